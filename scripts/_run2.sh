python -m pytest tests/test_hip_scale.py -m gpu -q -x -k "three_graph" 2>&1 | tail -25
