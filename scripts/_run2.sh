python -m pytest tests/test_hip_bf16.py -m gpu -q -x 2>&1 | tail -3
python scripts/ffn_probe.py 2>&1 | tail -1
python bench.py --config c3 --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print('c3', d['value'], d['ms_per_step'])"
