python -m pytest tests/test_hip_bf16.py tests/test_hip_kernels.py -m gpu -q -k "bf16 or ln_ or layernorm or LN" 2>&1 | tail -4
python bench.py --config c3 --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print('c3', d['value'], d['ms_per_step'])"
python bench.py --config c3 --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print('c3', d['value'], d['ms_per_step'])"
