python -m pytest tests/test_hip_kernels.py -m gpu -q -x -k "attention_half_forward" 2>&1 | tail -5
python -m pytest tests/test_hip_model.py -m gpu -q -x -k "golden or deep_variant" 2>&1 | tail -3
for m in n48 fused; do DG_ATTN_HALF_F32=$m python bench.py --config c5 --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print('c5 $m', d['value'], d['ms_per_step'])"; done
