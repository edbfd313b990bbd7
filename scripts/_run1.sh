python -m pytest tests/test_hip_kernels.py -m gpu -q -k "hidden_fp16 or single_product or fused_ffn" 2>&1 | tail -6
timeout 300 python scripts/h16_probe.py 2>&1 | sed -n 2,4p
for m in dh16 f32 dh16 f32; do DG_HIDDEN=$m python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print(d['value'], d['ms_per_step'], d['config']['hidden_storage'])"; done
DG_DH_PRODUCTS=3 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print(d['value'], d['ms_per_step'], 'dh16 3 products')"
python scripts/parity_report.py dh16 2>&1 | tail -14
