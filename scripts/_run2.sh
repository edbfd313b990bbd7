python -m pytest tests/test_hip_kernels.py -m gpu -q -x -k "wgrad or weight_gradient or riding" 2>&1 | tail -4
for m in 2 3 2 3; do DG_WGRAD128_PRODUCTS=$m python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print(d['value'], d['ms_per_step'], d['roofline_attention']['kernel'], d['roofline_attention']['frac'], '$m')"; done
python scripts/parity_report.py dh16 2>&1 | tail -13
