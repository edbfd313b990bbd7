python -m pytest tests/test_hip_kernels.py -m gpu -q -x -k "hidden_fp16 or two_product or fused_ffn or riding or paired" 2>&1 | tail -5
timeout 300 python scripts/h16_probe.py 2>&1 | sed -n 2,9p
for m in split f32 split f32; do DG_HIDDEN_FWD=$m python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print(d['value'], d['ms_per_step'], d['config']['hidden_storage'], '$m')"; done
python scripts/parity_report.py dh16 2>&1 | tail -13
