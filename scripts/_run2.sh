python -m pytest tests/test_hip_kernels.py -m gpu -q -k "hidden or fused_ffn or paired" 2>&1 | tail -3
timeout 300 python scripts/h16_probe.py 2>&1 | sed -n 2,3p
for m in 1 2 3; do python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print(d['value'], d['ms_per_step'])"; done
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
