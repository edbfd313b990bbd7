mkdir -p gpurun_out/r05d
python -m pytest tests -m gpu -q > gpurun_out/r05d/t_all.log 2>&1
grep -E "^FAILED|passed|failed" gpurun_out/r05d/t_all.log | tail -8
