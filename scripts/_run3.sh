python -m pytest tests -m gpu -q 2>&1 | tail -8 > gpurun_out/test_gpu.log
bash scripts/collect_profiles.sh r05 c1ac385 > gpurun_out/collect.log 2>&1
tail -8 gpurun_out/collect.log
cat gpurun_out/test_gpu.log
