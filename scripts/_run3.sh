R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05c; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/trace_c2 -o t -- python $R/bench.py --config c2 --steps 5 --warmup 2 --no-cpu-baseline --no-extra > $O/trace_c2.log 2>&1
python $R/scripts/rocprof_summary.py $(find $O/trace_c2 -name "*.db" | head -1) 60 > $O/c2_kernel_stats.txt
python $R/scripts/step_timeline.py $(find $O/trace_c2 -name "*.db" | head -1) 10 > $O/small_launches_c2.txt
rm -rf $O/trace_c2
head -45 $O/c2_kernel_stats.txt | cut -c1-170
