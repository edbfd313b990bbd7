R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05c; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/trace_c3 -o t -- python $R/bench.py --config c3 --steps 5 --warmup 2 --no-cpu-baseline --no-extra > $O/trace_c3.log 2>&1
python $R/scripts/rocprof_summary.py $(find $O/trace_c3 -name "*.db" | head -1) 70 > $O/c3_kernel_stats.txt
rm -rf $O/trace_c3
sed -n 1,75p $O/c3_kernel_stats.txt | cut -c1-150
