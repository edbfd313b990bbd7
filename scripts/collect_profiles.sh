#!/bin/bash
# Round-end evidence: bench lines of every BASELINE config, rocprofv3 kernel-trace summaries of the headline (c2, fp32)
# and configs[2] (c3, bf16) steps, FETCH_SIZE / WRITE_SIZE passes of both (-> profiles/traffic.json).  Everything lands in
# gpurun_out/<tag>/ ; copy what should be judged into profiles/.
#   gpurun -- 'scripts/collect_profiles.sh r05 <commit>'
set -u
tag=${1:-r05}; commit=${2:-unknown}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py"
# ---- bench lines
$B --steps 20 --warmup 5 2>$O/bench_c2.err | tail -1 > $O/bench_c2.json
cp $R/gpurun_out/bench_detail_c2_f32.json $O/bench_detail_c2_f32.json 2>/dev/null
$B --config c3 --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_c3.json
$B --config c4 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_c4.json
$B --config c5 --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_c5.json
$B --config c5 --dtype bf16 --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_c5_bf16.json
# ---- kernel traces
for cfg in c2 c3; do
  rocprofv3 --kernel-trace --stats -d $O/trace_$cfg -o t -- $B --config $cfg --steps 5 --warmup 2 --no-cpu-baseline --no-extra --eager > $O/trace_$cfg.log 2>&1
  python $R/scripts/rocprof_summary.py $(find $O/trace_$cfg -name "*.db" | head -1) 60 > $O/${cfg}_kernel_stats.txt
  # (the traced command runs 5 + 2 steps and bench.py's instrumented pre-pass of 3: 10 steps)
  [ $cfg = c2 ] && python $R/scripts/step_timeline.py $(find $O/trace_$cfg -name "*.db" | head -1) 10 > $O/small_launches_c2.txt
done
# ---- HBM traffic (separate passes per counter)
for cfg in c2 c3; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $ctr -d $O/pmc_${cfg}_$ctr -o p --output-format csv -- $B --config $cfg --steps 3 --warmup 1 --no-cpu-baseline --no-extra --eager > $O/pmc_${cfg}_$ctr.log 2>&1
  done
done
cd $R
# ---- SQ counters of the kernels this round worked on (one rocprofv3 pass per counter group, kernel trace only)
scripts/pmc_run.sh $tag/pmc_half attn_half -- python $R/scripts/half_probe.py 2048 45 > $O/pmc_attn_half_fwd.txt 2>&1
scripts/pmc_run.sh $tag/pmc_hb attn_half_bwd -- python $R/scripts/half_bwd_probe.py 2048 45 > $O/pmc_attn_half_bwd.txt 2>&1
scripts/pmc_run.sh $tag/pmc_h32 attn_half_f32_fwd -- python $R/scripts/half_f32_probe.py > $O/pmc_attn_half_f32_fwd.txt 2>&1
scripts/pmc_run.sh $tag/pmc_h32b attn_half_f32_bwd1 -- python $R/scripts/half_f32_bwd_check.py > $O/pmc_attn_half_f32_bwd1.txt 2>&1
scripts/pmc_run.sh $tag/pmc_wg wgrad_stream -- python $R/scripts/wgrad_probe.py > $O/pmc_wgrad.txt 2>&1
scripts/pmc_run.sh $tag/pmc_hid row_gemm -- python $R/scripts/h16_probe.py > $O/pmc_hidden_storage_gemms.txt 2>&1
python scripts/h16_probe.py > $O/hidden_storage_probe.txt 2>&1
python scripts/lnb_probe.py > $O/lnb_probe.txt 2>&1
rm -rf $O/pmc_half $O/pmc_hb $O/pmc_h32 $O/pmc_h32b $O/pmc_wg $O/pmc_hid $O/*.p[0-9].log
cp profiles/traffic.json $O/traffic.json 2>/dev/null || echo '{"records": []}' > $O/traffic.json
python scripts/pmc_traffic.py $(find $O/pmc_c2_FETCH_SIZE -name "*counter_collection.csv") $(find $O/pmc_c2_WRITE_SIZE -name "*counter_collection.csv") c2 f32 256 $commit $O/traffic.json 150e6 > $O/traffic_c2.txt
python scripts/pmc_traffic.py $(find $O/pmc_c3_FETCH_SIZE -name "*counter_collection.csv") $(find $O/pmc_c3_WRITE_SIZE -name "*counter_collection.csv") c3 bf16 2048 $commit $O/traffic.json 600e6 > $O/traffic_c3.txt
# the raw traces / counter dumps are large: keep only the summaries
rm -rf $O/trace_c2 $O/trace_c3
find $O -name '*counter_collection.csv' | while read f; do gzip -9 "$f"; done; find $O -name '*.csv' -delete
for f in bench_c2 bench_c3 bench_c4 bench_c5 bench_c5_bf16; do
  python - <<PY
import json
try:
    d = json.load(open("$O/$f.json"))
    print("$f", round(d["value"]), "mol/s", round(d["ms_per_step"], 2), "ms", d["roofline"].get("kernel"), round(d["roofline"].get("frac", 0), 3),
          "peak GB", round(d.get("peak_memory_GB", 0), 1), "bf16_configs2", round(d.get("bf16_configs2", {}).get("value", 0)))
except Exception as e:
    print("$f", "FAILED", e)
PY
done
