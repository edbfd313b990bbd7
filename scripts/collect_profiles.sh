#!/bin/bash
# Round-end evidence in ONE gpurun call: bench lines of every BASELINE config, the rocprofv3 kernel-trace summary of the headline
# step (c2, fp32) and of configs[2] (c3, bf16), FETCH_SIZE / WRITE_SIZE passes of the headline (-> profiles/traffic.json), SQ
# counters of the fused feed-forward kernel, the parity table per hidden-storage mode.  Everything lands in gpurun_out/<tag>/ ;
# copy what should be judged into profiles/.
#   gpurun -- 'scripts/collect_profiles.sh r06 <commit>'
set -u
tag=${1:-r06}; commit=${2:-unknown}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py"
# ---- bench lines
$B 2>$O/bench_c2.err | tail -1 > $O/bench_c2.json
cp $R/gpurun_out/bench_detail_c2_f32.json $O/bench_detail_c2_f32.json 2>/dev/null
$B --config c3 --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_c3.json
cp $R/gpurun_out/bench_detail_c3_bf16.json $O/bench_detail_c3_bf16.json 2>/dev/null      # (before the traced runs overwrite it: events cost ~100 us per launch under the tracer)
$B --config c4 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_c4.json
$B --config c5 --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_c5.json
$B --config c5 --batch 512 --steps 3 --warmup 1 --no-cpu-baseline 2>$O/bench_c5_b512.err | tail -1 > $O/bench_c5_b512.json
$B --config c5 --dtype bf16 --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_c5_bf16.json
# ---- kernel traces
for cfg in c2 c3; do
  rocprofv3 --kernel-trace --stats -d $O/trace_$cfg -o t -- $B --config $cfg --steps 5 --warmup 2 --no-cpu-baseline --no-extra --eager > $O/trace_$cfg.log 2>&1
  python $R/scripts/rocprof_summary.py $(find $O/trace_$cfg -name "*.db" | head -1) 60 > $O/${cfg}_kernel_stats.txt
  # (the traced command runs 5 + 2 steps and bench.py's instrumented pre-pass of 3: 10 steps)
  [ $cfg = c2 ] && python $R/scripts/step_timeline.py $(find $O/trace_$cfg -name "*.db" | head -1) 10 > $O/small_launches_c2.txt
done
# ---- HBM traffic (separate passes per counter)
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $ctr -d $O/pmc_c2_$ctr -o p --output-format csv -- $B --config c2 --steps 3 --warmup 1 --no-cpu-baseline --no-extra --eager > $O/pmc_c2_$ctr.log 2>&1
done
cd $R
# ---- SQ counters of the kernel this round built (one rocprofv3 pass per counter group, kernel trace only)
scripts/pmc_run.sh $tag/pmc_ffn ffn_fused -- python $R/scripts/ffn_f32_probe.py time > $O/pmc_ffn_fused_f32.txt 2>&1
python scripts/ffn_f32_probe.py all > $O/ffn_f32_probe.txt 2>&1
python scripts/parity_report.py dh16 f32 dh24 > $O/parity_by_hidden_storage.txt 2>&1
rm -rf $O/pmc_ffn $O/*.p[0-9].log
cp profiles/traffic.json $O/traffic.json 2>/dev/null || echo '{"records": []}' > $O/traffic.json
python scripts/pmc_traffic.py $(find $O/pmc_c2_FETCH_SIZE -name "*counter_collection.csv") $(find $O/pmc_c2_WRITE_SIZE -name "*counter_collection.csv") c2 f32 256 $commit $O/traffic.json 150e6 > $O/traffic_c2.txt
# the raw traces / counter dumps are large: keep only the summaries
rm -rf $O/trace_c2 $O/trace_c3
find $O -name '*counter_collection.csv' | while read f; do gzip -9 "$f"; done; find $O -name '*.csv' -delete
for f in bench_c2 bench_c3 bench_c4 bench_c5 bench_c5_b512 bench_c5_bf16; do
  python - <<PY
import json
try:
    d = json.load(open("$O/$f.json"))
    print("$f", round(d["value"]), "mol/s", round(d["ms_per_step"], 2), "ms", d["roofline"].get("kernel"), d["roofline"].get("bound"), round(d["roofline"].get("frac", 0), 3),
          "peak GB", round(d.get("peak_memory_GB", 0), 1), "bf16_configs2", round(d.get("bf16_configs2", {}).get("value", 0)))
except Exception as e:
    print("$f", "FAILED", e)
PY
done
