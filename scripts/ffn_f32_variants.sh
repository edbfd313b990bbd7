#!/bin/bash
# Ablation timing of the fused float32 feed-forward forward (profiles/r06_ffn_fused_ablation.txt): one line per -DFF_DBG=<bits>
# build, all inside one GPU call.  Build first, on the CPU box:
#   for v in 0 3 7 19 35 67 131 259 515 243 1015 4 16 32 256 240 496; do scripts/build_variant.sh ffv$v ffn_fused_f32.hip -DFF_DBG=$v; done
cd ${GRAFT_REPO_ROOT:-/root/repo}
python scripts/ffn_f32_variant_time.py 2>&1 | tail -1
for so in druggen_amd/lib/variants/ffv*.so; do
  DG_LIB=$PWD/$so timeout 200 python scripts/ffn_f32_variant_time.py 2>&1 | tail -1
done
