#!/bin/bash
# Ablation timing of the fused float32 attention-half forward: one line per -DAH_DBG=<bits> build (scripts/build_variant.sh
# ah_<bits> attn_half_f32.hip -DAH_DBG=<bits>), all inside one GPU call.
cd ${GRAFT_REPO_ROOT:-/root/repo}
for v in druggen_amd/lib/variants/ah_*.so; do
  echo "$(basename $v) $(DG_LIB=$PWD/$v python scripts/half_f32_probe.py 2>&1 | grep 'keep' | sed 's/three launches.*//' | tr '\n' ' ')"
done
