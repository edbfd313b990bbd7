#!/bin/bash
# Ablation timing of the 384 -> 128 producer / consumer row GEMM: one line per -DK3_DBG=<bits> build (scripts/build_variant.sh
# k3_<bits> row_gemm_k384.hip -DK3_DBG=<bits>), all inside one GPU call.
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do
for v in druggen_amd/lib/variants/k3_*.so; do
  DG_LIB=$PWD/$v python scripts/gemm_variants.py 2>&1 | grep "K=384"
done
done
