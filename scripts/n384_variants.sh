#!/bin/bash
# Ablation timing of the 128 -> 384 producer / consumer row GEMM: one line per -DN3_DBG=<bits> build (scripts/build_variant.sh
# n3_<bits> row_gemm_n384.hip -DN3_DBG=<bits>), all inside one GPU call.
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do
for v in druggen_amd/lib/variants/n3_*.so; do
  DG_LIB=$PWD/$v python scripts/gemm_variants.py 2>&1 | grep "N=384"
done
done
