#!/bin/bash
# A/B of the weight-gradient kernels inside ONE GPU call: symmetric (DG_WGRAD=sym) vs producer / consumer (default).
for R in 518400 1036800 11520 23040; do
  for mode in sym h3; do
    DG_WGRAD=$mode python scripts/wgrad_probe.py $R
  done
done
