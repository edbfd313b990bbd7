#!/bin/bash
# Developer A/B builds: scripts/build_variant.sh <name> <source.hip> <extra hipcc flags...> -> druggen_amd/lib/variants/<name>.so
# (the shipped objects of the other translation units + a re-compiled <source>; load with DG_LIB=<path>).
set -e
name=$1; src=$2; shift 2
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $R/druggen_amd/lib/variants
obj=$R/druggen_amd/lib/variants/$name.$(basename $src).o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast "$@" -c $R/druggen_amd/csrc/$src -o $obj
others=$(ls $R/druggen_amd/lib/*.hip.o | grep -v "/$(basename $src).o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/druggen_amd/lib/variants/$name.so $others $obj
echo $R/druggen_amd/lib/variants/$name.so
