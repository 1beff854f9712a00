#!/bin/bash
# scripts/build_variant.sh <name> [extra hipcc flags...]  -> scripts/ubench/lib_<name>.so   (kernel A/B builds)
R=/root/repo; n=$1; shift
rm -f $R/scripts/ubench/lib_$n.so
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-pass-failed "$@" -o $R/scripts/ubench/lib_$n.so $R/rome.jl_amd/csrc/rome_kernels.hip $R/rome.jl_amd/csrc/rome_parametric.hip $R/rome.jl_amd/csrc/rome_product.hip $R/rome.jl_amd/csrc/rome_kde.hip $R/rome.jl_amd/csrc/rome_capi.hip 2>&1 | grep -E "error" | head
test -f $R/scripts/ubench/lib_$n.so
