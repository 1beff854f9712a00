#!/bin/bash
# scripts/build_variant.sh <name> [extra hipcc flags...]  -> scripts/ubench/lib_<name>.so   (kernel A/B builds)
# only rome_kernels.hip is recompiled with the flags; the other units come from rome.jl_amd/build (python rome.jl_amd/_build.py first)
R=/root/repo; n=$1; shift
rm -f $R/scripts/ubench/lib_$n.so
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-pass-failed "$@" -c $R/rome.jl_amd/csrc/rome_kernels.hip -o /tmp/variant_$n.o 2>&1 | grep -E "error" | head
hipcc --offload-arch=gfx950 -shared -fPIC -o $R/scripts/ubench/lib_$n.so /tmp/variant_$n.o $(ls $R/rome.jl_amd/build/*.o | grep -v rome_kernels.o)
test -f $R/scripts/ubench/lib_$n.so
