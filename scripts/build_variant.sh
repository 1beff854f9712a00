#!/bin/bash
# scripts/build_variant.sh <name> [UNIT=rome_kde] [extra hipcc flags...]  -> scripts/ubench/lib_<name>.so   (kernel A/B builds)
# only ONE unit (default rome_kernels; first argument of the form UNIT=<file stem> selects another) is recompiled with the flags; the
# other units come from rome.jl_amd/build (python rome.jl_amd/_build.py first).  Run with ROME_MI355_LIB=scripts/ubench/lib_<name>.so
R=/root/repo; n=$1; shift
u=rome_kernels
case "$1" in UNIT=*) u=${1#UNIT=}; shift;; esac
rm -f $R/scripts/ubench/lib_$n.so
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-pass-failed "$@" -c $R/rome.jl_amd/csrc/$u.hip -o /tmp/variant_$n.o 2>&1 | grep -E "error" | head
hipcc --offload-arch=gfx950 -shared -fPIC -o $R/scripts/ubench/lib_$n.so /tmp/variant_$n.o $(ls $R/rome.jl_amd/build/*.o | grep -v "/$u.o")
test -f $R/scripts/ubench/lib_$n.so
