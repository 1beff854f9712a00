#!/bin/bash
# static instruction census of one kernel (device-only assembly): scripts/isa_count.sh <file.hip> <mangled-name-substring>
set -e
src=$1; pat=$2
out=/tmp/isa_$(basename $src .hip).s
[ -n "$REUSE" -a -f $out ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-pass-failed -S --cuda-device-only -o $out $src 2>/dev/null
awk -v pat="$pat" '
  $0 ~ ("^_Z[^:]*" pat "[^:]*:") {on=1; name=$1; next}
  
  on && /^\.Lfunc_end/ {on=0; printf "%s\n  valu %d (f64 %d, f32 %d, cvt %d, dpp %d, cndmask %d, mad_u64 %d)  salu %d  vmem %d  lds %d  branches %d  s_waitcnt %d  total %d\n", name, v, f64, f32, cvt, dpp, cnd, mad, s, vm, lds, br, wc, tot; v=f64=f32=cvt=dpp=cnd=mad=s=vm=lds=br=wc=tot=0}
  on && /^[ \t]+[a-z]/ {
    tot++
    if ($1 ~ /^v_/) { v++; if ($1 ~ /_f64/) f64++; if ($1 ~ /_f32/ && $1 !~ /cvt/) f32++; if ($1 ~ /cvt/) cvt++; if ($0 ~ /dpp|row_|quad_perm/) dpp++; if ($1 ~ /cndmask/) cnd++; if ($1 ~ /mad_u64/) mad++ }
    else if ($1 ~ /^s_waitcnt/) wc++
    else if ($1 ~ /^s_(c)?branch/) br++
    else if ($1 ~ /^s_/) s++
    else if ($1 ~ /^(global|buffer|flat|scratch)_/) vm++
    else if ($1 ~ /^ds_/) lds++
  }' $out
