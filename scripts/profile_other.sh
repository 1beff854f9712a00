#!/bin/bash
# scripts/profile_other.sh <tag>  (on the GPU box, through gpurun): rocprofv3 evidence for the NON-headline factor kernels --
# k_conv_flat<BR<0>>, k_conv<BR<1>,*>, k_conv_flat<P3P3>, k_conv<*,gauss_newton>, k_conv<P2P2,nelder_mead> -- on
# scripts/other_factors.py (helix 10k Pose3 + MIT-like bearing-range graph): kernel trace, FETCH_SIZE / WRITE_SIZE (separate passes
# + copy8 calibration), SQ counters.  Summary -> gpurun_out/<tag>/<tag>_other_factors_trace.md
tag=${1:-r04}
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$tag; T=/tmp/prof_other_$tag; mkdir -p $O $T
cd /tmp && export TMPDIR=/tmp
export ROME_OTHER_QUICK=1
timeout 600 rocprofv3 --kernel-trace --stats -d $T/trace -o t -- python $R/scripts/other_factors.py > $T/trace.log 2>&1
[ -x $R/scripts/ubench/copy8 ] || hipcc --offload-arch=gfx950 -O3 -o $R/scripts/ubench/copy8 $R/scripts/ubench/copy8.hip
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --pmc $c --kernel-trace -d $T/cal_$c -o cal -- $R/scripts/ubench/copy8 > $T/cal_$c.log 2>&1
  timeout 600 rocprofv3 --pmc $c --kernel-trace -d $T/pmc_$c -o p -- python $R/scripts/other_factors.py > $T/pmc_$c.log 2>&1
done
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU --kernel-trace -d $T/sq -o s -- python $R/scripts/other_factors.py > $T/sq.log 2>&1
grep -E "Pose3Pose3|BearingRange" $T/trace.log > $O/${tag}_other_factors_under_trace.txt
python3 - <<PY
import sqlite3, glob, re
T, O, tag = "$T", "$O", "$tag"
def q(pattern, sql):
    out = []
    for d in sorted(glob.glob(pattern, recursive=True)):
        out += sqlite3.connect(d).execute(sql).fetchall()
    return out
dur = {n: (c, a, v, s, l, sc) for n, c, a, v, s, l, sc in q(T + "/trace/**/*_results.db",
      "select name, count(*), avg(duration), max(vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size) from kernels group by name")}
def ctr(dirn):
    out = {}
    for n, cn, k, a in q(T + "/" + dirn + "/**/*_results.db", "select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name"):
        out.setdefault(n, {})[cn] = a
    return out
cal_f = [a for n, cn, k, a in q(T + "/cal_FETCH_SIZE/**/*_results.db", "select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name") if "copy" in n][0]
scale = 2.0 if cal_f < 0.75 * 2097152 else 1.0
F, W, S = ctr("pmc_FETCH_SIZE"), ctr("pmc_WRITE_SIZE"), ctr("sq")
# workload sizes of scripts/other_factors.py: helix 23991 rows (23990 relative + 1 prior), MIT-like 5978 sightings; N = 100
def alg(name):
    if "k_sweep_fused" in name:   # the whole MIT-like sweep in one launch: 16158 relative + 1 prior p2p2 rows, 5978 + 5978 sightings
        return 16158 * 100 * 48 + 2400 + 5978 * 100 * 64 + 5978 * 100 * 40, 16159 + 2 * 5978
    m = re.search(r"k_conv(_flat)?<rome::(P2P2|P3P3|BR<(\d)>), *(\w+)", name)
    if not m: return None
    flat, fam, d, sv = m.group(1), m.group(2), m.group(3), m.group(4)
    start = sv in ("2", "3")                       # Nelder-Mead / Gauss-Newton read the start points; closed form (0) of unique-root factors does not
    if fam == "P3P3": return 23990 * 100 * (144 if start else 96) + 100 * 48, 23991
    if fam == "BR<0>": return 5978 * 100 * (56 if start else 40), 5978
    if fam == "BR<1>": return 5978 * 100 * 64, 5978
    return None
lines = ["Non-headline factor kernels, N = 100, scripts/other_factors.py under rocprofv3 (ROME_OTHER_QUICK=1): kernel-trace average duration; HBM bytes per",
         "launch from --pmc FETCH_SIZE (x%.0f, gfx950 correction calibrated on scripts/ubench/copy8 in the same run) and --pmc WRITE_SIZE (separate passes);" % scale,
         "SQ counters from a third pass.  alg = algorithmic bytes per launch (SURVEY 8(d); unique-root closed form / Newton do not read the start points).",
         "solver template argument: 0 closed form (= NEWTON without status), 2 Nelder-Mead, 3 Gauss-Newton; k_conv_flat = packed unique-root sweep (closed form or Gauss-Newton).", "",
         "| kernel | launches | avg µs | vgpr | scratch | rows | alg MB | alg GB/s | frac of 8 TB/s | HBM MB (PMC) | HBM GB/s | VALU/wave | VALU busy | FP64-VALU note |",
         "|---|---|---|---|---|---|---|---|---|---|---|---|---|---|"]
for n in sorted(dur, key=lambda k: -dur[k][0] * dur[k][1]):
    if "rome::k_conv" not in n and "rome::k_sweep_fused" not in n: continue
    c, a, v, s, l, sc = dur[n]
    ab = alg(n)
    f = F.get(n, {}).get("FETCH_SIZE"); w = W.get(n, {}).get("WRITE_SIZE")
    hb = (f * 1024 * scale + w * 1024) if (f is not None and w is not None) else None
    sq = S.get(n, {})
    vw = sq["SQ_INSTS_VALU"] / sq["SQ_WAVES"] if sq.get("SQ_WAVES") else None
    busy = 4.0 * sq["SQ_ACTIVE_INST_VALU"] / (256 * 4 * 2.4 * a) if sq.get("SQ_ACTIVE_INST_VALU") else None
    note = ""
    if ", 2," in n or ", 3," in n:   # iterative solvers: FP64-VALU issue is the bound that applies (SURVEY 8(d) secondary roofline)
        note = "iterative: VALU-issue bound, busy fraction is the roofline that applies"
    lines.append("| \`%s\` | %d | %.2f | %s | %s | %s | %s | %s | %s | %s | %s | %s | %s | %s |" % (
        n.replace("void ", "")[:90], c, a / 1e3, v, sc, ab[1] if ab else "", "%.1f" % (ab[0] / 1e6) if ab else "",
        "%.0f" % (ab[0] / a) if ab else "", "%.3f" % (ab[0] / a / 8000) if ab else "", "%.1f" % (hb / 1e6) if hb else "",
        "%.0f" % (hb / a) if hb else "", "%.0f" % vw if vw else "", "%.2f" % busy if busy else "", note))
open(O + "/" + tag + "_other_factors_trace.md", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
rm -rf $T
