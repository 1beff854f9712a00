"""vgpr / sgpr / spill / scratch of the kernels in a device assembly file (hipcc -S --cuda-device-only) whose mangled name matches a pattern
    python scripts/kmeta.py file.s [regex]"""
import re
import sys

pat = re.compile(sys.argv[2] if len(sys.argv) > 2 else ".")
cur = {}
keys = (".vgpr_count", ".agpr_count", ".sgpr_count", ".vgpr_spill_count", ".private_segment_fixed_size", ".group_segment_fixed_size")
for line in open(sys.argv[1], errors="replace"):
    s = line.strip()
    if s.startswith("- .agpr_count") or s.startswith("- .args"):
        cur = {}
        s = s[2:]
    for k in keys:
        if s.startswith(k + ":"):
            cur[k] = s.split(":")[1].strip()
    if s.startswith(".name:"):
        cur["name"] = s.split(":", 1)[1].strip()
    if s.startswith(".wavefront_size") and cur.get("name") and pat.search(cur["name"]):
        print(cur["name"], " ".join("%s=%s" % (k[1:], cur.get(k)) for k in keys))
