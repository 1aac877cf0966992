#!/usr/bin/env bash
# tools/kinfo.sh [extra hipcc flags] — registers, spills, LDS and occupancy of the gfx950 kernels, from the compiler's own
# resource remarks (no GPU needed).  Prints one line per kernel whose name matches $KINFO_FILTER (default: the per-step kernels).
set -euo pipefail
root="$(cd "$(dirname "$0")/.." && pwd)"
filter="${KINFO_FILTER:-k_force_list|k_density_list|k_build_list|k_rebuild|k_check|k_step}"
cd "$root/pi-sph-fluid_amd"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -fPIC -I../include -Icsrc --cuda-device-only \
    -c csrc/sph_kernels.hip -o /tmp/kinfo.$$.co -Rpass-analysis=kernel-resource-usage "$@" 2> /tmp/kinfo.$$.txt || { cat /tmp/kinfo.$$.txt; exit 1; }
python3 - "$filter" /tmp/kinfo.$$.txt <<'PY'
import re, sys, subprocess
flt, path = re.compile(sys.argv[1]), sys.argv[2]
cur = None
rows = []
for line in open(path):
    m = re.search(r"remark: (.*?) *\[-Rpass", line)
    if not m: continue
    t = m.group(1)
    if t.startswith("Function Name:"):
        name = t.split(":", 1)[1].strip()
        try: name = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip().split("(")[0]
        except Exception: pass
        cur = {"name": name.replace("void sph::", "")}
        rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
print("%-44s %5s %5s %7s %6s %6s %4s" % ("kernel", "VGPR", "SGPR", "scratch", "vspill", "LDS", "occ"))
for r in rows:
    if flt.search(r["name"]):
        print("%-44s %5s %5s %7s %6s %6s %4s" % (r["name"][:44], r.get("VGPRs"), r.get("TotalSGPRs"), r.get("ScratchSize [bytes/lane]"),
              r.get("VGPRs Spill"), r.get("LDS Size [bytes/block]"), r.get("Occupancy [waves/SIMD]")))
PY
rm -f /tmp/kinfo.$$.co /tmp/kinfo.$$.txt
