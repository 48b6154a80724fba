#!/usr/bin/env bash
# Register / scratch / LDS use of every kernel of one translation unit (compiler remarks, gfx950):  tools/kres.sh csrc/<unit>.hip [-D...]
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")/../python-wlsqm_amd" && pwd)"
src="$1"; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I"$HERE/../include" -I"$HERE/csrc" --cuda-device-only -c "$HERE/$src" -o /dev/null \
  -Rpass-analysis=kernel-resource-usage "$@" 2>&1 | python3 -c '
import sys, re, subprocess
rows = []; cur = None
for line in sys.stdin:
    m = re.search(r"remark: [^ ]+ +(?:Function )?Name: (\S+)", line) or re.search(r"Name: (\S+)", line)
    if m:
        cur = {"name": m.group(1)}; rows.append(cur); continue
    m = re.search(r"(VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|VGPRs Spill|LDS Size \[bytes/block\]): (\d+)", line)
    if m and cur is not None: cur[m.group(1).split(" [")[0]] = int(m.group(2))
for r in rows:
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"\(wlsqm::KParams.*", "", name).replace("void wlsqm::", "")
    print("%-60s vgpr %3d agpr %3d scratch %4d spill %3d occ %d lds %6d" % (name, r.get("VGPRs", -1), r.get("AGPRs", -1), r.get("ScratchSize", -1), r.get("VGPRs Spill", -1), r.get("Occupancy", -1), r.get("LDS Size", -1)))
'
