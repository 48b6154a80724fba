#!/usr/bin/env bash
# usage: bash tools/isa_stats.sh [file.hip ...] > profiles/isa_r03.txt      (no GPU needed: hipcc cross-compiles gfx950)
# Per kernel of the given translation units (default: the ones that hold the BASELINE configs' kernels): registers the
# compiler allocated, registers spilled, scratch bytes per lane and static LDS, from the code object metadata
# (hipcc -S --cuda-device-only: .vgpr_count / .vgpr_spill_count / .private_segment_fixed_size / .group_segment_fixed_size).
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
FILES=("$@")
if (( ${#FILES[@]} == 0 )); then FILES=(fit_tile.hip fit_ring.hip fit_ring_gather.hip solve_op.hip fit_chunk.hip fit_strict.hip fit_rows.hip fit_tilek.hip); fi
FILTER="${ISA_FILTER:-.}"
# the record is stamped with the sources it was made from (the lines of libwlsqm_hip.manifest, hashed): tests/test_abi_and_host.py fails
# when profiles/isa_rNN.txt of the current round and the tree disagree
echo "# sources sha256: $(cd "$HERE/python-wlsqm_amd" && export LC_ALL=C && sha256sum csrc/*.hip csrc/*.hpp ../include/*.h | sed 's#\.\./include#include#' | sha256sum | cut -d' ' -f1)"
for f in "${FILES[@]}"; do
  src="$HERE/python-wlsqm_amd/csrc/$f"
  out="$(mktemp /tmp/isa_XXXX.s)"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I"$HERE/include" -I"$HERE/python-wlsqm_amd/csrc" -S --cuda-device-only -o "$out" "$src" 2>/dev/null
  echo "== $f"
  python3 - "$out" "$FILTER" <<'PY'
import re, subprocess, sys
txt = open(sys.argv[1]).read()
flt = re.compile(sys.argv[2])
# amdhsa.kernels metadata: one YAML-ish block per kernel
for blk in txt.split("  - .agpr_count:")[1:]:
    get = lambda k: (re.search(r"\.%s:\s+(\S+)" % k, blk) or [None, "?"])[1]
    name = get("name")
    try:
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    except Exception:
        dem = name
    dem = re.sub(r"\(wlsqm::\w+(, .*)?\)$", "", dem).replace("void wlsqm::", "")
    if not flt.search(dem):
        continue
    print("%-92s vgpr %4s  spilled %4s  scratch %5s B  lds %6s B  sgpr %4s" % (dem[:92], get("vgpr_count"), get("vgpr_spill_count"),
          get("private_segment_fixed_size"), get("group_segment_fixed_size"), get("sgpr_count")))
PY
  rm -f "$out"
done
