#!/usr/bin/env bash
# Builds libwlsqm_hip.so for gfx950 (cross-compiles without a GPU).  In-tree output:
#   python-wlsqm_amd/wlsqm/_lib/libwlsqm_hip.so   (+ libwlsqm_hip.manifest: sha256 of every source it was built from)
#
# Incremental by default: every object is compiled with -MMD, and a translation unit is rebuilt when its .hip or ANY header
# its dependency file names (csrc/*.hpp, include/*.h) is newer than the object, when the dependency file is missing, or
# when the compiler flags changed.  WLSQM_FORCE_REBUILD=1 recompiles everything regardless.  The last line of the output
# says how many translation units were compiled ("compiled N of M").
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
OUT="$HERE/wlsqm/_lib"
OBJ="$HERE/build"
mkdir -p "$OUT" "$OBJ"
FLAGS=(--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fopenmp -I"$HERE/../include" -I"$HERE/csrc" -Wall -Wno-unused-function)
FORCE="${WLSQM_FORCE_REBUILD:-0}"
JOBS="${WLSQM_BUILD_JOBS:-8}"

# the flags are part of the build's identity
flagsig="$(printf '%s\n' "$HIPCC" "${FLAGS[@]}" | sha256sum | cut -d' ' -f1)"
if [[ ! -f "$OBJ/.flags" || "$(cat "$OBJ/.flags")" != "$flagsig" ]]; then FORCE=1; fi

stale() {   # stale <obj> <dep> <src>
  local obj="$1" dep="$2" src="$3" f
  [[ "$FORCE" == "1" || ! -f "$obj" || ! -f "$dep" || "$src" -nt "$obj" ]] && return 0
  # every prerequisite listed by the compiler (-MMD); a header that disappeared also forces a rebuild
  for f in $(sed -e 's/^[^:]*://' -e 's/\\$//' "$dep"); do
    [[ ! -e "$f" || "$f" -nt "$obj" ]] && return 0
  done
  return 1
}

units=()
for src in "$HERE"/csrc/*.hip; do units+=("$(basename "${src%.hip}")"); done
pids=(); compiled=0
for f in "${units[@]}"; do
  src="$HERE/csrc/$f.hip"; obj="$OBJ/$f.o"; dep="$OBJ/$f.d"
  if stale "$obj" "$dep" "$src"; then
    while (( $(jobs -rp | wc -l) >= JOBS )); do wait -n; done
    "$HIPCC" "${FLAGS[@]}" -MMD -MF "$dep" -c "$src" -o "$obj" &
    pids+=($!); compiled=$((compiled + 1))
  fi
done
for p in "${pids[@]:-}"; do [[ -n "$p" ]] && wait "$p"; done
# objects of translation units that no longer exist must not be linked: the library is relinked whenever the LIST of units
# changed (a removed or renamed .hip compiles nothing, but the old .so would still hold its object)
objs=(); for f in "${units[@]}"; do objs+=("$OBJ/$f.o"); done
unitsig="$(printf '%s\n' "${units[@]}" | sort | sha256sum | cut -d' ' -f1)"
relink=0
if [[ ! -f "$OBJ/.units" || "$(cat "$OBJ/.units")" != "$unitsig" ]]; then relink=1; fi
if (( compiled > 0 )) || (( relink )) || [[ ! -f "$OUT/libwlsqm_hip.so" ]]; then
  rm -f "$OUT/libwlsqm_hip.manifest"
  "$HIPCC" --offload-arch=gfx950 -shared -fPIC -fopenmp -o "$OUT/libwlsqm_hip.so" "${objs[@]}"
fi
echo "$unitsig" > "$OBJ/.units"
echo "$flagsig" > "$OBJ/.flags"
# manifest: what the library was built from (checked by wlsqm._binding.build_manifest_ok and tests/test_abi_and_host.py)
( cd "$HERE" && sha256sum csrc/*.hip csrc/*.hpp ../include/*.h | sed 's#\.\./include#include#' ) > "$OUT/libwlsqm_hip.manifest"
echo "built $OUT/libwlsqm_hip.so (compiled $compiled of ${#units[@]} translation units; force=$FORCE)"
