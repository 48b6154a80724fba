#!/usr/bin/env bash
# Builds libwlsqm_hip.so for gfx950 (cross-compiles without a GPU).  In-tree output:
#   python-wlsqm_amd/wlsqm/_lib/libwlsqm_hip.so
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
OUT="$HERE/wlsqm/_lib"
OBJ="$HERE/build"
mkdir -p "$OUT" "$OBJ"
FLAGS=(--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fopenmp -I"$HERE/../include" -I"$HERE/csrc" -Wall -Wno-unused-function)
pids=()
for f in fit_tile_even fit_tile_gather fit_tile_big fit_tile api expert conds interp fit_lane fit_tilek fit_wave fit_moment solve_many knn fit_rows; do
  src="$HERE/csrc/$f.hip"; obj="$OBJ/$f.o"
  if [[ ! -f "$obj" || "$src" -nt "$obj" || "$HERE/csrc/wlsqm_kernels.hpp" -nt "$obj" || "$HERE/csrc/wlsqm_internal.hpp" -nt "$obj" || "$HERE/csrc/wlsqm_interp.hpp" -nt "$obj" || "$HERE/csrc/hostio.hpp" -nt "$obj" || "$HERE/csrc/wlsqm_moments.hpp" -nt "$obj" || "$HERE/csrc/wlsqm_tile1.hpp" -nt "$obj" || "$HERE/csrc/wlsqm_tile.hpp" -nt "$obj" || "$HERE/../include/wlsqm_hip.h" -nt "$obj" ]]; then
    "$HIPCC" "${FLAGS[@]}" -c "$src" -o "$obj" &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [[ -n "$p" ]] && wait "$p"; done
"$HIPCC" --offload-arch=gfx950 -shared -fPIC -fopenmp -o "$OUT/libwlsqm_hip.so" "$OBJ"/api.o "$OBJ"/expert.o "$OBJ"/conds.o "$OBJ"/interp.o "$OBJ"/fit_lane.o "$OBJ"/fit_tile.o "$OBJ"/fit_tilek.o "$OBJ"/fit_wave.o "$OBJ"/fit_moment.o "$OBJ"/solve_many.o "$OBJ"/knn.o "$OBJ"/fit_rows.o "$OBJ"/fit_tile_even.o "$OBJ"/fit_tile_gather.o "$OBJ"/fit_tile_big.o
echo "built $OUT/libwlsqm_hip.so"
