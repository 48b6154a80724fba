#!/usr/bin/env bash
# usage (GPU box, repo root): bash tools/ab_cmd.sh UNIT "COMMAND" "-DFLAG=1" ...   — tools/ab_unit.sh with any timing COMMAND
# (e.g. "python3 tools/time_shape.py 2 3 1000000 25"): runs it on the library as built, then with csrc/UNIT.hip rebuilt under each flag set,
# then as built again; the library as built comes back whatever happens.
set -euo pipefail
UNIT="$1"; CMD="$2"; shift 2
PKG=python-wlsqm_amd
ORIG="$(mktemp /tmp/lib_orig_XXXXXX.so)"; VAR="$(mktemp /tmp/unit_var_XXXXXX.o)"
cp $PKG/wlsqm/_lib/libwlsqm_hip.so "$ORIG"
trap 'cp "$ORIG" $PKG/wlsqm/_lib/libwlsqm_hip.so; rm -f "$ORIG" "$VAR"' EXIT
echo "== as built"; bash -c "$CMD"
for flags in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fopenmp -I include -I $PKG/csrc $flags -c $PKG/csrc/$UNIT.hip -o $VAR 
  objs=(); for o in $PKG/build/*.o; do [[ "$(basename $o)" == "$UNIT.o" ]] && objs+=("$VAR") || objs+=("$o"); done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fopenmp -o $PKG/wlsqm/_lib/libwlsqm_hip.so "${objs[@]}"
  echo "== [$flags]"; bash -c "$CMD"
done
cp "$ORIG" $PKG/wlsqm/_lib/libwlsqm_hip.so
echo "== as built again"; bash -c "$CMD"
