#!/usr/bin/env bash
# usage (GPU box, repo root): bash tools/with_variant.sh UNIT "-DFLAG=1 ..." COMMAND [ARGS...]
# Runs COMMAND with csrc/UNIT.hip rebuilt under the given flags and relinked into the library; the library as built comes back
# whatever happens (e.g. a profile of a compile-time variant: ... bash tools/profile_config.sh TAG C5 1000000 1404000000).
set -euo pipefail
UNIT="$1"; FLAGS="$2"; shift 2
PKG=python-wlsqm_amd
ORIG="$(mktemp /tmp/lib_orig_XXXXXX.so)"; VAR="$(mktemp /tmp/unit_var_XXXXXX.o)"
cp $PKG/wlsqm/_lib/libwlsqm_hip.so "$ORIG"
trap 'cp "$ORIG" $PKG/wlsqm/_lib/libwlsqm_hip.so; rm -f "$ORIG" "$VAR"' EXIT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fopenmp -I include -I $PKG/csrc $FLAGS -c $PKG/csrc/$UNIT.hip -o $VAR
objs=(); for o in $PKG/build/*.o; do [[ "$(basename $o)" == "$UNIT.o" ]] && objs+=("$VAR") || objs+=("$o"); done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fopenmp -o $PKG/wlsqm/_lib/libwlsqm_hip.so "${objs[@]}"
"$@"
