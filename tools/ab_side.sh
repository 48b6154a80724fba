#!/usr/bin/env bash
# usage (GPU box, repo root): bash tools/ab_side.sh UNIT SIDE_KEY "-DFLAG=1" ...   — tools/ab_unit.sh for a SIDE line of bench.py
# (WLSQM_BENCH_SIDE=SIDE_KEY): times the library as built, then rebuilds csrc/UNIT.hip with each flag set, relinks, times again, restores.
set -euo pipefail
UNIT="$1"; KEY="$2"; shift 2
PKG=python-wlsqm_amd
run() { for i in 1 2 3; do WLSQM_BENCH_SIDE="$KEY" python3 bench.py --steps 10 --warmup 3 --no-parity --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['configs_summary'])"; done; }
ORIG="$(mktemp /tmp/lib_orig_XXXXXX.so)"; VAR="$(mktemp /tmp/unit_var_XXXXXX.o)"
cp $PKG/wlsqm/_lib/libwlsqm_hip.so "$ORIG"
trap 'cp "$ORIG" $PKG/wlsqm/_lib/libwlsqm_hip.so; rm -f "$ORIG" "$VAR"' EXIT
run "as-built"
for flags in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fopenmp -I include -I $PKG/csrc $flags -c $PKG/csrc/$UNIT.hip -o $VAR 
  objs=(); for o in $PKG/build/*.o; do [[ "$(basename $o)" == "$UNIT.o" ]] && objs+=("$VAR") || objs+=("$o"); done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fopenmp -o $PKG/wlsqm/_lib/libwlsqm_hip.so "${objs[@]}"
  run "[$flags]"
done
