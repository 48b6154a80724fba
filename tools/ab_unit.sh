#!/usr/bin/env bash
# usage (GPU box, repo root): bash tools/ab_unit.sh UNIT CONFIG "-DFLAG=1" ...   — A/B of compile-time variants of one translation unit
# (csrc/UNIT.hip) on one BASELINE config of bench.py: times the library as built, then rebuilds UNIT.o with each flag set given,
# relinks and times again (kernel ms, fraction of the HBM peak), three passes each, and restores the library.
set -euo pipefail
UNIT="$1"; CFG="$2"; shift 2
PKG=python-wlsqm_amd
run() { for i in 1 2 3; do python3 bench.py --config "$CFG" ${AB_NCASES:+--ncases $AB_NCASES} --steps 20 --warmup 5 --no-parity --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['roofline']['kernel_ms'], d['roofline']['frac'])"; done; }
ORIG="$(mktemp /tmp/lib_orig_XXXXXX.so)"; VAR="$(mktemp /tmp/unit_var_XXXXXX.o)"
cp $PKG/wlsqm/_lib/libwlsqm_hip.so "$ORIG"
# whatever happens below (a failing compile, link or timing command under set -e): the library as built comes back
trap 'cp "$ORIG" $PKG/wlsqm/_lib/libwlsqm_hip.so; rm -f "$ORIG" "$VAR"' EXIT
run "as-built"
for flags in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fopenmp -I include -I $PKG/csrc $flags -c $PKG/csrc/$UNIT.hip -o $VAR
  objs=(); for o in $PKG/build/*.o; do [[ "$(basename $o)" == "$UNIT.o" ]] && objs+=("$VAR") || objs+=("$o"); done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fopenmp -o $PKG/wlsqm/_lib/libwlsqm_hip.so "${objs[@]}"
  run "[$flags]"
done
cp "$ORIG" $PKG/wlsqm/_lib/libwlsqm_hip.so
run "as-built-again"
