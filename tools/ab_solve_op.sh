#!/usr/bin/env bash
# usage (GPU box, repo root): bash tools/ab_solve_op.sh "-DWLSQM_OP_NSET16=4" ...   — A/B of solve_op.hip compile-time variants on BASELINE configs[3]:
# times the library as built, then rebuilds solve_op.o with each flag set given, relinks and times again (C4 kernel ms, frac).
set -euo pipefail
PKG=python-wlsqm_amd
run() { python3 bench.py --config C4 --steps 10 --warmup 3 --no-parity 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['roofline']['kernel_ms'], d['roofline']['frac'])"; }
ORIG="$(mktemp /tmp/lib_orig_XXXXXX.so)"; VAR="$(mktemp /tmp/unit_var_XXXXXX.o)"
cp $PKG/wlsqm/_lib/libwlsqm_hip.so "$ORIG"
# whatever happens below (a failing compile, link or timing command under set -e): the library as built comes back
trap 'cp "$ORIG" $PKG/wlsqm/_lib/libwlsqm_hip.so; rm -f "$ORIG" "$VAR"' EXIT
run "as-built"
for flags in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fopenmp -I include -I $PKG/csrc $flags -c $PKG/csrc/solve_op.hip -o "$VAR"
  objs=(); for o in $PKG/build/*.o; do [[ "$(basename $o)" == "solve_op.o" ]] && objs+=("$VAR") || objs+=("$o"); done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fopenmp -o $PKG/wlsqm/_lib/libwlsqm_hip.so "${objs[@]}"
  run "[$flags]"
done
cp "$ORIG" $PKG/wlsqm/_lib/libwlsqm_hip.so
run "as-built-again"
