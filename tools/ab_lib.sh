#!/usr/bin/env bash
# usage (GPU box, repo root): bash tools/ab_lib.sh OTHER.so "COMMAND"   — runs the timing COMMAND on the library as built, on a PREBUILT
# other library (e.g. the previous round's, built in a worktree and parked under tools/ab_libs/), and as built again; the library as
# built comes back whatever happens.
set -euo pipefail
OTHER="$1"; CMD="$2"
PKG=python-wlsqm_amd
ORIG="$(mktemp /tmp/lib_orig_XXXXXX.so)"
cp $PKG/wlsqm/_lib/libwlsqm_hip.so "$ORIG"
trap 'cp "$ORIG" $PKG/wlsqm/_lib/libwlsqm_hip.so; rm -f "$ORIG"' EXIT
echo "== as built"; bash -c "$CMD"
cp "$OTHER" $PKG/wlsqm/_lib/libwlsqm_hip.so
echo "== [$OTHER]"; bash -c "$CMD"
cp "$ORIG" $PKG/wlsqm/_lib/libwlsqm_hip.so
echo "== as built again"; bash -c "$CMD"
