#!/usr/bin/env bash
# usage (GPU box, repo root): bash tools/prof_cmd.sh TAG python-script [args]   — kernel stats of one python tool, top kernels printed
set -euo pipefail
TAG="$1"; shift
OUT="gpurun_out/prof/$TAG"; mkdir -p "$OUT"; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o stats -- python3 "$@" > "$OUT/run.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:int(14)]:
    print("%-110s calls %5s avg %10.1f us  total %6.1f%%" % (r["Name"][:110], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
