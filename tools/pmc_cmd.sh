#!/usr/bin/env bash
# usage (GPU box, repo root): bash tools/pmc_cmd.sh TAG "CTR1 CTR2 ..." KERNEL_SUBSTRING python-script [args]
# One rocprofv3 --pmc pass (with --kernel-trace only) of a python tool; per-kernel mean of every counter for kernels matching the substring.
set -euo pipefail
TAG="$1"; CTRS="$2"; KSUB="$3"; shift 3
OUT="gpurun_out/prof/$TAG"; mkdir -p "$OUT"; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d "$OUT" -o pmc -- python3 "$@" > "$OUT/run.log" 2>&1
python3 - "$OUT" "$KSUB" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    if sys.argv[2] in r["Kernel_Name"]:
        acc[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print("    %-28s mean %16.1f  (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
