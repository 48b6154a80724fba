#!/usr/bin/env python3
"""profiles/traffic_<config>.json from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) of `bench.py --config <config>`.
usage: python tools/make_traffic.py CONFIG fetch_counter_collection.csv write_counter_collection.csv cases_per_launch algorithmic_bytes
Per launch of the fit = one dispatch of each of this package's kernels that the launch consists of (the two kernels of
the moment path are summed).  FETCH_SIZE is doubled (gfx950 tallies the 128-B requests of 16-B-per-lane coalesced
streams at 64 B; MI355X_MICROARCH.md, HBM section); WRITE_SIZE is taken as read.  Units: KB."""
import csv, json, statistics, sys, collections
cfg, ffetch, fwrite, cases, alg = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), float(sys.argv[5])
def per_kernel(path, counter):
    vals = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if "wlsqm::" in r["Kernel_Name"] and "knn" not in r["Kernel_Name"] and r["Counter_Name"] == counter:
            vals[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: dict(dispatches=len(v), median=statistics.median(v), min=min(v), max=max(v)) for k, v in vals.items()}
fetch, write = per_kernel(ffetch, "FETCH_SIZE"), per_kernel(fwrite, "WRITE_SIZE")
rd = sum(v["median"] for v in fetch.values()) * 1024 * 2
wr = sum(v["median"] for v in write.values()) * 1024
out = dict(config=cfg, kernels=sorted(fetch), cases_per_launch=cases,
           correction="FETCH_SIZE x2 (gfx950, 16-B/lane coalesced streams; MI355X_MICROARCH.md HBM section); WRITE_SIZE as read",
           hbm_read_bytes_per_launch=rd, hbm_write_bytes_per_launch=wr, hbm_bytes_per_launch=rd + wr,
           algorithmic_bytes_per_launch=alg, detail=dict(fetch=fetch, write=write))
json.dump(out, open("profiles/traffic_%s.json" % cfg, "w"), indent=1)
print(cfg, "read %.1f MB + written %.1f MB = %.1f MB per launch; algorithmic %.1f MB" % (rd / 1e6, wr / 1e6, (rd + wr) / 1e6, alg / 1e6))
