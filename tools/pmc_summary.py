#!/usr/bin/env python3
"""Summaries of the rocprofv3 passes tools/profile_config.sh makes.
usage: python tools/pmc_summary.py TAG CONFIG RAW_DIR [CASES_PER_LAUNCH ALGORITHMIC_BYTES_PER_LAUNCH]
Writes gpurun_out/profiles/TAG_CONFIG_kernel_stats.csv (this package's kernels of the --stats pass) and
gpurun_out/profiles/TAG_CONFIG_pmc_summary.json (gpurun brings back only gpurun_out/; copy them into profiles/ afterwards): for every kernel of the package (the neighbour search of the setup excluded) the
per-dispatch median of each counter, the register / LDS footprint the dispatch reports, and derived ratios:
  hbm_read_bytes = FETCH_SIZE x 1024 x 2 (gfx950 tallies the 128-B requests of wide coalesced reads at 64 B;
  MI355X_MICROARCH.md, HBM section), hbm_write_bytes = WRITE_SIZE x 1024;
  valu_busy = SQ_ACTIVE_INST_VALU x 4 / (GRBM_GUI_ACTIVE / 8 x CUs x 4 SIMDs)   (SQ_* in quad-cycles, GRBM summed over 8 XCDs);
  wait_any / wait_inst / active_any as fractions of SQ_WAVE_CYCLES (they are disjoint and add up to ~1)."""
import collections
import csv
import glob
import json
import os
import statistics
import sys

tag, cfg, raw = sys.argv[1], sys.argv[2], sys.argv[3]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEST = os.path.join(ROOT, "gpurun_out", "profiles")
os.makedirs(DEST, exist_ok=True)
CUS = 256


def find(pattern):
    hits = sorted(glob.glob(os.path.join(raw, "**", pattern), recursive=True))
    return hits[-1] if hits else None


def ours(name):
    return "wlsqm::" in name and "knn" not in name and "bbox" not in name and "cell" not in name


# ---- kernel stats
stats = find("stats*kernel_stats.csv")
rows = []
if stats:
    with open(stats) as f:
        r = csv.DictReader(f)
        fields = r.fieldnames
        rows = [x for x in r if ours(x["Name"])]
    with open(os.path.join(DEST, "%s_%s_kernel_stats.csv" % (tag, cfg)), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=fields)
        w.writeheader()
        w.writerows(rows)

# ---- counters
kern = collections.defaultdict(lambda: collections.defaultdict(list))
foot = {}
for name in ("fetch", "write", "sq1", "sq2", "sq3"):
    path = find("%s*counter_collection.csv" % name)
    if not path:
        continue
    with open(path) as f:
        for x in csv.DictReader(f):
            if not ours(x["Kernel_Name"]):
                continue
            kern[x["Kernel_Name"]][x["Counter_Name"]].append(float(x["Counter_Value"]))
            foot[x["Kernel_Name"]] = dict(vgpr=int(x.get("VGPR_Count", 0) or 0), agpr=int(x.get("Accum_VGPR_Count", 0) or 0),
                                          sgpr=int(x.get("SGPR_Count", 0) or 0), lds=int(x.get("LDS_Block_Size", 0) or 0),
                                          scratch=int(x.get("Scratch_Size", 0) or 0), workgroup=int(x.get("Workgroup_Size", 0) or 0),
                                          grid=int(x.get("Grid_Size", 0) or 0))
out = dict(tag=tag, config=cfg,
           command="tools/profile_config.sh %s %s: rocprofv3 --kernel-trace --pmc <one group> -- python3 bench.py --config %s "
                   "--steps 5 --warmup 1 (separate passes: FETCH_SIZE, WRITE_SIZE, three SQ groups)" % (tag, cfg, cfg),
           kernels={})
for k, cs in kern.items():
    med = {c: statistics.median(v) for c, v in cs.items()}
    d = dict(footprint=foot.get(k), dispatches={c: len(v) for c, v in cs.items()}, median=med)
    der = {}
    if "FETCH_SIZE" in med:
        der["hbm_read_bytes"] = med["FETCH_SIZE"] * 1024 * 2
    if "WRITE_SIZE" in med:
        der["hbm_write_bytes"] = med["WRITE_SIZE"] * 1024
    if "GRBM_GUI_ACTIVE" in med and "SQ_ACTIVE_INST_VALU" in med:
        cyc = med["GRBM_GUI_ACTIVE"] / 8.0
        der["kernel_cycles"] = cyc
        der["valu_busy"] = med["SQ_ACTIVE_INST_VALU"] * 4.0 / (cyc * CUS * 4)
    if "SQ_WAVE_CYCLES" in med:
        wc = med["SQ_WAVE_CYCLES"]
        for c, key in (("SQ_WAIT_INST_ANY", "wait_inst_frac"),):
            if c in med:
                der[key] = med[c] / wc
        if "GRBM_GUI_ACTIVE" in med:
            der["waves_resident_per_simd"] = wc * 4.0 / (med["GRBM_GUI_ACTIVE"] / 8.0 * CUS * 4)
    d["derived"] = der
    out["kernels"][k] = d
# second-group fractions need SQ_WAVE_CYCLES of group 1 (same kernel, same launch size)
for k, d in out["kernels"].items():
    med, der = d["median"], d["derived"]
    if "SQ_WAVE_CYCLES" in med:
        for c, key in (("SQ_WAIT_ANY", "wait_any_frac"), ("SQ_ACTIVE_INST_ANY", "active_any_frac"),
                       ("SQ_WAIT_INST_LDS", "wait_inst_lds_frac")):
            if c in med:
                der[key] = med[c] / med["SQ_WAVE_CYCLES"]
    if "SQ_LDS_IDX_ACTIVE" in med and "SQ_LDS_BANK_CONFLICT" in med and med["SQ_LDS_IDX_ACTIVE"]:
        der["lds_conflict_frac_of_lds_active"] = med["SQ_LDS_BANK_CONFLICT"] / med["SQ_LDS_IDX_ACTIVE"]
    if "SQ_LDS_IDX_ACTIVE" in med and "kernel_cycles" in der:
        der["lds_busy"] = med["SQ_LDS_IDX_ACTIVE"] / (der["kernel_cycles"] * CUS)
json.dump(out, open(os.path.join(DEST, "%s_%s_pmc_summary.json" % (tag, cfg)), "w"), indent=1)
# traffic_<config>.json in the format bench.py reads (sum over the kernels one fit launch consists of)
only = os.environ.get("PMC_TRAFFIC_KERNEL", "")          # substring: the kernels a launch of the config consists of (C4: solve_many)
sel = [d for k, d in out["kernels"].items() if only in k]
rd = sum(d["derived"].get("hbm_read_bytes", 0.0) for d in sel)
wr = sum(d["derived"].get("hbm_write_bytes", 0.0) for d in sel)
if rd and wr and len(sys.argv) > 5:
    cases, alg = int(sys.argv[4]), float(sys.argv[5])
    json.dump(dict(config=cfg, tag=tag, kernels=sorted(out["kernels"]), cases_per_launch=cases,
                   units_per_launch=cases * (int(os.environ.get("PMC_C4_NRHS", "256")) if cfg == "C4" else 1),      # (C4: a unit = a case x a stacked right-hand side: what bench.py's side line asks for)
                   correction="FETCH_SIZE x2 (gfx950, 16-B/lane coalesced streams; MI355X_MICROARCH.md HBM section); WRITE_SIZE as read",
                   hbm_read_bytes_per_launch=rd, hbm_write_bytes_per_launch=wr, hbm_bytes_per_launch=rd + wr,
                   algorithmic_bytes_per_launch=alg, source="%s_%s_pmc_summary.json" % (tag, cfg)),
              open(os.path.join(DEST, "traffic_%s.json" % cfg), "w"), indent=1)
    print("traffic %s: read %.1f MB + written %.1f MB = %.1f MB per launch; algorithmic %.1f MB" % (cfg, rd / 1e6, wr / 1e6, (rd + wr) / 1e6, alg / 1e6))
for r in rows:
    print("%s: %s calls, avg %.1f us" % (r["Name"][:110], r["Calls"], float(r["AverageNs"]) / 1e3))
for k, d in out["kernels"].items():
    print(k[:110]); print("   ", d["footprint"]); print("   ", {a: (round(b, 4) if b < 100 else int(b)) for a, b in d["derived"].items()})
