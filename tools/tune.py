#!/usr/bin/env python3
"""Interleaved A/B timing of kernel variants in ONE process (cdna_hip_programming.md §5.4 rule 24).
usage: python tools/tune.py CONFIG ncases var0 var1 ...   (variants = values of WLSQM_TILE_VARIANT; 'lane' = generic kernel; fix/w1/w4 = curated fixed-K shape / runtime-K one wave / four waves per tile)"""
import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "python-wlsqm_amd"))
import torch
import bench, synth
import wlsqm.hip as whip

cfgname = sys.argv[1]; n = int(sys.argv[2]); variants = sys.argv[3:]
cfg = bench.CONFIGS[cfgname]
dim, order, nk = cfg["dim"], int(os.environ.get("TUNE_ORDER", cfg["order"])), int(os.environ.get("TUNE_NK", cfg["nk"]))
cfg = dict(cfg, nk=nk, order=order)
no = bench.NDOF[dim][order]
dev = torch.device("cuda", 0)
S, F, hoods = bench.build_problem(cfg, n, 0)
S_d = torch.from_numpy(np.ascontiguousarray(S)).to(dev); F_d = torch.from_numpy(F).to(dev)
h_d = torch.from_numpy(hoods.astype(np.int64)).to(dev)
xk = S_d[h_d].contiguous(); fk = F_d[h_d].contiguous(); xi = S_d.clone()
fi = torch.zeros((n, no), dtype=torch.float64, device=dev); fi[:, 0] = F_d
nk_d = torch.full((n,), nk, dtype=torch.int32, device=dev)
kn = torch.full((n,), cfg["knowns"], dtype=torch.int64, device=dev)
wm = torch.full((n,), cfg["wm"], dtype=torch.int32, device=dev)
args = (dim, order, xk, fk, nk_d, xi, fi, kn, wm)
B = bench.bytes_per_fit(dim, order, nk, cfg["knowns"])
res = {v: [] for v in variants}
ref = None
for rnd in range(5):
    for v in variants:
        os.environ.pop("WLSQM_HIP_DISABLE_RING", None); os.environ.pop("WLSQM_HIP_RING_TILES", None)
        os.environ.pop("WLSQM_HIP_TILE_RUN_STORE", None)
        if v in ("runs0", "runs1"):        # tile kernel: fi rows as 8-byte pieces per lane / as the tile's contiguous run through LDS
            os.environ.pop("WLSQM_HIP_DISABLE_TILE", None); os.environ.pop("WLSQM_HIP_DISABLE_FIXEDK", None)
            os.environ["WLSQM_TILE_VARIANT"] = "0"; os.environ["WLSQM_HIP_TILE_RUN_STORE"] = v[-1]
        elif v == "noring":                  # 2D order 4: the two-kernel moment path instead of the one-kernel ring fit
            os.environ.pop("WLSQM_HIP_DISABLE_TILE", None); os.environ.pop("WLSQM_HIP_DISABLE_FIXEDK", None)
            os.environ["WLSQM_HIP_DISABLE_RING"] = "1"
        elif v.startswith("ring"):         # ring fit with N tiles per workgroup (ring8, ring16, ...)
            os.environ.pop("WLSQM_HIP_DISABLE_TILE", None); os.environ.pop("WLSQM_HIP_DISABLE_FIXEDK", None)
            if v[4:]: os.environ["WLSQM_HIP_RING_TILES"] = v[4:]
        elif v == "lane":
            os.environ["WLSQM_HIP_DISABLE_TILE"] = "1"
        elif v in ("w1", "w4"):            # runtime-K kernels (curated fixed-K shapes off): one wave / four waves per tile
            os.environ.pop("WLSQM_HIP_DISABLE_TILE", None); os.environ["WLSQM_HIP_DISABLE_FIXEDK"] = "1"
            os.environ["WLSQM_TILEK_SHAPE"] = v[1]
        elif v == "fix":                   # curated fixed-K shape
            os.environ.pop("WLSQM_HIP_DISABLE_TILE", None); os.environ.pop("WLSQM_HIP_DISABLE_FIXEDK", None)
        elif v.startswith("g"):            # grid size of the persistent launch as a multiple of the resident workgroups
            os.environ.pop("WLSQM_HIP_DISABLE_TILE", None); os.environ.pop("WLSQM_HIP_DISABLE_FIXEDK", None)
            os.environ["WLSQM_TILE_VARIANT"] = "0"; os.environ["WLSQM_HIP_GRID_MULT"] = v[1:]
        else:
            os.environ.pop("WLSQM_HIP_DISABLE_TILE", None); os.environ.pop("WLSQM_HIP_DISABLE_FIXEDK", None)
            os.environ.pop("WLSQM_HIP_GRID_MULT", None)
            os.environ["WLSQM_TILE_VARIANT"] = v
        ms = whip.time_fit_device(*args, reps=20)
        res[v].append(ms)
        if rnd == 0:
            out = fi.clone()
            if ref is None: ref = out
            else:
                sc = ref.abs().amax(0)
                print("variant %s vs first: col max rel diff %.2e" % (v, float(((out - ref).abs().amax(0) / sc).max())))
for v in variants:
    a = np.array(res[v]); med = np.median(a)
    print("variant %-5s median %.4f ms  min %.4f  -> %.3e fits/s  %.0f GB/s  (%.1f%% of 8 TB/s)" %
          (v, med, a.min(), n / med * 1e3, B * n / med / 1e6, B * n / med / 1e6 / 80))
