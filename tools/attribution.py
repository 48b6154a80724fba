#!/usr/bin/env python3
"""Attribution of the fast kernels' distance from the reference.

CPU only.  Runs oracle/variants.c — the oracle's arithmetic with ONE MI355X-first choice switched on at a time — on the
reference-generated goldens at the density the metric is quoted on (tests/golden/config_{C2,C3,C5}_1M.npz) and prints, per
variant, the column metric E_m = max_j |fi - fi_ref| / max_j |fi_ref| against the REFERENCE's output (max over columns, and
how many columns meet 1e-10).  The row with no switch must equal the oracle bit for bit."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _cases as K          # noqa: E402
import _parity as P         # noqa: E402
from oracle import oracle   # noqa: E402

MOMENT, SPLIT, FMA, FASTW, LDLT, SYM = 1, 2, 4, 8, 16, 32
VARIANTS = [("reference order (= oracle)", 0, 1), ("+ symmetric (upper triangle mirrored)", SYM, 1),
            ("+ FMA contraction", FMA, 1), ("+ weights by d2 * (1/max_d2)", FASTW, 1),
            ("+ sums split over 2 lanes", SPLIT, 2), ("+ sums split over 4 lanes", SPLIT, 4),
            ("+ moment form", MOMENT, 1), ("+ unscaled LDL^T (no Ruiz, no pivoting)", LDLT, 1),
            ("moment + split 2 + FMA + fast weights (fast kernel's assembly), reference's LU", MOMENT | SPLIT | FMA | FASTW, 2),
            ("entry form + LDL^T + FMA", LDLT | FMA, 1),
            ("everything (the fast kernels' arithmetic)", MOMENT | SPLIT | FMA | FASTW | LDLT, 2)]


def main():
    subprocess = __import__("subprocess")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    L = C.CDLL(os.path.join(ROOT, "oracle", "libwlsqm_variants.so"))
    L.wlsqm_variant_fit_many.argtypes = [C.c_int, C.c_int, C.c_int, C.c_long, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_longlong, C.c_int, C.c_int, C.c_int]
    for name in ("C2_1M", "C5_1M", "C3_1M"):
        c = K.config_dense(name)
        dim, order, no, n, nk = c["dim"], c["order"], c["no"], c["n"], c["nkv"]
        kn, wm = int(c["knowns_a"][0]), int(c["wm_a"][0])
        cols = [a for a in range(no) if not (kn >> a) & 1]
        ref = c["g"]["fi"]
        fi_o = c["fi0"].copy()
        oracle.fit_many(dim, c["xk"], c["fk"], c["nk_a"], c["xi"], fi_o, None, 0, c["order_a"], c["knowns_a"], c["wm_a"])
        truth = P.truth_fit(dim, c["xk"], c["fk"], c["nk_a"], c["xi"], c["fi0"], c["order_a"], c["knowns_a"], c["wm_a"])
        N = P.column_metric(ref, truth)
        print("\n%s (%dD order %d, %d neighbours, 1 024 cases of the full cloud): reference's own distance to the 80-bit solution "
              "N_max = %.2e" % (name, dim, order, nk, max(N[m] for m in cols)))
        print("| arithmetic | E_max vs reference | columns <= 1e-10 | E_max vs 80-bit truth |")
        print("|---|---|---|---|")
        for label, flags, ns in VARIANTS:
            fi = np.ascontiguousarray(c["fi0"].copy())
            xk = np.ascontiguousarray(c["xk"]); fk = np.ascontiguousarray(c["fk"]); xi = np.ascontiguousarray(c["xi"])
            rc = L.wlsqm_variant_fit_many(dim, order, no, n, nk, xk.ctypes.data, fk.ctypes.data, xi.ctypes.data, fi.ctypes.data, kn, wm, flags, ns)
            assert rc == 0
            if flags == 0:
                assert np.array_equal(fi, fi_o), "variant 0 must be the oracle bit for bit"
            E = P.column_metric(fi, ref); T = P.column_metric(fi, truth)
            print("| %s | %.2e | %d of %d | %.2e |" % (label, max(E[m] for m in cols), sum(E[m] <= 1e-10 for m in cols), len(cols),
                                                     max(T[m] for m in cols)))


if __name__ == "__main__":
    main()
