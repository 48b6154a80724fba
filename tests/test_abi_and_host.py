"""CPU-only checks of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/wlsqm_hip.h declares, the pure-integer host functions are bit-exact with the reference's
golden table, the Python mirror validates arguments like the reference's typed memoryviews, and the
product fails loudly (no CPU fallback) when there is no HIP device."""
import ctypes as C
import os
import re
import sys

import numpy as np
import pytest

import _cases as K

ROOT = K.ROOT


@pytest.fixture(scope="module")
def binding():
    from wlsqm import _binding
    if not os.path.exists(_binding.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    return _binding


def test_library_exports_every_declared_symbol(binding):
    hdr = open(os.path.join(ROOT, "include", "wlsqm_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = sorted(set(re.findall(r"\b(wlsqm_hip_\w+)\s*\(", hdr)))
    assert len(declared) >= 14
    lib = C.CDLL(binding.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), "libwlsqm_hip.so does not export %s" % name


def test_number_of_dofs_and_remap_bit_exact(binding):
    L = binding.lib()
    for dim in (1, 2, 3):
        assert [L.wlsqm_hip_number_of_dofs(dim, k) for k in range(5)] == K.NDOF[dim]
    assert L.wlsqm_hip_number_of_dofs(0, 2) == -1 and L.wlsqm_hip_number_of_dofs(2, 7) == -2
    tab = K.golden("remap.npz")["table"]
    for row in tab:
        n, mask, nr = int(row[0]), int(row[1]), int(row[2])
        o2r = np.full(n, -7, np.int32); r2o = np.full(n, -7, np.int32)
        k = L.wlsqm_hip_remap(o2r.ctypes.data, r2o.ctypes.data, n, mask)
        assert k == nr
        assert np.array_equal(o2r, row[3:3 + n]) and np.array_equal(r2o, row[38:38 + n])
        assert L.wlsqm_hip_number_of_reduced_dofs(n, mask) == n - bin(mask).count("1")


def test_public_api_surface():
    """Names the reference re-exports flat (wlsqm/__init__.py:25-28; tests/test_package.py:24-32)."""
    import wlsqm
    for dim in (1, 2, 3):
        for suffix in ("", "_iterative", "_many", "_iterative_many", "_many_parallel", "_iterative_many_parallel"):
            assert callable(getattr(wlsqm, "fit_%dD%s" % (dim, suffix)))
    for name in ("ExpertSolver", "number_of_dofs", "WEIGHT_UNIFORM", "WEIGHT_CENTER", "ALGO_BASIC", "ALGO_ITERATIVE",
                 "b1_F", "b2_F", "b3_F", "i2_X2", "i3_XYZ2", "b3_XYZ2", "i3_0th_end", "SIZE3"):
        assert hasattr(wlsqm, name), name
    assert not hasattr(wlsqm, "i1_0th_end") and not hasattr(wlsqm, "i2_0th_end")      # defs.pyx:316, 346
    assert wlsqm.b2_XY == 1 << 4 and wlsqm.b3_XZ == 1 << 9 and wlsqm.i3_X2YZ == 32
    assert re.match(r"^\d+\.\d+\.\d+(\.(dev|a|b|rc|post)\d+)?$", wlsqm.__version__)
    import inspect
    sig = inspect.signature(wlsqm.fit_2D_many_parallel)
    assert list(sig.parameters) == ["xk", "fk", "nk", "xi", "fi", "sens", "do_sens", "order", "knowns",
                                    "weighting_method", "ntasks", "debug"]
    assert sig.parameters["ntasks"].default == 8                                       # simple.pyx:192-194
    sig = inspect.signature(wlsqm.fit_3D_iterative)
    assert sig.parameters["knowns"].default == wlsqm.b3_F and sig.parameters["max_iter"].default == 10
    assert sig.parameters["weighting_method"].default == wlsqm.WEIGHT_CENTER and sig.parameters["order"].default == 2


def _batch(n=4, nk=7):
    rng = np.random.default_rng(0)
    return dict(xk=rng.uniform(-1, 1, (n, nk, 2)), fk=rng.uniform(-1, 1, (n, nk)), nk=np.full(n, nk, np.int32),
                xi=np.zeros((n, 2)), fi=np.zeros((n, 6)), sens=None, do_sens=0, order=np.full(n, 2, np.int32),
                knowns=np.zeros(n, np.int64), weighting_method=np.full(n, 2, np.int32))


def test_argument_validation_like_typed_memoryviews():
    import wlsqm
    b = _batch()
    bad = dict(b, nk=b["nk"].astype(np.int64))
    with pytest.raises(ValueError, match="dtype mismatch"):
        wlsqm.fit_2D_many(**bad)
    bad = dict(b, knowns=b["knowns"].astype(np.int32))
    with pytest.raises(ValueError, match="dtype mismatch"):
        wlsqm.fit_2D_many(**bad)
    bad = dict(b, xk=b["xk"].astype(np.float32))
    with pytest.raises(ValueError, match="dtype mismatch"):
        wlsqm.fit_2D_many(**bad)
    bad = dict(b, xk=np.asfortranarray(b["xk"]))                                        # TODO_DEFERRED.md "Issue #5"
    with pytest.raises(ValueError, match="not contiguous in the same dimension"):
        wlsqm.fit_2D_many(**bad)
    bad = dict(b, fi=np.zeros((4, 12))[:, ::2])
    with pytest.raises(ValueError, match="not contiguous in the same dimension"):
        wlsqm.fit_2D_many(**bad)
    bad = dict(b, fk=b["fk"][:, 0])
    with pytest.raises(ValueError, match="wrong number of dimensions"):
        wlsqm.fit_2D_many(**bad)
    bad = dict(b, order=np.full(4, 5, np.int32))
    with pytest.raises(ValueError):
        wlsqm.fit_2D_many(**bad)
    bad = dict(b, fi=np.zeros((4, 3)))
    with pytest.raises(ValueError, match="columns"):
        wlsqm.fit_2D_many(**bad)
    bad = dict(b, nk=np.full(4, 9, np.int32))
    with pytest.raises(ValueError, match="neighbour axis"):
        wlsqm.fit_2D_many(**bad)
    with pytest.raises(ValueError, match="ntasks"):
        wlsqm.fit_2D_many_parallel(ntasks=0, **b)
    with pytest.raises(ValueError, match="sens is None"):
        wlsqm.fit_2D_many(**dict(b, do_sens=1))


def test_expert_solver_argument_checks():
    import wlsqm
    n = 3
    nk = np.full(n, 5, np.int32); o = np.full(n, 1, np.int32); kn = np.zeros(n, np.int64); w = np.full(n, 2, np.int32)
    with pytest.raises(ValueError, match="same length"):                                 # expert.pyx:131-132
        wlsqm.ExpertSolver(2, nk, o[:2], kn, w)
    with pytest.raises(ValueError, match="Dimension must be 1, 2 or 3"):                 # :134-135
        wlsqm.ExpertSolver(4, nk, o, kn, w)
    with pytest.raises(ValueError, match="Unknown algorithm"):                           # :151-156
        wlsqm.ExpertSolver(2, nk, o, kn, w, algorithm=7)
    with pytest.raises(ValueError, match="ntasks must be >= 1"):                         # :158-159
        wlsqm.ExpertSolver(2, nk, o, kn, w, ntasks=0)
    with pytest.raises(ValueError, match="cannot be None"):                              # :139-148
        wlsqm.ExpertSolver(2, nk, o, kn, w, max_iter=None)


def test_fails_loudly_without_a_gpu(binding):
    """No HIP device -> RuntimeError from the library itself; nothing silently falls back to a CPU path."""
    import wlsqm
    if binding.lib().wlsqm_hip_device_count() > 0:
        pytest.skip("a HIP device is present")
    with pytest.raises(RuntimeError, match="no HIP device"):
        wlsqm.fit_2D_many(**_batch())
    nk = np.full(3, 5, np.int32)
    with pytest.raises(RuntimeError, match="no HIP device"):
        wlsqm.ExpertSolver(2, nk, np.full(3, 1, np.int32), np.zeros(3, np.int64), np.full(3, 2, np.int32))


def test_product_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under python-wlsqm_amd/ may reference it."""
    pkg = os.path.join(ROOT, "python-wlsqm_amd")
    for dirpath, _, files in os.walk(pkg):
        if "build" in dirpath.split(os.sep):
            continue
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".sh", ".cpp")):
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert "oracle" not in text.lower() or f == "nothing", "%s mentions the oracle" % os.path.join(dirpath, f)


def test_package_surface_matches_the_reference():
    """Import graph and flat re-exports a user of the reference relies on (reference tests/test_package.py:10-57):
    version string, the fitter submodules (impl / infra / polyeval expose no Python-callable API in the reference either),
    the public names, number_of_dofs for every (dimension, order), and the bit-exact remap contract."""
    import re
    import wlsqm
    assert re.match(r"^\d+\.\d+\.\d+(\.(dev|a|b|rc|post)\d+)?$", wlsqm.__version__)
    from wlsqm.fitter import defs, expert, impl, infra, interp, polyeval, simple  # noqa: F401
    for name in ("fit_1D", "fit_2D", "fit_3D", "fit_1D_many_parallel", "fit_2D_many_parallel", "fit_3D_many_parallel",
                 "fit_2D_iterative_many_parallel", "ExpertSolver", "WEIGHT_UNIFORM", "WEIGHT_CENTER", "ALGO_BASIC",
                 "ALGO_ITERATIVE", "number_of_dofs", "interpolate_fit", "lambdify_fit", "b2_F", "i3_XY"):
        assert hasattr(wlsqm, name), name
    assert [wlsqm.number_of_dofs(1, k) for k in range(5)] == [1, 2, 3, 4, 5]
    assert [wlsqm.number_of_dofs(2, k) for k in range(5)] == [1, 3, 6, 10, 15]
    assert [wlsqm.number_of_dofs(3, k) for k in range(5)] == [1, 4, 10, 20, 35]
    nr, o2r, r2o = infra.remap(6, 0b010010)
    assert nr == 4 and o2r.tolist() == [0, -1, 1, 2, -1, 3] and r2o.tolist() == [0, 2, 3, 5, -1, -1]
    assert infra.number_of_reduced_dofs(6, 0b010010) == 4


def test_product_never_touches_the_oracle_or_the_reference():
    """The oracle is test infrastructure: nothing under python-wlsqm_amd/ or include/ (the product) may mention it, and
    nothing there may read the reference tree at run time."""
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    bad = []
    for top in ("python-wlsqm_amd", "include"):
        for d, _, files in os.walk(os.path.join(root, top)):
            if "build" in d.split(os.sep) or "__pycache__" in d:
                continue
            for f in files:
                if f.endswith((".py", ".hip", ".hpp", ".h", ".sh")):
                    text = open(os.path.join(d, f), errors="replace").read()
                    # (C++ comments cite reference file:line; only scripts could read the tree at run time)
                    if "oracle" in text.lower() or (f.endswith((".py", ".sh")) and "/root/reference" in text):
                        bad.append(os.path.join(d, f))
    assert not bad, bad


def _build_c_example(tmp_path):
    import os, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "python-wlsqm_amd", "wlsqm", "_lib")
    exe = str(tmp_path / "fit_quadratic")
    subprocess.check_call(["gcc", "-O2", "-I", os.path.join(root, "include"), os.path.join(root, "examples", "c", "fit_quadratic.c"),
                           "-o", exe, "-L", lib, "-lwlsqm_hip", "-Wl,-rpath," + lib, "-lm"])
    return exe


def test_c_abi_links_from_plain_c(tmp_path):
    """include/wlsqm_hip.h is a real C header and the library a real C ABI: examples/c/fit_quadratic.c compiles with gcc
    and links against libwlsqm_hip.so (on a machine without a GPU it then reports 'no HIP device' and exits with 2)."""
    import subprocess
    exe = _build_c_example(tmp_path)
    rc = subprocess.call([exe])
    assert rc in (0, 2)


def test_missing_library_is_an_import_error(monkeypatch):
    """No silent fallback: if libwlsqm_hip.so is not there, the first call fails with build instructions."""
    from wlsqm import _binding
    monkeypatch.setattr(_binding, "_lib", None)
    monkeypatch.setattr(_binding, "LIB_PATH", "/nonexistent/libwlsqm_hip.so")
    with pytest.raises(ImportError, match="no CPU fallback"):
        _binding.lib()


def test_bench_headline_is_compact_and_parseable(capsys, tmp_path, monkeypatch):
    """The driver keeps a 2 KB tail of bench.py's stdout and parses its last line: the headline must fit, whatever the full
    record holds (round 2's one-line 20 KB record left BENCH_r02.parsed = null).  Canned data: the full record of round 2."""
    import json
    sys.path.insert(0, ROOT)
    import bench
    full = json.load(open(os.path.join(ROOT, "profiles", "r02_bench_final.json")))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    bench.emit(full)
    out = capsys.readouterr().out
    last = out[-2000:].splitlines()[-1]
    line = json.loads(last)
    assert len(last) < 1800 and len(last) <= bench.COMPACT_LIMIT
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert set(line["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"}
    assert set(line["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample"}
    assert line["config"]["workload"].startswith("C2") and "model" not in line["config"]
    assert abs(line["roofline"]["frac"] - full["roofline"]["frac"]) < 1e-3
    assert abs(line["value"] / full["value"] - 1) < 1e-5
    assert "configs_summary" in line and len(line["configs_summary"]) == len(full["configs"])
    # the full record is on an earlier line and in the side file
    assert out.splitlines()[0].startswith("# full: ")
    assert json.load(open(tmp_path / "gpurun_out" / "bench_full.json"))["value"] == full["value"]
    # a pathological record still yields a parseable headline under the limit
    full["configs"] = {"side%03d" % i: {"ms_per_step": 1.0, "roofline": {"frac": 0.5}} for i in range(400)}
    assert len(json.dumps(bench.compact_line(full), separators=(",", ":"))) <= bench.COMPACT_LIMIT


def test_strict_mode_flag_is_per_thread_and_restored(binding, monkeypatch):
    """wlsqm_hip_set_strict / get_strict (no GPU needed): the context manager restores the previous mode, another thread starts
    from the environment's default, and WLSQM_HIP_STRICT is read by a thread's first use."""
    import threading
    sys.path.insert(0, os.path.join(ROOT, "python-wlsqm_amd"))
    import wlsqm.hip as whip
    prev = whip.set_strict(False)
    try:
        assert whip.get_strict() is False
        with whip.strict():
            assert whip.get_strict() is True
            with whip.strict(False):
                assert whip.get_strict() is False
            assert whip.get_strict() is True
            with whip.strict(None):                      # None: leave the mode alone
                assert whip.get_strict() is True
        assert whip.get_strict() is False
        assert whip.set_strict(True) is False and whip.set_strict(False) is True
        seen = {}
        whip.set_strict(True)

        def other():
            seen["default"] = whip.get_strict()          # a new thread does not inherit this thread's mode
        t = threading.Thread(target=other); t.start(); t.join()
        assert seen["default"] is (os.environ.get("WLSQM_HIP_STRICT", "0") not in ("", "0"))
        monkeypatch.setenv("WLSQM_HIP_STRICT", "1")

        def third():
            seen["env"] = whip.get_strict()
        t = threading.Thread(target=third); t.start(); t.join()
        assert seen["env"] is True
    finally:
        whip.set_strict(prev)


def test_row_hints_are_per_thread_and_restored(binding):
    """wlsqm_hip_set_row_hint / wlsqm_hip_set_order_hint (no GPU needed): the setters return the previous value, wlsqm.hip.row_hint
    restores it (nested too, and when the body raises), another thread starts from the defaults (full rows, sorted by distance), and
    values outside the documented codes mean the default."""
    import threading
    sys.path.insert(0, os.path.join(ROOT, "python-wlsqm_amd"))
    import wlsqm.hip as whip
    L = binding.lib()

    def now():
        r = L.wlsqm_hip_set_row_hint(1); L.wlsqm_hip_set_row_hint(r)
        o = L.wlsqm_hip_set_order_hint(1); L.wlsqm_hip_set_order_hint(o)
        return r, o
    r0, o0 = now()
    try:
        L.wlsqm_hip_set_row_hint(1); L.wlsqm_hip_set_order_hint(1)
        with whip.row_hint("ragged", sorted=False):
            assert now() == (2, 0)
            with whip.row_hint(None):
                assert now() == (0, 1)
            assert now() == (2, 0)
            seen = {}

            def other():
                seen["other"] = now()
            t = threading.Thread(target=other); t.start(); t.join()
            assert seen["other"] == (1, 1)
        assert now() == (1, 1)
        with pytest.raises(RuntimeError):
            with whip.row_hint(sorted=False):
                assert now() == (1, 0)
                raise RuntimeError("body")
        assert now() == (1, 1)
        for bad in (-1, 3, 77):
            L.wlsqm_hip_set_row_hint(bad)
            assert now()[0] == 1
    finally:
        L.wlsqm_hip_set_row_hint(r0); L.wlsqm_hip_set_order_hint(o0)


def test_default_device_precedence(monkeypatch):
    """Host-array entry points: WLSQM_HIP_DEVICE, then a non-default torch device, then LOCAL_RANK, then torch's device."""
    import types
    sys.path.insert(0, os.path.join(ROOT, "python-wlsqm_amd"))
    from wlsqm import _binding as B

    def fake_torch(initialised, current):
        t = types.ModuleType("torch")
        t.cuda = types.SimpleNamespace(is_initialized=lambda: initialised, current_device=lambda: current)
        return t
    for k in ("WLSQM_HIP_DEVICE", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setitem(sys.modules, "torch", fake_torch(True, 0))
    assert B.default_device() == 0
    monkeypatch.setenv("LOCAL_RANK", "3")
    assert B.default_device() == 3                       # torch still on its default 0: the rank's own GPU
    monkeypatch.setitem(sys.modules, "torch", fake_torch(True, 5))
    assert B.default_device() == 5                       # the caller chose a device with torch: follow it
    monkeypatch.setitem(sys.modules, "torch", fake_torch(False, 5))
    assert B.default_device() == 3                       # torch has not touched the GPU: LOCAL_RANK
    # one visible GPU per rank (HIP_VISIBLE_DEVICES set by the launcher): LOCAL_RANK 3 is not an ordinal there
    class OneGpu:
        def wlsqm_hip_device_count(self):
            return 1
    monkeypatch.setattr(B, "lib", lambda: OneGpu())
    monkeypatch.setitem(sys.modules, "torch", fake_torch(True, 0))
    assert B.default_device() == 0
    monkeypatch.setenv("LOCAL_RANK", "0")
    assert B.default_device() == 0
    monkeypatch.undo()
    monkeypatch.setenv("WLSQM_HIP_DEVICE", "7")
    assert B.default_device() == 7


def test_numerics_mode_codes():
    """wlsqm.hip.set_strict / get_strict / strict() / accurate(): False = fast, True = strict, 2 = accurate; strings accepted; the
    mode is per thread and needs no device (wlsqm_hip_set_strict is host state)."""
    import wlsqm.hip as h
    prev = h.set_strict(False)
    try:
        assert h.get_strict() is False
        assert h.set_strict("accurate") is False and h.get_strict() == 2
        assert h.set_strict("strict") == 2 and h.get_strict() is True
        assert h.set_strict(0) is True and h.get_strict() is False
        with h.accurate():
            assert h.get_strict() == 2
            with h.strict():
                assert h.get_strict() is True
            assert h.get_strict() == 2
        assert h.get_strict() is False
        with pytest.raises(ValueError):
            h.set_strict("fastest")
    finally:
        h.set_strict(prev)


def test_isa_record_is_of_these_sources():
    """profiles/isa_r06.txt (tools/isa_stats.sh: registers, spills, scratch, LDS of every kernel the BASELINE configs run) must have been made
    from the sources in the tree: its first line carries the hash of what libwlsqm_hip.manifest lists (round 5's record
    predated the round's last kernels)."""
    import glob
    import hashlib
    rec = os.path.join(ROOT, "profiles", "isa_r06.txt")
    assert os.path.exists(rec), "profiles/isa_r06.txt is missing: bash tools/isa_stats.sh <units> > profiles/isa_r06.txt"
    first = open(rec).readline().strip()
    assert first.startswith("# sources sha256: "), first
    pkg = os.path.join(ROOT, "python-wlsqm_amd")
    names = sorted(glob.glob(os.path.join(pkg, "csrc", "*.hip"))) + sorted(glob.glob(os.path.join(pkg, "csrc", "*.hpp")))
    lines = ["%s  %s\n" % (hashlib.sha256(open(f, "rb").read()).hexdigest(), os.path.relpath(f, pkg)) for f in names]
    lines += ["%s  %s\n" % (hashlib.sha256(open(f, "rb").read()).hexdigest(), os.path.relpath(f, ROOT))
              for f in sorted(glob.glob(os.path.join(ROOT, "include", "*.h")))]
    want = hashlib.sha256("".join(lines).encode()).hexdigest()
    assert first.split(": ")[1] == want, "profiles/isa_r06.txt was made from other sources: regenerate it (tools/isa_stats.sh)"
