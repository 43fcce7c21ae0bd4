"""The C-ABI library loads (no GPU needed) and exports exactly what include/videoitg_hip.h declares."""
import os
import re

import pytest

from videoitg_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header():
    src = open(os.path.join(ROOT, "include", "videoitg_hip.h")).read()
    return re.sub(r"/\*.*?\*/", "", src, flags=re.S)


def header_symbols():
    return sorted(set(re.findall(r"\bint\s+(vitg_\w+)\s*\(", _header())))


def test_header_declares_entry_points():
    syms = header_symbols()
    assert "vitg_gemm" in syms and "vitg_attention" in syms and len(syms) >= 12


def test_library_exports_every_declared_symbol():
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    lib = _lib.load()
    for s in header_symbols():
        assert hasattr(lib, s), f"{s} declared in include/videoitg_hip.h but not exported"
    assert sorted(_lib.SIGNATURES) == header_symbols()
    assert lib.vitg_version() >= 100


def test_arg_counts_match_header():
    src = _header()
    for name, args in _lib.SIGNATURES.items():
        m = re.search(r"\bint\s+" + name + r"\s*\((.*?)\)\s*;", src, flags=re.S)
        assert m, name
        params = m.group(1).strip()
        n = 0 if params in ("", "void") else params.count(",") + 1
        assert n == len(args), (name, n, len(args))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "videoitg_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith(".py"):
                txt = open(os.path.join(dp, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt, os.path.join(dp, f)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.VitgError):
        _lib.load()


def test_product_library_ships_no_research_scaffolding():
    """the wrong-results-by-design switches and the opt-in GEMM variants live in libvideoitg_hip_prof.so
    (include/videoitg_hip_profiling.h); the product library must not export them and the product never loads the other one."""
    import ctypes
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    assert os.path.basename(_lib.LIB_PATH) == "libvideoitg_hip.so" and not _lib.PROFILING
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for s in _lib.PROFILING_SIGNATURES:
        assert not hasattr(lib, s), f"{s} belongs to the profiling library only"
    pkg = os.path.join(ROOT, "videoitg_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith(".py") and f != "_lib.py":
                assert "libvideoitg_hip_prof" not in open(os.path.join(dp, f)).read(), f


def test_alternative_library_path_is_confined_to_experiment_builds_of_this_library():
    """VITG_LIB_PATH (A/B measurements of csrc/build.sh --exp builds) accepts only libvideoitg_hip_exp_<NAME>.so next to the product library"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for bad in ("/tmp/libvideoitg_hip_exp_x.so", os.path.join(root, "videoitg_amd", "libsomething.so")):
        r = subprocess.run([sys.executable, "-c", "import videoitg_amd._lib"], env={**os.environ, "VITG_LIB_PATH": bad, "PYTHONPATH": root},
                           capture_output=True, text=True)
        assert r.returncode != 0 and "VITG_LIB_PATH must name" in r.stderr
    ok = os.path.join(root, "videoitg_amd", "libvideoitg_hip_exp_doesnotexist.so")
    r = subprocess.run([sys.executable, "-c", "import videoitg_amd._lib as l; print(l.LIB_PATH)"], env={**os.environ, "VITG_LIB_PATH": ok, "PYTHONPATH": root},
                       capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.strip() == ok          # (selecting is allowed; loading a missing file then fails loudly in load())
