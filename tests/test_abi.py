"""CPU: the C-ABI library builds, loads, and exports exactly what include/ytvln.h declares (no compute calls)."""
import ctypes
import os
import re
import subprocess

import pytest

from conftest import ROOT


def header_decls():
    text = open(os.path.join(ROOT, "include", "ytvln.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    decls = {}
    for m in re.finditer(r"\b(int64_t|int|const char\*)\s+(ytvln_\w+)\s*\(([^)]*)\)\s*;", text):
        args = [a.strip() for a in m.group(3).split(",") if a.strip() and a.strip() != "void"]
        decls[m.group(2)] = (m.group(1), args)
    return decls


def ctype_of(arg):
    if "*" in arg:
        return ctypes.c_void_p
    if arg.startswith("int64_t"):
        return ctypes.c_int64
    if arg.startswith("int "):
        return ctypes.c_int
    if arg.startswith("float "):
        return ctypes.c_float
    raise AssertionError(arg)


def test_build_entry_point_compiles_and_loads():
    import __graft_entry__ as g
    g.build()
    from ytvln import _lib
    lib = _lib.load()
    assert lib.ytvln_version() == _lib.ABI_VERSION == 2
    assert lib.ytvln_attn_problem_size() == __import__("ctypes").sizeof(_lib.AttnProblem) == 192
    assert lib.ytvln_last_error() is not None


def test_every_declared_symbol_is_exported_and_bound_with_the_declared_signature():
    from ytvln import _lib
    lib = _lib.load()
    decls = header_decls()
    assert len(decls) >= 24
    exported = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    exported = {l.split()[-1] for l in exported.splitlines() if " T " in l}
    for name, (ret, args) in decls.items():
        assert name in exported, f"{name} declared in ytvln.h but not exported"
        assert hasattr(lib, name)
        if name in ("ytvln_version", "ytvln_last_error"):
            continue
        assert name in _lib.SIGNATURES, f"{name} has no ctypes signature"
        want = [ctype_of(a) for a in args]
        assert _lib.SIGNATURES[name] == want, f"{name}: ctypes table {_lib.SIGNATURES[name]} != header {want}"
    assert set(_lib.SIGNATURES) <= set(decls), set(_lib.SIGNATURES) - set(decls)
    assert {e for e in exported if e.startswith("ytvln_")} == set(decls), "exported symbols and header disagree"


def test_bad_arguments_are_rejected_without_touching_the_gpu():
    from ytvln import _lib
    lib = _lib.load()
    rc = lib.ytvln_gemm_f32(None, 1, 0, None, 1, 0, None, 1, None, None, 0, 4, 4, 4, 0, 0.0, None, 0, 0, None)
    assert rc != 0 and b"null" in lib.ytvln_last_error()
    with pytest.raises(RuntimeError, match="ytvln_ln_fwd_f32 failed"):
        _lib.call("ytvln_ln_fwd_f32", 16, None, 16, 16, 16, None, None, None, 4, 30, 1e-12, 0.0, 0.0, None, 0, None)   # H % 4 != 0
    assert lib.ytvln_gemm_workspace_elems(1024, 1024, 16128, 0) > 0
    assert lib.ytvln_gemm_workspace_elems(16128, 1024, 1024, 0) == 0      # no split, persistent kernel off (default): no scratch
    prev = _lib.set_option("GEMM_SK", 1)
    try:          # stream-K form on: two banks of one partial 256x256 tile per workgroup, one workgroup per CU (256 when no device is visible)
        assert lib.ytvln_gemm_workspace_elems(16128, 1024, 1024, 0) % (2 * 256 * 256) == 0
        assert lib.ytvln_gemm_workspace_elems(16128, 1024, 1024, 0) >= 2 * 8 * 256 * 256
    finally:
        _lib.set_option("GEMM_SK", prev)
    assert lib.ytvln_gemm_workspace_elems(100, 64, 64, 0) == 0
    assert lib.ytvln_ln_bwd_blocks(16128) == 768


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from ytvln import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.YtvlnLibraryError, match="no CPU / PyTorch fallback"):
        _lib.load()


def test_cpu_tensors_are_rejected_no_silent_fallback():
    import torch
    from ytvln import ops
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.linear(torch.zeros(4, 8), torch.zeros(3, 8), None)
    with pytest.raises(RuntimeError, match="GPU"):
        ops.cross_entropy(torch.zeros(4, 8), torch.zeros(4, dtype=torch.long), -1)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "youtube-vln_amd", "ytvln")
    for f in os.listdir(pkg):
        if f.endswith(".py"):
            src = open(os.path.join(pkg, f)).read()
            assert "vilbert_ref" not in src and "import oracle" not in src and "from oracle" not in src, f


def test_gemm_launch_planner_host_logic():
    """ytvln_gemm_plan is pure host code: the cost model's choices for the BASELINE config-2 / config-4 shapes."""
    import ctypes
    from ytvln import _lib
    lib = _lib.load()

    def plan(M, N, K, transA=0, epi=0):
        tm, tn, sp = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        assert lib.ytvln_gemm_plan(M, N, K, transA, epi, ctypes.byref(tm), ctypes.byref(tn), ctypes.byref(sp)) == 0
        return tm.value, tn.value, sp.value

    assert plan(16128, 1024, 1024) == (256, 256, 1)            # image-stream projection: one wave of 252 big tiles
    assert plan(24192, 1024, 1024)[:2] != (256, 256)           # config 4: 95 x 4 = 380 tiles of 256x256 would leave half a wave idle
    assert plan(4480, 768, 3072) == (160, 256, 3)              # 84 tiles of 160x256 x 3 splits = 252 workgroups in one round (128x128 unsplit until round 5)
    assert plan(4480, 3072, 768) == (224, 256, 1)              # 4480 = 20 x 224: 240 whole tiles in one round (216 of 256x256, 12 of them half empty)
    assert plan(4480, 2304, 768) == (160, 256, 1)              # 4480 = 28 x 160: 252 whole tiles in one round
    # long contractions (round 6): one round of 160- / 224-row tiles instead of six splits of 128x128 -- the LM decoder's input gradient and its kin
    assert plan(4480, 768, 30528) == (160, 256, 3) and plan(4480, 768, 8192) == (160, 256, 3) and plan(4480, 1024, 8192) == (224, 256, 3)
    assert plan(4480, 30528, 768)[0] != 224                    # many rounds: no gain from the 224-row tile (measured), stays on the well-trodden ones
    assert plan(4480, 1024, 768)[:2] == (64, 64)               # 280 128x128 tiles = 55 % of two waves -> small tiles
    # weight gradients (M-contiguous A): few tiles, long contraction -> deterministic split-K; since round 2 the 256x256 tile competes there
    # (half the operand bytes per flop) and wins when tiles x splits fills one round of the 256 CUs
    assert plan(1024, 1024, 16128, transA=1) == (256, 256, 16)
    assert plan(2048, 1024, 16128, transA=1) == (256, 256, 8)
    assert plan(768, 3072, 4480, transA=1) == (256, 256, 7)
    assert plan(768, 768, 4480, transA=1)[:2] == (128, 128)    # 9 big tiles cannot fill the chip within 16 splits
    assert plan(1024, 1024, 16128, transA=1, epi=1)[2] == 1    # fused activations never split
    tm, tn, sp = plan(30528, 768, 4480, transA=1)              # LM-head weight gradient: 360 big tiles, at most a shallow split
    assert (tm, tn) == (256, 256) and sp <= 2
    assert lib.ytvln_gemm_plan(0, 8, 8, 0, 0, None, None, None) != 0

    def plan_x3(M, N, K, transA=0, epi=0):      # the three-bf16-term form of the fp32 GEMM has its own cost table (LABNOTES.md 5a)
        tm, tn, sp = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        assert lib.ytvln_gemm_plan_x3(M, N, K, transA, epi, ctypes.byref(tm), ctypes.byref(tn), ctypes.byref(sp)) == 0
        return tm.value, tn.value, sp.value

    assert plan_x3(16128, 1024, 1024) == (256, 256, 1)
    tm, tn, sp = plan_x3(1024, 1024, 16128, transA=1)          # weight gradient: 16 wide tiles x 16 k-slabs = one block per CU
    assert (tm, tn) == (256, 256) and sp >= 8
    tm, tn, sp = plan_x3(4480, 768, 3072)                      # 54 wide tiles: split-K fills the chip (native plan: 210 tiles of 128x128)
    assert (tm, tn) == (256, 256) and sp >= 2
    assert plan_x3(4480, 3072, 768, epi=1)[2] == 1             # fused activations never split
    assert plan_x3(768, 768, 4480, transA=1)[:2] != (256, 256)   # 9 wide tiles cannot fill 256 CUs even with 16 k-slabs


def test_integration_md_lists_every_knob():
    """INTEGRATION.md section 4a is exhaustive: (1) its run-time options table = the library's own table (names AND defaults, read through
    ytvln_option_name / ytvln_get_option in a clean environment); (2) its environment table = every YTVLN_* variable the host code reads
    (grep of the Python sources for environ lookups), nothing more and nothing less; (3) the library reads no other YTVLN_* variable."""
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = text[text.index("## 4a."):text.index("## 5.")]
    env_tab, opt_tab = sec.split("**Run-time options of the library**")
    doc_env = set(re.findall(r"^\| `(YTVLN_\w+)` \|", env_tab, flags=re.M))
    doc_opt = {m.group(1): int(m.group(2)) for m in re.finditer(r"^\| `([A-Z0-9_]+)` \| (-?\d+) \|", opt_tab, flags=re.M)}
    # (1) the library's table, defaults read in a child process with no YTVLN_* variable set
    code = ("import sys; sys.path.insert(0, %r); from ytvln import _lib; print(repr(_lib.options()))" % os.path.join(ROOT, "youtube-vln_amd"))
    env = {k: v for k, v in os.environ.items() if not k.startswith("YTVLN_")}
    lib_opt = eval(subprocess.run([os.sys.executable, "-c", code], env=env, check=True, capture_output=True, text=True).stdout.strip().splitlines()[-1])
    assert doc_opt == lib_opt, (doc_opt, lib_opt)
    # (2) environment reads of the host code (+ bench.py)
    read = set()
    files = [os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "__graft_entry__.py")]
    pkg = os.path.join(ROOT, "youtube-vln_amd", "ytvln")
    files += [os.path.join(pkg, f) for f in os.listdir(pkg) if f.endswith(".py")]
    for f in files:
        src = open(f).read()
        read |= set(re.findall(r"environ(?:\.get|\.setdefault)?\s*[\(\[]\s*\"(YTVLN_\w+)\"", src))
        read |= set(re.findall(r"\"(YTVLN_\w+)\"\s+(?:not\s+)?in\s+os\.environ", src))
    assert read == doc_env, (sorted(read - doc_env), sorted(doc_env - read))
    # (3) the library builds its variable names from the option table only ("YTVLN_%s"): no literal getenv("YTVLN_...") anywhere in csrc
    csrc = os.path.join(ROOT, "youtube-vln_amd", "csrc")
    for f in os.listdir(csrc):
        src = open(os.path.join(csrc, f)).read()
        assert not re.findall(r"getenv\s*\(\s*\"YTVLN_", src), f
        assert len(re.findall(r"getenv\s*\(", src)) == (1 if f == "misc.hip" else 0), f
