"""Persistent / stream-K fp32 GEMM against the launch-per-tile kernel, per cfg-2 shape (HIP events, ITERS launches each), and the
in-kernel timeline of one launch (ytvln_gemm_probe: 16 stamps per workgroup at 100 MHz).  SHAPES=img|text|all, PROBE=1 for timelines."""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "youtube-vln_amd"))
import numpy as np
import torch
from ytvln import ops, _lib
dev = torch.device("cuda", 0)
lib = _lib.load()
IMG = [(16128, 1024, 1024, 1), (16128, 1024, 1024, 0), (16128, 3072, 1024, 1), (16128, 1024, 3072, 0), (16128, 2048, 1024, 1), (16128, 1024, 2048, 0)]
TEXT = [(4480, 3072, 768, 1), (4480, 768, 3072, 1), (4480, 3072, 768, 0), (4480, 768, 3072, 0), (4480, 2304, 768, 1), (4480, 768, 2304, 0),
        (4480, 768, 768, 1), (4480, 30528, 768, 1)]
DW = [(1024, 1024, 16128, 2), (3072, 1024, 16128, 2), (2048, 1024, 16128, 2), (768, 3072, 4480, 2), (3072, 768, 4480, 2), (2304, 768, 4480, 2)]      # 2: weight-gradient layout (A = dY^T: M-contiguous, B = X [K, N])
CO = [(4480, 768, 2048, 0), (4480, 2048, 768, 1), (4480, 1024, 768, 1), (4480, 1024, 768, 0), (4480, 768, 1024, 1), (4480, 768, 1024, 0),
      (4480, 768, 768, 0), (16128, 1601, 1024, 1), (16128, 1024, 1601, 0)]      # co-attention text side, the 1601-way image head
which = os.environ.get("SHAPES", "all")
if os.environ.get("SHAPE"):          # SHAPE=M,N,K,tb[;M,N,K,tb...]: ad-hoc shapes (tb = 2: the weight-gradient layout)
    which = "custom"
CUSTOM = [tuple(int(x) for x in t.split(",")) for t in os.environ.get("SHAPE", "").split(";") if t]
shapes = {"custom": CUSTOM, "img": IMG, "text": TEXT, "dw": DW, "co": CO, "textco": TEXT + CO, "all": IMG + TEXT, "all3": IMG + TEXT + DW}[which]
ITERS = int(os.environ.get("ITERS", "20"))
CONFIGS = [("old", dict(GEMM_SK=0)), ("dp4", dict(GEMM_SK=2, GEMM_SK_TILE=4)), ("sk4", dict(GEMM_SK=3, GEMM_SK_TILE=4)),
           ("dp3", dict(GEMM_SK=2, GEMM_SK_TILE=3)), ("sk3", dict(GEMM_SK=3, GEMM_SK_TILE=3)),
           ("sw4", dict(GEMM_SK=0, GEMM_TILE=4, GEMM_SW=1)), ("sw3", dict(GEMM_SK=0, GEMM_TILE=3, GEMM_SW=1)), ("old3", dict(GEMM_SK=0, GEMM_TILE=3)),
           ("old4", dict(GEMM_SK=0, GEMM_TILE=4)), ("old0", dict(GEMM_SK=0, GEMM_TILE=0)), ("sk0", dict(GEMM_SK=3, GEMM_SK_TILE=0)),
           ("dp0", dict(GEMM_SK=2, GEMM_SK_TILE=0)), ("sk0g1", dict(GEMM_SK=3, GEMM_SK_TILE=0, GEMM_SK_GROUPS=1))]
# TSn_m: tile n forced with m splits (planner experiments), e.g. CONFIGS=old,TS5_4
CONFIGS += [(f"TS{t}_{sp}", dict(GEMM_SK=0, GEMM_TILE=t, GEMM_SPLITS=sp)) for t in (0, 4, 5, 6) for sp in (1, 2, 3, 4, 5, 6, 8)]
if os.environ.get("CONFIGS"):
    CONFIGS = [c for c in CONFIGS if c[0] in os.environ["CONFIGS"].split(",")]
DEFAULTS = dict(GEMM_SK=0, GEMM_SK_TILE=-1, GEMM_SK_GROUPS=8, GEMM_TILE=-1, GEMM_SW=0, GEMM_SPLITS=-1)


def setopts(d):
    for k, v in {**DEFAULTS, **d}.items():
        _lib.set_option(k, v)


def timed(fn, iters=ITERS):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1000


NWG = 1024
probe = torch.zeros(NWG * 16, dtype=torch.int64, device=dev)
PASSES = int(os.environ.get("PASSES", "3"))
hot_a, hot_b = torch.randn(8192, 8192, device=dev), torch.randn(8192, 8192, device=dev)


def heat(ms=400):
    """Clocks: a timing taken right after an idle gap (allocation, host work) reads up to 15 % slow -- every pass starts behind ~0.4 s of GEMM."""
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while True:
        for _ in range(10): torch.mm(hot_a, hot_b)
        e1.record(); torch.cuda.synchronize()
        if e0.elapsed_time(e1) > ms: return


for (M, N, K, tb) in shapes:
    ta = 1 if tb == 2 else 0
    tb = 0 if tb == 2 else tb
    A = torch.randn((K, M) if ta else (M, K), device=dev)
    B = torch.randn((N, K) if tb else (K, N), device=dev)
    C = torch.empty(M, N, device=dev)
    go = lambda: ops._gemm(A, A.stride(0), ta, B, B.stride(0), tb, C, N, M, N, K)
    Aop = A.t() if ta else A
    lib_go = (lambda: torch.mm(Aop, B.t(), out=C)) if tb else (lambda: torch.mm(Aop, B, out=C))
    best = {}
    for ps in range(PASSES):          # interleaved passes, the order reversed every other pass: min over passes per configuration
        heat()
        order = CONFIGS + [("hipblaslt", None)]
        for name, o in (order if ps % 2 == 0 else order[::-1]):
            if o is None:
                us = timed(lib_go)
            else:
                setopts(o)
                us = timed(go)
            best[name] = min(best.get(name, 1e30), us)
    line = f"{M:6d} {N:6d} {K:5d} tA{ta} tB{tb} "
    for name, _ in CONFIGS + [("hipblaslt", None)]:
        line += f" {name} {best[name]:7.1f}us {2.0 * M * N * K / best[name] / 1e6:6.1f}TF |"
    print(line, flush=True)
    if os.environ.get("PROBE"):
        for name, o in CONFIGS:
            if name.startswith("old"): continue
            setopts(o)
            go(); torch.cuda.synchronize()
            lib.ytvln_gemm_probe(ctypes.c_void_p(probe.data_ptr()))
            probe.zero_(); torch.cuda.synchronize()
            go(); torch.cuda.synchronize()
            lib.ytvln_gemm_probe(None)
            p = probe.view(NWG, 16).cpu().numpy().astype(np.float64) / 100.0      # us
            p = p[p[:, 0] > 0]
            t0 = p[:, 0].min()
            start, tick, first, end = p[:, 0] - t0, p[:, 1] - p[:, 0], p[:, 2] - p[:, 1], p[:, 15] - t0
            pm, pe = p[:, 3] - p[:, 2], p[:, 4] - p[:, 3]           # first piece: main loop, epilogue issue
            has2 = p[:, 5] > 0
            print(f"   probe {name}: start spread {start.max():5.1f}  ticket {tick.mean():4.1f}/{tick.max():4.1f}  first operands {first.mean():4.1f}/{first.max():4.1f}"
                  f"  piece0 main {pm.mean():6.1f}/{pm.max():6.1f} epi {pe.mean():5.1f}/{pe.max():5.1f}"
                  + (f"  piece1 main {(p[has2, 5] - p[has2, 4]).mean():6.1f} epi {(p[has2, 6] - p[has2, 5]).mean():5.1f}" if has2.any() else "")
                  + f"  end min/mean/max {end.min():6.1f}/{end.mean():6.1f}/{end.max():6.1f}", flush=True)
setopts({})
