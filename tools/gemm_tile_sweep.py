"""Hot-loop sweep of the fp32 GEMM planner's tile / split-K choices on the cfg-2 shapes (run-time options GEMM_TILE, GEMM_SPLITS): which forced
plan, if any, beats the cost model's.  usage: python tools/gemm_tile_sweep.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "youtube-vln_amd"))
import torch
from ytvln import ops, _lib
dev = torch.device("cuda", 0)
TILES = {0: "128x128", 1: "128x64", 3: "256x128", 4: "256x256"}
shapes = [(4480, 768, 768, 0, 1), (4480, 2304, 768, 0, 1), (4480, 3072, 768, 0, 1), (4480, 768, 3072, 0, 1),
          (4480, 768, 2304, 0, 0), (4480, 768, 3072, 0, 0), (4480, 3072, 768, 0, 0), (4480, 768, 768, 0, 0),
          (768, 768, 4480, 1, 0), (2304, 768, 4480, 1, 0), (3072, 768, 4480, 1, 0), (768, 3072, 4480, 1, 0),
          (16128, 1024, 1024, 0, 1), (16128, 1024, 1024, 0, 0), (1024, 1024, 16128, 1, 0)]

def timeit(f, iters=30):
    for _ in range(4): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1000

for M, N, K, ta, tb in shapes:
    A = torch.randn((K, M) if ta else (M, K), device=dev); B = torch.randn((N, K) if tb else (K, N), device=dev)
    C = torch.empty(M, N, device=dev)
    lda = M if ta else K; ldb = K if tb else N
    f = lambda: ops._gemm(A, lda, ta, B, ldb, tb, C, N, M, N, K)
    _lib.set_option("GEMM_TILE", -1); _lib.set_option("GEMM_SPLITS", -1); ops._WS_CACHE.clear()
    base = timeit(f)
    fl = 2.0 * M * N * K
    cells = []
    for t in TILES:
        for sp in (1, 2, 3, 4, 6):
            if sp > 1 and t not in (0, 4): continue
            if sp > 1 and not ta and t == 4: continue
            _lib.set_option("GEMM_TILE", t); _lib.set_option("GEMM_SPLITS", sp); ops._WS_CACHE.clear()
            try:
                us = timeit(f, 15)
                cells.append((us, f"{TILES[t]}/s{sp}"))
            except Exception as e:
                cells.append((1e9, f"{TILES[t]}/s{sp} ERR"))
    _lib.set_option("GEMM_TILE", -1); _lib.set_option("GEMM_SPLITS", -1); ops._WS_CACHE.clear()
    cells.sort()
    print(f"{M:6d} {N:5d} {K:6d} tA{ta} tB{tb}  planner {base:7.1f} us {fl / base / 1e6:6.1f} TF/s | best forced: " +
          "  ".join(f"{n} {u:.1f} ({fl / u / 1e6:.0f})" for u, n in cells[:4]), flush=True)
