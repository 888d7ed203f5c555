"""Per-k-tile timeline of the fp32 GEMM main loop from a -DYT_GEMM_PROBE=1 build of gemm.hip (YTVLN_LIB=.../libytvln_probe.so): waves 0 and 4 of
workgroup 3, k-tiles 4..13, cycles [wait+barrier | k-group 0 | 1 | 2 | 3] and the k-tile total, for the cfg-2 image shape and a weight-gradient shape."""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "youtube-vln_amd"))
import numpy as np
import torch
from ytvln import _lib, ops
dev = torch.device("cuda", 0)
lib = ctypes.CDLL(_lib.LIB_PATH)
heat = torch.randn(8192, 8192, device=dev)
for name, M, N, K, ta, tb in (("img fwd 16128x1024x1024", 16128, 1024, 1024, 0, 1), ("img dX 16128x1024x1024 (B=[K,N])", 16128, 1024, 1024, 0, 0),
                              ("txt fwd 4480x3072x768", 4480, 3072, 768, 0, 1)):
    A = torch.randn((K, M) if ta else (M, K), device=dev); B = torch.randn((N, K) if tb else (K, N), device=dev)
    C = torch.empty(M, N, device=dev)
    for _ in range(6):
        torch.matmul(heat, heat)
    for _ in range(3):
        ops._gemm(A, A.stride(0), ta, B, B.stride(0), tb, C, N, M, N, K)
    torch.cuda.synchronize()
    host = (ctypes.c_uint32 * 128)()
    assert lib.ytvln_gemm_f32_probe_read(host) == 0
    t = np.array(host[:], dtype=np.int64)
    print("==", name)
    for g in range(2):
        v = t[64 * g: 64 * g + 60].reshape(10, 6)
        d = np.diff(v, axis=1) & 0xFFFFFFFF
        nxt = (v[1:, 0] - v[:-1, 5]) & 0xFFFFFFFF
        tot = (v[1:, 0] - v[:-1, 0]) & 0xFFFFFFFF
        print(f" wave {4 * g}: k-tile total median {int(np.median(tot))}; [wait+barrier, g0, g1, g2, g3] medians {[int(np.median(d[:, i])) for i in range(5)]}; loop-back {int(np.median(nxt))}")
