"""Phase timeline of the bf16 GEMM main loop (ytvln_gemm_bf16_probe): waves 0 (group 0) and 4 (group 1) of one workgroup stamp s_memtime at
the end of each phase's own work and after the barrier behind it, k-tiles 8..15 of a 129024 x 1024 x 2048 forward GEMM.  Prints, per group and
k-tile, the cycles from one stamp to the next: L01 work | wait at barrier | M01 work | wait | L23 work | wait | M23 work | wait."""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "youtube-vln_amd"))
import numpy as np
import torch
from ytvln import _lib, ops
dev = torch.device("cuda", 0)
M, N, K = int(os.environ.get("M", 129024)), int(os.environ.get("N", 1024)), int(os.environ.get("K", 2048))
A = torch.randn(M, K, device=dev).bfloat16()
B = torch.randn(N, K, device=dev).bfloat16()
C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
heat = torch.randn(8192, 8192, device=dev).bfloat16()
buf = torch.zeros(128, dtype=torch.int32, device=dev)
names = ["L01", "bar", "M01", "bar", "L23", "bar", "M23", "bar"]
for mask in (0xFF, 0x55, 0xAA):
    for block in (3, 700):
        for _ in range(10):
            torch.matmul(heat, heat)
        for _ in range(3):
            ops._gemm_bf16(A, K, 0, B, K, 1, C, N, M, N, K)
        buf.zero_()
        _lib.call("ytvln_gemm_bf16_probe", buf.data_ptr(), block, mask)
        ops._gemm_bf16(A, K, 0, B, K, 1, C, N, M, N, K)
        torch.cuda.synchronize()
        _lib.call("ytvln_gemm_bf16_probe", None, 0, 0)
        t = buf.cpu().numpy().astype(np.int64) & 0xFFFFFFFF
        print(f"== mask {mask:#x} workgroup {block}")
        for g in range(2):
            v = t[64 * g: 64 * g + 64]
            idx = [i for i in range(64) if (mask >> (i % 8)) & 1]
            vv = v[idx]
            d = np.diff(vv) & 0xFFFFFFFF
            per = len(idx) // 8
            print(f" group {g}: total {int((vv[-1] - vv[0]) & 0xFFFFFFFF)} cycles over {len(idx) - 1} intervals = {float((vv[-1]-vv[0]) & 0xFFFFFFFF) / 7.875 if per == 8 else float((vv[-1]-vv[0]) & 0xFFFFFFFF) * per / (len(idx)-1):.0f} per k-tile")
            for kt in range(8):
                row = d[kt * per: (kt + 1) * per]
                labels = [names[(i + 1) % 8] for i in range(8) if (mask >> i) & 1]
                print(f"   kt {8 + kt}: " + "  ".join(f"{int(x):5d}" for x in row) + "     (interval ENDING at the next stamp; stamps after: " + " ".join(n for i, n in enumerate(names) if (mask >> i) & 1) + ")")
