"""Timeline of the four-wave bf16 GEMM (form 44 = form 4 with stamps): per wave and k-tile the cycles spent in steps 0-2, at the counted wait, at
the barrier and in step 3.
Needs a measurement library: compile youtube-vln_amd/csrc/gemm_bf16_w4.hip with -DYT_W4_MEASURE=1, relink, and run with YTVLN_LIB=<that .so>
(the shipped library runs form 44 as form 4, without stamps)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "youtube-vln_amd"))
import numpy as np
import torch
from ytvln import _lib, ops
dev = torch.device("cuda", 0)
M, N, K = 129024, 1024, 2048
A = torch.randn(M, K, device=dev).bfloat16(); B = torch.randn(N, K, device=dev).bfloat16()
C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
heat = torch.randn(8192, 8192, device=dev).bfloat16()
buf = torch.zeros(256, dtype=torch.int32, device=dev)
_lib.set_option("GEMM_BF16_FORM", 44)
for block in (3, 700, 1500):
    for _ in range(10):
        torch.matmul(heat, heat)
    for _ in range(3):
        ops._gemm_bf16(A, K, 0, B, K, 1, C, N, M, N, K)
    buf.zero_()
    _lib.call("ytvln_gemm_bf16_probe", buf.data_ptr(), block, 0xff)
    ops._gemm_bf16(A, K, 0, B, K, 1, C, N, M, N, K)
    torch.cuda.synchronize()
    _lib.call("ytvln_gemm_bf16_probe", None, 0, 0)
    t = buf.cpu().numpy().astype(np.int64) & 0xFFFFFFFF
    print(f"== workgroup {block}: per k-tile [steps 0-2 | wait | barrier | step 3]")
    for w in range(4):
        v = t[64 * w: 64 * w + 64]
        d = (np.diff(v) & 0xFFFFFFFF).tolist() + [0]
        rows = [(d[4 * k + 3 - 4] if k else 0, d[4 * k], d[4 * k + 1], d[4 * k + 2]) for k in range(16)]
        tot = int((v[-1] - v[3]) & 0xFFFFFFFF) / 15.0
        med = [int(np.median([r[i] for r in rows[1:]])) for i in range(4)]
        print(f" wave {w}: {tot:.0f} cycles per k-tile; medians steps0-2 {med[0]}  wait {med[1]}  barrier {med[2]}  step3 {med[3]};  waits: " + " ".join(str(r[1]) for r in rows[1:]) + " ; barriers: " + " ".join(str(r[2]) for r in rows[1:]))
_lib.set_option("GEMM_BF16_FORM", 0)
