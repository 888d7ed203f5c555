"""Driver for tools/gemm_bf16_pmc.sh: the cfg-5 image-side forward GEMM (129024 x 1024 x K, bf16) repeated per main-loop form and for the library."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "youtube-vln_amd"))
import torch
from ytvln import _lib, ops
dev = torch.device("cuda", 0)
M, N, K = 129024, 1024, int(os.environ.get("K", 2048))
A = torch.randn(M, K, device=dev).bfloat16(); B = torch.randn(N, K, device=dev).bfloat16()
C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
for form in [int(x) for x in os.environ.get("FORMS", "0,4").split(",")]:
    _lib.set_option("GEMM_BF16_FORM", form)
    for _ in range(12):
        ops._gemm_bf16(A, K, 0, B, K, 1, C, N, M, N, K)
    torch.cuda.synchronize()
_lib.set_option("GEMM_BF16_FORM", 0)
for _ in range(12):
    torch.matmul(A, B.t(), out=C)
torch.cuda.synchronize()
