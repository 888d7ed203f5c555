"""bf16 GEMMs of the bf16-resident path (BASELINE configs[4] shapes: 224 pairs x 576 regions / 80 tokens) against the vendor library behind
torch.matmul (hipBLASLt / rocBLAS), same operands in HBM, same orientation, bf16 or fp32 output, no epilogue on either side.
  fwd  y = x W^T       (A [M,K] k-contiguous, B [N,K] k-contiguous)
  dX   dx = dy W       (A [M,K] k-contiguous, B [K,N] k-major)
  dW   dw = dy^T x     (A [K,M] k-major,      B [K,N] k-major; fp32 output)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "youtube-vln_amd"))
import torch
from ytvln import _lib
if os.environ.get("YTVLN_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["YTVLN_LIB"])
from ytvln import ops
dev = torch.device("cuda", 0)
R, T = 224 * 576, 224 * 80
shapes = [("img fwd 1024->1024", R, 1024, 1024, 0, 1), ("img fwd 1024->3072", R, 3072, 1024, 0, 1), ("img fwd 2048->1024", R, 1024, 2048, 0, 1),
          ("img dX 1024<-1024", R, 1024, 1024, 0, 0), ("img dX 1024<-3072", R, 1024, 3072, 0, 0),
          ("img dW 1024x1024", 1024, 1024, R, 1, 0), ("img dW 3072x1024", 3072, 1024, R, 1, 0),
          ("txt fwd 768->2304", T, 2304, 768, 0, 1), ("txt fwd 768->3072", T, 3072, 768, 0, 1), ("txt fwd 3072->768", T, 768, 3072, 0, 1),
          ("txt fwd 768->768", T, 768, 768, 0, 1), ("txt dX 768<-3072", T, 768, 3072, 0, 0), ("txt dX 3072<-768", T, 3072, 768, 0, 0),
          ("txt dW 3072x768", 3072, 768, T, 1, 0), ("txt dW 768x768", 768, 768, T, 1, 0), ("lm decoder 768->30522", T, 30522, 768, 0, 1)]
only = os.environ.get("SHAPES")


def timed(f, n=10):
    for _ in range(3):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


print(f"{'shape':24s} {'M':>7s} {'N':>6s} {'K':>7s}  {'ytvln us':>9s} {'TF/s':>6s}  {'torch us':>9s} {'TF/s':>6s}  ytvln/torch", flush=True)
for name, M, N, K, ta, tb in shapes:
    if only and not any(o in name for o in only.split(",")):
        continue
    A = torch.randn((K, M) if ta else (M, K), device=dev).bfloat16()
    B = torch.randn((N, K) if tb else (K, N), device=dev).bfloat16()
    cdt = torch.float32 if ta else torch.bfloat16
    ldc = (N + 7) // 8 * 8
    C = torch.empty(M, ldc, device=dev, dtype=cdt)
    mine = timed(lambda: ops._gemm_bf16(A, A.stride(0), ta, B, B.stride(0), tb, C, ldc, M, N, K))
    At, Bt = (A.t() if ta else A), (B.t() if tb else B)
    if ta:          # the library has no bf16 x bf16 -> fp32 matmul through torch: bf16 output there (less traffic than ours)
        Ct = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    else:
        Ct = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    lib = timed(lambda: torch.matmul(At, Bt, out=Ct))
    ref = Ct.float()
    got = C[:, :N].float()
    err = float((got - ref).abs().max() / ref.abs().max())
    fl = 2.0 * M * N * K
    print(f"{name:24s} {M:7d} {N:6d} {K:7d}  {mine*1e3:9.1f} {fl/mine/1e9:6.0f}  {lib*1e3:9.1f} {fl/lib/1e9:6.0f}  {lib/mine:5.2f}x   (max rel diff {err:.1e})", flush=True)
