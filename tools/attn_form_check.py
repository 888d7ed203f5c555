"""Comparison of two forms of the attention kernels, selected through the library's run-time options (include/ytvln.h: ytvln_set_option):
ATTN_W1 is a bit mask of the one-wave-per-SIMD kernels (1 forward, 2 dQ, 4 dK/dV), ATTN_W1_DKV_ANY = 1 takes the one-wave dK/dV kernel for
launches of any size, ATTN_DSPLIT = 0 switches the d-split of half-filled two-wave forward workgroups off.  Both forms run in this process on
the same seeded inputs; outputs (ctx, lse and, with bwd, dq / dk / dv) must agree to rounding (the forms do the same arithmetic in the same
order; hipcc may contract (s - m) * log2(e) differently).  tests/test_attention_forms_gpu.py calls compare(); as a script:
    python tools/attn_form_check.py "ATTN_W1=0,ATTN_DSPLIT=0" "ATTN_W1=7,ATTN_W1_DKV_ANY=1,ATTN_DSPLIT=0" [bwd]"""
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "youtube-vln_amd"))
SHAPES = [  # N, heads, d, Tq, Tk, p, masked
    (3, 8, 128, 288, 288, 0.1, True), (3, 8, 128, 288, 80, 0.1, True), (3, 8, 128, 80, 288, 0.1, True), (2, 2, 128, 37, 101, 0.0, True),
    (2, 3, 96, 65, 33, 0.1, False), (1, 1, 128, 1, 1, 0.0, False), (2, 2, 128, 32, 32, 0.1, True), (2, 2, 128, 33, 64, 0.1, True),
    (1, 2, 128, 576, 576, 0.1, True), (2, 4, 68, 100, 31, 0.2, True),
    (3, 12, 64, 80, 80, 0.1, True), (2, 4, 64, 100, 37, 0.1, True), (1, 2, 64, 32, 32, 0.0, False), (2, 3, 64, 33, 288, 0.1, True),
    (1, 2, 128, 500, 512, 0.1, True), (1, 3, 64, 512, 490, 0.1, True), (1, 1, 128, 64, 481, 0.0, True)]          # the longest rows the one-wave kernels take


def run(options: dict, bwd: bool) -> dict:
    import torch
    from ytvln import _lib, ops
    dev = torch.device("cuda", 0)
    prev = {k: _lib.set_option(k, v) for k, v in options.items()}
    try:
        ops.DropoutState.manual_seed(1234)          # the same mask stream for both forms
        out = {}
        for i, (N, h, d, Tq, Tk, p, masked) in enumerate(SHAPES):
            g = torch.Generator(device="cpu").manual_seed(100 + i)
            H = h * d
            q, k, v = (torch.randn(N * T, H, generator=g).to(dev) * 1.5 for T in (Tq, Tk, Tk))
            mask = torch.zeros(N, Tk)
            if masked:
                for n in range(N):
                    mask[n, max(1, Tk - 3 * n - 2):] = -10000.0
                if N > 1:
                    mask[1, :] = -10000.0          # a fully masked row set
            mask = mask.to(dev)
            ctx = torch.empty(N * Tq, H, device=dev)
            st = ops.DropoutState(dev)
            sc = 1 / math.sqrt(d)
            lse = ops._attn_fwd(q, 0, H, k, 0, H, v, 0, H, mask, ctx, N, h, Tq, Tk, d, sc, p, st.tensor, 3)
            out[f"ctx{i}"], out[f"lse{i}"] = ctx.cpu().numpy(), lse.cpu().numpy()
            if bwd:
                dout = torch.randn(N * Tq, H, generator=g).to(dev)
                dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
                ops._attn_bwd(q, 0, H, k, 0, H, v, 0, H, mask, ctx, dout, lse, dq, 0, H, dk, 0, H, dv, 0, H, N, h, Tq, Tk, d, sc, p, st.tensor, 3)
                out[f"dq{i}"], out[f"dk{i}"], out[f"dv{i}"] = dq.cpu().numpy(), dk.cpu().numpy(), dv.cpu().numpy()
        torch.cuda.synchronize()
        return out
    finally:
        for k, v in prev.items():
            _lib.set_option(k, v)
        ops.DropoutState.manual_seed(None)


def compare(opts_a: dict, opts_b: dict, bwd: bool = False, tol: float = 2e-6, verbose: bool = True) -> int:
    """Number of tensors on which the two forms disagree beyond `tol` (relative to the tensor's maximum)."""
    res = [run(opts_a, bwd), run(opts_b, bwd)]
    bad = 0
    for key in res[0]:
        x, y = res[0][key], res[1][key]
        if np.array_equal(x, y, equal_nan=True):
            continue
        fin = np.isfinite(x) & np.isfinite(y)
        err = np.abs(x[fin] - y[fin]).max() / max(np.abs(x[fin]).max(), 1.0)      # (inputs are O(1): a tensor that is ~0 up to cancellation noise is compared absolutely)
        nonfin = not np.array_equal(np.isfinite(x), np.isfinite(y))
        ok = err < tol and not nonfin
        bad += 0 if ok else 1
        if verbose:
            print(f"{key}: {'close    ' if ok else 'DIFFERENT'}  max|diff|/max|x| {err:.3e}  mismatching {int((x != y).sum())} of {x.size}, "
                  f"non-finite a/b {(~np.isfinite(x)).sum()}/{(~np.isfinite(y)).sum()}")
    if verbose:
        print(f"{opts_a} vs {opts_b}: {len(res[0]) - bad} of {len(res[0])} tensors agree" + ("" if bad else "  -> OK"))
    return bad


if __name__ == "__main__":
    def parse(s):
        return {kv.split("=")[0]: int(kv.split("=")[1]) for kv in s.split(",") if kv}
    a = parse(sys.argv[1]) if len(sys.argv) > 1 else {"ATTN_W1": 0}
    b = parse(sys.argv[2]) if len(sys.argv) > 2 else {"ATTN_W1": 7}
    sys.exit(1 if compare(a, b, bwd=len(sys.argv) > 3) else 0)
