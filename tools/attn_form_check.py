"""Comparison of two forms of the attention kernels selected by an environment knob (default: YTVLN_ATTN_W1=0 vs 1; the one-wave-per-SIMD dQ and
dK/dV kernels: KNOB=YTVLN_ATTN_W1_DQ, and KNOB=YTVLN_ATTN_W1_DKV A=0 B=2 -- 2 forces that form for launches of any size).  Each form runs in its
own process (the knobs are read once), on the same seeded inputs; outputs (ctx, lse and, with BWD=1, dq/dk/dv) must agree to rounding (the forms
do the same arithmetic in the same order; hipcc may contract (s - m) * log2(e) differently).  tests/test_attention_forms_gpu.py runs it."""
import os, sys, math, subprocess, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = [  # N, heads, d, Tq, Tk, p, masked
    (3, 8, 128, 288, 288, 0.1, True), (3, 8, 128, 288, 80, 0.1, True), (3, 8, 128, 80, 288, 0.1, True), (2, 2, 128, 37, 101, 0.0, True),
    (2, 3, 96, 65, 33, 0.1, False), (1, 1, 128, 1, 1, 0.0, False), (2, 2, 128, 32, 32, 0.1, True), (2, 2, 128, 33, 64, 0.1, True),
    (1, 2, 128, 576, 576, 0.1, True), (2, 4, 68, 100, 31, 0.2, True),
    (3, 12, 64, 80, 80, 0.1, True), (2, 4, 64, 100, 37, 0.1, True), (1, 2, 64, 32, 32, 0.0, False), (2, 3, 64, 33, 288, 0.1, True),
    (1, 2, 128, 500, 512, 0.1, True), (1, 3, 64, 512, 490, 0.1, True), (1, 1, 128, 64, 481, 0.0, True)]          # the longest rows the one-wave kernels take

def child(path):
    sys.path.insert(0, os.path.join(ROOT, "youtube-vln_amd"))
    import torch
    from ytvln import ops
    dev = torch.device("cuda", 0)
    ops.DropoutState.manual_seed(1234)          # (the default stream is keyed by torch.initial_seed(): different in every process)
    out = {}
    bwd = bool(int(os.environ.get("BWD", "0")))
    for i, (N, h, d, Tq, Tk, p, masked) in enumerate(SHAPES):
        g = torch.Generator(device="cpu").manual_seed(100 + i)
        H = h * d
        q, k, v = (torch.randn(N * T, H, generator=g).to(dev) * 1.5 for T in (Tq, Tk, Tk))
        mask = torch.zeros(N, Tk)
        if masked:
            for n in range(N):
                mask[n, max(1, Tk - 3 * n - 2):] = -10000.0
            if N > 1: mask[1, :] = -10000.0          # a fully masked row set
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
    np.savez(path, **out)

if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(sys.argv[1]); sys.exit(0)
    knob, va, vb = os.environ.get("KNOB", "YTVLN_ATTN_W1"), os.environ.get("A", "0"), os.environ.get("B", "1")
    tmp = tempfile.mkdtemp()
    res = []
    for tag, val in (("a", va), ("b", vb)):
        path = os.path.join(tmp, tag + ".npz")
        subprocess.run([sys.executable, os.path.abspath(__file__), path], check=True, env={**os.environ, knob: val})
        res.append(np.load(path))
    bad = 0
    for key in res[0].files:
        x, y = res[0][key], res[1][key]
        same = np.array_equal(x, y, equal_nan=True)
        if not same:
            fin = np.isfinite(x) & np.isfinite(y)
            err = np.abs(x[fin] - y[fin]).max() / max(np.abs(x[fin]).max(), 1.0)      # (inputs are O(1): a tensor that is ~0 up to cancellation noise is compared absolutely)
            nonfin = not np.array_equal(np.isfinite(x), np.isfinite(y))
            # the compiler may contract (s - m) * log2(e) differently in the two forms: differences of a few ulp are not a defect
            ok = err < float(os.environ.get("TOL", "2e-6")) and not nonfin
            bad += 0 if ok else 1
            if not ok and key.startswith("ctx"):
                N, h, d, Tq, Tk, _, _ = SHAPES[int(key[3:])]
                rel = np.abs(x - y).reshape(N, Tq, h, d).max(axis=3) > 1e-6 * np.abs(x).max()      # [N, Tq, h]
                qt = sorted({int(q) // 32 for q in np.nonzero(rel)[1]})
                print(f"  {key}: bad (n, h) pairs {sorted({(int(a), int(c)) for a, _, c in zip(*np.nonzero(rel))})[:12]} ... query tiles {qt}; "
                      f"rows in tile {sorted({int(q) % 32 for q in np.nonzero(rel)[1]})}")
            print(f"{key}: {'close    ' if ok else 'DIFFERENT'}  max|diff|/max|x| {err:.3e}  mismatching {int((x != y).sum())} of {x.size}, non-finite a/b {(~np.isfinite(x)).sum()}/{(~np.isfinite(y)).sum()}")
    print(f"{knob}={va} vs {vb}: {len(res[0].files) - bad} of {len(res[0].files)} tensors agree" + ("" if bad else "  -> OK"))
    sys.exit(1 if bad else 0)
