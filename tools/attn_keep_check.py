"""Debug aid for the bf16 attention's stored dropout decisions: decode the keep buffer the forward wrote and compare it with the mask recovered
from the forward output (V = identity columns); then dQ / dK / dV one by one against fp64 with that mask."""
import math, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "youtube-vln_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from ytvln import ops
dev = torch.device("cuda", 0)
BF = torch.bfloat16
d = int(os.environ.get("D", "128"))
N, heads, Tq, p = 2, 2, 40, 0.2
Tk = 96 if d >= 96 else 64
H = heads * d
st = ops.DropoutState(dev)
site, scale = 5, 1 / math.sqrt(d)
z, zk = torch.zeros(N * Tq, H, device=dev, dtype=BF), torch.zeros(N * Tk, H, device=dev, dtype=BF)
eye = torch.zeros(N, Tk, heads, d, device=dev)
for j in range(Tk):
    eye[:, j, :, j] = 1.0
eye = eye.reshape(N * Tk, H).to(BF)
out = torch.empty(N * Tq, H, device=dev, dtype=BF)
lse = ops._attn_fwd(z, 0, H, zk, 0, H, eye, 0, H, None, out, N, heads, Tq, Tk, d, scale, p, st.tensor, site)
torch.cuda.synchronize()
keep = (out.float().view(N, Tq, heads, d)[..., :Tk] > 0).permute(0, 2, 1, 3)          # [N, heads, Tq, Tk]
kb = lse._ytvln_keep.view(torch.int64).cpu()
nqb, nkt = (Tq + 31) // 32, (Tk + 31) // 32
kb = kb.view(N * heads, nqb, nkt, 16)
dec = torch.zeros(N * heads, nqb * 32, nkt * 32, dtype=torch.bool)
for r in range(16):
    for lane in range(64):
        l31, half = lane & 31, lane >> 5
        key = (r & 3) + 8 * (r >> 2) + 4 * half
        dec[:, l31::32, key::32] = 0
for nh in range(N * heads):
    for qb in range(nqb):
        for t in range(nkt):
            for r in range(16):
                m = int(kb[nh, qb, t, r]) & 0xFFFFFFFFFFFFFFFF
                for lane in range(64):
                    l31, half = lane & 31, lane >> 5
                    key = (r & 3) + 8 * (r >> 2) + 4 * half
                    dec[nh, qb * 32 + l31, t * 32 + key] = (m >> lane) & 1
dec = dec.view(N, heads, nqb * 32, nkt * 32)[:, :, :Tq, :Tk]
mm = (dec != keep.cpu())
import collections
hr = collections.Counter(); hh = collections.Counter(); hq = collections.Counter(); ht = collections.Counter()
for (n_, h_, qi, kj) in mm.nonzero().tolist():
    c = kj % 32
    half = (c >> 2) & 1; r = (c & 3) + 4 * (c >> 3)
    hr[r] += 1; hh[half] += 1; hq[qi // 32] += 1; ht[kj // 32] += 1
print("by r", sorted(hr.items()), "by half", sorted(hh.items()), "by qblock", sorted(hq.items()), "by ktile", sorted(ht.items()))
print("first mismatches", mm.nonzero()[:12].tolist())
print("stored keep == forward keep:", bool(torch.equal(dec, keep.cpu())), "mismatches", int((dec != keep.cpu()).sum()), "of", dec.numel())
from test_bf16_gpu import _ref_attention
from test_kernels_gpu import rnd
from helpers import rel_l2
q, k, v = (rnd(dev, N * T, H, seed=sd).to(BF) for T, sd in ((Tq, 1), (Tk, 2), (Tk, 3)))
mask = torch.zeros(N, Tk, device=dev)
lse = ops._attn_fwd(q, 0, H, k, 0, H, v, 0, H, mask, out, N, heads, Tq, Tk, d, scale, p, st.tensor, site)
qd, kd, vd = (t.double().view(N, -1, H).requires_grad_(True) for t in (q, k, v))
ref = _ref_attention(qd, kd, vd, mask.double(), heads, keep.double(), p)
print("fwd", rel_l2(out.view(N, Tq, H), ref))
dout = rnd(dev, N * Tq, H, seed=4).to(BF)
ref.backward(dout.double().view(N, Tq, H))
dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
ops._attn_bwd(q, 0, H, k, 0, H, v, 0, H, mask, out, dout, lse, dq, 0, H, dk, 0, H, dv, 0, H, N, heads, Tq, Tk, d, scale, p, st.tensor, site)
print("dq", rel_l2(dq.view(N, Tq, H), qd.grad), "dk", rel_l2(dk.view(N, Tk, H), kd.grad), "dv", rel_l2(dv.view(N, Tk, H), vd.grad))
