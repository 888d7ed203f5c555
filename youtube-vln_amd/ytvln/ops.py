"""torch.autograd.Function wrappers over the C ABI (include/ytvln.h).

PyTorch supplies device memory, the current HIP stream and the autograd graph; every FLOP / byte of the hot path runs in
libytvln.so.  All wrappers require CUDA(HIP) fp32 tensors -- there is no CPU or eager-PyTorch fallback.
"""
from __future__ import annotations

import contextlib
import math
import ctypes
import os
from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import EPI_GELU, EPI_MUL_DGELU, EPI_NONE, EPI_RELU, GEMM_A_ZERO_PADDED, GEMM_SPLIT_BF16X3, call

Tensor = torch.Tensor
_ACT = {"none": EPI_NONE, None: EPI_NONE, "gelu": EPI_GELU, "relu": EPI_RELU}


# ------------------------------------------------------------------------------------------------------------------
# plumbing helpers
# ------------------------------------------------------------------------------------------------------------------
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream() -> int:
    """Raw hipStream_t of torch's current stream (the C accessor avoids building Stream objects ~2500 times per step)."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def _ptr(t: Optional[Tensor], offset_elems: int = 0):
    if t is None:
        return None
    return t.data_ptr() + 4 * offset_elems


def _check(t: Tensor, name: str, dtype=torch.float32):
    if not t.is_cuda:
        raise RuntimeError(f"ytvln.ops: `{name}` must live on the GPU (no CPU fallback); got device {t.device}")
    if t.dtype != dtype:
        raise RuntimeError(f"ytvln.ops: `{name}` must be {dtype}; got {t.dtype}")


def _rows2d(x: Tensor, name: str) -> Tuple[Tensor, int, int, int]:
    """View x as [M, K] rows with unit inner stride; returns (tensor-keeping-storage, M, K, leading dimension)."""
    _check(x, name)
    K = x.shape[-1]
    if x.dim() == 2 and x.stride(1) == 1 and x.stride(0) >= K:
        return x, x.shape[0], K, x.stride(0)
    if not x.is_contiguous():
        x = x.contiguous()
    return x, x.numel() // max(K, 1), K, K


def _rows2d_any(x: Tensor) -> Tuple[Tensor, int, int, int]:
    """_rows2d without the fp32 check (bf16 logits of the bf16-resident path; rows may carry a padded leading dimension)."""
    if not x.is_cuda:
        raise RuntimeError("ytvln.ops: tensor must live on the GPU")
    K = x.shape[-1]
    if x.dim() == 2 and x.stride(1) == 1 and x.stride(0) >= K:
        return x, x.shape[0], K, x.stride(0)
    if not x.is_contiguous():
        x = x.contiguous()
    return x, x.numel() // max(K, 1), K, K


def _i64(t: Tensor, name: str) -> Tensor:
    if not t.is_cuda:
        raise RuntimeError(f"ytvln.ops: `{name}` must live on the GPU")
    if t.dtype != torch.int64:
        t = t.to(torch.int64)
    return t.contiguous()


# ------------------------------------------------------------------------------------------------------------------
# dropout RNG state: (seed, forward counter) in device memory + host-side site ids
# ------------------------------------------------------------------------------------------------------------------
class DropoutState:
    """Philox key material for one forward pass.  `tensor` is an int64[2] DEVICE tensor (seed, counter) that the kernels
    read; every dropout site of the pass takes the next `site` id.  Backward regenerates masks from (tensor, site).

    Seeding follows the reference's `set_seed` (utils/misc.py:37-45: `torch.manual_seed(seed + local_rank)`): unless
    `manual_seed()` was called explicitly, the per-device stream is keyed by `torch.initial_seed()` at the moment the first
    training-mode forward creates it, so different `--seed`s / ranks draw independent masks.  `get_state()` / `set_state()`
    carry (seed, counter) through checkpoints so a resumed run continues the stream instead of replaying it."""

    _global = {}
    seed = None          # None: derive from torch.initial_seed() when a device's stream is first created

    def __init__(self, device):
        device = torch.device(device)
        g = DropoutState._global.get(device)
        if g is None:
            seed = DropoutState.seed if DropoutState.seed is not None else (torch.initial_seed() & 0x7FFFFFFFFFFFFFFF)
            g = torch.tensor([seed, 0], dtype=torch.int64, device=device)
            DropoutState._global[device] = g
        self.tensor = g.clone()          # frozen copy for this forward (and its backward)
        g[1] += 1                        # device-side increment: graph-capturable, no host sync
        self._site = 0

    def next_site(self) -> int:
        self._site += 1
        return self._site

    @classmethod
    def manual_seed(cls, seed):
        """Explicit seed for every device's stream (restarts the counters).  `None` returns to following torch.initial_seed()."""
        cls.seed = None if seed is None else int(seed) & 0x7FFFFFFFFFFFFFFF
        cls._global.clear()

    @classmethod
    def get_state(cls) -> dict:
        """{device string: (seed, counter)} of every mask stream THIS process owns (one per device it ran a training-mode forward on; one
        process per GPU: normally a single entry) -- one device->host read per entry, at checkpoint time only."""
        return {str(dev): tuple(int(v) for v in t.tolist()) for dev, t in cls._global.items()}

    @classmethod
    def set_state(cls, state: dict, device=None) -> None:
        """Restore a saved stream onto `device` -- default: the device of the one stream this process already owns, else the caller's
        current device -- whatever device string the checkpoint carries (one process per GPU: a checkpoint written by rank 0 on cuda:0 is
        resumed by rank k on cuda:k and must not make it touch cuda:0 -- ADVICE r2).  The entry saved under the target's own device string
        is preferred; otherwise the first saved entry is taken and, when the device indices differ, its seed is shifted by the index
        difference so that the ranks' mask streams stay distinct the way `set_seed(seed + local_rank)` made them (utils/misc.py:37-45).
        Pass the model's device when it is not the current one."""
        if not state or not torch.cuda.is_available():
            return
        if device is not None:
            cur = torch.device(device)
            if cur.type == "cuda" and cur.index is None:
                cur = torch.device("cuda", torch.cuda.current_device())
        elif len(cls._global) == 1:
            cur = next(iter(cls._global))
        else:
            cur = torch.device("cuda", torch.cuda.current_device())
            if cls._global and cur not in cls._global:
                import warnings
                warnings.warn(f"DropoutState.set_state: restoring onto the current device {cur} while this process owns streams on "
                              f"{sorted(map(str, cls._global))}; pass device= to choose")
        key = str(cur) if str(cur) in state else sorted(state)[0]
        seed, counter = state[key]
        saved = torch.device(key)
        shift = (cur.index or 0) - (saved.index or 0) if saved.type == "cuda" else 0
        cls._global[cur] = torch.tensor([(int(seed) + shift) & 0x7FFFFFFFFFFFFFFF, int(counter)], dtype=torch.int64, device=cur)


# ------------------------------------------------------------------------------------------------------------------
# two HIP streams for the two streams of the model
# ------------------------------------------------------------------------------------------------------------------
class TwoStream:
    """The text stream and the image stream of ViLBERT are independent between co-attention layers (vilbert.py:737-811: T0..T5 need
    nothing from the image side, V_i and T_6+i both start from the outputs of co-layer i, and behind the co-attention itself each side's
    BertBiOutput + FFN is its own chain).  With `set_two_stream(True)` the text-side launches go to a second HIP stream: the 4480-row
    text GEMMs fill 82 % of the chip's CUs in their last round of tiles and the small LayerNorm / attention launches of that side leave
    most of it idle -- the other side's workgroups take those CUs.  Every kernel is still the same deterministic launch and the Python
    call order (hence every dropout site id) is unchanged, so results are bit-identical to the one-stream run; autograd runs each
    node's backward on the stream of its forward, so the backward pass splits the same way, and under hipGraph capture the two
    streams become two branches of the graph.

    Allocator safety: a tensor produced on one stream and read on the other is `record_stream`ed by `side()` / `join()` (gradients
    crossing streams in the backward pass are recorded by the autograd engine)."""

    enabled = True
    _side = {}
    _last_main = {}      # device -> the stream the last region was opened on
    _quiet = False
    active = False
    main = None          # the stream the region was opened on
    fp = None            # event on the main stream that covers everything the text side may read from it

    @classmethod
    def side_stream(cls, device) -> "torch.cuda.Stream":
        device = torch.device(device)
        if device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        s = cls._side.get(device)
        if s is None:
            s = cls._side[device] = torch.cuda.Stream(device=device)
        return s

    @classmethod
    def begin(cls, device) -> bool:
        """Start a forked region on the caller's current stream; False when the mode is off or a region is already open."""
        if not cls.enabled or cls.active or torch.device(device).type != "cuda":
            return False
        if not cls._quiet:
            # a parameter both sides use (the word embedding tied to the language decoder) gets gradients from both streams by design: the
            # engine orders them; its advice to "initialise DDP under the same stream" does not apply
            quiet = getattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch", None)
            if quiet is not None:
                quiet(False)
            cls._quiet = True
        cls.active = True
        cls.main = torch.cuda.current_stream()
        cls._last_main[cls.main.device] = cls.main
        cls.mark()
        return True

    @classmethod
    def mark(cls) -> None:
        """New fork point: the text side may read whatever the main stream has been given up to here."""
        if cls.active:
            ev = torch.cuda.Event()
            ev.record(cls.main)
            cls.fp = ev

    @classmethod
    def side(cls, *inputs):
        """Context: launches inside go to the side stream, ordered behind the last fork point.  `inputs` = tensors the body reads that
        were produced on the main stream."""
        if not cls.active:
            return contextlib.nullcontext()
        s = cls.side_stream(cls.main.device)
        s.wait_event(cls.fp)
        for t in inputs:
            if t is not None and t.is_cuda:
                t.record_stream(s)
        return torch.cuda.stream(s)

    @classmethod
    def join(cls, *outs) -> None:
        """The main stream waits for the text side; `outs` = tensors produced there that the main stream reads next."""
        if not cls.active:
            return
        main = cls.main
        main.wait_stream(cls.side_stream(main.device))
        for t in outs:
            if t is not None and t.is_cuda:
                t.record_stream(main)

    @classmethod
    @contextlib.contextmanager
    def shared_write(cls):
        """Context for a write to memory that BOTH sides read and that is not part of the forward's dataflow (the lazily created / refreshed
        bf16 copy of the weight arena): inside a two-stream region the write is enqueued on the main stream, becomes the new fork point,
        and a caller that is currently on the side stream waits for it too.  Outside a region: a plain pass-through."""
        if not cls.active:
            yield
            return
        cur = torch.cuda.current_stream()
        with torch.cuda.stream(cls.main):
            yield
        cls.mark()
        if cur != cls.main:
            cur.wait_event(cls.fp)

    @classmethod
    def gather_streams(cls) -> None:
        """Make the CURRENT stream wait for both sides of the model.  For code that runs inside a backward pass (a gradient hook runs on the
        stream of the parameter it fires for) and is about to hand a block of gradients that came from BOTH sides to something ordered
        behind the current stream only: the bucket collectives of ytvln.distributed."""
        if not cls.enabled or not cls._side or torch.cuda.is_current_stream_capturing():
            return
        cur = torch.cuda.current_stream()
        for s in (cls._side.get(cur.device), cls._last_main.get(cur.device)):
            if s is not None and s != cur:
                cur.wait_stream(s)

    @classmethod
    def join_backward(cls) -> None:
        """Call after a backward pass that ran (partly) on the side stream and before anything reads the gradients on the current
        stream.  The autograd engine joins the streams of the leaves it ran at the end of a backward pass, but a parameter's
        AccumulateGrad node keeps the stream it was CREATED on: one pinned earlier on the null stream (a post-accumulate hook registered
        before the model ran) takes no part in a stream capture, and the text side's last kernels would then have no edge to the
        capturing stream.  This join does not depend on where those nodes live.  No-op when the side stream has nothing in flight for
        the current capture."""
        if not cls.enabled or not cls._side or not torch.cuda.is_available():
            return
        cur = torch.cuda.current_stream()
        s = cls._side.get(cur.device)
        if s is None:
            return
        if torch.cuda.is_current_stream_capturing():
            with torch.cuda.stream(s):
                if not torch.cuda.is_current_stream_capturing():
                    return
        cur.wait_stream(s)

    @classmethod
    def end(cls, *outs) -> None:
        cls.join(*outs)
        cls.active = False
        cls.fp = cls.main = None


def set_two_stream(flag: bool) -> None:
    """Text-side launches on a second HIP stream (see TwoStream).  Process-wide switch, read at the start of a model forward."""
    TwoStream.enabled = bool(flag)


def get_two_stream() -> bool:
    return TwoStream.enabled


# ------------------------------------------------------------------------------------------------------------------
# GEMM primitives
# ------------------------------------------------------------------------------------------------------------------
_WS_CACHE = {}


_MATMUL_PRECISION = "fp32"


def set_matmul_precision(mode: str) -> None:
    """"fp32" (default, the reference's arithmetic: v_mfma_f32_32x32x2_f32), "bf16" or "fp32x3".
    "bf16" (BASELINE configs[4]): the bf16-RESIDENT path -- hidden states, their gradients and a copy of the weights live in HBM as bf16;
    projections (ytvln_gemm_bf16) and attention (ytvln_attn_*_bf16) read them in place and write bf16, LayerNorm reads / writes bf16 rows;
    accumulation, softmax, LayerNorm statistics, logits, losses, master weights, weight gradients and the optimizer stay fp32.
    "fp32x3": dense projections keep fp32 operands and split every value exactly into three bf16 terms in registers; each product is
    accumulated in fp32 from the six largest cross terms on the bf16 matrix instruction (YTVLN_GEMM_SPLIT_BF16X3: error per product
    of the order of one fp32 rounding).  Attention and everything else as in "fp32".  Process-wide switch, read at call time."""
    global _MATMUL_PRECISION
    if mode not in ("fp32", "bf16", "fp32x3"):
        raise ValueError(f"matmul precision must be 'fp32', 'bf16' or 'fp32x3', got {mode!r}")
    _MATMUL_PRECISION = mode


def get_matmul_precision() -> str:
    return _MATMUL_PRECISION


def _bf16_eligible(M: int, N: int, K: int) -> bool:
    """bf16 mode: does a projection whose INPUT arrives as fp32 (network inputs, pooled vectors) enter the bf16-resident path?  Large ones
    do (the region-feature projection); the one- and two-column heads on the pooled vectors stay on the fp32 kernels."""
    return _MATMUL_PRECISION == "bf16" and M >= 64 and N >= 64 and K >= 64


def _gemm(A, lda, transA, B, ldb, transB, C, ldc, M, N, K, bias=None, aux=None, ldaux=0, epi=EPI_NONE, beta=0.0, flags=0, rowsum=None):
    """fp32 operands (native fp32 matrix instruction, or the three-bf16-term form under "fp32x3").  `rowsum` = an [M] tensor that should
    receive sum_k op(A)[m, k] (the bias gradient riding on a weight-gradient GEMM, ytvln_gemm_f32_rowsum); returns True when the launch
    produced it, False when the caller has to run `colsum` itself."""
    sk = _lib.option("GEMM_SK") != 0          # the persistent kernel (opt-in): its partial-tile scratch and control block exist only then
    key = (M, N, K, epi, sk)
    need = _WS_CACHE.get(key)
    if need is None:
        need = _WS_CACHE[key] = int(_lib.load().ytvln_gemm_workspace_elems(M, N, K, epi))
    ws = torch.empty(need, dtype=torch.float32, device=C.device) if need else None      # split-K / stream-K scratch (caching allocator)
    if _MATMUL_PRECISION == "fp32x3":
        flags = int(flags) | GEMM_SPLIT_BF16X3
    ctl = _sk_ctl(C.device) if sk else None
    if rowsum is not None:
        done = ctypes.c_int(0)
        call("ytvln_gemm_f32_sk", _ptr(A), lda, int(transA), _ptr(B), ldb, int(transB), _ptr(C), ldc, _ptr(bias), _ptr(aux), ldaux,
             M, N, K, epi, float(beta), _ptr(ws), need, int(flags), _ptr(rowsum), ctypes.byref(done), _ptr(ctl), _stream())
        return bool(done.value)
    call("ytvln_gemm_f32_sk", _ptr(A), lda, int(transA), _ptr(B), ldb, int(transB), _ptr(C), ldc, _ptr(bias), _ptr(aux), ldaux,
         M, N, K, epi, float(beta), _ptr(ws), need, int(flags), None, None, _ptr(ctl), _stream())
    return False


_SK_CTL = {}


def _sk_ctl(device):
    """Control block of the persistent GEMM (ytvln_gemm_f32_sk: tickets, done counters, partial-tile flags; zero between launches).  A block
    must never be shared by two launches that can run at the same time, so there is one per device and ROLE: the side stream of
    `TwoStream` and everything else (eager, captured and replayed launches of a role are ordered among themselves).  Created outside
    stream capture; a first use inside a capture gets None = the launch-per-tile kernel."""
    pair = _SK_CTL.get(device)
    if pair is None:
        if torch.cuda.is_current_stream_capturing():
            return None
        n = int(_lib.load().ytvln_gemm_sk_ctl_elems())
        # (block of the main role, block of the side role, the stream that owns the main role = the one the first launch was enqueued on)
        pair = _SK_CTL[device] = (torch.zeros(n, dtype=torch.int32, device=device), torch.zeros(n, dtype=torch.int32, device=device), _stream())
        torch.cuda.synchronize(device)
    side = TwoStream._side.get(device)
    cur = _stream()
    if side is not None and cur == side.cuda_stream:
        return pair[1]
    # any other stream than the two roles (a user stream, a communication stream, a second graph): two launches sharing one block could run at
    # the same time -- those take the launch-per-tile kernel
    return pair[0] if cur == pair[2] else None


def colsum(x: Tensor, M: int, N: int, ld: int, out: Optional[Tensor] = None) -> Tensor:
    """Deterministic column sums of the [M, N] matrix at x (leading dimension ld) -> [N] (written into `out` if given,
    e.g. a gradient-arena slot)."""
    first = True
    while True:
        nstrips = (N + 255) // 256          # the kernel sweeps 256-column strips (16 bytes per lane) whenever N % 4 == 0
        nb = max(1, min((M + 31) // 32, max(1, 1024 // nstrips))) if first else 1      # two stages at most
        first = False
        rpb = (M + nb - 1) // nb
        nb = (M + rpb - 1) // rpb
        if nb == 1:
            res = out if out is not None else torch.empty(N, dtype=torch.float32, device=x.device)
            call("ytvln_colsum_f32", _ptr(x), ld, M, N, _ptr(res), N, rpb, _stream())
            return res
        part = torch.empty((nb, N), dtype=torch.float32, device=x.device)
        call("ytvln_colsum_f32", _ptr(x), ld, M, N, _ptr(part), N, rpb, _stream())
        x, M, ld = part, nb, N


def colsum_by_index(x: Tensor, M: int, N: int, ld: int, KT: int, idx_f32: Optional[Tensor] = None, idx_stride: int = 1,
                    idx_f32_offset: int = 0, idx_i64: Optional[Tensor] = None) -> Tensor:
    """out[k, n] = sum of rows r with idx[r] == k -> [KT, N] (KT <= 32)."""
    nstrips = (N + 63) // 64
    nb = max(1, min((M + 31) // 32, max(1, 1024 // nstrips)))
    rpb = (M + nb - 1) // nb
    nb = (M + rpb - 1) // rpb
    out = torch.empty((nb, KT, N), dtype=torch.float32, device=x.device)
    call("ytvln_colsum_by_index_f32", _ptr(x), ld, _ptr(idx_f32, idx_f32_offset) if idx_f32 is not None else None, idx_stride,
         _ptr(idx_i64) if idx_i64 is not None else None, M, N, KT, _ptr(out), rpb, _stream())
    if nb == 1:
        return out[0]
    return colsum(out, nb, KT * N, KT * N).view(KT, N)


# ------------------------------------------------------------------------------------------------------------------
# flat-arena hooks (registered by ytvln.optimization.AdamW once parameters / gradients live in flat arenas)
# ------------------------------------------------------------------------------------------------------------------
class ArenaSlot:
    """Where a parameter lives inside the optimizer's flat arenas.  Attached to the Parameter object as `_ytvln_slot` by
    ytvln.optimization.AdamW; `written` (shared per arena) records which gradient slots a GEMM has already written directly
    since the last optimizer step / zero_grad, so a weight used twice before one backward falls back to the additive path."""
    __slots__ = ("flat_p", "flat_g", "off", "numel", "written", "owner")

    def __init__(self, flat_p, flat_g, off, numel, written, owner=None):
        self.flat_p, self.flat_g, self.off, self.numel, self.written = flat_p, flat_g, off, numel, written
        self.owner = owner          # weakref to the optimizer that owns the arenas (its bf16 weight arena: bf16_arena())

    def valid_for(self, t) -> bool:
        return t.numel() == self.numel and t.data_ptr() == self.flat_p.data_ptr() + 4 * self.off


def _slot_of(t):
    sl = getattr(t, "_ytvln_slot", None)
    return sl if (sl is not None and sl.valid_for(t)) else None


def _targets_of(weight):
    """The parameters a weight operand stands for: itself, or the members of a packed (concatenated) view."""
    if isinstance(weight, torch.nn.Parameter):
        return (weight,)
    return getattr(weight, "_ytvln_pack_params", None)


def _direct_grad(targets, shape):
    """A fresh view of the flat gradient arena covering `targets` (adjacent there, none of them holding a gradient yet and
    none written directly since the last step), shaped `shape` -- the weight-gradient GEMM then writes its result straight
    into the arena and autograd's AccumulateGrad adopts the view without a copy or a read-modify-write pass.  None when the
    conditions do not hold (the caller then allocates a normal gradient tensor)."""
    if not targets:
        return None
    first = _slot_of(targets[0])
    if first is None:
        return None
    off = first.off
    for t in targets:
        sl = _slot_of(t)
        if sl is None or sl.flat_g is not first.flat_g or sl.off != off or t.grad is not None or sl.off in sl.written:
            return None
        off += sl.numel
    for t in targets:
        first.written.add(t._ytvln_slot.off)
    return first.flat_g[first.off:off].view(shape)


class PackRowsFn(torch.autograd.Function):
    """cat(params, dim=0) for the packed Q|K|V (or K|V) projection.  Once the optimizer has moved the parameters into its
    flat arena the members are adjacent in memory and the packed operand is a zero-copy view; the backward hands each member
    its row block of the packed gradient as a view."""

    @staticmethod
    def forward(ctx, *ts):
        ctx.set_materialize_grads(False)
        ctx.rows = [t.shape[0] for t in ts]
        first = _slot_of(ts[0])
        if first is not None:
            off, ok = first.off, True
            for t in ts:
                sl = _slot_of(t)
                if sl is None or sl.flat_p is not first.flat_p or sl.off != off:
                    ok = False
                    break
                off += sl.numel
            if ok:
                return first.flat_p[first.off:off].view((sum(ctx.rows),) + tuple(ts[0].shape[1:]))
        return torch.cat([t.detach() for t in ts], dim=0)

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return (None,) * len(ctx.rows)
        out, r0 = [], 0
        for r in ctx.rows:
            out.append(g.narrow(0, r0, r))
            r0 += r
        return tuple(out)


def pack_rows(*params):
    w = PackRowsFn.apply(*params)
    w._ytvln_pack_params = tuple(params)
    return w


# ------------------------------------------------------------------------------------------------------------------
# wide, oddly sized outputs (30522-way language logits, 1601-way vision logits) live in buffers whose leading dimension is
# rounded up to 32 floats: rows stay 16-byte aligned and their gradients can feed the LDS-DMA GEMM (zero-padded K tail).
# ------------------------------------------------------------------------------------------------------------------
import weakref as _weakref

_ZERO_PADDED = {}       # storage data_ptr -> leading dimension, for gradient buffers whose pad columns are known to be zero


def _pad_ld(n: int) -> int:
    return (n + 31) // 32 * 32 if (n >= 1024 and n % 32) else n


def _alloc_rows(M: int, N: int, device, zero_pad: bool = False):
    """[M, N] fp32 view with leading dimension _pad_ld(N); with zero_pad the pad columns are cleared and remembered."""
    ld = _pad_ld(N)
    if ld == N:
        return torch.empty((M, N), dtype=torch.float32, device=device), N
    full = torch.empty((M, ld), dtype=torch.float32, device=device)
    if zero_pad:
        full[:, N:].zero_()
        key = full.untyped_storage().data_ptr()
        _ZERO_PADDED[key] = ld
        _weakref.finalize(full, _ZERO_PADDED.pop, key, None)      # `full` stays alive as the base of the returned view
    return full[:, :N], ld


def _view_rows_as(rows: Tensor, ld: int, shape):
    """View a [M, N] row buffer with leading dimension ld as `shape` (= (*lead, N))."""
    shape = tuple(shape)
    if ld == shape[-1]:
        return rows.view(shape)
    strides, acc = [], ld
    for dim in reversed(shape[:-1]):
        strides.append(acc)
        acc *= dim
    return torch.as_strided(rows, shape, tuple(reversed(strides)) + (1,))


def _rows_of_grad(dy: Tensor, M: int, N: int):
    """(dy as [M, N] rows, leading dimension, GEMM flags): keeps a padded-leading-dimension gradient in place."""
    d2 = dy.reshape(M, N)
    if d2.stride(1) == 1 and d2.stride(0) > N and d2.stride(0) % 4 == 0 and d2.data_ptr() % 16 == 0:
        known = _ZERO_PADDED.get(d2.untyped_storage().data_ptr()) == d2.stride(0) and d2.storage_offset() == 0
        return d2, d2.stride(0), (GEMM_A_ZERO_PADDED if known else 0), known
    if not d2.is_contiguous():
        d2 = d2.contiguous()
    return d2, N, 0, False


def _overlaps(a: Tensor, b: Tensor) -> bool:
    if a is None or b is None or a.untyped_storage().data_ptr() != b.untyped_storage().data_ptr():
        return False
    a0, b0 = a.data_ptr(), b.data_ptr()
    return a0 < b0 + b.numel() * b.element_size() and b0 < a0 + a.numel() * a.element_size()


def _accumulate_target(dres, M: int, K: int, *live):
    """The [M, K] row view of a residual-branch gradient the input-gradient GEMM may accumulate into (beta = 1), or None.
    `live`: tensors this backward still reads (the output gradient): with dropout p = 0 AddLayerNormFn hands the SAME storage to the
    sublayer-output gradient and to the residual gradient, and accumulating in place would overwrite an operand that is being read
     -- the caller then falls back to a separate buffer + add."""
    if dres is None or dres.dtype != torch.float32 or not dres.is_cuda or dres.numel() != M * K or not dres.is_contiguous():
        return None
    if any(_overlaps(dres, t) for t in live):
        return None
    return dres.view(M, K)


class LinearFn(torch.autograd.Function):
    """y = act(x W^T + b)  -- nn.Linear + optional erf-GELU / ReLU epilogue, all on the fp32 MFMA GEMM.

    `passthrough=True` returns (y, x): the second output is x itself, to be used by the RESIDUAL connection that skips the sublayer this
    projection opens (vilbert.py:322-325, 365-368: `LayerNorm(dropout(dense(...)) + input_tensor)`).  The residual branch's gradient then
    arrives here as a second output gradient and the input-gradient GEMM accumulates into that buffer (beta = 1) instead of autograd
    launching a separate elementwise add for the fan-out of x."""

    @staticmethod
    def forward(ctx, x, weight, bias, act, passthrough=False):
        ctx.set_materialize_grads(False)
        ctx.passthrough = bool(passthrough)
        x2, M, K, lda = _rows2d(x, "x")
        _check(weight, "weight")
        if weight.stride(-1) != 1:
            weight = weight.contiguous()
        N = weight.shape[0]
        assert weight.shape[1] == K, (weight.shape, K)
        need_grad = any(ctx.needs_input_grad)
        epi = _ACT[act]
        if epi == EPI_NONE:
            y, ldy = _alloc_rows(M, N, x.device)          # wide odd widths (logits) get a 32-float aligned leading dimension
        else:
            y, ldy = torch.empty((M, N), dtype=torch.float32, device=x.device), N
        z = torch.empty_like(y) if (epi == EPI_GELU and need_grad) else None
        _gemm(x2, lda, 0, weight, weight.stride(0), 1, y, ldy, M, N, K, bias=bias, aux=z, ldaux=N, epi=epi)
        ctx.epi, ctx.dims, ctx.lda, ctx.has_bias = epi, (M, N, K), lda, bias is not None
        ctx.in_shape = x.shape
        ctx.targets = _targets_of(weight)
        ctx.btargets = _targets_of(bias) if bias is not None else None
        ctx.save_for_backward(x2, weight, z if epi == EPI_GELU else (y if epi == EPI_RELU else None))
        out = _view_rows_as(y, ldy, tuple(x.shape[:-1]) + (N,))
        return (out, x) if ctx.passthrough else out

    @staticmethod
    def backward(ctx, dy, dres=None):
        if dy is None:
            return (dres if ctx.needs_input_grad[0] else None), None, None, None, None
        x2, weight, aux = ctx.saved_tensors
        M, N, K = ctx.dims
        if ctx.epi != EPI_NONE:
            dy = dy.reshape(M, N)
            if not dy.is_contiguous():
                dy = dy.contiguous()
            dz = torch.empty_like(dy)
            call("ytvln_act_bwd_f32", _ptr(dy), _ptr(aux), _ptr(dz), dy.numel(), ctx.epi, _stream())
            dy, ldy, flags = dz, N, 0
        else:
            dy, ldy, flags, _ = _rows_of_grad(dy, M, N)
        dx = dw = db = None
        want_db = ctx.has_bias and ctx.needs_input_grad[2]
        if want_db:
            db = _direct_grad(ctx.btargets, (N,))
            if db is None:
                db = torch.empty(N, dtype=torch.float32, device=dy.device)
        db_done = False
        if ctx.needs_input_grad[0]:
            acc = _accumulate_target(dres, M, K, dy)
            if acc is not None:          # dx = d(residual) + dY W, written over the residual branch's gradient
                _gemm(dy, ldy, 0, weight, weight.stride(0), 0, acc, K, M, K, N, beta=1.0, flags=flags)
                dx = acc.view(ctx.in_shape)
            else:
                dx = torch.empty((M, K), dtype=torch.float32, device=dy.device)
                _gemm(dy, ldy, 0, weight, weight.stride(0), 0, dx, K, M, K, N, flags=flags)
                dx = dx.view(ctx.in_shape)
                if dres is not None:
                    dx = dx + dres.reshape(ctx.in_shape)
        if ctx.needs_input_grad[1]:
            dw = _direct_grad(ctx.targets, (N, K))
            if dw is None:
                dw = torch.empty((N, K), dtype=torch.float32, device=dy.device)
            # db = column sums of dY = row sums of the A operand (dY^T) of this launch: rides on the GEMM when it can
            db_done = _gemm(dy, ldy, 1, x2, ctx.lda, 0, dw, K, N, K, M, flags=flags, rowsum=db if want_db else None)
        if want_db and not db_done:
            colsum(dy, M, N, ldy, out=db)
        return dx, dw, db, None, None


def _wants_bf16(x: Tensor, weight: Tensor) -> bool:
    """bf16-resident path for this projection?  Yes when its input already is a bf16 hidden state; in bf16 mode also for a LARGE projection
    whose input arrives as fp32 (the 2048-d region features: cast once on the way in)."""
    if x.dtype == torch.bfloat16:
        return True
    K = x.shape[-1]
    return x.dtype == torch.float32 and _bf16_eligible(x.numel() // max(K, 1), weight.shape[0], K)


def linear(x: Tensor, weight: Tensor, bias: Optional[Tensor] = None, act: Optional[str] = None, out_fp32: bool = False) -> Tensor:
    """`out_fp32` only matters on the bf16-resident path: the output leaves the path as fp32 (the pooled vectors and the small heads on them)."""
    if _wants_bf16(x, weight):
        return LinearBf16Fn.apply(x, weight, bias, act, False, out_fp32)
    return LinearFn.apply(x, weight, bias, act)


def linear_res(x: Tensor, weight: Tensor, bias: Optional[Tensor] = None, act: Optional[str] = None) -> Tuple[Tensor, Tensor]:
    """(linear(x, ...), x): use the second value for the residual connection around the sublayer (see LinearFn)."""
    if _wants_bf16(x, weight):
        return LinearBf16Fn.apply(x, weight, bias, act, True, False)
    return LinearFn.apply(x, weight, bias, act, True)


class FFNFn(torch.autograd.Function):
    """out = gelu(x W1^T + b1) W2^T + b2   (BertIntermediate + BertOutput.dense, vilbert.py:351-354, 365).
    Backward fuses the GELU derivative into the epilogue of the dX GEMM of the second projection."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, passthrough=False):
        ctx.set_materialize_grads(False)
        ctx.passthrough = bool(passthrough)
        x2, M, K, lda = _rows2d(x, "x")
        I, N = w1.shape[0], w2.shape[0]
        need_grad = any(ctx.needs_input_grad)
        h = torch.empty((M, I), dtype=torch.float32, device=x.device)
        z = torch.empty_like(h) if need_grad else None
        _gemm(x2, lda, 0, w1, w1.stride(0), 1, h, I, M, I, K, bias=b1, aux=z, ldaux=I, epi=EPI_GELU)
        y = torch.empty((M, N), dtype=torch.float32, device=x.device)
        _gemm(h, I, 0, w2, w2.stride(0), 1, y, N, M, N, I, bias=b2)
        ctx.dims, ctx.lda, ctx.in_shape = (M, K, I, N), lda, x.shape
        ctx.targets = (_targets_of(w1), _targets_of(w2), _targets_of(b1), _targets_of(b2))
        ctx.save_for_backward(x2, w1, w2, z, h)
        out = y.view(*x.shape[:-1], N)
        return (out, x) if ctx.passthrough else out

    @staticmethod
    def backward(ctx, dy, dres=None):
        if dy is None:
            return (dres if ctx.needs_input_grad[0] else None), None, None, None, None, None
        x2, w1, w2, z, h = ctx.saved_tensors
        M, K, I, N = ctx.dims
        dy = dy.reshape(M, N)
        if not dy.is_contiguous():
            dy = dy.contiguous()
        dev = dy.device
        dz = torch.empty((M, I), dtype=torch.float32, device=dev)
        db2 = _direct_grad(ctx.targets[3], (N,))
        if db2 is None:
            db2 = torch.empty(N, dtype=torch.float32, device=dev)
        db1 = _direct_grad(ctx.targets[2], (I,))
        if db1 is None:
            db1 = torch.empty(I, dtype=torch.float32, device=dev)
        _gemm(dy, N, 0, w2, w2.stride(0), 0, dz, I, M, I, N, aux=z, ldaux=I, epi=EPI_MUL_DGELU)   # dH * gelu'(z)
        dw2 = _direct_grad(ctx.targets[1], (N, I))
        if dw2 is None:
            dw2 = torch.empty((N, I), dtype=torch.float32, device=dev)
        if not _gemm(dy, N, 1, h, I, 0, dw2, I, N, I, M, rowsum=db2):          # db2 rides on the dW2 GEMM
            colsum(dy, M, N, N, out=db2)
        dx = None
        if ctx.needs_input_grad[0]:
            acc = _accumulate_target(dres, M, K, dy)      # (dy is consumed by now, but it may be retained / hooked upstream)
            if acc is not None:          # the residual around the FFN: accumulate into its gradient (see LinearFn)
                _gemm(dz, I, 0, w1, w1.stride(0), 0, acc, K, M, K, I, beta=1.0)
                dx = acc.view(ctx.in_shape)
            else:
                dx = torch.empty((M, K), dtype=torch.float32, device=dev)
                _gemm(dz, I, 0, w1, w1.stride(0), 0, dx, K, M, K, I)
                dx = dx.view(ctx.in_shape)
                if dres is not None:
                    dx = dx + dres.reshape(ctx.in_shape)
        dw1 = _direct_grad(ctx.targets[0], (I, K))
        if dw1 is None:
            dw1 = torch.empty((I, K), dtype=torch.float32, device=dev)
        if not _gemm(dz, I, 1, x2, ctx.lda, 0, dw1, K, I, K, M, rowsum=db1):
            colsum(dz, M, I, I, out=db1)
        return dx, dw1, db1, dw2, db2, None


def ffn(x, w1, b1, w2, b2) -> Tensor:
    if x.dtype == torch.bfloat16:
        return FFNBf16Fn.apply(x, w1, b1, w2, b2)
    return FFNFn.apply(x, w1, b1, w2, b2)


def ffn_res(x, w1, b1, w2, b2) -> Tuple[Tensor, Tensor]:
    """(ffn(x, ...), x) -- the second value feeds the residual add behind the FFN (see LinearFn)."""
    if x.dtype == torch.bfloat16:
        return FFNBf16Fn.apply(x, w1, b1, w2, b2, True)
    return FFNFn.apply(x, w1, b1, w2, b2, True)



# ------------------------------------------------------------------------------------------------------------------
# bf16-resident path (BASELINE configs[4]; set_matmul_precision("bf16")): hidden states, their gradients and a copy of the weights
# are bf16 tensors in HBM; ytvln_gemm_bf16 / ytvln_attn_*_bf16 / ytvln_ln_*_bf16 read and write them in place.  No operand is
# staged, cast or transposed per call: the producer of a tensor writes the bf16 its consumers read.
# ------------------------------------------------------------------------------------------------------------------
_BF16 = torch.bfloat16
_BF16_WS_CACHE = {}
_DT_CODE = {torch.float32: _lib.DT_F32, torch.bfloat16: _lib.DT_BF16}


def _pad8(n: int) -> int:
    """Leading dimension (elements) of a bf16 row buffer whose width is not a multiple of 8: rounded up to 64, the padding ZERO."""
    return n if n % 8 == 0 else (n + 63) // 64 * 64


def cast_bf16(x: Tensor) -> Tensor:
    """bf16 copy (round to nearest even) of a contiguous fp32 tensor: network inputs entering the bf16-resident path, parameters outside the
    optimizer's arenas, small fp32 gradients coming back from the fp32 heads."""
    _check(x, "x")
    xc = x if x.is_contiguous() else x.contiguous()
    out = torch.empty(xc.shape, dtype=_BF16, device=x.device)
    cols = xc.shape[-1] if xc.dim() > 1 else xc.numel()
    rows = xc.numel() // max(cols, 1)
    if xc.numel():
        call("ytvln_cast_f32_bf16", _ptr(xc), cols, rows, cols, out.data_ptr(), cols, _stream())
    return out


def _bf16_weight(weight: Tensor) -> Tensor:
    """The bf16 copy of a weight operand ([out, in], contiguous).  Parameters that live in the optimizer's flat arena (after the first
    optimizer step: all that receive gradients) have their copy in a bf16 arena at the same offsets, refreshed by the AdamW kernel itself
    (ytvln_adamw_f32_bf16copy) -- a packed Q|K|V operand is then a zero-copy view of it.  Everything else (the first eager steps, frozen
    or inference-only parameters) is cast on the spot and cached against the parameter's version counter."""
    targets = _targets_of(weight)
    if targets:
        first = _slot_of(targets[0])
        if first is not None and first.owner is not None:
            off, ok = first.off, True
            for t in targets:
                sl = _slot_of(t)
                if sl is None or sl.flat_p is not first.flat_p or sl.off != off:
                    ok = False
                    break
                off += sl.numel
            opt = first.owner() if ok else None
            if opt is not None:
                pb = opt.bf16_arena(targets)
                if pb is not None:
                    return pb[first.off:off].view(weight.shape)
        # not (all) adjacent in one arena.  Arena members are updated by the fused AdamW kernel through raw pointers, which does not bump
        # their version counters: a cache keyed on versions would go stale after every optimizer step, so they are cast on the spot.
        if any(_slot_of(t) is not None for t in targets):
            return cast_bf16(weight.detach())
        # per-parameter cache keyed by (storage pointer, version)
        key = tuple((t.data_ptr(), t._version) for t in targets)
        holder = targets[0]
        cached = getattr(holder, "_ytvln_bf16_cache", None)
        if cached is not None and cached[0] == key and cached[1].shape == weight.shape:
            return cached[1]
        wb = cast_bf16(weight.detach())
        if not torch.cuda.is_current_stream_capturing():
            holder._ytvln_bf16_cache = (key, wb)
        return wb
    return cast_bf16(weight.detach())


def _gemm_bf16(A, lda, transA, B, ldb, transB, C, ldc, M, N, K, bias=None, aux=None, ldaux=0, epi=EPI_NONE, beta=0.0, flags=0, rowsum=None):
    """ytvln_gemm_bf16: bf16 operands read in place (transA = 1 / transB = 0: the contraction index is the operand's ROW, gathered by the
    transposing LDS read), fp32 accumulation, C bf16 or fp32.  Returns True when `rowsum` ([M] fp32: sum_k op(A)[m, k]) was produced."""
    key = (M, N, K, epi)
    need = _BF16_WS_CACHE.get(key)
    if need is None:
        need = _BF16_WS_CACHE[key] = int(_lib.load().ytvln_gemm_bf16_workspace_elems(M, N, K, epi))
    ws = torch.empty(need, dtype=torch.float32, device=C.device) if need else None
    done = ctypes.c_int(0)
    call("ytvln_gemm_bf16", A.data_ptr(), lda, int(transA), B.data_ptr(), ldb, int(transB), C.data_ptr(), ldc, _DT_CODE[C.dtype], _ptr(bias),
         aux.data_ptr() if aux is not None else None, ldaux, M, N, K, epi, float(beta), _ptr(ws), need, int(flags),
         _ptr(rowsum) if rowsum is not None else None, ctypes.byref(done) if rowsum is not None else None, _stream())
    return bool(done.value)


def _rows2d_bf16(x: Tensor) -> Tuple[Tensor, int, int, int]:
    """View a bf16 tensor as [M, K] rows with unit inner stride and a leading dimension that is a multiple of 8."""
    K = x.shape[-1]
    if x.dim() == 2 and x.stride(1) == 1 and x.stride(0) >= K and x.stride(0) % 8 == 0 and x.data_ptr() % 16 == 0:
        return x, x.shape[0], K, x.stride(0)
    if not x.is_contiguous():
        x = x.contiguous()
    return x, x.numel() // max(K, 1), K, K


def _grad_rows_bf16(dy: Tensor, M: int, N: int):
    """(dy as bf16 [M, N] rows, leading dimension, GEMM flags).  bf16 gradients with a padded leading dimension come from the loss kernels
    (ytvln_ce_bwd_bf16 / ytvln_kl_bwd_bf16: zero padding written by the kernel) and are used in place; fp32 gradients (the small heads on the
    pooled vectors) are cast; anything unaligned is repacked into a zero-padded buffer."""
    if dy.dtype == torch.float32:
        dy = cast_bf16(dy.reshape(M, N))
    d2 = dy.reshape(M, N)
    if d2.stride(1) == 1 and d2.stride(0) % 8 == 0 and d2.data_ptr() % 16 == 0 and (d2.stride(0) == N or
                                                                                   _ZERO_PADDED.get(d2.untyped_storage().data_ptr()) == d2.stride(0)):
        return d2, d2.stride(0), (GEMM_A_ZERO_PADDED if d2.stride(0) != N else 0)
    ld = _pad8(N)
    if ld == N:
        return d2.contiguous(), N, 0
    full = torch.zeros((M, ld), dtype=_BF16, device=dy.device)
    full[:, :N].copy_(d2)
    return full[:, :N], ld, GEMM_A_ZERO_PADDED


def _alloc_rows_bf16(M: int, N: int, device):
    """[M, N] bf16 view with leading dimension _pad8(N); the caller's kernel writes zeros into the padding (and that is remembered)."""
    ld = _pad8(N)
    if ld == N:
        return torch.empty((M, N), dtype=_BF16, device=device), N
    full = torch.empty((M, ld), dtype=_BF16, device=device)
    key = full.untyped_storage().data_ptr()
    _ZERO_PADDED[key] = ld
    _weakref.finalize(full, _ZERO_PADDED.pop, key, None)
    return full[:, :N], ld


class LinearBf16Fn(torch.autograd.Function):
    """y = act(x W^T + b) on the bf16-resident path (same contract as LinearFn, incl. `passthrough`).  x: bf16 hidden states (or fp32 network
    inputs, cast once); W: the bf16 copy of the fp32 master weight; y: bf16, or fp32 when `out_fp32` (pooled vectors).  Backward:
    dX = dY W and dW = dY^T X read dY, W and X as they lie in HBM (k-major operands through the transposing LDS read); dW and db are fp32
    and go straight into the gradient arena; db rides on the dW launch."""

    @staticmethod
    def forward(ctx, x, weight, bias, act, passthrough=False, out_fp32=False):
        ctx.set_materialize_grads(False)
        ctx.passthrough = bool(passthrough)
        ctx.x_was_f32 = x.dtype == torch.float32
        xb = cast_bf16(x) if ctx.x_was_f32 else x
        _check(xb, "x", _BF16)
        x2, M, K, lda = _rows2d_bf16(xb)
        _check(weight, "weight")
        N = weight.shape[0]
        assert weight.shape[1] == K, (weight.shape, K)
        wb = _bf16_weight(weight)
        need_grad = any(ctx.needs_input_grad)
        epi = _ACT[act]
        if out_fp32:
            y, ldy = _alloc_rows(M, N, x.device)
        else:
            y, ldy = torch.empty((M, N), dtype=_BF16, device=x.device), N
        z = torch.empty((M, N), dtype=_BF16, device=x.device) if (epi == EPI_GELU and need_grad) else None
        _gemm_bf16(x2, lda, 0, wb, K, 1, y, ldy, M, N, K, bias=bias, aux=z, ldaux=N, epi=epi)
        ctx.epi, ctx.dims, ctx.lda, ctx.has_bias = epi, (M, N, K), lda, bias is not None
        ctx.in_shape = x.shape
        ctx.targets = _targets_of(weight)
        ctx.btargets = _targets_of(bias) if bias is not None else None
        # the ReLU backward needs the sign of the OUTPUT: a bf16 copy of it when the output itself leaves as fp32
        aux = z if epi == EPI_GELU else ((y if y.dtype == _BF16 else cast_bf16(y)) if (epi == EPI_RELU and need_grad) else None)
        ctx.save_for_backward(x2, weight, aux)
        out = _view_rows_as(y, ldy, tuple(x.shape[:-1]) + (N,))
        return (out, x) if ctx.passthrough else out

    @staticmethod
    def backward(ctx, dy, dres=None):
        if dy is None:
            return (dres if ctx.needs_input_grad[0] else None), None, None, None, None, None
        x2, weight, aux = ctx.saved_tensors
        M, N, K = ctx.dims
        dy, ldy, flags = _grad_rows_bf16(dy, M, N)
        if ctx.epi != EPI_NONE:
            if ldy != N:
                dy, ldy, flags = dy.contiguous(), N, 0
            dz = torch.empty((M, N), dtype=_BF16, device=dy.device)
            call("ytvln_act_bwd_bf16", dy.data_ptr(), aux.data_ptr(), dz.data_ptr(), dy.numel(), ctx.epi, _stream())
            dy = dz
        dx = dw = db = None
        want_db = ctx.has_bias and ctx.needs_input_grad[2]
        if want_db:
            db = _direct_grad(ctx.btargets, (N,))
            if db is None:
                db = torch.empty(N, dtype=torch.float32, device=dy.device)
        if ctx.needs_input_grad[0]:
            wb = _bf16_weight(weight)
            acc = None
            if not ctx.x_was_f32 and dres is not None and dres.dtype == _BF16 and dres.is_contiguous() and dres.numel() == M * K and \
                    not _overlaps(dres, dy):
                acc = dres.view(M, K)
            if acc is not None:          # dx = d(residual) + dY W, accumulated over the residual branch's gradient (beta = 1)
                _gemm_bf16(dy, ldy, 0, wb, K, 0, acc, K, M, K, N, beta=1.0, flags=flags)
                dx = acc.view(ctx.in_shape)
            else:
                dx = torch.empty((M, K), dtype=_BF16, device=dy.device)
                _gemm_bf16(dy, ldy, 0, wb, K, 0, dx, K, M, K, N, flags=flags)
                dx = dx.view(ctx.in_shape)
                if dres is not None:
                    dx = dx + dres.reshape(ctx.in_shape)
                if ctx.x_was_f32:
                    dx = dx.float()
        db_done = False
        if ctx.needs_input_grad[1]:
            dw = _direct_grad(ctx.targets, (N, K))
            if dw is None:
                dw = torch.empty((N, K), dtype=torch.float32, device=dy.device)
            # dW[N, K] = dY^T X: both operands k-major (the contraction index M is their row); db = row sums of dY^T ride on the launch
            db_done = _gemm_bf16(dy, ldy, 1, x2, ctx.lda, 0, dw, K, N, K, M, flags=flags, rowsum=db if want_db else None)
        if want_db and not db_done:
            colsum(dy.float() if ldy == N else dy.contiguous().float(), M, N, N, out=db)
        return dx, dw, db, None, None, None


class FFNBf16Fn(torch.autograd.Function):
    """out = gelu(x W1^T + b1) W2^T + b2 on the bf16-resident path (BertIntermediate + BertOutput.dense, vilbert.py:351-354, 365): x, the
    GELU pre-activation z, the hidden h and out are bf16; the backward fuses GELU' into the epilogue of the dX GEMM of the second projection."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, passthrough=False):
        ctx.set_materialize_grads(False)
        ctx.passthrough = bool(passthrough)
        _check(x, "x", _BF16)
        x2, M, K, lda = _rows2d_bf16(x)
        I, N = w1.shape[0], w2.shape[0]
        need_grad = any(ctx.needs_input_grad)
        dev = x.device
        h = torch.empty((M, I), dtype=_BF16, device=dev)
        z = torch.empty((M, I), dtype=_BF16, device=dev) if need_grad else None
        _gemm_bf16(x2, lda, 0, _bf16_weight(w1), K, 1, h, I, M, I, K, bias=b1, aux=z, ldaux=I, epi=EPI_GELU)
        y = torch.empty((M, N), dtype=_BF16, device=dev)
        _gemm_bf16(h, I, 0, _bf16_weight(w2), I, 1, y, N, M, N, I, bias=b2)
        ctx.dims, ctx.lda, ctx.in_shape = (M, K, I, N), lda, x.shape
        ctx.targets = (_targets_of(w1), _targets_of(w2), _targets_of(b1), _targets_of(b2))
        ctx.save_for_backward(x2, w1, w2, z, h)
        out = y.view(*x.shape[:-1], N)
        return (out, x) if ctx.passthrough else out

    @staticmethod
    def backward(ctx, dy, dres=None):
        if dy is None:
            return (dres if ctx.needs_input_grad[0] else None), None, None, None, None, None
        x2, w1, w2, z, h = ctx.saved_tensors
        M, K, I, N = ctx.dims
        dy, ldy, _ = _grad_rows_bf16(dy, M, N)
        dev = dy.device
        dz = torch.empty((M, I), dtype=_BF16, device=dev)
        db2 = _direct_grad(ctx.targets[3], (N,))
        if db2 is None:
            db2 = torch.empty(N, dtype=torch.float32, device=dev)
        db1 = _direct_grad(ctx.targets[2], (I,))
        if db1 is None:
            db1 = torch.empty(I, dtype=torch.float32, device=dev)
        _gemm_bf16(dy, ldy, 0, _bf16_weight(w2), I, 0, dz, I, M, I, N, aux=z, ldaux=I, epi=EPI_MUL_DGELU)   # dH * gelu'(z)
        dw2 = _direct_grad(ctx.targets[1], (N, I))
        if dw2 is None:
            dw2 = torch.empty((N, I), dtype=torch.float32, device=dev)
        if not _gemm_bf16(dy, ldy, 1, h, I, 0, dw2, I, N, I, M, rowsum=db2):
            colsum(dy.float(), M, N, N, out=db2)
        dx = None
        if ctx.needs_input_grad[0]:
            acc = None
            if dres is not None and dres.dtype == _BF16 and dres.is_contiguous() and dres.numel() == M * K and not _overlaps(dres, dy):
                acc = dres.view(M, K)
            if acc is not None:          # the residual around the FFN: accumulate into its gradient
                _gemm_bf16(dz, I, 0, _bf16_weight(w1), K, 0, acc, K, M, K, I, beta=1.0)
                dx = acc.view(ctx.in_shape)
            else:
                dx = torch.empty((M, K), dtype=_BF16, device=dev)
                _gemm_bf16(dz, I, 0, _bf16_weight(w1), K, 0, dx, K, M, K, I)
                dx = dx.view(ctx.in_shape)
                if dres is not None:
                    dx = dx + dres.reshape(ctx.in_shape)
        dw1 = _direct_grad(ctx.targets[0], (I, K))
        if dw1 is None:
            dw1 = torch.empty((I, K), dtype=torch.float32, device=dev)
        if not _gemm_bf16(dz, I, 1, x2, ctx.lda, 0, dw1, K, I, K, M, rowsum=db1):
            colsum(dz.float(), M, I, I, out=db1)
        return dx, dw1, db1, dw2, db2, None


def _ln_bwd_bf16(dy, s, mean, rstd, gamma, rows, H, p_pre, p_post, rng, site, gb_targets=None, want_bf16=True, want_f32=False):
    """-> (ds bf16 | None, dx bf16 (= ds when p_pre == 0), ds fp32 | None, dgamma, dbeta)."""
    if dy.dtype != _BF16:
        dy = cast_bf16(dy.reshape(rows, H))
    dy = dy.reshape(rows, H)
    if not dy.is_contiguous():
        dy = dy.contiguous()
    ds = torch.empty((rows, H), dtype=_BF16, device=dy.device) if want_bf16 else None
    dx = torch.empty((rows, H), dtype=_BF16, device=dy.device) if p_pre > 0 else None
    ds32 = torch.empty((rows, H), dtype=torch.float32, device=dy.device) if want_f32 else None
    nb = _lib.load().ytvln_ln_bwd_blocks(rows)
    partial = torch.empty((nb, 2 * H), dtype=torch.float32, device=dy.device)
    call("ytvln_ln_bwd_bf16", dy.data_ptr(), s.data_ptr(), _ptr(mean), _ptr(rstd), _ptr(gamma), ds.data_ptr() if ds is not None else None,
         dx.data_ptr() if dx is not None else None, _ptr(ds32), _ptr(partial), rows, H, float(p_pre), float(p_post),
         _ptr(rng) if rng is not None else None, int(site), _stream())
    gb = colsum(partial, nb, 2 * H, 2 * H, out=_direct_grad(gb_targets, (2 * H,)))
    return ds, (dx if dx is not None else ds), ds32, gb[:H], gb[H:]


class AddLayerNormBf16Fn(torch.autograd.Function):
    """y = LN(dropout(x) + residual) on bf16 rows (statistics, gamma / beta and their gradients fp32); saves s = dropout(x) + residual as bf16."""

    @staticmethod
    def forward(ctx, x, res, gamma, beta, eps, p_pre, p_post, rng, site):
        ctx.set_materialize_grads(False)
        _check(x, "x", _BF16)
        H = x.shape[-1]
        xc = x if x.is_contiguous() else x.contiguous()
        rc = None
        if res is not None:
            _check(res, "residual", _BF16)
            rc = res if res.is_contiguous() else res.contiguous()
        rows = xc.numel() // H
        need_grad = any(ctx.needs_input_grad)
        y = torch.empty_like(xc)
        s = torch.empty_like(xc) if need_grad else None
        mean = torch.empty(rows, dtype=torch.float32, device=x.device) if need_grad else None
        rstd = torch.empty_like(mean) if need_grad else None
        call("ytvln_ln_fwd_bf16", xc.data_ptr(), rc.data_ptr() if rc is not None else None, _ptr(gamma), _ptr(beta), y.data_ptr(),
             s.data_ptr() if s is not None else None, _ptr(mean), _ptr(rstd), rows, H, float(eps), float(p_pre), float(p_post),
             _ptr(rng) if rng is not None else None, int(site), _stream())
        ctx.meta = (rows, H, p_pre, p_post, site, res is not None)
        ctx.gb = (gamma, beta) if isinstance(gamma, torch.nn.Parameter) and isinstance(beta, torch.nn.Parameter) else None
        ctx.save_for_backward(s, mean, rstd, gamma, rng)
        return y

    @staticmethod
    def backward(ctx, dy):
        if dy is None:
            return (None,) * 9
        s, mean, rstd, gamma, rng = ctx.saved_tensors
        rows, H, p_pre, p_post, site, has_res = ctx.meta
        ds, dx, _, dg, db = _ln_bwd_bf16(dy, s, mean, rstd, gamma, rows, H, p_pre, p_post, rng, site, ctx.gb)
        shape = dy.shape
        return (dx.view(shape) if ctx.needs_input_grad[0] else None,
                ds.view(shape) if (has_res and ctx.needs_input_grad[1]) else None, dg, db, None, None, None, None, None)


# ------------------------------------------------------------------------------------------------------------------
# LayerNorm family
# ------------------------------------------------------------------------------------------------------------------
def _ln_bwd(dy, s, mean, rstd, gamma, rows, H, p_pre, p_post, rng, site, gb_targets=None):
    dy = dy.reshape(rows, H)
    if not dy.is_contiguous():
        dy = dy.contiguous()
    ds = torch.empty((rows, H), dtype=torch.float32, device=dy.device)
    dx = torch.empty_like(ds) if p_pre > 0 else None
    nb = _lib.load().ytvln_ln_bwd_blocks(rows)
    partial = torch.empty((nb, 2 * H), dtype=torch.float32, device=dy.device)
    call("ytvln_ln_bwd_f32", _ptr(dy), _ptr(s), _ptr(mean), _ptr(rstd), _ptr(gamma), _ptr(ds), _ptr(dx), _ptr(partial), rows, H,
         float(p_pre), float(p_post), _ptr(rng) if rng is not None else None, int(site), _stream())
    gb = colsum(partial, nb, 2 * H, 2 * H, out=_direct_grad(gb_targets, (2 * H,)))      # LayerNorm.weight | LayerNorm.bias
    return ds, (dx if dx is not None else ds), gb[:H], gb[H:]


class AddLayerNormFn(torch.autograd.Function):
    """y = LN(dropout(x) + residual) [dropout after] -- BertLayerNorm with its surrounding dropout / residual add."""

    @staticmethod
    def forward(ctx, x, res, gamma, beta, eps, p_pre, p_post, rng, site):
        ctx.set_materialize_grads(False)
        _check(x, "x")
        H = x.shape[-1]
        xc = x if x.is_contiguous() else x.contiguous()
        rc = None
        if res is not None:
            rc = res if res.is_contiguous() else res.contiguous()
        rows = xc.numel() // H
        need_grad = any(ctx.needs_input_grad)
        y = torch.empty_like(xc)
        s = torch.empty_like(xc) if need_grad else None
        mean = torch.empty(rows, dtype=torch.float32, device=x.device) if need_grad else None
        rstd = torch.empty_like(mean) if need_grad else None
        call("ytvln_ln_fwd_f32", _ptr(xc), _ptr(rc), _ptr(gamma), _ptr(beta), _ptr(y), _ptr(s), _ptr(mean), _ptr(rstd), rows, H,
             float(eps), float(p_pre), float(p_post), _ptr(rng) if rng is not None else None, int(site), _stream())
        ctx.meta = (rows, H, p_pre, p_post, site, res is not None)
        ctx.gb = (gamma, beta) if isinstance(gamma, torch.nn.Parameter) and isinstance(beta, torch.nn.Parameter) else None
        ctx.save_for_backward(s, mean, rstd, gamma, rng)
        return y

    @staticmethod
    def backward(ctx, dy):
        if dy is None:
            return (None,) * 9
        s, mean, rstd, gamma, rng = ctx.saved_tensors
        rows, H, p_pre, p_post, site, has_res = ctx.meta
        ds, dx, dg, db = _ln_bwd(dy, s, mean, rstd, gamma, rows, H, p_pre, p_post, rng, site, ctx.gb)
        shape = dy.shape
        return (dx.view(shape) if ctx.needs_input_grad[0] else None,
                ds.view(shape) if (has_res and ctx.needs_input_grad[1]) else None, dg, db, None, None, None, None, None)


def add_layer_norm(x, res, gamma, beta, eps=1e-12, p_pre=0.0, p_post=0.0, drop: Optional[DropoutState] = None) -> Tensor:
    if (p_pre > 0 or p_post > 0) and drop is None:
        raise RuntimeError("dropout requested without a DropoutState")
    use = drop is not None and (p_pre > 0 or p_post > 0)
    fn = AddLayerNormBf16Fn if x.dtype == torch.bfloat16 else AddLayerNormFn
    return fn.apply(x, res, gamma, beta, eps, p_pre if use else 0.0, p_post if use else 0.0,
                    drop.tensor if use else None, drop.next_site() if use else 0)


class TextEmbedFn(torch.autograd.Function):
    """BertEmbeddings.forward (vilbert.py:240-256) fused: gather x3 + add + LayerNorm + dropout."""

    @staticmethod
    def forward(ctx, ids, type_ids, word, pos, typ, gamma, beta, eps, p_post, rng, site):
        ctx.set_materialize_grads(False)
        ids = _i64(ids, "input_ids")
        N, T = ids.shape
        H = word.shape[1]
        if T > pos.shape[0]:
            raise RuntimeError(f"sequence length {T} exceeds max_position_embeddings {pos.shape[0]}")
        tt = _i64(type_ids, "token_type_ids") if type_ids is not None else None
        rows = N * T
        need_grad = any(ctx.needs_input_grad)
        bf16 = _MATMUL_PRECISION == "bf16"          # bf16-resident path: the hidden states leave the embedding as bf16
        y = torch.empty((N, T, H), dtype=torch.bfloat16 if bf16 else torch.float32, device=word.device)
        s = torch.empty_like(y) if need_grad else None
        mean = torch.empty(rows, dtype=torch.float32, device=word.device) if need_grad else None
        rstd = torch.empty_like(mean) if need_grad else None
        call("ytvln_text_embed_fwd_bf16" if bf16 else "ytvln_text_embed_fwd_f32", _ptr(ids), _ptr(tt), _ptr(word), _ptr(pos), _ptr(typ),
             _ptr(gamma), _ptr(beta), _ptr(y), _ptr(s), _ptr(mean), _ptr(rstd), rows, T, H, float(eps), float(p_post),
             _ptr(rng) if rng is not None else None, int(site), _stream())
        ctx.meta = (N, T, H, p_post, site, word.shape, pos.shape, typ.shape)
        ctx.save_for_backward(ids, tt, s, mean, rstd, gamma, rng)
        return y

    @staticmethod
    def backward(ctx, dy):
        if dy is None:
            return (None,) * 11
        ids, tt, s, mean, rstd, gamma, rng = ctx.saved_tensors
        N, T, H, p_post, site, wshape, pshape, tshape = ctx.meta
        rows = N * T
        if s.dtype == torch.bfloat16:          # bf16-resident path: the table-gradient kernels below read an fp32 copy of ds
            _, _, ds, dg, db = _ln_bwd_bf16(dy, s, mean, rstd, gamma, rows, H, 0.0, p_post, rng, site, want_bf16=False, want_f32=True)
        else:
            ds, _, dg, db = _ln_bwd(dy, s, mean, rstd, gamma, rows, H, 0.0, p_post, rng, site)
        dev = ds.device
        dword = torch.zeros(wshape, dtype=torch.float32, device=dev)
        # token ids repeat inside a batch: sorted (stable) so that one wave owns each id's run -- no atomics, reproducible sums
        sorted_ids, perm = torch.sort(ids.reshape(-1), stable=True)
        call("ytvln_scatter_add_rows_sorted_f32", _ptr(ds), H, _ptr(sorted_ids), _ptr(perm), rows, H, _ptr(dword), 0, _stream())   # padding_idx = 0
        dpos = torch.zeros(pshape, dtype=torch.float32, device=dev)
        dpos[:T] = colsum(ds, N, T * H, T * H).view(T, H)
        if tt is None:
            dtyp = torch.zeros(tshape, dtype=torch.float32, device=dev)
            dtyp[0] = colsum(ds, rows, H, H)
        elif tshape[0] <= 32:
            dtyp = colsum_by_index(ds, rows, H, H, tshape[0], idx_i64=tt)
        else:
            dtyp = torch.zeros(tshape, dtype=torch.float32, device=dev)
            call("ytvln_scatter_add_rows_f32", _ptr(ds), H, _ptr(tt), rows, H, _ptr(dtyp), -1, _stream())
        return None, None, dword, dpos, dtyp, dg, db, None, None, None, None


_CHECK_INDICES = os.environ.get("YTVLN_CHECK_INDICES", "0") != "0"


def _check_index_range(idx: Tensor, n: int, what: str) -> None:
    """Debug aid (YTVLN_CHECK_INDICES=1; one host sync per call): the embedding kernels gather rows by raw index like torch's own
    kernels, which assert on the device; production keeps the sync-free path."""
    if _CHECK_INDICES and idx.numel():
        lo, hi = int(idx.min()), int(idx.max())
        if lo < 0 or hi >= n:
            raise IndexError(f"{what}: index range [{lo}, {hi}] outside the table of {n} rows")


def text_embed(ids, type_ids, word, pos, typ, gamma, beta, eps=1e-12, p=0.0, drop: Optional[DropoutState] = None) -> Tensor:
    _check_index_range(ids, word.shape[0], "token ids")
    if type_ids is not None:
        _check_index_range(type_ids, typ.shape[0], "token type ids")
    use = drop is not None and p > 0
    return TextEmbedFn.apply(ids, type_ids, word, pos, typ, gamma, beta, eps, p if use else 0.0, drop.tensor if use else None,
                             drop.next_site() if use else 0)


class ImageEmbedFn(torch.autograd.Function):
    """BertImageEmbeddings.forward after the feature projection (vilbert.py:1361-1368), fused:
    location / orientation / next-orientation linears (K = 5, 4, 2) + frame-index gather + add + LayerNorm + dropout."""

    @staticmethod
    def forward(ctx, img, loc, W5, b5, W4, b4, W2, b2, E, gamma, beta, eps, p_post, rng, site):
        ctx.set_materialize_grads(False)
        bf16 = img.dtype == torch.bfloat16          # bf16-resident path: img is the bf16 output of the feature projection
        _check(img, "img", img.dtype if bf16 else torch.float32)
        _check(loc, "image_loc")
        H = img.shape[-1]
        imgc = img if img.is_contiguous() else img.contiguous()
        locc = loc if loc.is_contiguous() else loc.contiguous()
        if locc.shape[-1] != 12:
            raise RuntimeError(f"image_loc must have 12 columns (5 box + 4 orientation + 2 next-orientation + frame); got {locc.shape}")
        rows = imgc.numel() // H
        need_grad = any(ctx.needs_input_grad)
        y = torch.empty_like(imgc)
        s = torch.empty_like(imgc) if need_grad else None
        mean = torch.empty(rows, dtype=torch.float32, device=img.device) if need_grad else None
        rstd = torch.empty_like(mean) if need_grad else None
        ws = [w if w.is_contiguous() else w.contiguous() for w in (W5, W4, W2, E)]
        call("ytvln_image_embed_fwd_bf16" if bf16 else "ytvln_image_embed_fwd_f32", _ptr(imgc), _ptr(locc), _ptr(ws[0]), _ptr(b5), _ptr(ws[1]), _ptr(b4), _ptr(ws[2]), _ptr(b2),
             _ptr(ws[3]), _ptr(gamma), _ptr(beta), _ptr(y), _ptr(s), _ptr(mean), _ptr(rstd), rows, H, float(eps), float(p_post),
             _ptr(rng) if rng is not None else None, int(site), _stream())
        ctx.meta = (rows, H, p_post, site, E.shape[0])
        ctx.save_for_backward(locc, s, mean, rstd, gamma, rng)
        return y

    @staticmethod
    def backward(ctx, dy):
        if dy is None:
            return (None,) * 15
        locc, s, mean, rstd, gamma, rng = ctx.saved_tensors
        rows, H, p_post, site, KT = ctx.meta
        dimg = None
        if s.dtype == torch.bfloat16:          # bf16-resident path: bf16 ds for the feature projection's backward, an fp32 copy for the small kernels below
            dimg, _, ds, dg, db = _ln_bwd_bf16(dy, s, mean, rstd, gamma, rows, H, 0.0, p_post, rng, site, want_bf16=True, want_f32=True)
        else:
            ds, _, dg, db = _ln_bwd(dy, s, mean, rstd, gamma, rows, H, 0.0, p_post, rng, site)
        dev = ds.device
        dwall = torch.empty((H, 12), dtype=torch.float32, device=dev)
        _gemm(ds, H, 1, locc, 12, 0, dwall, 12, H, 12, rows)               # ds^T . loc  -> [H, 12]
        dbias = colsum(ds, rows, H, H)
        dE = colsum_by_index(ds, rows, H, H, KT, idx_f32=locc, idx_stride=12, idx_f32_offset=11)
        return ((dimg if dimg is not None else ds).view(dy.shape), None, dwall[:, 0:5].contiguous(), dbias, dwall[:, 5:9].contiguous(), dbias,
                dwall[:, 9:11].contiguous(), dbias, dE, dg, db, None, None, None, None)


def image_embed(img, loc, W5, b5, W4, b4, W2, b2, E, gamma, beta, eps=1e-12, p=0.0, drop: Optional[DropoutState] = None) -> Tensor:
    if _CHECK_INDICES:
        _check_index_range(loc[..., 11].long(), E.shape[0], "frame index (image_loc[..., 11])")
    use = drop is not None and p > 0
    if _MATMUL_PRECISION == "bf16" and img.dtype == torch.float32:
        # both streams of the bf16-resident path are bf16 (co-attention reads them side by side); a feature projection too small for the bf16
        # GEMM (K < 64: toy configs) leaves its output fp32 -- round it here, autograd casts the gradient back
        img = img.to(torch.bfloat16)
    return ImageEmbedFn.apply(img, loc, W5, b5, W4, b4, W2, b2, E, gamma, beta, eps, p if use else 0.0,
                              drop.tensor if use else None, drop.next_site() if use else 0)


class GatherRowsFn(torch.autograd.Function):
    """out[j] = x[idx[j]] over rows of a [M, H] matrix (idx < 0 -> zero row); backward scatters the row gradients back."""

    @staticmethod
    def forward(ctx, x, idx):
        ctx.set_materialize_grads(False)
        x2, M, H, ld = _rows2d(x, "x")
        idx = _i64(idx, "idx").reshape(-1)
        out = torch.empty((idx.numel(), H), dtype=torch.float32, device=x.device)
        call("ytvln_gather_rows_f32", _ptr(x2), ld, _ptr(idx), idx.numel(), H, _ptr(out), _stream())
        ctx.meta = (M, H, x.shape)
        ctx.save_for_backward(idx)
        return out

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return None, None
        (idx,) = ctx.saved_tensors
        M, H, shape = ctx.meta
        g = g if g.is_contiguous() else g.contiguous()
        dx = torch.zeros((M, H), dtype=torch.float32, device=g.device)
        call("ytvln_scatter_add_rows_f32", _ptr(g), H, _ptr(idx), idx.numel(), H, _ptr(dx), -1, _stream())
        return dx.view(shape), None


def gather_rows(x: Tensor, idx: Tensor) -> Tensor:
    return GatherRowsFn.apply(x, idx)


def select_rows(flag: Tensor, capacity: int) -> Tensor:
    """Static-shape, sync-free row selection: the first `capacity` indices of a stable sort that puts flagged rows first.
    If fewer rows are flagged the tail holds un-flagged rows (their targets are "ignore", so they change no loss); callers
    size `capacity` well above the expected count and may check `flag.sum() <= capacity` lazily."""
    order = torch.argsort((~flag.reshape(-1).bool()).to(torch.int8), stable=True)
    return order[:capacity].contiguous()


class DropoutFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p, rng, site):
        ctx.set_materialize_grads(False)
        _check(x, "x")
        xc = x if x.is_contiguous() else x.contiguous()
        y = torch.empty_like(xc)
        call("ytvln_dropout_f32", _ptr(xc), _ptr(y), xc.numel(), float(p), _ptr(rng), int(site), _stream())
        ctx.p, ctx.site = p, site
        ctx.save_for_backward(rng)
        return y

    @staticmethod
    def backward(ctx, dy):
        if dy is None:
            return None, None, None, None
        (rng,) = ctx.saved_tensors
        dyc = dy if dy.is_contiguous() else dy.contiguous()
        dx = torch.empty_like(dyc)
        call("ytvln_dropout_f32", _ptr(dyc), _ptr(dx), dyc.numel(), float(ctx.p), _ptr(rng), int(ctx.site), _stream())
        return dx, None, None, None


def dropout(x: Tensor, p: float, training: bool, drop: Optional[DropoutState]) -> Tensor:
    if not training or p <= 0.0:
        return x
    if drop is None:
        raise RuntimeError("dropout requested without a DropoutState")
    return DropoutFn.apply(x, p, drop.tensor, drop.next_site())


# ------------------------------------------------------------------------------------------------------------------
# attention
# ------------------------------------------------------------------------------------------------------------------
def _eptr(t: Optional[Tensor], offset_elems: int = 0):
    return None if t is None else t.data_ptr() + t.element_size() * offset_elems


def _attn_problem(q, q_off, ldq, k, k_off, ldk, v, v_off, ldv, mask, Tq, Tk, p, site, ctx=None, lse=None, ctx_in=None, dctx=None,
                  lse_in=None, delta=None, dq=None, dq_off=0, lddq=0, dk=None, dk_off=0, lddk=0, dv=None, dv_off=0, lddv=0, keep=None):
    """One `ytvln_attn_problem` record (include/ytvln.h); q / k / v / ctx / gradients are fp32 or bf16 tensors (offsets in elements)."""
    pr = _lib.AttnProblem()
    pr.q, pr.k, pr.v, pr.mask = _eptr(q, q_off), _eptr(k, k_off), _eptr(v, v_off), _ptr(mask)
    pr.ctx, pr.lse = _eptr(ctx), _ptr(lse)
    pr.ctx_in, pr.dctx, pr.lse_in, pr.delta = _eptr(ctx_in), _eptr(dctx), _ptr(lse_in), _ptr(delta)
    pr.dq, pr.dk, pr.dv = _eptr(dq, dq_off), _eptr(dk, dk_off), _eptr(dv, dv_off)
    o = ctx if ctx is not None else ctx_in
    pr.ldq, pr.ldk, pr.ldv, pr.ldo, pr.lddq, pr.lddk, pr.lddv = ldq, ldk, ldv, o.shape[-1], lddq, lddk, lddv
    pr.Tq, pr.Tk, pr.p_drop, pr.site = Tq, Tk, float(p), int(site)
    pr.keep = keep.data_ptr() if keep is not None else None
    return pr


def _attn_keep(N, heads, Tq, Tk, p, device):
    """bf16-resident attention with dropout: the buffer the forward kernel writes its keep decisions to and the backward kernels read them from
    (`ytvln_attn_problem.keep`); None without dropout."""
    if not p > 0:
        return None
    return torch.empty(int(_lib.load().ytvln_attn_keep_bytes(N, heads, Tq, Tk)), dtype=torch.uint8, device=device)


def _attn_launch(backward: bool, bf16: bool, pa, pb, N, heads, d, scale, rng):
    """One launch over one problem (pb = None) or the two directions of BertBiAttention."""
    rp = _ptr(rng) if rng is not None else None
    if bf16:
        call("ytvln_attn_bwd_bf16" if backward else "ytvln_attn_fwd_bf16", ctypes.addressof(pa), ctypes.addressof(pb) if pb is not None else None,
             N, heads, d, float(scale), rp, _stream())
    else:
        call("ytvln_attn_bwd_pair" if backward else "ytvln_attn_fwd_pair", ctypes.addressof(pa), ctypes.addressof(pb), N, heads, d, float(scale),
             rp, _stream())


def _attn_fwd(q, q_off, ldq, k, k_off, ldk, v, v_off, ldv, mask, out, N, heads, Tq, Tk, d, scale, p, rng, site):
    lse = torch.empty((N, heads, Tq), dtype=torch.float32, device=out.device)
    if q.dtype == torch.bfloat16:
        keep = _attn_keep(N, heads, Tq, Tk, p, out.device)
        _attn_launch(False, True, _attn_problem(q, q_off, ldq, k, k_off, ldk, v, v_off, ldv, mask, Tq, Tk, p, site, ctx=out, lse=lse, keep=keep),
                     None, N, heads, d, scale, rng)
        lse._ytvln_keep = keep          # direct callers hand `lse` to _attn_bwd, which finds the forward's keep decisions here; the autograd
        return lse                      # functions save the buffer explicitly (a saved tensor is unpacked into a new Python object)
    call("ytvln_attn_fwd_f32", _ptr(q, q_off), ldq, _ptr(k, k_off), ldk, _ptr(v, v_off), ldv, _ptr(mask), _ptr(out), out.shape[-1],
         _ptr(lse), N, heads, Tq, Tk, d, float(scale), float(p), _ptr(rng) if rng is not None else None, int(site), _stream())
    return lse


def _attn_bwd(q, q_off, ldq, k, k_off, ldk, v, v_off, ldv, mask, out, dout, lse, dq, dq_off, lddq, dk, dk_off, lddk, dv, dv_off,
              lddv, N, heads, Tq, Tk, d, scale, p, rng, site, keep=None):
    delta = torch.empty_like(lse)
    if q.dtype == torch.bfloat16:
        if keep is None:
            keep = getattr(lse, "_ytvln_keep", None)
        if p > 0 and keep is None:
            raise RuntimeError("bf16 attention backward with dropout needs the keep decisions its forward wrote (pass keep= or the lse tensor _attn_fwd returned)")
        _attn_launch(True, True, _attn_problem(q, q_off, ldq, k, k_off, ldk, v, v_off, ldv, mask, Tq, Tk, p, site, ctx_in=out, dctx=dout,
                                               lse_in=lse, delta=delta, dq=dq, dq_off=dq_off, lddq=lddq, dk=dk, dk_off=dk_off, lddk=lddk,
                                               dv=dv, dv_off=dv_off, lddv=lddv, keep=keep), None, N, heads, d, scale, rng)
        return
    call("ytvln_attn_bwd_f32", _ptr(q, q_off), ldq, _ptr(k, k_off), ldk, _ptr(v, v_off), ldv, _ptr(mask), _ptr(out), _ptr(dout),
         out.shape[-1], _ptr(lse), _ptr(delta), _ptr(dq, dq_off), lddq, _ptr(dk, dk_off), lddk, _ptr(dv, dv_off), lddv, N, heads, Tq,
         Tk, d, float(scale), float(p), _ptr(rng) if rng is not None else None, int(site), _stream())


def attn_probs(q, q_off, ldq, k, k_off, ldk, mask, lse, N, heads, Tq, Tk, d, scale) -> Tensor:
    probs = torch.empty((N, heads, Tq, Tk), dtype=torch.float32, device=lse.device)
    if q.dtype == torch.bfloat16 or k.dtype == torch.bfloat16:
        # bf16-resident path (output_all_attention_masks is a rare diagnostic there): the probability kernel reads fp32 rows, so the
        # bf16 projections are widened first -- offsets and leading dimensions are in elements and carry over unchanged
        q, k = q.float(), k.float()
    elif q.dtype != torch.float32 or k.dtype != torch.float32:
        raise RuntimeError(f"attn_probs: q / k must be float32 or bfloat16, got {q.dtype} / {k.dtype}")
    call("ytvln_attn_probs_f32", _ptr(q, q_off), ldq, _ptr(k, k_off), ldk, _ptr(mask), _ptr(lse), _ptr(probs), N, heads, Tq, Tk, d,
         float(scale), _stream())
    return probs


class SelfAttentionFn(torch.autograd.Function):
    """ctx = MHA(qkv) for a packed [N*T, 3H] projection (query | key | value column blocks); vilbert.py:284-311."""

    @staticmethod
    def forward(ctx, qkv, mask, N, T, heads, p, rng, site):
        ctx.set_materialize_grads(False)
        _check(qkv, "qkv", qkv.dtype if qkv.dtype == torch.bfloat16 else torch.float32)
        assert qkv.is_contiguous() and qkv.dim() == 2 and qkv.shape[0] == N * T
        H = qkv.shape[1] // 3
        d = H // heads
        if qkv.dtype == torch.bfloat16 and d not in (64, 128):
            raise NotImplementedError(f"bf16-resident attention is built for head dimensions 64 and 128 (got {d}); use the fp32 path")
        scale = 1.0 / math.sqrt(d)
        out = torch.empty((N * T, H), dtype=qkv.dtype, device=qkv.device)
        lse = _attn_fwd(qkv, 0, 3 * H, qkv, H, 3 * H, qkv, 2 * H, 3 * H, mask, out, N, heads, T, T, d, scale, p, rng, site)
        keep = getattr(lse, "_ytvln_keep", None)
        ctx.meta = (N, T, heads, H, d, scale, p, site)
        ctx.save_for_backward(qkv, mask, out, lse, rng, keep)
        ctx.mark_non_differentiable(lse)
        return out, lse

    @staticmethod
    def backward(ctx, dout, _dlse):
        if dout is None:
            return (None,) * 8
        qkv, mask, out, lse, rng, keep = ctx.saved_tensors
        N, T, heads, H, d, scale, p, site = ctx.meta
        dout = dout if dout.is_contiguous() else dout.contiguous()
        if dout.dtype != qkv.dtype:
            dout = dout.to(qkv.dtype)
        dqkv = torch.empty_like(qkv)
        _attn_bwd(qkv, 0, 3 * H, qkv, H, 3 * H, qkv, 2 * H, 3 * H, mask, out, dout, lse, dqkv, 0, 3 * H, dqkv, H, 3 * H, dqkv, 2 * H,
                  3 * H, N, heads, T, T, d, scale, p, rng, site, keep=keep)
        return dqkv, None, None, None, None, None, None, None


class CoAttentionFn(torch.autograd.Function):
    """BertBiAttention (vilbert.py:552-618).  Projections are packed PER DIRECTION so that a direction whose context is
    never used downstream (e.g. the vision side of the last co-layer under --masked_language only) propagates `None`
    gradients exactly like the reference's autograd graph -- AdamW must not touch those tensors:
        q1  [N*R, Hb]  = query1(image)        kv1 [N*R, 2Hb] = key1|value1(image)
        q2  [N*T, Hb]  = query2(text)         kv2 [N*T, 2Hb] = key2|value2(text)
        ctx1 [N*T, Hb] = attend(q2; kv1, image mask)     ctx2 [N*R, Hb] = attend(q1; kv2, text mask)"""

    @staticmethod
    def forward(ctx, q1, kv1, q2, kv2, mask1, mask2, N, R, T, heads, p1, p2, rng, site1, site2):
        bf16 = q1.dtype == torch.bfloat16
        for t, nme in ((q1, "q1"), (kv1, "kv1"), (q2, "q2"), (kv2, "kv2")):
            _check(t, nme, torch.bfloat16 if bf16 else torch.float32)
            assert t.is_contiguous() and t.dim() == 2
        Hb = q1.shape[1]
        d = Hb // heads
        if bf16 and d not in (64, 128):
            raise NotImplementedError(f"bf16-resident attention is built for head dimensions 64 and 128 (got {d}); use the fp32 path")
        scale = 1.0 / math.sqrt(d)
        ctx1 = torch.empty((N * T, Hb), dtype=q1.dtype, device=q1.device)
        ctx2 = torch.empty((N * R, Hb), dtype=q1.dtype, device=q1.device)
        lse1 = torch.empty((N, heads, T), dtype=torch.float32, device=q1.device)
        lse2 = torch.empty((N, heads, R), dtype=torch.float32, device=q1.device)
        # both directions in one launch (text queries over regions | region queries over text)
        pk = max(p1, p2)          # one launch: dropout in either direction makes the kernels record decisions for both
        keep1 = _attn_keep(N, heads, T, R, pk, q1.device) if bf16 else None
        keep2 = _attn_keep(N, heads, R, T, pk, q1.device) if bf16 else None
        _attn_launch(False, bf16,
                     _attn_problem(q2, 0, Hb, kv1, 0, 2 * Hb, kv1, Hb, 2 * Hb, mask1, T, R, p1, site1, ctx=ctx1, lse=lse1, keep=keep1),
                     _attn_problem(q1, 0, Hb, kv2, 0, 2 * Hb, kv2, Hb, 2 * Hb, mask2, R, T, p2, site2, ctx=ctx2, lse=lse2, keep=keep2),
                     N, heads, d, scale, rng)
        ctx.meta = (N, R, T, heads, Hb, d, scale, p1, p2, site1, site2)
        ctx.save_for_backward(q1, kv1, q2, kv2, mask1, mask2, ctx1, ctx2, lse1, lse2, rng, keep1, keep2)
        ctx.mark_non_differentiable(lse1, lse2)
        ctx.set_materialize_grads(False)
        return ctx1, ctx2, lse1, lse2

    @staticmethod
    def backward(ctx, d1, d2, _a, _b):
        q1, kv1, q2, kv2, mask1, mask2, ctx1, ctx2, lse1, lse2, rng, keep1, keep2 = ctx.saved_tensors
        N, R, T, heads, Hb, d, scale, p1, p2, site1, site2 = ctx.meta
        gq1 = gkv1 = gq2 = gkv2 = None
        if d1 is not None and d2 is not None:       # the usual case: both directions in one launch per kernel
            d1 = d1 if d1.is_contiguous() else d1.contiguous()
            d2 = d2 if d2.is_contiguous() else d2.contiguous()
            gq2, gkv1, gq1, gkv2 = torch.empty_like(q2), torch.empty_like(kv1), torch.empty_like(q1), torch.empty_like(kv2)
            delta1, delta2 = torch.empty_like(lse1), torch.empty_like(lse2)
            _attn_launch(True, q1.dtype == torch.bfloat16,
                       _attn_problem(q2, 0, Hb, kv1, 0, 2 * Hb, kv1, Hb, 2 * Hb, mask1, T, R, p1, site1, ctx_in=ctx1, dctx=d1, lse_in=lse1,
                                     delta=delta1, dq=gq2, lddq=Hb, dk=gkv1, lddk=2 * Hb, dv=gkv1, dv_off=Hb, lddv=2 * Hb, keep=keep1),
                       _attn_problem(q1, 0, Hb, kv2, 0, 2 * Hb, kv2, Hb, 2 * Hb, mask2, R, T, p2, site2, ctx_in=ctx2, dctx=d2, lse_in=lse2,
                                     delta=delta2, dq=gq1, lddq=Hb, dk=gkv2, lddk=2 * Hb, dv=gkv2, dv_off=Hb, lddv=2 * Hb, keep=keep2),
                       N, heads, d, scale, rng)
            return (gq1, gkv1, gq2, gkv2) + (None,) * 11
        if d1 is not None:      # text queries over image keys/values -> dq2, dk1|dv1
            d1 = d1 if d1.is_contiguous() else d1.contiguous()
            gq2, gkv1 = torch.empty_like(q2), torch.empty_like(kv1)
            _attn_bwd(q2, 0, Hb, kv1, 0, 2 * Hb, kv1, Hb, 2 * Hb, mask1, ctx1, d1, lse1, gq2, 0, Hb, gkv1, 0, 2 * Hb, gkv1, Hb,
                      2 * Hb, N, heads, T, R, d, scale, p1, rng, site1, keep=keep1)
        if d2 is not None:      # image queries over text keys/values -> dq1, dk2|dv2
            d2 = d2 if d2.is_contiguous() else d2.contiguous()
            gq1, gkv2 = torch.empty_like(q1), torch.empty_like(kv2)
            _attn_bwd(q1, 0, Hb, kv2, 0, 2 * Hb, kv2, Hb, 2 * Hb, mask2, ctx2, d2, lse2, gq1, 0, Hb, gkv2, 0, 2 * Hb, gkv2, Hb,
                      2 * Hb, N, heads, R, T, d, scale, p2, rng, site2, keep=keep2)
        return (gq1, gkv1, gq2, gkv2) + (None,) * 11


# ------------------------------------------------------------------------------------------------------------------
# losses
# ------------------------------------------------------------------------------------------------------------------
class CrossEntropyFn(torch.autograd.Function):
    """F.cross_entropy(logits, target, ignore_index) with mean reduction (utils_init.py:133-135, :141)."""

    @staticmethod
    def forward(ctx, logits, target, ignore_index):
        ctx.set_materialize_grads(False)
        bf = logits.dtype == _BF16             # bf16-resident path: bf16 logits in, bf16 gradient out (same kernels, 2-byte loads/stores)
        lg, M, V, ld = _rows2d_any(logits) if bf else _rows2d(logits, "logits")
        tg = _i64(target, "target").reshape(-1)
        assert tg.numel() == M, (tg.shape, M)
        dev = lg.device
        row_lse = torch.empty(M, dtype=torch.float32, device=dev)
        row_loss = torch.empty(M, dtype=torch.float32, device=dev)
        out = torch.empty(2, dtype=torch.float32, device=dev)
        call("ytvln_ce_fwd_bf16" if bf else "ytvln_ce_fwd_f32", lg.data_ptr(), ld, _ptr(tg), int(ignore_index), _ptr(row_lse), _ptr(row_loss),
             _ptr(out), M, V, _stream())
        ctx.meta = (M, V, ld, int(ignore_index), logits.shape)
        ctx.bf16_grad = bf
        ctx.save_for_backward(lg, tg, row_lse, out)
        return out[0]

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return None, None, None
        lg, tg, row_lse, out = ctx.saved_tensors
        M, V, ld, ign, shape = ctx.meta
        g = g.reshape(1).contiguous().float()
        if ctx.bf16_grad:          # rounded once, zero padding written by the kernel: feeds ytvln_gemm_bf16 as it stands
            dl, ldd = _alloc_rows_bf16(M, V, lg.device)
            call("ytvln_ce_bwd_bf16", lg.data_ptr(), ld, _ptr(tg), ign, _ptr(row_lse), _ptr(out), _ptr(g), dl.data_ptr(), ldd, M, V, _stream())
            return _view_rows_as(dl, ldd, shape), None, None
        dl, ldd = _alloc_rows(M, V, lg.device, zero_pad=True)
        call("ytvln_ce_bwd_f32", _ptr(lg), ld, _ptr(tg), ign, _ptr(row_lse), _ptr(out), _ptr(g), _ptr(dl), ldd, M, V, _stream())
        return _view_rows_as(dl, ldd, shape), None, None


def cross_entropy(logits: Tensor, target: Tensor, ignore_index: int = -100) -> Tensor:
    return CrossEntropyFn.apply(logits, target, ignore_index)


class KLMaskedFn(torch.autograd.Function):
    """sum(mask * kl_div(log_softmax(pred), target)) / max(1, sum(mask))  (utils_init.py:117-128), no host sync."""

    @staticmethod
    def forward(ctx, pred, target, mask):
        ctx.set_materialize_grads(False)
        bf = pred.dtype == _BF16
        pr, M, Cc, ld = _rows2d_any(pred) if bf else _rows2d(pred, "pred")
        tg, M2, C2, ldt = _rows2d(target, "target")
        assert (M, Cc) == (M2, C2), (pred.shape, target.shape)
        mk = _i64(mask, "mask").reshape(-1)
        dev = pr.device
        row_lse = torch.empty(M, dtype=torch.float32, device=dev)
        row_loss = torch.empty(M, dtype=torch.float32, device=dev)
        out = torch.empty(2, dtype=torch.float32, device=dev)
        call("ytvln_kl_fwd_bf16" if bf else "ytvln_kl_fwd_f32", pr.data_ptr(), ld, _ptr(tg), ldt, _ptr(mk), _ptr(row_lse), _ptr(row_loss),
             _ptr(out), M, Cc, _stream())
        ctx.meta = (M, Cc, ld, ldt, pred.shape)
        ctx.bf16_grad = bf
        ctx.save_for_backward(pr, tg, mk, row_lse, out)
        return out[0]

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return None, None, None
        pr, tg, mk, row_lse, out = ctx.saved_tensors
        M, Cc, ld, ldt, shape = ctx.meta
        g = g.reshape(1).contiguous().float()
        if ctx.bf16_grad:
            dp, ldd = _alloc_rows_bf16(M, Cc, pr.device)
            call("ytvln_kl_bwd_bf16", pr.data_ptr(), ld, _ptr(tg), ldt, _ptr(mk), _ptr(row_lse), _ptr(out), _ptr(g), dp.data_ptr(), ldd, M, Cc, _stream())
            return _view_rows_as(dp, ldd, shape), None, None
        dp, ldd = _alloc_rows(M, Cc, pr.device, zero_pad=True)
        call("ytvln_kl_bwd_f32", _ptr(pr), ld, _ptr(tg), ldt, _ptr(mk), _ptr(row_lse), _ptr(out), _ptr(g), _ptr(dp), ldd, M, Cc, _stream())
        return _view_rows_as(dp, ldd, shape), None, None


def kl_masked(pred: Tensor, target: Tensor, mask: Tensor) -> Tensor:
    return KLMaskedFn.apply(pred, target, mask)


class BCEWithLogitsFn(torch.autograd.Function):
    """F.binary_cross_entropy_with_logits(x, t, pos_weight=w), mean reduction (utils_init.py:143, :160-161)."""

    @staticmethod
    def forward(ctx, x, t, pos_weight):
        ctx.set_materialize_grads(False)
        _check(x, "x")
        xc = x.contiguous()
        tc = t.to(torch.float32).contiguous()
        pw = pos_weight.to(torch.float32).reshape(1).contiguous() if pos_weight is not None else None
        out = torch.empty(1, dtype=torch.float32, device=x.device)
        call("ytvln_bce_fwd_f32", _ptr(xc), _ptr(tc), _ptr(pw), _ptr(out), xc.numel(), _stream())
        ctx.save_for_backward(xc, tc, pw)
        return out[0]

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return None, None, None
        xc, tc, pw = ctx.saved_tensors
        g = g.reshape(1).contiguous().float()
        dx = torch.empty_like(xc)
        call("ytvln_bce_bwd_f32", _ptr(xc), _ptr(tc), _ptr(pw), _ptr(g), _ptr(dx), xc.numel(), _stream())
        return dx, None, None


def bce_with_logits(x: Tensor, t: Tensor, pos_weight: Optional[Tensor] = None) -> Tensor:
    return BCEWithLogitsFn.apply(x, t, pos_weight)


# ------------------------------------------------------------------------------------------------------------------
# optimizer kernel
# ------------------------------------------------------------------------------------------------------------------
def adamw_step(p: Tensor, g: Tensor, m: Tensor, v: Tensor, chunks: Tensor, nchunks: int, hyper: Tensor, grad_scale: float = 1.0,
               p_bf16: Optional[Tensor] = None):
    """`p_bf16`: the bf16 parameter arena of the bf16-resident path (same offsets): refreshed by the same kernel pass."""
    for t, nme in ((p, "p"), (g, "g"), (m, "m"), (v, "v"), (hyper, "hyper")):
        _check(t, nme)
    if p_bf16 is not None:
        call("ytvln_adamw_f32_bf16copy", _ptr(p), _ptr(g), _ptr(m), _ptr(v), p_bf16.data_ptr(), chunks.data_ptr(), int(nchunks), _ptr(hyper),
             float(grad_scale), _stream())
        return
    call("ytvln_adamw_f32", _ptr(p), _ptr(g), _ptr(m), _ptr(v), chunks.data_ptr(), int(nchunks), _ptr(hyper), float(grad_scale),
         _stream())
