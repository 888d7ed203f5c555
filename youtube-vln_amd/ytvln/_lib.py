"""ctypes binding of libytvln.so (the C ABI declared in include/ytvln.h).

There is NO fallback: if the shared library is missing or a symbol cannot be resolved the import of the compute path
fails loudly (`YtvlnLibraryError`).  `torch` is imported first on purpose so that the HIP runtime the library binds to
(`libamdhip64.so.7`) is the one PyTorch already loaded -- streams and device pointers are then interchangeable.
"""
from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  (must precede the dlopen below)

HERE = os.path.dirname(os.path.abspath(__file__))
# YTVLN_LIB: an alternative build of the SAME ABI (A/B experiments: tools/, scratch/); a missing file still fails loudly in load()
LIB_PATH = os.path.abspath(os.environ["YTVLN_LIB"]) if os.environ.get("YTVLN_LIB") else os.path.join(HERE, "lib", "libytvln.so")


class YtvlnLibraryError(RuntimeError):
    pass


P, I64, I32, F32 = C.c_void_p, C.c_int64, C.c_int, C.c_float

# name -> argtypes, exactly mirroring include/ytvln.h (tests/test_abi.py checks header <-> table <-> exported symbols)
class AttnProblem(C.Structure):
    """`ytvln_attn_problem` of include/ytvln.h."""
    _fields_ = [(n, C.c_void_p) for n in ("q", "k", "v", "mask", "ctx_in", "dctx", "lse_in", "ctx", "lse", "delta", "dq", "dk", "dv")] + \
               [(n, C.c_int64) for n in ("ldq", "ldk", "ldv", "ldo", "lddq", "lddk", "lddv")] + \
               [("Tq", C.c_int32), ("Tk", C.c_int32), ("p_drop", C.c_float), ("reserved", C.c_int32), ("site", C.c_int64), ("keep", C.c_void_p)]


SIGNATURES = {
    "ytvln_gemm_workspace_elems": [I32, I32, I32, I32],
    "ytvln_gemm_f32": [P, I64, I32, P, I64, I32, P, I64, P, P, I64, I32, I32, I32, I32, F32, P, I64, I32, P],
    "ytvln_gemm_f32_rowsum": [P, I64, I32, P, I64, I32, P, I64, P, P, I64, I32, I32, I32, I32, F32, P, I64, I32, P, P, P],
    "ytvln_gemm_sk_ctl_elems": [],
    "ytvln_gemm_f32_sk": [P, I64, I32, P, I64, I32, P, I64, P, P, I64, I32, I32, I32, I32, F32, P, I64, I32, P, P, P, P],
    "ytvln_gemm_sk_plan": [I32, I32, I32, I32, I32, P, P, P, P, P],
    "ytvln_gemm_probe": [P],
    "ytvln_gemm_bf16_probe": [P, I32, I32],
    "ytvln_colsum_f32": [P, I64, I32, I32, P, I64, I32, P],
    "ytvln_colsum_by_index_f32": [P, I64, P, I64, P, I32, I32, I32, P, I32, P],
    "ytvln_scatter_add_rows_f32": [P, I64, P, I32, I32, P, I64, P],
    "ytvln_gather_rows_f32": [P, I64, P, I32, I32, P, P],
    "ytvln_scatter_add_rows_sorted_f32": [P, I64, P, P, I32, I32, P, I64, P],
    "ytvln_randomize_tokens": [P, P, I64, I32, I64, P, P, P, I64, P, P, P],
    "ytvln_randomize_regions": [P, I64, P, P, I64, I32, I32, P, P, I64, P, P, P],
    "ytvln_attn_keep_bytes": [I32, I32, I32, I32],
    "ytvln_attn_fwd_bf16": [P, P, I32, I32, I32, F32, P, P],
    "ytvln_attn_bwd_bf16": [P, P, I32, I32, I32, F32, P, P],
    "ytvln_attn_fwd_pair": [P, P, I32, I32, I32, F32, P, P],
    "ytvln_attn_bwd_pair": [P, P, I32, I32, I32, F32, P, P],
    "ytvln_gemm_plan": [I32, I32, I32, I32, I32, P, P, P],
    "ytvln_option_count": [],
    "ytvln_option_name": [I32],
    "ytvln_set_option": [P, I32],
    "ytvln_get_option": [P, P],
    "ytvln_gemm_plan_x3": [I32, I32, I32, I32, I32, P, P, P],
    "ytvln_gemm_bf16_workspace_elems": [I32, I32, I32, I32],
    "ytvln_gemm_bf16": [P, I64, I32, P, I64, I32, P, I64, I32, P, P, I64, I32, I32, I32, I32, F32, P, I64, I32, P, P, P],
    "ytvln_cast_f32_bf16": [P, I64, I64, I32, P, I64, P],
    "ytvln_ln_fwd_bf16": [P, P, P, P, P, P, P, P, I64, I32, F32, F32, F32, P, I64, P],
    "ytvln_ln_bwd_bf16": [P, P, P, P, P, P, P, P, P, I64, I32, F32, F32, P, I64, P],
    "ytvln_text_embed_fwd_bf16": [P, P, P, P, P, P, P, P, P, P, P, I64, I32, I32, F32, F32, P, I64, P],
    "ytvln_image_embed_fwd_bf16": [P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, I64, I32, F32, F32, P, I64, P],
    "ytvln_act_bwd_bf16": [P, P, P, I64, I32, P],
    "ytvln_ce_fwd_bf16": [P, I64, P, I64, P, P, P, I32, I32, P],
    "ytvln_ce_bwd_bf16": [P, I64, P, I64, P, P, P, P, I64, I32, I32, P],
    "ytvln_kl_fwd_bf16": [P, I64, P, I64, P, P, P, P, I32, I32, P],
    "ytvln_kl_bwd_bf16": [P, I64, P, I64, P, P, P, P, P, I64, I32, I32, P],
    "ytvln_adamw_f32_bf16copy": [P, P, P, P, P, P, I32, P, F32, P],
    "ytvln_ln_fwd_f32": [P, P, P, P, P, P, P, P, I64, I32, F32, F32, F32, P, I64, P],
    "ytvln_ln_bwd_blocks": [I64],
    "ytvln_ln_bwd_f32": [P, P, P, P, P, P, P, P, I64, I32, F32, F32, P, I64, P],
    "ytvln_text_embed_fwd_f32": [P, P, P, P, P, P, P, P, P, P, P, I64, I32, I32, F32, F32, P, I64, P],
    "ytvln_image_embed_fwd_f32": [P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, I64, I32, F32, F32, P, I64, P],
    "ytvln_act_bwd_f32": [P, P, P, I64, I32, P],
    "ytvln_dropout_f32": [P, P, I64, F32, P, I64, P],
    "ytvln_attn_fwd_f32": [P, I64, P, I64, P, I64, P, P, I64, P, I32, I32, I32, I32, I32, F32, F32, P, I64, P],
    "ytvln_attn_bwd_f32": [P, I64, P, I64, P, I64, P, P, P, I64, P, P, P, I64, P, I64, P, I64, I32, I32, I32, I32, I32,
                           F32, F32, P, I64, P],
    "ytvln_attn_probs_f32": [P, I64, P, I64, P, P, P, I32, I32, I32, I32, I32, F32, P],
    "ytvln_ce_fwd_f32": [P, I64, P, I64, P, P, P, I32, I32, P],
    "ytvln_ce_bwd_f32": [P, I64, P, I64, P, P, P, P, I64, I32, I32, P],
    "ytvln_kl_fwd_f32": [P, I64, P, I64, P, P, P, P, I32, I32, P],
    "ytvln_kl_bwd_f32": [P, I64, P, I64, P, P, P, P, P, I64, I32, I32, P],
    "ytvln_bce_fwd_f32": [P, P, P, P, I32, P],
    "ytvln_bce_bwd_f32": [P, P, P, P, P, I32, P],
    "ytvln_adamw_f32": [P, P, P, P, P, I32, P, F32, P],
    # RCCL binding (csrc/rccl.hip): host calls, the communicator is an opaque handle
    "ytvln_rccl_load": [P],
    "ytvln_rccl_library_path": [],
    "ytvln_rccl_version": [P],
    "ytvln_rccl_unique_id": [P, I64],
    "ytvln_rccl_init": [P, P, I64, I32, I32, I32],
    "ytvln_rccl_allreduce": [P, P, I64, I32, I32, P],
    "ytvln_rccl_allreduce_slices_f32": [P, P, P, P, I32, P],
    "ytvln_rccl_allreduce_slices": [P, P, I32, P, P, I32, P],
    "ytvln_set_host_wait": [I32, I32],
    "ytvln_attn_problem_size": [],
    "ytvln_rccl_broadcast": [P, P, I64, I32, P],
    "ytvln_rccl_async_error": [P],
    "ytvln_rccl_destroy": [P],
}
RESTYPES = {"ytvln_attn_problem_size": I64, "ytvln_gemm_workspace_elems": I64, "ytvln_attn_keep_bytes": I64, "ytvln_gemm_sk_ctl_elems": I64, "ytvln_gemm_bf16_workspace_elems": I64, "ytvln_rccl_library_path": C.c_char_p, "ytvln_option_name": C.c_char_p}
DT_F32, DT_F64, DT_BF16, DT_I64, DT_U8 = 0, 1, 2, 3, 4
RED_SUM, RED_MAX, RED_MIN = 0, 1, 2
RCCL_UNIQUE_ID_BYTES = 128

EPI_NONE, EPI_GELU, EPI_RELU, EPI_MUL_DGELU, EPI_MUL_DRELU = 0, 1, 2, 3, 4
GEMM_A_ZERO_PADDED = 1
GEMM_SPLIT_BF16X3 = 2
ABI_VERSION = 2

_lib = None


def load():
    """dlopen the library once, bind every symbol, verify the ABI version."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise YtvlnLibraryError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(there is no CPU / PyTorch fallback for the compute path).")
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover
        raise YtvlnLibraryError(f"cannot load {LIB_PATH}: {e}") from e
    lib.ytvln_version.restype = I32
    lib.ytvln_version.argtypes = []
    lib.ytvln_last_error.restype = C.c_char_p
    lib.ytvln_last_error.argtypes = []
    for name, argtypes in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise YtvlnLibraryError(f"{LIB_PATH} does not export {name}") from e
        fn.argtypes = argtypes
        fn.restype = RESTYPES.get(name, I32)
    if lib.ytvln_version() != ABI_VERSION:
        raise YtvlnLibraryError(f"ABI mismatch: library {lib.ytvln_version()} != binding {ABI_VERSION}")
    if lib.ytvln_attn_problem_size() != C.sizeof(AttnProblem):          # a record the library would read past
        raise YtvlnLibraryError(f"ytvln_attn_problem: library {lib.ytvln_attn_problem_size()} bytes != binding {C.sizeof(AttnProblem)}")
    _lib = lib
    _FN.update({name: getattr(lib, name) for name in SIGNATURES})
    return lib


_FN = {}


def call(name: str, *args):
    """Invoke an entry point; non-zero status -> RuntimeError carrying ytvln_last_error()."""
    fn = _FN.get(name)
    if fn is None:
        load()
        fn = _FN[name]
    rc = fn(*args)
    if rc != 0:
        raise RuntimeError(f"{name} failed ({rc}): {_lib.ytvln_last_error().decode(errors='replace')}")


def options() -> dict:
    """{name: current value} of every run-time option of the library (include/ytvln.h: ytvln_option_*)."""
    lib = load()
    out = {}
    for i in range(lib.ytvln_option_count()):
        name = lib.ytvln_option_name(i).decode()
        v = C.c_int(0)
        call("ytvln_get_option", name.encode(), C.byref(v))
        out[name] = v.value
    return out


_OPTION_CACHE = {}


def option(name: str) -> int:
    """Current value of one run-time option (cached on the Python side; `set_option` is the only writer after the first read)."""
    v = _OPTION_CACHE.get(name)
    if v is None:
        c = C.c_int(0)
        call("ytvln_get_option", name.encode(), C.byref(c))
        v = _OPTION_CACHE[name] = c.value
    return v


def set_option(name: str, value: int) -> int:
    """Set a run-time option (kernel-form selection for tests / experiments); returns the previous value."""
    prev = C.c_int(0)
    call("ytvln_get_option", name.encode(), C.byref(prev))
    call("ytvln_set_option", name.encode(), int(value))
    _OPTION_CACHE.pop(name[6:] if name.startswith("YTVLN_") else name, None)
    return prev.value
