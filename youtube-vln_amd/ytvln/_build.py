"""Build libytvln.so (gfx950) in-tree with hipcc.  Called by `__graft_entry__.build()`; cross-compiles without a GPU.

The library is placed at youtube-vln_amd/ytvln/lib/libytvln.so so it travels with the repo snapshot to the GPU box
(built objects are git-ignored, not gpurun-ignored).  Incremental: a source is recompiled only when it (or a header)
is newer than its object.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(HERE), "csrc")
INCLUDE = os.path.join(os.path.dirname(os.path.dirname(HERE)), "include")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(LIBDIR, "obj")
LIB = os.path.join(LIBDIR, "libytvln.so")
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-munsafe-fp-atomics", "-Wno-unused-result"]


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found; cannot build libytvln.so")


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _src_mtime(src: str) -> float:
    """mtime of a source and of the .hip files it includes (gemm_x3.hip is gemm.hip compiled with another macro set)."""
    t = os.path.getmtime(src)
    for line in open(src):
        line = line.strip()
        if line.startswith('#include "') and line.endswith('.hip"'):
            t = max(t, os.path.getmtime(os.path.join(CSRC, line[10:-1])))
    return t


def _headers_mtime() -> float:
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hs += [os.path.join(INCLUDE, f) for f in os.listdir(INCLUDE) if f.endswith(".h")]
    return max(os.path.getmtime(h) for h in hs)


# Kernels whose epilogues issue loads from hand-written asm and wait for them with a hand-counted s_waitcnt (gemm_tiles.h epilogue_interior,
# gemm_bf16.hip bf_epilogue_interior): the destination registers are only safe while the register allocator neither spills nor reloads anything
# between the load and the wait.  A kernel without a scratch segment cannot do either, so the build REFUSES one that has scratch.
AUDITED_KERNELS = ("gemm_dma_kernel", "gemm_bf16_kernel", "gemm_bf16_h_kernel", "gemm_bf16_w4_kernel", "gemm_sw_kernel", "gemm_sk_kernel")


# ... except where the kernel was written for it: the 128x128 form of the persistent kernel (opt-in GEMM_SK, 2-4 spilled registers at its 128-register
# budget) takes the compiler-counted epilogue loads (gemm_epilogue<..., HAND = false>) and issues its partial-tile loads together with their wait
SCRATCH_OK = ("gemm_sk_kernelILi128ELi128E",)


def parse_resource_usage(stderr: str) -> dict:
    """{mangled kernel name: {"vgprs", "agprs", "sgprs", "scratch", "vgpr_spill", "sgpr_spill", "lds", "occupancy"}} from the remarks of
    -Rpass-analysis=kernel-resource-usage."""
    import re
    out, cur = {}, None
    keys = {"VGPRs": "vgprs", "AGPRs": "agprs", "TotalSGPRs": "sgprs", "ScratchSize [bytes/lane]": "scratch", "VGPRs Spill": "vgpr_spill",
            "SGPRs Spill": "sgpr_spill", "LDS Size [bytes/block]": "lds", "Occupancy [waves/SIMD]": "occupancy"}
    for line in stderr.splitlines():
        m = re.search(r"remark: .*Function Name: (\S+)", line)
        if m:
            cur = out.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark:\s+([A-Za-z][^:]*): (\d+) \[-Rpass", line)
        if m and cur is not None and m.group(1).strip() in keys:
            cur[keys[m.group(1).strip()]] = int(m.group(2))
    return out


def audit_gemm_kernels(src: str, obj: str, stderr: str) -> None:
    import json
    usage = parse_resource_usage(stderr)
    with open(obj[:-2] + ".usage.json", "w") as f:
        json.dump(usage, f, indent=0, sort_keys=True)
    bad = {k: v for k, v in usage.items() if any(a in k for a in AUDITED_KERNELS) and not any(a in k for a in SCRATCH_OK)
           and (v.get("scratch", 0) or v.get("vgpr_spill", 0))}
    if bad:
        raise RuntimeError(f"{os.path.basename(src)}: kernels with hand-counted epilogue loads must not use scratch (a spill between such a load "
                           f"and its wait reads the register before the data lands): {bad}")
    if not any(any(a in k for a in AUDITED_KERNELS) for k in usage):
        raise RuntimeError(f"{os.path.basename(src)}: no kernel-resource-usage remarks parsed (compiler output format changed?)")


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(OBJDIR, exist_ok=True)
    hipcc = _hipcc()
    hm = _headers_mtime()
    jobs = []
    for src in sources():
        obj = os.path.join(OBJDIR, os.path.basename(src)[:-4] + ".o")
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(_src_mtime(src), hm):
            jobs.append((src, obj))

    def compile_one(job):
        src, obj = job
        audited = os.path.basename(src).startswith("gemm")
        cmd = [hipcc, *FLAGS, *(["-Rpass-analysis=kernel-resource-usage"] if audited else []), "-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stderr[-4000:]}")
        if audited:
            audit_gemm_kernels(src, obj, r.stderr)
        return src

    if jobs:
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4)) as ex:
            for done in ex.map(compile_one, jobs):
                if verbose:
                    print(f"[ytvln build] compiled {os.path.basename(done)}", file=sys.stderr)
    objs = [os.path.join(OBJDIR, os.path.basename(s)[:-4] + ".o") for s in sources()]
    if jobs or not os.path.exists(LIB) or os.path.getmtime(LIB) < max(os.path.getmtime(o) for o in objs):
        cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-Wl,-z,defs", *objs, "-o", LIB]   # unresolved symbols fail the link
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr[-4000:]}")
        if verbose:
            print(f"[ytvln build] linked {LIB}", file=sys.stderr)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
