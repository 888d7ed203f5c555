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
        cmd = [hipcc, *FLAGS, "-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stderr[-4000:]}")
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
