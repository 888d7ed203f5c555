"""Micro-benchmark generator (round 3): what do LDS reads / LDS-DMA issues cost the fp32 matrix pipe of a gfx950 SIMD?

Writes scratch/mfma_lds_bench.hip: one kernel per variant, each a hand-placed inline-asm loop of N matrix instructions (1024 pipe cycles
per iteration) with R `ds_read_b128` and G `global_load_lds_dwordx4` placed one behind every matrix instruction, accumulators in the
AGPR or the VGPR half of the register file, v_mfma_f32_32x32x2_f32 or v_mfma_f32_16x16x4_f32, one or two waves per SIMD (256 / 512
threads, one workgroup per CU).  The host times each kernel with HIP events and prints pipe cycles per iteration at 2.4 GHz.
Build + run:  python tools/lab/gen_mfma_lds_bench.py && hipcc -O2 --offload-arch=gfx950 scratch/mfma_lds_bench.hip -o scratch/mfma_lds_bench
"""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
variants = []
for mf in ("32", "16"):
    for accf in ("a", "v"):
        for threads in (256, 512):
            for (R, G, W) in ((0, 0, 0), (4, 0, 0), (8, 0, 0), (16, 0, 0), (0, 2, 0), (0, 4, 0), (8, 4, 0), (0, 0, 8)):
                variants.append((mf, accf, threads, R, G, W))
# reads issued by ONE wave of the pair only (the other runs matrix instructions only): role split, 512 threads
for mf in ("32",):
    for accf in ("a", "v"):
        for (R, G) in ((8, 0), (16, 0), (0, 4)):
            variants.append((mf, accf, 512, R, G, -1))


def kernel(idx, mf, accf, threads, R, G, W):
    split = W == -1
    W = max(W, 0)
    n_mfma = 16 if mf == "32" else 32
    acc_regs = 16 if mf == "32" else 4
    n_acc = 2 if mf == "32" else 8
    base = 0 if accf == "a" else 128          # a[0:..] or v[128:..]
    lines = []
    extra = []                                  # (kind, k) placed behind matrix instruction number k
    for r in range(R):
        extra.append(("r", r))
    for g_ in range(G):
        extra.append(("g", g_))
    for w in range(W):
        extra.append(("w", w))
    per = max(1, n_mfma // max(1, len(extra))) if extra else 0

    def body(with_extra):
        out = []
        e = 0
        for k in range(n_mfma):
            a = k % n_acc
            lo = base + a * acc_regs
            reg = f"{accf}[{lo}:{lo + acc_regs - 1}]"
            op = "v_mfma_f32_32x32x2_f32" if mf == "32" else "v_mfma_f32_16x16x4_f32"
            out.append(f"{op} {reg}, v{20 + (k % 4)}, v{24 + (k % 4)}, {reg}")
            if with_extra and extra and k % per == 0 and e < len(extra):
                kind, j = extra[e]; e += 1
                if kind == "r":
                    out.append(f"ds_read_b128 v[{40 + 4 * (j % 16)}:{43 + 4 * (j % 16)}], %[addr] offset:{(j % 16) * 1024}")
                elif kind == "w":
                    out.append(f"ds_write_b128 %[addr], v[{40 + 4 * (j % 8)}:{43 + 4 * (j % 8)}] offset:{32768 + (j % 8) * 1024}")
                else:
                    out.append(f"s_add_u32 m0, %[ldsb], {16384 + (j % 8) * 1024}")
                    out.append("global_load_lds_dwordx4 %[gp], off")
        while with_extra and e < len(extra):
            kind, j = extra[e]; e += 1
            out.append(f"ds_read_b128 v[{40 + 4 * (j % 16)}:{43 + 4 * (j % 16)}], %[addr] offset:{(j % 16) * 1024}" if kind == "r" else "s_nop 0")
        return out

    asm = ["s_mov_b32 s20, %[iters]"]
    if split:
        asm += ["s_cmp_lg_u32 %[role], 0", "s_cbranch_scc1 LB%=", "LA%=:"] + body(True) + \
               ["s_waitcnt vmcnt(0) lgkmcnt(0)", "s_sub_u32 s20, s20, 1", "s_cmp_lg_u32 s20, 0", "s_cbranch_scc1 LA%=", "s_branch LE%=",
                "LB%=:"] + body(False) + ["s_sub_u32 s20, s20, 1", "s_cmp_lg_u32 s20, 0", "s_cbranch_scc1 LB%=", "LE%=:"]
    else:
        asm += ["LA%=:"] + body(True) + ["s_waitcnt vmcnt(0) lgkmcnt(0)", "s_sub_u32 s20, s20, 1", "s_cmp_lg_u32 s20, 0", "s_cbranch_scc1 LA%="]
    asm += ["s_nop 15", "s_nop 15"]
    if accf == "a":
        asm += [f"v_accvgpr_read_b32 v30, a0"]
    else:
        asm += [f"v_mov_b32 v30, v128"]
    asm += ["v_mov_b32 %[res], v30"]
    clob = [f"v{r}" for r in range(20, 104)] + [f"v{r}" for r in range(128, 160)] + [f"a{r}" for r in range(0, 32)] + ["s20", "scc", "m0", "memory"]
    text = "\\n\\t".join(asm)
    name = f"k{idx}"
    src = f"""
__global__ __launch_bounds__({threads}, {threads // 256}) void {name}(const float* __restrict__ gsrc, float* __restrict__ out, int iters) {{
    __shared__ __attribute__((aligned(16))) float lds[24576];
    for (int i = threadIdx.x; i < 24576; i += {threads}) lds[i] = 1.0f + 0.001f * (float)(i & 255);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned addr = (unsigned)(lane * 16);
    unsigned ldsb = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)lds + wave * 0;
    int role = wave >> 2;
    const float* gp = gsrc + (blockIdx.x * {threads // 64} + wave) * 2048 + lane * 4;
    float res = 0.f;
    float seed = 0.5f + 0.01f * (float)lane;
    asm volatile("v_mov_b32 v20, %0\\n\\tv_mov_b32 v21, %0\\n\\tv_mov_b32 v22, %0\\n\\tv_mov_b32 v23, %0\\n\\tv_mov_b32 v24, %0\\n\\tv_mov_b32 v25, %0\\n\\tv_mov_b32 v26, %0\\n\\tv_mov_b32 v27, %0"
                 :: "v"(seed) : "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27");
    asm volatile("{text}"
                 : [res] "=v"(res)
                 : [addr] "v"(addr), [iters] "s"(iters), [ldsb] "s"(ldsb), [gp] "v"(gp), [role] "s"(role)
                 : {", ".join('"%s"' % c for c in clob)});
    out[blockIdx.x * {threads} + threadIdx.x] = res;
}}
"""
    desc = f"mfma{mf} acc={accf} thr={threads} R={R} G={G} W={W}" + (" split" if split else "")
    return name, desc, src


parts = ["#include <hip/hip_runtime.h>\n#include <stdio.h>\n#include <stdint.h>\n"]
table = []
for i, v in enumerate(variants):
    n, d, s = kernel(i, *v)
    parts.append(s)
    table.append((n, d, v[2]))
main = ["int main() {", "  float *g, *o; hipMalloc(&g, 64 << 20); hipMalloc(&o, 4 << 20); hipMemset(g, 0, 64 << 20);",
        "  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); const int iters = 4000; float ms;"]
for n, d, thr in table:
    main.append(f'  hipLaunchKernelGGL({n}, dim3(256), dim3({thr}), 0, 0, g, o, 100); hipDeviceSynchronize();')
    main.append(f'  hipEventRecord(e0); hipLaunchKernelGGL({n}, dim3(256), dim3({thr}), 0, 0, g, o, iters); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);')
    main.append(f'  printf("%-44s %8.1f cycles/iter (ideal {1024 * (thr // 256)})  err=%d\\n", "{d}", ms * 1e-3 * 2.4e9 / iters, (int)hipGetLastError());')
main.append("  return 0; }")
parts.append("\n".join(main))
os.makedirs(os.path.join(ROOT, "scratch"), exist_ok=True)
open(os.path.join(ROOT, "scratch", "mfma_lds_bench.hip"), "w").write("\n".join(parts))
print(len(variants), "variants")
