"""Micro-benchmark (round 3): is the fp32 matrix pipe of MI355X power-limited, and which instruction delivers more FLOP/s under the cap?

Writes scratch/mfma_power_bench.hip: hand-placed loops of v_mfma_f32_32x32x2_f32 or v_mfma_f32_16x16x4_f32 (same FLOPs per iteration: 8192
pipe cycles) on 16 A and 16 B operand registers holding pseudo-random floats (so the multiplier inputs toggle like real data), 256 or 512
threads per workgroup, one workgroup per CU, every CU busy.  Prints wall time per iteration, the shader clock measured inside the kernel
(s_memtime cycles / s_memrealtime) and the resulting TFLOP/s.  Constant operands (ZERO=1) show the unthrottled rate.
"""
import os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
variants = [(mf, "a", 256, 0, R) for R in (0, 16) for mf in ("32", "16")] * 3


def kernel(idx, mf, accf, threads, zero, R=0):
    n_mfma = 128 if mf == "32" else 256
    acc_regs = 16 if mf == "32" else 4
    n_acc = 8 if mf == "32" else 32           # 128 accumulator registers either way
    base = 0 if accf == "a" else 128
    body = []
    for k in range(n_mfma):
        a = k % n_acc
        lo = base + a * acc_regs
        reg = f"{accf}[{lo}:{lo + acc_regs - 1}]"
        op = "v_mfma_f32_32x32x2_f32" if mf == "32" else "v_mfma_f32_16x16x4_f32"
        body.append(f"{op} {reg}, v{20 + (k * 7 + k // n_acc) % 16}, v{36 + (k * 5 + k // n_acc) % 16}, {reg}")
        per = n_mfma // R if R else 0
        if R and k % per == 0:      # operand registers refilled from (random) LDS data: 8 x 16-byte reads rotate over v20..v51
            j = (k // per) % 8
            body.append(f"ds_read_b128 v[{20 + 4 * j}:{23 + 4 * j}], %[addr] offset:{((k // per) % 16) * 1024}")
    asm = ["s_mov_b32 s20, %[iters]", "s_memtime s[22:23]", "s_memrealtime s[24:25]", "s_waitcnt lgkmcnt(0)", "LA%=:"] + body + \
          ["s_waitcnt lgkmcnt(0)", "s_sub_u32 s20, s20, 1", "s_cmp_lg_u32 s20, 0", "s_cbranch_scc1 LA%=", "s_memtime s[26:27]", "s_memrealtime s[28:29]", "s_waitcnt lgkmcnt(0)",
           "s_sub_u32 s26, s26, s22", "s_subb_u32 s27, s27, s23", "s_sub_u32 s28, s28, s24", "s_subb_u32 s29, s29, s25",
           "v_mov_b32 %[c0], s26", "v_mov_b32 %[c1], s28", "s_nop 15", "s_nop 15"]
    asm += [f"v_accvgpr_read_b32 %[res], a0" if accf == "a" else "v_mov_b32 %[res], v128"]
    clob = [f"v{r}" for r in range(20, 52)] + [f"v{r}" for r in range(128, 256)] + [f"a{r}" for r in range(0, 128)] + \
           [f"s{r}" for r in range(20, 30)] + ["scc", "memory"]
    init = "\\n\\t".join(f"v_mov_b32 v{20 + r}, %{r}" for r in range(32))
    ins = ", ".join(f'"v"(x[{r}])' for r in range(32))
    text = "\\n\\t".join(asm)
    src = f"""
__global__ __launch_bounds__({threads}, {threads // 256}) void k{idx}(float* __restrict__ out, unsigned* __restrict__ clk, int iters) {{
    __shared__ __attribute__((aligned(16))) float pad[24576];
    {{ unsigned q = threadIdx.x * 747796405u + 2891336453u;
      for (int i = threadIdx.x; i < 24576; i += {threads}) {{ q = q * 1664525u + 1013904223u; pad[i] = {"0.f" if zero else "((int)(q >> 9) - (1 << 22)) * (1.0f / (1 << 22))"}; }} }}
    __syncthreads();
    unsigned addr = (threadIdx.x & 63) * 16;
    float x[32];
    unsigned h = threadIdx.x * 2654435761u + 12345u;
    for (int r = 0; r < 32; ++r) {{ h = h * 1664525u + 1013904223u; x[r] = {"0.f" if zero else "((int)(h >> 9) - (1 << 22)) * (1.0f / (1 << 22))"}; }}
    float res; unsigned c0, c1;
    asm volatile("{init}" :: {ins} : {", ".join('"v%d"' % (20 + r) for r in range(32))});
    asm volatile("{text}" : [res] "=v"(res), [c0] "=v"(c0), [c1] "=v"(c1) : [iters] "s"(iters), [addr] "v"(addr) : {", ".join('"%s"' % c for c in clob)});
    out[blockIdx.x * {threads} + threadIdx.x] = res + pad[0];
    if (threadIdx.x == 0) {{ clk[2 * blockIdx.x] = c0; clk[2 * blockIdx.x + 1] = c1; }}
}}
"""
    return f"k{idx}", f"mfma{mf} acc={accf} thr={threads} {'zero' if zero else 'rand'} R={R}", src, threads


parts = ["#include <hip/hip_runtime.h>\n#include <stdio.h>\n#include <stdint.h>\n"]
main = ["int main() {", "  float* o; unsigned* c; hipMalloc(&o, 4 << 20); hipMalloc(&c, 4096);",
        "  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); const int iters = 120000; float ms; unsigned hc[512];"]
for i, v in enumerate(variants):
    n, d, s, thr = kernel(i, *v)
    parts.append(s)
    main.append(f'  hipLaunchKernelGGL({n}, dim3(256), dim3({thr}), 0, 0, o, c, 200); hipDeviceSynchronize();')
    main.append(f'  hipEventRecord(e0); hipLaunchKernelGGL({n}, dim3(256), dim3({thr}), 0, 0, o, c, iters); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);')
    main.append(f'  hipMemcpy(hc, c, 2048, hipMemcpyDeviceToHost);')
    main.append(f'  printf("%-30s %8.2f us/iter  clock %.3f GHz  cycles/iter %7.0f (ideal {8192 * (thr // 256)})  %6.1f TFLOP/s\\n", "{d}", ms * 1e3 / iters, hc[0] / (hc[1] * 10.0), (double)hc[0] / iters, 256.0 * {thr // 64} * 524288.0 * iters / (ms * 1e-3) / 1e12);')
main.append("  return 0; }")
parts.append("\n".join(main))
open(os.path.join(ROOT, "scratch", "mfma_power_bench.hip"), "w").write("\n".join(parts))
print(len(variants), "variants")
