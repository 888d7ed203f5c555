// Translation unit of the one-wave-per-SIMD software-pipelined instantiations (gemm_sw_kernel, round 3; opt-in YTVLN_GEMM_SW=1).  The kernel
// template and everything it needs live in gemm.hip; compiling them here lets the parts of the GEMM code build in parallel.
#define YT_GEMM_SW_TU 1
#include "gemm.hip"
