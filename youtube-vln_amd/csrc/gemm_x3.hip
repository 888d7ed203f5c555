// Translation unit of the three-term (fp32 as hi + mid + lo bf16, YTVLN_GEMM_SPLIT_BF16X3) instantiations of gemm_dma_kernel.
// The kernel template and everything it needs live in gemm.hip; compiling them here lets the two halves build in parallel.
#define YT_GEMM_X3_TU 1
#include "gemm.hip"
