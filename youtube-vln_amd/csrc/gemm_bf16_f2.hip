// Translation unit of the FORM = 2 instantiations of gemm_bf16_kernel (see bf_launch_form in gemm_bf16.hip): the kernel template and everything
// it needs live there; compiling the forms in their own units lets them build in parallel.
#define YT_BF16_FORM_TU 2
#include "gemm_bf16.hip"
