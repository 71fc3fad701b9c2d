// The 256 x 128 ping-pong kernel's instantiations (gemm_n128.h): convolutions.
#include "gemm_n128.h"

int pb_gemm_n128_conv_f16(hipStream_t s, const GemmArgs &a) { return launch_g8n<A_CONV, EPI_STD, false>(s, a); }
int pb_gemm_n128_conv_mx(hipStream_t s, const GemmArgs &a) { return launch_g8n<A_CONV, EPI_STD, true>(s, a); }
