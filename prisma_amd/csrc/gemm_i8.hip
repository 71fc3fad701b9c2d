// The 256 x 128 ping-pong kernel's instantiations (gemm_n128.h): dense GEMMs.
#include "gemm_n128.h"

int pb_gemm_n128_dense_f16(hipStream_t s, const GemmArgs &a) { return launch_g8n<A_DENSE, EPI_STD, false>(s, a); }
int pb_gemm_n128_dense_mx(hipStream_t s, const GemmArgs &a) { return launch_g8n<A_DENSE, EPI_STD, true>(s, a); }
