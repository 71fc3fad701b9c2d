// One translation unit of the GEMM kernel instantiations (the templates live in gemm_kernels.h; split so that make -j compiles them in parallel).
#include "gemm_kernels.h"

int pb_gemm_dense_std_mx(hipStream_t s, int tile, const GemmArgs &a) { return launch_tile<A_DENSE, EPI_STD, true>(s, tile, a); }
