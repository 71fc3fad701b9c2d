// One translation unit of the GEMM kernel instantiations (the templates live in gemm_kernels.h; split so that make -j compiles them in parallel).
#include "gemm_kernels.h"

int pb_gemm_conv_pixshuf_f16(hipStream_t s, int tile, const GemmArgs &a) { return launch_tile<A_CONV, EPI_PIXSHUF, false>(s, tile, a); }
int pb_gemm_conv_pixshuf_mx(hipStream_t s, int tile, const GemmArgs &a) { return launch_tile<A_CONV, EPI_PIXSHUF, true>(s, tile, a); }
int pb_gemm_conv_head_f16(hipStream_t s, int tile, const GemmArgs &a) { return launch_tile<A_CONV, EPI_HEAD, false>(s, tile, a); }
int pb_gemm_conv_head_mx(hipStream_t s, int tile, const GemmArgs &a) { return launch_tile<A_CONV, EPI_HEAD, true>(s, tile, a); }
