// Shared device/host helpers for libprisma_bands (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <string>

typedef _Float16 f16;
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
// Q is stored pre-multiplied by head_dim^-0.5 * log2(e) (head_dim = 64): attention then needs a subtract + v_exp_f32 per score
#define PB_QSCALE (0.125f * 1.4426950408889634f)
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define PB_WAVE 64

// ---- error plumbing (host) -------------------------------------------------------------------
void pb_set_error(const char *fmt, ...);

#define PB_HIP(expr)                                                                          \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) {                                                               \
            pb_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return -2;                                                                        \
        }                                                                                     \
    } while (0)

#define PB_CHECK(cond, code, ...)                                                             \
    do {                                                                                      \
        if (!(cond)) {                                                                        \
            pb_set_error(__VA_ARGS__);                                                        \
            return (code);                                                                    \
        }                                                                                     \
    } while (0)

static inline int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }
// integer tuning knob from the environment (read where an engine is constructed; the default is the shipped value)
static inline int pb_env_int(const char *name, int dflt) {
    const char *e = getenv(name);
    return e && *e ? atoi(e) : dflt;
}

// ---- device helpers --------------------------------------------------------------------------
#if defined(__HIPCC__)
// Direct HBM -> LDS copy of 16 bytes per lane.  The LDS destination is wave-uniform `lds_base`
// + lane * 16 (hardware adds the lane term); the global source is per lane.
__device__ __forceinline__ void glds16(const void *gsrc, void *lds_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gsrc,
                                     (__attribute__((address_space(3))) void *)lds_base, 16, 0, 0);
}

// Same LDS-DMA through the buffer path: `rsrc` addresses the whole tensor, `voff` is this lane's byte offset, `soff` a
// wave-uniform byte offset (e.g. the K-tile advance, so the per-lane offsets stay loop invariant); out-of-range offsets
// read zeros.
__device__ __forceinline__ void glds16_buf(__amdgpu_buffer_rsrc_t rsrc, int voff, int soff, void *lds_base) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void *)lds_base, 16, voff, soff, 0, 0);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, (int)bytes, 0x00020000);
}

// Separately rounded IEEE operations for the byte-exact kernels (encodes, masks): numpy evaluates `a * b + c` as two rounded
// operations, while hipcc contracts it into an FMA by default (-ffp-contract=fast) - and HIP's __fmul_rn / __fadd_rn / ... are
// plain operators (contractable) and __fsqrt_rn is the NATIVE (approximate) square root unless OCML_BASIC_ROUNDED_OPERATIONS is
// defined.  These helpers carry `fp contract(off)` (the instructions keep no `contract` flag after inlining) and use the
// correctly rounded divide / sqrt (hipcc's default -fhip-fp32-correctly-rounded-divide-sqrt).
__device__ __forceinline__ float ex_fmul(float a, float b) {
#pragma clang fp contract(off)
    return a * b;
}
__device__ __forceinline__ float ex_fadd(float a, float b) {
#pragma clang fp contract(off)
    return a + b;
}
__device__ __forceinline__ float ex_fsub(float a, float b) {
#pragma clang fp contract(off)
    return a - b;
}
__device__ __forceinline__ float ex_fdiv(float a, float b) {
#pragma clang fp contract(off)
    return a / b;
}
__device__ __forceinline__ float ex_fsqrt(float a) { return __builtin_sqrtf(a); }
__device__ __forceinline__ double ex_dmul(double a, double b) {
#pragma clang fp contract(off)
    return a * b;
}
__device__ __forceinline__ double ex_dadd(double a, double b) {
#pragma clang fp contract(off)
    return a + b;
}
__device__ __forceinline__ double ex_dsub(double a, double b) {
#pragma clang fp contract(off)
    return a - b;
}
__device__ __forceinline__ double ex_ddiv(double a, double b) {
#pragma clang fp contract(off)
    return a / b;
}

// two / four floats -> OCP e4m3 bytes (saturating at +-448), for the fp8 copies the MX correction segments read (gemm.h nk16)
__device__ __forceinline__ unsigned short pb_fp8x2(float a, float b) {
    a = __builtin_amdgcn_fmed3f(a, -448.f, 448.f); b = __builtin_amdgcn_fmed3f(b, -448.f, 448.f);
    return (unsigned short)__builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
}
__device__ __forceinline__ int pb_fp8x4(float a, float b, float c, float d) {
    a = __builtin_amdgcn_fmed3f(a, -448.f, 448.f); b = __builtin_amdgcn_fmed3f(b, -448.f, 448.f);
    c = __builtin_amdgcn_fmed3f(c, -448.f, 448.f); d = __builtin_amdgcn_fmed3f(d, -448.f, 448.f);
    int r = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
    return __builtin_amdgcn_cvt_pk_fp8_f32(c, d, r, true);
}

__device__ __forceinline__ float gelu_erf(float x) {
    return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
#endif
