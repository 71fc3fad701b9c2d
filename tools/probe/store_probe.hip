// Epilogue access patterns of the 256 x 256 GEMM tile, all CUs at once (tools/probe, not product code).
// The ping-pong GEMM's epilogue is issue-bound (profiles/r03n_gemm8_phase_cycles.log: 64 four-byte stores per wave take ~170 cycles
// each); this probe measures, with the same geometry (256 persistent workgroups of 8 waves, one per CU, 128 KB of LDS claimed),
// what each candidate pattern costs per tile:
//   fp16 outputs  V0  64 x global_store_dword   wave tile 128 x 64, lane = columns 2 li, 2 li + 1 (the shipped epilogue)
//                 V1  64 x buffer_store_dword   same addresses through a buffer resource (32-bit offsets)
//                 V2  32 x global_store_dwordx2 wave tile 64 x 128, lane = columns 4 li .. 4 li + 3 (half wave = 256 contiguous bytes)
//                 V3  32 x buffer_store_dwordx2
//   fp32 residual V4 128 x global_store_dword   wave tile 128 x 64, lane = column tn * 32 + li (the shipped resid_io)
//                 V5  64 x global_store_dwordx2 wave tile 128 x 64, columns 2 li, 2 li + 1
//                 V6  32 x global_store_dwordx4 wave tile 64 x 128, columns 4 li .. 4 li + 3 (half wave = 512 contiguous bytes)
//                 V7  32 x buffer_store_dwordx4
//   fp32 residual loads V8 128 x global_load_dword, V9 32 x global_load_dwordx4, V10 32 x buffer_load_dwordx4 (same maps as V4 / V6)
// mode 0: the accesses back to back (T tiles per workgroup); mode 1: ~`mf` MFMAs per wave between two tiles' accesses (a K loop's worth),
// so that stores that only need to be ISSUED can drain under the next tile's matrix work.
// build: hipcc --offload-arch=gfx950 -O3 tools/probe/store_probe.hip -o tools/probe/store_probe.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

typedef _Float16 f16;
typedef f16 f16x2 __attribute__((ext_vector_type(2)));
typedef f16 f16x4 __attribute__((ext_vector_type(4)));
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, (int)bytes, 0x00020000);
}

template <int V>
__global__ __launch_bounds__(512) void probe(char *out, int ld, int tilesN, int T, int mf, const f16 *src, long long *stamps, float *sink) {
    extern __shared__ char smem[];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int li = lane & 31, lh = lane >> 5;
    constexpr bool MIRROR = V == 2 || V == 3 || V == 6 || V == 7 || V == 9 || V == 10;     // wave tile 64 x 128 (4 x 2 waves) instead of 128 x 64 (2 x 4)
    constexpr bool F32 = V >= 4;
    constexpr int ES = F32 ? 4 : 2;
    const int wm = MIRROR ? (wave >> 2) * 128 + ((wave >> 1) & 1) * 64 : (wave >> 2) * 128;
    const int wn = MIRROR ? (wave & 1) * 128 : (wave & 3) * 64;
    f32x16 acc[8];
    f16x8 a[2], b[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        a[i] = *(const f16x8 *)(src + ((i * 64 + lane) * 8));
        b[i] = *(const f16x8 *)(src + (((2 + i) * 64 + lane) * 8));
    }
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = (float)(i * 16 + r + lane);
    if (tid == 0) smem[0] = 1;
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(out, 0xFFFFF000u);
    long long issue = 0, t_begin = __builtin_readcyclecounter();
    for (int t = 0; t < T; ++t) {
        const int tile = t * gridDim.x + blockIdx.x;
        const int tm = (tile / tilesN) % 300, tn = tile % tilesN;
        const int m0 = tm * 256 + wm, n0 = tn * 256 + wn;
        if (mf > 0) {
            for (int it = 0; it < mf / 8; ++it) {
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i & 1], b[(i >> 1) & 1], acc[i], 0, 0, 0);
            }
        }
        const long long c0 = __builtin_readcyclecounter();
        // 128 values per lane = acc[8][16]; row of register r inside a 32-row MFMA tile
        if constexpr (V == 0 || V == 1) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + q * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh, n = n0 + 2 * li;
                    f16x2 o; o[0] = (f16)acc[2 * q][r]; o[1] = (f16)acc[2 * q + 1][r];
                    const size_t off = ((size_t)m * ld + n) * 2;
                    if constexpr (V == 0) *(f16x2 *)(out + off) = o;
                    else __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, o), rs, (unsigned)off, 0, 0);
                }
        } else if constexpr (V == 2 || V == 3) {
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + q * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh, n = n0 + 4 * li;
                    f16x4 o;
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = (f16)acc[4 * q + j][r];
                    const size_t off = ((size_t)m * ld + n) * 2;
                    if constexpr (V == 2) *(f16x4 *)(out + off) = o;
                    else __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, o), rs, (unsigned)off, 0, 0);
                }
        } else if constexpr (V == 4) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int m = m0 + q * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh, n = n0 + j * 32 + li;
                        *(float *)(out + ((size_t)m * ld + n) * 4) = acc[2 * q + j][r];
                    }
        } else if constexpr (V == 5) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + q * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh, n = n0 + 2 * li;
                    f32x2 o = {acc[2 * q][r], acc[2 * q + 1][r]};
                    *(f32x2 *)(out + ((size_t)m * ld + n) * 4) = o;
                }
        } else if constexpr (V == 6 || V == 7) {
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + q * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh, n = n0 + 4 * li;
                    f32x4 o = {acc[4 * q][r], acc[4 * q + 1][r], acc[4 * q + 2][r], acc[4 * q + 3][r]};
                    const size_t off = ((size_t)m * ld + n) * 4;
                    if constexpr (V == 6) *(f32x4 *)(out + off) = o;
                    else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), rs, (unsigned)off, 0, 0);
                }
        } else if constexpr (V == 8) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int m = m0 + q * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh, n = n0 + j * 32 + li;
                        acc[2 * q + j][r] += *(const float *)(out + ((size_t)m * ld + n) * 4);
                    }
        } else if constexpr (V == 9 || V == 10) {
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + q * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh, n = n0 + 4 * li;
                    const size_t off = ((size_t)m * ld + n) * 4;
                    f32x4 x;
                    if constexpr (V == 9) x = *(const f32x4 *)(out + off);
                    else x = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)off, 0, 0));
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[4 * q + j][r] += x[j];
                }
        }
        if constexpr (V >= 8) {               // loads: the values must have arrived before the tile counts as done
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("" ::"v"(acc[i]));
        }
        issue += __builtin_readcyclecounter() - c0;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long t_end = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 12345.678f) sink[0] = s + (float)ES;
    if (lane == 0) { stamps[(blockIdx.x * 8 + wave) * 2] = issue; stamps[(blockIdx.x * 8 + wave) * 2 + 1] = t_end - t_begin; }
}

template <int V>
static void run(const char *name, char *out, int ld, int tilesN, int T, int mf, const f16 *src, long long *stamps, float *sink, int cus, int ninstr, double bytes_per_tile) {
    auto kern = probe<V>;
    CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    std::vector<long long> st(cus * 16);
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(kern, dim3(cus), dim3(512), 131072, 0, out, ld, tilesN, T, mf, src, stamps, sink);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && ms < best) { best = ms; CK(hipMemcpy(st.data(), stamps, cus * 128, hipMemcpyDeviceToHost)); }
    }
    std::vector<double> is, tot;
    for (int i = 0; i < cus * 8; ++i) { is.push_back((double)st[2 * i] / T); tot.push_back((double)st[2 * i + 1] / T); }
    std::sort(is.begin(), is.end()); std::sort(tot.begin(), tot.end());
    printf("%-44s mf %4d: %8.2f us/tile wall, per wave and tile: access issue %7.0f cyc (%5.0f per instruction), tile total %7.0f cyc; %.2f TB/s chip\n", name, mf,
           best * 1e3 / T, is[is.size() / 2], is[is.size() / 2] / ninstr, tot[tot.size() / 2], bytes_per_tile * cus * T / (best * 1e-3) * 1e-12);
    fflush(stdout);
}

int main(int argc, char **argv) {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount, T = 16;
    const int M = 78336 + 512;
    char *out; f16 *src; long long *stamps; float *sink;
    const size_t bytes = (size_t)M * 4096 * 2;                 // fp16 [M, 4096] = fp32 [M, 2048] >= fp32 [M, 1024]
    CK(hipMalloc(&out, bytes)); CK(hipMemset(out, 0, bytes));
    CK(hipMalloc(&src, 1 << 16)); CK(hipMemset(src, 0, 1 << 16));
    CK(hipMalloc(&stamps, cus * 128)); CK(hipMalloc(&sink, 64));
    printf("# %s, %d CUs; %d tiles of 256 x 256 per workgroup, one workgroup per CU\n", prop.name, cus, T);
    for (int mode = 0; mode < 2; ++mode) {
        const int mf = mode ? 512 : 0;
        run<0>("fp16 64 x global_store_dword (shipped)", out, 4096, 16, T, mf, src, stamps, sink, cus, 64, 131072.0);
        run<1>("fp16 64 x buffer_store_dword", out, 4096, 16, T, mf, src, stamps, sink, cus, 64, 131072.0);
        run<2>("fp16 32 x global_store_dwordx2 (64x128 wave)", out, 4096, 16, T, mf, src, stamps, sink, cus, 32, 131072.0);
        run<3>("fp16 32 x buffer_store_dwordx2 (64x128 wave)", out, 4096, 16, T, mf, src, stamps, sink, cus, 32, 131072.0);
        run<4>("fp32 128 x global_store_dword (shipped)", out, 1024, 4, T, mf, src, stamps, sink, cus, 128, 262144.0);
        run<5>("fp32 64 x global_store_dwordx2", out, 1024, 4, T, mf, src, stamps, sink, cus, 64, 262144.0);
        run<6>("fp32 32 x global_store_dwordx4 (64x128 wave)", out, 1024, 4, T, mf, src, stamps, sink, cus, 32, 262144.0);
        run<7>("fp32 32 x buffer_store_dwordx4 (64x128 wave)", out, 1024, 4, T, mf, src, stamps, sink, cus, 32, 262144.0);
        run<8>("fp32 128 x global_load_dword (shipped)", out, 1024, 4, T, mf, src, stamps, sink, cus, 128, 262144.0);
        run<9>("fp32 32 x global_load_dwordx4 (64x128 wave)", out, 1024, 4, T, mf, src, stamps, sink, cus, 32, 262144.0);
        run<10>("fp32 32 x buffer_load_dwordx4 (64x128 wave)", out, 1024, 4, T, mf, src, stamps, sink, cus, 32, 262144.0);
    }
    return 0;
}
