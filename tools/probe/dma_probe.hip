// What one CU can move from the L2 into its LDS per cycle, by transport (tools/probe, not product code).
// Every GEMM kernel of the bands - 256 x 256 ping-pong, 256 x 128 ping-pong, generic 128 x 128 - runs its K loop at 27-31 bytes of
// LDS-DMA per cycle and CU whatever its MFMA share (profiles/r05a_n128_tile_stamps.txt); this probe asks whether that is the transport's ceiling.
//   V0  buffer_load_dwordx4 ... lds   (LDS-DMA, 1 KB per wave instruction; what the kernels use)
//   V1  buffer_load_dwordx4 -> VGPRs  (no LDS write)
//   V2  buffer_load_dwordx4 -> VGPRs -> ds_write_b128
//   V3  V0 with half of the waves reading the LDS (ds_read_b128) instead of loading   (fragment-read traffic beside the DMA)
//   V4  V0 with `mf` MFMAs per wave between bursts (matrix pipe busy beside the DMA)
//   V5  no global traffic at all: every wave issues ds_read_b128 (1 KB per wave instruction, conflict-free) back to back - what the LDS itself delivers
//   V6  V5 with half of the waves issuing LDS-DMA instead (fragment reads beside DMA writes, the GEMM kernels' mix)
//   V7  V6 with the loading waves going through registers (buffer_load -> VGPRs -> ds_write_b128) instead of LDS-DMA
// Geometry: one workgroup of NW waves per CU, each workgroup walks its own S-byte window of an L2-resident buffer (S * 32 CUs < 4 MB per XCD),
// `per` loads per wave back to back, then one wait; T rounds.  Prints bytes per cycle and CU (s_memtime) and GB/s (wall).
// build: hipcc --offload-arch=gfx950 -O3 tools/probe/dma_probe.hip -o tools/probe/dma_probe.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

typedef _Float16 f16;
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, (int)bytes, 0x00020000);
}

template <int V, int PER>
__global__ __launch_bounds__(512) void probe(const char *src, int S, int T, int mf, long long *stamps, float *sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int nw = blockDim.x >> 6;
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(src + (size_t)blockIdx.x * S, (unsigned)S);
    f32x16 acc[4];
    f16x8 a = {(f16)1.f, (f16)0.5f, (f16)0.25f, (f16)2.f, (f16)1.f, (f16)0.5f, (f16)0.25f, (f16)2.f};
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = (float)(lane + r);
    f32x4 keep = {0.f, 0.f, 0.f, 0.f};
    const bool reader = (V == 3 || V == 6 || V == 7) && (wave & 1);
    const int loaders = (V == 3 || V == 6 || V == 7) ? nw / 2 : nw, lw = (V == 3 || V == 6 || V == 7) ? wave >> 1 : wave;
    if constexpr (V == 5 || V == 6 || V == 7) {
        if (V == 5 || reader) {
            // PER ds_read_b128 per round, straight into registers nobody reads (asm volatile: they are issued), one wait per round
            const unsigned a0 = (unsigned)(size_t)(smem) + lane * 16 + wave * 1024;
            const long long t0 = __builtin_readcyclecounter();
            for (int t = 0; t < T; ++t) {
#pragma unroll
                for (int i = 0; i < PER; ++i) {
                    f32x4 v;
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(a0), "n"((i & 7) * 4096));
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            const long long t1 = __builtin_readcyclecounter();
            if (tid == 64 * (V == 5 ? 0 : 1)) { stamps[blockIdx.x * 2] = t0; stamps[blockIdx.x * 2 + 1] = t1; }
            return;
        }
    }
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    int off = lw * PER * 1024;                               // this wave's first byte of the round
    for (int t = 0; t < T; ++t) {
        if (reader) {
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                const f32x4 v = *(const f32x4 *)(smem + ((i * 8 + wave) & 63) * 1024 + lane * 16);
                keep += v;
            }
        } else if constexpr (V == 0 || V == 3 || V == 4 || V == 6) {
#pragma unroll
            for (int i = 0; i < PER; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)(smem + ((i * nw + wave) & 63) * 1024), 16,
                                                         lane * 16, off + i * 1024, 0, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            f32x4 v[PER];
#pragma unroll
            for (int i = 0; i < PER; ++i) v[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, off + i * 1024, 0));
            if constexpr (V == 2 || V == 7) {
#pragma unroll
                for (int i = 0; i < PER; ++i) *(f32x4 *)(smem + ((i * nw + wave) & 63) * 1024 + lane * 16) = v[i];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            } else {
#pragma unroll
                for (int i = 0; i < PER; ++i) keep += v[i];
            }
        }
        if (V == 4 && mf > 0) {
            for (int it = 0; it < mf / 4; ++it) {
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, a, acc[i], 0, 0, 0);
            }
        }
        off += loaders * PER * 1024;
        off = off + loaders * PER * 1024 > S ? lw * PER * 1024 : off;
    }
    const long long t1 = __builtin_readcyclecounter();
    if (tid == 0 && V != 6 && V != 7) { stamps[blockIdx.x * 2] = t0; stamps[blockIdx.x * 2 + 1] = t1; }
    float s = keep[0] + keep[1] + keep[2] + keep[3];
#pragma unroll
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][5];
    if (s == 12345.678f) sink[tid] = s + smem[tid];
}

template <int V, int PER>
void run(const char *name, const char *src, int S, int nw, int T, int mf, long long *d_st, float *d_sink, int ncu) {
    auto k = probe<V, PER>;
    CK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k, dim3(ncu), dim3(nw * 64), 65536, 0, src, S, 8, mf, d_st, d_sink);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k, dim3(ncu), dim3(nw * 64), 65536, 0, src, S, T, mf, d_st, d_sink);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<long long> st(ncu * 2);
    CK(hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> cyc;
    for (int i = 0; i < ncu; ++i) cyc.push_back((double)(st[2 * i + 1] - st[2 * i]));
    std::sort(cyc.begin(), cyc.end());
    const int loaders = (V == 3 || V == 6 || V == 7) ? nw / 2 : nw;       // (V6: the stamps are a READER wave's; its share = the other half of the waves)
    const double bytes = (double)T * loaders * PER * 1024;
    printf("%-34s waves %d per %2d mf %3d: %6.1f B/cycle/CU (median WG, %7.0f cycles / round), %7.1f GB/s chip, %.3f ms\n", name, nw, PER, mf,
           bytes / cyc[ncu / 2], cyc[ncu / 2] / T, bytes * ncu / (ms * 1e-3) / 1e9, ms);
}

int main(int argc, char **argv) {
    int ncu = 256;
    const int S = 96 * 1024, T = 400;
    char *src;
    long long *d_st;
    float *d_sink;
    CK(hipMalloc(&src, (size_t)ncu * S + 4096));
    CK(hipMemset(src, 1, (size_t)ncu * S + 4096));
    CK(hipMalloc(&d_st, ncu * 16));
    CK(hipMalloc(&d_sink, 4096));
    for (int nw : {4, 8}) {
        run<0, 2>("V0 lds-dma", src, S, nw, T, 0, d_st, d_sink, ncu);
        run<0, 4>("V0 lds-dma", src, S, nw, T, 0, d_st, d_sink, ncu);
        run<0, 8>("V0 lds-dma", src, S, nw, T, 0, d_st, d_sink, ncu);
        run<1, 4>("V1 load -> vgpr", src, S, nw, T, 0, d_st, d_sink, ncu);
        run<1, 8>("V1 load -> vgpr", src, S, nw, T, 0, d_st, d_sink, ncu);
        run<2, 4>("V2 load -> vgpr -> ds_write", src, S, nw, T, 0, d_st, d_sink, ncu);
        run<2, 8>("V2 load -> vgpr -> ds_write", src, S, nw, T, 0, d_st, d_sink, ncu);
        run<3, 4>("V3 lds-dma | half the waves ds_read", src, S, nw, T, 0, d_st, d_sink, ncu);
        run<3, 8>("V3 lds-dma | half the waves ds_read", src, S, nw, T, 0, d_st, d_sink, ncu);
        run<5, 8>("V5 ds_read_b128 only", src, S, nw, T, 0, d_st, d_sink, ncu);
        run<5, 16>("V5 ds_read_b128 only", src, S, nw, T, 0, d_st, d_sink, ncu);
        run<6, 8>("V6 ds_read_b128 (counted) | half the waves lds-dma", src, S, nw, T, 0, d_st, d_sink, ncu);
        run<7, 8>("V7 ds_read_b128 (counted) | half the waves load -> vgpr -> ds_write", src, S, nw, T, 0, d_st, d_sink, ncu);
        run<4, 2>("V4 lds-dma + mfma", src, S, nw, T, 8, d_st, d_sink, ncu);
        run<4, 3>("V4 lds-dma + mfma", src, S, nw, T, 8, d_st, d_sink, ncu);
        run<4, 4>("V4 lds-dma + mfma", src, S, nw, T, 16, d_st, d_sink, ncu);
        run<4, 8>("V4 lds-dma + mfma", src, S, nw, T, 32, d_st, d_sink, ncu);
    }
    return 0;
}
