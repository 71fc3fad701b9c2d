// What the correlation volume's store pattern costs on its own (tools/probe, not product code).
// corr_volume_kernel (prisma_amd/csrc/volume.hip) writes 27.7 GB per step at 2.2 TB/s while its matrix work needs a quarter of that
// time; tools/probe/store_probe.hip writes fp16 rows at 5.3 TB/s.  The difference between the two patterns: there a workgroup covers 512
// contiguous bytes of a row at once, here 128 B (one 64-column tile) and the next 128 B of the same row a tile later.  This probe issues
// ONLY the stores (no loads, no MFMAs), same grid and workgroup -> (row tile, column group) mapping as volume.hip, by geometry:
//   P0  workgroup 128 rows x 64 columns per tile  (4 waves x 32 rows; lane = 2 columns, dword stores: 128 B of a row per half wave)   <- volume.hip
//   P1  workgroup 128 rows x 128 columns per tile (lane = 4 columns, dwordx2: 256 B per half wave)
//   P2  workgroup 128 rows x 256 columns per tile (lane = 8 columns, dwordx4: 512 B per half wave)
//   P3  workgroup 32 rows x 256 columns per tile  (4 waves side by side, 64 columns each, dword stores)
//   P4  workgroup 64 rows x 128 columns per tile  (2 x 2 waves, 64 columns each, dword stores)
// One tile's stores per wave in flight at most (s_waitcnt vmcnt(16) before the next tile's), as in the kernel.
// build: hipcc --offload-arch=gfx950 -O3 tools/probe/volume_store_probe.hip -o tools/probe/volume_store_probe.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, (int)bytes, 0x00020000);
}

// WR = rows per workgroup, WC = columns per workgroup tile, LC = columns per lane (2, 4, 8)
template <int WR, int WC, int LC>
__global__ __launch_bounds__(256, 2) void probe(uint16_t *out, int M, int N, int64_t ldo, int tiles_per_wg, int delay) {
    extern __shared__ char smem[];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, li = lane & 31, lh = lane >> 5;
    const int tilesM = (M + WR - 1) / WR, tilesN = (N + WC - 1) / WC;
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int qd = nwg >> 3, rm = nwg & 7, xcd = bid & 7;
    const int swz = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (bid >> 3);
    const int grp = swz / tilesM, tile_m = swz - grp * tilesM;
    constexpr int WAVES_M = WR / 32, WAVES_N = 4 / WAVES_M;           // wave grid inside the workgroup
    constexpr int CW = WC / WAVES_N;                                   // columns per wave = 32 lanes x LC
    static_assert(CW == 32 * LC, "wave tile");
    const int wm = wave % WAVES_M, wn = wave / WAVES_M;
    const int m0 = tile_m * WR + wm * 32;
    const int nt0 = grp * tiles_per_wg;
    const int ntl = (tilesN - nt0) < tiles_per_wg ? (tilesN - nt0) : tiles_per_wg;
    if (ntl <= 0) return;
    const int rows_w = (M - m0) < 32 ? (M - m0 < 0 ? 0 : M - m0) : 32;
    const int ldb = (int)ldo * 2;
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(out + (int64_t)m0 * ldo, (unsigned)(rows_w * ldb));
    const int vo_lane = 4 * lh * ldb + li * LC * 2 + wn * CW * 2;
    const unsigned v = 0x3c003c00u + lane;
    for (int j = 0; j < ntl; ++j) {
        const int n0 = (nt0 + j) * WC;
        if (n0 + wn * CW + li * LC < N) {
            const int vo = vo_lane + n0 * 2;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int so = ((r & 3) + 8 * (r >> 2)) * ldb;
                if constexpr (LC == 2) __builtin_amdgcn_raw_buffer_store_b32(v, rs, vo, so, 0);
                else if constexpr (LC == 4) __builtin_amdgcn_raw_buffer_store_b64(u32x2{v, v}, rs, vo, so, 0);
                else __builtin_amdgcn_raw_buffer_store_b128(u32x4{v, v, v, v}, rs, vo, so, 0);
            }
        }
        for (int d = 0; d < delay; ++d) __builtin_amdgcn_s_sleep(8);
        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    }
}

template <int WR, int WC, int LC>
void run(const char *name, uint16_t *out, int M, int N, int64_t ldo, int delay) {
    auto k = probe<WR, WC, LC>;
    CK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    const int tilesM = (M + WR - 1) / WR, tilesN = (N + WC - 1) / WC;
    int groups = (2048 + tilesM - 1) / tilesM;
    groups = groups < 1 ? 1 : (groups > tilesN ? tilesN : groups);
    const int tpw = (tilesN + groups - 1) / groups;
    groups = (tilesN + tpw - 1) / tpw;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k, dim3(tilesM * groups), dim3(256), 65536, 0, out, M, N, ldo, tpw, delay);
    CK(hipEventRecord(e0));
    const int reps = 4;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k, dim3(tilesM * groups), dim3(256), 65536, 0, out, M, N, ldo, tpw, delay);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    printf("%-58s grid %5d x %3d tiles/wg, %2d x s_sleep 8: %7.3f ms per volume, %5.2f TB/s\n", name, tilesM * groups, tpw, delay, ms, (double)M * N * 2 / (ms * 1e-3) / 1e12);
}

int main() {
    const int M = 18360, N = 18360;
    const int64_t ldo = 19200;
    uint16_t *out;
    CK(hipMalloc(&out, (size_t)M * ldo * 2));
    CK(hipMemset(out, 0, (size_t)M * ldo * 2));
    for (int delay : {0, 4}) {
        run<128, 64, 2>("P0 wg 128 rows x 64 cols, dword (volume.hip)", out, M, N, ldo, delay);
        run<128, 128, 4>("P1 wg 128 rows x 128 cols, dwordx2", out, M, N, ldo, delay);
        run<128, 256, 8>("P2 wg 128 rows x 256 cols, dwordx4", out, M, N, ldo, delay);
        run<32, 256, 2>("P3 wg 32 rows x 256 cols (4 waves side by side), dword", out, M, N, ldo, delay);
        run<64, 128, 2>("P4 wg 64 rows x 128 cols (2 x 2 waves), dword", out, M, N, ldo, delay);
    }
    // reference: the same bytes as one flat fill
    {
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipMemsetAsync(out, 1, (size_t)M * ldo * 2));
        CK(hipEventRecord(e0));
        for (int i = 0; i < 4; ++i) CK(hipMemsetAsync(out, 1, (size_t)M * ldo * 2));
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("hipMemset of the padded volume: %7.3f ms, %5.2f TB/s\n", ms / 4, (double)M * ldo * 2 / (ms / 4 * 1e-3) / 1e12);
    }
    return 0;
}
