// Probe of v_mfma_scale_f32_32x32x64_f8f6f4 on gfx950: operand K layout and E8M0 scale semantics (tools/probe, not product code).
// build: hipcc --offload-arch=gfx950 -O2 tools/probe/mx_probe.hip -o tools/probe/mx_probe.bin ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void k(const unsigned char *A, const unsigned char *B, float *D, int hyp, int sa, int sb) {
    // A: [32][64] row-major fp8 (i, k); B: [64][32] (k, j) stored as Bt[j][k]; lane l supplies 32 bytes
    const int l = threadIdx.x, r = l & 31, h = l >> 5;
    unsigned char a[32], b[32];
    for (int t = 0; t < 32; ++t) {
        int kk;
        if (hyp == 0) kk = 32 * h + t;                          // contiguous 32 per lane half
        else kk = 16 * (2 * (t >> 4) + h) + (t & 15);           // two 16-byte groups interleaved across the halves
        a[t] = A[r * 64 + kk];
        b[t] = B[r * 64 + kk];
    }
    i32x8 av, bv;
    memcpy(&av, a, 32); memcpy(&bv, b, 32);
    f32x16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, acc, 0, 0, 0, sa, 0, sb);
    for (int i = 0; i < 16; ++i) {
        const int row = (i & 3) + 8 * (i >> 2) + 4 * h, col = r;
        D[row * 32 + col] = acc[i];
    }
}

static unsigned char enc(int v) {            // OCP e4m3: small integers
    static const unsigned char t[4] = {0x00, 0x38, 0x40, 0x44};
    return v < 0 ? (unsigned char)(t[-v] | 0x80) : t[v];
}

int main() {
    int Ai[32][64], Bi[64][32];
    unsigned char hA[32 * 64], hB[32 * 64];
    srand(7);
    for (int i = 0; i < 32; ++i) for (int k2 = 0; k2 < 64; ++k2) { Ai[i][k2] = rand() % 7 - 3; hA[i * 64 + k2] = enc(Ai[i][k2]); }
    for (int k2 = 0; k2 < 64; ++k2) for (int j = 0; j < 32; ++j) { Bi[k2][j] = rand() % 7 - 3; hB[j * 64 + k2] = enc(Bi[k2][j]); }
    float ref[32 * 32];
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { int s = 0; for (int k2 = 0; k2 < 64; ++k2) s += Ai[i][k2] * Bi[k2][j]; ref[i * 32 + j] = (float)s; }
    unsigned char *dA, *dB; float *dD; float hD[32 * 32];
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, sizeof hD);
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    for (int hyp = 0; hyp < 2; ++hyp) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD, hyp, 0x7F7F7F7F, 0x7F7F7F7F);
        hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
        int bad = 0; float maxd = 0;
        for (int i = 0; i < 1024; ++i) { float d = hD[i] - ref[i]; if (d != 0) ++bad; if (fabsf(d) > maxd) maxd = fabsf(d); }
        printf("K-layout hypothesis %d (same mapping for A and B), scale 1.0: %d / 1024 wrong, max |diff| %g\n", hyp, bad, maxd);
    }
    const int scales[4][2] = {{0x7F, 0x7F}, {0x80, 0x7F}, {0x7F, 0x7D}, {0x73, 0x8B}};
    for (int s = 0; s < 4; ++s) {
        const int sa = scales[s][0] * 0x01010101, sb = scales[s][1] * 0x01010101;
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD, 0, sa, sb);
        hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
        double num = 0, den = 0;
        for (int i = 0; i < 1024; ++i) { num += (double)hD[i] * ref[i]; den += (double)ref[i] * ref[i]; }
        printf("scale bytes a=0x%02X b=0x%02X: D / ref = %g (expected %g)\n", scales[s][0], scales[s][1], num / den,
               ldexp(1.0, scales[s][0] - 127 + scales[s][1] - 127));
    }
    // asymmetric probe: does any A/B k-permutation matter?  feed A with hypothesis 0 and B with hypothesis 1 -> must be wrong if the
    // hardware pairs byte t of A with byte t of B
    return 0;
}
