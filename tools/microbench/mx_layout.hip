// Pins the operand layout of v_mfma_scale_f32_32x32x64_f8f6f4 with fp8 (e4m3) operands on gfx950, using values that are
// exact in e4m3 and an fp64 host reference:  D[i][j] = sum_k A[i][k] * 2^(sa[i][k/32]-127) * B[k][j] * 2^(sb[j][k/32]-127).
// Hypothesis checked: lane l (r = l & 31, h = l >> 5) holds A[r][32h + e], B[32h + e][r] in byte e = 0..31 of its 8 VGPRs,
// and the scale byte selected by opsel applies to that lane's 32 k-values (block h of row/column r).
//   hipcc --offload-arch=gfx950 -O2 tools/microbench/mx_layout.hip -o tools/microbench/mx_layout && ./mx_layout
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>

typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__global__ void k(const uint8_t *A, const uint8_t *B, const uint8_t *sa, const uint8_t *sb, float *D, int opsel_variant) {
    const int l = threadIdx.x, r = l & 31, h = l >> 5;
    i32x8 a, b;
    const int *ap = reinterpret_cast<const int *>(A + r * 64 + 32 * h);      // A[r][32h .. 32h+31], row-major [32][64]
    const int *bp = reinterpret_cast<const int *>(B + r * 64 + 32 * h);      // B stored column-major: Bt[col r][k]
    for (int e = 0; e < 8; ++e) { a[e] = ap[e]; b[e] = bp[e]; }
    // scale VGPR: byte 0 = this lane's block scale; other bytes poisoned to catch a wrong opsel reading
    const int sav = sa[r * 2 + h] | 0x55aa3300, sbv = sb[r * 2 + h] | 0x33cc5500;
    f32x16 c = {0};
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, sav, 0, sbv);   // cbsz=0 (fp8), blgp=0 (fp8)
    for (int g = 0; g < 16; ++g) D[((g & 3) + 8 * (g >> 2) + 4 * h) * 32 + r] = c[g];
}

static float e4m3_to_f(uint8_t v) {
    const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
    float x = e == 0 ? ldexpf((float)m, -9) : ldexpf(1.0f + m / 8.0f, e - 7);
    return s ? -x : x;
}

static int run(bool unit_scales, int membership) {
    std::vector<uint8_t> A(32 * 64), Bt(32 * 64), sa(64), sb(64);
    srand(7);
    // values with |x| <= 1.875 so that products and sums stay small and exact
    for (auto &v : A) v = (uint8_t)((rand() & 0x80) | (0x30 + (rand() % 16)));
    for (auto &v : Bt) v = (uint8_t)((rand() & 0x80) | (0x30 + (rand() % 16)));
    for (auto &v : sa) v = unit_scales ? 127 : (uint8_t)(124 + rand() % 7);
    for (auto &v : sb) v = unit_scales ? 127 : (uint8_t)(124 + rand() % 7);
    uint8_t *dA, *dB, *dsa, *dsb; float *dD;
    hipMalloc(&dA, A.size()); hipMalloc(&dB, Bt.size()); hipMalloc(&dsa, 64); hipMalloc(&dsb, 64); hipMalloc(&dD, 32 * 32 * 4);
    hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice); hipMemcpy(dB, Bt.data(), Bt.size(), hipMemcpyHostToDevice);
    hipMemcpy(dsa, sa.data(), 64, hipMemcpyHostToDevice); hipMemcpy(dsb, sb.data(), 64, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dsa, dsb, dD, 0);
    std::vector<float> D(32 * 32);
    hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
    int bad = 0; double maxrel = 0;
    for (int i = 0; i < 32; ++i)
        for (int j = 0; j < 32; ++j) {
            double ref = 0;
            // memory byte kk of a row = lane half (kk / 32), byte e = kk % 32 of that lane
            for (int kk = 0; kk < 64; ++kk) {
                const int h = kk / 32, e = kk % 32;
                const int blk = membership == 0 ? h : e / 16;      // which scale block this byte belongs to
                ref += (double)e4m3_to_f(A[i * 64 + kk]) * ldexp(1.0, sa[i * 2 + blk] - 127) *
                       (double)e4m3_to_f(Bt[j * 64 + kk]) * ldexp(1.0, sb[j * 2 + blk] - 127);
            }
            const double err = fabs(ref - D[i * 32 + j]), rel = err / (fabs(ref) + 1e-30);
            if (rel > maxrel) maxrel = rel;
            if (err > 1e-3 * (1 + fabs(ref))) { if (bad < 3) printf("  mismatch D[%d][%d] = %g, ref %g\n", i, j, D[i * 32 + j], ref); ++bad; }
        }
    printf("mx_layout unit_scales=%d membership=%s: %d mismatches of 1024\n", unit_scales, membership == 0 ? "block=h" : "block=e/16", bad);
    return bad;
}

int main() {
    run(true, 0);
    run(false, 0);
    run(false, 1);
    return 0;
}
