// Micro-benchmark of the tile loop shared by cos_topk_partial, gemm_xres2 and ln_rows_gemm (gfx950): per step a workgroup
// optionally issues LDS-DMA pieces, passes a barrier, and every wave runs NM MFMAs, each fed by one ds_read_b128 of an
// XOR-swizzled 24-KiB LDS tile (the other operand is resident in registers).  Reports shader cycles per step (s_memtime,
// median over workgroups) against the 32 * NM * waves-per-SIMD cycles the matrix pipe needs.
//   hipcc --offload-arch=gfx950 -O3 -o tools/microbench/mfma_loop tools/microbench/mfma_loop.hip && tools/microbench/mfma_loop
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <type_traits>
#include <utility>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

template <int... I, class F>
__device__ __forceinline__ void static_for(std::integer_sequence<int, I...>, F &&f) { (f(std::integral_constant<int, I>{}), ...); }
template <int OFF>
__device__ __forceinline__ void lds_rd(u32x4 &dst, uint32_t addr) { asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory"); }
template <int N>
__device__ __forceinline__ void lgkm(u32x4 &r) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(r) : "n"(N) : "memory"); }
template <int N>
__device__ __forceinline__ void vmwait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void glds16(const void *g, void *l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g, (__attribute__((address_space(3))) void *)l, 16, 0, 0);
}

// NW waves; READS: fragment reads from LDS (else the MFMA operand stays in a register); BAR: barrier per SPB steps; DMA: pieces per
// wave and step (0: none); M16: 16x16x32 form (2 MFMAs per read); LATE: waves >= NW/2 issue their DMA after their MFMAs
// ST: 0 none; 1: two 16-byte stores per wave and step in gemm_xres2's pattern (32 token rows x 32 B, rows 2 304 B apart);
// 2: the same bytes as two contiguous 1-KiB stores; 3: pattern 1 issued inside the MFMA stream (behind MFMAs 14 and 20)
template <int NW, bool READS, int SPB, int DMA, bool M16, bool LATE, int ST = 0>
__global__ __launch_bounds__(NW * 64) void loop_kernel(const char *__restrict__ src, int steps, unsigned long long *__restrict__ cyc, float *__restrict__ sink,
                                                       char *__restrict__ dst = nullptr) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TILE = 24576, NSLOT = 4, NM = 24;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, h = lane >> 5;
    // fill LDS with pseudo-random bf16 bits (finite)
    for (int i = threadIdx.x; i < NSLOT * TILE / 4; i += NW * 64) {
        uint32_t v = (uint32_t)(i * 2654435761u) ^ (blockIdx.x * 40503u);
        reinterpret_cast<uint32_t *>(smem)[i] = (v & 0x3f7f3f7fu) | 0x3c003c00u;
    }
    __syncthreads();
    bf16x8 b;
#pragma unroll
    for (int j = 0; j < 8; ++j) b[j] = (__bf16)(0.01f * (float)((lane * 7 + j * 13) % 17 - 8));
    f32x16 acc[3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int g = 0; g < 16; ++g) acc[i][g] = 0.f;
    const uint32_t lbase = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) char *)smem);
    uint32_t ck[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) ck[ks] = lbase + r * 256 + (((2 * ks + h) ^ (r & 15)) << 4);
    const char *my = src + ((size_t)(blockIdx.x % 64) * TILE) + lane * 16;
    const bool late = LATE && wave >= NW / 2;
    auto issue = [&](int st) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < DMA; ++i)
            glds16(my + ((wave * (DMA > 0 ? DMA : 1) + i) % 24) * 1024, smem + ((st + 2) % NSLOT) * TILE + ((wave * (DMA > 0 ? DMA : 1) + i) % 24) * 1024);
    };
    char *drow = dst + (size_t)blockIdx.x * (256 * 2304) + (ST == 2 ? (size_t)wave * 32 * 2304 + lane * 16 : (size_t)(wave * 32 + r) * 2304 + 16 * h);
    auto do_store = [&](int st, int which) __attribute__((always_inline)) {
        const u32x4 v = {(uint32_t)st, (uint32_t)lane, 0x3c003c00u, (uint32_t)which};
        char *a = drow + ((st * 2 + which) % 36) * (ST == 2 ? 1024 : 64);
        if (ST == 2) a = drow + ((st * 2 + which) % 64) * 1024;
        *reinterpret_cast<u32x4 *>(a) = v;
    };
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int st = 0; st < steps; ++st) {
        if (st % SPB == 0) {
            if constexpr (DMA > 0) vmwait<DMA>();
            if constexpr (SPB < 1000) __builtin_amdgcn_s_barrier();
        }
        if constexpr (DMA > 0) { if (!late) issue(st); }
        const uint32_t so = (st % NSLOT) * TILE;
        constexpr int PF = 4;
        u32x4 fr[PF + 1];
        auto rd = [&](auto nc) __attribute__((always_inline)) {
            constexpr int n = decltype(nc)::value;
            if constexpr (READS) lds_rd<(n % 3) * 8192>(fr[n % (PF + 1)], ck[n / 3] + so);
        };
        if constexpr (!READS) {
#pragma unroll
            for (int i = 0; i <= PF; ++i) fr[i] = u32x4{0x3c003c01u + i, 0x3c103c00u, 0x3c003c20u, 0x3c303c00u + lane};
        }
        static_for(std::make_integer_sequence<int, PF>{}, rd);
        static_for(std::make_integer_sequence<int, NM>{}, [&](auto nc) __attribute__((always_inline)) {
            constexpr int n = decltype(nc)::value;
            if constexpr (n + PF < NM) rd(std::integral_constant<int, n + PF>{});
            if constexpr (READS) { constexpr int y = n + PF < NM ? PF : NM - 1 - n; lgkm<y>(fr[n % (PF + 1)]); }
            if constexpr (M16) {
                // two 16x16x32 per fragment: same pipe time as one 32x32x16
                f32x4 *a4 = reinterpret_cast<f32x4 *>(&acc[n % 3]);
                typedef __attribute__((ext_vector_type(8))) __bf16 v8;
                a4[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8, fr[n % (PF + 1)]), b, a4[0], 0, 0, 0);
                a4[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8, fr[n % (PF + 1)]), b, a4[1], 0, 0, 0);
            } else {
                acc[n % 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fr[n % (PF + 1)]), b, acc[n % 3], 0, 0, 0);
            }
            if constexpr (ST == 3 && n == 14) do_store(st, 0);
            if constexpr (ST == 3 && n == 20) do_store(st, 1);
        });
        if constexpr (ST == 1 || ST == 2) {
#pragma unroll
            for (int i = 0; i < 3; ++i) asm volatile("" : "+v"(acc[i]));
            do_store(st, 0);
            do_store(st, 1);
        }
        if constexpr (DMA > 0) {
            if (late) {
#pragma unroll
                for (int i = 0; i < 3; ++i) asm volatile("" : "+v"(acc[i]));
                issue(st);
            }
        }
    }
    vmwait<0>();
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int g = 0; g < 16; ++g) s += acc[i][g];
    if (s == 12345.678f) sink[threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NW, bool READS, int SPB, int DMA, bool M16, bool LATE, int ST = 0>
static void run(const char *name, const char *src, unsigned long long *cyc, float *sink, int steps, char *dst = nullptr) {
    auto k = loop_kernel<NW, READS, SPB, DMA, M16, LATE, ST>;
    const int lds = 4 * 24576;
    hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(256), dim3(NW * 64), lds, 0, src, steps, cyc, sink, dst);
        hipEventRecord(e1);
        hipDeviceSynchronize();
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(256);
    hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double per = (double)h[128] / steps, pipe = 32.0 * 24 * (NW / 4);
    printf("%-58s %8.0f cycles/step  pipe %5.0f  busy %5.1f %%   %.3f ms  clock %.2f GHz\n", name, per, pipe, 100.0 * pipe / per, ms, h[128] / (ms * 1e6));
}

int main() {
    char *src; unsigned long long *cyc; float *sink;
    hipMalloc(&src, 64 * 24576 + 4096); hipMemset(src, 0x3c, 64 * 24576 + 4096);
    hipMalloc(&cyc, 256 * 8); hipMalloc(&sink, 4096);
    const int S = 4000;
    run<8, false, 100000, 0, false, false>("8 waves, MFMA only (operand in registers)", src, cyc, sink, S);
    run<8, true, 100000, 0, false, false>("8 waves, + ds_read_b128 per MFMA", src, cyc, sink, S);
    run<8, true, 1, 0, false, false>("8 waves, + reads + barrier per step", src, cyc, sink, S);
    run<8, true, 2, 0, false, false>("8 waves, + reads + barrier per 2 steps", src, cyc, sink, S);
    run<8, true, 1, 3, false, false>("8 waves, reads + barrier + 3 DMA pieces/wave/step", src, cyc, sink, S);
    run<8, true, 1, 3, false, true>("8 waves, same, waves 4-7 issue after their MFMAs", src, cyc, sink, S);
    run<8, true, 1, 5, false, false>("8 waves, reads + barrier + 5 DMA pieces/wave/step", src, cyc, sink, S);
    run<8, true, 1, 8, false, false>("8 waves, reads + barrier + 8 DMA pieces/wave/step", src, cyc, sink, S);
    run<8, true, 1, 3, true, false>("8 waves, 16x16x32 form, reads + barrier + 3 DMA", src, cyc, sink, S);
    run<8, true, 1, 0, true, false>("8 waves, 16x16x32 form, reads + barrier", src, cyc, sink, S);
    char *dst;
    hipMalloc(&dst, (size_t)256 * 256 * 2304 + (1 << 20));
    run<8, true, 1, 3, false, false, 1>("8 waves, reads + barrier + 3 DMA + 2 stores (32 rows x 32 B)", src, cyc, sink, S, dst);
    run<8, true, 1, 3, false, false, 2>("8 waves, reads + barrier + 3 DMA + 2 stores (contiguous 1 KiB)", src, cyc, sink, S, dst);
    run<8, true, 1, 3, false, false, 3>("8 waves, reads + barrier + 3 DMA + 2 stores (32 x 32 B) in stream", src, cyc, sink, S, dst);
    run<4, false, 100000, 0, false, false>("4 waves, MFMA only", src, cyc, sink, S);
    run<4, true, 100000, 0, false, false>("4 waves, + ds_read_b128 per MFMA", src, cyc, sink, S);
    run<4, true, 1, 6, false, false>("4 waves, reads + barrier + 6 DMA pieces/wave/step", src, cyc, sink, S);
    return 0;
}
