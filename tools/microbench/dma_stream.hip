// Microbenchmark: per-CU rate of global_load_lds_dwordx4 streaming (no compute), as a function of ring depth,
// pieces per wave and tile, waves per workgroup, and where the bytes come from (a small buffer every workgroup
// re-reads = L2, or disjoint slices of a large buffer = HBM).   hipcc --offload-arch=gfx950 -O3 dma_stream.hip -o dma_stream
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int NW, int PPW, int NSLOT>
__global__ __launch_bounds__(NW * 64) void stream(const char *src, size_t span, size_t wg_stride, int tiles, int *sink, int rot) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TILE = NW * PPW * 1024;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const char *base = src + (size_t)blockIdx.x * wg_stride;
    auto issue = [&](int t, int slot) __attribute__((always_inline)) {
        size_t off = ((size_t)(t + (int)blockIdx.x * rot) * TILE) % span;   // rot: workgroups start at different tiles of a shared buffer
#pragma unroll
        for (int i = 0; i < PPW; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(base + off + (wave * PPW + i) * 1024 + lane * 16),
                                             (__attribute__((address_space(3))) void *)(smem + slot * TILE + (wave * PPW + i) * 1024), 16, 0, 0);
    };
#pragma unroll
    for (int i = 0; i < NSLOT - 1; ++i) issue(i, i);
    int acc = 0;
    for (int t = 0; t < tiles; ++t) {
        wait_vmcnt<(NSLOT - 2) * PPW>();
        __builtin_amdgcn_s_barrier();
        issue(t + NSLOT - 1, (t + NSLOT - 1) % NSLOT);
        acc += *reinterpret_cast<const int *>(smem + (t % NSLOT) * TILE + threadIdx.x * 4);   // touch the tile
    }
    wait_vmcnt<0>();
    if (acc == 0x12345678) sink[0] = acc;
}

// gather shape of a GEMM operand tile: each 1-KiB piece = 8 rows x 128 B, rows `rs` bytes apart (rs = K*2)
template <int NW, int PPW, int NSLOT>
__global__ __launch_bounds__(NW * 64) void stream_rows(const char *src, size_t span, int rs, int tiles, int *sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TILE = NW * PPW * 1024;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    auto issue = [&](int t, int slot) __attribute__((always_inline)) {
        // tile t = k-tile t of a [rows = NW*PPW*8][K] matrix: column offset t*128 bytes (wrapping inside a row)
        const size_t col = ((size_t)t * 128) % (size_t)rs;
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int row = (wave * PPW + i) * 8 + (lane >> 3);
            const size_t off = ((size_t)row * rs + col + (lane & 7) * 16) % span;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + off),
                                             (__attribute__((address_space(3))) void *)(smem + slot * TILE + (wave * PPW + i) * 1024), 16, 0, 0);
        }
    };
#pragma unroll
    for (int i = 0; i < NSLOT - 1; ++i) issue(i, i);
    int acc = 0;
    for (int t = 0; t < tiles; ++t) {
        wait_vmcnt<(NSLOT - 2) * PPW>();
        __builtin_amdgcn_s_barrier();
        issue(t + NSLOT - 1, (t + NSLOT - 1) % NSLOT);
        acc += *reinterpret_cast<const int *>(smem + (t % NSLOT) * TILE + threadIdx.x * 4);
    }
    wait_vmcnt<0>();
    if (acc == 0x12345678) sink[0] = acc;
}

template <int NW, int PPW, int NSLOT>
static void run_rows(const char *name, const char *buf, size_t span, int rs, int grid, int tiles, int *sink) {
    constexpr int lds = NSLOT * NW * PPW * 1024;
    auto k = stream_rows<NW, PPW, NSLOT>;
    hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(grid), dim3(NW * 64), lds, 0, buf, span, rs, tiles, sink);
    hipEventRecord(e0);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k, dim3(grid), dim3(NW * 64), lds, 0, buf, span, rs, tiles, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
    const double bytes = (double)grid * tiles * NW * PPW * 1024;
    printf("%-34s NW=%d PPW=%2d NSLOT=%d tile=%3d KB rowstride=%5d grid=%4d: %7.3f ms  %6.2f TB/s  %6.1f GB/s per CU\n", name, NW, PPW,
           NSLOT, NW * PPW, rs, grid, ms, bytes / ms / 1e9, bytes / ms / 1e6 / 256);
}

template <int NW, int PPW, int NSLOT>
static void run(const char *name, const char *buf, size_t span, size_t wg_stride, int grid, int tiles, int *sink, int rot = 0) {
    constexpr int lds = NSLOT * NW * PPW * 1024;
    auto k = stream<NW, PPW, NSLOT>;
    hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(grid), dim3(NW * 64), lds, 0, buf, span, wg_stride, tiles, sink, rot);
    hipEventRecord(e0);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k, dim3(grid), dim3(NW * 64), lds, 0, buf, span, wg_stride, tiles, sink, rot);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
    const double bytes = (double)grid * tiles * NW * PPW * 1024;
    printf("%-34s NW=%d PPW=%2d NSLOT=%d tile=%3d KB lds=%3d KB grid=%4d: %7.3f ms  %6.2f TB/s  %6.1f GB/s per CU\n", name, NW, PPW, NSLOT,
           NW * PPW, lds >> 10, grid, ms, bytes / ms / 1e9, bytes / ms / 1e6 / 256);
}

int main() {
    const size_t big = (size_t)3 << 30;
    char *buf; int *sink;
    hipMalloc(&buf, big); hipMalloc(&sink, 64);
    hipMemset(buf, 1, big);
    const size_t MB = 1 << 20;
    // L2-resident source shared by all workgroups (like W of a GEMM): 1 MB span
    run<8, 3, 3>("L2 shared 1MB", buf, 1 * MB, 0, 256, 4000, sink);
    run<8, 6, 2>("L2 shared 1MB", buf, 1 * MB, 0, 256, 2000, sink);
    run<8, 8, 2>("L2 shared 1MB", buf, 1 * MB, 0, 256, 2000, sink);
    run<8, 4, 4>("L2 shared 1MB", buf, 1 * MB, 0, 256, 4000, sink);
    run<8, 2, 8>("L2 shared 1MB", buf, 1 * MB, 0, 256, 8000, sink);
    run<8, 1, 16>("L2 shared 1MB", buf, 1 * MB, 0, 256, 16000, sink);
    run<4, 6, 3>("L2 shared 1MB", buf, 1 * MB, 0, 256, 4000, sink);
    run<4, 6, 3>("L2 shared 1MB, 2 WG/CU", buf, 1 * MB, 0, 512, 4000, sink);
    run<4, 4, 3>("L2 shared 1MB, 3 WG/CU", buf, 1 * MB, 0, 768, 4000, sink);
    run<16, 2, 3>("L2 shared 1MB", buf, 1 * MB, 0, 256, 4000, sink);
    // same tile stream read by all workgroups at the same time (like K1 at large Q): 768 MB span, stride 0
    run<8, 3, 3>("all WGs same 768MB stream", buf, 768 * MB, 0, 256, 4000, sink);
    // HBM: every workgroup its own 8 MB slice
    run<8, 3, 3>("HBM disjoint 8MB/WG", buf, 8 * MB, 8 * MB, 256, 4000, sink);
    run<8, 6, 2>("HBM disjoint 8MB/WG", buf, 8 * MB, 8 * MB, 256, 2000, sink);
    run<8, 4, 4>("HBM disjoint 8MB/WG", buf, 8 * MB, 8 * MB, 256, 3000, sink);
    run<8, 2, 8>("HBM disjoint 8MB/WG", buf, 8 * MB, 8 * MB, 256, 6000, sink);
    run<4, 6, 3>("HBM disjoint, 2 WG/CU", buf, 4 * MB, 4 * MB, 512, 2000, sink);
    // GEMM-shaped gathers from an L2-resident matrix: 8 rows x 128 B per piece
    run_rows<8, 6, 2>("W tile 384x64 bf16, K=1536", buf, 2 * MB, 3072, 256, 2000, sink);
    run_rows<8, 6, 2>("W tile 384x64 bf16, K=384", buf, 2 * MB, 768, 256, 2000, sink);
    run_rows<8, 3, 3>("W tile 192x64 bf16, K=384", buf, 2 * MB, 768, 256, 4000, sink);
    run_rows<8, 6, 2>("rows 128 B apart (contiguous)", buf, 2 * MB, 128, 256, 2000, sink);
    run_rows<8, 6, 2>("W tile, K=1536, HBM-size span", buf, 1024 * MB, 3072, 256, 2000, sink);
    // do workgroups that stream the SAME bytes at the SAME time (every GEMM's W operand here) get in each other's way?
    // rot = 0: lockstep; rot = 5 / 17: workgroup b starts b*rot tiles further into the shared buffer
    for (int rot : {0, 1, 5, 17}) {
        char nm[64];
        snprintf(nm, sizeof nm, "L2 shared 1.2MB rot=%d", rot);
        run<8, 3, 3>(nm, buf, 50 * 24576, 0, 256, 4000, sink, rot);
        run<8, 3, 5>(nm, buf, 50 * 24576, 0, 256, 4000, sink, rot);
        run<8, 6, 3>(nm, buf, 25 * 49152, 0, 256, 2000, sink, rot);
    }
    hipDeviceSynchronize();
    return 0;
}
