// Internal helpers shared by the HIP translation units of libtsim.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

#include "../../include/tsim.h"

namespace tsim {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef uint16_t bf16_t;  // storage type of a bfloat16 element
// Unit rows for the search are IEEE half: 11 significand bits against bf16's 8 make the MFMA selection scores 8x closer
// to the exact cosine (the elements of a unit row are <= 1 in magnitude, so the narrower exponent costs nothing; the f16
// MFMA runs at the bf16 rate), which is what keeps the exactness guard of the search (search.hip) quiet.
typedef _Float16 unit_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;

constexpr int WAVE = 64;

void set_error(const std::string &msg);
int fail(int code, const char *fmt, ...);

#define TSIM_HIP_CHECK(expr)                                                                      \
    do {                                                                                          \
        hipError_t _e = (expr);                                                                   \
        if (_e != hipSuccess)                                                                     \
            return ::tsim::fail(TSIM_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                                __FILE__, __LINE__);                                              \
    } while (0)

#define TSIM_REQUIRE(cond, ...)                                   \
    do {                                                          \
        if (!(cond)) return ::tsim::fail(TSIM_EINVAL, __VA_ARGS__); \
    } while (0)

// ---------------------------------------------------------------- device helpers
__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

// round-to-nearest-even float -> bf16 bits; the plain cast lowers to v_cvt_pk_bf16_f32 on gfx950 and
// keeps NaN a NaN (MI355X_MICROARCH.md "Correctness boundaries").
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(bf16_t, b);
}

__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    return (uint32_t)f32_to_bf16(lo) | ((uint32_t)f32_to_bf16(hi) << 16);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Canonical row scale for unit rows, reproducible bit for bit on the host (oracle/search_ref.l2_normalize):
// sum of squares in float64 (each lane's partial in element order j = lane, lane+64, ..., then an xor butterfly
// 32,16,..,1 — every lane ends with the same bits), inv = 1 / max(sqrt(ss), eps) in float64; an element is then
// double(x) * inv rounded ONCE to bf16 (f64_to_bf16).
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double canonical_inv_norm(double lane_partial_ss, float eps) {
    const double ss = wave_sum_f64(lane_partial_ss);
    return 1.0 / fmax(sqrt(ss), (double)eps);
}
// float64 -> bf16 with ONE correct rounding (nearest, ties to even).  Going through float32 would round twice: a
// value within half a float ulp of a bf16 midpoint lands ON the midpoint and the tie rule then decides.  The
// midpoint case is detected on the float bits and undone by stepping one float ulp back towards the true value.
__device__ __forceinline__ bf16_t f64_to_bf16(double v) {
    const float f = (float)v;
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)0x7fc0;  // NaN
    if ((u & 0xffffu) == 0x8000u) {
        const double fd = fabs((double)f), av = fabs(v);
        if (av < fd) u -= 1u;
        else if (av > fd) u += 1u;
    }
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}
// float64 -> IEEE half with ONE correct rounding: float32 by round-to-odd (truncate towards zero, set the last bit when
// anything was lost), then the hardware's nearest-even float32 -> half conversion; with 13+ spare bits the sticky last bit
// makes the second rounding see exactly which side of every half-precision midpoint the true value lies on (also in the
// subnormal half range).  numpy's float64 -> float16 cast (oracle/search_ref.unit_rows) rounds once as well.
__device__ __forceinline__ unit_t f64_to_f16(double v) {
    float f = (float)v;
    if ((double)f != v && f == f) {   // inexact (and not NaN)
        uint32_t u = __float_as_uint(f);
        if (fabs((double)f) > fabs(v)) u -= 1u;   // back to the truncation towards zero (magnitude bits only change)
        u |= 1u;
        f = __uint_as_float(u);
    }
    return (unit_t)f;
}
__device__ __forceinline__ unit_t canonical_unit_elem(float x, double inv) { return f64_to_f16((double)x * inv); }

// ---- the search's exactness guard (search.hip; restated in oracle/search_ref.guard_eps) ------------------------------------
// A stored half row is u^ = u + delta, u = x / max(|x|, eps) the exact unit row, rho = |delta|_2 its rounding residual.
// For two rows, by Cauchy-Schwarz (|u| <= 1):   | u^q . u^c  -  u_q . u_c |  <=  rho_q + rho_c + rho_q rho_c,
// and u_q . u_c IS the reference's F.cosine_similarity of the float32 rows.  The MFMA score differs from u^q . u^c by the
// float32 accumulation of ld products that are exact in float32 (11 x 11 significand bits): at most
// ld * 2^-23 * |u^q| |u^c| for ANY summation order with rounding or truncation to float32 at every step.  One more 2^-22
// covers the final rounding of the exact score to float32 and a tie on it.  Every constant is rounded UP.
//   rho_round_up: float upper bound of a float64 residual norm; NaN / oversized values (rows with NaN or inf elements) are
//                 clamped to 2, which sends every query that meets them to the brute-force pass.
__device__ __forceinline__ float rho_round_up(double r) {
    const float f = (float)(r * (1.0 + 1e-6));
    return f < 2.f ? f : 2.f;
}
// raise *word to rho (non-negative floats order like their bit patterns).  The word is read first: once it holds a value near
// the maximum almost no row beats it, so a million rows cost a handful of atomics instead of a million serialised ones
// (measured: 11 ms -> sub-ms for 1 M rows).
__device__ __forceinline__ void rho_publish(float *word, float rho) {
    const int bits = __float_as_int(rho);
    if (bits > __hip_atomic_load(reinterpret_cast<int *>(word), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
        atomicMax(reinterpret_cast<int *>(word), bits);
}
// a-priori residual bound of a canonical unit row (every element rounded to nearest half): relative 2^-11 per normal element,
// absolute 2^-25 per subnormal one  ->  rho <= 2^-11 |u| + sqrt(ld) 2^-25.  Used when the caller has no measured rho_max.
__host__ __device__ __forceinline__ float rho_apriori(int ld) {
    return 4.8828125e-4f * 1.000001f + sqrtf((float)ld) * 2.98023224e-8f * 1.000001f;
}
__device__ __forceinline__ float guard_eps(float rho_q, float rho_c, int ld) {
    const double acc = (double)ld * 1.1920928955078125e-7 * (1.0 + rho_q) * (1.0 + rho_c);
    const double e = (double)rho_q + (double)rho_c + (double)rho_q * (double)rho_c + acc + 2.384185791015625e-7;
    return (float)(e * (1.0 + 1e-6));
}

__device__ __forceinline__ float gelu_erf(float x) {
    // 0.5 x (1 + erf(x / sqrt 2)); erf by Abramowitz-Stegun 7.1.26 (|err| < 1.5e-7, far below bf16 output resolution),
    // raw v_rcp_f32 / v_exp_f32 (1 ulp) with the constants folded: 14 VALU instructions (erff(): ~30; the same formula
    // with an IEEE-correct reciprocal and a guarded exp: 29).  FFN1's epilogue is VALU-bound, so this matters.
    //   1 - erf(|x|/sqrt2) = poly(t) * t * exp(-x^2/2),  t = 1 / (1 + p |x| / sqrt2)
    //   gelu = hx + |hx| (1 - pe) with hx = x/2
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f * 0.70710678118654752f, fabsf(x), 1.0f));
    float poly = fmaf(1.061405429f, t, -1.453152027f);
    poly = fmaf(poly, t, 1.421413741f);
    poly = fmaf(poly, t, -0.284496736f);
    poly = fmaf(poly, t, 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(x * x * (-0.5f * 1.4426950408889634f));
    const float pe = poly * t * e;
    const float hx = 0.5f * x;
    return fmaf(-fabsf(hx), pe, hx + fabsf(hx));
}

// GELU of two values without transcendentals, in packed fp32 (v_pk_fma_f32 / v_pk_mul_f32: two values per issue slot).
//   gelu(x) = x * Phi(x),  Phi(x) ~ 1/2 + xc * S(xc^2),  xc = clamp(x, -4, 4),  S = degree-8 Chebyshev fit of
//   (Phi(sqrt u) - 1/2) / sqrt u on u in [0, 16]  (fp32 Horner: relative error <= 2.5e-5 for x > 0, absolute error
//   <= 4e-5 on [-4, 0] — two orders below the bf16 resolution of the stored result; beyond |x| = 4 Phi is frozen at
//   Phi(+-4) = 1 - 3.2e-5 / 3.2e-5).  26 issue cycles per value against 64 for gelu_erf (rcp + exp + 12 VALU): the FFN1
//   epilogue is VALU-bound.  Set TSIM_GELU_ERF at build time (-DTSIM_GELU_ERF) to fall back to the A&S erf form.
typedef __attribute__((ext_vector_type(2))) float f32x2;
__device__ __forceinline__ void gelu2(float &a, float &b) {
#ifdef TSIM_GELU_ERF
    a = gelu_erf(a);
    b = gelu_erf(b);
#else
    const f32x2 x = {a, b};
    const f32x2 xc = {__builtin_amdgcn_fmed3f(a, -4.0f, 4.0f), __builtin_amdgcn_fmed3f(b, -4.0f, 4.0f)};
    const f32x2 u = xc * xc;
    f32x2 p = {9.56756410e-11f, 9.56756410e-11f};
    p = __builtin_elementwise_fma(p, u, (f32x2){-8.02642397e-09f, -8.02642397e-09f});
    p = __builtin_elementwise_fma(p, u, (f32x2){3.00262883e-07f, 3.00262883e-07f});
    p = __builtin_elementwise_fma(p, u, (f32x2){-6.72069427e-06f, -6.72069427e-06f});
    p = __builtin_elementwise_fma(p, u, (f32x2){1.02510894e-04f, 1.02510894e-04f});
    p = __builtin_elementwise_fma(p, u, (f32x2){-1.15122017e-03f, -1.15122017e-03f});
    p = __builtin_elementwise_fma(p, u, (f32x2){9.92152281e-03f, 9.92152281e-03f});
    p = __builtin_elementwise_fma(p, u, (f32x2){-6.64609522e-02f, -6.64609522e-02f});
    p = __builtin_elementwise_fma(p, u, (f32x2){3.98939520e-01f, 3.98939520e-01f});
    const f32x2 phi = __builtin_elementwise_fma(xc, p, (f32x2){0.5f, 0.5f});
    const f32x2 y = x * phi;
    a = y[0];
    b = y[1];
#endif
}

// The same polynomial for N values at once as N INDEPENDENT scalar Horner chains, step by step: no instruction waits for
// the one before it.  gelu2's packed chain is the cheaper form when issue slots are the limit (a VALU-bound epilogue:
// 26 issue cycles per value), but a packed fp32 op beside MFMAs costs far more than its slot and every step waits for the
// previous one (measured in ffn_fused_kernel: ~240 cycles per gelu2 call); where the values sit in the shadow of an MFMA
// stream this form is the right one.  Bit-identical to gelu2 (same operations per value, fp32 fma).
template <int N>
__device__ __forceinline__ void gelu_n(float (&x)[N]) {
#ifdef TSIM_GELU_ERF
#pragma unroll
    for (int i = 0; i < N; ++i) x[i] = gelu_erf(x[i]);
#else
    float xc[N], u[N], p[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        xc[i] = __builtin_amdgcn_fmed3f(x[i], -4.0f, 4.0f);
        u[i] = xc[i] * xc[i];
        p[i] = 9.56756410e-11f;
    }
    constexpr float c[8] = {-8.02642397e-09f, 3.00262883e-07f, -6.72069427e-06f, 1.02510894e-04f,
                            -1.15122017e-03f, 9.92152281e-03f, -6.64609522e-02f, 3.98939520e-01f};
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
        for (int i = 0; i < N; ++i) p[i] = fmaf(p[i], u[i], c[k]);
#pragma unroll
    for (int i = 0; i < N; ++i) x[i] = x[i] * fmaf(xc[i], p[i], 0.5f);
#endif
}

// gelu_n cut into 11 stages (0: clamp + square, 1..8: one Horner step each, 9: Phi, 10: x * Phi) so that a caller can place
// one stage of N independent operations between the MFMAs of a dependent accumulation chain.  State: xc, u, p (N each).
template <int ST, int N>
__device__ __forceinline__ void gelu_stage(float (&x)[N], float (&xc)[N], float (&u)[N], float (&p)[N]) {
    constexpr float c[8] = {-8.02642397e-09f, 3.00262883e-07f, -6.72069427e-06f, 1.02510894e-04f,
                            -1.15122017e-03f, 9.92152281e-03f, -6.64609522e-02f, 3.98939520e-01f};
#pragma unroll
    for (int i = 0; i < N; ++i) {
        if constexpr (ST == 0) {
            xc[i] = __builtin_amdgcn_fmed3f(x[i], -4.0f, 4.0f);
            u[i] = xc[i] * xc[i];
            p[i] = 9.56756410e-11f;
        } else if constexpr (ST <= 8) {
            p[i] = fmaf(p[i], u[i], c[ST - 1]);
        } else if constexpr (ST == 9) {
            p[i] = fmaf(xc[i], p[i], 0.5f);
        } else {
            x[i] = x[i] * p[i];
        }
    }
}

// async global -> LDS copy of 16 bytes per lane; LDS destination = wave-uniform base + lane*16.
__device__ __forceinline__ void glds16(const void *gsrc, void *lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gsrc,
                                     (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, 0);
}

// ds_read_b128 at addr + immediate offset, and a COUNTED wait that names the register it releases (so that no use of it is
// scheduled above the wait): the building blocks of hand-rolled LDS prefetch pipelines (hipcc's own schedule tends to wait
// lgkmcnt(0) right behind the read it has just issued)
typedef __attribute__((ext_vector_type(4))) uint32_t lds_u32x4;
template <int OFF>
__device__ __forceinline__ void lds_read_b128_imm(lds_u32x4 &dst, uint32_t addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory");
}
template <int N>
__device__ __forceinline__ void lgkm_wait_counted(lds_u32x4 &released) {   // all but the N youngest LDS operations are done
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(released) : "n"(N) : "memory");
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void wait_lgkmcnt0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// Opt a kernel in to more than 64 KiB of dynamic LDS.  The attribute is per DEVICE (a process may drive several GPUs), so each
// call site keeps one flag per device: `static DevOnce once; TSIM_MAX_LDS(once, kern, bytes);`
struct DevOnce { bool done[64] = {}; };
#define TSIM_MAX_LDS(once, kern, bytes)                                                                              \
    do {                                                                                                             \
        int dev_ = 0;                                                                                                \
        TSIM_HIP_CHECK(hipGetDevice(&dev_));                                                                         \
        if (dev_ < 0 || dev_ >= 64 || !(once).done[dev_]) {                                                          \
            TSIM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),                                 \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (bytes)));                \
            if (dev_ >= 0 && dev_ < 64) (once).done[dev_] = true;                                                    \
        }                                                                                                            \
    } while (0)

}  // namespace tsim
