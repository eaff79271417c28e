// Device-side building blocks of the HGMM kernels (tree_kernels.hip: one cloud per call; tree_batch.hip: a forest of
// independent clouds per launch set).  Everything here is a template, a forceinline device function or a constant, so
// that both translation units run the SAME arithmetic -- the batched calls are bitwise the serial ones by construction.
#pragma once
#include "hgmm_ctx.h"
#include "wave_ops.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

namespace hgmm {

constexpr double TREE_EPS = 1.0e-15;                 // hgmm_cupy_cpu_working.py:29
constexpr double TWO_PI_POW_1_5 = 15.749609945722419; // (2 pi)^(3/2)
constexpr int PREP_N = 20;   // i00 i01 i02 i11 i12 i22 | mu0 mu1 mu2 | wE | wL | complexity | r00 r01 r02 r11 r12 r22 | kappa | kappa'
// [12..17] R: upper-triangular factor of Sigma^-1 / 2 (R^T R = Sigma^-1 / 2), so that the exponent of the pdf is
//          -(x-mu)^T Sigma^-1 (x-mu) / 2 = -|R (x - mu)|^2 : 9 fma/mul for a point given in coordinates where R mu is
//          precomputed, against 14 for the symmetric form (tree_loglik_kernel, full_fused_kernel).
// [18]     kappa = 1 / (2 lambda_max(Sigma)): the exponent is <= -kappa |x - mu|^2 for every x -- a whole node can be
//          rejected for a whole box of points once kappa dist(box, mu)^2 passes the underflow threshold.
// [19]     kappa' = lambda_max(Sigma^-1) / 2 (an upper bound of it): the exponent is >= -kappa' |x - mu|^2 for every x -- with
//          the farthest corner of a box of points that is a LOWER bound of the node's pdf over the box (the relative
//          reach test of tree_loglik_kernel); -1 where no bound is known.
// A node whose Sigma^-1 is not numerically positive definite although det >= eps (cannot happen for a covariance
// estimated from moments; a caller-supplied table may hold anything) raises bit 0 of the context's tree flags and the
// consumers fall back to the symmetric form.
constexpr int PREP_R = 12, PREP_KAPPA = 18, PREP_KAPPA2 = 19;
constexpr int CH = 256;      // points per chunk = threads per workgroup
constexpr int NMOM = 10;     // m0, m1[3], m2 unique[6] (xx xy xz yy yz zz)

__host__ __device__ inline int64_t level_first(int l) {  // 8 (8^l - 1) / 7
    int64_t p = 1;
    for (int i = 0; i < l; ++i) p *= 8;
    return 8 * (p - 1) / 7;
}

// d^T S d for a symmetric S given by its 6 unique entries, factored so that it costs 11 multiply /
// fma instead of the 17 of the term-by-term form:
//   d0 (s00 d0 + 2 (s01 d1 + s02 d2)) + d1 (s11 d1 + 2 s12 d2) + d2 (s22 d2)
__device__ __forceinline__ double sym3_quad(double s00, double s01, double s02, double s11, double s12,
                                            double s22, double d0, double d1, double d2) {
    const double t0 = fma(2.0, fma(s02, d2, s01 * d1), s00 * d0);
    const double t1 = fma(2.0, s12 * d2, s11 * d1);
    return fma(d2, s22 * d2, fma(d1, t1, d0 * t0));
}

// exp(y) for y <= 0, branch-free, <= 1 ulp, four (or two) independent evaluations interleaved in one asm block so
// that the dependent v_fma_f64 chain of one hides behind the others.
//   n = round(y * 128 / ln 2)  (add / subtract 1.5 * 2^52: the integer sits in the low mantissa word),
//   r = y - n ln2/128 in two pieces (|r| <= ln2/256 = 0.0027),   n = 128 m + j,
//   exp(y) = 2^m * T[j] * e^r,  T[j] = 2^(j/128) from a 1 KB table in LDS (any 64 lanes hit distinct banks or the
//   same word), e^r - 1 = r (1 + r (1/2 + r (1/6 + r (1/24 + r/120)))): degree 5 is enough at that range
//   (r^6/720 < 6e-19), the result is formed as fma(T, e^r - 1, T) and 2^m goes into the exponent field.
// 16 VALU instructions per value; the degree-13 polynomial on |r| <= ln2/2 that this replaces took 21.
// Arguments below -708 are clamped (result < 3.3e-308, i.e. nothing).  A NaN argument is NOT propagated: fmin(NaN, 0)
// is 0, the result is exp(0) = 1 -- a point with a NaN coordinate therefore contributes weight x 1 per node to the
// sums it takes part in instead of poisoning them with NaN (the reference propagates NaN into every moment of the
// nodes the point touches; neither result means anything -- callers must not pass non-finite points).
static __device__ const double EXP2_TAB[128] = {
    0x1.0000000000000p+0, 0x1.0163da9fb3335p+0, 0x1.02c9a3e778061p+0, 0x1.04315e86e7f85p+0,
    0x1.059b0d3158574p+0, 0x1.0706b29ddf6dep+0, 0x1.0874518759bc8p+0, 0x1.09e3ecac6f383p+0,
    0x1.0b5586cf9890fp+0, 0x1.0cc922b7247f7p+0, 0x1.0e3ec32d3d1a2p+0, 0x1.0fb66affed31bp+0,
    0x1.11301d0125b51p+0, 0x1.12abdc06c31ccp+0, 0x1.1429aaea92de0p+0, 0x1.15a98c8a58e51p+0,
    0x1.172b83c7d517bp+0, 0x1.18af9388c8deap+0, 0x1.1a35beb6fcb75p+0, 0x1.1bbe084045cd4p+0,
    0x1.1d4873168b9aap+0, 0x1.1ed5022fcd91dp+0, 0x1.2063b88628cd6p+0, 0x1.21f49917ddc96p+0,
    0x1.2387a6e756238p+0, 0x1.251ce4fb2a63fp+0, 0x1.26b4565e27cddp+0, 0x1.284dfe1f56381p+0,
    0x1.29e9df51fdee1p+0, 0x1.2b87fd0dad990p+0, 0x1.2d285a6e4030bp+0, 0x1.2ecafa93e2f56p+0,
    0x1.306fe0a31b715p+0, 0x1.32170fc4cd831p+0, 0x1.33c08b26416ffp+0, 0x1.356c55f929ff1p+0,
    0x1.371a7373aa9cbp+0, 0x1.38cae6d05d866p+0, 0x1.3a7db34e59ff7p+0, 0x1.3c32dc313a8e5p+0,
    0x1.3dea64c123422p+0, 0x1.3fa4504ac801cp+0, 0x1.4160a21f72e2ap+0, 0x1.431f5d950a897p+0,
    0x1.44e086061892dp+0, 0x1.46a41ed1d0057p+0, 0x1.486a2b5c13cd0p+0, 0x1.4a32af0d7d3dep+0,
    0x1.4bfdad5362a27p+0, 0x1.4dcb299fddd0dp+0, 0x1.4f9b2769d2ca7p+0, 0x1.516daa2cf6642p+0,
    0x1.5342b569d4f82p+0, 0x1.551a4ca5d920fp+0, 0x1.56f4736b527dap+0, 0x1.58d12d497c7fdp+0,
    0x1.5ab07dd485429p+0, 0x1.5c9268a5946b7p+0, 0x1.5e76f15ad2148p+0, 0x1.605e1b976dc09p+0,
    0x1.6247eb03a5585p+0, 0x1.6434634ccc320p+0, 0x1.6623882552225p+0, 0x1.68155d44ca973p+0,
    0x1.6a09e667f3bcdp+0, 0x1.6c012750bdabfp+0, 0x1.6dfb23c651a2fp+0, 0x1.6ff7df9519484p+0,
    0x1.71f75e8ec5f74p+0, 0x1.73f9a48a58174p+0, 0x1.75feb564267c9p+0, 0x1.780694fde5d3fp+0,
    0x1.7a11473eb0187p+0, 0x1.7c1ed0130c132p+0, 0x1.7e2f336cf4e62p+0, 0x1.80427543e1a12p+0,
    0x1.82589994cce13p+0, 0x1.8471a4623c7adp+0, 0x1.868d99b4492edp+0, 0x1.88ac7d98a6699p+0,
    0x1.8ace5422aa0dbp+0, 0x1.8cf3216b5448cp+0, 0x1.8f1ae99157736p+0, 0x1.9145b0b91ffc6p+0,
    0x1.93737b0cdc5e5p+0, 0x1.95a44cbc8520fp+0, 0x1.97d829fde4e50p+0, 0x1.9a0f170ca07bap+0,
    0x1.9c49182a3f090p+0, 0x1.9e86319e32323p+0, 0x1.a0c667b5de565p+0, 0x1.a309bec4a2d33p+0,
    0x1.a5503b23e255dp+0, 0x1.a799e1330b358p+0, 0x1.a9e6b5579fdbfp+0, 0x1.ac36bbfd3f37ap+0,
    0x1.ae89f995ad3adp+0, 0x1.b0e07298db666p+0, 0x1.b33a2b84f15fbp+0, 0x1.b59728de5593ap+0,
    0x1.b7f76f2fb5e47p+0, 0x1.ba5b030a1064ap+0, 0x1.bcc1e904bc1d2p+0, 0x1.bf2c25bd71e09p+0,
    0x1.c199bdd85529cp+0, 0x1.c40ab5fffd07ap+0, 0x1.c67f12e57d14bp+0, 0x1.c8f6d9406e7b5p+0,
    0x1.cb720dcef9069p+0, 0x1.cdf0b555dc3fap+0, 0x1.d072d4a07897cp+0, 0x1.d2f87080d89f2p+0,
    0x1.d5818dcfba487p+0, 0x1.d80e316c98398p+0, 0x1.da9e603db3285p+0, 0x1.dd321f301b460p+0,
    0x1.dfc97337b9b5fp+0, 0x1.e264614f5a129p+0, 0x1.e502ee78b3ff6p+0, 0x1.e7a51fbc74c83p+0,
    0x1.ea4afa2a490dap+0, 0x1.ecf482d8e67f1p+0, 0x1.efa1bee615a27p+0, 0x1.f252b376bba97p+0,
    0x1.f50765b6e4540p+0, 0x1.f7bfdad9cbe14p+0, 0x1.fa7c1819e90d8p+0, 0x1.fd3c22b8f71f1p+0,
};
constexpr int EXP_TAB_N = 128;
// copy the table into LDS (first 128 threads of the workgroup; the caller synchronises)
__device__ __forceinline__ void exp_tab_load(double* __restrict__ tab_lds) {
    if (threadIdx.x < EXP_TAB_N) tab_lds[threadIdx.x] = EXP2_TAB[threadIdx.x];
}

#define HGMM_EXP_CONSTS                                                                                       \
    constexpr double MAGIC = 6755399441055744.0;          /* 1.5 * 2^52 */                                      \
    constexpr double INV = 184.6649652337873;             /* 128 / ln 2 */                                      \
    constexpr double C_HI = 6.93147180369123816490e-01 / 128.0, C_LO = 1.90821492927058770002e-10 / 128.0

__device__ __forceinline__ void exp_nonpos4(const double (&yin)[4], double (&out)[4], const double* __restrict__ tab) {
    HGMM_EXP_CONSTS;
    double r[4], T[4], q[4];
    int m[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const double y = fmax(fmin(yin[k], 0.0), -708.0);
        const double t = fma(y, INV, MAGIC);
        const double nf = t - MAGIC;
        r[k] = fma(nf, -C_LO, fma(nf, -C_HI, y));
        const int n = __double2loint(t);                  // low mantissa word of t = n (two's complement)
        T[k] = tab[n & (EXP_TAB_N - 1)];
        m[k] = n >> 7;
    }
    // (one block of three-address v_fma_f64, consecutive instructions independent: hipcc would pick the two-address
    //  v_fmac_f64 and pay a v_mov_b64 per step, and a block per value is a chain of dependent fp64 fmas)
    asm("v_fma_f64 %0, %4, %12, %13\n\tv_fma_f64 %1, %5, %12, %13\n\tv_fma_f64 %2, %6, %12, %13\n\t"
        "v_fma_f64 %3, %7, %12, %13\n\t"
        "v_fma_f64 %0, %0, %4, %14\n\tv_fma_f64 %1, %1, %5, %14\n\tv_fma_f64 %2, %2, %6, %14\n\t"
        "v_fma_f64 %3, %3, %7, %14\n\t"
        "v_fma_f64 %0, %0, %4, 0.5\n\tv_fma_f64 %1, %1, %5, 0.5\n\tv_fma_f64 %2, %2, %6, 0.5\n\t"
        "v_fma_f64 %3, %3, %7, 0.5\n\t"
        "v_fma_f64 %0, %0, %4, 1.0\n\tv_fma_f64 %1, %1, %5, 1.0\n\tv_fma_f64 %2, %2, %6, 1.0\n\t"
        "v_fma_f64 %3, %3, %7, 1.0\n\t"
        "v_mul_f64 %0, %0, %4\n\tv_mul_f64 %1, %1, %5\n\tv_mul_f64 %2, %2, %6\n\tv_mul_f64 %3, %3, %7\n\t"
        "v_fma_f64 %0, %8, %0, %8\n\tv_fma_f64 %1, %9, %1, %9\n\tv_fma_f64 %2, %10, %2, %10\n\t"
        "v_fma_f64 %3, %11, %3, %11"
        : "=&v"(q[0]), "=&v"(q[1]), "=&v"(q[2]), "=&v"(q[3])
        : "v"(r[0]), "v"(r[1]), "v"(r[2]), "v"(r[3]), "v"(T[0]), "v"(T[1]), "v"(T[2]), "v"(T[3]),
          "v"(1.0 / 120.0), "v"(1.0 / 24.0), "v"(1.0 / 6.0));
#pragma unroll
    for (int k = 0; k < 4; ++k)
        out[k] = __hiloint2double(__double2hiint(q[k]) + (m[k] << 20), __double2loint(q[k]));
}

__device__ __forceinline__ void exp_nonpos2(const double (&yin)[2], double (&out)[2], const double* __restrict__ tab) {
    HGMM_EXP_CONSTS;
    double r[2], T[2], q[2];
    int m[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const double y = fmax(fmin(yin[k], 0.0), -708.0);
        const double t = fma(y, INV, MAGIC);
        const double nf = t - MAGIC;
        r[k] = fma(nf, -C_LO, fma(nf, -C_HI, y));
        const int n = __double2loint(t);
        T[k] = tab[n & (EXP_TAB_N - 1)];
        m[k] = n >> 7;
    }
    asm("v_fma_f64 %0, %2, %6, %7\n\tv_fma_f64 %1, %3, %6, %7\n\t"
        "v_fma_f64 %0, %0, %2, %8\n\tv_fma_f64 %1, %1, %3, %8\n\t"
        "v_fma_f64 %0, %0, %2, 0.5\n\tv_fma_f64 %1, %1, %3, 0.5\n\t"
        "v_fma_f64 %0, %0, %2, 1.0\n\tv_fma_f64 %1, %1, %3, 1.0\n\t"
        "v_mul_f64 %0, %0, %2\n\tv_mul_f64 %1, %1, %3\n\t"
        "v_fma_f64 %0, %4, %0, %4\n\tv_fma_f64 %1, %5, %1, %5"
        : "=&v"(q[0]), "=&v"(q[1])
        : "v"(r[0]), "v"(r[1]), "v"(T[0]), "v"(T[1]), "v"(1.0 / 120.0), "v"(1.0 / 24.0), "v"(1.0 / 6.0));
#pragma unroll
    for (int k = 0; k < 2; ++k)
        out[k] = __hiloint2double(__double2hiint(q[k]) + (m[k] << 20), __double2loint(q[k]));
}
#undef HGMM_EXP_CONSTS

// The same construction on a 2048-entry table (16 KB of LDS) for the throughput-bound kernels: n = round(y 2048 / ln 2),
// |r| <= ln2 / 4096 = 1.7e-4, so e^r - 1 = r (1 + r (1/2 + r / 6)) is enough (r^4 / 24 < 3.5e-17): 4 instead of 6
// instructions for the polynomial; and for exponents that are NON-POSITIVE BY CONSTRUCTION (-|R d|^2) the upper clamp
// goes: 14 VALU instructions per value instead of 17.  The table (correctly rounded 2^(j/2048), formed in long double
// on the host) lives in a context buffer and is copied to LDS by the kernel.
constexpr int EXP_TAB2_BITS = 11;
constexpr int EXP_TAB2_N = 1 << EXP_TAB2_BITS;
template <bool NONPOS>
__device__ __forceinline__ void exp_t11_4(const double (&yin)[4], double (&out)[4], const double* __restrict__ tab) {
    constexpr double MAGIC = 6755399441055744.0;          /* 1.5 * 2^52 */
    constexpr double INV = 2954.6394437405972;            /* 2048 / ln 2 */
    constexpr double C_HI = 6.93147180369123816490e-01 / 2048.0, C_LO = 1.90821492927058770002e-10 / 2048.0;
    double r[4], T[4], q[4];
    int mh[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const double y = NONPOS ? fmax(yin[k], -708.0) : fmax(fmin(yin[k], 0.0), -708.0);
        const double t = fma(y, INV, MAGIC);
        const double nf = t - MAGIC;
        r[k] = fma(nf, -C_LO, fma(nf, -C_HI, y));
        const int n = __double2loint(t);                  // low mantissa word of t = n (two's complement)
        T[k] = tab[n & (EXP_TAB2_N - 1)];
        mh[k] = (int)((unsigned int)(n & ~(EXP_TAB2_N - 1)) << (20 - EXP_TAB2_BITS));   // (n >> 11) << 20: exponent-field increment
    }
    asm("v_fma_f64 %0, %4, %12, 0.5\n\tv_fma_f64 %1, %5, %12, 0.5\n\tv_fma_f64 %2, %6, %12, 0.5\n\t"
        "v_fma_f64 %3, %7, %12, 0.5\n\t"
        "v_fma_f64 %0, %0, %4, 1.0\n\tv_fma_f64 %1, %1, %5, 1.0\n\tv_fma_f64 %2, %2, %6, 1.0\n\t"
        "v_fma_f64 %3, %3, %7, 1.0\n\t"
        "v_mul_f64 %0, %0, %4\n\tv_mul_f64 %1, %1, %5\n\tv_mul_f64 %2, %2, %6\n\tv_mul_f64 %3, %3, %7\n\t"
        "v_fma_f64 %0, %8, %0, %8\n\tv_fma_f64 %1, %9, %1, %9\n\tv_fma_f64 %2, %10, %2, %10\n\t"
        "v_fma_f64 %3, %11, %3, %11"
        : "=&v"(q[0]), "=&v"(q[1]), "=&v"(q[2]), "=&v"(q[3])
        : "v"(r[0]), "v"(r[1]), "v"(r[2]), "v"(r[3]), "v"(T[0]), "v"(T[1]), "v"(T[2]), "v"(T[3]), "v"(1.0 / 6.0));
#pragma unroll
    for (int k = 0; k < 4; ++k) out[k] = __hiloint2double(__double2hiint(q[k]) + mh[k], __double2loint(q[k]));
}
// The same exp in three stages, so that a caller can put the table look-ups of MANY values in flight before the first
// one is needed (the one-block form above takes the table values as inputs of its asm block: the block cannot start
// before every look-up has returned, and the polynomial does not overlap the LDS latency):
//   exp_t11_head   clamp, n = round(y 2048 / ln 2), r = y - n ln2 / 2048, REQUEST T = 2^((n mod 2048) / 2048)
//   exp_t11_poly4  p = e^r - 1 for four values (one asm block, inputs r only)
//   exp_t11_tail   2^(n div 2048) * fma(T, p, T)
struct ExpHead { double r, T; int mh; };
template <bool NONPOS>
__device__ __forceinline__ ExpHead exp_t11_head(double yin, const double* __restrict__ tab) {
    constexpr double MAGIC = 6755399441055744.0, INV = 2954.6394437405972;
    constexpr double C_HI = 6.93147180369123816490e-01 / 2048.0, C_LO = 1.90821492927058770002e-10 / 2048.0;
    const double y = NONPOS ? fmax(yin, -708.0) : fmax(fmin(yin, 0.0), -708.0);
    const double t = fma(y, INV, MAGIC);
    const double nf = t - MAGIC;
    ExpHead h;
    h.r = fma(nf, -C_LO, fma(nf, -C_HI, y));
    const int n = __double2loint(t);
    h.T = tab[n & (EXP_TAB2_N - 1)];
    h.mh = (int)((unsigned int)(n & ~(EXP_TAB2_N - 1)) << (20 - EXP_TAB2_BITS));
    return h;
}
__device__ __forceinline__ void exp_t11_poly4(double r0, double r1, double r2, double r3, double (&p)[4]) {
    asm("v_fma_f64 %0, %4, %8, 0.5\n\tv_fma_f64 %1, %5, %8, 0.5\n\tv_fma_f64 %2, %6, %8, 0.5\n\t"
        "v_fma_f64 %3, %7, %8, 0.5\n\t"
        "v_fma_f64 %0, %0, %4, 1.0\n\tv_fma_f64 %1, %1, %5, 1.0\n\tv_fma_f64 %2, %2, %6, 1.0\n\t"
        "v_fma_f64 %3, %3, %7, 1.0\n\t"
        "v_mul_f64 %0, %0, %4\n\tv_mul_f64 %1, %1, %5\n\tv_mul_f64 %2, %2, %6\n\tv_mul_f64 %3, %3, %7"
        : "=&v"(p[0]), "=&v"(p[1]), "=&v"(p[2]), "=&v"(p[3])
        : "v"(r0), "v"(r1), "v"(r2), "v"(r3), "v"(1.0 / 6.0));
}
__device__ __forceinline__ double exp_t11_tail(const ExpHead& h, double p) {
    const double q = fma(h.T, p, h.T);
    return __hiloint2double(__double2hiint(q) + h.mh, __double2loint(q));
}
__device__ __forceinline__ void exp_tab2_load(double* __restrict__ tab_lds, const double* __restrict__ tab_g) {
    for (int e = threadIdx.x; e < EXP_TAB2_N; e += blockDim.x) tab_lds[e] = tab_g[e];
}

// log(x) for a positive, finite, NORMAL x (the callers clamp at eps = 1e-15 or test for 0 first): the classic reduction
// x = 2^k (1 + f), sqrt(1/2) < 1 + f < sqrt(2), s = f / (2 + f), log(1 + f) = f - f^2/2 + s (f^2/2 + R(s^2)) with the
// degree-7 minimax R of fdlibm's e_log.c (error < 1 ulp) -- ~40 VALU instructions against the library call's ~95 with its
// special cases.  Used where a logarithm is taken per point and launch (the float32-pdf log-likelihood, the level-0 q of
// the E-step); the float64 log-likelihood kernels keep the library call and with it round 5's bits.
constexpr double LN_TREE_EPS = -34.538776394910684;  // log(1e-15)
__device__ __forceinline__ double log_pos_f64(double x) {
    constexpr double LN2_HI = 6.93147180369123816490e-01, LN2_LO = 1.90821492927058770002e-10;
    constexpr double LG1 = 6.666666666666735130e-01, LG2 = 3.999999999940941908e-01, LG3 = 2.857142874366239149e-01,
                     LG4 = 2.222219843214978396e-01, LG5 = 1.818357216161805012e-01, LG6 = 1.531383769920937332e-01,
                     LG7 = 1.479819860511658591e-01;
    int hx = __double2hiint(x);
    const int lx = __double2loint(x);
    int k = (hx >> 20) - 1023;
    hx &= 0x000fffff;
    const int i = (hx + 0x95f64) & 0x100000;
    const double m = __hiloint2double(hx | (i ^ 0x3ff00000), lx);      // x / 2^k normalised into [sqrt(1/2), sqrt(2))
    k += i >> 20;
    const double f = m - 1.0;
    const double s = f / (2.0 + f);
    const double dk = (double)k;
    const double z = s * s, w = z * z;
    const double t1 = w * fma(w, fma(w, LG6, LG4), LG2);
    const double t2 = z * fma(w, fma(w, fma(w, LG7, LG5), LG3), LG1);
    const double R = t2 + t1;
    const double hfsq = 0.5 * f * f;
    return dk * LN2_HI - ((hfsq - (s * (hfsq + R) + dk * LN2_LO)) - f);
}

// ------------------------------------------------------------------------------------------
// per-node preparation
// ------------------------------------------------------------------------------------------
__device__ inline double sym3_min_eig_over_trace(double a00, double a01, double a02, double a11,
                                                 double a12, double a22) {
    // closed-form eigenvalues of a symmetric 3x3 (trigonometric solution)
    const double tr = a00 + a11 + a22;
    const double p1 = a01 * a01 + a02 * a02 + a12 * a12;
    double e_min;
    if (p1 == 0.0) {
        e_min = fmin(a00, fmin(a11, a22));
    } else {
        const double q = tr / 3.0;
        const double b00 = a00 - q, b11 = a11 - q, b22 = a22 - q;
        const double p2 = b00 * b00 + b11 * b11 + b22 * b22 + 2.0 * p1;
        const double p = sqrt(p2 / 6.0);
        const double ip = 1.0 / p;
        const double c00 = b00 * ip, c01 = a01 * ip, c02 = a02 * ip, c11 = b11 * ip, c12 = a12 * ip,
                     c22 = b22 * ip;
        double r = 0.5 * (c00 * (c11 * c22 - c12 * c12) - c01 * (c01 * c22 - c12 * c02) +
                          c02 * (c01 * c12 - c11 * c02));
        r = r < -1.0 ? -1.0 : (r > 1.0 ? 1.0 : r);
        const double phi = acos(r) / 3.0;
        e_min = q + 2.0 * p * cos(phi + 2.0943951023931953);   // + 2 pi / 3
    }
    return e_min / tr;
}

// An upper bound of the largest eigenvalue of a symmetric 3x3 that converges to it: Newton's iteration on the
// characteristic polynomial started at the Frobenius norm (>= the spectral radius).  To the right of the largest
// root the polynomial is increasing and convex, so the iterates decrease monotonically towards the root and EVERY
// iterate is a valid bound -- which is all the reach test of the log-likelihood kernel needs.  ~60 flops; the
// closed form (sqrt, acos, cos) this replaces cost more than the whole rest of a node's M-step.
__device__ inline double sym3_max_eig_upper(double a00, double a01, double a02, double a11, double a12, double a22) {
    const double c2 = a00 + a11 + a22;
    const double c1 = (a00 * a11 - a01 * a01) + (a00 * a22 - a02 * a02) + (a11 * a22 - a12 * a12);
    const double c0 = a00 * (a11 * a22 - a12 * a12) - a01 * (a01 * a22 - a12 * a02) + a02 * (a01 * a12 - a11 * a02);
    double lam = sqrt(a00 * a00 + a11 * a11 + a22 * a22 + 2.0 * (a01 * a01 + a02 * a02 + a12 * a12));
    if (!(lam > 0.0)) return lam;
#pragma unroll
    for (int it = 0; it < 6; ++it) {
        const double pv = fma(fma(lam - c2, lam, c1), lam, -c0);                 // p(lam)
        const double dv = fma(fma(3.0, lam, -2.0 * c2), lam, c1);                // p'(lam)
        const double nxt = (dv > 0.0) ? lam - pv / dv : lam;
        lam = (nxt < lam && nxt > 0.0) ? nxt : lam;                              // never move up, never leave the right branch
    }
    return lam * (1.0 + 1.0e-9);
}

__device__ __forceinline__ void prep_node(double p, double m0, double m1, double m2, double c00, double c01,
                                          double c02, double c10, double c11, double c12, double c20,
                                          double c21, double c22, double* __restrict__ o, int* __restrict__ flags,
                                          bool with_complexity = true) {
    const double det = c00 * (c11 * c22 - c12 * c21) - c01 * (c10 * c22 - c12 * c20) +
                       c02 * (c10 * c21 - c11 * c20);
#pragma unroll
    for (int e = PREP_R; e < PREP_N; ++e) o[e] = 0.0;
    o[PREP_KAPPA2] = -1.0;
    if (det < TREE_EPS) {           // gaussianPdf returns 0 (hgmm_cupy_cpu_working.py:65-67)
        o[0] = o[1] = o[2] = o[3] = o[4] = o[5] = 0.0;
        o[9] = 0.0;
        o[10] = 0.0;
    } else {
        const double id = 1.0 / det;
        // symmetric part of the adjugate (covariances are symmetric by construction)
        o[0] = (c11 * c22 - c12 * c21) * id;
        o[1] = (c02 * c21 - c01 * c22) * id;
        o[2] = (c01 * c12 - c02 * c11) * id;
        o[3] = (c00 * c22 - c02 * c20) * id;
        o[4] = (c02 * c10 - c00 * c12) * id;
        o[5] = (c00 * c11 - c01 * c10) * id;
        const double coef = 1.0 / (sqrt(det) * TWO_PI_POW_1_5);
        o[9] = p * coef;
        o[10] = (p < TREE_EPS) ? 0.0 : p * coef;   // logLikelihoodValue skips pi < eps (C:80)
        // Cholesky factor of A = Sigma^-1 / 2:  A = R^T R, R upper triangular
        const double a00 = 0.5 * o[0], a01 = 0.5 * o[1], a02 = 0.5 * o[2], a11 = 0.5 * o[3], a12 = 0.5 * o[4],
                     a22 = 0.5 * o[5];
        const double r00 = sqrt(a00);
        const double r01 = a01 / r00, r02 = a02 / r00;
        const double r11 = sqrt(a11 - r01 * r01);
        const double r12 = (a12 - r01 * r02) / r11;
        const double r22 = sqrt(a22 - r02 * r02 - r12 * r12);
        const bool pd = (a00 > 0.0) && (r11 > 0.0) && (r22 > 0.0) && (r22 == r22) && (r11 == r11) &&
                        (fabs(r12) < 1.0e300) && (fabs(r02) < 1.0e300);
        if (pd) {
            o[PREP_R + 0] = r00; o[PREP_R + 1] = r01; o[PREP_R + 2] = r02;
            o[PREP_R + 3] = r11; o[PREP_R + 4] = r12; o[PREP_R + 5] = r22;
            const double lmax = sym3_max_eig_upper(c00, c01, c02, c11, c12, c22);
            o[PREP_KAPPA] = (lmax > 0.0 && lmax == lmax && lmax < 1.0e300) ? 0.5 / lmax : 0.0;
            const double imax = sym3_max_eig_upper(o[0], o[1], o[2], o[3], o[4], o[5]);
            o[PREP_KAPPA2] = (imax > 0.0 && imax == imax && imax < 1.0e300) ? 0.5 * imax : -1.0;
        } else if (flags) {
            atomicOr(flags, 1);
        }
    }
    o[6] = m0; o[7] = m1; o[8] = m2;
    // (the 'complexity' ratio feeds the registration E-step only: the build takes it once, when the tree is finished)
    if (with_complexity) o[11] = sym3_min_eig_over_trace(c00, c01, c02, c11, c12, c22);
}

// ------------------------------------------------------------------------------------------
// The level's stop rule WITHOUT a reduction tail: every workgroup of the NEXT launch in the stream (the next
// iteration's E-step, or the one-workgroup tree_close_kernel behind a level's last budgeted iteration) adds up the
// previous iteration's shares of q for itself -- same fixed order everywhere, so the same q and the same verdict -- and
// workgroup 0 records it (trace, iteration count, stop flag, the host's progress word).  The log-likelihood kernels
// just store their shares: no ticket, no atomic, no last-workgroup pass (store_block_q's chain of a drained store, two
// arrival counters and a read-back was ~3 us at the end of every C4 iteration), and the build path is free of atomics.
// State is double-buffered by launch parity (launch e reads slot (e - 1) & 1 and writes slot e & 1): a workgroup that
// starts late must not read what workgroup 0 of its own launch has just written.
// ------------------------------------------------------------------------------------------
struct TreeLoopState { int it; int pad; double prev_q; };
struct TreeFollow {                      // q_blocks == nullptr: nothing to follow (a level's first iteration)
    const double* q_blocks; int nb;
    const TreeLoopState* prev; TreeLoopState* next;
    int* done; double ls; int max_iters; double* trace; int trace_cap; unsigned long long* host_word;
    int* final_it = nullptr;             // forest (tree_batch.hip): receives the level's iteration count together with *done
};
// what the workgroup that speaks for the loop leaves behind (one thread)
__device__ __forceinline__ void tree_follow_record(const TreeFollow& f, int it, double q, bool stop_now) {
    if (it < f.trace_cap) f.trace[it] = q;
    f.next->it = it + 1;
    f.next->prev_q = q;
    if (stop_now) {
        if (f.final_it) *f.final_it = it + 1;
        *f.done = 1;
    }
    if (f.host_word)
        __hip_atomic_store(f.host_word, ((unsigned long long)(stop_now ? 1 : 0) << 32) | (unsigned long long)(unsigned)(it + 1),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// true: the level has stopped -- this launch has nothing to do.  Called by all threads of a CH-thread workgroup.
// `speaker`: this workgroup records the verdict (serial build: workgroup 0 of the launch; forest: the cloud's first)
__device__ __forceinline__ bool tree_follow(const TreeFollow& f, int stop_flag, double* sh4, bool speaker) {
    const double prev_q = f.prev->prev_q;                 // (requested together with the shares)
    const int it = f.prev->it;
    double acc = 0.0;
    for (int i = threadIdx.x; i < f.nb; i += CH) acc += f.q_blocks[i];
    if (stop_flag) return true;                           // kernel-uniform
    acc = wave_sum_f64(acc);                              // (the order of store_block_q's last workgroup / tree_sum_kernel)
    if (lane_id() == 0) sh4[wave_in_block()] = acc;
    __syncthreads();
    const double q = sh4[0] + sh4[1] + sh4[2] + sh4[3];
    const bool stop_now = fabs(q - prev_q) < f.ls || it + 1 >= f.max_iters;      // (hgmm_gpu.py:532; prev_q starts at 0)
    if (speaker && threadIdx.x == 0) {
        tree_follow_record(f, it, q, stop_now);
    }
    return stop_now;
}
// The same for a workgroup of ONE wave (tree_moments_kernel): lane t adds up what threads t, t + 64, t + 128, t + 192 of a
// CH-thread workgroup would have -- four separate sums, reduced and combined exactly as above, so the same q bit for bit.
struct TreeFollowLoads { double acc[CH / 64]; double prev_q; int it; };
// (the loads first -- the caller puts its own requests behind them and only then asks for the verdict, so that the two
//  chains of trips to memory run side by side)
__device__ __forceinline__ TreeFollowLoads tree_follow_wave_load(const TreeFollow& f) {
    TreeFollowLoads r;
    r.prev_q = f.prev->prev_q;
    r.it = f.prev->it;
#pragma unroll
    for (int w = 0; w < CH / 64; ++w) r.acc[w] = 0.0;
    for (int i0 = threadIdx.x; i0 < f.nb; i0 += CH) {
#pragma unroll
        for (int w = 0; w < CH / 64; ++w)
            if (i0 + 64 * w < f.nb) r.acc[w] += f.q_blocks[i0 + 64 * w];
    }
    return r;
}
__device__ __forceinline__ bool tree_follow_wave_verdict(const TreeFollow& f, TreeFollowLoads r, int stop_flag, bool speaker) {
    if (stop_flag) return true;
#pragma unroll
    for (int w = 0; w < CH / 64; ++w) r.acc[w] = wave_sum_f64(r.acc[w]);
    double q = r.acc[0];
#pragma unroll
    for (int w = 1; w < CH / 64; ++w) q += r.acc[w];
    const double prev_q = r.prev_q;
    const int it = r.it;
    const bool stop_now = fabs(q - prev_q) < f.ls || it + 1 >= f.max_iters;
    if (speaker && threadIdx.x == 0) {
        tree_follow_record(f, it, q, stop_now);
    }
    return stop_now;
}

// ------------------------------------------------------------------------------------------
// E-step of one tree level
// ------------------------------------------------------------------------------------------
constexpr int ES_LD = 66;            // LDS row stride (doubles) of the moment contraction's operands: conflict-free fragment reads
typedef double double4_es __attribute__((ext_vector_type(4)));
// HALF: the wave's 64 points go through the LDS transpose 32 at a time (18 rows x 34 instead of x 66 doubles per wave:
// 23 KB of LDS per workgroup instead of 42) -- same products, same order of accumulation, so the same sums bit for bit.
// A million-point level is 3907 chunks = 15 workgroups per CU, of which the LDS admitted 3 at a time: the launch was
// five rounds of one latency chain each.  Small clouds (less than one workgroup per CU) keep the one-pass form.
struct TreeEstepArgs {
    const double* xs; int64_t n_pad; const double* prep; const int* chunk_desc; const int* n_chunks;
    int64_t parent_level_first; int level; double* partials; int* cur_sorted; const int* done;
    // LEVEL 0 of the overlapped builds (small clouds, forests): the level's nodes ARE the root's eight children, so the
    // log-likelihood of iteration e -- sum_i log max(sum_j [pi_j >= eps] pi_j N(x_i; j), eps), logLikelihoodValue C:72-85 --
    // is a sum over the very eight terms iteration e + 1's E-step forms from the same parameters.  With q_shares set the
    // E-step stores its chunk's share of it (chunk c -> q_shares[c]) and no log-likelihood workgroup runs for the level.
    double* q_shares = nullptr;
};
// A FOREST (tree_batch.hip): B independent clouds whose points lie back to back in one resident cloud and whose trees are
// built by the same launches.  At level l the forest has B 8^l parent segments; segment p belongs to cloud p >> 3 l and is
// the (p mod 8^l)-th parent of that cloud's level; cloud b's nodes are [b T, (b + 1) T) of the node tables.  Every
// workgroup works for exactly one cloud and runs the arithmetic of the serial build of that cloud -- the same chunks, the
// same orders of summation -- so a forest's trees are the serial trees bit for bit.
struct alignas(128) ForestCloud {
    int pt_first, pt_count;          // its points in the forest's order (every level's partition keeps a cloud contiguous)
    int ll_gx, ll_gy, ll_per_chunk;  // the log-likelihood decomposition the SERIAL build uses for this cloud at this level: gx
                                     //   blocks of PTS CH points x gy node chunks (reproduced for the order of its sums)
    int q_first, q_count;            // its shares of q in block_q
    int done;                        // device-written: the level has stopped ...
    int final_it;                    //   ... after this many iterations
    int pad;
    double n_total;                  // pi = m0 / n_total
    TreeLoopState st[2];             // tree_follow's state, double-buffered by launch parity
};
struct ForestArgs {
    ForestCloud* clouds; int B; int T; int shift;       // shift = 3 l
    double ls; int max_iters; double* trace_base; int trace_cap; int L; int level; unsigned long long* host_words;
};
// the follow of cloud b for the launch that follows iteration e - 1 (fc: the cloud's entry, already loaded)
__device__ __forceinline__ TreeFollow forest_follow(const ForestArgs& fa, int b, int e, const double* block_q, int q_first,
                                                    int q_count) {
    ForestCloud* fc = fa.clouds + b;
    return TreeFollow{block_q + q_first, q_count, fc->st + ((e - 1) & 1), fc->st + (e & 1), &fc->done, fa.ls, fa.max_iters,
                      fa.trace_base + ((size_t)b * fa.L + fa.level) * fa.trace_cap, fa.trace_cap,
                      fa.host_words ? fa.host_words + b : nullptr, &fc->final_it};
}
// (c = the workgroup's chunk: blockIdx.x in tree_estep_kernel, an offset of it in tree_ll_estep_kernel)
// LDS of one E-step workgroup, in doubles: exp table, tree_follow's four, the waves' [8][NMOM] sums, gamma rows, feature rows
// (the waves' [8][NMOM] sums live in the first rows of the wave's own gamma block: a wave writes them once its last
//  fragment reads have landed in registers -- 20.6 KB per workgroup in the two-pass form, seven workgroups per CU)
template <bool HALF>
constexpr int tree_estep_lds() { return EXP_TAB_N + 4 + (CH / 64) * (8 + NMOM) * (HALF ? ES_LD / 2 + 1 : ES_LD); }
template <bool HALF, bool FOREST = false>
__device__ __forceinline__ void tree_estep_body(const int c, const TreeEstepArgs& a, const TreeFollow& follow,
                                                double* __restrict__ smem, const ForestArgs* fa = nullptr) {
    const double* __restrict__ xs = a.xs;
    const int64_t n_pad = a.n_pad;
    const double* __restrict__ prep = a.prep;
    const int* __restrict__ chunk_desc = a.chunk_desc;
    const int* __restrict__ n_chunks = a.n_chunks;
    const int64_t parent_level_first = a.parent_level_first;
    const int level = a.level;
    double* __restrict__ partials = a.partials;
    int* __restrict__ cur_sorted = a.cur_sorted;
    const int* __restrict__ done = a.done;
    // (the stop flag, the chunk count and this chunk's descriptor are requested together -- the descriptor table is
    //  allocated for the whole grid, so the read is safe before the count is known: one trip to memory instead of three)
    int stop_flag = (!FOREST && done) ? *done : 0; // the level converged in an earlier iteration of this batch
    const int chunks_now = *n_chunks;
    const int p = chunk_desc[3 * c + 0], begin = chunk_desc[3 * c + 1], end = chunk_desc[3 * c + 2];
    double* exp_tab = smem;
    double* sh_follow = smem + EXP_TAB_N;
    exp_tab_load(exp_tab);                         // (synchronised below, once the points' loads are on their way)
    if (c >= chunks_now) return;                   // (workgroup 0 always owns a chunk)
    const int i = begin + (int)threadIdx.x;
    const bool active = i < end;
    double x0 = 0.0, x1 = 0.0, x2 = 0.0;
    int pl = p;                                    // the parent's index within its cloud's level
    int64_t node_off = 0;                          // first node of the cloud's tree
    if constexpr (FOREST) {
        // (the cloud's stop flag hangs on the descriptor: it is requested together with the points)
        const int b = p >> fa->shift;
        pl = p & ((1 << fa->shift) - 1);
        node_off = (int64_t)b * fa->T;
        stop_flag = fa->clouds[b].done;
        if (active) { x0 = xs[i]; x1 = xs[n_pad + i]; x2 = xs[2 * n_pad + i]; }
        if (stop_flag) return;                     // (a forest's E-steps never follow: launch e = 0 and the speculative ones)
    } else {
        if (follow.q_blocks) {
            if (tree_follow(follow, stop_flag, sh_follow, blockIdx.x == 0)) return;   // the previous iteration's q stopped the level
        } else if (stop_flag) {
            return;
        }
    }
    // (The children's parameters are read by scalar loads, ~25 per wave.  Round 6 suspected them -- in a forest every CU sees
    //  the nodes of every cloud in turn and the 16 KB scalar cache misses, SQ_INST_LEVEL_SMEM / SQ_INSTS_SMEM -- and
    //  staged the 8 x 11 values through LDS with ONE vector load per workgroup, requested with the points: 24 more
    //  registers, and the build of 32 bunny scans went from 9.04 to 9.52 ms.  The scalar loads stay.)
    // node id of the parent: level 0 -> pseudo-parent -1; child(j) = 8 (j + 1)
    const int64_t parent_node = (level == 0) ? -1 : parent_level_first + pl;
    const int64_t j0 = node_off + 8 * (parent_node + 1);
    if constexpr (!FOREST) {
        if (active) { x0 = xs[i]; x1 = xs[n_pad + i]; x2 = xs[2 * n_pad + i]; }
    }
    __syncthreads();                               // exp_tab

    double g[8];
    double den = 0.0;
    {
        // the 8 children's exponents, then two interleaved branch-free exponentials of four (a child with
        // wE = 0 -- pi = 0 or a singular covariance -- has an all-zero inverse: exponent 0, weight 0)
        double yv[8], ev[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const double* pr = prep + PREP_N * (j0 + k);
            const double d0 = x0 - pr[6], d1 = x1 - pr[7], d2 = x2 - pr[8];
            yv[k] = -0.5 * sym3_quad(pr[0], pr[1], pr[2], pr[3], pr[4], pr[5], d0, d1, d2);
        }
        const double ya[4] = {yv[0], yv[1], yv[2], yv[3]}, yb[4] = {yv[4], yv[5], yv[6], yv[7]};
        double ea[4], eb[4];
        exp_nonpos4(ya, ea, exp_tab);
        exp_nonpos4(yb, eb, exp_tab);
#pragma unroll
        for (int k = 0; k < 4; ++k) { ev[k] = ea[k]; ev[4 + k] = eb[k]; }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const double wE = prep[PREP_N * (j0 + k) + 9];
            // below the exponent range the library exp returns exactly 0; keep that (the clamp gives 3e-308)
            g[k] = (wE == 0.0 || yv[k] < -745.0) ? 0.0 : wE * ev[k];
            den += g[k];
        }
    }
    if (a.q_shares) {                                  // (kernel-uniform)
        // wL = wE, or 0 where pi < eps (prep_node): the log-likelihood leaves those nodes out, the E-step does not
        double den_l = 0.0;
#pragma unroll
        for (int k = 0; k < 8; ++k) den_l += (prep[PREP_N * (j0 + k) + 10] == 0.0) ? 0.0 : g[k];
        const double lq = wave_sum_f64(active ? log_pos_f64(fmax(den_l, TREE_EPS)) : 0.0);
        if (lane_id() == 0) sh_follow[wave_in_block()] = lq;      // (read behind the barrier in front of the partials)
    }
    // gamma = g / den if den > eps else 0; arg-max = first maximum  (C:174-187)
    // (round 6: ONE division per point -- gamma_k = g_k (1 / den), within an ulp of g_k / den; eight fp64 divisions were
    //  ~90 of the E-step's ~550 instructions per point)
    const bool good = den > TREE_EPS;
    const double inv_den = good ? 1.0 / den : 0.0;
    int am = 0;
    double best = -1.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        g[k] = g[k] * inv_den;
        if (g[k] > best) { best = g[k]; am = k; }
        if (g[k] < TREE_EPS || !active) g[k] = 0.0;      // accumulate() ignores gamma < eps (C:100)
    }
    if (active) cur_sorted[i] = (int)(j0 + am);

    // The wave's 8 x 10 moment sums  M[k][m] = sum_lanes gamma[k] f[m]  as ONE dense contraction on the fp64 matrix
    // cores: gamma and the features go through LDS (component- / feature-major, one row per child resp. feature),
    // 16 v_mfma_f64_16x16x4_f64 walk the wave's 64 points four at a time.  (Round 2 took 80 DPP wave reductions here,
    // ~1400 dependent fp64 instructions per wave -- a third of this latency-bound kernel's time at C4.)  Rows 8..15 of
    // the A operand and columns 10..15 of B alias rows that exist: their products land in accumulator entries nobody reads.
    constexpr int LD = HALF ? ES_LD / 2 + 1 : ES_LD;          // 34 / 66: the same bank pattern (stride = 4 mod 64 dwords)
    constexpr int PASS_PTS = HALF ? 32 : 64;
    double (*GS)[8][LD] = reinterpret_cast<double (*)[8][LD]>(smem + EXP_TAB_N + 4);
    double (*FS)[NMOM][LD] = reinterpret_cast<double (*)[NMOM][LD]>(smem + EXP_TAB_N + 4 + (CH / 64) * 8 * LD);
    static_assert(8 * LD >= 8 * NMOM, "the wave's sums fit into its gamma block");
    auto sh = [&](int ww) -> double* { return &GS[ww][0][0]; };      // wave ww's [8][NMOM] sums (aliases its gamma rows)
    const int w = wave_in_block();
    const int lane = lane_id();
    {
        const double f[NMOM] = {1.0, x0, x1, x2, x0 * x0, x0 * x1, x0 * x2, x1 * x1, x1 * x2, x2 * x2};
        const int a_idx = lane & 15, b_idx = lane >> 4;
        const double* ga = &GS[w][a_idx & 7][b_idx];
        const double* fb = &FS[w][a_idx < NMOM ? a_idx : NMOM - 1][b_idx];
        double4_es acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int pass = 0; pass < 64 / PASS_PTS; ++pass) {
            if (pass > 0) __builtin_amdgcn_wave_barrier();     // the previous pass's fragment reads are done (same wave, in order)
            if (lane / PASS_PTS == pass) {
                const int col = lane % PASS_PTS;
#pragma unroll
                for (int k = 0; k < 8; ++k) GS[w][k][col] = g[k];
#pragma unroll
                for (int m = 0; m < NMOM; ++m) FS[w][m][col] = f[m];
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);        // lgkmcnt(0): this wave's LDS writes have landed
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int st = 0; st < PASS_PTS / 4; ++st)
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(ga[4 * st], fb[4 * st], acc, 0, 0, 0);
        }
        // D layout (f64 16x16x4): row (child) = (lane >> 4) + 4 r, column (feature) = lane & 15
        __builtin_amdgcn_s_waitcnt(0xc07f);            // lgkmcnt(0): the last fragments are in registers
        __builtin_amdgcn_wave_barrier();
        if (a_idx < NMOM) {
            sh(w)[b_idx * NMOM + a_idx] = acc[0];
            sh(w)[(b_idx + 4) * NMOM + a_idx] = acc[1];
        }
    }
    __syncthreads();
    if (threadIdx.x < 8 * NMOM) {
        double t = 0.0;
#pragma unroll
        for (int ww = 0; ww < CH / 64; ++ww) t += sh(ww)[threadIdx.x];
        partials[(size_t)c * (8 * NMOM) + threadIdx.x] = t;
    }
    if (a.q_shares && threadIdx.x == CH - 1) {
        double t = 0.0;
#pragma unroll
        for (int ww = 0; ww < CH / 64; ++ww) t += sh_follow[ww];
        a.q_shares[c] = t;
    }
}

// one wave per child node of the level: fixed-order sum of its parent's chunk partials
// ML estimate of one node from its moments (mlEstimator, hgmm_cupy_cpu_working.py:109-119) followed by
// the node's E-step preparation, so that no separate prep launch is needed.
__device__ __forceinline__ void mstep_node(const double* __restrict__ m, int64_t j, double n_points_total,
                                           double ld, double* __restrict__ pi, double* __restrict__ mu,
                                           double* __restrict__ cov, double* __restrict__ prep,
                                           int* __restrict__ flags, bool with_complexity = true) {
    const double m0 = m[0];
    double* c = cov + 9 * j;
    if (m0 < ld) {
        pi[j] = 0.0;
        mu[3 * j] = mu[3 * j + 1] = mu[3 * j + 2] = 0.0;
        for (int e = 0; e < 9; ++e) c[e] = (e % 4 == 0) ? 1.0 : 0.0;
        if (prep) prep_node(0.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 1.0, prep + PREP_N * j, flags,
                            with_complexity);
        return;
    }
    const double p = m0 / n_points_total;
    pi[j] = p;
    const double u0 = m[1] / m0, u1 = m[2] / m0, u2 = m[3] / m0;
    mu[3 * j] = u0; mu[3 * j + 1] = u1; mu[3 * j + 2] = u2;
    const double s00 = m[4] / m0 - u0 * u0, s01 = m[5] / m0 - u0 * u1, s02 = m[6] / m0 - u0 * u2,
                 s11 = m[7] / m0 - u1 * u1, s12 = m[8] / m0 - u1 * u2, s22 = m[9] / m0 - u2 * u2;
    c[0] = s00; c[1] = s01; c[2] = s02; c[3] = s01; c[4] = s11; c[5] = s12; c[6] = s02; c[7] = s12; c[8] = s22;
    if (prep) prep_node(p, u0, u1, u2, s00, s01, s02, s01, s11, s12, s02, s12, s22, prep + PREP_N * j, flags,
                        with_complexity);
}

// ------------------------------------------------------------------------------------------
// level log-likelihood over ALL nodes of the level (logLikelihoodValue, C:72-85)
// ------------------------------------------------------------------------------------------
constexpr int LL_TILE = 256;
// grid = (point blocks, node chunks).  With one chunk the per-point log is taken here; with several
// (small clouds: not enough point blocks to fill 256 CUs) the per-chunk sums go to `partial`
// [chunk][point] and tree_loglik_finish_kernel adds them in fixed order.
// Stores a workgroup's share of q; with `ticket` the workgroup that finishes last also adds up all
// shares -- in the same fixed order as tree_sum_kernel -- so that no separate reduction launch is
// needed (the result does not depend on which workgroup happens to be last).
struct TreeCtl { int done; int it; double prev_q; };
// host_word (may be null): a word of pinned HOST memory that receives (done << 32 | iterations) after every update, so
// that the host can follow the loop without a copy, an event or a synchronisation (hgmm_tree_build)
struct TreeStop { TreeCtl* ctl; double ls; int max_iters; double* trace; int trace_cap; unsigned long long* host_word = nullptr; };   // ctl == nullptr: not here
// the stop rule of one tree level (see tree_ctl_kernel); one thread
__device__ __forceinline__ void tree_ctl_update(double q, const TreeStop& st) {
    TreeCtl* ctl = st.ctl;
    const int it = ctl->it;
    if (it < st.trace_cap) st.trace[it] = q;
    ctl->it = it + 1;
    const int stop_now = (fabs(q - ctl->prev_q) < st.ls || it + 1 >= st.max_iters) ? 1 : 0;
    if (stop_now) ctl->done = 1;
    ctl->prev_q = q;
    if (st.host_word)
        __hip_atomic_store(st.host_word, ((unsigned long long)stop_now << 32) | (unsigned long long)(unsigned)(it + 1),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// (With `stop.ctl` set the workgroup that finishes last also applies the level's stop rule -- every other workgroup
//  of the launch has passed its own look at the flag by then -- which saves the one-thread launch per iteration.)
// Hand-off between workgroups without fences (guide, inter-workgroup visibility: "8-byte agent-scope atomics on both
// sides" is a complete protocol): a share is published with ONE relaxed agent-scope atomic store (write-through to
// memory, not parked in this XCD's L2), the publishing lane drains its store, then takes a ticket; the workgroup
// that draws the last ticket reads the shares back with relaxed agent-scope atomic loads (served below the L1 of its
// CU).  Round 2 used __threadfence() on both sides -- L2 write-back + L1 invalidate, ~3.5 us each on this chip:
// most of the duration of these microsecond kernels at C4.
// Tickets are taken in TWO levels: workgroup b draws from counter 1 + (b mod NG), the workgroup that completes a
// group draws from counter 0, the one that completes counter 0 is last.  Agent-scope atomics on ONE address are
// served one after the other at ~20 ns each on this chip: with a single counter the 629 workgroups of the
// small-cloud log-likelihood spent 12 us queueing for their tickets (measured: 15.5 us for an 8-node level).
constexpr int TICKET_GROUPS = 64;                        // counter 0 = top level, counters 1 .. 64 = groups
// ... and the counters sit 4 KB apart: device-scope atomics are executed at the memory side, one queue per channel --
// 65 counters in three cache lines still queued behind each other (15.3 us for the 8-node level, unchanged).
constexpr int TICKET_STRIDE = 1024;                      // unsigned ints between two counters
__device__ __forceinline__ void store_block_q(double value, double* __restrict__ block_q, const int bx, int nb,
                                              unsigned int* __restrict__ ticket, double* __restrict__ q_out,
                                              const TreeStop& stop) {
    __shared__ bool is_last;
    __shared__ double sh_fin[4];
    if (threadIdx.x == 0) {
        is_last = false;
        if (ticket) {
            __hip_atomic_store(block_q + bx, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the share has left this CU before the ticket does
            const int ng = nb < TICKET_GROUPS ? nb : TICKET_GROUPS;
            const int g = (int)((unsigned)bx % (unsigned)ng);
            const unsigned int members = (unsigned int)((nb - g + ng - 1) / ng);
            unsigned int* mine = ticket + (size_t)(1 + g) * TICKET_STRIDE;
            if (__hip_atomic_fetch_add(mine, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == members - 1) {
                __hip_atomic_store(mine, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);             // ready for the next launch
                is_last = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned int)(ng - 1);
            }
        } else {
            block_q[bx] = value;
        }
    }
    __syncthreads();
    if (!is_last) return;
    double acc = 0.0;
    for (int i = threadIdx.x; i < nb; i += CH)
        acc += __hip_atomic_load(block_q + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    acc = wave_sum_f64(acc);
    if (lane_id() == 0) sh_fin[wave_in_block()] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double q = sh_fin[0] + sh_fin[1] + sh_fin[2] + sh_fin[3];
        *q_out = q;
        __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (stop.ctl) tree_ctl_update(q, stop);
    }
}

// PTS points per thread: the node parameters come out of LDS as wave-wide broadcast reads (10 x 8 B x
// 64 lanes per node and wave, ~40 LDS cycles for a CU whose four SIMDs need ~12 VALU cycles each for the
// quadratic form), so the kernel is LDS-bound at one point per thread; every further point reuses the
// same ten values.
//
// Round 3:
//   * LOCAL ORIGIN + TRIANGULAR FORM.  A workgroup's points are neighbours (the cloud is regrouped per parent at every
//     level), so they are expressed relative to the workgroup's first point c; the exponent is taken as
//     -|R (x - c) - R (mu - c)|^2 with R^T R = Sigma^-1 / 2 (prep[12..17]) and b = R (mu - c) formed once per node and
//     workgroup while the tile is loaded: 9 fma / mul per (point, node) pair instead of 14 (3 subtractions + the
//     symmetric form).  Both terms are of the size of the workgroup's extent, so nothing is lost to cancellation
//     (the global-coordinate version of the same form would lose |x| / sigma).
//   * DEAD AND OUT-OF-REACH NODES NEVER ENTER THE TILE.  While a 256-node tile is loaded every thread looks at one
//     node: pi < eps / singular nodes (weight 0) are dropped, and so is a node whose pdf underflows for EVERY point
//     of the workgroup: exponent <= -kappa dist(box, mu)^2 with kappa = 1 / (2 lambda_max(Sigma)) and box = the
//     bounding box of the workgroup's points.  Terms skipped this way are terms the wave-uniform test below would
//     have skipped too (they are exactly 0 in float64), so q does not change by a bit.  The survivors are compacted
//     in node order (ballot + popcount), the inner loop runs over them without a test per node.
//   pair_count (optional): += (points of this workgroup) x (nodes that entered its tiles) -- the pairs actually evaluated.
constexpr double LL_SKIP = -750.0;       // exp(y) == 0 in float64 below this exponent (denormals end at -745.13)
constexpr double LL_CULL = 751.0;        // a node is out of reach when kappa dist^2 exceeds this (margin over LL_SKIP)
constexpr double LL_REL_DROP = 46.1;     // ln(1e20) + margin: see the relative reach test in tree_loglik_kernel
constexpr int LL_REL_MIN_NODES = 512;    // levels with fewer nodes skip the extra pass (measured: nothing to drop there)
struct TreeLoglikArgs {
    const double* xs; int64_t n; int64_t n_pad; const double* prep; int64_t lb; int n_level_nodes; int nodes_per_chunk;
    double* partial; double* block_q; unsigned int* ticket; double* q_out; const int* done; TreeStop stop;
    const int* flags; unsigned long long* pair_count; const double* exp2_tab;
    int64_t i_base = 0; int q_count = 0;     // forest: the cloud's first point (n = one past its last), its number of q shares
};
// (bx, by) of a (gx, gy) grid: blockIdx / gridDim in tree_loglik_kernel, a slice of a 1-d grid in tree_ll_estep_kernel
// LDS of one log-likelihood workgroup, in doubles: node tile, exp table, the waves' q / boxes / lref, the waves' counts (ints)
template <bool BIGTAB>
constexpr int tree_loglik_lds() { return LL_TILE * 10 + (BIGTAB ? EXP_TAB2_N : EXP_TAB_N) + (CH / 64) * (1 + 6 + 1) + (CH / 64) / 2; }
// FOREST (tree_batch.hip): the workgroup takes ALL node chunks of its point block one after the other and adds the chunk
// sums in chunk order, then forms the shares of q the serial build's finish kernel would (one per 256 points) -- the serial
// sums in the serial order, without the serial form's [chunk][point] round trip (a forest fills the chip without
// splitting the nodes); a node's parameters are requested only once its weight has turned out non-zero.
template <int PTS, bool BIGTAB, bool FOREST = false>
__device__ __forceinline__ void tree_loglik_body(const int bx, const int by, const int gx, const int gy,
                                                 const TreeLoglikArgs& a, double* __restrict__ smem) {
    const double* __restrict__ xs = a.xs;
    const int64_t n = a.n, n_pad = a.n_pad;
    const double* __restrict__ prep = a.prep;
    const int64_t lb = a.lb;
    const int n_level_nodes = a.n_level_nodes, nodes_per_chunk = a.nodes_per_chunk;
    double* __restrict__ partial = a.partial;
    double* __restrict__ block_q = a.block_q;
    unsigned int* __restrict__ ticket = a.ticket;
    double* __restrict__ q_out = a.q_out;
    const int* __restrict__ done = a.done;
    const TreeStop& stop = a.stop;
    const int* __restrict__ flags = a.flags;
    unsigned long long* __restrict__ pair_count = a.pair_count;
    const double* __restrict__ exp2_tab = a.exp2_tab;
    const int stop_flag = done ? *done : 0;                // (looked at below, once the other requests are on their way)
    constexpr int TAB_N = BIGTAB ? EXP_TAB2_N : EXP_TAB_N;
    double (*tile)[10] = reinterpret_cast<double (*)[10]>(smem);
    double* exp_tab = smem + LL_TILE * 10;
    double* shq = exp_tab + TAB_N;
    double (*shbox)[6] = reinterpret_cast<double (*)[6]>(shq + CH / 64);
    double* shl = shq + (CH / 64) * 7;
    int* wcnt = reinterpret_cast<int*>(shl + CH / 64);
    if (BIGTAB) exp_tab2_load(exp_tab, exp2_tab); else exp_tab_load(exp_tab);   // (the tile loop's first barrier covers it)
    const int fl = flags ? *flags : 0;
    const int w = wave_in_block(), lane = lane_id();
    // origin: the workgroup's first point
    const int64_t i_first = (FOREST ? a.i_base : (int64_t)0) + (int64_t)bx * PTS * CH;
    const int64_t i_c = i_first < n ? i_first : n - 1;
    const double c0 = xs[i_c], c1 = xs[n_pad + i_c], c2 = xs[2 * n_pad + i_c];
    int64_t i[PTS];
    bool active[PTS];
    double x0[PTS], x1[PTS], x2[PTS], tot[PTS];
    double lo0 = 0.0, lo1 = 0.0, lo2 = 0.0, hi0 = 0.0, hi1 = 0.0, hi2 = 0.0;    // the origin itself is in the box
    // (the stop flag, the form flag, the origin and the points are requested together: the launches of a small cloud are
    //  chains of trips to memory, ~1.5 us each, and every trip taken side by side instead of in turn is that much less)
    double r0[PTS], r1[PTS], r2[PTS];
#pragma unroll
    for (int p = 0; p < PTS; ++p) {
        i[p] = i_first + (int64_t)p * CH + threadIdx.x;
        active[p] = i[p] < n;
        const int64_t il = active[p] ? i[p] : i_c;
        r0[p] = xs[il]; r1[p] = xs[n_pad + il]; r2[p] = xs[2 * n_pad + il];
    }
    // Small clouds: the WEIGHTS of this thread's nodes in the first LL_WPRE tiles are requested with the points.  A level's
    // dead nodes (pi < eps: 3370 of C4's 4096 level-3 slots) then cost 8 bytes each instead of the 17 parameters the tile
    // loop asks for at once -- 70 KB per workgroup, 88 MB per launch of a kernel that lasts 15 us (C4: 2.32 -> 2.27 ms).
    // (Also tried for these instantiations and dropped: two nodes per step of the evaluation loop, four exponentials
    //  interleaved -- 2.29 ms: the loop is not what these launches wait for.)
    constexpr int LL_WPRE = (BIGTAB || FOREST) ? 0 : 2;
    int node_begin = by * nodes_per_chunk;
    int node_end = (node_begin + nodes_per_chunk < n_level_nodes) ? node_begin + nodes_per_chunk : n_level_nodes;
    double wpre[LL_WPRE > 0 ? LL_WPRE : 1];
#pragma unroll
    for (int t = 0; t < LL_WPRE; ++t) {
        const int nd = node_begin + t * LL_TILE + (int)threadIdx.x;
        wpre[t] = nd < node_end ? prep[PREP_N * (lb + nd) + 10] : 0.0;
    }
    if (stop_flag) return;
    const bool use_chol = !(fl & 1);                       // kernel-uniform
#pragma unroll
    for (int p = 0; p < PTS; ++p) {
        x0[p] = x1[p] = x2[p] = tot[p] = 0.0;              // inactive slots sit on the origin
        if (active[p]) {
            x0[p] = r0[p] - c0; x1[p] = r1[p] - c1; x2[p] = r2[p] - c2;
        }
        lo0 = fmin(lo0, x0[p]); hi0 = fmax(hi0, x0[p]);
        lo1 = fmin(lo1, x1[p]); hi1 = fmax(hi1, x1[p]);
        lo2 = fmin(lo2, x2[p]); hi2 = fmax(hi2, x2[p]);
    }
    {
        const double b0 = -wave_max_f64(-lo0), b1 = -wave_max_f64(-lo1), b2 = -wave_max_f64(-lo2);
        const double b3 = wave_max_f64(hi0), b4 = wave_max_f64(hi1), b5 = wave_max_f64(hi2);
        if (lane == 0) {
            shbox[w][0] = b0; shbox[w][1] = b1; shbox[w][2] = b2; shbox[w][3] = b3; shbox[w][4] = b4; shbox[w][5] = b5;
        }
    }
    __syncthreads();
    lo0 = fmin(fmin(shbox[0][0], shbox[1][0]), fmin(shbox[2][0], shbox[3][0]));
    lo1 = fmin(fmin(shbox[0][1], shbox[1][1]), fmin(shbox[2][1], shbox[3][1]));
    lo2 = fmin(fmin(shbox[0][2], shbox[1][2]), fmin(shbox[2][2], shbox[3][2]));
    hi0 = fmax(fmax(shbox[0][3], shbox[1][3]), fmax(shbox[2][3], shbox[3][3]));
    hi1 = fmax(fmax(shbox[0][4], shbox[1][4]), fmax(shbox[2][4], shbox[3][4]));
    hi2 = fmax(fmax(shbox[0][5], shbox[1][5]), fmax(shbox[2][5], shbox[3][5]));

    // ---- relative reach: a lower bound of log(sum_j w_j pdf_j(x)) that holds for EVERY point of the workgroup ----
    // For node j and any x in the box: log(w_j pdf_j(x)) >= log w_j - kappa'_j D_j^2 with D_j the distance from the mean
    // to the farthest corner; the largest of these over the level's nodes, lref, bounds every point's sum from below.
    // A node whose UPPER bound over the box, log w_j - kappa_j dist(box, mu_j)^2, is below lref - ln(1e20 n_nodes)
    // cannot contribute more than 1e-20 of any point's sum even together with every other node dropped this way:
    // four orders of magnitude below half an ulp of the sum, i.e. below what the ORDER of the additions already
    // decides.  (The absolute test alone keeps a node until its pdf underflows, 38 sigma away; this one lets go of it
    // ~10 sigma beyond the box.)  Every workgroup -- also the ones that take a chunk of the nodes -- looks at ALL of
    // the level's nodes here, so that every chunk uses the same lref.
    double lref = -INFINITY;
    const double rel_margin = LL_REL_DROP + log((double)n_level_nodes);
    // OPT-IN (HGMM_TREE_REL=1 -> bit 1 of the flags) and only in the large-cloud instantiation.  Measured: at 10^6
    // points it takes the evaluated pairs from 22 % to 18 % of the reference's, 5.19 -> 5.04 ms per build -- 3 %, for
    // which the default does not give up "q is bitwise the full sum's"; the 40 256-point build loses 0.2 ms to the
    // extra pass (its launches are chains of latencies, and a converged bunny tree has 26 / 44 / 34 live nodes per level).
    if (BIGTAB && n_level_nodes >= LL_REL_MIN_NODES && (fl & 2)) {
        double best = -INFINITY;
        for (int n0 = (int)threadIdx.x; n0 < n_level_nodes; n0 += 4 * CH) {
            double wl[4], k2[4], u0[4], u1[4], u2[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {                  // (all twenty loads of the round in flight together)
                const int nd = n0 + q * CH < n_level_nodes ? n0 + q * CH : n_level_nodes - 1;
                const double* pr = prep + PREP_N * (lb + nd);
                wl[q] = pr[10]; k2[q] = pr[PREP_KAPPA2]; u0[q] = pr[6]; u1[q] = pr[7]; u2[q] = pr[8];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (n0 + q * CH < n_level_nodes && wl[q] > 0.0 && k2[q] >= 0.0) {
                    const double m0 = u0[q] - c0, m1 = u1[q] - c1, m2 = u2[q] - c2;
                    const double f0 = fmax(m0 - lo0, hi0 - m0), f1 = fmax(m1 - lo1, hi1 - m1), f2 = fmax(m2 - lo2, hi2 - m2);
                    best = fmax(best, log(wl[q]) - k2[q] * (f0 * f0 + f1 * f1 + f2 * f2));
                }
            }
        }
        best = wave_max_f64(best);
        if (lane == 0) shl[w] = best;
        __syncthreads();
        lref = fmax(fmax(shl[0], shl[1]), fmax(shl[2], shl[3]));
    }

    int entered = 0;                                       // nodes that made it into this workgroup's tiles
    [[maybe_unused]] double totsum[PTS];                   // forest: the chunk sums added in chunk order (tree_loglik_finish_kernel)
#pragma unroll
    for (int p = 0; p < PTS; ++p) totsum[p] = 0.0;
    const int cy_end = FOREST ? gy : 1;
    for (int cy = 0; cy < cy_end; ++cy) {
    if constexpr (FOREST) {
        node_begin = cy * nodes_per_chunk;
        node_end = (node_begin + nodes_per_chunk < n_level_nodes) ? node_begin + nodes_per_chunk : n_level_nodes;
    }
    for (int base = node_begin; base < node_end; base += LL_TILE) {
        // ---- this thread's node of the tile: weight, reach test, parameters in workgroup coordinates ----
        const int node = base + (int)threadIdx.x;
        bool live = false;
        double v[10];
#pragma unroll
        for (int e = 0; e < 10; ++e) v[e] = 0.0;
        bool known_dead = false;                            // (weight already here and zero: nothing else is requested)
        if constexpr (LL_WPRE > 0) {
            const int ti = (base - node_begin) / LL_TILE;
            if (ti < LL_WPRE) known_dead = (ti == 0 ? wpre[0] : wpre[LL_WPRE - 1]) == 0.0;
        }
        if constexpr (FOREST) {
            if (node < node_end) known_dead = prep[PREP_N * (lb + node) + 10] == 0.0;
        }
        if (node < node_end && !known_dead) {
            const double* pr = prep + PREP_N * (lb + node);
            // (every field requested at once: two dependent rounds of loads are two trips to memory in a kernel
            //  that lasts a handful of them)
            const double wL = pr[10], kap = pr[PREP_KAPPA], u0 = pr[6], u1 = pr[7], u2 = pr[8];
            const int fo = use_chol ? PREP_R : 0;
            const double f0 = pr[fo], f1 = pr[fo + 1], f2 = pr[fo + 2], f3 = pr[fo + 3], f4 = pr[fo + 4], f5 = pr[fo + 5];
            if (wL != 0.0) {
                const double m0 = u0 - c0, m1 = u1 - c1, m2 = u2 - c2;              // mean relative to the origin
                const double g0 = fmax(fmax(lo0 - m0, m0 - hi0), 0.0), g1 = fmax(fmax(lo1 - m1, m1 - hi1), 0.0),
                             g2 = fmax(fmax(lo2 - m2, m2 - hi2), 0.0);
                const double d2 = g0 * g0 + g1 * g1 + g2 * g2;                      // squared distance box <-> mean
                live = !(kap * d2 > LL_CULL) && !(log(wL) - kap * d2 < lref - rel_margin);
                if (live) {
                    if (use_chol) {
                        const double r00 = f0, r01 = f1, r02 = f2, r11 = f3, r12 = f4, r22 = f5;
                        v[0] = r00; v[1] = r01; v[2] = r02; v[3] = r11; v[4] = r12; v[5] = r22;
                        v[6] = -fma(r02, m2, fma(r01, m1, r00 * m0));               // -b = -R (mu - c)
                        v[7] = -fma(r12, m2, r11 * m1);
                        v[8] = -(r22 * m2);
                    } else {
                        // -1/2 Sigma^-1: the symmetric form is then the (non-positive) exponent itself
                        v[0] = -0.5 * f0; v[1] = -0.5 * f1; v[2] = -0.5 * f2; v[3] = -0.5 * f3;
                        v[4] = -0.5 * f4; v[5] = -0.5 * f5;
                        v[6] = m0; v[7] = m1; v[8] = m2;
                    }
                    v[9] = wL;
                }
            }
        }
        const unsigned long long mask = __ballot(live);
        const int before = __popcll(mask & ((1ull << lane) - 1ull));
        // (wcnt of the previous tile was read before that tile's second barrier, which every wave has passed)
        if (lane == 0) wcnt[w] = __popcll(mask);
        __syncthreads();                                   // also: every wave is done with the previous tile
        int off = 0, cnt = 0;
#pragma unroll
        for (int ww = 0; ww < CH / 64; ++ww) {
            const int t = wcnt[ww];
            if (ww < w) off += t;
            cnt += t;
        }
        if (live) {
            double* dst = tile[off + before];
#pragma unroll
            for (int e = 0; e < 10; ++e) dst[e] = v[e];
        }
        __syncthreads();
        entered += cnt;
        for (int k = 0; k < cnt; ++k) {
            // (node parameters through the LDS tile: reading them with wave-uniform scalar loads
            //  instead was measured 60 % slower for the C4 build, 8.6 vs 5.2 ms)
            const double t0 = tile[k][0], t1 = tile[k][1], t2 = tile[k][2], t3 = tile[k][3], t4 = tile[k][4],
                         t5 = tile[k][5], t6 = tile[k][6], t7 = tile[k][7], t8 = tile[k][8], wL = tile[k][9];
            double yv[PTS];
            bool need = false;
#pragma unroll
            for (int p = 0; p < PTS; ++p) {
                if (use_chol) {
                    const double z0 = fma(t2, x2[p], fma(t1, x1[p], fma(t0, x0[p], t6)));
                    const double z1 = fma(t4, x2[p], fma(t3, x1[p], t7));
                    const double z2 = fma(t5, x2[p], t8);
                    yv[p] = -fma(z2, z2, fma(z1, z1, z0 * z0));                    // = -q / 2
                } else {
                    yv[p] = sym3_quad(t0, t1, t2, t3, t4, t5, x0[p] - t6, x1[p] - t7, x2[p] - t8);
                }
                need = need || (yv[p] > LL_SKIP);
            }
            // exp(-q / 2) underflows to exactly 0 in float64 beyond q ~ 1490: skip the exponentials when no lane
            // of the wave needs one (points are sorted spatially); otherwise all PTS of them go through the
            // branch-free interleaved exp (an argument below the range comes back as < 3.3e-308: nothing)
            if (__any(need)) {
                if constexpr (PTS == 4) {
                    double e[4];
                    if constexpr (BIGTAB) {
                        if (use_chol) exp_t11_4<true>(yv, e, exp_tab); else exp_t11_4<false>(yv, e, exp_tab);
                    } else {
                        exp_nonpos4(yv, e, exp_tab);
                    }
#pragma unroll
                    for (int p = 0; p < 4; ++p) tot[p] = fma(wL, e[p], tot[p]);
                } else if constexpr (PTS == 2) {
                    double e[2];
                    exp_nonpos2(yv, e, exp_tab);
                    tot[0] = fma(wL, e[0], tot[0]);
                    tot[1] = fma(wL, e[1], tot[1]);
                } else {
                    const double y2[2] = {yv[0], yv[0]};
                    double e[2];
                    exp_nonpos2(y2, e, exp_tab);
                    tot[0] = fma(wL, e[0], tot[0]);
                }
            }
        }
    }
    if constexpr (FOREST) {
        if (gy > 1) {
#pragma unroll
            for (int p = 0; p < PTS; ++p) { totsum[p] += tot[p]; tot[p] = 0.0; }
        }
    }
    }
    if constexpr (FOREST) {
        if (gy > 1) {
            // the serial build's tree_loglik_finish_kernel: one share per 256 points = per p of this workgroup
            __syncthreads();
#pragma unroll
            for (int p = 0; p < PTS; ++p) {
                double lq = active[p] ? log(fmax(totsum[p], TREE_EPS)) : 0.0;
                lq = wave_sum_f64(lq);
                if (p > 0) __syncthreads();                // the previous share has been summed
                if (lane_id() == 0) shq[wave_in_block()] = lq;
                __syncthreads();
                double t = 0.0;
                for (int ww = 0; ww < CH / 64; ++ww) t += shq[ww];
                const int share = bx * PTS + p;
                if (threadIdx.x == 0 && share < a.q_count) block_q[share] = t;
            }
            return;
        }
    }
    if (pair_count && threadIdx.x == 0) {
        const int64_t rest = n - i_first;
        const int64_t pts = rest <= 0 ? 0 : (rest < (int64_t)PTS * CH ? rest : (int64_t)PTS * CH);
        atomicAdd(pair_count, (unsigned long long)(pts * entered));
    }
    if (gy > 1) {
#pragma unroll
        for (int p = 0; p < PTS; ++p)
            if (active[p]) partial[(size_t)by * n_pad + i[p]] = tot[p];
        return;
    }
    double lq = 0.0;
#pragma unroll
    for (int p = 0; p < PTS; ++p) lq += active[p] ? log(fmax(tot[p], TREE_EPS)) : 0.0;
    lq = wave_sum_f64(lq);
    __syncthreads();                                       // (shq is not aliased, but keep the tile loop's last readers behind)
    if (lane_id() == 0) shq[wave_in_block()] = lq;
    __syncthreads();
    double t = 0.0;
    for (int ww = 0; ww < CH / 64; ++ww) t += shq[ww];
    store_block_q(t, block_q, bx, gx, ticket, q_out, stop);
}

// ------------------------------------------------------------------------------------------
// The same level log-likelihood with the pdfs in FLOAT32 (hgmm_tree_set_precision(ctx, HGMM_PRECISION_F32_PDF): the type
// of the reference's GPU file, hgmm_gpu.py:472-484 -- float32 points, float32 node and moment arrays), any cloud size (round 6: small clouds and forests too).
//   * What stays float64: the points' coordinates relative to the workgroup's first point and every node's parameters in
//     those coordinates (R, -R (mu - c), weight) are formed in float64 exactly as above and only THEN rounded -- the
//     float32 numbers are of the size of the workgroup's extent over sigma, not of |x| over sigma; log(max(sum, eps))
//     per point and the sum q over the points are float64.
//   * What becomes float32: z = R d - b, y = -|z|^2 (R pre-scaled by sqrt(log2 e): 2^y is the pdf's exponential), 2^y by
//     v_exp_f32, sum_j w_j 2^y_j per point: 20 packed instructions (two points per v_pk_fma_f32) + 4 v_exp_f32 per node
//     for the thread's FOUR points, against 96 float64 instructions above.
//   * Range: the reference clamps the sum at eps = 1e-15 before the logarithm (logLikelihoodValue, C:83), so a term
//     below eps * 2^-30 / n_nodes cannot move a point's logarithm by 1e-9 -- nodes whose UPPER bound over the
//     workgroup's box, log w - kappa dist(box, mu)^2, is below that never enter the tile, and a wave skips the
//     exponentials of a node whose exponents are all below it: float32's exponent range (2^-126) is never approached.
//   * Conditioning: z = R x - R m in float32 carries 2^-23 of |R| x (extent of the workgroup about its origin), so a
//     term's error grows with extent / sigma -- a few units while a workgroup's points are neighbours (the usual case),
//     ~1000 when a parent holds tight, far-apart clusters.  The errors have random sign and q sums 10^5 ... 10^6 terms:
//     MEASURED on 16 clusters of sigma = 5e-4 scattered over a unit cube (tests/test_tree_gpu.py::
//     test_float32_pdf_mode_on_tight_far_apart_clusters) |dq| / |q| = 3e-8, |dq| = 0.09 against ls = 20.  A guarded
//     variant (nodes beyond |R| x extent = 64 evaluated from head + tail differences, errors relative to |x - mu|) was
//     built and measured: 8e-9 there, but 114 instead of 96 VGPRs and a 20 KB tile made the 10^6-point build 2.97 instead
//     of 2.41 ms -- removed again (profiles/r05/tree_f32_probe.log keeps both figures).
//   Accuracy of q against the float64 kernel, measured: |dq| = 0.02 ... 0.04 = 3-5e-8 per point on every cloud tried
//   (the uniform million, clustered clouds of 0.4 - 0.9 M points built to convergence at L = 1 ... 3, seeded random
//   shapes, the ill-conditioned case above; tests/test_tree_gpu.py) -- under 1 % of the smallest stop threshold in use
//   (ls = 5) and 0.03 % of the bench's (ls = 80).  The
//   E-step and the moments do NOT go through this kernel: as long as a level stops after the same number of iterations
//   the tree is the float64 tree bit for bit.
// ------------------------------------------------------------------------------------------
typedef float f2t __attribute__((ext_vector_type(2)));
typedef float f4t __attribute__((ext_vector_type(4)));
constexpr double LLF_SQRT_LOG2E = 1.2011224087864498;    // sqrt(log2 e): |sqrt(log2 e) z|^2 = log2(e) |z|^2
constexpr double LLF_LOG2E = 1.4426950408889634;
constexpr double LLF_REL_BITS = 30.0;                    // a dropped term is below eps * 2^-30 / n_nodes
__device__ __forceinline__ f2t llf_bc(float v) { return f2t{v, v}; }
__device__ __forceinline__ f2t llf_fma(f2t a, f2t b, f2t c) { return __builtin_elementwise_fma(a, b, c); }
// float32 node parameters stay finite AND leave room for z = R x - b and |z|^2: a node tighter than sigma ~ 1e-15 would
// overflow float32 (inf - inf = NaN in the z sums, and a NaN exponent silently drops the node from a point's sum); clamped,
// its exponent is a huge negative number for every point float32 can tell from the mean, i.e. its pdf is 0 there
__device__ __forceinline__ float llf_f32(double v) { return (float)fmax(fmin(v, 1.0e15), -1.0e15); }

// LDS of one float32 log-likelihood workgroup, in doubles: node tile (three float4 per node), the waves' q, boxes, counts
constexpr int tree_loglik_f32_lds() { return LL_TILE * 3 * 2 + (CH / 64) * (1 + 6) + (CH / 64) / 2; }
// PTS = 4 (clouds of >= 400 000 points) or 2 (small clouds); FOREST as in tree_loglik_body: the workgroup takes ALL node
// chunks of its point block one after the other and forms the serial build's shares of q (the chunk sums -- float32 sums
// widened to float64, exactly what the serial form parks in `partial` -- added in chunk order).
template <int PTS, bool FOREST>
__device__ __forceinline__ void tree_loglik_f32_body(const int bx, const int by, const int gx, const int gy,
                                                     const TreeLoglikArgs& a, double* __restrict__ smem) {
    static_assert(PTS == 2 || PTS == 4, "points go through the packed instructions two at a time");
    static_assert(CH / 64 == 4 && LL_TILE == CH, "the box / count reductions below are written for four waves and one node per thread");
    constexpr int NH = PTS / 2;
    f4t* tile = reinterpret_cast<f4t*>(smem);              // per node: (r00 r01 r02 r11) (r12 r22 -b0 -b1) (-b2 w yskip 0)
    double* shq = smem + LL_TILE * 3 * 2;
    double (*shbox)[6] = reinterpret_cast<double (*)[6]>(shq + CH / 64);
    int* wcnt = reinterpret_cast<int*>(shq + (CH / 64) * 7);
    const double* __restrict__ xs = a.xs;
    const int64_t n = a.n, n_pad = a.n_pad;
    const double* __restrict__ prep = a.prep;
    const int64_t lb = a.lb;
    const int n_level_nodes = a.n_level_nodes, nodes_per_chunk = a.nodes_per_chunk;
    const int stop_flag = a.done ? *a.done : 0;
    const int fl = a.flags ? *a.flags : 0;
    const int w = wave_in_block(), lane = lane_id();
    const int64_t i_first = (FOREST ? a.i_base : (int64_t)0) + (int64_t)bx * PTS * CH;
    const int64_t i_c = i_first < n ? i_first : n - 1;
    const double c0 = xs[i_c], c1 = xs[n_pad + i_c], c2 = xs[2 * n_pad + i_c];
    int64_t i[PTS];
    bool active[PTS];
    double r0[PTS], r1[PTS], r2[PTS];
#pragma unroll
    for (int p = 0; p < PTS; ++p) {
        i[p] = i_first + (int64_t)p * CH + threadIdx.x;
        active[p] = i[p] < n;
        const int64_t il = active[p] ? i[p] : i_c;
        r0[p] = xs[il]; r1[p] = xs[n_pad + il]; r2[p] = xs[2 * n_pad + il];
    }
    if (stop_flag) return;
    const bool use_chol = !(fl & 1);                       // kernel-uniform
    double lo0 = 0.0, lo1 = 0.0, lo2 = 0.0, hi0 = 0.0, hi1 = 0.0, hi2 = 0.0;
    float xf0[PTS], xf1[PTS], xf2[PTS];
#pragma unroll
    for (int p = 0; p < PTS; ++p) {
        double d0 = 0.0, d1 = 0.0, d2 = 0.0;               // inactive slots sit on the origin
        if (active[p]) { d0 = r0[p] - c0; d1 = r1[p] - c1; d2 = r2[p] - c2; }
        lo0 = fmin(lo0, d0); hi0 = fmax(hi0, d0);
        lo1 = fmin(lo1, d1); hi1 = fmax(hi1, d1);
        lo2 = fmin(lo2, d2); hi2 = fmax(hi2, d2);
        xf0[p] = (float)d0; xf1[p] = (float)d1; xf2[p] = (float)d2;
    }
    {
        const double b0 = -wave_max_f64(-lo0), b1 = -wave_max_f64(-lo1), b2 = -wave_max_f64(-lo2);
        const double b3 = wave_max_f64(hi0), b4 = wave_max_f64(hi1), b5 = wave_max_f64(hi2);
        if (lane == 0) {
            shbox[w][0] = b0; shbox[w][1] = b1; shbox[w][2] = b2; shbox[w][3] = b3; shbox[w][4] = b4; shbox[w][5] = b5;
        }
    }
    __syncthreads();
    lo0 = fmin(fmin(shbox[0][0], shbox[1][0]), fmin(shbox[2][0], shbox[3][0]));
    lo1 = fmin(fmin(shbox[0][1], shbox[1][1]), fmin(shbox[2][1], shbox[3][1]));
    lo2 = fmin(fmin(shbox[0][2], shbox[1][2]), fmin(shbox[2][2], shbox[3][2]));
    hi0 = fmax(fmax(shbox[0][3], shbox[1][3]), fmax(shbox[2][3], shbox[3][3]));
    hi1 = fmax(fmax(shbox[0][4], shbox[1][4]), fmax(shbox[2][4], shbox[3][4]));
    hi2 = fmax(fmax(shbox[0][5], shbox[1][5]), fmax(shbox[2][5], shbox[3][5]));
    // a term below this (natural log) cannot move any point's log(max(sum, eps)) by 2^-30
    // (a tree level has 8^(l+1) nodes: the logarithm of a power of two needs no log)
    const double ln_nodes = (n_level_nodes & (n_level_nodes - 1)) == 0 ? 0.6931471805599453 * (double)(31 - __clz(n_level_nodes))
                                                                       : log((double)n_level_nodes);
    const double abs_floor = LN_TREE_EPS - LLF_REL_BITS * 0.6931471805599453 - ln_nodes;
    // (A RELATIVE floor -- lref - 30 ln 2 - ln n with lref = max_j [log w_j - kappa'_j D_j^2], a lower bound of every point's
    //  sum over the workgroup's box -- was built and measured in round 6: on the scans' thin, surface-like Gaussians
    //  kappa' D^2 over a 512-point box runs into the hundreds, lref never rises above the absolute floor, and the extra pass
    //  over the level's nodes only cost: 8.41e7 -> 8.43e7 VALU instructions per level-2 launch of 32 bunny scans.)
    f2t X0[NH], X1[NH], X2[NH], TOT[NH];
#pragma unroll
    for (int h = 0; h < NH; ++h) {
        X0[h] = f2t{xf0[2 * h], xf0[2 * h + 1]};
        X1[h] = f2t{xf1[2 * h], xf1[2 * h + 1]};
        X2[h] = f2t{xf2[2 * h], xf2[2 * h + 1]};
        TOT[h] = f2t{0.f, 0.f};
    }
    [[maybe_unused]] double totsum[PTS];                   // forest: the chunk sums added in chunk order (tree_loglik_finish_kernel)
#pragma unroll
    for (int p = 0; p < PTS; ++p) totsum[p] = 0.0;

    int node_begin = by * nodes_per_chunk;
    int node_end = (node_begin + nodes_per_chunk < n_level_nodes) ? node_begin + nodes_per_chunk : n_level_nodes;
    int entered = 0;
    const int cy_end = FOREST ? gy : 1;
    for (int cy = 0; cy < cy_end; ++cy) {
    if constexpr (FOREST) {
        node_begin = cy * nodes_per_chunk;
        node_end = (node_begin + nodes_per_chunk < n_level_nodes) ? node_begin + nodes_per_chunk : n_level_nodes;
    }
    for (int base = node_begin; base < node_end; base += LL_TILE) {
        const int node = base + (int)threadIdx.x;
        bool live = false;
        f4t va = f4t{0.f, 0.f, 0.f, 0.f}, vb = va, vc = va;
        bool known_dead = false;                            // (forest: a node's parameters are requested once its weight is known)
        if constexpr (FOREST) {
            if (node < node_end) known_dead = prep[PREP_N * (lb + node) + 10] == 0.0;
        }
        if (node < node_end && !known_dead) {
            const double* pr = prep + PREP_N * (lb + node);
            const double wL = pr[10], kap = pr[PREP_KAPPA], u0 = pr[6], u1 = pr[7], u2 = pr[8];
            const int fo = use_chol ? PREP_R : 0;
            const double f0 = pr[fo], f1 = pr[fo + 1], f2 = pr[fo + 2], f3 = pr[fo + 3], f4 = pr[fo + 4], f5 = pr[fo + 5];
            if (wL != 0.0) {
                const double m0 = u0 - c0, m1 = u1 - c1, m2 = u2 - c2;
                const double g0 = fmax(fmax(lo0 - m0, m0 - hi0), 0.0), g1 = fmax(fmax(lo1 - m1, m1 - hi1), 0.0),
                             g2 = fmax(fmax(lo2 - m2, m2 - hi2), 0.0);
                const double d2 = g0 * g0 + g1 * g1 + g2 * g2;
                const double lw = log_pos_f64(wL);
                live = !(lw - kap * d2 < abs_floor);
                if (live) {
                    const float ysk = (float)((abs_floor - lw) * LLF_LOG2E);            // skip threshold of 2^y, log2 units
                    if (use_chol) {
                        const double S = LLF_SQRT_LOG2E;
                        va = f4t{llf_f32(S * f0), llf_f32(S * f1), llf_f32(S * f2), llf_f32(S * f3)};
                        vb = f4t{llf_f32(S * f4), llf_f32(S * f5), llf_f32(-S * fma(f2, m2, fma(f1, m1, f0 * m0))),
                                 llf_f32(-S * fma(f4, m2, f3 * m1))};
                        vc = f4t{llf_f32(-S * (f5 * m2)), (float)wL, ysk, 0.f};
                    } else {
                        // symmetric form: (-log2(e) / 2) Sigma^-1 and the mean, one float32 quadratic form per point
                        const double H = -0.5 * LLF_LOG2E;
                        va = f4t{llf_f32(H * f0), llf_f32(H * f1), llf_f32(H * f2), llf_f32(H * f3)};
                        vb = f4t{llf_f32(H * f4), llf_f32(H * f5), (float)m0, (float)m1};
                        vc = f4t{(float)m2, (float)wL, ysk, 0.f};
                    }
                }
            }
        }
        const unsigned long long mask = __ballot(live);
        const int before = __popcll(mask & ((1ull << lane) - 1ull));
        if (lane == 0) wcnt[w] = __popcll(mask);
        __syncthreads();                                   // also: every wave is done with the previous tile
        int off = 0, cnt = 0;
#pragma unroll
        for (int ww = 0; ww < CH / 64; ++ww) {
            const int t = wcnt[ww];
            if (ww < w) off += t;
            cnt += t;
        }
        if (live) {
            f4t* dst = tile + 3 * (off + before);
            dst[0] = va; dst[1] = vb; dst[2] = vc;
        }
        __syncthreads();
        entered += cnt;
        for (int k = 0; k < cnt; ++k) {
            const f4t ta = tile[3 * k], tb = tile[3 * k + 1], tc = tile[3 * k + 2];
            f2t y[NH];
            if (use_chol) {
#pragma unroll
                for (int h = 0; h < NH; ++h) {
                    const f2t z0 = llf_fma(llf_bc(ta.z), X2[h], llf_fma(llf_bc(ta.y), X1[h], llf_fma(llf_bc(ta.x), X0[h], llf_bc(tb.z))));
                    const f2t z1 = llf_fma(llf_bc(tb.x), X2[h], llf_fma(llf_bc(ta.w), X1[h], llf_bc(tb.w)));
                    const f2t z2 = llf_fma(llf_bc(tb.y), X2[h], llf_bc(tc.x));
                    f2t t = -(z0 * z0);
                    t = llf_fma(-z1, z1, t);
                    y[h] = llf_fma(-z2, z2, t);
                }
            } else {
#pragma unroll
                for (int h = 0; h < NH; ++h) {
                    const f2t d0 = X0[h] - llf_bc(tb.z), d1 = X1[h] - llf_bc(tb.w), d2 = X2[h] - llf_bc(tc.x);
                    const f2t t0 = llf_fma(llf_bc(2.f), llf_fma(llf_bc(ta.z), d2, llf_bc(ta.y) * d1), llf_bc(ta.x) * d0);
                    const f2t t1 = llf_fma(llf_bc(2.f), llf_bc(tb.x) * d2, llf_bc(ta.w) * d1);
                    y[h] = llf_fma(d2, llf_bc(tb.y) * d2, llf_fma(d1, t1, d0 * t0));
                }
            }
            float ymax = fmaxf(y[0].x, y[0].y);
            if constexpr (NH == 2) ymax = fmaxf(ymax, fmaxf(y[1].x, y[1].y));
            if (__any(ymax > tc.z)) {
#pragma unroll
                for (int h = 0; h < NH; ++h) {
                    const f2t e = f2t{__builtin_amdgcn_exp2f(y[h].x), __builtin_amdgcn_exp2f(y[h].y)};
                    TOT[h] = llf_fma(llf_bc(tc.y), e, TOT[h]);
                }
            }
        }
    }
    if constexpr (FOREST) {
        if (gy > 1) {
#pragma unroll
            for (int h = 0; h < NH; ++h) {
                totsum[2 * h] += (double)TOT[h].x;
                totsum[2 * h + 1] += (double)TOT[h].y;
                TOT[h] = f2t{0.f, 0.f};
            }
        }
    }
    }
    if constexpr (FOREST) {
        if (gy > 1) {
            // the serial build's tree_loglik_finish_kernel: one share per 256 points = per p of this workgroup
            __syncthreads();
#pragma unroll
            for (int p = 0; p < PTS; ++p) {
                double lq = active[p] ? log_pos_f64(fmax(totsum[p], TREE_EPS)) : 0.0;
                lq = wave_sum_f64(lq);
                if (p > 0) __syncthreads();                // the previous share has been summed
                if (lane_id() == 0) shq[wave_in_block()] = lq;
                __syncthreads();
                double t = 0.0;
                for (int ww = 0; ww < CH / 64; ++ww) t += shq[ww];
                const int share = bx * PTS + p;
                if (threadIdx.x == 0 && share < a.q_count) a.block_q[share] = t;
            }
            return;
        }
    }
    if (a.pair_count && threadIdx.x == 0) {
        const int64_t rest = n - i_first;
        const int64_t pts = rest <= 0 ? 0 : (rest < (int64_t)PTS * CH ? rest : (int64_t)PTS * CH);
        atomicAdd(a.pair_count, (unsigned long long)(pts * entered));
    }
    float tot[PTS];
#pragma unroll
    for (int h = 0; h < NH; ++h) { tot[2 * h] = TOT[h].x; tot[2 * h + 1] = TOT[h].y; }
    if (gy > 1) {
#pragma unroll
        for (int p = 0; p < PTS; ++p)
            if (active[p]) a.partial[(size_t)by * n_pad + i[p]] = (double)tot[p];
        return;
    }
    double lq = 0.0;
#pragma unroll
    for (int p = 0; p < PTS; ++p) lq += active[p] ? log_pos_f64(fmax((double)tot[p], TREE_EPS)) : 0.0;
    lq = wave_sum_f64(lq);
    __syncthreads();
    if (lane_id() == 0) shq[wave_in_block()] = lq;
    __syncthreads();
    double t = 0.0;
    for (int ww = 0; ww < CH / 64; ++ww) t += shq[ww];
    store_block_q(t, a.block_q, bx, gx, a.ticket, a.q_out, a.stop);
}

struct OpAddInt { __device__ __forceinline__ int operator()(int a, int b) const { return a + b; } };
// inclusive prefix sum of one int per lane over the wave (DPP row_shr with zero fill + row broadcasts, no LDS)
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ int dpp_i32_zero(int v) {
    return __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xF, true);
}
__device__ __forceinline__ int wave_scan_i32(int v) {
    v += dpp_i32_zero<0x111>(v);                  // row_shr:1
    v += dpp_i32_zero<0x112>(v);                  // row_shr:2
    v += dpp_i32_zero<0x114>(v);                  // row_shr:4
    v += dpp_i32_zero<0x118>(v);                  // row_shr:8
    v += dpp_i32_zero<DPP_ROW_BCAST15, 0xA>(v);
    v += dpp_i32_zero<DPP_ROW_BCAST31, 0xC>(v);
    return v;
}

// ------------------------------------------------------------------------------------------
// registration E-step (gmmTreeRegESTep, hgmm_cupy_cpu_working.py:202-228)
// ------------------------------------------------------------------------------------------
struct Rigid { double r[9]; double t[3]; double s; };


// The moment accumulation of this E-step is the one place of the library where contributions from arbitrary
// workgroups meet in the same memory words (a target point may land in any node; the points are not grouped by
// node, and their tree paths change with every (R, t)).  Floating-point atomics would make the result depend on
// arrival order, so the sums are taken in 64-bit FIXED POINT, where addition is associative: every run, every
// grid shape and every sharding of the target gives bit-identical moments.
//   * moments are taken about the node's own mean and scaled by the extent D (a power of two >= any |x - mu|):
//     gamma (x - mu) / D and gamma (x - mu)(x - mu)^T / D^2 lie in [-1, 1], gamma in [0, 1];
//   * scale 2^F, F = 62 - ceil(log2(n + 1)): n terms cannot overflow, resolution 2^-F (n = 40 k: 1.4e-14,
//     n = 1 M: 2.3e-13, relative to D resp. D^2) -- below the 1e-12 the parity tests allow;
//   * the registration M-step needs exactly these centred quantities (s - mu = c1 / m0), see tree_reg_normal_kernel.
// NMQ = 10 (m0, c1[3], C2 unique[6]) for the API's full moment set, 4 (m0, c1) inside the registration loop.
template <int CTRL, int MASK>
__device__ __forceinline__ long long dpp_i64_or0(long long x) {           // lanes outside the row mask read 0
    int lo = (int)(x & 0xffffffffLL), hi = (int)(x >> 32);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, MASK, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, MASK, 0xF, true);
    return (long long)(((unsigned long long)(unsigned int)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ long long wave_sum_i64(long long v) {          // exact: integer adds, any order
    v += dpp_i64_or0<DPP_QUAD_XOR1, 0xF>(v);
    v += dpp_i64_or0<DPP_QUAD_XOR2, 0xF>(v);
    v += dpp_i64_or0<DPP_ROW_HALF_MIRROR, 0xF>(v);
    v += dpp_i64_or0<DPP_ROW_MIRROR, 0xF>(v);
    v += dpp_i64_or0<DPP_ROW_BCAST15, 0xA>(v);
    v += dpp_i64_or0<DPP_ROW_BCAST31, 0xC>(v);
    const int lo = __builtin_amdgcn_readlane((int)(v & 0xffffffffLL), 63);
    const int hi = __builtin_amdgcn_readlane((int)(v >> 32), 63);
    return (long long)(((unsigned long long)(unsigned int)hi << 32) | (unsigned int)lo);
}

// (i: the thread's target point in `tg`, alive: it exists; prep / momq: the node table and the sums of THIS tree)
constexpr int REG_LDS_NODES = 584;                       // levels 0..2 (8 + 64 + 512 nodes)
template <int NMQ>
__device__ __forceinline__ void tree_reg_estep_body(const int64_t i, bool alive, const double* __restrict__ tg,
                                                    int64_t n_pad, const Rigid& tf, const double* __restrict__ prep, int L,
                                                    double lambda_c, double inv_d, double fix_scale,
                                                    unsigned long long* __restrict__ momq,
                                                    unsigned long long* __restrict__ tab /* LDS [REG_LDS_NODES * NMQ] */) {
    const int lds_nodes = (int)(level_first(L < 3 ? L : 3));
    __shared__ double exp_tab[EXP_TAB_N];                  // the build's exponential (exp_nonpos4): 17 instructions per value
    exp_tab_load(exp_tab);
    for (int e = threadIdx.x; e < lds_nodes * NMQ; e += CH) tab[e] = 0ull;
    __syncthreads();
    double x0 = 0.0, x1 = 0.0, x2 = 0.0;
    if (alive) {
        const double a = tg[i], b = tg[n_pad + i], c = tg[2 * n_pad + i];
        // (explicit fused operations: the serial and the batched kernel must round alike whatever the compiler would pick)
        x0 = fma(tf.s, fma(tf.r[2], c, fma(tf.r[1], b, tf.r[0] * a)), tf.t[0]);
        x1 = fma(tf.s, fma(tf.r[5], c, fma(tf.r[4], b, tf.r[3] * a)), tf.t[1]);
        x2 = fma(tf.s, fma(tf.r[8], c, fma(tf.r[7], b, tf.r[6] * a)), tf.t[2]);
    }
    int64_t search = -1;
    for (int l = 0; l < L; ++l) {
        if (!__any(alive)) break;
        const int64_t j0 = 8 * (search + 1);
        double g[8];
        double den = 0.0;
        if (alive) {
            // (round 6: the build E-step's arithmetic -- two interleaved table exponentials of four instead of eight
            //  library calls -- and ONE division: only the largest responsibility is used, and g_k / den is monotone in g_k)
            double yv[8], wE[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const double* pr = prep + PREP_N * (j0 + k);
                const double d0 = x0 - pr[6], d1 = x1 - pr[7], d2 = x2 - pr[8];
                yv[k] = -0.5 * sym3_quad(pr[0], pr[1], pr[2], pr[3], pr[4], pr[5], d0, d1, d2);
                wE[k] = pr[9];
            }
            const double ya[4] = {yv[0], yv[1], yv[2], yv[3]}, yb[4] = {yv[4], yv[5], yv[6], yv[7]};
            double ea[4], eb[4];
            exp_nonpos4(ya, ea, exp_tab);
            exp_nonpos4(yb, eb, exp_tab);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const double ev = k < 4 ? ea[k & 3] : eb[k & 3];
                g[k] = (wE[k] == 0.0 || yv[k] < -745.0) ? 0.0 : wE[k] * ev;     // (below the range the library exp is exactly 0)
                den += g[k];
            }
        }
        double gs = 0.0;
        int64_t s = 0;
        bool contribute = false;
        if (alive) {
            const bool good = den > TREE_EPS;
            int am = 0;
            double gbest = -1.0;
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (g[k] > gbest) { gbest = g[k]; am = k; }                     // first maximum (C:187 argmax)
            if (!good) am = 0;                                                  // all responsibilities are 0: the first child
            const double best = good ? gbest / den : 0.0;
            s = j0 + am;
            search = s;
            if (prep[PREP_N * s + 11] <= lambda_c) {       // complexity(cov_s) <= lambda_c: stop
                alive = false;
            } else {
                gs = best;
                contribute = !(gs < TREE_EPS);
            }
        }
        // this lane's contribution in fixed point, about the node's mean, in units of D
        long long q[NMQ];
#pragma unroll
        for (int m = 0; m < NMQ; ++m) q[m] = 0;
        if (contribute) {
            const double* pr = prep + PREP_N * s;
            const double u0 = (x0 - pr[6]) * inv_d, u1 = (x1 - pr[7]) * inv_d, u2 = (x2 - pr[8]) * inv_d;
            const double gq = gs * fix_scale;
            q[0] = __double2ll_rn(gq);
            q[1] = __double2ll_rn(gq * u0); q[2] = __double2ll_rn(gq * u1); q[3] = __double2ll_rn(gq * u2);
            if (NMQ == 10) {
                q[4] = __double2ll_rn(gq * u0 * u0); q[5] = __double2ll_rn(gq * u0 * u1);
                q[6] = __double2ll_rn(gq * u0 * u2); q[7] = __double2ll_rn(gq * u1 * u1);
                q[8] = __double2ll_rn(gq * u1 * u2); q[9] = __double2ll_rn(gq * u2 * u2);
            }
        }
        // The upper levels (few nodes, every workgroup hits all of them) are summed in LDS first and flushed once
        // per workgroup; deeper nodes are spread thinly enough for direct atomics.  Integer adds commute, so
        // neither the LDS order nor the arrival order of the global atomics can change the totals.
        if (contribute) {
            if (s < lds_nodes) {
#pragma unroll
                for (int m = 0; m < NMQ; ++m) atomicAdd(tab + NMQ * s + m, (unsigned long long)q[m]);
            } else {
#pragma unroll
                for (int m = 0; m < NMQ; ++m) atomicAdd(momq + NMQ * s + m, (unsigned long long)q[m]);
            }
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < lds_nodes * NMQ; e += CH) {
        const unsigned long long v = tab[e];
        if (v != 0ull) atomicAdd(momq + e, v);
    }
}


// Normal equations of the registration M-step (GMMTree.maximization_step, hgmm_gpu.py:729-752).  The reference
// stacks, for every node i with m0_i >= float32 eps, the three rows  [ s_i x n_c | n_c ] x = n_c . (mu_i - s_i),
// n_c = the columns of V_i sqrt(m0_i / lambda_i)  (eigh of Sigma_i), s_i = m1_i / m0_i, and solves by lstsq.
// Since sum_c n_c n_c^T = m0_i Sigma_i^-1 =: W_i (no eigen-decomposition needed) the normal equations are
//   A^T A = sum_i P_i W_i P_i^T,  A^T b = sum_i P_i W_i d_i,  b^T b = sum_i d_i^T W_i d_i,   P_i = [ [s_i]_x ; I ],
//   d_i = mu_i - s_i = -c1_i / m0_i.
// out[28] = 21 upper-triangle entries of A^T A (row-major), 6 of A^T b, b^T b.  One workgroup, fixed order.
// Input: the fixed-point sums (m0, c1) [T][4] themselves (already all-reduced over the ranks); they are set back
// to zero here, so that the next iteration's E-step needs no separate clearing launch.
__device__ __forceinline__ void tree_reg_normal_body(unsigned long long* __restrict__ momq /*[T][4]*/,
                                                     double d_ext, double inv_scale,
                                                     const double* __restrict__ prep, int64_t T,
                                                     double* __restrict__ out, double* host_out,
                                                     unsigned long long* host_seq, unsigned long long seq) {
    double acc[28];
#pragma unroll
    for (int k = 0; k < 28; ++k) acc[k] = 0.0;
    for (int64_t j = threadIdx.x; j < T; j += 256) {
        const double z = (double)(long long)momq[4 * j] * inv_scale;
        const double c10 = (double)(long long)momq[4 * j + 1] * (d_ext * inv_scale);
        const double c11 = (double)(long long)momq[4 * j + 2] * (d_ext * inv_scale);
        const double c12 = (double)(long long)momq[4 * j + 3] * (d_ext * inv_scale);
        momq[4 * j] = momq[4 * j + 1] = momq[4 * j + 2] = momq[4 * j + 3] = 0ull;
        if (z < 1.1920928955078125e-07) continue;                  // np.finfo(np.float32).eps (hgmm_gpu.py:733)
        const double* pr = prep + PREP_N * j;
        const double w00 = z * pr[0], w01 = z * pr[1], w02 = z * pr[2], w11 = z * pr[3], w12 = z * pr[4], w22 = z * pr[5];
        const double iz = 1.0 / z;
        const double d0 = -c10 * iz, d1 = -c11 * iz, d2 = -c12 * iz;
        const double s0 = pr[6] - d0, s1 = pr[7] - d1, s2 = pr[8] - d2;
        const double W[3][3] = {{w00, w01, w02}, {w01, w11, w12}, {w02, w12, w22}};
        // SW = [s]_x W : column c of SW = s x W[:,c]
        double SW[3][3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            SW[0][c] = s1 * W[2][c] - s2 * W[1][c];
            SW[1][c] = s2 * W[0][c] - s0 * W[2][c];
            SW[2][c] = s0 * W[1][c] - s1 * W[0][c];
        }
        // SWS^T = SW [s]_x^T : row r of it = -(SW[r,:] x s) ... (SW S^T)[r][c] = sum_k SW[r][k] S[c][k]
        const double S[3][3] = {{0.0, -s2, s1}, {s2, 0.0, -s0}, {-s1, s0, 0.0}};
        double TL[3][3];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) TL[r][c] = SW[r][0] * S[c][0] + SW[r][1] * S[c][1] + SW[r][2] * S[c][2];
        const double Wd[3] = {W[0][0] * d0 + W[0][1] * d1 + W[0][2] * d2, W[1][0] * d0 + W[1][1] * d1 + W[1][2] * d2,
                              W[2][0] * d0 + W[2][1] * d1 + W[2][2] * d2};
        const double SWd[3] = {s1 * Wd[2] - s2 * Wd[1], s2 * Wd[0] - s0 * Wd[2], s0 * Wd[1] - s1 * Wd[0]};
        // upper triangle of the 6x6, row-major: rows 0-2 = [TL | SW], rows 3-5 = [. | W]
        int k = 0;
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int c = r; c < 6; ++c) {
                double v;
                if (r < 3 && c < 3) v = TL[r][c];
                else if (r < 3) v = SW[r][c - 3];
                else v = W[r - 3][c - 3];
                acc[k++] += v;
            }
#pragma unroll
        for (int r = 0; r < 3; ++r) { acc[21 + r] += SWd[r]; acc[24 + r] += Wd[r]; }
        acc[27] += d0 * Wd[0] + d1 * Wd[1] + d2 * Wd[2];
    }
    __shared__ double sh[4][28];
#pragma unroll
    for (int k = 0; k < 28; ++k) {
        const double v = wave_sum_f64(acc[k]);
        if (lane_id() == 0) sh[wave_in_block()][k] = v;
    }
    __syncthreads();
    // host_out / host_seq: coherent pinned HOST memory -- the 28 numbers, then (behind a system-scope release) the
    // sequence number the host is polling for: the registration loop's one hand-over per iteration without a copy
    // packet and a stream synchronisation
    if (threadIdx.x < 64) {                                  // wave 0
        if (threadIdx.x < 28) {
            const double v = sh[0][threadIdx.x] + sh[1][threadIdx.x] + sh[2][threadIdx.x] + sh[3][threadIdx.x];
            out[threadIdx.x] = v;
            if (host_out) host_out[threadIdx.x] = v;
        }
        if (host_out) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
            if (threadIdx.x == 0) __hip_atomic_store(host_seq, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}


// ---- the registration loop (hgmm_gpu.py:754-768) with its 6 x 6 M-step on the host side of this library ----------
// eigenvalue range of a symmetric 6 x 6 matrix by cyclic Jacobi sweeps (for the conditioning test only)
inline void sym6_eig_range(const double (&A)[6][6], double* lo, double* hi) {
    double a[6][6];
    for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) a[i][j] = A[i][j];
    for (int sweep = 0; sweep < 30; ++sweep) {
        double off = 0.0, diag = 0.0;
        for (int i = 0; i < 6; ++i) {
            diag += a[i][i] * a[i][i];
            for (int j = i + 1; j < 6; ++j) off += a[i][j] * a[i][j];
        }
        // (the range feeds a test against 1e-11 of the largest eigenvalue: off-diagonal mass below 1e-32 of the diagonal's
        //  moves no eigenvalue by 1e-16 of that scale.  Round 5 iterated to off == 0.0 -- a dozen more sweeps of nothing,
        //  ~10 us of host time per pair and iteration, which is what bounds a BATCH of registrations)
        if (off <= 1.0e-32 * diag) break;
        for (int p = 0; p < 6; ++p)
            for (int q = p + 1; q < 6; ++q) {
                if (a[p][q] == 0.0) continue;
                const double theta = (a[q][q] - a[p][p]) / (2.0 * a[p][q]);
                const double tt = (theta >= 0.0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                const double cs = 1.0 / std::sqrt(tt * tt + 1.0), sn = tt * cs;
                for (int k = 0; k < 6; ++k) {
                    const double akp = a[k][p], akq = a[k][q];
                    a[k][p] = cs * akp - sn * akq;
                    a[k][q] = sn * akp + cs * akq;
                }
                for (int k = 0; k < 6; ++k) {
                    const double apk = a[p][k], aqk = a[q][k];
                    a[p][k] = cs * apk - sn * aqk;
                    a[q][k] = sn * apk + cs * aqk;
                }
            }
    }
    *lo = *hi = a[0][0];
    for (int i = 1; i < 6; ++i) { *lo = std::min(*lo, a[i][i]); *hi = std::max(*hi, a[i][i]); }
}
// A x = b by Gaussian elimination with partial pivoting (what LAPACK's gesv does); false: singular
inline bool solve6(const double (&A)[6][6], const double (&b)[6], double (&x)[6]) {
    double m[6][7];
    for (int i = 0; i < 6; ++i) { for (int j = 0; j < 6; ++j) m[i][j] = A[i][j]; m[i][6] = b[i]; }
    for (int col = 0; col < 6; ++col) {
        int piv = col;
        for (int r = col + 1; r < 6; ++r) if (std::fabs(m[r][col]) > std::fabs(m[piv][col])) piv = r;
        if (m[piv][col] == 0.0) return false;
        if (piv != col) for (int j = 0; j < 7; ++j) std::swap(m[piv][j], m[col][j]);
        for (int r = col + 1; r < 6; ++r) {
            const double f = m[r][col] / m[col][col];
            for (int j = col; j < 7; ++j) m[r][j] -= f * m[col][j];
        }
    }
    for (int i = 5; i >= 0; --i) {
        double s = m[i][6];
        for (int j = i + 1; j < 6; ++j) s -= m[i][j] * x[j];
        x[i] = s / m[i][i];
    }
    return true;
}
// (rot, t) <- (dR rot, dR t + v), dR = exp([omega]_x) by Rodrigues' formula   (twist_mul, hgmm_gpu.py:634-664)
inline void twist_compose(const double (&x)[6], double* rot, double* t) {
    const double w0 = x[0], w1 = x[1], w2 = x[2];
    const double angle = std::sqrt(w0 * w0 + w1 * w1 + w2 * w2);
    double d[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    if (angle != 0.0) {
        const double a = w0 / angle, b = w1 / angle, c = w2 / angle;
        const double k[3][3] = {{0.0, -c, b}, {c, 0.0, -a}, {-b, a, 0.0}};
        const double sn = std::sin(angle), oc = 1.0 - std::cos(angle);
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                double kk = 0.0;
                for (int l = 0; l < 3; ++l) kk += k[i][l] * k[l][j];
                d[i][j] += sn * k[i][j] + oc * kk;
            }
    }
    double r2[9], t2[3];
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) {
            double v = 0.0;
            for (int l = 0; l < 3; ++l) v += d[i][l] * rot[3 * l + j];
            r2[3 * i + j] = v;
        }
        double v = 0.0;
        for (int l = 0; l < 3; ++l) v += d[i][l] * t[l];
        t2[i] = v + x[3 + i];
    }
    for (int i = 0; i < 9; ++i) rot[i] = r2[i];
    for (int i = 0; i < 3; ++i) t[i] = t2[i];
}

// The host side of ONE registration iteration, shared by hgmm_tree_register and hgmm_tree_register_batch: o[28] = the
// normal equations the device produced (tree_reg_normal_kernel).  -> 0: (rot, t) updated, go on; 1: updated and stopped by
// |q - q_prev| < tol; 2: too ill-conditioned for normal equations (nothing updated): the caller takes the reference's
// stacked least-squares M-step.  *q_out = this iteration's q (status 0 / 1).
inline int reg_host_step(const double* o, double* rot, double* t, double* q_prev_inout, double tol, double* q_out) {
    double A[6][6], b[6], x[6];
    bool finite = true;
    for (int i = 0, k = 0; i < 6; ++i)
        for (int j = i; j < 6; ++j, ++k) { A[i][j] = A[j][i] = o[k]; finite = finite && std::isfinite(o[k]); }
    for (int i = 0; i < 6; ++i) { b[i] = o[21 + i]; finite = finite && std::isfinite(b[i]); }
    // The conditioning rule is lambda_min <= 1e-11 lambda_max.  A cheap certificate of the opposite first: with A = L L^T,
    // lambda_min >= 1 / ||A^-1||_inf and lambda_max <= ||A||_inf, so 1 / ||A^-1||_inf > 1e-11 ||A||_inf settles it without
    // the eigenvalues (~0.2 us instead of the Jacobi sweeps' ~2 us per pair and iteration -- the host's share of a batched
    // registration); only a system that fails the certificate gets the exact test.  Same verdicts, same solve.
    bool well = false;
    if (finite) {
        double Lm[6][6], Li[6][6];
        bool spd = true;
        for (int j = 0; j < 6 && spd; ++j) {
            double sdiag = A[j][j];
            for (int k = 0; k < j; ++k) sdiag -= Lm[j][k] * Lm[j][k];
            if (!(sdiag > 0.0)) { spd = false; break; }
            Lm[j][j] = std::sqrt(sdiag);
            for (int i = j + 1; i < 6; ++i) {
                double v = A[i][j];
                for (int k = 0; k < j; ++k) v -= Lm[i][k] * Lm[j][k];
                Lm[i][j] = v / Lm[j][j];
            }
        }
        if (spd) {
            for (int c2 = 0; c2 < 6; ++c2)                         // Li = L^-1 (lower triangular), column by column
                for (int i = 0; i < 6; ++i) {
                    if (i < c2) { Li[i][c2] = 0.0; continue; }
                    double v = (i == c2) ? 1.0 : 0.0;
                    for (int k = c2; k < i; ++k) v -= Lm[i][k] * Li[k][c2];
                    Li[i][c2] = v / Lm[i][i];
                }
            double ninv = 0.0, na = 0.0;
            for (int i = 0; i < 6; ++i) {
                double ri = 0.0, ra = 0.0;
                for (int j = 0; j < 6; ++j) {
                    double aij = 0.0;                              // (A^-1)_ij = sum_k Li[k][i] Li[k][j]
                    for (int k = (i > j ? i : j); k < 6; ++k) aij += Li[k][i] * Li[k][j];
                    ri += std::fabs(aij);
                    ra += std::fabs(A[i][j]);
                }
                ninv = std::max(ninv, ri);
                na = std::max(na, ra);
            }
            well = std::isfinite(ninv) && ninv > 0.0 && 1.0 / ninv > 1.0e-10 * na;      // (a decade of margin for its own rounding)
        }
    }
    double lo = 0.0, hi = 0.0;
    if (finite && !well) sym6_eig_range(A, &lo, &hi);
    if (!finite || (!well && (!(hi > 0.0) || lo <= 1e-11 * hi)) || !solve6(A, b, x)) return 2;
    double xb = 0.0;
    for (int i = 0; i < 6; ++i) xb += x[i] * b[i];
    const double q = std::max(o[27] - xb, 0.0);
    twist_compose(x, rot, t);
    *q_out = q;
    const double qp = *q_prev_inout;
    *q_prev_inout = q;
    return (qp == qp && std::fabs(q - qp) < tol) ? 1 : 0;                                   // (NaN: no previous q)
}

// the lanes' shares of one node's moments: chunk partials of its parent, lane t takes chunks c0 + t, c0 + t + 64, ...
// (tree_moments_kernel and its forest twin: the same order, then the same wave_sum_f64 per moment)
__device__ __forceinline__ void tree_moments_gather(const double* __restrict__ partials, int c0, int c1, int k,
                                                    double (&acc)[NMOM]) {
#pragma unroll
    for (int m = 0; m < NMOM; ++m) acc[m] = 0.0;
    for (int c = c0 + (int)threadIdx.x; c < c1; c += 64) {
        const double* src = partials + (size_t)c * (8 * NMOM) + k * NMOM;
#pragma unroll
        for (int m = 0; m < NMOM; ++m) acc[m] += src[m];
    }
}

// The same sums, bit for bit, for the EIGHT children of one parent in one wave (round 6: small clouds and forests, where
// a parent has a handful of chunks and a wave per child spent ~700 instructions -- ten 64-lane reductions, the stop rule,
// the M-step in one lane -- on sixteen numbers): lane 8 k + j works for child k.  tree_moments_kernel's wave adds, per
// moment, a_t = sum_i partial[c0 + t + 64 i] in lane t and then reduces the 64 lanes by the butterfly of wave_sum_f64:
// rows of eight lanes first (R_m, m = t / 8: quad xor 1, quad xor 2, half-row mirror), then
// ((R_7 + R_6) + (R_5 + R_4)) + ((R_3 + R_2) + (R_1 + R_0)).  Here the eight lanes of a child take the eight rows one
// after the other -- the same three steps inside the row, the same tree above it -- and a row no chunk falls into is the
// +0.0 it would have been.  Every lane of the child's group ends up with the child's ten totals.
__device__ __forceinline__ void tree_moments_gather8(const double* __restrict__ partials, int c0, int c1,
                                                     double (&tot)[NMOM]) {
    const int k = (int)threadIdx.x >> 3, j = (int)threadIdx.x & 7;
    auto row = [&](int m, double (&r)[NMOM]) {
#pragma unroll
        for (int q = 0; q < NMOM; ++q) r[q] = 0.0;
        if (c0 + 8 * m >= c1) return;                     // (wave-uniform: all of a wave's lanes work for the same parent)
        for (int c = c0 + 8 * m + j; c < c1; c += 64) {
            const double* src = partials + (size_t)c * (8 * NMOM) + k * NMOM;
#pragma unroll
            for (int q = 0; q < NMOM; ++q) r[q] += src[q];
        }
#pragma unroll
        for (int q = 0; q < NMOM; ++q) {
            r[q] += dpp_f64<DPP_QUAD_XOR1>(r[q]);
            r[q] += dpp_f64<DPP_QUAD_XOR2>(r[q]);
            r[q] += dpp_f64<DPP_ROW_HALF_MIRROR>(r[q]);
        }
    };
    double a[NMOM], b[NMOM], c[NMOM];
    row(0, a); row(1, b);
#pragma unroll
    for (int q = 0; q < NMOM; ++q) a[q] += b[q];          // S0 = R0 + R1
    row(2, b); row(3, c);
#pragma unroll
    for (int q = 0; q < NMOM; ++q) a[q] = (b[q] + c[q]) + a[q];   // S1 + S0
    row(4, b); row(5, c);
#pragma unroll
    for (int q = 0; q < NMOM; ++q) b[q] += c[q];          // S2
    double d[NMOM];
    row(6, c); row(7, d);
#pragma unroll
    for (int q = 0; q < NMOM; ++q) tot[q] = ((c[q] + d[q]) + b[q]) + a[q];   // (S3 + S2) + (S1 + S0)
}

// How hgmm_tree_build splits a level's nodes over gridDim.y of the log-likelihood launch (small clouds: not enough point
// blocks to fill the chip).  The split fixes the ORDER in which a point's node terms are added, so the forest build asks the
// same function for every cloud.  llblocks = point blocks of the launch.
inline void tree_ll_split(int llblocks, int n_level, int cus, int* chunks_out, int* per_chunk_out) {
    int chunks = 1;
    if (llblocks < 2 * cus && n_level > LL_TILE) {            // (N = 1e6: 977 workgroups are plenty -- no split, no finish pass)
        chunks = (4 * cus + llblocks - 1) / llblocks;
        const int max_chunks_l = (n_level + LL_TILE - 1) / LL_TILE;
        if (chunks > max_chunks_l) chunks = max_chunks_l;
    }
    const int per_chunk = ((n_level + chunks - 1) / chunks + LL_TILE - 1) / LL_TILE * LL_TILE;
    *per_chunk_out = per_chunk;
    *chunks_out = (n_level + per_chunk - 1) / per_chunk;
}

// largest |mu_j| of a node table (host)
inline double tree_mu_rmax(const double* mu, int64_t T) {
    double m2 = 0.0;
    for (int64_t j = 0; j < T; ++j) {
        const double v = mu[3 * j] * mu[3 * j] + mu[3 * j + 1] * mu[3 * j + 1] + mu[3 * j + 2] * mu[3 * j + 2];
        if (v > m2 && std::isfinite(v)) m2 = v;
    }
    return std::sqrt(m2);
}
// extent of the moved target about any node mean: |s R x + t - mu| <= |s| (Frobenius bound on R) max|x| + |t| + max|mu|
__host__ __device__ inline double reg_extent(const Rigid& tf, double tgt_rmax, double mu_rmax) {
    double rn = 0.0, tn = 0.0;
    for (int i = 0; i < 9; ++i) rn += tf.r[i] * tf.r[i];
    for (int i = 0; i < 3; ++i) tn += tf.t[i] * tf.t[i];
    return fabs(tf.s) * sqrt(rn) * tgt_rmax + sqrt(tn) + mu_rmax;
}
// encoding of the registration E-step's fixed-point sums: D = the extent rounded up to a power of two, F fractional bits
// such that n_all terms cannot overflow 62 bits
__host__ __device__ inline void reg_encoding(double ext, double n_all, double* D_out, int* F_out) {
    if (!(ext > 0.0) || !(ext < 1.0e300)) ext = 1.0;              // (not positive, NaN or infinite)
    int e2 = 0;
    (void)frexp(ext, &e2);                                        // ext < 2^e2
    *D_out = ldexp(1.0, e2);
    int nbits = 1;
    while (ldexp(1.0, nbits) <= n_all) ++nbits;
    *F_out = 62 - nbits;
}

// One pair of a batched registration (tree_batch.hip).  The first block is what the E-step and the normal-equation kernels
// read; the second is the loop's state when the device runs it alone (reg_device_solve, reg_device_step below).
struct ForestRegPair {
    Rigid tf;
    double inv_d, fix_scale, d_ext, inv_scale;
    int tg_first, tg_count;
    int active, pad;
    double q_prev, tg_rmax, mu_rmax;     // q of the previous iteration (valid with has_q), extent bounds of the encoding
    int has_q, it, status, pad2;         // iterations done; 0: running / budget used up, 1: |dq| < tol, 2: ill-conditioned
};

// The host side of a registration iteration (reg_host_step) done by ONE device thread (per-context option
// reg_device_solve, off by default: north_star keeps the rigid solve on the host, and the host path is the parity
// reference): o[28] = the normal equations; solves the 6 x 6 system, composes the twist, applies the stop rule and
// prepares the fixed-point encoding of the next E-step, all in the pair's table entry -- the next launch reads (R, t)
// from device memory and the host only follows a progress word.  Differences to the host path, both inside the 1e-8 the
// per-iteration trace is held to: Cholesky instead of elimination with pivoting, the device's sin / cos, fused
// multiply-adds; the conditioning test is on the Cholesky pivots (smallest pivot <= 1e-11 largest diagonal entry --
// implied by the host's eigenvalue test failing, not equivalent to it): such a pair stops with status 2 and the caller
// finishes it on the host path, as in the host loop.
__device__ inline void reg_device_step(const double* __restrict__ o, ForestRegPair* pr, double tol, int max_iter,
                                       double* __restrict__ trace_row) {
    double A[6][6], b[6], x[6], y[6];
    bool ok = true;
    double dmax = 0.0;
    for (int i = 0, k = 0; i < 6; ++i)
        for (int j = i; j < 6; ++j, ++k) { A[i][j] = A[j][i] = o[k]; ok = ok && (fabs(o[k]) < 1.0e300); }
    for (int i = 0; i < 6; ++i) { b[i] = o[21 + i]; ok = ok && (fabs(b[i]) < 1.0e300); dmax = fmax(dmax, A[i][i]); }
    double pmin = 1.0e300;
    for (int j = 0; j < 6 && ok; ++j) {                       // A = L L^T, L in the lower triangle of A
        double s = A[j][j];
        for (int k = 0; k < j; ++k) s -= A[j][k] * A[j][k];
        pmin = fmin(pmin, s);
        if (!(s > 0.0)) { ok = false; break; }
        const double d = sqrt(s);
        A[j][j] = d;
        for (int i = j + 1; i < 6; ++i) {
            double v = A[i][j];
            for (int k = 0; k < j; ++k) v -= A[i][k] * A[j][k];
            A[i][j] = v / d;
        }
    }
    if (!ok || !(dmax > 0.0) || pmin <= 1.0e-11 * dmax) {
        pr->status = 2;
        pr->active = 0;
        return;
    }
    for (int i = 0; i < 6; ++i) {
        double v = b[i];
        for (int k = 0; k < i; ++k) v -= A[i][k] * y[k];
        y[i] = v / A[i][i];
    }
    for (int i = 5; i >= 0; --i) {
        double v = y[i];
        for (int k = i + 1; k < 6; ++k) v -= A[k][i] * x[k];
        x[i] = v / A[i][i];
    }
    double xb = 0.0;
    for (int i = 0; i < 6; ++i) xb += x[i] * b[i];
    const double q = fmax(o[27] - xb, 0.0);
    // (rot, t) <- (dR rot, dR t + v), dR = exp([omega]_x)   (twist_compose)
    Rigid tf = pr->tf;
    {
        const double w0 = x[0], w1 = x[1], w2 = x[2];
        const double angle = sqrt(w0 * w0 + w1 * w1 + w2 * w2);
        double d[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
        if (angle != 0.0) {
            const double a = w0 / angle, bb = w1 / angle, c = w2 / angle;
            const double k[3][3] = {{0.0, -c, bb}, {c, 0.0, -a}, {-bb, a, 0.0}};
            const double sn = sin(angle), oc = 1.0 - cos(angle);
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j) {
                    double kk = 0.0;
                    for (int l = 0; l < 3; ++l) kk += k[i][l] * k[l][j];
                    d[i][j] += sn * k[i][j] + oc * kk;
                }
        }
        double r2[9], t2[3];
        for (int i = 0; i < 3; ++i) {
            for (int j = 0; j < 3; ++j) {
                double v = 0.0;
                for (int l = 0; l < 3; ++l) v += d[i][l] * tf.r[3 * l + j];
                r2[3 * i + j] = v;
            }
            double v = 0.0;
            for (int l = 0; l < 3; ++l) v += d[i][l] * tf.t[l];
            t2[i] = v + x[3 + i];
        }
        for (int i = 0; i < 9; ++i) tf.r[i] = r2[i];
        for (int i = 0; i < 3; ++i) tf.t[i] = t2[i];
    }
    pr->tf = tf;
    const int it = pr->it;
    if (trace_row) {
        for (int i = 0; i < 9; ++i) trace_row[i] = tf.r[i];
        for (int i = 0; i < 3; ++i) trace_row[9 + i] = tf.t[i];
        trace_row[12] = q;
    }
    const bool stop = pr->has_q && fabs(q - pr->q_prev) < tol;
    pr->q_prev = q;
    pr->has_q = 1;
    pr->it = it + 1;
    if (stop) pr->status = 1;
    if (stop || it + 1 >= max_iter) { pr->active = 0; return; }
    double D = 1.0;
    int F = 0;
    reg_encoding(reg_extent(tf, pr->tg_rmax, pr->mu_rmax), (double)pr->tg_count, &D, &F);
    pr->inv_d = 1.0 / D;
    pr->fix_scale = ldexp(1.0, F);
    pr->d_ext = D;
    pr->inv_scale = ldexp(1.0, -F);
}

// the registration loop on the device alone (tree_batch.hip; hgmm_tree_register uses it with B = 1 on the serial buffers)
int forest_register_on_device(::hgmm_ctx* c, int B, const double* tg, int64_t tg_pad, const int64_t* tg_first,
                              const int64_t* tg_counts, const double* tg_rmax, const double* mu_rmax, const double* prep, int T,
                              int L, unsigned long long* momq, double* rot, double* t, double scale, double lambda_c,
                              int max_iter, double tol, double* q_prev_inout, int32_t* iters_out, int32_t* status_out,
                              double* trace);

// ---- kernels defined in tree_kernels.hip that the batched path (tree_batch.hip) launches as they are -----------------
constexpr int OFF_BLOCK = 256;
__global__ void tree_prep_kernel(const double* __restrict__ pi, const double* __restrict__ mu,
                                 const double* __restrict__ cov, int64_t j_begin, int64_t j_end,
                                 double* __restrict__ prep, int* __restrict__ flags);
__global__ void tree_complexity_kernel(const double* __restrict__ cov, int64_t j_begin, int64_t j_end,
                                       double* __restrict__ prep);
__global__ void tree_init_nodes_kernel(const double* __restrict__ init_mu, double sig2, int64_t T,
                                       double* pi, double* mu, double* cov);
__global__ void tree_chunks_kernel(const int* __restrict__ seg_start, int P, int* __restrict__ chunk_first,
                                   int* __restrict__ chunk_desc, int* __restrict__ n_chunks_out);
__global__ __launch_bounds__(OFF_BLOCK) void tree_offsets_kernel(const int* __restrict__ hist,
                                                                 const int* __restrict__ chunk_first,
                                                                 const int* __restrict__ seg_start, int P,
                                                                 int* __restrict__ chunk_off /*[chunks][8]*/,
                                                                 int* __restrict__ new_seg_start /*[8P+1]*/);
__global__ void tree_iota_kernel(int* perm, int64_t n);
inline unsigned nblk(int64_t n, int b) { return (unsigned)((n + b - 1) / b); }

}  // namespace hgmm
