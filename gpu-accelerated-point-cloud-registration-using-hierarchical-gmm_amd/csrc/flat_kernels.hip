// Flat (non-hierarchical) GMM EM for gfx950: diag / spherical covariance, float32 arithmetic.
//
// Replaces the reference's array-library EM (src/python/gmm_waymo/src/gmm_impl.py:53-155 and
// src/python/gmmreg_gpu/gmm_impl.py:36-91): per EM iteration the reference runs 5 SGEMMs with
// K = 3 / K = N and ~10 elementwise passes over N x J temporaries.  Here:
//
//   flat_estep_kernel   one wavefront per point row, lanes across components.  The row's
//                       x is wave-uniform (scalar loads, prefetched one row ahead); each lane
//                       keeps the packed parameters (mu, 0.5 log2(e)/sigma^2, const) of its
//                       components in VGPRs for the whole kernel, evaluates the *centred*
//                       quadratic form in the log2 domain (v_exp_f32 / v_log_f32 are base 2),
//                       does the log-sum-exp with DPP wave reductions and streams log_resp[N,J]
//                       out with coalesced 16-byte stores.  HBM-write-bound: 4 N J bytes.
//   flat_fused_kernel   same mapping, but instead of writing N x J it accumulates the
//                       7 sufficient statistics per component in registers (centred about
//                       the current mean, so no raw-moment cancellation), combines the waves
//                       of a workgroup through LDS in a fixed order and writes one partial per
//                       workgroup.  No atomics, run-to-run deterministic.
//   flat_mstep_kernel   moments from a materialised resp / log_resp matrix (HBM-read-bound).
//   flat_reduce_kernel  fp64 second-stage reduction of the per-workgroup partials; its output
//                       is the buffer the RCCL all-reduce works on.
//   flat_finalize_kernel  fp64 M-step (both reference flavours), next E-step's packed
//                       parameters, log-likelihood trace and the device-side stop rule.
#include "hgmm_ctx.h"
#include "wave_ops.h"

#include <type_traits>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

namespace hgmm {

constexpr float NEG_INF = -__builtin_huge_valf();
constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;
constexpr int PK_MU = 0, PK_G = 3, PK_C = 6, PK_W = 7;   // rows of the packed parameter table (PK_W: the raw weight, for the fused kernel's origin)
constexpr int PK_ROWS = 8;
typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int WAVES_PER_BLOCK = 4;
constexpr int BLOCK = WAVES_PER_BLOCK * 64;

// ------------------------------------------------------------------------------------------
// parameter packing (log2 domain):  wl2_ij = c2_j - sum_d g_jd (x_id - mu_jd)^2 = log2e * wlp_ij
//   g_jd = 0.5 * log2(e) * inv_std_jd^2
//   c2_j = log2(e) * ( -0.5 * 3 * log(2 pi) + sum_d log(inv_std_jd + eps) + log(w_j [+ eps]) )
// (estimate_log_prob / estimate_log_prob_spherical / e_step, gmm_waymo gmm_impl.py:53-116)
// ------------------------------------------------------------------------------------------
// (m, i: the component's mean and inverse standard deviations, spherical already expanded to three axes)
__device__ inline void pack_values(int j, bool valid, int Jpad, int variant, const float (&m)[3], const float (&i)[3],
                                   float wj, float* pack) {
    float m0 = 0.f, m1 = 0.f, m2 = 0.f, g0 = 0.f, g1 = 0.f, g2 = 0.f, c = NEG_INF;
    if (valid) {
        const double eps = (double)FLAT_EPS;
        const double l2e = 1.4426950408889634074;
        const double i0 = i[0], i1 = i[1], i2 = i[2];
        const double log2pi = (double)1.8378770351409912f;   // reference casts log(2 pi) to float32
        double half_log_det = log(i0 + eps) + log(i1 + eps) + log(i2 + eps);
        double lw = (variant == HGMM_VARIANT_W) ? log((double)wj + eps) : log((double)wj);
        double cc = (-0.5 * 3.0 * log2pi + half_log_det + lw) * l2e;
        m0 = m[0]; m1 = m[1]; m2 = m[2];
        g0 = (float)(0.5 * l2e * i0 * i0); g1 = (float)(0.5 * l2e * i1 * i1); g2 = (float)(0.5 * l2e * i2 * i2);
        c = (cc != cc) ? NEG_INF : (float)cc;
    }
    pack[(PK_MU + 0) * Jpad + j] = m0;
    pack[(PK_MU + 1) * Jpad + j] = m1;
    pack[(PK_MU + 2) * Jpad + j] = m2;
    pack[(PK_G + 0) * Jpad + j] = g0;
    pack[(PK_G + 1) * Jpad + j] = g1;
    pack[(PK_G + 2) * Jpad + j] = g2;
    pack[PK_C * Jpad + j] = c;
    // (a component only counts towards the fused kernel's origin with a usable mean and a positive, finite weight)
    const bool usable = valid && wj > 0.f && wj < 3.0e38f && fabsf(m0) < 3.0e38f && fabsf(m1) < 3.0e38f && fabsf(m2) < 3.0e38f;
    pack[PK_W * Jpad + j] = usable ? wj : 0.f;
}
__device__ inline void pack_component(int j, int J, int Jpad, int cov_type, int variant,
                                      const float* mu, const float* inv, const float* w,
                                      float* pack) {
    float m[3] = {0.f, 0.f, 0.f}, i[3] = {0.f, 0.f, 0.f}, wj = 0.f;
    if (j < J) {
        if (cov_type == HGMM_COV_DIAG) { i[0] = inv[3 * j + 0]; i[1] = inv[3 * j + 1]; i[2] = inv[3 * j + 2]; }
        else i[0] = i[1] = i[2] = inv[j];
        m[0] = mu[3 * j + 0]; m[1] = mu[3 * j + 1]; m[2] = mu[3 * j + 2];
        wj = w[j];
    }
    pack_values(j, j < J, Jpad, variant, m, i, wj, pack);
}

__global__ void flat_pack_kernel(int J, int Jpad, int cov_type, int variant, const float* mu,
                                 const float* inv, const float* w, float* pack) {
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < Jpad) pack_component(j, J, Jpad, cov_type, variant, mu, inv, w, pack);
}

// ------------------------------------------------------------------------------------------
// lane <-> component layout.  A lane owns K = 4*NV4 + NV1 components:
//   k <  4*NV4 : "vector" slots, j = ((k/4)*64 + lane)*4 + k%4      (16-byte loads/stores)
//   k >= 4*NV4 : "scalar" slots, j = 256*NV4 + (k-4*NV4)*64 + lane  (4-byte loads/stores)
// J = 800 -> NV4 = 3, NV1 = 1: 13 components per lane, 3 x 1 KiB + 1 x 128 B stores per row.
// ------------------------------------------------------------------------------------------
template <int NV4, int NV1>
struct Layout {
    static constexpr int K = 4 * NV4 + NV1;
    static constexpr int CAP = 256 * NV4 + 64 * NV1;
    __device__ static __forceinline__ int j_of(int k, int lane) {
        return (k < 4 * NV4) ? ((k >> 2) * 64 + lane) * 4 + (k & 3) : 256 * NV4 + (k - 4 * NV4) * 64 + lane;
    }
};

template <int NV4, int NV1>
struct LaneParams {
    using L = Layout<NV4, NV1>;
    static constexpr int K = L::K;
    float mu0[K], mu1[K], mu2[K], g0[K], g1[K], g2[K], c[K];

    __device__ __forceinline__ void load(const float* __restrict__ pack, int Jpad, int lane) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int j = L::j_of(k, lane);
            mu0[k] = pack[(PK_MU + 0) * Jpad + j];
            mu1[k] = pack[(PK_MU + 1) * Jpad + j];
            mu2[k] = pack[(PK_MU + 2) * Jpad + j];
            g0[k] = pack[(PK_G + 0) * Jpad + j];
            g1[k] = pack[(PK_G + 1) * Jpad + j];
            g2[k] = pack[(PK_G + 2) * Jpad + j];
            c[k] = pack[PK_C * Jpad + j];
        }
    }
};

// weighted log2-probabilities of one row for this lane's components; returns the lane-local max
template <int NV4, int NV1>
__device__ __forceinline__ float row_wl2(const LaneParams<NV4, NV1>& P, float x0, float x1, float x2,
                                         float (&wl)[4 * NV4 + NV1]) {
    constexpr int K = 4 * NV4 + NV1;
    float m = NEG_INF;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const float d0 = x0 - P.mu0[k], d1 = x1 - P.mu1[k], d2 = x2 - P.mu2[k];
        float a = fmaf(-(d0 * P.g0[k]), d0, P.c[k]);
        a = fmaf(-(d1 * P.g1[k]), d1, a);
        a = fmaf(-(d2 * P.g2[k]), d2, a);
        wl[k] = a;
        m = fmaxf(m, a);
    }
    return m;
}

// The reference's normaliser  log( sum_j exp(wlp_j) + eps )  (gmm_waymo gmm_impl.py:113, no
// max-shift there) from the wave maximum m2 and S = sum_j 2^(wl2_j - m2), all in log2 units:
//   lpn2 = mc + log2( S' + eps 2^-mc ),  mc = max(m2, -64),  S' = S (or 0 when m2 < -64: then
//   sum_j exp(wlp_j) < 1e-19 J is below float32 resolution of eps = 1e-8).
// inv_den satisfies  r_j = 2^(wl2_j - m2) * inv_den.
constexpr float CS_MAX_SHIFT = 60.0f;      // constant-shift log-sum-exp is used while max_j c2_j <= this (flat_fused_pk_kernel)
__device__ __forceinline__ float lpn2_from(float m2, float s, float& inv_den) {
    const bool tiny = m2 < -64.0f;
    const float mc = tiny ? -64.0f : m2;
    const float den = fmaf(FLAT_EPS, __builtin_amdgcn_exp2f(-mc), tiny ? 0.0f : s);
    inv_den = tiny ? 0.0f : __builtin_amdgcn_rcpf(den);
    return mc + __builtin_amdgcn_logf(den);
}

__device__ __forceinline__ void wave_row_range(int64_t n, int64_t& r0, int64_t& r1) {
    const int64_t nw = (int64_t)gridDim.x * WAVES_PER_BLOCK;
    const int64_t gw = (int64_t)blockIdx.x * WAVES_PER_BLOCK + wave_in_block();
    const int64_t per = (n + nw - 1) / nw;
    r0 = gw * per;
    r1 = r0 + per < n ? r0 + per : n;
    if (r0 > n) r0 = n;
}

template <bool NT>
__device__ __forceinline__ void store_f4(float* p, float a, float b, float c, float d) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    f4 v = {a, b, c, d};
    if (NT) __builtin_nontemporal_store(v, reinterpret_cast<f4*>(p));
    else *reinterpret_cast<f4*>(p) = v;
}
template <bool NT>
__device__ __forceinline__ void store_f1(float* p, float a) {
    if (NT) __builtin_nontemporal_store(a, p);
    else *p = a;
}

// ------------------------------------------------------------------------------------------
// materialising E-step (NORMALISE) / predict (!NORMALISE: arg-max of wlp only)
// ------------------------------------------------------------------------------------------
template <int NV4, int NV1, bool NORMALISE, bool NT>
__global__ __launch_bounds__(BLOCK) void flat_estep_kernel(
    const float* __restrict__ X, const float* __restrict__ pack, int64_t n, int J, int Jpad,
    float* __restrict__ log_resp, float* __restrict__ lpn_out, int32_t* __restrict__ argmax_out,
    double* __restrict__ lpn_partials, int round_robin) {
    using L = Layout<NV4, NV1>;
    constexpr int K = L::K;
    const int lane = lane_id();
    LaneParams<NV4, NV1> P;
    P.load(pack, Jpad, lane);

    // rows of this wave: base + it * stride, it < cnt.  round_robin = 0: one contiguous range per
    // wave; 1: rows dealt round-robin, so that the waves in flight write one contiguous window
    int64_t base, stride, cnt;
    {
        const int64_t nw = (int64_t)gridDim.x * WAVES_PER_BLOCK;
        const int64_t gw = (int64_t)blockIdx.x * WAVES_PER_BLOCK + wave_in_block();
        if (round_robin) {
            base = gw; stride = nw; cnt = (gw < n) ? (n - gw + nw - 1) / nw : 0;
        } else {
            int64_t r0, r1;
            wave_row_range(n, r0, r1);
            base = r0; stride = 1; cnt = r1 - r0;
        }
    }
    double lsum = 0.0;
    float keep_lpn = 0.f;
    int keep_arg = 0;
    float x0 = 0.f, x1 = 0.f, x2 = 0.f;
    if (cnt > 0) { const float* xp = X + 3 * base; x0 = xp[0]; x1 = xp[1]; x2 = xp[2]; }
    for (int64_t it = 0; it < cnt; ++it) {
        const int64_t row = base + it * stride;
        // prefetch the next row's coordinates (wave-uniform scalar loads) behind this row's math
        const int64_t nrow = (it + 1 < cnt) ? row + stride : row;
        const float* xn = X + 3 * nrow;
        const float nx0 = xn[0], nx1 = xn[1], nx2 = xn[2];

        float wl[K];
        float m = row_wl2<NV4, NV1>(P, x0, x1, x2, wl);
        m = wave_max_dpp(m);
        const int slot = (int)(it & 63);
        if (NORMALISE) {
            if (m == NEG_INF) m = 0.f;            // every component has zero weight: avoid inf - inf
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < K; ++k) s += __builtin_amdgcn_exp2f(wl[k] - m);
            s = wave_sum_dpp(s);
            float inv_den;
            const float lpn2 = lpn2_from(m, s, inv_den);
            const float lpn = lpn2 * LN2;
            lsum += (double)lpn;
            if (log_resp) {
                float* out = log_resp + row * (int64_t)J;
#pragma unroll
                for (int v = 0; v < NV4; ++v) {
                    const int jb = (v * 64 + lane) * 4;
                    if (jb < J)
                        store_f4<NT>(out + jb, fmaf(wl[4 * v + 0], LN2, -lpn), fmaf(wl[4 * v + 1], LN2, -lpn),
                                     fmaf(wl[4 * v + 2], LN2, -lpn), fmaf(wl[4 * v + 3], LN2, -lpn));
                }
#pragma unroll
                for (int v = 0; v < NV1; ++v) {
                    const int j = 256 * NV4 + v * 64 + lane;
                    if (j < J) store_f1<NT>(out + j, fmaf(wl[4 * NV4 + v], LN2, -lpn));
                }
            }
            if (lane == slot) keep_lpn = lpn;
        } else if (log_resp) {
            // raw weighted log-probabilities (estimate_log_prob + log weights), natural log; the vector slots as
            // 16-byte stores like the normalised path (written one float at a time this call ran at 3.6 TB/s)
            float* out = log_resp + row * (int64_t)J;
#pragma unroll
            for (int v = 0; v < NV4; ++v) {
                const int jb = (v * 64 + lane) * 4;
                if (jb + 3 < J)
                    store_f4<NT>(out + jb, wl[4 * v + 0] * LN2, wl[4 * v + 1] * LN2, wl[4 * v + 2] * LN2, wl[4 * v + 3] * LN2);
                else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (jb + e < J) out[jb + e] = wl[4 * v + e] * LN2;
                }
            }
#pragma unroll
            for (int v = 0; v < NV1; ++v) {
                const int j = 256 * NV4 + v * 64 + lane;
                if (j < J) out[j] = wl[4 * NV4 + v] * LN2;
            }
        }
        if (argmax_out) {
            // first index attaining the row maximum (numpy argmax tie rule)
            int best = 0x7fffffff;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const int j = L::j_of(k, lane);
                if (wl[k] == m && j < J && j < best) best = j;
            }
            best = wave_reduce_i(best, OpMinI());
            if (best == 0x7fffffff) best = 0;
            if (lane == slot) keep_arg = best;
        }
        if (slot == 63 || it + 1 == cnt) {
            // one store for the last (up to) 64 rows' scalars: lane l holds row base + (it - slot + l) * stride
            if (lane <= slot) {
                const int64_t r = base + (it - slot + lane) * stride;
                if (NORMALISE && lpn_out) lpn_out[r] = keep_lpn;
                if (argmax_out) argmax_out[r] = keep_arg;
            }
        }
        x0 = nx0; x1 = nx1; x2 = nx2;
    }
    if (NORMALISE && lpn_partials) {
        __shared__ double sh[WAVES_PER_BLOCK];
        if (lane == 0) sh[wave_in_block()] = lsum;
        __syncthreads();
        if (threadIdx.x == 0) {
            double t = 0.0;
            for (int i = 0; i < WAVES_PER_BLOCK; ++i) t += sh[i];
            lpn_partials[blockIdx.x] = t;
        }
    }
}

// ------------------------------------------------------------------------------------------
// Store pacing.  The HBM write path of this chip delivers LESS the more stores are waiting for it: a grid that offers
// the 3200-byte rows faster than memory drains them falls from ~6.0 to ~5.3 TB/s (tools/logprob_sweep.py: the raw-table
// kernel, which has nothing to do but store, ran 0.60 ms at every grid from 128 to 4096 workgroups -- and 0.536 ms once
// every wave idled ~1000 cycles behind each group of rows).  The E-step kernel had found the same sweet spot by
// accident: at exactly 3/4 of the CUs its arithmetic happened to offer the rows at the drain rate -- at one core clock;
// behind a memory-bound kernel or in an unwaited-for stream the clock differs and the optimum moved (estep_rows_grid).
// So the rate is made explicit: every wave issues its i-th group of rows no earlier than t0 + i * period on the
// constant-rate wall clock (100 MHz), period chosen on the host so that all waves together offer the target rate.
// Enough waves are launched for the arithmetic to keep up at any clock; the clock then no longer matters.
// (The single-row kernel that serves the layouts without 16-byte stores -- J not a multiple of 4 -- is not paced: it is
//  bound by its per-row reduction chain, not by the write path, and reaches 6.0 - 6.2 TB/s un-paced; paced at the same
//  target it measured 2 - 3 % slower: J = 801 / 515 / 1001 at N = 1e6: 0.529 / 0.348 / 0.648 ms against 0.541 / 0.360 / 0.666.)
// ------------------------------------------------------------------------------------------
// (struct StorePacer: wave_ops.h -- the write-ceiling probe of hgmm_api.hip paces its stores with the same code)

// ------------------------------------------------------------------------------------------
// materialising E-step, ROWS consecutive rows in flight per wave, components paired into float2.
// No accumulators live in this kernel (parameters 7K + ROWS*K values), so the independent max / sum
// reduction chains of several rows interleave inside ONE wave; that lets the kernel run with few
// waves per CU, and few, orderly writers is what the HBM write path rewards (tools/fillbench.py:
// 1024 waves writing a common 1 MiB window reach 6.4 TB/s, 2048 waves 5.3 TB/s, private streams
// 5.5 TB/s).  Row groups are dealt round-robin over the waves so the waves in flight write one
// window.  With one wave per SIMD the kernel's own instruction count matters, hence the pairing
// (v_pk_add/mul/fma_f32): pairs (4v, 4v+1), (4v+2, 4v+3) of every vector slot, pairs of scalar-tail
// slots, at most one trailing single.
// Measured (kbench, N = 1e6, J = 800, same box): single-row kernel, 2 workgroups/CU 0.672 ms;
// 6 rows unpaired 0.603 ms; paired 4 / 6 / 8 rows 0.546 / 0.561 / 0.604 ms.  What counts is the number
// of waves writing at once (row-maximum loop): 1024 waves (4 per workgroup, 1 workgroup per CU) 0.54 - 0.56 ms, 2048 waves
// 0.65 ms whether as 8 waves per workgroup or as 2 workgroups per CU, with 2 or 4 rows each.  The 4 rows of a
// wave must be consecutive (one 12.8 KB run): dealing single rows round-robin over the waves (each wave's 4
// rows 3.2 MB apart) drops to 0.86 ms.
// ------------------------------------------------------------------------------------------
template <int NV4, int NV1, int ROWS, bool NT>
__global__ __launch_bounds__(BLOCK) void flat_estep_rows_pk_kernel(
    const float* __restrict__ X, const float* __restrict__ pack, int64_t n, int J, int Jpad,
    float* __restrict__ log_resp, float* __restrict__ lpn_out, int32_t* __restrict__ argmax_out,
    double* __restrict__ lpn_partials, int allow_const_shift, int pace,
    unsigned long long* __restrict__ stamp /*[2][gridDim.x] wall-clock start / end per workgroup, pinned host memory; may be null*/) {
    using L = Layout<NV4, NV1>;
    constexpr int K = L::K;
    constexpr int KP = 2 * NV4 + NV1 / 2;          // pairs
    constexpr bool ODD = (NV1 & 1) != 0;           // trailing single = component K - 1
    const int lane = lane_id();
    if (stamp && threadIdx.x == 0) stamp[blockIdx.x] = wall_clock64();
    f2 mu0[KP + 1], mu1[KP + 1], mu2[KP + 1], g0[KP + 1], g1[KP + 1], g2[KP + 1], cc[KP + 1];
    float mu0s = 0.f, mu1s = 0.f, mu2s = 0.f, g0s = 0.f, g1s = 0.f, g2s = 0.f, cs = NEG_INF;
    auto ld = [&](int row, int j) { return pack[row * Jpad + j]; };
#pragma unroll
    for (int p = 0; p < KP; ++p) {
        const int ka = 2 * p, kb = 2 * p + 1;
        const int ja = L::j_of(ka, lane), jb = L::j_of(kb, lane);
        mu0[p] = f2{ld(PK_MU + 0, ja), ld(PK_MU + 0, jb)};
        mu1[p] = f2{ld(PK_MU + 1, ja), ld(PK_MU + 1, jb)};
        mu2[p] = f2{ld(PK_MU + 2, ja), ld(PK_MU + 2, jb)};
        g0[p] = f2{ld(PK_G + 0, ja), ld(PK_G + 0, jb)};
        g1[p] = f2{ld(PK_G + 1, ja), ld(PK_G + 1, jb)};
        g2[p] = f2{ld(PK_G + 2, ja), ld(PK_G + 2, jb)};
        cc[p] = f2{ld(PK_C, ja), ld(PK_C, jb)};
    }
    if (ODD) {
        const int j = L::j_of(K - 1, lane);
        mu0s = ld(PK_MU + 0, j); mu1s = ld(PK_MU + 1, j); mu2s = ld(PK_MU + 2, j);
        g0s = ld(PK_G + 0, j); g1s = ld(PK_G + 1, j); g2s = ld(PK_G + 2, j);
        cs = ld(PK_C, j);
    }
    // constant-shift log-sum-exp (see flat_fused_pk_kernel): M0 = max_j c2_j bounds every wl2, so with M0 folded into
    // the constants the per-row maximum (2 v_max per pair, a wave reduction, one subtraction per pair) disappears.
    // With ONE wave per SIMD -- what the HBM write path wants -- a wave issues one VALU instruction per ~5 cycles
    // whatever its ILP, and this kernel was bound by exactly that (0.52 ms with its stores switched off, 0.55 with
    // them): fewer instructions is the only lever.  The arg-max output needs the row maximum: generic loop.
    float m0 = cs;
#pragma unroll
    for (int p = 0; p < KP; ++p) m0 = fmaxf(m0, fmaxf(cc[p].x, cc[p].y));
    m0 = wave_max_dpp(m0);
    // Here every log_resp value is an OUTPUT (the fused kernel only sums): with the shift folded in, wl and the
    // log-denominator both carry a magnitude of ~M0 and their difference inherits ulp(M0) -- 2e-6 at 24, 8e-6 at 60
    // against the 1e-5 parity bar on the responsibilities -- so the loop is taken for moderate shifts only.
    constexpr float ESTEP_CS_MAX_SHIFT = 24.0f;
    const bool small_shift = allow_const_shift && !argmax_out && m0 <= ESTEP_CS_MAX_SHIFT && m0 >= -24.0f;

    const int64_t nw = (int64_t)gridDim.x * WAVES_PER_BLOCK;
    const int64_t gw = (int64_t)blockIdx.x * WAVES_PER_BLOCK + wave_in_block();
    const int64_t ngroups = (n + ROWS - 1) / ROWS;
    double lsum = 0.0;
    float x[ROWS][3];
    auto load_group = [&](int64_t g, float (&dst)[ROWS][3]) {
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            int64_t row = g * ROWS + r;
            row = row < n ? row : n - 1;
            const float* xp = X + 3 * row;
            dst[r][0] = xp[0]; dst[r][1] = xp[1]; dst[r][2] = xp[2];
        }
    };
    auto groups = [&](auto cs_tag) {
    constexpr bool CS = decltype(cs_tag)::value;
    if (CS) {
        const f2 M0 = f2{m0, m0};
#pragma unroll
        for (int p = 0; p < KP; ++p) cc[p] = cc[p] - M0;
        cs -= m0;
    }
    const float eps_scale = __builtin_amdgcn_exp2f(-m0);
    StorePacer pacer(pace, gw, nw);
    if (gw < ngroups) load_group(gw, x);
    for (int64_t g = gw; g < ngroups; g += nw) {
        float nx[ROWS][3];
        load_group((g + nw < ngroups) ? g + nw : g, nx);          // prefetch (scalar loads)
        f2 wl[ROWS][KP + 1];
        float wls[ROWS];
        float m[ROWS], s[ROWS];
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const f2 X0 = f2{x[r][0], x[r][0]}, X1 = f2{x[r][1], x[r][1]}, X2 = f2{x[r][2], x[r][2]};
            float mm = NEG_INF;
            f2 acc = f2{0.f, 0.f};
#pragma unroll
            for (int p = 0; p < KP; ++p) {
                const f2 d0 = X0 - mu0[p], d1 = X1 - mu1[p], d2 = X2 - mu2[p];
                f2 a = cc[p] - (d0 * g0[p]) * d0;
                a = a - (d1 * g1[p]) * d1;
                a = a - (d2 * g2[p]) * d2;
                wl[r][p] = a;
                if (CS) acc += f2{__builtin_amdgcn_exp2f(a.x), __builtin_amdgcn_exp2f(a.y)};
                else mm = fmaxf(mm, fmaxf(a.x, a.y));
            }
            wls[r] = NEG_INF;
            if (ODD) {
                const float d0 = x[r][0] - mu0s, d1 = x[r][1] - mu1s, d2 = x[r][2] - mu2s;
                float a = fmaf(-(d0 * g0s), d0, cs);
                a = fmaf(-(d1 * g1s), d1, a);
                a = fmaf(-(d2 * g2s), d2, a);
                wls[r] = a;
                mm = fmaxf(mm, a);
            }
            m[r] = mm;
            if (CS) {
                float a = acc.x + acc.y;
                if (ODD) a += __builtin_amdgcn_exp2f(wls[r]);
                s[r] = a;
            }
        }
        if (!CS) {
            if constexpr (ROWS == 4) wave_max4_dpp(m);
            else {
#pragma unroll
                for (int r = 0; r < ROWS; ++r) m[r] = wave_max_dpp(m[r]);
            }
#pragma unroll
            for (int r = 0; r < ROWS; ++r)
                if (m[r] == NEG_INF) m[r] = 0.f;
#pragma unroll
            for (int r = 0; r < ROWS; ++r) {
                const f2 M = f2{m[r], m[r]};
                f2 acc = f2{0.f, 0.f};
#pragma unroll
                for (int p = 0; p < KP; ++p) {
                    const f2 t = wl[r][p] - M;
                    acc += f2{__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)};
                }
                float a = acc.x + acc.y;
                if (ODD) a += __builtin_amdgcn_exp2f(wls[r] - m[r]);
                s[r] = a;
            }
        }
        if constexpr (ROWS == 4) wave_sum4_dpp(s);
        else {
#pragma unroll
            for (int r = 0; r < ROWS; ++r) s[r] = wave_sum_dpp(s[r]);
        }
        float keep_lpn = 0.f;
        int keep_arg = 0;
        pacer.wait();
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const int64_t row = g * ROWS + r;
            float lpn;          // the row's log-normaliser
            float sub;          // what is subtracted from wl * ln 2 (wl carries the shift -M0 under CS)
            if (CS) {
                // lpn2_from with m = m0 in [-64, CS_MAX_SHIFT]: no "tiny" case, eps 2^-m0 is loop-invariant
                const float lden = __builtin_amdgcn_logf(fmaf(FLAT_EPS, eps_scale, s[r]));
                lpn = (m0 + lden) * LN2;
                sub = lden * LN2;
            } else {
                float inv_den;
                lpn = lpn2_from(m[r], s[r], inv_den) * LN2;
                sub = lpn;
            }
            const f2 LNV = f2{LN2, LN2}, NL = f2{-sub, -sub};
            if (row < n) {                                          // wave-uniform
                lsum += (double)lpn;
                float* out = log_resp + row * (int64_t)J;
#pragma unroll
                for (int v = 0; v < NV4; ++v) {
                    const int jb = (v * 64 + lane) * 4;
                    const f2 lo = wl[r][2 * v] * LNV + NL, hi = wl[r][2 * v + 1] * LNV + NL;
                    if (jb < J) store_f4<NT>(out + jb, lo.x, lo.y, hi.x, hi.y);
                }
#pragma unroll
                for (int v = 0; v < NV1; ++v) {
                    const int j = 256 * NV4 + v * 64 + lane;
                    const int k = 4 * NV4 + v;
                    const float val = (ODD && k == K - 1) ? wls[r] : ((k & 1) ? wl[r][k >> 1].y : wl[r][k >> 1].x);
                    if (j < J) store_f1<NT>(out + j, fmaf(val, LN2, -sub));
                }
            }
            if (lane == r) keep_lpn = lpn;
            if (!CS && argmax_out) {
                int best = 0x7fffffff;
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const int j = L::j_of(k, lane);
                    const float val = (ODD && k == K - 1) ? wls[r] : ((k & 1) ? wl[r][k >> 1].y : wl[r][k >> 1].x);
                    if (val == m[r] && j < J && j < best) best = j;
                }
                best = wave_reduce_i(best, OpMinI());
                if (best == 0x7fffffff) best = 0;
                if (lane == r) keep_arg = best;
            }
        }
        if (lane < ROWS) {
            const int64_t row = g * ROWS + lane;
            if (row < n) {
                if (lpn_out) lpn_out[row] = keep_lpn;
                if (!CS && argmax_out) argmax_out[row] = keep_arg;
            }
        }
#pragma unroll
        for (int r = 0; r < ROWS; ++r) { x[r][0] = nx[r][0]; x[r][1] = nx[r][1]; x[r][2] = nx[r][2]; }
    }
    };
    if (small_shift) groups(std::true_type{});
    else groups(std::false_type{});
    if (lpn_partials) {
        __shared__ double sh[WAVES_PER_BLOCK];
        if (lane == 0) sh[wave_in_block()] = lsum;
        __syncthreads();
        if (threadIdx.x == 0) {
            double t = 0.0;
            for (int i = 0; i < WAVES_PER_BLOCK; ++i) t += sh[i];
            lpn_partials[blockIdx.x] = t;
        }
    }
    if (stamp) {                                   // every wave's stores have been acknowledged, then the workgroup's end
        __builtin_amdgcn_s_waitcnt(0x0F70);        // vmcnt(0)
        __syncthreads();
        if (threadIdx.x == 0) stamp[gridDim.x + blockIdx.x] = wall_clock64();
    }
}

// ------------------------------------------------------------------------------------------
// estimate_log_prob (gmm_impl.py:53-78): the un-normalised table, natural log.  The materialising E-step's layout, four
// rows in flight and packed arithmetic with everything but the quadratic forms taken out: no maximum, no exponential,
// no wave reduction -- 10 packed instructions per pair of components and row (9 for the form, one for log2 -> ln),
// then the same 16-byte non-temporal stores.  (Round 3 ran this call on the single-row kernel: 0.69 ms per
// 10^6 x 800 table = 4.7 TB/s while e_step, which does strictly more work on the same bytes, ran at 5.9.)
// ------------------------------------------------------------------------------------------
template <int NV4, int NV1, bool LATE>
__global__ __launch_bounds__(BLOCK) void flat_logprob_rows_pk_kernel(
    const float* __restrict__ X, const float* __restrict__ pack, int64_t n, int J, int Jpad,
    float* __restrict__ log_prob, int pace) {
    using L = Layout<NV4, NV1>;
    constexpr int K = L::K;
    constexpr int KP = 2 * NV4 + NV1 / 2;          // pairs
    constexpr bool ODD = (NV1 & 1) != 0;           // trailing single = component K - 1
    constexpr int ROWS = 4;
    const int lane = lane_id();
    f2 mu0[KP + 1], mu1[KP + 1], mu2[KP + 1], g0[KP + 1], g1[KP + 1], g2[KP + 1], cc[KP + 1];
    float mu0s = 0.f, mu1s = 0.f, mu2s = 0.f, g0s = 0.f, g1s = 0.f, g2s = 0.f, cs = NEG_INF;
    auto ld = [&](int row, int j) { return pack[row * Jpad + j]; };
    // the table is kept in the log2 domain (pack_values); here the NATURAL log is the output, so the factor ln 2 is
    // folded into the lane's copy of (g, c) once instead of costing an instruction per pair and row
    const f2 LNV = f2{LN2, LN2};
#pragma unroll
    for (int p = 0; p < KP; ++p) {
        const int ja = L::j_of(2 * p, lane), jb = L::j_of(2 * p + 1, lane);
        mu0[p] = f2{ld(PK_MU + 0, ja), ld(PK_MU + 0, jb)};
        mu1[p] = f2{ld(PK_MU + 1, ja), ld(PK_MU + 1, jb)};
        mu2[p] = f2{ld(PK_MU + 2, ja), ld(PK_MU + 2, jb)};
        g0[p] = f2{ld(PK_G + 0, ja), ld(PK_G + 0, jb)} * LNV;
        g1[p] = f2{ld(PK_G + 1, ja), ld(PK_G + 1, jb)} * LNV;
        g2[p] = f2{ld(PK_G + 2, ja), ld(PK_G + 2, jb)} * LNV;
        cc[p] = f2{ld(PK_C, ja), ld(PK_C, jb)} * LNV;
    }
    if (ODD) {
        const int j = L::j_of(K - 1, lane);
        mu0s = ld(PK_MU + 0, j); mu1s = ld(PK_MU + 1, j); mu2s = ld(PK_MU + 2, j);
        g0s = ld(PK_G + 0, j) * LN2; g1s = ld(PK_G + 1, j) * LN2; g2s = ld(PK_G + 2, j) * LN2;
        cs = ld(PK_C, j) * LN2;
    }
    const int64_t nw = (int64_t)gridDim.x * WAVES_PER_BLOCK;
    const int64_t gw = (int64_t)blockIdx.x * WAVES_PER_BLOCK + wave_in_block();
    const int64_t ngroups = (n + ROWS - 1) / ROWS;
    float x[ROWS][3];
    auto load_group = [&](int64_t g, float (&dst)[ROWS][3]) {
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            int64_t row = g * ROWS + r;
            row = row < n ? row : n - 1;
            const float* xp = X + 3 * row;
            dst[r][0] = xp[0]; dst[r][1] = xp[1]; dst[r][2] = xp[2];
        }
    };
    StorePacer pacer(pace, gw, nw);
    if (gw < ngroups) load_group(gw, x);
    for (int64_t g = gw; g < ngroups; g += nw) {
        float nx[ROWS][3];
        load_group((g + nw < ngroups) ? g + nw : g, nx);          // prefetch (scalar loads)
        f2 wl[ROWS][KP + 1];
        float wls[ROWS];
        auto compute = [&](int r) {
            const f2 X0 = f2{x[r][0], x[r][0]}, X1 = f2{x[r][1], x[r][1]}, X2 = f2{x[r][2], x[r][2]};
#pragma unroll
            for (int p = 0; p < KP; ++p) {
                const f2 d0 = X0 - mu0[p], d1 = X1 - mu1[p], d2 = X2 - mu2[p];
                f2 a = cc[p] - (d0 * g0[p]) * d0;
                a = a - (d1 * g1[p]) * d1;
                a = a - (d2 * g2[p]) * d2;
                wl[r][p] = a;
            }
            wls[r] = NEG_INF;
            if (ODD) {
                const float d0 = x[r][0] - mu0s, d1 = x[r][1] - mu1s, d2 = x[r][2] - mu2s;
                float a = fmaf(-(d0 * g0s), d0, cs);
                a = fmaf(-(d1 * g1s), d1, a);
                a = fmaf(-(d2 * g2s), d2, a);
                wls[r] = a;
            }
        };
        auto store = [&](int r) {
            const int64_t row = g * ROWS + r;
            if (row < n) {                                          // wave-uniform
                float* out = log_prob + row * (int64_t)J;
#pragma unroll
                for (int v = 0; v < NV4; ++v) {
                    const int jb = (v * 64 + lane) * 4;
                    if (jb < J) store_f4<true>(out + jb, wl[r][2 * v].x, wl[r][2 * v].y, wl[r][2 * v + 1].x, wl[r][2 * v + 1].y);
                }
#pragma unroll
                for (int v = 0; v < NV1; ++v) {
                    const int j = 256 * NV4 + v * 64 + lane;
                    const int k = 4 * NV4 + v;
                    const float val = (ODD && k == K - 1) ? wls[r] : ((k & 1) ? wl[r][k >> 1].y : wl[r][k >> 1].x);
                    if (j < J) store_f1<true>(out + j, val);
                }
            }
        };
        if (LATE) {             // the four rows' forms first, then one 12.8 KB run of stores (the E-step's pattern)
#pragma unroll
            for (int r = 0; r < ROWS; ++r) compute(r);
            asm volatile("" ::: "memory");
            pacer.wait();
#pragma unroll
            for (int r = 0; r < ROWS; ++r) store(r);
        } else {
#pragma unroll
            for (int r = 0; r < ROWS; ++r) { compute(r); store(r); }
        }
#pragma unroll
        for (int r = 0; r < ROWS; ++r) { x[r][0] = nx[r][0]; x[r][1] = nx[r][1]; x[r][2] = nx[r][2]; }
    }
}

// ------------------------------------------------------------------------------------------
// predict(): arg-max_j of the weighted log-probabilities, nothing else (gmm_impl.py:147-155).  Same lane <-> component
// layout and the same packed arithmetic as flat_estep_rows_pk_kernel (so the values compared are the ones e_step's own
// arg-max sees), four rows in flight per wave -- and no exponentials, no normaliser, no N x J traffic: per row the
// quadratic forms (9 packed instructions per pair of components), the row maximum (one 4-wide DPP reduction for the
// four rows) and the smallest index that attains it (indices ride as negated floats through the same 4-wide max
// reduction: a second compare-and-select pass instead of an index carried through the first).
// (The single-row kernel this replaces for predict ran 0.33 ms per 10^6 x 800 frame; its row maximum, index search
//  and two 6-step wave reductions were one dependent chain per row.)
// ------------------------------------------------------------------------------------------
// Every lane searches its own K values for the row maximum (compare + select per component on all 64 lanes) and a second
// 4-wide DPP reduction finds the smallest index.  (Round 5 built the alternative -- ballot the lanes that hold the maximum,
// read the single holder's K values into SGPRs, compare on the scalar unit: 157 instead of 174 registers, three waves per
// SIMD -- and measured 0.251 against 0.246 ms at J = 800, 0.334 against 0.276 at J = 1024: removed in round 6,
// profiles/r05/predict_modes.log keeps the figures.)
template <int NV4, int NV1>
__global__ __launch_bounds__(BLOCK) void flat_predict_rows_kernel(
    const float* __restrict__ X, const float* __restrict__ pack, int64_t n, int J, int Jpad,
    int32_t* __restrict__ labels) {
    using L = Layout<NV4, NV1>;
    constexpr int K = L::K;
    constexpr int KP = 2 * NV4 + NV1 / 2;          // pairs
    constexpr bool ODD = (NV1 & 1) != 0;           // trailing single = component K - 1
    constexpr int ROWS = 4;
    const int lane = lane_id();
    f2 mu0[KP + 1], mu1[KP + 1], mu2[KP + 1], g0[KP + 1], g1[KP + 1], g2[KP + 1], cc[KP + 1];
    float mu0s = 0.f, mu1s = 0.f, mu2s = 0.f, g0s = 0.f, g1s = 0.f, g2s = 0.f, cs = NEG_INF;
    auto ld = [&](int row, int j) { return pack[row * Jpad + j]; };
#pragma unroll
    for (int p = 0; p < KP; ++p) {
        const int ja = L::j_of(2 * p, lane), jb = L::j_of(2 * p + 1, lane);
        mu0[p] = f2{ld(PK_MU + 0, ja), ld(PK_MU + 0, jb)};
        mu1[p] = f2{ld(PK_MU + 1, ja), ld(PK_MU + 1, jb)};
        mu2[p] = f2{ld(PK_MU + 2, ja), ld(PK_MU + 2, jb)};
        g0[p] = f2{ld(PK_G + 0, ja), ld(PK_G + 0, jb)};
        g1[p] = f2{ld(PK_G + 1, ja), ld(PK_G + 1, jb)};
        g2[p] = f2{ld(PK_G + 2, ja), ld(PK_G + 2, jb)};
        cc[p] = f2{ld(PK_C, ja), ld(PK_C, jb)};
    }
    if (ODD) {
        const int j = L::j_of(K - 1, lane);
        mu0s = ld(PK_MU + 0, j); mu1s = ld(PK_MU + 1, j); mu2s = ld(PK_MU + 2, j);
        g0s = ld(PK_G + 0, j); g1s = ld(PK_G + 1, j); g2s = ld(PK_G + 2, j);
        cs = ld(PK_C, j);
    }
    float njf[K];                                  // -(index): the smallest index is the largest of these
#pragma unroll
    for (int k = 0; k < K; ++k) njf[k] = -(float)L::j_of(k, lane);

    const int64_t nw = (int64_t)gridDim.x * WAVES_PER_BLOCK;
    const int64_t gw = (int64_t)blockIdx.x * WAVES_PER_BLOCK + wave_in_block();
    const int64_t ngroups = (n + ROWS - 1) / ROWS;
    float x[ROWS][3];
    auto load_group = [&](int64_t g, float (&dst)[ROWS][3]) {
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            int64_t row = g * ROWS + r;
            row = row < n ? row : n - 1;
            const float* xp = X + 3 * row;
            dst[r][0] = xp[0]; dst[r][1] = xp[1]; dst[r][2] = xp[2];
        }
    };
    if (gw < ngroups) load_group(gw, x);
    for (int64_t g = gw; g < ngroups; g += nw) {
        float nx[ROWS][3];
        load_group((g + nw < ngroups) ? g + nw : g, nx);          // prefetch (scalar loads)
        f2 wl[ROWS][KP + 1];
        float wls[ROWS];
        float m[ROWS];
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const f2 X0 = f2{x[r][0], x[r][0]}, X1 = f2{x[r][1], x[r][1]}, X2 = f2{x[r][2], x[r][2]};
            float mm = NEG_INF;
#pragma unroll
            for (int p = 0; p < KP; ++p) {
                const f2 d0 = X0 - mu0[p], d1 = X1 - mu1[p], d2 = X2 - mu2[p];
                f2 a = cc[p] - (d0 * g0[p]) * d0;
                a = a - (d1 * g1[p]) * d1;
                a = a - (d2 * g2[p]) * d2;
                wl[r][p] = a;
                mm = fmaxf(mm, fmaxf(a.x, a.y));
            }
            wls[r] = NEG_INF;
            if (ODD) {
                const float d0 = x[r][0] - mu0s, d1 = x[r][1] - mu1s, d2 = x[r][2] - mu2s;
                float a = fmaf(-(d0 * g0s), d0, cs);
                a = fmaf(-(d1 * g1s), d1, a);
                a = fmaf(-(d2 * g2s), d2, a);
                wls[r] = a;
                mm = fmaxf(mm, a);
            }
            m[r] = mm;
        }
        wave_max4_dpp(m);
        float b[ROWS];
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            float best = NEG_INF;                                  // descending k: the last hit is the lane's smallest index
#pragma unroll
            for (int k = K - 1; k >= 0; --k) {
                const float val = (ODD && k == K - 1) ? wls[r] : ((k & 1) ? wl[r][k >> 1].y : wl[r][k >> 1].x);
                best = (val == m[r]) ? njf[k] : best;
            }
            b[r] = best;
        }
        wave_max4_dpp(b);
        if (lane < ROWS) {
            const int64_t row = g * ROWS + lane;
            const float mb = lane == 0 ? m[0] : lane == 1 ? m[1] : lane == 2 ? m[2] : m[3];
            const float bb = lane == 0 ? b[0] : lane == 1 ? b[1] : lane == 2 ? b[2] : b[3];
            // every component at -inf (zero weights) or NaN everywhere: index 0, like argmax of a constant row
            int lab = (mb == NEG_INF || !(bb > NEG_INF)) ? 0 : (int)(-bb);
            if (lab >= J) lab = 0;
            if (row < n) labels[row] = lab;
        }
#pragma unroll
        for (int r = 0; r < ROWS; ++r) { x[r][0] = nx[r][0]; x[r][1] = nx[r][1]; x[r][2] = nx[r][2]; }
    }
}

// ------------------------------------------------------------------------------------------
// fused E+M: sufficient statistics without the N x J round trip
//   s0_j = sum_i r_ij,  a_jd = sum_i r_ij (x_id - mu_jd),  b_jd = sum_i r_ij (x_id - mu_jd)^2
// (the un-paired kernel of round 1 -- one component per VALU lane-instruction -- stayed behind HGMM_FUSED_PK=0 until
//  round 6 and went with the switch: nothing else reached it)
// ------------------------------------------------------------------------------------------
// fused E+M with explicitly paired arithmetic: components (2p, 2p+1) of a lane travel together as
// a float2, so that the centred quadratic form and the moment update issue as v_pk_add/mul/fma_f32
// (two components per VALU instruction; the fp32 peak of gfx950 is only reachable with packed
// ops).
// ------------------------------------------------------------------------------------------
//
// CS = constant-shift log-sum-exp.  wl2_j <= c2_j for every point (the quadratic form is <= 0),
// so M0 = max_j c2_j bounds every row and 2^(wl2 - M0) cannot overflow: with M0 folded into the
// constants the per-row maximum (K v_max + one wave reduction + K subtractions) disappears and the
// exponentials are taken straight out of the quadratic form.  What the fixed shift flushes to
// zero is < 2^(M0 - 126), i.e. < 1e-19 relative to the eps = 1e-8 of the reference's normaliser
// while M0 <= CS_MAX_SHIFT, and the denominator stays >= eps 2^-M0 >> FLT_MIN.  A model with a
// larger M0 (sigma < ~1e-7: only the clipped-covariance flavour can get there) is served by the
// row-maximum loop of the same kernel: every wave derives the same M0 from the table and picks its loop.

template <int NSLOT>
__global__ __launch_bounds__(BLOCK) void flat_fused_pk_kernel(
    const float* __restrict__ X, const float* __restrict__ pack, int64_t n, int J, int Jpad,
    float* __restrict__ partials, double* __restrict__ lpn_partials,
    const int* __restrict__ done_flag, int allow_const_shift) {
    if (done_flag && *done_flag) return;
    constexpr int K = NSLOT;
    constexpr int KP = K / 2;          // full pairs
    constexpr bool ODD = (K & 1) != 0; // one trailing single component
    const int lane = lane_id();
    f2 mu0[KP + 1], mu1[KP + 1], mu2[KP + 1], g0[KP + 1], g1[KP + 1], g2[KP + 1], cc[KP + 1];
    f2 s0[KP + 1], a0[KP + 1], a1[KP + 1], a2[KP + 1], b0[KP + 1], b1[KP + 1], b2[KP + 1];
    float mu0s = 0.f, mu1s = 0.f, mu2s = 0.f, g0s = 0.f, g1s = 0.f, g2s = 0.f, cs = NEG_INF;
    float s0s = 0.f, a0s = 0.f, a1s = 0.f, a2s = 0.f, b0s = 0.f, b1s = 0.f, b2s = 0.f;
#pragma unroll
    for (int p = 0; p < KP; ++p) {
        const int ja = (2 * p) * 64 + lane, jb = (2 * p + 1) * 64 + lane;
        mu0[p] = f2{pack[(PK_MU + 0) * Jpad + ja], pack[(PK_MU + 0) * Jpad + jb]};
        mu1[p] = f2{pack[(PK_MU + 1) * Jpad + ja], pack[(PK_MU + 1) * Jpad + jb]};
        mu2[p] = f2{pack[(PK_MU + 2) * Jpad + ja], pack[(PK_MU + 2) * Jpad + jb]};
        g0[p] = f2{pack[(PK_G + 0) * Jpad + ja], pack[(PK_G + 0) * Jpad + jb]};
        g1[p] = f2{pack[(PK_G + 1) * Jpad + ja], pack[(PK_G + 1) * Jpad + jb]};
        g2[p] = f2{pack[(PK_G + 2) * Jpad + ja], pack[(PK_G + 2) * Jpad + jb]};
        cc[p] = f2{pack[PK_C * Jpad + ja], pack[PK_C * Jpad + jb]};
        s0[p] = a0[p] = a1[p] = a2[p] = b0[p] = b1[p] = b2[p] = f2{0.f, 0.f};
    }
    if (ODD) {
        const int j = (K - 1) * 64 + lane;
        mu0s = pack[(PK_MU + 0) * Jpad + j]; mu1s = pack[(PK_MU + 1) * Jpad + j]; mu2s = pack[(PK_MU + 2) * Jpad + j];
        g0s = pack[(PK_G + 0) * Jpad + j]; g1s = pack[(PK_G + 1) * Jpad + j]; g2s = pack[(PK_G + 2) * Jpad + j];
        cs = pack[PK_C * Jpad + j];
    }

    // largest constant of the table = upper bound of every wl2 (identical in every wave)
    float m0 = cs;
#pragma unroll
    for (int p = 0; p < KP; ++p) m0 = fmaxf(m0, fmaxf(cc[p].x, cc[p].y));
    m0 = wave_max_dpp(m0);
    // (false for NaN as well; below -64 the row-maximum loop's "tiny" rule applies, see lpn2_from)
    const bool small_shift = allow_const_shift && m0 <= CS_MAX_SHIFT && m0 >= -64.0f;

    int64_t r0, r1;
    wave_row_range(n, r0, r1);
    double lsum = 0.0;
    // First moments are summed about ONE origin for all components and moved to the component's mean when the wave is
    // done:  sum r (x - mu) = sum r (x - o) - (mu - o) sum r.  The row's x - o is a wave-uniform operand, so the update
    // is one packed FMA per axis -- and the squares (x - mu)^2 the quadratic form needs anyway are what the second
    // moments need: the per-component differences themselves do not have to survive the row's reduction.  18 instead of
    // 21 packed instructions per pair of components (round 4: 0.374 -> 0.33 ms at C3).  The origin is the weighted mean
    // of the component means (every wave derives the same one from the table) -- inside the model whatever the
    // coordinates' offset and whatever outliers the cloud has.  The price: a first moment's
    // rounding error is relative to the MODEL's extent instead of the component's, 2^-24 |x - o| per addition -- the
    // reference's own float32 resp.T @ X has it relative to |x| (tests: test_fit_far_from_the_origin,
    // test_train_far_points_and_mixed_scales).
    // ... WEIGHTED by the mixing weights (table row PK_W; 0 for padding, dead and non-finite components): a component
    // that has lost its points sits wherever its last M-step left it -- at the coordinate origin, which for a cloud in
    // map coordinates is kilometres outside the model -- and must not pull the common origin (and with it every
    // component's rounding error) towards itself.
    float o0 = 0.f, o1 = 0.f, o2 = 0.f, ow = 0.f;
#pragma unroll
    for (int p = 0; p < KP; ++p) {
        const float wa = pack[PK_W * Jpad + (2 * p) * 64 + lane], wb = pack[PK_W * Jpad + (2 * p + 1) * 64 + lane];
        o0 = fmaf(wa, mu0[p].x, fmaf(wb, mu0[p].y, o0));
        o1 = fmaf(wa, mu1[p].x, fmaf(wb, mu1[p].y, o1));
        o2 = fmaf(wa, mu2[p].x, fmaf(wb, mu2[p].y, o2));
        ow += wa + wb;
    }
    if (ODD) {
        const float ws = pack[PK_W * Jpad + (K - 1) * 64 + lane];
        o0 = fmaf(ws, mu0s, o0); o1 = fmaf(ws, mu1s, o1); o2 = fmaf(ws, mu2s, o2);
        ow += ws;
    }
    {
        ow = wave_sum_dpp(ow);
        const float inv_w = ow > 0.f ? 1.0f / ow : 0.f;      // (no usable component at all: the coordinate origin)
        o0 = wave_sum_dpp(o0) * inv_w; o1 = wave_sum_dpp(o1) * inv_w; o2 = wave_sum_dpp(o2) * inv_w;
        if (!(fabsf(o0) < 3.0e38f)) o0 = 0.f;
        if (!(fabsf(o1) < 3.0e38f)) o1 = 0.f;
        if (!(fabsf(o2) < 3.0e38f)) o2 = 0.f;
    }
    // the row loop exists twice in this kernel, once per log-sum-exp variant; the choice is uniform
    // over the whole grid (every wave derives the same m0 from the table)
    auto rows = [&](auto cs_tag) {
    constexpr bool CS = decltype(cs_tag)::value;
    if (CS) {
        const f2 M0 = f2{m0, m0};
#pragma unroll
        for (int p = 0; p < KP; ++p) cc[p] = cc[p] - M0;
        cs -= m0;
    }
    const float eps_scale = __builtin_amdgcn_exp2f(-m0);
    float lacc = 0.f;                      // sum of log2(den) of the rows since the last flush
    float x0 = 0.f, x1 = 0.f, x2 = 0.f;
    if (r0 < r1) { const float* xp = X + 3 * r0; x0 = xp[0]; x1 = xp[1]; x2 = xp[2]; }
    for (int64_t row = r0; row < r1; ++row) {
        const int64_t nrow = (row + 1 < r1) ? row + 1 : row;
        const float* xn = X + 3 * nrow;
        const float nx0 = xn[0], nx1 = xn[1], nx2 = xn[2];
        const f2 X0 = f2{x0, x0}, X1 = f2{x1, x1}, X2 = f2{x2, x2};
        const float xo0 = x0 - o0, xo1 = x1 - o1, xo2 = x2 - o2;
        const f2 XO0 = f2{xo0, xo0}, XO1 = f2{xo1, xo1}, XO2 = f2{xo2, xo2};

        f2 wl[KP + 1];
        f2 q0[KP + 1], q1[KP + 1], q2[KP + 1];
        float q0s = 0.f, q1s = 0.f, q2s = 0.f;
        float wls = NEG_INF;
        float m = NEG_INF;
        f2 sacc = f2{0.f, 0.f};
#pragma unroll
        for (int p = 0; p < KP; ++p) {
            const f2 d0 = X0 - mu0[p], d1 = X1 - mu1[p], d2 = X2 - mu2[p];
            q0[p] = d0 * d0; q1[p] = d1 * d1; q2[p] = d2 * d2;
            f2 a = cc[p] - g0[p] * q0[p];
            a = a - g1[p] * q1[p];
            a = a - g2[p] * q2[p];
            if (CS) {
                wl[p] = f2{__builtin_amdgcn_exp2f(a.x), __builtin_amdgcn_exp2f(a.y)};
                sacc += wl[p];
            } else {
                wl[p] = a;
                m = fmaxf(m, fmaxf(a.x, a.y));
            }
        }
        if (ODD) {
            const float d0 = x0 - mu0s, d1 = x1 - mu1s, d2 = x2 - mu2s;
            q0s = d0 * d0; q1s = d1 * d1; q2s = d2 * d2;
            float a = fmaf(-g0s, q0s, cs);
            a = fmaf(-g1s, q1s, a);
            a = fmaf(-g2s, q2s, a);
            wls = CS ? __builtin_amdgcn_exp2f(a) : a;
            m = fmaxf(m, a);
        }
        if (CS) {
            m = m0;
        } else {
            m = wave_max_dpp(m);
            if (m == NEG_INF) m = 0.f;
            const f2 M = f2{m, m};
#pragma unroll
            for (int p = 0; p < KP; ++p) {
                const f2 t = wl[p] - M;
                wl[p] = f2{__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)};
                sacc += wl[p];
            }
            if (ODD) wls = __builtin_amdgcn_exp2f(wls - m);
        }
        float s = sacc.x + sacc.y;
        if (ODD) s += wls;
        s = wave_sum_dpp(s);
        float inv_den;
        if (CS) {
            // lpn2_from with m = m0 in [-64, CS_MAX_SHIFT]: no "tiny" case, eps 2^-m0 is loop-invariant;
            // the log-normalisers are summed in float32 over 8 rows before they join the float64 total
            const float den = fmaf(FLAT_EPS, eps_scale, s);
            inv_den = __builtin_amdgcn_rcpf(den);
            lacc += __builtin_amdgcn_logf(den);
            if (((row - r0) & 7) == 7) {
                asm volatile("" ::: "memory");          // keeps this a (wave-uniform) branch, not selects
                lsum += (double)lacc;
                lacc = 0.f;
            }
        } else {
            const float lpn2 = lpn2_from(m, s, inv_den);
            lsum += (double)(lpn2 * LN2);
        }
        const f2 INV = f2{inv_den, inv_den};
#pragma unroll
        for (int p = 0; p < KP; ++p) {
            const f2 rr = wl[p] * INV;
            s0[p] += rr;
            a0[p] += rr * XO0; a1[p] += rr * XO1; a2[p] += rr * XO2;
            b0[p] += rr * q0[p]; b1[p] += rr * q1[p]; b2[p] += rr * q2[p];
        }
        if (ODD) {
            const float rr = wls * inv_den;
            s0s += rr;
            a0s = fmaf(rr, xo0, a0s); a1s = fmaf(rr, xo1, a1s); a2s = fmaf(rr, xo2, a2s);
            b0s = fmaf(rr, q0s, b0s); b1s = fmaf(rr, q1s, b1s); b2s = fmaf(rr, q2s, b2s);
        }
        x0 = nx0; x1 = nx1; x2 = nx2;
    }
    if (CS) lsum = (lsum + (double)lacc + (double)(r1 > r0 ? r1 - r0 : 0) * (double)m0) * (double)LN2;
    };
    if (small_shift) rows(std::true_type{});
    else rows(std::false_type{});
    // first moments: from the common origin to the component's mean
    {
        const f2 O0 = f2{o0, o0}, O1 = f2{o1, o1}, O2 = f2{o2, o2};
#pragma unroll
        for (int p = 0; p < KP; ++p) {
            a0[p] -= (mu0[p] - O0) * s0[p]; a1[p] -= (mu1[p] - O1) * s0[p]; a2[p] -= (mu2[p] - O2) * s0[p];
        }
        if (ODD) {
            a0s = fmaf(-(mu0s - o0), s0s, a0s); a1s = fmaf(-(mu1s - o1), s0s, a1s); a2s = fmaf(-(mu2s - o2), s0s, a2s);
        }
    }

    __shared__ float sh[FLAT_NSTAT * NSLOT * 64];
    __shared__ double shl[WAVES_PER_BLOCK];
    const int w = wave_in_block();
    constexpr int ST = NSLOT * 64;
    if (lane == 0) shl[w] = lsum;
    auto put = [&](int k, bool first, float v0, float v1, float v2, float v3, float v4, float v5, float v6) {
        float* q = sh + k * 64 + lane;
        if (first) { q[0 * ST] = v0; q[1 * ST] = v1; q[2 * ST] = v2; q[3 * ST] = v3; q[4 * ST] = v4; q[5 * ST] = v5; q[6 * ST] = v6; }
        else { q[0 * ST] += v0; q[1 * ST] += v1; q[2 * ST] += v2; q[3 * ST] += v3; q[4 * ST] += v4; q[5 * ST] += v5; q[6 * ST] += v6; }
    };
    for (int turn = 0; turn < WAVES_PER_BLOCK; ++turn) {
        if (w == turn) {
#pragma unroll
            for (int p = 0; p < KP; ++p) {
                put(2 * p, turn == 0, s0[p].x, a0[p].x, a1[p].x, a2[p].x, b0[p].x, b1[p].x, b2[p].x);
                put(2 * p + 1, turn == 0, s0[p].y, a0[p].y, a1[p].y, a2[p].y, b0[p].y, b1[p].y, b2[p].y);
            }
            if (ODD) put(K - 1, turn == 0, s0s, a0s, a1s, a2s, b0s, b1s, b2s);
        }
        __syncthreads();
    }
    float* outp = partials + (size_t)blockIdx.x * FLAT_NSTAT * Jpad;
    for (int idx = threadIdx.x; idx < FLAT_NSTAT * ST; idx += BLOCK) {
        const int st = idx / ST, j = idx % ST;
        outp[st * Jpad + j] = sh[idx];
    }
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int i = 0; i < WAVES_PER_BLOCK; ++i) t += shl[i];
        lpn_partials[blockIdx.x] = t;
    }
}

// ------------------------------------------------------------------------------------------
// M-step moments from a materialised responsibility matrix (m_step(X, resp))
// ------------------------------------------------------------------------------------------
template <int NV4, int NV1, bool ISLOG, bool NTLOAD>
__global__ __launch_bounds__(BLOCK) void flat_mstep_kernel(
    const float* __restrict__ X, const float* __restrict__ resp,
    const float* __restrict__ hint /*[3][Jpad]*/, int64_t n, int J, int Jpad,
    float* __restrict__ partials, int64_t ld, int round_robin) {
    // resp / hint / partials point at this launch's first column; J = valid columns from there,
    // ld = row stride of resp (the full component count).  Components are paired into float2 like
    // in the fused kernel (pairs (4v, 4v+1), (4v+2, 4v+3), pairs of scalar slots, one single).
    using L = Layout<NV4, NV1>;
    constexpr int K = L::K;
    constexpr int KP = 2 * NV4 + NV1 / 2;
    constexpr bool ODD = (NV1 & 1) != 0;
    const int lane = lane_id();
    f2 c0[KP + 1], c1[KP + 1], c2[KP + 1];
    f2 s0[KP + 1], a0[KP + 1], a1[KP + 1], a2[KP + 1], b0[KP + 1], b1[KP + 1], b2[KP + 1];
    float c0s = 0.f, c1s = 0.f, c2s = 0.f, s0s = 0.f, a0s = 0.f, a1s = 0.f, a2s = 0.f, b0s = 0.f, b1s = 0.f, b2s = 0.f;
#pragma unroll
    for (int p = 0; p < KP; ++p) {
        const int ja = L::j_of(2 * p, lane), jb = L::j_of(2 * p + 1, lane);
        c0[p] = f2{hint[0 * Jpad + ja], hint[0 * Jpad + jb]};
        c1[p] = f2{hint[1 * Jpad + ja], hint[1 * Jpad + jb]};
        c2[p] = f2{hint[2 * Jpad + ja], hint[2 * Jpad + jb]};
        s0[p] = a0[p] = a1[p] = a2[p] = b0[p] = b1[p] = b2[p] = f2{0.f, 0.f};
    }
    if (ODD) {
        const int j = L::j_of(K - 1, lane);
        c0s = hint[0 * Jpad + j]; c1s = hint[1 * Jpad + j]; c2s = hint[2 * Jpad + j];
    }
    // rows of this wave: base + it * stride (contiguous range, or dealt round-robin so that the
    // waves in flight read one contiguous window of the matrix)
    int64_t base, stride, cnt;
    {
        const int64_t nw = (int64_t)gridDim.x * WAVES_PER_BLOCK;
        const int64_t gw = (int64_t)blockIdx.x * WAVES_PER_BLOCK + wave_in_block();
        if (round_robin) {
            base = gw; stride = nw; cnt = (gw < n) ? (n - gw + nw - 1) / nw : 0;
        } else {
            int64_t r0, r1;
            wave_row_range(n, r0, r1);
            base = r0; stride = 1; cnt = r1 - r0;
        }
    }

    constexpr bool is_log = ISLOG;
    // Every load of the row loop is unconditional: a lane whose column is past J reads column 0 instead (its sums are
    // zeroed on the way out), and the look-ahead past a wave's last row re-reads that row.  With conditional loads the
    // compiler cannot count what is in flight and waits with vmcnt(0) -- behind the NEXT row's loads -- before touching the
    // current row, which serialises each row's load and arithmetic (round 4: 0.526 -> 0.48 ms together with the
    // non-temporal loads, which a pure read of the same matrix shows to be worth 6.5 -> 7.1 TB/s: profiles/r04/read_probe.log).
    int off4[NV4 > 0 ? NV4 : 1], off1[NV1 > 0 ? NV1 : 1];
#pragma unroll
    for (int s = 0; s < NV4; ++s) { const int jb = (s * 64 + lane) * 4; off4[s] = (jb < J) ? jb : 0; }
#pragma unroll
    for (int s = 0; s < NV1; ++s) { const int j = 256 * NV4 + s * 64 + lane; off1[s] = (j < J) ? j : 0; }
    auto load_row = [&](int64_t row, float (&v)[K]) {
        const float* in = resp + row * ld;
#pragma unroll
        for (int s = 0; s < NV4; ++s) {
            typedef float f4v __attribute__((ext_vector_type(4)));
            const f4v* q = reinterpret_cast<const f4v*>(in + off4[s]);
            const f4v t = NTLOAD ? __builtin_nontemporal_load(q) : *q;
            v[s * 4 + 0] = t.x; v[s * 4 + 1] = t.y; v[s * 4 + 2] = t.z; v[s * 4 + 3] = t.w;
        }
#pragma unroll
        for (int s = 0; s < NV1; ++s) v[4 * NV4 + s] = NTLOAD ? __builtin_nontemporal_load(in + off1[s]) : in[off1[s]];
    };
    auto accumulate = [&](const float (&v)[K], float x0, float x1, float x2) {
        const f2 X0 = f2{x0, x0}, X1 = f2{x1, x1}, X2 = f2{x2, x2}, L2 = f2{LOG2E, LOG2E};
#pragma unroll
        for (int p = 0; p < KP; ++p) {
            f2 r = f2{v[2 * p], v[2 * p + 1]};
            if (is_log) {
                const f2 t = r * L2;
                r = f2{__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)};
            }
            const f2 d0 = X0 - c0[p], d1 = X1 - c1[p], d2 = X2 - c2[p];
            const f2 rd0 = r * d0, rd1 = r * d1, rd2 = r * d2;
            s0[p] += r;
            a0[p] += rd0; a1[p] += rd1; a2[p] += rd2;
            b0[p] += rd0 * d0; b1[p] += rd1 * d1; b2[p] += rd2 * d2;
        }
        if (ODD) {
            const float r = is_log ? __builtin_amdgcn_exp2f(v[K - 1] * LOG2E) : v[K - 1];
            const float d0 = x0 - c0s, d1 = x1 - c1s, d2 = x2 - c2s;
            const float rd0 = r * d0, rd1 = r * d1, rd2 = r * d2;
            s0s += r;
            a0s += rd0; a1s += rd1; a2s += rd2;
            b0s = fmaf(rd0, d0, b0s); b1s = fmaf(rd1, d1, b1s); b2s = fmaf(rd2, d2, b2s);
        }
    };
    // Two row buffers used in turn (no register copies), rows in order, so the sums are those of the plain loop bit for
    // bit.  (Round 4, measured and dropped: a StorePacer in front of every row load, 0.55 - 0.56 ms at every target
    // rate: profiles/r04/mstep_pace.log.)
    if (cnt > 0) {
        float bufa[K], bufb[K];
        load_row(base, bufa);
        const float* xp = X + 3 * base;
        float xa0 = xp[0], xa1 = xp[1], xa2 = xp[2];
        int64_t it = 0;
        for (; it + 2 <= cnt; it += 2) {
            const int64_t rb = base + (it + 1) * stride;
            __builtin_amdgcn_sched_barrier(0);
            load_row(rb, bufb);
            const float* xq = X + 3 * rb;
            const float xb0 = xq[0], xb1 = xq[1], xb2 = xq[2];
            __builtin_amdgcn_sched_barrier(0);   // (the scheduler otherwise sinks the loads to just before their use)
            accumulate(bufa, xa0, xa1, xa2);
            __builtin_amdgcn_sched_barrier(0);
            const int64_t ra = base + ((it + 2 < cnt) ? it + 2 : cnt - 1) * stride;
            load_row(ra, bufa);
            const float* xr = X + 3 * ra;
            xa0 = xr[0]; xa1 = xr[1]; xa2 = xr[2];
            __builtin_amdgcn_sched_barrier(0);
            accumulate(bufb, xb0, xb1, xb2);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (it < cnt) accumulate(bufa, xa0, xa1, xa2);
    }
    __shared__ float sh[FLAT_NSTAT * L::CAP];
    const int w = wave_in_block();
    constexpr int ST = L::CAP;
    auto put = [&](int k, bool first, float v0, float v1, float v2, float v3, float v4, float v5, float v6) {
        float* q = sh + L::j_of(k, lane);
        if (L::j_of(k, lane) >= J) v0 = v1 = v2 = v3 = v4 = v5 = v6 = 0.f;   // (that lane summed column 0's values)
        if (first) { q[0 * ST] = v0; q[1 * ST] = v1; q[2 * ST] = v2; q[3 * ST] = v3; q[4 * ST] = v4; q[5 * ST] = v5; q[6 * ST] = v6; }
        else { q[0 * ST] += v0; q[1 * ST] += v1; q[2 * ST] += v2; q[3 * ST] += v3; q[4 * ST] += v4; q[5 * ST] += v5; q[6 * ST] += v6; }
    };
    for (int turn = 0; turn < WAVES_PER_BLOCK; ++turn) {
        if (w == turn) {
#pragma unroll
            for (int p = 0; p < KP; ++p) {
                put(2 * p, turn == 0, s0[p].x, a0[p].x, a1[p].x, a2[p].x, b0[p].x, b1[p].x, b2[p].x);
                put(2 * p + 1, turn == 0, s0[p].y, a0[p].y, a1[p].y, a2[p].y, b0[p].y, b1[p].y, b2[p].y);
            }
            if (ODD) put(K - 1, turn == 0, s0s, a0s, a1s, a2s, b0s, b1s, b2s);
        }
        __syncthreads();
    }
    float* outp = partials + (size_t)blockIdx.x * FLAT_NSTAT * Jpad;
    for (int idx = threadIdx.x; idx < FLAT_NSTAT * ST; idx += BLOCK) {
        const int st = idx / ST, j = idx % ST;
        outp[st * Jpad + j] = sh[idx];
    }
}

// ------------------------------------------------------------------------------------------
// Large J (> 1024): the same register-resident mapping, applied to chunks of CH_J = 832
// components (13 slots of 64).  A row's log-sum-exp is assembled from per-chunk (max, sum) pairs,
// after which every chunk is revisited with the row's final normaliser:
//   flat_chunk_lse_kernel      per chunk: m_c = max_j wl2, s_c = sum_j 2^(wl2 - m_c), arg-max
//   flat_chunk_combine_kernel  per row: lpn2 = log2(sum_c s_c 2^(m_c) + eps), arg-max over chunks
//   flat_chunk_fused_kernel    per chunk: r = 2^(wl2 - lpn2) (no reductions), statistics
//   flat_chunk_write_kernel    per chunk: log_resp[:, chunk] = (wl2 - lpn2) ln 2
// ------------------------------------------------------------------------------------------
constexpr int CH_SLOTS = 13;
constexpr int CH_J = CH_SLOTS * 64;

__global__ __launch_bounds__(BLOCK) void flat_chunk_lse_kernel(
    const float* __restrict__ X, const float* __restrict__ pack /*at the chunk's first column*/,
    int64_t n, int jvalid, int jbase, int Jpad, float* __restrict__ cm, float* __restrict__ cs,
    int* __restrict__ ca) {
    constexpr int K = CH_SLOTS;
    const int lane = lane_id();
    LaneParams<0, CH_SLOTS> P;
    P.load(pack, Jpad, lane);
    int64_t r0, r1;
    wave_row_range(n, r0, r1);
    float keep_m = 0.f, keep_s = 0.f;
    int keep_a = 0;
    float x0 = 0.f, x1 = 0.f, x2 = 0.f;
    if (r0 < r1) { const float* xp = X + 3 * r0; x0 = xp[0]; x1 = xp[1]; x2 = xp[2]; }
    for (int64_t row = r0; row < r1; ++row) {
        const int64_t nrow = (row + 1 < r1) ? row + 1 : row;
        const float* xn = X + 3 * nrow;
        const float nx0 = xn[0], nx1 = xn[1], nx2 = xn[2];
        float wl[K];
        float m = row_wl2<0, CH_SLOTS>(P, x0, x1, x2, wl);
        m = wave_max_dpp(m);
        float s = 0.f;
        if (m != NEG_INF) {
#pragma unroll
            for (int k = 0; k < K; ++k) s += __builtin_amdgcn_exp2f(wl[k] - m);
            s = wave_sum_dpp(s);
        }
        const int slot = (int)((row - r0) & 63);
        if (ca) {
            int best = 0x7fffffff;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const int j = k * 64 + lane;
                if (wl[k] == m && j < jvalid && j < best) best = j;
            }
            best = wave_reduce_i(best, OpMinI());
            if (lane == slot) keep_a = (best == 0x7fffffff) ? jbase : jbase + best;
        }
        if (lane == slot) { keep_m = m; keep_s = s; }
        if (slot == 63 || row + 1 == r1) {
            const int64_t base = row - slot;
            if (lane <= slot) {
                cm[base + lane] = keep_m;
                cs[base + lane] = keep_s;
                if (ca) ca[base + lane] = keep_a;
            }
        }
        x0 = nx0; x1 = nx1; x2 = nx2;
    }
}

__global__ __launch_bounds__(256) void flat_chunk_combine_kernel(
    const float* __restrict__ cm, const float* __restrict__ cs, const int* __restrict__ ca, int nchunks,
    int64_t n, float* __restrict__ lpn2_out, float* __restrict__ lpn_out, int32_t* __restrict__ argmax_out,
    double* __restrict__ lpn_partials) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    double lp = 0.0;
    if (i < n) {
        float M = NEG_INF;
        int best_c = 0;
        for (int c = 0; c < nchunks; ++c) {
            const float m = cm[(size_t)c * n + i];
            if (m > M) { M = m; best_c = c; }           // strict >: first chunk wins ties (numpy argmax)
        }
        float S = 0.f;
        if (M == NEG_INF) M = 0.f;
        else
            for (int c = 0; c < nchunks; ++c) {
                const float m = cm[(size_t)c * n + i];
                if (m != NEG_INF) S += cs[(size_t)c * n + i] * __builtin_amdgcn_exp2f(m - M);
            }
        float inv_den;
        const float lpn2 = lpn2_from(M, S, inv_den);
        lpn2_out[i] = lpn2;
        const float lpn = lpn2 * LN2;
        if (lpn_out) lpn_out[i] = lpn;
        if (argmax_out) argmax_out[i] = ca ? ca[(size_t)best_c * n + i] : 0;
        lp = (double)lpn;
    }
    if (lpn_partials) {
        __shared__ double sh[4];
        lp = wave_sum_f64(lp);
        if (lane_id() == 0) sh[wave_in_block()] = lp;
        __syncthreads();
        if (threadIdx.x == 0) lpn_partials[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
    }
}

__global__ __launch_bounds__(BLOCK) void flat_chunk_fused_kernel(
    const float* __restrict__ X, const float* __restrict__ pack, int64_t n, int Jpad,
    const float* __restrict__ lpn2, float* __restrict__ partials /*at the chunk's first column*/,
    const int* __restrict__ done_flag) {
    if (done_flag && *done_flag) return;
    constexpr int K = CH_SLOTS;
    const int lane = lane_id();
    LaneParams<0, CH_SLOTS> P;
    P.load(pack, Jpad, lane);
    float a_s0[K], a_a0[K], a_a1[K], a_a2[K], a_b0[K], a_b1[K], a_b2[K];
#pragma unroll
    for (int k = 0; k < K; ++k) a_s0[k] = a_a0[k] = a_a1[k] = a_a2[k] = a_b0[k] = a_b1[k] = a_b2[k] = 0.f;
    int64_t r0, r1;
    wave_row_range(n, r0, r1);
    float x0 = 0.f, x1 = 0.f, x2 = 0.f, l2 = 0.f;
    if (r0 < r1) { const float* xp = X + 3 * r0; x0 = xp[0]; x1 = xp[1]; x2 = xp[2]; l2 = lpn2[r0]; }
    for (int64_t row = r0; row < r1; ++row) {
        const int64_t nrow = (row + 1 < r1) ? row + 1 : row;
        const float* xn = X + 3 * nrow;
        const float nx0 = xn[0], nx1 = xn[1], nx2 = xn[2], nl2 = lpn2[nrow];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const float d0 = x0 - P.mu0[k], d1 = x1 - P.mu1[k], d2 = x2 - P.mu2[k];
            float a = fmaf(-(d0 * P.g0[k]), d0, P.c[k] - l2);
            a = fmaf(-(d1 * P.g1[k]), d1, a);
            a = fmaf(-(d2 * P.g2[k]), d2, a);
            const float rr = __builtin_amdgcn_exp2f(a);      // r = 2^(wl2 - lpn2) = exp(wlp - lpn)
            const float rd0 = rr * d0, rd1 = rr * d1, rd2 = rr * d2;
            a_s0[k] += rr;
            a_a0[k] += rd0; a_a1[k] += rd1; a_a2[k] += rd2;
            a_b0[k] = fmaf(rd0, d0, a_b0[k]);
            a_b1[k] = fmaf(rd1, d1, a_b1[k]);
            a_b2[k] = fmaf(rd2, d2, a_b2[k]);
        }
        x0 = nx0; x1 = nx1; x2 = nx2; l2 = nl2;
    }
    __shared__ float sh[FLAT_NSTAT * CH_J];
    const int w = wave_in_block();
    for (int turn = 0; turn < WAVES_PER_BLOCK; ++turn) {
        if (w == turn) {
#pragma unroll
            for (int k = 0; k < K; ++k) {
                float* p = sh + k * 64 + lane;
                constexpr int ST = CH_J;
                if (turn == 0) {
                    p[0 * ST] = a_s0[k]; p[1 * ST] = a_a0[k]; p[2 * ST] = a_a1[k]; p[3 * ST] = a_a2[k];
                    p[4 * ST] = a_b0[k]; p[5 * ST] = a_b1[k]; p[6 * ST] = a_b2[k];
                } else {
                    p[0 * ST] += a_s0[k]; p[1 * ST] += a_a0[k]; p[2 * ST] += a_a1[k]; p[3 * ST] += a_a2[k];
                    p[4 * ST] += a_b0[k]; p[5 * ST] += a_b1[k]; p[6 * ST] += a_b2[k];
                }
            }
        }
        __syncthreads();
    }
    float* outp = partials + (size_t)blockIdx.x * FLAT_NSTAT * Jpad;
    for (int idx = threadIdx.x; idx < FLAT_NSTAT * CH_J; idx += BLOCK) {
        const int st = idx / CH_J, j = idx % CH_J;
        outp[st * Jpad + j] = sh[idx];
    }
}

// out[row, jbase + j] = (wl2 - lpn2[row]) ln 2   (lpn2 == nullptr: raw weighted log-probabilities)
__global__ __launch_bounds__(BLOCK) void flat_chunk_write_kernel(
    const float* __restrict__ X, const float* __restrict__ pack, int64_t n, int jvalid, int Jpad,
    const float* __restrict__ lpn2, float* __restrict__ out /*at the chunk's first column*/, int64_t ld) {
    constexpr int K = CH_SLOTS;
    const int lane = lane_id();
    LaneParams<0, CH_SLOTS> P;
    P.load(pack, Jpad, lane);
    int64_t r0, r1;
    wave_row_range(n, r0, r1);
    const bool nt_rows = (ld & 3) == 0;               // (rows of whole 16-byte pieces: see launch_estep)
    for (int64_t row = r0; row < r1; ++row) {
        const float* xp = X + 3 * row;
        const float x0 = xp[0], x1 = xp[1], x2 = xp[2];
        const float l = lpn2 ? lpn2[row] * LN2 : 0.f;
        float wl[K];
        (void)row_wl2<0, CH_SLOTS>(P, x0, x1, x2, wl);
        float* o = out + row * ld;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int j = k * 64 + lane;
            if (j < jvalid) {
                const float v = fmaf(wl[k], LN2, -l);
                if (nt_rows) __builtin_nontemporal_store(v, o + j); else o[j] = v;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// second stage: fp64 sum of the per-workgroup partials -> stats[7][Jpad], sum lpn, n
// ------------------------------------------------------------------------------------------
constexpr int RED_IDX = 32;      // consecutive statistics per workgroup (one 128-byte line)
constexpr int RED_SLICES = 32;   // partial-block slices summed in parallel, combined in fixed order
constexpr int RED_BATCH = 16;    // loads a thread has in flight: 512 workgroups' partials = ONE round of loads
// This kernel sits between two EM iterations and is nothing but latency: every load is issued before the first
// add (index clamped, value masked -- no branch between the loads), the stop flag is fetched alongside and only
// gates the store.
__global__ __launch_bounds__(RED_IDX * RED_SLICES) void flat_reduce_kernel(
    const float* __restrict__ partials, const double* __restrict__ lpn_partials, int nblocks,
    int n_lpn_blocks, int valid_j, int Jpad, double n_local, double* __restrict__ stats,
    const int* __restrict__ done_flag) {
    // (an atomic load is a VECTOR load: it is waited for where `done` is used, a scalar load would be waited for
    //  at the next kernel-argument fetch, i.e. before the partials' loads are even issued)
    // (fetched through a per-lane address the compiler cannot prove uniform: a uniform value would be moved to an
    //  SGPR -- and waited for -- on the spot)
    int lane_zero = 0;
    asm volatile("" : "+v"(lane_zero));
    const int done = __hip_atomic_load(done_flag + lane_zero, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // never null
    const int total = FLAT_NSTAT * Jpad;
    const int stat_blocks = (total + RED_IDX - 1) / RED_IDX;
    __shared__ double sh[RED_SLICES][RED_IDX];
    if ((int)blockIdx.x < stat_blocks) {
        const int li = threadIdx.x % RED_IDX, slice = threadIdx.x / RED_IDX;
        const int idx = blockIdx.x * RED_IDX + li;
        double acc = 0.0;
        if (idx < total && (idx % Jpad) < valid_j) {
            const float* src = partials + idx;
            for (int b0 = slice; b0 < nblocks; b0 += RED_BATCH * RED_SLICES) {
                float v[RED_BATCH];
#pragma unroll
                for (int u = 0; u < RED_BATCH; ++u) {
                    const int b = b0 + u * RED_SLICES;
                    v[u] = src[(size_t)(b < nblocks ? b : nblocks - 1) * total];
                }
#pragma unroll
                for (int u = 0; u < RED_BATCH; ++u) acc += (b0 + u * RED_SLICES < nblocks) ? (double)v[u] : 0.0;
            }
        }
        sh[slice][li] = acc;
        __syncthreads();
        if (slice == 0 && idx < total && !done) {
            double t = 0.0;
#pragma unroll
            for (int s = 0; s < RED_SLICES; ++s) t += sh[s][li];
            stats[idx] = t;
        }
    } else {
        // last workgroup: sum of the per-workgroup log-normaliser partials + the point count
        double acc = 0.0;
        if (lpn_partials)
            for (int b = threadIdx.x; b < n_lpn_blocks; b += RED_IDX * RED_SLICES) acc += lpn_partials[b];
        acc = wave_sum_f64(acc);
        double* shw = &sh[0][0];
        if (lane_id() == 0) shw[wave_in_block()] = acc;
        __syncthreads();
        if (threadIdx.x == 0 && !done) {
            double t = 0.0;
            for (int w = 0; w < RED_IDX * RED_SLICES / 64; ++w) t += shw[w];
            stats[total] = t;
            stats[total + 1] = n_local;
        }
    }
}

// ------------------------------------------------------------------------------------------
// M-step in fp64 from centred statistics (single workgroup, Jpad <= 1024 threads).
//   Sx = a + c s0,  Sxx = b + 2 c a + c^2 s0   (c = centre the statistics were taken about)
//   W (gmm_waymo gmm_impl.py:81-103,134): nk = s0 + eps; mu = Sx/nk; cov = Sxx/nk - mu^2 + 1e-6;
//        spherical: mean over axes; w = nk / N; inv_std = 1/(sqrt(cov + 1e-6) + eps)
//   G (gmmreg_gpu gmm_impl.py:46-52,74): nk = s0; mu = Sx/(nk+eps); cov = max(Sxx/(nk+eps) - mu^2, 0);
//        w = nk / N; inv_std = 1/(sqrt(cov) + eps)
// ctl: [0] done, [1] n_iter, [2] converged; prev_ll lives in ctl_f[0].
// ------------------------------------------------------------------------------------------
struct FlatComponent { float mu[3], cov[3], inv[3], w; };

// the new parameters of component j, rounded to the reference's storage type (float32) before inv_std is derived
__device__ inline FlatComponent finalize_values(int j, const double* __restrict__ stats,
                                                const float* __restrict__ centre, int Jpad, int cov_type, int variant) {
    const double eps = (double)FLAT_EPS;
    const double n_total = stats[FLAT_NSTAT * Jpad + 1];
    const double s0 = stats[0 * Jpad + j];
    double st_a[3], st_b[3], ce[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {                      // every load before the first divide
        ce[d] = (double)centre[d * Jpad + j];
        st_a[d] = stats[(1 + d) * Jpad + j];
        st_b[d] = stats[(4 + d) * Jpad + j];
    }
    double nmu[3], ncov[3];
    const double nk = (variant == HGMM_VARIANT_W) ? s0 + eps : s0;
    const double den = (variant == HGMM_VARIANT_W) ? nk : nk + eps;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const double c = ce[d], a = st_a[d], b = st_b[d];
        const double sx = a + c * s0;
        const double sxx = b + 2.0 * c * a + c * c * s0;
        const double m = sx / den;
        double v = sxx / den - m * m;
        if (variant == HGMM_VARIANT_W) v += 1e-6; else v = v < 0.0 ? 0.0 : v;
        nmu[d] = m;
        ncov[d] = v;
    }
    FlatComponent r;
    if (cov_type == HGMM_COV_SPHERICAL) {
        const float sph = (float)((ncov[0] + ncov[1] + ncov[2]) / 3.0);
        r.cov[0] = r.cov[1] = r.cov[2] = sph;
    } else {
#pragma unroll
        for (int d = 0; d < 3; ++d) r.cov[d] = (float)ncov[d];
    }
#pragma unroll
    for (int d = 0; d < 3; ++d) r.mu[d] = (float)nmu[d];
    r.w = (float)(nk / n_total);
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const double cv = (double)r.cov[d];
        r.inv[d] = (float)((variant == HGMM_VARIANT_W) ? 1.0 / (sqrt(cv + 1e-6) + eps) : 1.0 / (sqrt(cv) + eps));
    }
    return r;
}
__device__ inline void store_component(int j, const FlatComponent& r, int cov_type, float* mu, float* cov, float* w,
                                       float* inv) {
    if (cov_type == HGMM_COV_SPHERICAL) cov[j] = r.cov[0];
    else { cov[3 * j + 0] = r.cov[0]; cov[3 * j + 1] = r.cov[1]; cov[3 * j + 2] = r.cov[2]; }
    mu[3 * j + 0] = r.mu[0]; mu[3 * j + 1] = r.mu[1]; mu[3 * j + 2] = r.mu[2];
    w[j] = r.w;
    if (inv) {
        if (cov_type == HGMM_COV_SPHERICAL) inv[j] = r.inv[0];
        else { inv[3 * j + 0] = r.inv[0]; inv[3 * j + 1] = r.inv[1]; inv[3 * j + 2] = r.inv[2]; }
    }
}
__device__ inline void finalize_component(int j, const double* __restrict__ stats,
                                          const float* __restrict__ centre, int J, int Jpad, int cov_type,
                                          int variant, float* mu, float* cov, float* w, float* inv) {
    store_component(j, finalize_values(j, stats, centre, Jpad, cov_type, variant), cov_type, mu, cov, w, inv);
}

__device__ inline void ctl_update(const double* __restrict__ stats, int Jpad, float* lls, int lls_cap,
                                  float tol, int* ctl, float* ctl_f, int* stop_out) {
    const double n_total = stats[FLAT_NSTAT * Jpad + 1];
    const float ll = (float)(stats[FLAT_NSTAT * Jpad] / n_total);
    const int it = ctl[1];
    if (it < lls_cap) lls[it] = ll;
    const float change = ll - ctl_f[0];          // prev starts at -inf (gmm_impl.py:120)
    ctl_f[0] = ll;
    ctl[1] = it + 1;
    if (fabsf(change) < tol) { *stop_out = 1; ctl[2] = 1; }
}

// M-step + next packed table + stop rule in one launch (Jpad <= 1024: Jpad / 256 workgroups, a component per thread).
// Like the reduction this is a latency chain between two iterations: the stop flag is fetched WITH the statistics
// and only gates the stores, the packed table is formed from the registers that hold the new parameters (no store
// -> barrier -> re-load), and the ~800 fp64 instructions per component (divides, square roots, four logarithms)
// are spread over several CUs instead of queueing 16 waves deep on one.
// The stop flag is double-buffered by iteration parity: the launch of iteration k reads `stop` = flag[k & 1] -- which
// nothing in this launch writes -- and thread 0 records its verdict in `stop_next` = flag[(k + 1) & 1], the word
// the kernels of iteration k + 1 look at.  (With ONE flag a workgroup that starts late could see the verdict of its
// own launch and skip its share of the update.)
__global__ __launch_bounds__(256) void flat_finalize_kernel(const double* __restrict__ stats,
                                     const float* __restrict__ centre /*[3][Jpad] or pack mu rows*/,
                                     int J, int Jpad, int cov_type, int variant,
                                     float* mu, float* cov, float* w, float* inv, float* pack,
                                     float* lls, int lls_cap, float tol, int* ctl, float* ctl_f,
                                     const int* stop /*never null*/, int* stop_next /*null: no device loop*/) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int done = __hip_atomic_load(stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // vector load, see flat_reduce_kernel
    FlatComponent r = {};
    if (j < J) r = finalize_values(j, stats, centre, Jpad, cov_type, variant);
    if (done) {
        if (j == 0 && stop_next) *stop_next = 1;           // stays stopped
        return;
    }
    if (j < J) store_component(j, r, cov_type, mu, cov, w, inv);
    if (pack && j < Jpad) pack_values(j, j < J, Jpad, variant, r.mu, r.inv, r.w, pack);
    if (j == 0 && ctl) ctl_update(stats, Jpad, lls, lls_cap, tol, ctl, ctl_f, stop_next);
}

// many workgroups (large J): the stop rule moves to flat_ctl_kernel, launched afterwards, so no
// workgroup can observe a `done` flag written during this launch
__global__ void flat_finalize_mb_kernel(const double* __restrict__ stats, const float* __restrict__ centre,
                                        int J, int Jpad, int cov_type, int variant, float* mu, float* cov,
                                        float* w, float* inv, float* pack, const int* ctl) {
    if (ctl && ctl[0]) return;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < J) finalize_component(j, stats, centre, J, Jpad, cov_type, variant, mu, cov, w, inv);
    if (pack && j < Jpad) pack_component(j, J, Jpad, cov_type, variant, mu, inv, w, pack);
}
__global__ void flat_ctl_kernel(const double* __restrict__ stats, int Jpad, float* lls, int lls_cap, float tol,
                                int* ctl, float* ctl_f) {
    if (ctl[0]) return;
    ctl_update(stats, Jpad, lls, lls_cap, tol, ctl, ctl_f, ctl);
}


// hint table [3][Jpad] from host-layout mu [J,3]
__global__ void flat_hint_kernel(const float* mu, int J, int Jpad, float* hint) {
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= Jpad) return;
    for (int d = 0; d < 3; ++d) hint[d * Jpad + j] = (j < J) ? mu[3 * j + d] : 0.f;
}
__global__ void flat_hint_const_kernel(float c0, float c1, float c2, int Jpad, float* hint) {
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= Jpad) return;
    hint[0 * Jpad + j] = c0; hint[1 * Jpad + j] = c1; hint[2 * Jpad + j] = c2;
}

// ------------------------------------------------------------------------------------------
// host-side launch logic
// ------------------------------------------------------------------------------------------
static int round_up(int v, int m) { return (v + m - 1) / m * m; }

static int grid_for(hgmm_ctx* c, int64_t n, int blocks_per_cu) {
    int64_t want = (n + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK;     // one row per wave at least
    if (blocks_per_cu < 1) blocks_per_cu = 1;
    if (blocks_per_cu > 8) blocks_per_cu = 8;
    int64_t cap = (int64_t)c->cus * blocks_per_cu;
    if (cap > FLAT_MAX_BLOCKS) cap = FLAT_MAX_BLOCKS;
    int64_t g = want < cap ? want : cap;
    return (int)(g < 1 ? 1 : g);
}

static int flat_check(hgmm_ctx* c, int cov_type, int variant, int J) {
    if (!c->have_f32 || c->n <= 0) return fail(c, HGMM_ERR_STATE, "flat EM: call hgmm_set_points_f32 first");
    if (cov_type != HGMM_COV_DIAG && cov_type != HGMM_COV_SPHERICAL)
        return fail(c, HGMM_ERR_ARG, "cov_type must be 0 (diag) or 1 (spherical)");
    if (variant != HGMM_VARIANT_W && variant != HGMM_VARIANT_G)
        return fail(c, HGMM_ERR_ARG, "variant must be 0 (W) or 1 (G)");
    if (variant == HGMM_VARIANT_G && cov_type != HGMM_COV_DIAG)
        return fail(c, HGMM_ERR_ARG, "variant G (gmmreg_gpu) is diag-only");
    if (J < 1 || J > FLAT_MAX_J_CHUNKED)
        return fail(c, HGMM_ERR_ARG, "J = %d outside the supported range 1..%d", J, FLAT_MAX_J_CHUNKED);
    return HGMM_OK;
}

static int flat_setup(hgmm_ctx* c, int cov_type, int variant, int J) {
    HGMM_HIP(c, hipSetDevice(c->device));
    const bool chunked = J > FLAT_MAX_J;
    const int nchunks = chunked ? (J + CH_J - 1) / CH_J : 1;
    const int Jpad = chunked ? round_up(nchunks * CH_J, 256) : round_up(J, 256);
    c->flat.cov_type = cov_type; c->flat.variant = variant; c->flat.J = J; c->flat.Jpad = Jpad;
    c->flat.chunked = chunked; c->flat.nchunks = nchunks;
    if (chunked) {
        HGMM_TRY(ensure(c, c->f_cm, sizeof(float) * (size_t)nchunks * c->n));
        HGMM_TRY(ensure(c, c->f_cs, sizeof(float) * (size_t)nchunks * c->n));
        HGMM_TRY(ensure(c, c->f_ca, sizeof(int) * (size_t)nchunks * c->n));
        HGMM_TRY(ensure(c, c->f_lpn2, sizeof(float) * (size_t)c->n));
    }
    // the four parameter arrays are slices of ONE allocation, [cov | mu | w | inv] with a stride of Jpad per row:
    // what an E-step call uploads (mu, w, inv) and what an M-step call downloads (cov, mu, w) are each one contiguous
    // range -- one DMA packet per call instead of three (API-granular calls are a chain of such packets)
    HGMM_TRY(ensure(c, c->f_block, sizeof(float) * 10 * (size_t)Jpad));
    {
        float* blk = c->f_block.as<float>();
        c->f_cov.p = blk;                  c->f_cov.cap = sizeof(float) * 3 * Jpad;
        c->f_mu.p = blk + 3 * (size_t)Jpad; c->f_mu.cap = sizeof(float) * 3 * Jpad;
        c->f_w.p = blk + 6 * (size_t)Jpad;  c->f_w.cap = sizeof(float) * Jpad;
        c->f_inv.p = blk + 7 * (size_t)Jpad; c->f_inv.cap = sizeof(float) * 3 * Jpad;
    }
    HGMM_TRY(ensure(c, c->f_pack, sizeof(float) * PK_ROWS * Jpad));
    HGMM_TRY(ensure(c, c->f_hint, sizeof(float) * 3 * Jpad));
    // grid_for() never launches more than min(FLAT_MAX_BLOCKS, 8 workgroups per CU)
    const size_t max_blocks = std::min<size_t>(FLAT_MAX_BLOCKS, (size_t)c->cus * 8);
    HGMM_TRY(ensure(c, c->f_partials, sizeof(float) * max_blocks * FLAT_NSTAT * Jpad));
    HGMM_TRY(ensure(c, c->f_lpn_partials,
                    sizeof(double) * std::max<size_t>(FLAT_MAX_BLOCKS, (size_t)((c->n + 255) / 256))));
    HGMM_TRY(ensure(c, c->f_stats, sizeof(double) * (FLAT_NSTAT * Jpad + 2)));
    const bool fresh_ctl = c->f_ctl.p == nullptr;
    HGMM_TRY(ensure(c, c->f_ctl, 256));
    // int [16] of the buffer is never written again: the "not stopped" flag for launches outside a device loop
    if (fresh_ctl) HGMM_HIP(c, hipMemsetAsync(c->f_ctl.p, 0, 256, c->stream));
    return HGMM_OK;
}

static size_t cov_elems(int cov_type, int J) { return cov_type == HGMM_COV_DIAG ? (size_t)3 * J : (size_t)J; }

// ---- pinned staging ring ----------------------------------------------------------------------
// Small host <-> device transfers of the API-granular calls (parameters in, M-step results out) go through a
// pinned ring owned by the context: the host side is a memcpy, the device side a genuinely asynchronous DMA
// that queues behind the kernels already on the stream -- so hgmm_flat_estep_enqueue returns while its kernel
// runs and the next call's host work overlaps it.  A region is reused only after a stream synchronisation.
constexpr size_t STAGE_BYTES = 4u << 20;
int stage_reserve(hgmm_ctx* c, size_t bytes, void** out) {            // (declared in hgmm_ctx.h: the tree build's downloads use it too)
    if (!c->h_stage) {
        HGMM_HIP(c, hipHostMalloc(&c->h_stage, STAGE_BYTES, hipHostMallocDefault));
        c->h_stage_cap = STAGE_BYTES;
        c->h_stage_off = 0;
    }
    bytes = (bytes + 255) & ~(size_t)255;
    if (bytes > c->h_stage_cap) return fail(c, HGMM_ERR_ARG, "staging request of %zu bytes", bytes);
    if (c->h_stage_off + bytes > c->h_stage_cap) {
        HGMM_HIP(c, ctx_stream_sync(c));         // every earlier region has been consumed
        c->h_stage_off = 0;
    }
    *out = static_cast<char*>(c->h_stage) + c->h_stage_off;
    c->h_stage_off += bytes;
    return HGMM_OK;
}
static int stage_h2d(hgmm_ctx* c, void* dev, const void* host, size_t bytes) {
    void* st = nullptr;
    HGMM_TRY(stage_reserve(c, bytes, &st));
    std::memcpy(st, host, bytes);
    HGMM_HIP(c, hipMemcpyAsync(dev, st, bytes, hipMemcpyHostToDevice, c->stream));
    return HGMM_OK;
}

static int flat_upload(hgmm_ctx* c, const float* mu, const float* inv_or_cov, bool is_cov, const float* w) {
    const int J = c->flat.J, Jpad = c->flat.Jpad;
    const size_t ce = cov_elems(c->flat.cov_type, J);
    // an image of the device block's range [cov | mu | w] (is_cov) or [mu | w | inv] in the pinned ring, ONE copy
    void* st = nullptr;
    HGMM_TRY(stage_reserve(c, sizeof(float) * 7 * (size_t)Jpad, &st));
    float* img = static_cast<float*>(st);
    float* dev = is_cov ? c->f_cov.as<float>() : c->f_mu.as<float>();
    float* i_cov = img, *i_mu = img + (is_cov ? 3 * (size_t)Jpad : 0), *i_w = i_mu + 3 * (size_t)Jpad,
         * i_inv = i_w + Jpad;
    std::memcpy(i_mu, mu, sizeof(float) * 3 * J);
    std::memcpy(i_w, w, sizeof(float) * J);
    std::memcpy(is_cov ? i_cov : i_inv, inv_or_cov, sizeof(float) * ce);
    // extent actually needed: up to the end of the last array written
    const size_t used = is_cov ? (size_t)(i_w - img) + J : (size_t)(i_inv - img) + ce;
    HGMM_HIP(c, hipMemcpyAsync(dev, img, sizeof(float) * used, hipMemcpyHostToDevice, c->stream));
    return HGMM_OK;
}

static void launch_pack(hgmm_ctx* c) {
    const FlatState& f = c->flat;
    flat_pack_kernel<<<(f.Jpad + 255) / 256, 256, 0, c->stream>>>(
        f.J, f.Jpad, f.cov_type, f.variant, c->f_mu.as<float>(), c->f_inv.as<float>(), c->f_w.as<float>(),
        c->f_pack.as<float>());
}

// layout choice: vector slots for full 256-component groups, scalar slots for a short remainder
static void pick_layout(int J, int* nv4, int* nv1) {
    if (J % 4 == 0) {
        int a = J / 256, rem = J - 256 * (J / 256), b = 0;
        if (rem > 128) { a += 1; } else if (rem > 0) { b = (rem + 63) / 64; }
        *nv4 = a; *nv1 = b;
    } else {
        const int ns = (J + 63) / 64;
        const int steps[] = {1, 2, 3, 4, 6, 8, 12, 16};
        int b = 16;
        for (int s : steps) if (s >= ns) { b = s; break; }
        *nv4 = 0; *nv1 = b;
    }
}


// LAYOUT_DISPATCH(nv4, nv1, M): expands M(NV4, NV1) for the supported (NV4, NV1) pairs
#define LAYOUT_DISPATCH(nv4, nv1, M)                                                           \
    do {                                                                                       \
        const int key_ = (nv4) * 100 + (nv1);                                                  \
        switch (key_) {                                                                        \
            case 1: M(0, 1); break;   case 2: M(0, 2); break;   case 3: M(0, 3); break;        \
            case 4: M(0, 4); break;   case 6: M(0, 6); break;   case 8: M(0, 8); break;        \
            case 12: M(0, 12); break; case 16: M(0, 16); break;                                \
            case 100: M(1, 0); break; case 101: M(1, 1); break; case 102: M(1, 2); break;      \
            case 200: M(2, 0); break; case 201: M(2, 1); break; case 202: M(2, 2); break;      \
            case 300: M(3, 0); break; case 301: M(3, 1); break; case 302: M(3, 2); break;      \
            case 400: M(4, 0); break;                                                          \
            default: return fail(c, HGMM_ERR_ARG, "unsupported layout %d/%d", (nv4), (nv1));   \
        }                                                                                      \
    } while (0)

// Offered store rate of the paced N x J writers (materialising E-step, estimate_log_prob), GB/s.  The knee of the write
// path sat at 6.8 - 7.0 TB/s on every box of round 4, but it is a property of the memory system's state (refresh rate
// with temperature, the stacks of the particular chip), and past it the kernel does not degrade gracefully -- it drops
// to ~5.3 TB/s; the knee is soft besides: at 6700 one launch in three already ran 3 - 18 % long in bursts, at 6600 the
// 150 launches of a bench run stayed within 488 - 506 us of their 487 (profiles/r04).  So the rate starts at 6600 and is
// CONTROLLED: a paced launch that runs more than 6 % longer than its
// target explains counts as a strike, three strikes in a row lower the target by 2 % (never below PACE_FLOOR_GBS).  The
// evidence comes from event pairs around the last few launches that are queried -- never waited for -- at the next
// launch, so the stream is not disturbed.  Launches that are not store-bound by construction (the row-maximum loop,
// small tables) neither adapt nor are they judged.  HGMM_ESTEP_TARGET_GBS=<n> fixes the rate (0: un-paced), HGMM_ESTEP_ADAPT=0
// keeps the initial one.
constexpr int ESTEP_TARGET_GBS = 6600;
constexpr double PACE_FLOOR_GBS = 5800.0;
constexpr double PACE_STRIKE_RATIO = 1.06;
constexpr double PACE_STEP = 0.98;
constexpr double PACE_MIN_BYTES = 256.0 * 1024 * 1024;
// ... and upwards, carefully: the knee sat at 7.0 TB/s on one box and at 7.6 on another (profiles/r04/paced_fill_lease*.log).
// After PACE_PROBE_AFTER clean launches in a row the rate is raised by 2 % on probation: three clean launches keep it, the
// first long one takes it back, remembers the rate as a ceiling and doubles the wait before the next probe.
// ... and a ceiling is a statement about the memory system's state WHEN it was learnt (another stream writing at the same
// time, a hot chip), not for life: after PACE_FORGET_AFTER clean launches below it (HGMM_PACE_FORGET=<n>) it is forgotten and
// the probes start over with their first waiting time.  hgmm_pace_reset() forgets everything at once.
constexpr int PACE_FORGET_AFTER = 10000;
constexpr int PACE_PROBE_AFTER = 24, PACE_PROBE_AFTER_MAX = 4096;
constexpr double PACE_PROBE_STEP = 1.02;
constexpr double PACE_CEILING_GBS = 7600.0;

// the rate the next paced launch of J-float rows offers
static double pace_target(hgmm_ctx* c, int J) {
    const int fixed = c->cfg[CFG_ESTEP_TARGET_GBS];
    if (fixed >= 0) return (double)fixed;
    PaceCtl& p = c->pace;
    if (p.target <= 0.0 || p.J != J) {
        p.target = (double)c->cfg[CFG_PACE_START];
        p.strikes = p.clean = p.probe_seen = 0;
        p.probe_after = PACE_PROBE_AFTER;
        p.probe_base = 0.0;
        p.ceiling = 1e30;
        p.since_ceiling = 0;
        p.J = J;
    }
    return p.target;
}
// look at the launches that have finished since the last call (no waiting)
static void pace_poll(hgmm_ctx* c) {
    PaceCtl& p = c->pace;
    while (p.tail != p.head) {
        const unsigned s = p.tail % PaceCtl::RING;
        if (hipEventQuery(p.ev[s][1]) != hipSuccess) { (void)hipGetLastError(); break; }
        float ms = 0.f;
        bool have_ms = false;
        if (p.stamps && p.grid_at[s] > 0 && c->wall_khz > 0) {
            // the launch is complete (its end event has passed): first workgroup start to last workgroup end
            const unsigned long long* st = p.stamps + (size_t)s * 2 * PaceCtl::STAMP_WG;
            unsigned long long t0 = ~0ull, t1 = 0ull;
            for (int b = 0; b < p.grid_at[s]; ++b) {
                t0 = st[b] < t0 ? st[b] : t0;
                t1 = st[p.grid_at[s] + b] > t1 ? st[p.grid_at[s] + b] : t1;
            }
            if (t1 > t0) { ms = (float)((double)(t1 - t0) / (double)c->wall_khz); have_ms = true; }
        } else {
            have_ms = hipEventElapsedTime(&ms, p.ev[s][0], p.ev[s][1]) == hipSuccess;
        }
        if (have_ms && p.tgt_at[s] == p.target) {
            const double ideal_ms = p.bytes_at[s] / (p.tgt_at[s] * 1e9) * 1e3;
            const bool slow = (double)ms > PACE_STRIKE_RATIO * ideal_ms;
            if (p.probe_base > 0.0) {
                // a probe is running: ONE long launch ends it (the memory system has said no), three clean ones accept it
                if (slow) {
                    p.ceiling = p.target;
                    p.since_ceiling = 0;
                    p.target = p.probe_base;
                    p.probe_base = 0.0;
                    p.probe_after = std::min(PACE_PROBE_AFTER_MAX, 2 * p.probe_after);
                    p.clean = 0;
                    p.strikes = 0;
                } else if (++p.probe_seen >= 3) {
                    p.probe_base = 0.0;
                    p.steps_up++;
                    p.clean = 0;
                }
            } else if (slow) {
                p.clean = 0;
                if (++p.strikes >= 3) {
                    p.ceiling = std::min(p.ceiling, p.target);
                    p.since_ceiling = 0;
                    p.target = std::max(PACE_FLOOR_GBS, p.target * PACE_STEP);
                    p.strikes = 0;
                    p.steps_down++;
                }
            } else {
                p.strikes = 0;
                if (p.ceiling < 1e29 && ++p.since_ceiling >= c->cfg[CFG_PACE_FORGET]) {
                    p.ceiling = 1e30;                       // what congested then need not congest now: probe afresh
                    p.since_ceiling = 0;
                    p.probe_after = PACE_PROBE_AFTER;
                    p.ceilings_forgotten++;
                }
                const double up = p.target * PACE_PROBE_STEP;
                if (++p.clean >= p.probe_after && up < p.ceiling * 0.995 && up <= PACE_CEILING_GBS) {
                    p.probe_base = p.target;
                    p.target = up;
                    p.probe_seen = 0;
                }
            }
        }
        p.tail++;
    }
}
// bracket a paced launch: begin() before the kernel, end() behind it; false: this launch is not observed
// (*stamp_out: where this launch's workgroups leave their wall-clock stamps, null when the launch is judged by its events)
static bool pace_observe_begin(hgmm_ctx* c, double target, double bytes, int grid, unsigned long long** stamp_out) {
    PaceCtl& p = c->pace;
    *stamp_out = nullptr;
    if (c->cfg[CFG_ESTEP_TARGET_GBS] >= 0) return false;            // a fixed rate: nothing to control
    if (bytes < PACE_MIN_BYTES || target <= 0.0) return false;
    if (!p.have_events) {
        for (auto& pr : p.ev)
            if (hipEventCreate(&pr[0]) != hipSuccess || hipEventCreate(&pr[1]) != hipSuccess) return false;
        p.have_events = true;
        {
            void* h = nullptr;
            void* d = nullptr;
            const size_t bytes_st = sizeof(unsigned long long) * PaceCtl::RING * 2 * PaceCtl::STAMP_WG;
            if (hipHostMalloc(&h, bytes_st, hipHostMallocDefault) == hipSuccess && hipHostGetDevicePointer(&d, h, 0) == hipSuccess) {
                p.stamps = static_cast<unsigned long long*>(h);
                p.stamps_dev = static_cast<unsigned long long*>(d);
            } else {
                if (h) (void)hipHostFree(h);
                (void)hipGetLastError();
            }
        }
    }
    pace_poll(c);
    if (p.head - p.tail >= (unsigned)PaceCtl::RING) return false;          // every pair is still in flight
    const unsigned s = p.head % PaceCtl::RING;
    p.tgt_at[s] = target;
    p.bytes_at[s] = bytes;
    p.grid_at[s] = 0;
    if (p.stamps && grid <= PaceCtl::STAMP_WG) {
        // the kernel writes start[b] at st[b] and end[b] at st[grid + b]: the slot is laid out for THIS grid
        p.grid_at[s] = grid;
        *stamp_out = p.stamps_dev + (size_t)s * 2 * PaceCtl::STAMP_WG;
    }
    return hipEventRecord(p.ev[s][0], c->stream) == hipSuccess;
}
static void pace_observe_end(hgmm_ctx* c) {
    PaceCtl& p = c->pace;
    if (hipEventRecord(p.ev[p.head % PaceCtl::RING][1], c->stream) == hipSuccess) p.head++;
}

// StorePacer period (1/16 ticks of the 100 MHz wall clock) per group of 4 rows: all `grid` x 4 waves together offer
// `target_gbs` GB/s of J-float rows.  0 = no pacing.
static int store_pace16(hgmm_ctx* c, int grid, int J, double target_gbs) {
    if (target_gbs <= 0.0) return 0;
    int khz = c->wall_khz;
    if (khz <= 0) {
        if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, c->device) != hipSuccess || khz <= 0) khz = 100000;
        c->wall_khz = khz;
    }
    const double bytes_per_group = 4.0 * 4.0 * (double)J;
    const double seconds = bytes_per_group * (double)grid * WAVES_PER_BLOCK / (target_gbs * 1e9);
    const double p16 = seconds * (double)khz * 1e3 * 16.0;
    return p16 < 1.0 ? 0 : (p16 > 2e9 ? 2000000000 : (int)(p16 + 0.5));
}

// the 4-rows-per-wave materialising kernel for one (layout, grid, log-sum-exp variant); false: layout not instantiated
static bool launch_estep_rows(hgmm_ctx* c, int nv4, int nv1, int grid_r, bool cshift, float* log_resp, float* lpn,
                              int32_t* argmax, int pace, unsigned long long* stamp) {
    const FlatState& f = c->flat;
    const float* X = c->x_aos.as<float>();
    const float* pk = c->f_pack.as<float>();
    double* lp = c->f_lpn_partials.as<double>();
#define ESTEP_R(A, B)                                                                              \
    flat_estep_rows_pk_kernel<A, B, 4, true><<<grid_r, BLOCK, 0, c->stream>>>(                      \
        X, pk, c->n, f.J, f.Jpad, log_resp, lpn, argmax, lp, cshift ? 1 : 0, pace, stamp)
    if (nv4 == 3 && nv1 == 1) ESTEP_R(3, 1);
    else if (nv4 == 3 && nv1 == 0) ESTEP_R(3, 0);
    else if (nv4 == 3 && nv1 == 2) ESTEP_R(3, 2);
    else if (nv4 == 4 && nv1 == 0) ESTEP_R(4, 0);
    else if (nv4 == 2 && nv1 == 0) ESTEP_R(2, 0);
    else if (nv4 == 2 && nv1 == 1) ESTEP_R(2, 1);
    else if (nv4 == 2 && nv1 == 2) ESTEP_R(2, 2);
    else if (nv4 == 1 && nv1 == 0) ESTEP_R(1, 0);
    else if (nv4 == 1 && nv1 == 1) ESTEP_R(1, 1);
    else if (nv4 == 1 && nv1 == 2) ESTEP_R(1, 2);
    else if (nv4 == 0 && nv1 == 1) ESTEP_R(0, 1);
    else if (nv4 == 0 && nv1 == 2) ESTEP_R(0, 2);
    else return false;
#undef ESTEP_R
    return true;
}

// estimate_log_prob on the four-rows-in-flight kernel (the layouts the materialising E-step is instantiated for); false: not instantiated
static bool launch_logprob_rows(hgmm_ctx* c, int nv4, int nv1, float* log_prob) {
    const FlatState& f = c->flat;
    const bool have = (nv4 >= 1 && nv4 <= 3 && nv1 <= 2) || (nv4 == 4 && nv1 == 0) || (nv4 == 0 && (nv1 == 1 || nv1 == 2));
    if (!have) return false;
    // Grid: with no reduction and no exponential left, the store stream is all this kernel waits for.  Un-paced it ran
    // 0.60 ms at every grid from 128 to 4096 workgroups on the box where the E-step took 0.54 (tools/logprob_sweep.py):
    // the write path was over-subscribed.  Paced like the E-step (StorePacer), two workgroups per CU: 0.506 ms.
    const int64_t groups = (c->n + 3) / 4;
    const int64_t g64 = (int64_t)c->cus * 2;
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(g64, (groups + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK));
    const float* X = c->x_aos.as<float>();
    const float* pk = c->f_pack.as<float>();
    ProfScope prof(c, HGMM_K_FLAT_ESTEP);
    // (the raw table follows the rate the E-step's controller has settled on for this row length)
    const int pace = store_pace16(c, grid, f.J, pace_target(c, f.J));
#define LOGP_R(A, B) flat_logprob_rows_pk_kernel<A, B, true><<<grid, BLOCK, 0, c->stream>>>(X, pk, c->n, f.J, f.Jpad, log_prob, pace)
    if (nv4 == 3 && nv1 == 1) LOGP_R(3, 1);
    else if (nv4 == 3 && nv1 == 0) LOGP_R(3, 0);
    else if (nv4 == 3 && nv1 == 2) LOGP_R(3, 2);
    else if (nv4 == 4 && nv1 == 0) LOGP_R(4, 0);
    else if (nv4 == 2 && nv1 == 0) LOGP_R(2, 0);
    else if (nv4 == 2 && nv1 == 1) LOGP_R(2, 1);
    else if (nv4 == 2 && nv1 == 2) LOGP_R(2, 2);
    else if (nv4 == 1 && nv1 == 0) LOGP_R(1, 0);
    else if (nv4 == 1 && nv1 == 1) LOGP_R(1, 1);
    else if (nv4 == 1 && nv1 == 2) LOGP_R(1, 2);
    else if (nv4 == 0 && nv1 == 1) LOGP_R(0, 1);
    else if (nv4 == 0 && nv1 == 2) LOGP_R(0, 2);
    else return false;
#undef LOGP_R
    return true;
}

// predict(): the four-rows-in-flight arg-max kernel for the layouts it is instantiated for (the others take the general kernel)
static bool predict_rows_layout(const hgmm_ctx* c, int nv4, int nv1) {
    if (c->cfg[CFG_PREDICT_SINGLE_ROW]) return false;
    return (nv4 == 3 && nv1 <= 2) || (nv4 == 4 && nv1 == 0) || (nv4 == 2 && nv1 <= 2) || (nv4 == 1 && nv1 <= 2) ||
           (nv4 == 0 && (nv1 == 1 || nv1 == 2));
}
static bool launch_predict_rows(hgmm_ctx* c, int nv4, int nv1, int32_t* labels) {
    const FlatState& f = c->flat;
    // (one frame of 10^6 x 800: 0.38 / 0.26 / 0.26 / 0.24 ms with 1 / 2 / 3 / 4 workgroups per CU; the single-row kernel 0.36)
    // (N = 1e6: 8 workgroups per CU beat 4 at every J -- 0.258 -> 0.235 ms at J = 800, 0.080 -> 0.072 at J = 100; profiles/r04/small_j_grids.log)
    const int grid = grid_for(c, (c->n + 3) / 4, c->n >= 400000 ? 8 : 4);
    const float* X = c->x_aos.as<float>();
    const float* pk = c->f_pack.as<float>();
    // (the marginal cost per component is the VALU floor of 4.5 packed + 2.5 plain instructions; what remains is ~0.06 ms
    //  that does not depend on J)
#define PRED_R(A, B) flat_predict_rows_kernel<A, B><<<grid, BLOCK, 0, c->stream>>>(X, pk, c->n, f.J, f.Jpad, labels)
    if (nv4 == 3 && nv1 == 1) PRED_R(3, 1);
    else if (nv4 == 3 && nv1 == 0) PRED_R(3, 0);
    else if (nv4 == 3 && nv1 == 2) PRED_R(3, 2);
    else if (nv4 == 4 && nv1 == 0) PRED_R(4, 0);
    else if (nv4 == 2 && nv1 == 0) PRED_R(2, 0);
    else if (nv4 == 2 && nv1 == 1) PRED_R(2, 1);
    else if (nv4 == 2 && nv1 == 2) PRED_R(2, 2);
    else if (nv4 == 1 && nv1 == 0) PRED_R(1, 0);
    else if (nv4 == 1 && nv1 == 1) PRED_R(1, 1);
    else if (nv4 == 1 && nv1 == 2) PRED_R(1, 2);
    else if (nv4 == 0 && nv1 == 1) PRED_R(0, 1);
    else if (nv4 == 0 && nv1 == 2) PRED_R(0, 2);
    else return false;
#undef PRED_R
    return true;
}

// Number of workgroups of the materialising E-step.  The kernel sits where two limits meet: with one wave per SIMD
// its arithmetic is issue-bound, and the HBM write path delivers LESS the more waves write at once (fillbench) --
// so there is a best grid and it is sharp: constant-shift loop 0.559 / 0.543 / 0.559 / 0.580 / 0.616 ms at 184 / 192 /
// 200 / 208 / 256 workgroups, row-maximum loop 0.576 / 0.540 / 0.565 / 0.589 at 208 / 224 / 240 / 256 (one box; the
// generic loop's optimum sat at 256 on the next one, the constant-shift loop's at 192 on all three that were tried:
// 0.543 / 0.556 / 0.538 ms against 0.589 / 0.564 / 0.551 for the round-1 shape).  Timing the candidates at first
// use was tried and dropped: two launches after an idle moment run at other clocks than a stream of them, and picked
// 176.  Per-row results do not depend on the grid; the mean log-normaliser's summation order does (last bits).
// Grid of the materialising E-step.  The kernel sits where two limits meet (its header): one wave per SIMD issues one
// VALU instruction per ~5 cycles whatever its ILP, and the HBM write path delivers less the more waves write at once --
// so in a stream of E-steps 3/4 of the CUs with one workgroup each is the optimum (0.536 ms; 2 per CU: 0.565).  That
// optimum is FRAGILE: right behind the HBM-read-bound M-step (the `e_step -> m_step` loop of a caller of the module
// functions) the same launch takes 0.81 ms -- the chip comes out of a memory-bound kernel at a lower core clock and the
// issue-bound side of the kernel pays for it -- while two workgroups per CU take 0.60 ms there
// (tools/em_loop_probe.py, profiles/r03/em_loop_probe.log: 160 / 192 / 224 / 256 / 320 / 384 / 512 workgroups behind an
// M-step: 0.94 / 0.81 / 0.74 / 0.69 / 0.59 / 0.60 / 0.60 ms; alone: 0.63 / 0.54 / 0.54 / 0.56 / 0.70 / 0.61 / 0.57).
// The same holds for a STREAM of E-steps that nobody waits for (hgmm_flat_estep_async / _dev back to back): 0.70 ms at
// 192 workgroups, 0.62 at 512 -- the 0.54 of the 3/4 grid needs the ~25 us of idle time a blocking call leaves behind
// every launch (tools/em_loop_probe2.py, profiles/r03/em_loop_probe2.log).
// The grid therefore follows what the context did last: behind an M-step, or behind an E-step the host has not
// waited for, two workgroups per CU; after an idle moment 3/4 of the CUs.
static int estep_rows_grid(hgmm_ctx* c, bool cshift) {
    const bool after_mstep = c->flat.last_kernel == 2 || (c->flat.last_kernel == 1 && !c->flat.idle_since_launch);
    const int full = grid_for(c, (c->n + 3) / 4, after_mstep ? 2 : 1);
    if (after_mstep) return full;
    return cshift ? std::min(full, std::max(1, c->cus * 3 / 4)) : full;
}

template <bool NORMALISE>
static int launch_estep(hgmm_ctx* c, float* log_resp, float* lpn, int32_t* argmax, int* grid_out) {
    const FlatState& f = c->flat;
    const int grid = grid_for(c, c->n, 2);
    *grid_out = grid;
    const float* X = c->x_aos.as<float>();
    const float* pk = c->f_pack.as<float>();
    double* lp = c->f_lpn_partials.as<double>();
    int nv4, nv1;
    pick_layout(f.J, &nv4, &nv1);
    // (single-row kernel) non-temporal stores only for rows that are whole 16-byte pieces: J = 513 wrote its table in 0.364 ms
    // with them and in 0.251 ms without -- a line two rows share is written twice, each time in part
    // (profiles/r04/estep_nt_by_J.log)
    const bool nt = NORMALISE && f.J % 4 == 0;
    const int rr = 0;
    if (!NORMALISE && !log_resp && !lpn && argmax && predict_rows_layout(c, nv4, nv1)) {      // predict(): labels only
        ProfScope prof(c, HGMM_K_FLAT_ESTEP);
        (void)launch_predict_rows(c, nv4, nv1, argmax);
        HGMM_HIP(c, hipGetLastError());
        return HGMM_OK;
    }
    if (!NORMALISE && log_resp && !lpn && !argmax) {                                        // estimate_log_prob: the raw table
        if (launch_logprob_rows(c, nv4, nv1, log_resp)) {
            c->flat.last_kernel = 1;
            c->flat.idle_since_launch = false;
            HGMM_HIP(c, hipGetLastError());
            return HGMM_OK;
        }
    }
    // Materialising path: 4 rows in flight per wave, at most ONE workgroup per CU (see the kernel's header)
    const int rows = (NORMALISE && log_resp) ? 4 : 1;
    if (rows > 1) {
        const bool cshift = !argmax;
        // Paced stores (StorePacer): two workgroups per CU -- all resident at once, which the pacer's period assumes, and
        // enough waves for the arithmetic to keep up at any core clock -- offering the rows at the controlled target
        // rate (ESTEP_TARGET_GBS above).  The write path takes an evenly paced, phase-staggered 6.8 TB/s of these rows in
        // every call pattern and collapses to ~5.3 at 7.0 - 7.4 (tools/pace_sweep.py, profiles/r04/pace_sweep.log).
        // HGMM_ESTEP_TARGET_GBS=0: the un-paced launch with round 3's grid policy (estep_rows_grid).
        const double target = pace_target(c, f.J);
        // (few components per lane leave the register file empty: on a large cloud four workgroups per CU -- all
        //  resident, as the pacer's period assumes -- hide more of the rows' reduction chains: J = 64 0.110 -> 0.082 ms at
        //  N = 1e6; from J = 256 on two are as good or better, and more than fit at once break the pacing: profiles/r04/small_j_grids.log)
        const int bpc_rows = (c->n >= 400000 && f.J <= 128) ? 4 : 2;
        const int grid_r = target > 0.0 ? grid_for(c, (c->n + 3) / 4, bpc_rows)
                                        : estep_rows_grid(c, cshift);
        const int pace = store_pace16(c, grid_r, f.J, target);
        c->flat.last_kernel = 1;
        c->flat.idle_since_launch = false;
        // (only the constant-shift loop is store-bound by construction: the row-maximum loop is not judged)
        unsigned long long* stamp = nullptr;
        const bool observed = cshift && pace_observe_begin(c, target, 4.0 * (double)c->n * (double)f.J, grid_r, &stamp);
        bool launched;
        {
            ProfScope prof(c, HGMM_K_FLAT_ESTEP);
            launched = launch_estep_rows(c, nv4, nv1, grid_r, cshift, log_resp, lpn, argmax, pace, observed ? stamp : nullptr);
        }
        if (launched) {
            if (observed) pace_observe_end(c);
            *grid_out = grid_r;
            HGMM_HIP(c, hipGetLastError());
            return HGMM_OK;
        }
    }
    ProfScope prof(c, HGMM_K_FLAT_ESTEP);
#define ESTEP_M(A, B)                                                                              \
    do {                                                                                           \
        if (nt) flat_estep_kernel<A, B, NORMALISE, NORMALISE><<<grid, BLOCK, 0, c->stream>>>(       \
                    X, pk, c->n, f.J, f.Jpad, log_resp, lpn, argmax, lp, rr);                       \
        else flat_estep_kernel<A, B, NORMALISE, false><<<grid, BLOCK, 0, c->stream>>>(              \
                    X, pk, c->n, f.J, f.Jpad, log_resp, lpn, argmax, lp, rr);                       \
    } while (0)
    LAYOUT_DISPATCH(nv4, nv1, ESTEP_M);
#undef ESTEP_M
    HGMM_HIP(c, hipGetLastError());
    return HGMM_OK;
}

static int launch_fused(hgmm_ctx* c, const int* done_flag, int* grid_out, int* valid_j) {
    c->flat.last_kernel = 3;
    const FlatState& f = c->flat;
    if (!done_flag) done_flag = c->f_ctl.as<int>() + 16;        // always 0 (flat_setup)
    const int ns = (f.J + 63) / 64;
    // Grid.  Two workgroups per CU are the floor (a small cloud wants every CU busy: bun000 is fastest at 512 workgroups of
    // 20 rows per wave, tools/bunny_grid.py); a large cloud takes more, as long as a wave keeps ~100 rows: J = 800 (254
    // registers, two waves per SIMD resident either way) gains 1.6 % from 4 per CU -- a finer tail -- and small J, whose
    // lanes hold few components and leave the register file empty, gain 10 - 28 % from 6 - 8 per CU: more resident waves
    // hide the row's reduction chain (N = 1e6: J = 64 0.105 -> 0.075 ms per iteration, J = 100 0.117 -> 0.098, J = 400
    // 0.227 -> 0.202; profiles/r04/fused_bpc_by_J.log).  More workgroups are more partials for the reduction to read:
    // the iteration times above include it.  
    int grid;
    {
        const int bpc_max = ns <= 4 ? 8 : (ns <= 8 ? 6 : 4);
        const int64_t by_rows = c->n / (WAVES_PER_BLOCK * 100);
        const int64_t big = std::min<int64_t>(by_rows, std::min<int64_t>((int64_t)c->cus * bpc_max, FLAT_MAX_BLOCKS));
        grid = (int)std::max<int64_t>(grid_for(c, c->n, 2), big);
    }
    *grid_out = grid;
    const float* X = c->x_aos.as<float>();
    const float* pk = c->f_pack.as<float>();
    float* part = c->f_partials.as<float>();
    double* lp = c->f_lpn_partials.as<double>();
    // Design notes (measured on MI355X at N = 1e6, J = 800; see DESIGN.md section 6 for the list):
    //  * more rows in flight per wave lose: 1 row (249 VGPRs, 2 waves/SIMD) 0.544 ms, 2 rows
    //    (310 regs, 1 wave/SIMD) 0.677 ms, 4 rows (424 regs) 0.772 ms;
    //  * reusing the previous row's log-sum as the LSE shift (no max reduction) costs 3 VGPRs,
    //    which tips J = 800 over the 256-register line: 0.66 - 0.78 ms;
    //  * wave-uniform skipping of 64-component slots whose responsibilities are all < 1e-10
    //    never triggers while components are broad: 0 % gain.
    //  * two rows in flight in the constant-shift kernel (differences recomputed instead of kept, so that
    //    only the exponentials of a row live between its phases): 216 instead of 221 instructions per row,
    //    but 256 VGPRs + 4 AGPRs = one wave per SIMD, 0.46 - 0.48 ms; forced to two waves per SIMD it
    //    spills 372 B per lane, 0.43 ms; one row (248 VGPRs, two waves per SIMD) stays the best, 0.405 ms;
    //  * constant-shift log-sum-exp: see flat_fused_pk_kernel; 0.514 -> 0.440 ms (a table whose largest constant
    //    leaves the safe range takes the kernel's row-maximum loop by itself).
#define FUSED_CASE(S)                                                                           \
    do {                                                                                        \
        flat_fused_pk_kernel<S><<<grid, BLOCK, 0, c->stream>>>(X, pk, c->n, f.J, f.Jpad, part,  \
                                                              lp, done_flag, 1);              \
        *valid_j = S * 64;                                                                      \
    } while (0)
    ProfScope prof(c, HGMM_K_FLAT_FUSED);
    switch (ns) {
        case 1: FUSED_CASE(1); break;
        case 2: FUSED_CASE(2); break;
        case 3: FUSED_CASE(3); break;
        case 4: FUSED_CASE(4); break;
        case 5: FUSED_CASE(5); break;
        case 6: FUSED_CASE(6); break;
        case 7: FUSED_CASE(7); break;
        case 8: FUSED_CASE(8); break;
        case 9: FUSED_CASE(9); break;
        case 10: FUSED_CASE(10); break;
        case 11: FUSED_CASE(11); break;
        case 12: FUSED_CASE(12); break;
        case 13: FUSED_CASE(13); break;
        case 14: FUSED_CASE(14); break;
        case 15: FUSED_CASE(15); break;
        default: FUSED_CASE(16); break;
    }
#undef FUSED_CASE
    HGMM_HIP(c, hipGetLastError());
    return HGMM_OK;
}

static int launch_reduce(hgmm_ctx* c, int nblocks, int valid_j, bool with_lpn, const int* done_flag,
                         int n_lpn_blocks = -1) {
    const FlatState& f = c->flat;
    const int total = FLAT_NSTAT * f.Jpad;
    // Under a communicator the all-reduce below runs in place on f_stats in EVERY enqueued iteration, also
    // the ones after the device-side stop (whose kernels return at once).  The local statistics are therefore
    // rebuilt from the (then unchanged) partials each time instead of being left to be summed over the ranks
    // again and again.
    if (c->comm_on() || !done_flag) done_flag = c->f_ctl.as<int>() + 16;        // always 0 (flat_setup)
    flat_reduce_kernel<<<(total + RED_IDX - 1) / RED_IDX + 1, RED_IDX * RED_SLICES, 0, c->stream>>>(
        c->f_partials.as<float>(), with_lpn ? c->f_lpn_partials.as<double>() : nullptr, nblocks,
        n_lpn_blocks < 0 ? nblocks : n_lpn_blocks, valid_j, f.Jpad, (double)c->n, c->f_stats.as<double>(),
        done_flag);
    HGMM_HIP(c, hipGetLastError());
    if (c->comm_on()) HGMM_TRY(allreduce_f64_dev(c, c->f_stats.as<double>(), (size_t)total + 2));
    return HGMM_OK;
}

// ---- chunked (J > 1024) launch helpers ---------------------------------------------------------
static int chunk_valid(const FlatState& f, int ci) {
    const int rem = f.J - ci * CH_J;
    return rem < CH_J ? rem : CH_J;
}

// per-chunk (max, sum[, arg-max]) + per-row combine -> f_lpn2 [, lpn_out, argmax_out, lpn partials]
static int chunk_normalisers(hgmm_ctx* c, bool want_arg, float* lpn_out, int32_t* argmax_out,
                             bool want_partials, int* n_lpn_blocks) {
    const FlatState& f = c->flat;
    const int grid = grid_for(c, c->n, 3);
    float* cm = c->f_cm.as<float>();
    float* cs = c->f_cs.as<float>();
    int* ca = c->f_ca.as<int>();
    for (int ci = 0; ci < f.nchunks; ++ci)
        flat_chunk_lse_kernel<<<grid, BLOCK, 0, c->stream>>>(
            c->x_aos.as<float>(), c->f_pack.as<float>() + ci * CH_J, c->n, chunk_valid(f, ci), ci * CH_J, f.Jpad,
            cm + (size_t)ci * c->n, cs + (size_t)ci * c->n, want_arg ? ca + (size_t)ci * c->n : nullptr);
    const int cblocks = (int)((c->n + 255) / 256);
    flat_chunk_combine_kernel<<<cblocks, 256, 0, c->stream>>>(cm, cs, want_arg ? ca : nullptr, f.nchunks, c->n,
                                                            c->f_lpn2.as<float>(), lpn_out, argmax_out,
                                                            want_partials ? c->f_lpn_partials.as<double>() : nullptr);
    HGMM_HIP(c, hipGetLastError());
    if (n_lpn_blocks) *n_lpn_blocks = cblocks;
    return HGMM_OK;
}

// statistics of all chunks with the rows' final normalisers -> f_partials (grid returned)
static int chunk_statistics(hgmm_ctx* c, const int* done_flag, int* grid_out) {
    const FlatState& f = c->flat;
    const int grid = grid_for(c, c->n, 2);
    {
        ProfScope prof(c, HGMM_K_FLAT_FUSED);
        for (int ci = 0; ci < f.nchunks; ++ci)
            flat_chunk_fused_kernel<<<grid, BLOCK, 0, c->stream>>>(
                c->x_aos.as<float>(), c->f_pack.as<float>() + ci * CH_J, c->n, f.Jpad, c->f_lpn2.as<float>(),
                c->f_partials.as<float>() + ci * CH_J, done_flag);
    }
    HGMM_HIP(c, hipGetLastError());
    *grid_out = grid;
    return HGMM_OK;
}

// one EM iteration, everything asynchronous on the context's stream
static int enqueue_em_iteration(hgmm_ctx* c) {
    FlatState& f = c->flat;
    int* ctl = c->f_ctl.as<int>();
    float* ctl_f = reinterpret_cast<float*>(ctl + 8);
    int grid = 0, valid_j = 0;
    if (f.chunked) {
        int n_lpn = 0;
        HGMM_TRY(chunk_normalisers(c, false, nullptr, nullptr, true, &n_lpn));
        HGMM_TRY(chunk_statistics(c, ctl, &grid));
        HGMM_TRY(launch_reduce(c, grid, f.nchunks * CH_J, true, ctl, n_lpn));
        flat_finalize_mb_kernel<<<(f.Jpad + 255) / 256, 256, 0, c->stream>>>(
            c->f_stats.as<double>(), c->f_pack.as<float>() + PK_MU * f.Jpad, f.J, f.Jpad, f.cov_type, f.variant,
            c->f_mu.as<float>(), c->f_cov.as<float>(), c->f_w.as<float>(), c->f_inv.as<float>(),
            c->f_pack.as<float>(), ctl);
        flat_ctl_kernel<<<1, 1, 0, c->stream>>>(c->f_stats.as<double>(), f.Jpad, c->f_lls.as<float>(), f.lls_cap,
                                               f.tol, ctl, ctl_f);
        HGMM_HIP(c, hipGetLastError());
        f.launched++;
        return HGMM_OK;
    }
    // stop flag of this iteration / of the next one (see flat_finalize_kernel): ctl[0] and ctl[3] in turn
    int* stop = ctl + ((f.launched & 1) ? 3 : 0);
    int* stop_next = ctl + ((f.launched & 1) ? 0 : 3);
    HGMM_TRY(launch_fused(c, stop, &grid, &valid_j));
    HGMM_TRY(launch_reduce(c, grid, valid_j, true, stop));
    // (reduction + finalisation in ONE launch -- the last workgroup to finish, found by a ticket, runs the
    //  M-step -- was measured no faster: 0.4253 vs 0.4254 ms per iteration at C3, 35 - 36 k vs 38 k it/s
    //  on bun000 J = 100: the serial tail in one 256-thread workgroup costs what the launch saved.  Round 4 tried the form
    //  WITHOUT a hand-over -- a workgroup owns 8 components, adds up their 7 statistics over all partial blocks in the
    //  reduction kernel's own slices and orders (the same fit bit for bit) and runs their M-step: slower again, 0.3465 vs
    //  0.3416 ms per iteration at C3 and 0.45 vs 0.385 ms per 20-iteration fit on bun000 J = 100 -- 100 workgroups reading
    //  32-byte pieces of 512 partial blocks take longer than the 225 well-coalesced workgroups of the reduction kernel
    //  plus a launch.)
    // the statistics were centred about the means the E-step used = rows PK_MU.. of pack
    flat_finalize_kernel<<<f.Jpad / 256, 256, 0, c->stream>>>(
        c->f_stats.as<double>(), c->f_pack.as<float>() + PK_MU * f.Jpad, f.J, f.Jpad, f.cov_type,
        f.variant, c->f_mu.as<float>(), c->f_cov.as<float>(), c->f_w.as<float>(), c->f_inv.as<float>(),
        c->f_pack.as<float>(), c->f_lls.as<float>(), f.lls_cap, f.tol, ctl, ctl_f, stop, stop_next);
    HGMM_HIP(c, hipGetLastError());
    f.launched++;
    return HGMM_OK;
}

}  // namespace hgmm

using namespace hgmm;

// ==========================================================================================
// C ABI
// ==========================================================================================
// mean of the per-point normalisers from the per-workgroup partial sums (fixed order), one workgroup
__global__ __launch_bounds__(256) void flat_mean_lpn_kernel(const double* __restrict__ partials, int nblocks, double n,
                                                            double* __restrict__ out) {
    __shared__ double sh[4];
    double acc = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += 256) acc += partials[i];
    acc = wave_sum_f64(acc);
    if (lane_id() == 0) sh[wave_in_block()] = acc;
    __syncthreads();
    if (threadIdx.x == 0) *out = (sh[0] + sh[1] + sh[2] + sh[3]) / n;
}

static int flat_estep_enqueue(hgmm_ctx* c, int cov_type, int variant, int J, const float* mu, const float* inv_std,
                              const float* w, float* dev_log_resp, float* dev_lpn, int32_t* dev_argmax, int* grid_out,
                              bool params_on_device = false) {
    HGMM_TRY(flat_check(c, cov_type, variant, J));
    HGMM_TRY(flat_setup(c, cov_type, variant, J));
    if (params_on_device) {
        // the packed table is formed straight from the caller's device arrays: no staging, no copy
        if (!mu || !inv_std || !w) return fail(c, HGMM_ERR_ARG, "device parameter array is NULL");
        const FlatState& f = c->flat;
        flat_pack_kernel<<<(f.Jpad + 255) / 256, 256, 0, c->stream>>>(f.J, f.Jpad, f.cov_type, f.variant, mu, inv_std, w,
                                                                    c->f_pack.as<float>());
    } else {
        HGMM_TRY(flat_upload(c, mu, inv_std, false, w));
        launch_pack(c);
    }
    int grid = 0;
    if (c->flat.chunked) {
        const FlatState& f = c->flat;
        ProfScope prof(c, HGMM_K_FLAT_ESTEP);
        HGMM_TRY(chunk_normalisers(c, dev_argmax != nullptr, dev_lpn, dev_argmax, true, &grid));
        if (dev_log_resp) {
            const int g2 = grid_for(c, c->n, 2);
            for (int ci = 0; ci < f.nchunks; ++ci)
                flat_chunk_write_kernel<<<g2, BLOCK, 0, c->stream>>>(
                    c->x_aos.as<float>(), c->f_pack.as<float>() + ci * CH_J, c->n, chunk_valid(f, ci), f.Jpad,
                    c->f_lpn2.as<float>(), dev_log_resp + ci * CH_J, (int64_t)J);
            HGMM_HIP(c, hipGetLastError());
        }
    } else {
        HGMM_TRY(launch_estep<true>(c, dev_log_resp, dev_lpn, dev_argmax, &grid));
    }
    *grid_out = grid;
    return HGMM_OK;
}

extern "C" int hgmm_flat_estep(hgmm_ctx* c, int cov_type, int variant, int J, const float* mu,
                               const float* inv_std, const float* w, float* dev_log_resp,
                               float* dev_lpn, int32_t* dev_argmax, double* mean_lpn_out) {
    HGMM_ENTER(c);
    int grid = 0;
    HGMM_TRY(flat_estep_enqueue(c, cov_type, variant, J, mu, inv_std, w, dev_log_resp, dev_lpn, dev_argmax, &grid));
    if (mean_lpn_out) {
        std::vector<double> h(grid);
        HGMM_HIP(c, hipMemcpyAsync(h.data(), c->f_lpn_partials.p, sizeof(double) * grid,
                                   hipMemcpyDeviceToHost, c->stream));
        HGMM_HIP(c, ctx_stream_sync(c));
        double t = 0.0;
        for (double v : h) t += v;
        *mean_lpn_out = t / (double)c->n;
    }
    return HGMM_OK;
}

extern "C" int hgmm_flat_estep_async(hgmm_ctx* c, int cov_type, int variant, int J, const float* mu,
                                     const float* inv_std, const float* w, float* dev_log_resp,
                                     float* dev_lpn, int32_t* dev_argmax, double* dev_mean_lpn) {
    HGMM_ENTER(c);
    int grid = 0;
    HGMM_TRY(flat_estep_enqueue(c, cov_type, variant, J, mu, inv_std, w, dev_log_resp, dev_lpn, dev_argmax, &grid));
    if (dev_mean_lpn) {
        flat_mean_lpn_kernel<<<1, 256, 0, c->stream>>>(c->f_lpn_partials.as<double>(), grid, (double)c->n, dev_mean_lpn);
        HGMM_HIP(c, hipGetLastError());
    }
    return HGMM_OK;                                   // nothing waited for: the caller's arrays were copied to the ring
}

extern "C" int hgmm_flat_estep_dev(hgmm_ctx* c, int cov_type, int variant, int J, const float* dev_mu,
                                   const float* dev_inv_std, const float* dev_w, float* dev_log_resp,
                                   float* dev_lpn, int32_t* dev_argmax, double* dev_mean_lpn) {
    HGMM_ENTER(c);
    int grid = 0;
    HGMM_TRY(flat_estep_enqueue(c, cov_type, variant, J, dev_mu, dev_inv_std, dev_w, dev_log_resp, dev_lpn, dev_argmax,
                                &grid, /*params_on_device=*/true));
    if (dev_mean_lpn) {
        flat_mean_lpn_kernel<<<1, 256, 0, c->stream>>>(c->f_lpn_partials.as<double>(), grid, (double)c->n, dev_mean_lpn);
        HGMM_HIP(c, hipGetLastError());
    }
    return HGMM_OK;
}

extern "C" int hgmm_pace_reset(hgmm_ctx* c) {
    HGMM_ENTER(c);
    PaceCtl& p = c->pace;
    p.target = 0.0;                                        // the next paced launch starts over (pace_target)
    p.steps_down = p.steps_up = p.ceilings_forgotten = 0;
    p.tail = p.head;                                       // launches still in flight are not judged any more
    return HGMM_OK;
}

extern "C" int hgmm_pace_info(hgmm_ctx* c, double* target_gbs_out, int* steps_down_out, int* steps_up_out) {
    HGMM_ENTER(c);
    pace_poll(c);
    const int fixed = c->cfg[CFG_ESTEP_TARGET_GBS];
    if (target_gbs_out) *target_gbs_out = fixed >= 0 ? (double)fixed : (c->pace.target > 0.0 ? c->pace.target : (double)ESTEP_TARGET_GBS);
    if (steps_down_out) *steps_down_out = c->pace.steps_down;
    if (steps_up_out) *steps_up_out = c->pace.steps_up;
    return HGMM_OK;
}

extern "C" int hgmm_flat_predict(hgmm_ctx* c, int cov_type, int variant, int J, const float* mu,
                                 const float* inv_std, const float* w, int32_t* dev_labels) {
    HGMM_ENTER(c);
    if (!c || !dev_labels) return c ? fail(c, HGMM_ERR_ARG, "dev_labels is NULL") : HGMM_ERR_ARG;
    HGMM_TRY(flat_check(c, cov_type, variant, J));
    HGMM_TRY(flat_setup(c, cov_type, variant, J));
    HGMM_TRY(flat_upload(c, mu, inv_std, false, w));
    launch_pack(c);
    int grid = 0;
    if (c->flat.chunked) return chunk_normalisers(c, true, nullptr, dev_labels, false, nullptr);
    HGMM_TRY(launch_estep<false>(c, nullptr, nullptr, dev_labels, &grid));
    return HGMM_OK;
}

extern "C" int hgmm_flat_predict_dev(hgmm_ctx* c, int cov_type, int variant, int J, const float* dev_mu,
                                     const float* dev_inv_std, const float* dev_w, int32_t* dev_labels) {
    HGMM_ENTER(c);
    if (!c || !dev_labels) return c ? fail(c, HGMM_ERR_ARG, "dev_labels is NULL") : HGMM_ERR_ARG;
    if (!dev_mu || !dev_inv_std || !dev_w) return fail(c, HGMM_ERR_ARG, "device parameter array is NULL");
    HGMM_TRY(flat_check(c, cov_type, variant, J));
    HGMM_TRY(flat_setup(c, cov_type, variant, J));
    const FlatState& f = c->flat;
    flat_pack_kernel<<<(f.Jpad + 255) / 256, 256, 0, c->stream>>>(f.J, f.Jpad, f.cov_type, f.variant, dev_mu, dev_inv_std,
                                                                dev_w, c->f_pack.as<float>());
    int grid = 0;
    if (c->flat.chunked) return chunk_normalisers(c, true, nullptr, dev_labels, false, nullptr);
    HGMM_TRY(launch_estep<false>(c, nullptr, nullptr, dev_labels, &grid));
    return HGMM_OK;
}

extern "C" int hgmm_flat_log_prob(hgmm_ctx* c, int cov_type, int J, const float* mu, const float* inv_std,
                                  float* dev_log_prob) {
    HGMM_ENTER(c);
    if (!c || !dev_log_prob) return c ? fail(c, HGMM_ERR_ARG, "dev_log_prob is NULL") : HGMM_ERR_ARG;
    // weights = 1 under flavour G (log 1 = 0, no eps) gives the bare log-density
    HGMM_TRY(flat_check(c, cov_type, cov_type == HGMM_COV_DIAG ? HGMM_VARIANT_G : HGMM_VARIANT_W, J));
    HGMM_TRY(flat_setup(c, cov_type, HGMM_VARIANT_G, J));
    std::vector<float> ones((size_t)J, 1.0f);
    HGMM_TRY(flat_upload(c, mu, inv_std, false, ones.data()));
    HGMM_HIP(c, ctx_stream_sync(c));   // `ones` is pageable host memory
    launch_pack(c);
    int grid = 0;
    if (c->flat.chunked) {
        const FlatState& f = c->flat;
        const int g2 = grid_for(c, c->n, 2);
        for (int ci = 0; ci < f.nchunks; ++ci)
            flat_chunk_write_kernel<<<g2, BLOCK, 0, c->stream>>>(
                c->x_aos.as<float>(), c->f_pack.as<float>() + ci * CH_J, c->n, chunk_valid(f, ci), f.Jpad, nullptr,
                dev_log_prob + ci * CH_J, (int64_t)J);
        HGMM_HIP(c, hipGetLastError());
        return HGMM_OK;
    }
    HGMM_TRY(launch_estep<false>(c, dev_log_prob, nullptr, nullptr, &grid));
    return HGMM_OK;
}

// The M-step's launches.  centre_hint: host [J,3] (staged), device [J,3] (hint_on_device) or NULL; the new parameters
// go to d_w [J], d_mu [J,3], d_cov [J,3] / [J] -- the context's own block or the caller's device arrays.
static int flat_mstep_enqueue(hgmm_ctx* c, int cov_type, int variant, int J, const float* dev_resp, int is_log,
                              const float* centre_hint, bool hint_on_device, float* d_w, float* d_mu, float* d_cov) {
    FlatState& f = c->flat;
    if (centre_hint && hint_on_device) {
        flat_hint_kernel<<<(f.Jpad + 255) / 256, 256, 0, c->stream>>>(centre_hint, J, f.Jpad, c->f_hint.as<float>());
    } else if (centre_hint) {
        HGMM_TRY(stage_h2d(c, c->f_mu.p, centre_hint, sizeof(float) * 3 * J));
        flat_hint_kernel<<<(f.Jpad + 255) / 256, 256, 0, c->stream>>>(c->f_mu.as<float>(), J, f.Jpad,
                                                                    c->f_hint.as<float>());
    } else {
        flat_hint_const_kernel<<<(f.Jpad + 255) / 256, 256, 0, c->stream>>>(0.f, 0.f, 0.f, f.Jpad,
                                                                          c->f_hint.as<float>());
    }
    // (workgroups per CU: two at J = 800 (three or four lose: mstep_nt_rr_sweep.log); short rows leave registers and
    //  memory-level parallelism unused -- N = 1e6: J = 64 0.133 -> 0.061 ms with 8, J = 100 0.153 -> 0.082, J = 200 0.183 -> 0.131
    //  with 4, J = 400 0.271 -> 0.257; from J = 512 on two again: profiles/r04/small_j_grids.log)
    const int bpc_m = c->n >= 400000 ? (J <= 128 ? 8 : (J <= 448 ? 4 : 2)) : 2;
    const int grid = grid_for(c, c->n, bpc_m);
    const int rr = 1;                               // rows dealt round-robin: 0.480 against 0.492 ms (profiles/r04/mstep_nt_rr_sweep.log)
    const float* X = c->x_aos.as<float>();
    float* part = c->f_partials.as<float>();
    const float* hint = c->f_hint.as<float>();
    int valid_j = 0;
    // non-temporal loads: 6.1 -> 6.8 TB/s at J = 800, a gain or a tie for every row length that is a whole number of
    // 16-byte pieces -- and a loss when rows straddle them (J = 513: 0.380 vs 0.351 ms, J = 37: 0.176 vs 0.142: a line shared by
    // two rows is then fetched for each of them; profiles/r04/mstep_nt_by_J.log)
    const bool ntload = J % 4 == 0;
#define MSTEP_LAUNCH(A, B, RESP, HINT, JV, PART)                                                               \
    do {                                                                                                       \
        if (is_log) {                                                                                          \
            if (ntload) flat_mstep_kernel<A, B, true, true><<<grid, BLOCK, 0, c->stream>>>(X, RESP, HINT, c->n, JV, f.Jpad, PART, (int64_t)J, rr);  \
            else flat_mstep_kernel<A, B, true, false><<<grid, BLOCK, 0, c->stream>>>(X, RESP, HINT, c->n, JV, f.Jpad, PART, (int64_t)J, rr);       \
        } else {                                                                                               \
            if (ntload) flat_mstep_kernel<A, B, false, true><<<grid, BLOCK, 0, c->stream>>>(X, RESP, HINT, c->n, JV, f.Jpad, PART, (int64_t)J, rr); \
            else flat_mstep_kernel<A, B, false, false><<<grid, BLOCK, 0, c->stream>>>(X, RESP, HINT, c->n, JV, f.Jpad, PART, (int64_t)J, rr);      \
        }                                                                                                      \
    } while (0)
    if (f.chunked) {
        ProfScope prof(c, HGMM_K_FLAT_MSTEP);
        for (int ci = 0; ci < f.nchunks; ++ci)
            MSTEP_LAUNCH(0, CH_SLOTS, dev_resp + ci * CH_J, hint + ci * CH_J, chunk_valid(f, ci), part + ci * CH_J);
        valid_j = f.nchunks * CH_J;
    } else {
        ProfScope prof(c, HGMM_K_FLAT_MSTEP);
        int nv4, nv1;
        pick_layout(J, &nv4, &nv1);
#define MSTEP_M(A, B)                                                                              \
    do {                                                                                           \
        MSTEP_LAUNCH(A, B, dev_resp, hint, J, part);                                               \
        valid_j = 256 * A + 64 * B;                                                                \
    } while (0)
        LAYOUT_DISPATCH(nv4, nv1, MSTEP_M);
#undef MSTEP_M
    }
#undef MSTEP_LAUNCH
    HGMM_HIP(c, hipGetLastError());
    c->flat.last_kernel = 2;                               // (the next E-step's grid looks at this, estep_rows_grid)
    c->flat.idle_since_launch = false;
    HGMM_TRY(launch_reduce(c, grid, valid_j, false, nullptr));
    if (f.chunked)
        flat_finalize_mb_kernel<<<(f.Jpad + 255) / 256, 256, 0, c->stream>>>(
            c->f_stats.as<double>(), hint, J, f.Jpad, cov_type, variant, d_mu, d_cov, d_w, nullptr, nullptr, nullptr);
    else
        flat_finalize_kernel<<<f.Jpad / 256, 256, 0, c->stream>>>(
            c->f_stats.as<double>(), hint, J, f.Jpad, cov_type, variant, d_mu, d_cov, d_w, nullptr, nullptr,
            nullptr, 0, 0.f, nullptr, nullptr, c->f_ctl.as<int>() + 16, nullptr);
    HGMM_HIP(c, hipGetLastError());
    return HGMM_OK;
}

extern "C" int hgmm_flat_mstep_dev(hgmm_ctx* c, int cov_type, int variant, int J, const float* dev_resp, int is_log,
                                   const float* dev_centre_hint, float* dev_w, float* dev_mu, float* dev_cov) {
    HGMM_ENTER(c);
    if (!c || !dev_resp) return c ? fail(c, HGMM_ERR_ARG, "dev_resp is NULL") : HGMM_ERR_ARG;
    if (!dev_w || !dev_mu || !dev_cov) return fail(c, HGMM_ERR_ARG, "device output array is NULL");
    HGMM_TRY(flat_check(c, cov_type, variant, J));
    HGMM_TRY(flat_setup(c, cov_type, variant, J));
    return flat_mstep_enqueue(c, cov_type, variant, J, dev_resp, is_log, dev_centre_hint, true, dev_w, dev_mu, dev_cov);
}

extern "C" int hgmm_flat_mstep(hgmm_ctx* c, int cov_type, int variant, int J, const float* dev_resp,
                               int is_log, const float* centre_hint, float* w_out, float* mu_out,
                               float* cov_out) {
    HGMM_ENTER(c);
    if (!c || !dev_resp) return c ? fail(c, HGMM_ERR_ARG, "dev_resp is NULL") : HGMM_ERR_ARG;
    HGMM_TRY(flat_check(c, cov_type, variant, J));
    HGMM_TRY(flat_setup(c, cov_type, variant, J));
    FlatState& f = c->flat;
    HGMM_TRY(flat_mstep_enqueue(c, cov_type, variant, J, dev_resp, is_log, centre_hint, false, c->f_w.as<float>(),
                                c->f_mu.as<float>(), c->f_cov.as<float>()));
    // results: [cov | mu | w] is one contiguous range of the parameter block: ONE DMA packet into the pinned ring, ONE
    // synchronisation, then plain memcpys
    const size_t b_mu = sizeof(float) * 3 * J, b_cov = sizeof(float) * cov_elems(cov_type, J), b_w = sizeof(float) * J;
    const size_t span = sizeof(float) * (6 * (size_t)f.Jpad + J);
    void* st = nullptr;
    HGMM_TRY(stage_reserve(c, span, &st));
    char* h = static_cast<char*>(st);
    HGMM_HIP(c, hipMemcpyAsync(h, c->f_cov.p, span, hipMemcpyDeviceToHost, c->stream));
    HGMM_HIP(c, ctx_stream_sync(c));
    c->h_stage_off = 0;                                    // the stream is idle: every region of the ring is free
    std::memcpy(cov_out, h, b_cov);
    std::memcpy(mu_out, h + sizeof(float) * 3 * (size_t)f.Jpad, b_mu);
    std::memcpy(w_out, h + sizeof(float) * 6 * (size_t)f.Jpad, b_w);
    return HGMM_OK;
}

// ---- elementwise float32 arithmetic on small device arrays -----------------------------------------------------
// What `inv_cov = 1 / (xp.sqrt(covariances + 1e-6) + eps)` (gmm_impl.py:134) is under CuPy: a few tiny kernels on
// device arrays, so that a caller's own EM loop never brings the parameters to the host.  IEEE float32 operations
// (hipcc rounds fp32 divide and sqrt correctly by default): the results are bitwise NumPy's.
__global__ void flat_elementwise_kernel(int op, int64_t n, const float* __restrict__ a, const float* __restrict__ b,
                                        float s, float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = a[i];
    const float y = b ? b[i] : s;
    float r;
    switch (op) {
        case HGMM_EW_ADD: r = x + y; break;
        case HGMM_EW_SUB: r = x - y; break;
        case HGMM_EW_RSUB: r = y - x; break;
        case HGMM_EW_MUL: r = x * y; break;
        case HGMM_EW_DIV: r = x / y; break;
        case HGMM_EW_RDIV: r = y / x; break;
        case HGMM_EW_SQRT: r = sqrtf(x); break;
        case HGMM_EW_EXP: r = expf(x); break;
        case HGMM_EW_LOG: r = logf(x); break;
        case HGMM_EW_MAX: r = fmaxf(x, y); break;
        case HGMM_EW_MIN: r = fminf(x, y); break;
        default: r = x; break;
    }
    out[i] = r;
}
extern "C" int hgmm_elementwise_f32(hgmm_ctx* c, int op, int64_t n, const float* dev_a, const float* dev_b,
                                    float scalar, float* dev_out) {
    HGMM_ENTER(c);
    if (op < 0 || op > HGMM_EW_MIN) return fail(c, HGMM_ERR_ARG, "elementwise op %d", op);
    if (n < 0 || (n > 0 && (!dev_a || !dev_out))) return fail(c, HGMM_ERR_ARG, "elementwise: NULL array");
    if (n == 0) return HGMM_OK;
    HGMM_HIP(c, hipSetDevice(c->device));
    flat_elementwise_kernel<<<(unsigned)((n + 255) / 256), 256, 0, c->stream>>>(op, n, dev_a, dev_b, scalar, dev_out);
    HGMM_HIP(c, hipGetLastError());
    return HGMM_OK;
}


// Everything a fit needs before its first iteration, in ONE launch: initial inv_std from the covariances, the packed
// table from them, the loop's control words.  (Three launches before round 4; at the start of a call -- the stream has
// been idle -- each of them came with ~13 us of launch latency: 40 us of a 620 us bun000 fit.)
__global__ void flat_begin_kernel(int J, int Jpad, int cov_type, int variant, const float* __restrict__ mu,
                                  const float* __restrict__ cov, const float* __restrict__ w, float* __restrict__ inv,
                                  float* __restrict__ pack, int* __restrict__ ctl, float* __restrict__ ctl_f) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j == 0) {
        ctl[0] = 0; ctl[1] = 0; ctl[2] = 0; ctl[3] = 0;
        ctl_f[0] = -__builtin_huge_valf();
    }
    if (j >= Jpad) return;
    float m[3] = {0.f, 0.f, 0.f}, i[3] = {0.f, 0.f, 0.f}, wj = 0.f;
    if (j < J) {
        // initial inv_std = 1/sqrt(cov)  (gmm_waymo gmm_impl.py:122, gmmreg_gpu gmm_impl.py:67)
        if (cov_type == HGMM_COV_DIAG) {
#pragma unroll
            for (int d = 0; d < 3; ++d) { i[d] = 1.0f / sqrtf(cov[3 * j + d]); inv[3 * j + d] = i[d]; }
        } else {
            i[0] = i[1] = i[2] = 1.0f / sqrtf(cov[j]);
            inv[j] = i[0];
        }
        m[0] = mu[3 * j + 0]; m[1] = mu[3 * j + 1]; m[2] = mu[3 * j + 2];
        wj = w[j];
    }
    pack_values(j, j < J, Jpad, variant, m, i, wj, pack);
}

extern "C" int hgmm_flat_train_begin(hgmm_ctx* c, int cov_type, int variant, int J, float tol,
                                     const float* mu, const float* cov, const float* w,
                                     int lls_capacity) {
    HGMM_ENTER(c);
    HGMM_TRY(flat_check(c, cov_type, variant, J));
    HGMM_TRY(flat_setup(c, cov_type, variant, J));
    if (lls_capacity < 1) lls_capacity = 1;
    HGMM_TRY(ensure(c, c->f_lls, sizeof(float) * lls_capacity));
    FlatState& f = c->flat;
    f.tol = tol; f.lls_cap = lls_capacity; f.launched = 0; f.active = true;
    HGMM_TRY(flat_upload(c, mu, cov, true, w));
    int* ctl = c->f_ctl.as<int>();
    flat_begin_kernel<<<(f.Jpad + 255) / 256, 256, 0, c->stream>>>(J, f.Jpad, cov_type, variant, c->f_mu.as<float>(),
                                                                  c->f_cov.as<float>(), c->f_w.as<float>(),
                                                                  c->f_inv.as<float>(), c->f_pack.as<float>(), ctl,
                                                                  reinterpret_cast<float*>(ctl + 8));
    HGMM_HIP(c, hipGetLastError());
    return HGMM_OK;
}

extern "C" int hgmm_flat_train_step(hgmm_ctx* c, int iters) {
    HGMM_ENTER(c);
    if (!c->flat.active) return fail(c, HGMM_ERR_STATE, "hgmm_flat_train_step before hgmm_flat_train_begin");
    for (int i = 0; i < iters; ++i) HGMM_TRY(enqueue_em_iteration(c));
    return HGMM_OK;
}

extern "C" int hgmm_flat_train_end(hgmm_ctx* c, float* mu, float* cov, float* w, float* inv_std_out,
                                   float* lls_out, int* n_iter_out, int* converged_out) {
    HGMM_ENTER(c);
    FlatState& f = c->flat;
    if (!f.active) return fail(c, HGMM_ERR_STATE, "hgmm_flat_train_end before hgmm_flat_train_begin");
    // The results come back through the pinned ring: the whole parameter block [cov | mu | w | inv], the control words
    // and the head of the lls trace are three DMA packets that queue behind the last iteration at once, ONE
    // synchronisation, then plain memcpys.  (Copied straight into the caller's pageable arrays every one of the six
    // copies was staged by the runtime and waited for in turn, ~17 us each: a fifth of a 20-iteration bun000 fit.)
    int ctl[4] = {0, 0, 0, 0};
    const int J = f.J;
    const size_t ce = cov_elems(f.cov_type, J);
    const size_t span = sizeof(float) * 10 * (size_t)f.Jpad;
    const int eager = f.lls_cap < 16384 ? f.lls_cap : 16384;
    int n_it = 0;
    if (span + 512 + sizeof(float) * (size_t)eager <= STAGE_RING_BYTES / 2) {
        void *s_blk = nullptr, *s_ctl = nullptr, *s_lls = nullptr;
        HGMM_TRY(stage_reserve(c, span, &s_blk));
        HGMM_TRY(stage_reserve(c, sizeof ctl, &s_ctl));
        HGMM_TRY(stage_reserve(c, sizeof(float) * (size_t)eager, &s_lls));
        HGMM_HIP(c, hipMemcpyAsync(s_blk, c->f_block.p, span, hipMemcpyDeviceToHost, c->stream));
        HGMM_HIP(c, hipMemcpyAsync(s_ctl, c->f_ctl.p, sizeof ctl, hipMemcpyDeviceToHost, c->stream));
        HGMM_HIP(c, hipMemcpyAsync(s_lls, c->f_lls.p, sizeof(float) * (size_t)eager, hipMemcpyDeviceToHost, c->stream));
        HGMM_HIP(c, ctx_stream_sync(c));
        c->h_stage_off = 0;                                // the stream is idle: every region of the ring is free
        std::memcpy(ctl, s_ctl, sizeof ctl);
        const float* blk = static_cast<const float*>(s_blk);
        if (cov) std::memcpy(cov, blk, sizeof(float) * ce);
        if (mu) std::memcpy(mu, blk + 3 * (size_t)f.Jpad, sizeof(float) * 3 * J);
        if (w) std::memcpy(w, blk + 6 * (size_t)f.Jpad, sizeof(float) * J);
        if (inv_std_out) std::memcpy(inv_std_out, blk + 7 * (size_t)f.Jpad, sizeof(float) * ce);
        n_it = ctl[1];
        if (lls_out && n_it > 0) {
            const int cnt = n_it < f.lls_cap ? n_it : f.lls_cap;
            std::memcpy(lls_out, s_lls, sizeof(float) * (size_t)(cnt < eager ? cnt : eager));
            if (cnt > eager)
                HGMM_HIP(c, hipMemcpy(lls_out + eager, c->f_lls.as<float>() + eager, sizeof(float) * (size_t)(cnt - eager),
                                      hipMemcpyDeviceToHost));
        }
    } else {                                               // (very large J: the block does not fit the ring)
        HGMM_HIP(c, hipMemcpyAsync(ctl, c->f_ctl.p, sizeof ctl, hipMemcpyDeviceToHost, c->stream));
        if (mu) HGMM_HIP(c, hipMemcpyAsync(mu, c->f_mu.p, sizeof(float) * 3 * J, hipMemcpyDeviceToHost, c->stream));
        if (cov) HGMM_HIP(c, hipMemcpyAsync(cov, c->f_cov.p, sizeof(float) * ce, hipMemcpyDeviceToHost, c->stream));
        if (w) HGMM_HIP(c, hipMemcpyAsync(w, c->f_w.p, sizeof(float) * J, hipMemcpyDeviceToHost, c->stream));
        if (inv_std_out) HGMM_HIP(c, hipMemcpyAsync(inv_std_out, c->f_inv.p, sizeof(float) * ce, hipMemcpyDeviceToHost, c->stream));
        HGMM_HIP(c, ctx_stream_sync(c));
        n_it = ctl[1];
        if (lls_out && n_it > 0) {
            const int cnt = n_it < f.lls_cap ? n_it : f.lls_cap;
            HGMM_HIP(c, hipMemcpy(lls_out, c->f_lls.p, sizeof(float) * cnt, hipMemcpyDeviceToHost));
        }
    }
    if (n_iter_out) *n_iter_out = n_it;
    if (converged_out) *converged_out = ctl[2];
    f.active = false;
    return HGMM_OK;
}

extern "C" int hgmm_flat_train(hgmm_ctx* c, int cov_type, int variant, int J, int max_iter, float tol,
                               float* mu, float* cov, float* w, float* inv_std_out, float* lls_out,
                               int* n_iter_out, int* converged_out) {
    HGMM_ENTER(c);
    if (max_iter < 0) return fail(c, HGMM_ERR_ARG, "max_iter < 0");
    HGMM_TRY(hgmm_flat_train_begin(c, cov_type, variant, J, tol, mu, cov, w, max_iter));
    HGMM_TRY(hgmm_flat_train_step(c, max_iter));
    return hgmm_flat_train_end(c, mu, cov, w, inv_std_out, lls_out, n_iter_out, converged_out);
}

extern "C" int hgmm_flat_stats(hgmm_ctx* c, int cov_type, int variant, int J, const float* mu,
                               const float* inv_std, const float* w, double* stats_out,
                               double* sum_lpn_out, double* n_points_out) {
    HGMM_ENTER(c);
    HGMM_TRY(flat_check(c, cov_type, variant, J));
    HGMM_TRY(flat_setup(c, cov_type, variant, J));
    HGMM_TRY(flat_upload(c, mu, inv_std, false, w));
    launch_pack(c);
    int grid = 0, valid_j = 0;
    if (c->flat.chunked) {
        int n_lpn = 0;
        HGMM_TRY(chunk_normalisers(c, false, nullptr, nullptr, true, &n_lpn));
        HGMM_TRY(chunk_statistics(c, nullptr, &grid));
        HGMM_TRY(launch_reduce(c, grid, c->flat.nchunks * CH_J, true, nullptr, n_lpn));
    } else {
        HGMM_TRY(launch_fused(c, nullptr, &grid, &valid_j));
        HGMM_TRY(launch_reduce(c, grid, valid_j, true, nullptr));
    }
    const FlatState& f = c->flat;
    std::vector<double> h((size_t)FLAT_NSTAT * f.Jpad + 2);
    HGMM_HIP(c, hipMemcpyAsync(h.data(), c->f_stats.p, sizeof(double) * h.size(), hipMemcpyDeviceToHost, c->stream));
    HGMM_HIP(c, ctx_stream_sync(c));
    if (stats_out)
        for (int j = 0; j < J; ++j)
            for (int s = 0; s < FLAT_NSTAT; ++s) stats_out[(size_t)j * FLAT_NSTAT + s] = h[(size_t)s * f.Jpad + j];
    if (sum_lpn_out) *sum_lpn_out = h[(size_t)FLAT_NSTAT * f.Jpad];
    if (n_points_out) *n_points_out = h[(size_t)FLAT_NSTAT * f.Jpad + 1];
    return HGMM_OK;
}
