// KMeans initialiser of the GMMReg flavour for gfx950, float64 arithmetic.
//
// The reference seeds its flat EM with scikit-learn on the host
// (src/python/gmmreg_gpu/gmm_impl.py:18-24: KMeans(k, random_state=1, max_iter=50, n_init=1),
// called on the caller's float64 Open3D points, gmm.py:79).  scikit-learn's algorithm is
// k-means++ seeding (greedy, 2 + log k local trials per centre) followed by Lloyd iterations;
// both are O(N k) per step and dominate a fit at Waymo scale, so they run here on the resident
// float64 cloud (x_soa64, lanes across points).  The random draws, the stop rule and the
// empty-cluster relocation stay on the host (kmeans.py), exactly as scikit-learn orders them.
//
//   kmpp_first / kmpp_pick / kmpp_eval / kmpp_select / kmpp_update
//       one seeding step per centre, no host round trip: closest[i] = min_c d2(x_i, c) is kept in
//       HBM with one partial sum per 256-point block; 'pick' scans the block sums and walks into
//       the block that holds each random threshold (searchsorted on the running sum), 'eval'
//       scores the candidates, 'select' takes the one with the lowest potential, 'update' folds
//       it into closest[].
//   kmeans_assign_kernel   Lloyd E-step: lanes across points (2 per thread), centres through
//       wave-uniform scalar loads, strict '<' so the first nearest centre wins.
//   kmeans_accum_kernel    per-cluster sums: lanes across CENTRES (one wave walks a contiguous
//       run of points; the point's label is wave-uniform, the owning lane adds into its
//       registers), waves combined through LDS in fixed order: no atomics, deterministic.
//   kmeans_reduce_kernel   fixed-order fp64 sum of the workgroup partials -> [k][4] (+ inertia,
//       + number of changed labels) = the buffer an RCCL all-reduce works on.
#include "hgmm_ctx.h"
#include "wave_ops.h"

#include <algorithm>
#include <cmath>
#include <vector>

namespace hgmm {

constexpr int KM_BLOCK = 256;
constexpr int KM_MAX_TRIALS = 16;
constexpr int KM_ACC_SLOTS = 16;                   // 64-centre slots per accumulate pass
constexpr int KM_MAX_K = 16384;

struct OpAddI { __device__ __forceinline__ int operator()(int a, int b) const { return a + b; } };

static unsigned km_nblk(int64_t n, int b) { return (unsigned)((n + b - 1) / b); }

__device__ __forceinline__ double dist2(double x, double y, double z, double cx, double cy, double cz) {
    const double dx = x - cx, dy = y - cy, dz = z - cz;
    // explicit fused operations: every call site must round identically (the Lloyd assignment recognises the winner
    // of a group of centres by evaluating the group a second time and comparing for equality)
    return fma(dz, dz, fma(dy, dy, dx * dx));
}

// (readlane_f64: wave_ops.h)

// inclusive prefix sum over the 64 lanes: Hillis-Steele inside each DPP row of 16 (row_shr 1, 2, 4, 8 with zero
// fill), then the row totals carried across with row_bcast 15 / 31 -- no LDS round trips (ds_bpermute costs a
// dependent ~130 cycles per step, and the seeding tail is nothing but a latency chain)
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ double dpp_f64_zero(double v) {
    const long long b = __double_as_longlong(v);
    int lo = (int)(b & 0xffffffffLL), hi = (int)(b >> 32);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xF, true);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double wave_scan_f64(double v) {
    v += dpp_f64_zero<0x111>(v);                  // row_shr:1
    v += dpp_f64_zero<0x112>(v);                  // row_shr:2
    v += dpp_f64_zero<0x114>(v);                  // row_shr:4
    v += dpp_f64_zero<0x118>(v);                  // row_shr:8
    v += dpp_f64_zero<DPP_ROW_BCAST15, 0xA>(v);   // rows 1, 3 += total of rows 0, 2
    v += dpp_f64_zero<DPP_ROW_BCAST31, 0xC>(v);   // rows 2, 3 += total of rows 0 + 1
    return v;
}

// sum of one value per thread over a 256-thread workgroup, fixed order; valid in thread 0
__device__ __forceinline__ double block_sum_256(double v, double* sh /* [4] */) {
    const double w = wave_sum_f64(v);
    if (lane_id() == 0) sh[wave_in_block()] = w;
    __syncthreads();
    const double t = (sh[0] + sh[1]) + (sh[2] + sh[3]);
    __syncthreads();
    return t;
}

// ------------------------------------------------------------------------------------------
// k-means++ seeding
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(KM_BLOCK) void kmpp_first_kernel(const double* __restrict__ xs, int64_t n,
                                                              int64_t n_pad, int64_t first,
                                                              double* __restrict__ closest,
                                                              double* __restrict__ bsum,
                                                              double* __restrict__ centres,
                                                              int64_t* __restrict__ ids) {
    __shared__ double sh[4];
    const int64_t i = (int64_t)blockIdx.x * KM_BLOCK + threadIdx.x;
    const double cx = xs[first], cy = xs[n_pad + first], cz = xs[2 * n_pad + first];
    double d = 0.0;
    if (i < n) d = dist2(xs[i], xs[n_pad + i], xs[2 * n_pad + i], cx, cy, cz);
    closest[i] = d;
    const double t = block_sum_256(d, sh);
    if (threadIdx.x == 0) {
        bsum[blockIdx.x] = t;
        if (blockIdx.x == 0) {
            centres[0] = cx; centres[1] = cy; centres[2] = cz;
            ids[0] = first;
        }
    }
}

// ------------------------------------------------------------------------------------------
// Fused seeding step: ONE pass over the points per centre instead of two (eval + update).
//
// The potential of candidate t,  sum_i min(closest_i, |x_i - cand_t|^2),  is accumulated per 256-point block
// (`part[t][b]`) -- and those block sums ARE the block sums of the closest-distance array after cand_t has been
// accepted.  So the winner's row of `part` replaces the separate update pass as far as the next draw's prefix
// search is concerned; the element-wise minimum itself is folded into the NEXT step's pass (`fold` = the centre
// accepted last), and inside the one block the search lands in it is formed on the fly.  Per centre:
//   kmpp_step_kernel   closest <- min(closest, d(., centre fold)); part[t][b] for the T candidates, and their sums
//                      over the workgroup's 16 blocks, part16[t][g]                              (grid G = B / 16)
//   kmpp_tail_kernel   winner = first arg-min_t sum_g part16[t][g]  ->  centre j;  prefix scan of its part16 row;
//                      the T candidates of centre j + 1 by np.searchsorted(cumsum, rand * pot): group, then block
//                      inside the group (16 sums of `part`), then point inside the block         (one workgroup)
// The tail is ONE workgroup on ONE CU reading what 245 other CUs wrote a moment ago: every cache line it touches
// is a trip through the fabric, and a CU keeps only so many of them in flight.  The two-level layout keeps the
// lines on its critical path to ~120 (all part16 rows) + 2 per draw + the 64 lines of the block a draw lands in;
// a flat per-block table (3907 sums per row at N = 1M) cost twice the time.
// Round 3: step and tail as ONE launch per centre (kmpp_fused_kernel: every workgroup of the pass runs the previous
// centre's tail for itself first, 245 times the same arithmetic on the same numbers) -- 16.3 -> 12.8 us per centre,
// 13.1 -> 10.2 ms for k = 800 at N = 1M.  Where the 12 us of a launch go (HGMM_KMPP_DEBUG=1, thread 0's clock): first
// round of loads + the candidates' totals 2.0, arg-min 1.3, prefix scan of the winner's row 1.1, group search 0.6, the
// group's 16 block sums 0.9, the block's points 0.9, hand-over 0.5 -- then the pass itself, whose operands are in
// registers by then: 4.6 us of fp64 issue (1M points x 8 candidates x 8 instructions, 16 waves on each CU) and 2 us
// until the last wave of the workgroup is through.  The two-launch form stays for clouds beyond 16.7 M points.
// Fixed summation orders throughout, hence the same seeds run to run.  (The four-kernel form of round 1 -- kmpp_pick /
// eval / select / update behind HGMM_KMPP_UNFUSED -- left the library in round 6 with its switch.)
// ------------------------------------------------------------------------------------------
constexpr int KM_GROUP = 16;                             // 256-point blocks per step workgroup (one per wave)
constexpr int KM_STEP_BLOCK = 64 * KM_GROUP;

// group sums of the first kernel's block sums, in the order the step kernel uses (sequential over the 16 blocks)
__global__ void kmpp_group_kernel(const double* __restrict__ bsum, int B, int G, double* __restrict__ g16) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= G) return;
    double s = 0.0;
    for (int q = 0; q < KM_GROUP; ++q) s += (g * KM_GROUP + q < B) ? bsum[g * KM_GROUP + q] : 0.0;
    g16[g] = s;
}

// One WAVE per 256-point block (4 consecutive points per lane): the block sum of a candidate is one wave reduction
// of the lanes' 4-point partial sums.
template <int TMAX>                       // unrolled candidate slots (8 covers k <= 1096 under scikit-learn's 2 + log k)
__global__ __launch_bounds__(KM_STEP_BLOCK) void kmpp_step_kernel(const double* __restrict__ xs, int64_t n, int64_t n_pad,
                                                                  double* __restrict__ closest,
                                                                  const double* __restrict__ centres, int fold,
                                                                  const double* __restrict__ cand_xyz, int T,
                                                                  double* __restrict__ part, int B,
                                                                  double* __restrict__ part16) {
    __shared__ double shg[KM_MAX_TRIALS][KM_GROUP];
    const int lane = lane_id();
    const int64_t blk = (int64_t)blockIdx.x * KM_GROUP + wave_in_block();     // 256-point block of this wave
    const bool have = blk < B;
    const int64_t i0 = (have ? blk : 0) * KM_BLOCK + 4 * lane;                 // n_pad is a multiple of 256
    // every candidate's coordinates through ONE batch of scalar loads (index clamped, not branched on, so that the
    // loads can leave the per-candidate blocks): the tail's CU wrote them a moment ago and each cache line of them
    // is a trip through the fabric -- a load per loop iteration would chain those trips
    // (issued before the points' loads so that the two kinds of trip overlap)
    double cc[TMAX][3];
#pragma unroll
    for (int t = 0; t < TMAX; ++t) {
        const int tt = t < T ? t : T - 1;
        cc[t][0] = cand_xyz[3 * tt]; cc[t][1] = cand_xyz[3 * tt + 1]; cc[t][2] = cand_xyz[3 * tt + 2];
    }
    double x[4], y[4], z[4], cl[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { x[q] = xs[i0 + q]; y[q] = xs[n_pad + i0 + q]; z[q] = xs[2 * n_pad + i0 + q]; cl[q] = closest[i0 + q]; }
    if (fold >= 0 && have) {
        const double fx = centres[3 * fold], fy = centres[3 * fold + 1], fz = centres[3 * fold + 2];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const double d = dist2(x[q], y[q], z[q], fx, fy, fz);
            if (i0 + q < n && d < cl[q]) { cl[q] = d; closest[i0 + q] = d; }   // few points move once j is large
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) if (!have || !(i0 + q < n)) cl[q] = 0.0;      // padding contributes min(0, d) = 0
#pragma unroll
    for (int t = 0; t < TMAX; ++t) {
        if (t < T) {
            const double cx = cc[t][0], cy = cc[t][1], cz = cc[t][2];
            double sacc = 0.0;
#pragma unroll
            for (int q = 0; q < 4; ++q) sacc += fmin(cl[q], dist2(x[q], y[q], z[q], cx, cy, cz));
            const double w = wave_sum_f64(sacc);
            if (lane == 0) {
                if (have) part[(size_t)t * B + blk] = w;
                shg[t][wave_in_block()] = have ? w : 0.0;
            }
        }
    }
    __syncthreads();
    // the workgroup's 16 block sums, added in block order: what the tail's winner selection and group search read
    if ((int)threadIdx.x < T) {
        double s = 0.0;
#pragma unroll
        for (int q = 0; q < KM_GROUP; ++q) s += shg[threadIdx.x][q];
        part16[(size_t)threadIdx.x * gridDim.x + blockIdx.x] = s;
    }
}

// One workgroup of 1024 threads.  `select`: pick the winner among the T candidates of centre j from `part16`.
// `draw`: draw the T candidates of the next centre from the winner's rows of `part16` / `part` (or, for the first
// centre, the first kernel's sums `g0` / `bsum0`); `closest` lacks the fold of centre `pend` (-1: it is current),
// applied on the fly inside the blocks the draws land in.
constexpr int KM_TAIL_LDS_GROUPS = 4096;                 // group-prefix table kept in LDS up to this many groups
// HGMM_KMPP_DEBUG=1: thread 0 of workgroups 0 and 128 adds the time since its kernel started (100 MHz ticks) at nine
// points of kmpp_fused_kernel into dbg[wg slot][9] (tools/kmpp_prof.py prints the averages per centre)
__device__ __forceinline__ void km_stamp(unsigned long long* dbg, int k, unsigned long long t0) {
    if (dbg && threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == 128))
        atomicAdd(dbg + (blockIdx.x ? 16 : 0) + k, (unsigned long long)wall_clock64() - t0);
}
struct KmTailShared {                                    // the tail's LDS (declared by the kernel that runs the body)
    double wsum[KM_MAX_TRIALS][16];
    double wave_tot[16];
    double seg_end[1024];
    double pre_sh[KM_TAIL_LDS_GROUPS];                   // inclusive prefix of the group sums (global `gprefix` beyond)
    double total_sh;
    int best_sh;
};
// The candidates of centre c and their potentials live in buffer c & 1 (`cand_in` / `part` are the ones being judged,
// `cand_out` the ones being drawn): kmpp_fused_kernel has every workgroup run this body while others already write
// the next potentials.  `writer`: this workgroup stores the results to memory (always, in the one-workgroup kernel);
// `sh_cand` (LDS [T][3], may be null) receives the drawn candidates' coordinates for the code that follows in the
// same workgroup; (pcx, pcy, pcz) = the accepted centre's coordinates in every thread (select).
template <int TMAX>
__device__ __forceinline__ void kmpp_tail_body(const double* __restrict__ xs, int64_t n, int64_t n_pad,
                                               const double* closest,
                                               const double* __restrict__ part,
                                               const double* __restrict__ part16, int G,
                                               const double* __restrict__ bsum0,
                                               const double* __restrict__ g0, int B, int T, int j,
                                               int select, int draw, const double* __restrict__ rand_c,
                                               const int64_t* cand_in, const double* cand_xyz_in,
                                               int64_t* cand_out, double* cand_xyz_out,
                                               double* __restrict__ centres,
                                               int64_t* __restrict__ ids, double* gprefix, bool writer,
                                               double* sh_cand, double& pcx, double& pcy, double& pcz,
                                               KmTailShared& ts, unsigned long long* dbg = nullptr,
                                               unsigned long long t_start = 0, double* pre_pts = nullptr,
                                               int64_t pre_i0 = 0) {
    auto& wsum = ts.wsum;
    auto& wave_tot = ts.wave_tot;
    auto& seg_end = ts.seg_end;
    auto& pre_sh = ts.pre_sh;
    double& total_sh = ts.total_sh;
    int& best_sh = ts.best_sh;
    const int tid = threadIdx.x, lane = lane_id(), wave = wave_in_block();
    const bool in_lds = G <= KM_TAIL_LDS_GROUPS;
    const int seg = (G + 1023) / 1024;                      // groups per thread in the prefix scan (1 up to N = 4M)
    const int g_lo = tid * seg, g_hi = min(G, g_lo + seg);
    // waves 0 .. nwg-1 own groups; the totals of the others are exact zeros, and x + 0.0 == x: leaving them out of the
    // fixed-order sums below changes no bit and takes up to 12 dependent LDS reads out of two steps of the chain
    const int nwg = min(16, (G + 64 * seg - 1) / (64 * seg));
    double loc = 0.0;                                       // the thread's run of the winner's group sums
    // Everything whose ADDRESS is known now is requested now, next to the first round of loads, instead of one trip
    // each further down the chain: this wave's uniform, and every candidate's id and coordinates (lane t holds
    // candidate t; the winner's are picked out of the registers once it is known).
    const double u_rand = draw ? rand_c[wave < T ? wave : T - 1] : 0.0;
    const int tl = lane < T ? lane : 0;
    int64_t my_cand = 0;
    double my_cx = 0.0, my_cy = 0.0, my_cz = 0.0;
    if (select) {
        my_cand = cand_in[tl];
        my_cx = cand_xyz_in[3 * tl]; my_cy = cand_xyz_in[3 * tl + 1]; my_cz = cand_xyz_in[3 * tl + 2];
    }
    if (select) {
        double acc[TMAX];
#pragma unroll
        for (int t = 0; t < TMAX; ++t) acc[t] = 0.0;
        // thread tid owns groups [g_lo, g_hi) of every row: the totals AND (for the winner) its piece of the prefix
        // scan come from this one round of loads
        // (row index clamped instead of branched on: the loads of one group are issued together, not one trip each)
        for (int g = g_lo; g < g_hi; ++g) {
            double v[TMAX];
#pragma unroll
            for (int t = 0; t < TMAX; ++t) v[t] = part16[(size_t)(t < T ? t : T - 1) * G + g];
#pragma unroll
            for (int t = 0; t < TMAX; ++t) acc[t] += (t < T) ? v[t] : 0.0;
        }
        // (kmpp_fused_kernel: the caller's 16 point values are requested HERE, behind the loads the chain waits for --
        //  the counter of outstanding loads retires in order, so requested first they would hold the chain up)
        if (pre_pts) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                pre_pts[q] = xs[pre_i0 + q]; pre_pts[4 + q] = xs[n_pad + pre_i0 + q];
                pre_pts[8 + q] = xs[2 * n_pad + pre_i0 + q]; pre_pts[12 + q] = closest[pre_i0 + q];
            }
        }
        // (a wave none of whose threads owns a group -- 12 of the 16 at N = 1M -- has nothing to add up: eight fp64
        //  wave reductions by every wave of the workgroup were a microsecond of this chain)
        const bool wave_has_groups = wave * 64 * seg < G;
#pragma unroll
        for (int t = 0; t < TMAX; ++t) {
            if (t < T) {
                const double w = wave_has_groups ? wave_sum_f64(acc[t]) : 0.0;
                if (lane == 0) wsum[t][wave] = w;
            }
        }
        km_stamp(dbg, 1, t_start);
        __syncthreads();
        if (tid < 64) {
            // lanes 0..T-1 add up their candidate's 16 wave totals (fixed order); first minimum wins (np.argmin)
            double p = 0.0;
            if (tid < T)
                for (int w = 0; w < nwg; ++w) p += wsum[tid][w];
            int best = 0;
            double best_pot = readlane_f64(p, 0);
#pragma unroll
            for (int t = 1; t < TMAX; ++t) {
                const double pt = readlane_f64(p, t);
                if (t < T && pt < best_pot) { best = t; best_pot = pt; }
            }
            if (tid == 0) best_sh = best;
        }
        __syncthreads();
        km_stamp(dbg, 2, t_start);
        const int best = best_sh;
#pragma unroll
        for (int t = 0; t < TMAX; ++t) if (t == best) loc = acc[t];
    } else if (draw) {
        for (int g = g_lo; g < g_hi; ++g) loc += g0[g];
    }
    int64_t win = 0;
    pcx = 0.0; pcy = 0.0; pcz = 0.0;
    const int best = select ? best_sh : 0;
    if (select) {
        const int bl = __builtin_amdgcn_readfirstlane(best);
        win = (int64_t)__double_as_longlong(readlane_f64(__longlong_as_double((long long)my_cand), bl));
        pcx = readlane_f64(my_cx, bl); pcy = readlane_f64(my_cy, bl); pcz = readlane_f64(my_cz, bl);
        if (tid == 0 && writer) {
            ids[j] = win;
            centres[3 * j + 0] = pcx; centres[3 * j + 1] = pcy; centres[3 * j + 2] = pcz;
        }
    }
    if (!draw) return;
    const double* gs = select ? part16 + (size_t)best * G : g0;      // group sums of the winner
    const double* bs = select ? part + (size_t)best * B : bsum0;     // its block sums
    const int pend = select ? j : -1;                       // centre whose fold `closest` is still missing
    const double incl = (wave * 64 * seg < G) ? wave_scan_f64(loc) : 0.0;     // (loc = 0 in a wave without groups)
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();                                        // (also: everybody has read cand[best] before it is overwritten)
    double off = 0.0;
    for (int w = 0; w < min(wave, nwg); ++w) off += wave_tot[w];
    if (seg == 1) {
        if (g_lo < g_hi) pre_sh[g_lo] = off + incl;
    } else {
        double run = off + (incl - loc);
        for (int g = g_lo; g < g_hi; ++g) {
            run += gs[g];                                   // a re-read of lines this CU fetched a moment ago
            if (in_lds) pre_sh[g] = run; else gprefix[g] = run;
        }
    }
    seg_end[tid] = off + incl;
    if (tid == 1023) total_sh = off + incl;
    if (!in_lds) __threadfence();
    __syncthreads();
    const double pot = total_sh;
    km_stamp(dbg, 3, t_start);
    if (wave >= T) return;                               // (no barrier of this body follows)
    const double v = u_rand * pot;
    const volatile double* gp = gprefix;                 // written by other waves of this workgroup
    auto pre = [&](int g) -> double { return in_lds ? pre_sh[g] : gp[g]; };
    // first thread segment whose end >= v = number of segment ends below v (the ends never decrease): the wave
    // counts them 64 at a time instead of walking a dependent binary search through LDS
    int lo = 0;
    const int nseg = min(1024, (G + seg - 1) / seg);     // segments that hold groups; the rest repeat the total
    for (int sbase = 0; sbase < nseg; sbase += 64)
        lo += __popcll(__ballot(sbase + lane < nseg && seg_end[sbase + lane] < v));
    int hi = min(G, (lo + 1) * seg);                     // ... then the first group inside it
    lo = min(G, lo * seg);
    while (lo < hi && !(pre(lo) >= v)) ++lo;
    if (lo == hi && hi < G) lo = hi;                     // rounding at the segment's end
    km_stamp(dbg, 4, t_start);
    int64_t found = n - 1;                               // np.clip(..., n - 1)
    bool have_xyz = false;                               // (wave-uniform) the drawn point's coordinates are in registers
    double fdx = 0.0, fdy = 0.0, fdz = 0.0;
    if (lo < G) {
        // the block inside group `lo`: running sum of its 16 block sums in the order the step kernel added them
        const double gbase = lo > 0 ? pre(lo - 1) : 0.0;
        const int bfirst = lo * KM_GROUP;
        const double mine = (lane < KM_GROUP && bfirst + lane < B) ? bs[bfirst + lane] : 0.0;
        double run = 0.0, before = 0.0;
        int blk = -1;
#pragma unroll
        for (int q = 0; q < KM_GROUP; ++q) {
            before = run;
            run += readlane_f64(mine, q);
            if (blk < 0 && bfirst + q < B && gbase + run >= v) { blk = bfirst + q; break; }
        }
        km_stamp(dbg, 5, t_start);
        if (blk < 0) {                                   // rounding at the group's end: first element after it
            const int64_t nxt = (int64_t)min(B, bfirst + KM_GROUP) * KM_BLOCK;
            found = nxt < n ? nxt : n - 1;
        } else {
            const double base = gbase + before;
            const int64_t e0 = (int64_t)blk * KM_BLOCK + 4 * lane;
            // the block's 4 x 4 values per lane in ONE round of loads (the padding up to n_pad exists in both arrays and
            // is masked below; written with a test per element this was four dependent trips), and the coordinates
            // double as the drawn point's: no further trip for them
            double c[4], cv[4], xv[4], yv[4], zv[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                cv[q] = closest[e0 + q]; xv[q] = xs[e0 + q]; yv[q] = xs[n_pad + e0 + q]; zv[q] = xs[2 * n_pad + e0 + q];
            }
            double s = 0.0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                double val = cv[q];
                if (pend >= 0) val = fmin(val, dist2(xv[q], yv[q], zv[q], pcx, pcy, pcz));
                if (!(e0 + q < n)) val = 0.0;
                s += val;
                c[q] = s;
            }
            const double ex = base + (wave_scan_f64(s) - s);
            int first_q = 4;
#pragma unroll
            for (int q = 3; q >= 0; --q)
                if (e0 + q < n && ex + c[q] >= v) first_q = q;
            const unsigned long long hit = __ballot(first_q < 4);
            if (hit) {
                const int l = __ffsll((long long)hit) - 1;
                const int q = __builtin_amdgcn_readlane(first_q, l);
                found = (int64_t)blk * KM_BLOCK + 4 * l + q;
                const double sx = q == 0 ? xv[0] : q == 1 ? xv[1] : q == 2 ? xv[2] : xv[3];
                const double sy = q == 0 ? yv[0] : q == 1 ? yv[1] : q == 2 ? yv[2] : yv[3];
                const double sz = q == 0 ? zv[0] : q == 1 ? zv[1] : q == 2 ? zv[2] : zv[3];
                fdx = readlane_f64(sx, l); fdy = readlane_f64(sy, l); fdz = readlane_f64(sz, l);
                have_xyz = true;
            } else {
                // rounding at the block's end: the threshold falls on the first element after it
                const int64_t nxt = (int64_t)(blk + 1) * KM_BLOCK;
                found = nxt < n ? nxt : n - 1;
            }
        }
    }
    km_stamp(dbg, 6, t_start);
    if (!have_xyz) { fdx = xs[found]; fdy = xs[n_pad + found]; fdz = xs[2 * n_pad + found]; }   // (the rounding cases)
    if (lane < 3) {                                      // for the next step + tail
        const double v3 = lane == 0 ? fdx : lane == 1 ? fdy : fdz;
        if (writer) cand_xyz_out[3 * wave + lane] = v3;
        if (sh_cand) sh_cand[3 * wave + lane] = v3;
    }
    if (lane == 0 && writer) cand_out[wave] = found;
}

template <int TMAX>
__global__ __launch_bounds__(1024) void kmpp_tail_kernel(const double* __restrict__ xs, int64_t n, int64_t n_pad,
                                                         const double* __restrict__ closest,
                                                         const double* __restrict__ part,
                                                         const double* __restrict__ part16, int G,
                                                         const double* __restrict__ bsum0,
                                                         const double* __restrict__ g0, int B, int T, int j,
                                                         int select, int draw, const double* __restrict__ rand_c,
                                                         const int64_t* cand_in, const double* cand_xyz_in,
                                                         int64_t* cand_out, double* cand_xyz_out,
                                                         double* __restrict__ centres,
                                                         int64_t* __restrict__ ids, double* __restrict__ gprefix) {
    __shared__ KmTailShared ts;
    double pcx, pcy, pcz;
    kmpp_tail_body<TMAX>(xs, n, n_pad, closest, part, part16, G, bsum0, g0, B, T, j, select, draw, rand_c, cand_in,
                         cand_xyz_in, cand_out, cand_xyz_out, centres, ids, gprefix, true, nullptr, pcx, pcy, pcz, ts);
}

// ONE launch per centre: every workgroup of the pass over the points first runs the tail of the PREVIOUS centre for
// itself -- accept centre j - 1 among its candidates (potentials in buffer (j - 1) & 1), draw the candidates of centre
// j -- and then its share of the pass: fold centre j - 1 into `closest`, potentials of the new candidates into buffer
// j & 1.  The tail is a chain of ~5 dependent trips to memory (~6 us); run by one workgroup in a launch of its own it
// was 8.5 of the 16.4 us per centre.  Here the workgroup's points are requested BEFORE the chain starts, so the pass's
// 32 MB stream rides under it, and the launch boundary between tail and pass is gone.  All workgroups compute the
// same draws (same code on the same numbers: the only memory they read that others write meanwhile is `closest`,
// where a value is either before or after the fold of centre j - 1 -- and the draw applies that fold on the fly,
// min(min(c, d), d) = min(c, d)); workgroup 0 records them.  Needs the group-prefix table in LDS (G <= 4096 groups
// = 16.7 M points); larger clouds take the two-launch form.
template <int TMAX>
__global__ __launch_bounds__(KM_STEP_BLOCK) void kmpp_fused_kernel(const double* __restrict__ xs, int64_t n, int64_t n_pad,
                                                                   double* closest,
                                                                   const double* __restrict__ part_prev,
                                                                   const double* __restrict__ part16_prev,
                                                                   double* __restrict__ part_next,
                                                                   double* __restrict__ part16_next, int G, int B, int T,
                                                                   int j, const double* __restrict__ rand_c,
                                                                   const int64_t* cand_prev, const double* cand_xyz_prev,
                                                                   int64_t* cand_next, double* cand_xyz_next,
                                                                   double* __restrict__ centres,
                                                                   int64_t* __restrict__ ids,
                                                                   unsigned long long* dbg) {
    const unsigned long long t_start = dbg ? (unsigned long long)wall_clock64() : 0ull;
    static_assert(KM_STEP_BLOCK == 1024, "the tail body is written for 1024 threads");
    __shared__ double shg[KM_MAX_TRIALS][KM_GROUP];
    __shared__ double sh_cand[3 * KM_MAX_TRIALS];
    __shared__ KmTailShared ts;
    const int lane = lane_id();
    const int64_t blk = (int64_t)blockIdx.x * KM_GROUP + wave_in_block();     // 256-point block of this wave
    const bool have = blk < B;
    const int64_t i0 = (have ? blk : 0) * KM_BLOCK + 4 * lane;                 // n_pad is a multiple of 256
    // the pass's operands do not depend on the tail: requested inside its first round of loads (see there)
    double pts[16];
    double fx, fy, fz;
    kmpp_tail_body<TMAX>(xs, n, n_pad, closest, part_prev, part16_prev, G, nullptr, nullptr, B, T, j - 1, 1, 1, rand_c,
                         cand_prev, cand_xyz_prev, cand_next, cand_xyz_next, centres, ids, nullptr, blockIdx.x == 0,
                         sh_cand, fx, fy, fz, ts, dbg, t_start, pts, i0);
    __syncthreads();                                         // sh_cand
    km_stamp(dbg, 7, t_start);
    double x[4], y[4], z[4], cl[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { x[q] = pts[q]; y[q] = pts[4 + q]; z[q] = pts[8 + q]; cl[q] = pts[12 + q]; }
    if (have) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const double d = dist2(x[q], y[q], z[q], fx, fy, fz);
            if (i0 + q < n && d < cl[q]) { cl[q] = d; closest[i0 + q] = d; }
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) if (!have || !(i0 + q < n)) cl[q] = 0.0;      // padding contributes min(0, d) = 0
#pragma unroll
    for (int t = 0; t < TMAX; ++t) {
        if (t < T) {
            const double cx = sh_cand[3 * t], cy = sh_cand[3 * t + 1], cz = sh_cand[3 * t + 2];
            double sacc = 0.0;
#pragma unroll
            for (int q = 0; q < 4; ++q) sacc += fmin(cl[q], dist2(x[q], y[q], z[q], cx, cy, cz));
            const double w = wave_sum_f64(sacc);
            if (lane == 0) {
                if (have) part_next[(size_t)t * B + blk] = w;
                shg[t][wave_in_block()] = have ? w : 0.0;
            }
        }
    }
    km_stamp(dbg, 9, t_start);
    __syncthreads();
    if ((int)threadIdx.x < T) {
        double s2 = 0.0;
#pragma unroll
        for (int q = 0; q < KM_GROUP; ++q) s2 += shg[threadIdx.x][q];
        part16_next[(size_t)threadIdx.x * gridDim.x + blockIdx.x] = s2;
    }
    km_stamp(dbg, 8, t_start);
}

// ------------------------------------------------------------------------------------------
// Lloyd iteration
// ------------------------------------------------------------------------------------------
// Two points per thread (512 per workgroup); the centre table [k][4] (x y z pad) is read with a
// wave-uniform index, i.e. through the scalar cache.  ~10 fp64 lane-ops per (point, centre).
__global__ __launch_bounds__(KM_BLOCK) void kmeans_assign_kernel(
    const double* __restrict__ xs, int64_t n, int64_t n_pad, const double* __restrict__ c4, int k,
    int32_t* __restrict__ labels, double* __restrict__ mind2, double* __restrict__ inertia_part,
    unsigned long long* __restrict__ n_changed, const int* __restrict__ done) {
    if (done && *done) return;
    __shared__ double sh[4];
    __shared__ int shc[4];
    const int64_t i0 = (int64_t)blockIdx.x * (2 * KM_BLOCK) + threadIdx.x;
    const int64_t i1 = i0 + KM_BLOCK;
    const bool l0 = i0 < n, l1 = i1 < n;
    const int64_t j0 = l0 ? i0 : 0, j1 = l1 ? i1 : 0;
    const double x0 = xs[j0], y0 = xs[n_pad + j0], z0 = xs[2 * n_pad + j0];
    const double x1 = xs[j1], y1 = xs[n_pad + j1], z1 = xs[2 * n_pad + j1];
    // Four centres at a time: their minimum (3 v_min_f64) is compared with the running best once, and only the
    // group's first index is recorded (1 compare + 3 selects per FOUR pairs instead of per pair: 7.75 instead of 10
    // lane-operations per pair); which of the four it was is settled afterwards by evaluating the winning group
    // again -- same arithmetic, so the same values, and the first index that attains the minimum wins as before.
    double b0 = INFINITY, b1 = INFINITY;
    int a0 = 0, a1 = 0;
    const int k4 = k & ~3;
    for (int j = 0; j < k4; j += 4) {
        double e0[4], e1[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const double cx = c4[4 * (j + u)], cy = c4[4 * (j + u) + 1], cz = c4[4 * (j + u) + 2];
            e0[u] = dist2(x0, y0, z0, cx, cy, cz);
            e1[u] = dist2(x1, y1, z1, cx, cy, cz);
        }
        const double m0 = fmin(fmin(e0[0], e0[1]), fmin(e0[2], e0[3]));
        const double m1 = fmin(fmin(e1[0], e1[1]), fmin(e1[2], e1[3]));
        if (m0 < b0) { b0 = m0; a0 = j; }
        if (m1 < b1) { b1 = m1; a1 = j; }
    }
    if (k4 > 0) {
        // the member of the winning group (divergent group index: vector loads, once per point)
        int w0 = a0, w1 = a1;
#pragma unroll
        for (int u = 3; u >= 0; --u) {
            const double* p0 = c4 + 4 * (a0 + u);
            const double* p1 = c4 + 4 * (a1 + u);
            if (dist2(x0, y0, z0, p0[0], p0[1], p0[2]) == b0) w0 = a0 + u;
            if (dist2(x1, y1, z1, p1[0], p1[1], p1[2]) == b1) w1 = a1 + u;
        }
        a0 = w0; a1 = w1;
    }
    for (int j = k4; j < k; ++j) {
        const double cx = c4[4 * j], cy = c4[4 * j + 1], cz = c4[4 * j + 2];
        const double d0 = dist2(x0, y0, z0, cx, cy, cz);
        const double d1 = dist2(x1, y1, z1, cx, cy, cz);
        if (d0 < b0) { b0 = d0; a0 = j; }
        if (d1 < b1) { b1 = d1; a1 = j; }
    }
    int changed = 0;
    double in = 0.0;
    if (l0) {
        changed += labels[i0] != a0;
        labels[i0] = a0;
        mind2[i0] = b0;
        in += b0;
    }
    if (l1) {
        changed += labels[i1] != a1;
        labels[i1] = a1;
        mind2[i1] = b1;
        in += b1;
    }
    const int wc = wave_reduce_i(changed, OpAddI());
    if (lane_id() == 0) shc[wave_in_block()] = wc;
    const double t = block_sum_256(in, sh);
    if (threadIdx.x == 0) {
        inertia_part[blockIdx.x] = t;
        const int tc = shc[0] + shc[1] + shc[2] + shc[3];
        if (tc) atomicAdd(n_changed, (unsigned long long)tc);
    }
}

// One wave owns a contiguous run of points and NSLOT * 64 centres starting at slot0 * 64; lane l
// holds the sums of centres slot * 64 + l.  The 64 points of a step are loaded coalesced and then
// visited one by one (label and coordinates broadcast with v_readlane): the label is wave-uniform,
// so 'which register' is a scalar branch and 'which lane' an exec mask -- no atomics.
template <int NSLOT>
__global__ __launch_bounds__(KM_BLOCK) void kmeans_accum_kernel(const double* __restrict__ xs, int64_t n,
                                                                int64_t n_pad,
                                                                const int32_t* __restrict__ labels,
                                                                int slot0, int k_alloc,
                                                                double* __restrict__ partial,
                                                                const int* __restrict__ done) {
    if (done && *done) return;
    __shared__ double sh[4][64][4];
    const int lane = lane_id(), wave = wave_in_block();
    double ax[NSLOT], ay[NSLOT], az[NSLOT], ac[NSLOT];
#pragma unroll
    for (int s = 0; s < NSLOT; ++s) ax[s] = ay[s] = az[s] = ac[s] = 0.0;
    const int64_t waves = (int64_t)gridDim.x * 4;
    const int64_t per = ((n + waves - 1) / waves + 63) / 64 * 64;
    const int64_t gw = (int64_t)blockIdx.x * 4 + wave;
    const int64_t begin = gw * per;
    const int64_t end = begin + per < n ? begin + per : n;
    for (int64_t base = begin; base < end; base += 64) {
        const int64_t i = base + lane;
        const bool ok = i < end;
        const int64_t ii = ok ? i : 0;
        const int rel = ok ? labels[ii] - slot0 * 64 : -1;
        const double x = xs[ii], y = xs[n_pad + ii], z = xs[2 * n_pad + ii];
        const int cnt = (int)(end - base < 64 ? end - base : 64);
        for (int t = 0; t < cnt; ++t) {
            const int r = __builtin_amdgcn_readlane(rel, t);
            if (r < 0 || r >= NSLOT * 64) continue;
            const double px = readlane_f64(x, t), py = readlane_f64(y, t), pz = readlane_f64(z, t);
            const int s = r >> 6;
            if (lane == (r & 63)) {
#pragma unroll
                for (int u = 0; u < NSLOT; ++u)
                    if (u == s) { ax[u] += px; ay[u] += py; az[u] += pz; ac[u] += 1.0; }
            }
        }
    }
    double* out = partial + (size_t)blockIdx.x * k_alloc * 4;
#pragma unroll
    for (int s = 0; s < NSLOT; ++s) {
        sh[wave][lane][0] = ax[s];
        sh[wave][lane][1] = ay[s];
        sh[wave][lane][2] = az[s];
        sh[wave][lane][3] = ac[s];
        __syncthreads();
        const int l = threadIdx.x >> 2, f = threadIdx.x & 3;
        out[((size_t)(slot0 + s) * 64 + l) * 4 + f] = (sh[0][l][f] + sh[1][l][f]) + (sh[2][l][f] + sh[3][l][f]);
        __syncthreads();
    }
}

// k <= 1024: the sums live in LDS instead of registers.  Every wave owns a private table [k][4] (x, y, z, count)
// and walks its contiguous run of points; a point is ONE ds_add_f64 by four lanes (lane f adds feature f) at the
// address its label selects -- no 16-way predicated register update, no exec-mask juggling: ~5 instructions per
// point instead of ~80.  The table is private to the wave and LDS executes a wave's operations in order, so the
// additions happen in point order: deterministic, like the register form.  The four tables of a workgroup are added
// in fixed order on the way out.  One workgroup per CU (4 x 32 KB at k = 1024).
__global__ __launch_bounds__(KM_BLOCK) void kmeans_accum_lds_kernel(const double* __restrict__ xs, int64_t n,
                                                                    int64_t n_pad,
                                                                    const int32_t* __restrict__ labels, int k_alloc,
                                                                    double* __restrict__ partial,
                                                                    const int* __restrict__ done) {
    if (done && *done) return;
    extern __shared__ double km_lds[];
    const int lane = lane_id(), wave = wave_in_block();
    const int tsz = k_alloc * 4;
    double* T = km_lds + (size_t)wave * tsz;              // this wave's sums
    double* S = km_lds + (size_t)4 * tsz + wave * 256;    // [64 points][4] staging of the current batch
    for (int e = lane; e < tsz; e += 64) T[e] = 0.0;
    // Wave-level ordering contract, stated instead of assumed: lanes 0..3 read what OTHER lanes of this wave stored
    // (the staging tile, the cleared table).  The hardware executes one wave's LDS operations in order, but the
    // compiler may reorder accesses it cannot prove to alias; a wavefront-scope release / acquire fence pair plus the
    // (free) wave barrier pins the order in the generated code.
    auto wave_lds_sync = [] {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    wave_lds_sync();                                      // table cleared before the first ds_add
    const int64_t waves = (int64_t)gridDim.x * 4;
    const int64_t per = ((n + waves - 1) / waves + 63) / 64 * 64;
    const int64_t gw = (int64_t)blockIdx.x * 4 + wave;
    const int64_t begin = gw * per;
    const int64_t end = begin + per < n ? begin + per : n;
    const int f = lane & 3;
    for (int64_t base = begin; base < end; base += 64) {
        const int64_t i = base + lane;
        const bool ok = i < end;
        const int64_t ii = ok ? i : 0;
        const int lab = ok ? labels[ii] : 0;
        const double x = xs[ii], y = xs[n_pad + ii], z = xs[2 * n_pad + ii];
        wave_lds_sync();                                  // the previous batch has been consumed
        S[lane * 4 + 0] = x; S[lane * 4 + 1] = y; S[lane * 4 + 2] = z; S[lane * 4 + 3] = 1.0;
        wave_lds_sync();                                  // all 64 lanes' stores before lanes 0..3 read them
        const int cnt = (int)(end - base < 64 ? end - base : 64);
        if (lane < 4) {
            int t = 0;
            for (; t + 8 <= cnt; t += 8) {
                double v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = S[(t + u) * 4 + f];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int r = __builtin_amdgcn_readlane(lab, t + u);
                    (void)__hip_atomic_fetch_add(&T[r * 4 + f], v[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
            for (; t < cnt; ++t) {
                const int r = __builtin_amdgcn_readlane(lab, t);
                (void)__hip_atomic_fetch_add(&T[r * 4 + f], S[t * 4 + f], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
    }
    __syncthreads();
    double* out = partial + (size_t)blockIdx.x * tsz;
    for (int e = threadIdx.x; e < tsz; e += KM_BLOCK)
        out[e] = (km_lds[e] + km_lds[tsz + e]) + (km_lds[2 * tsz + e] + km_lds[3 * tsz + e]);
}

// out[e] = sum over workgroups of partial[b][e], e < 4 k: 32 entries x 8 block slices per
// workgroup, fixed order.  Workgroup 0 also adds up the inertia partials and appends
// (inertia, n_changed) behind the sums.
__global__ __launch_bounds__(KM_BLOCK) void kmeans_reduce_kernel(const double* __restrict__ partial, int nb,
                                                                 int k_alloc, int k,
                                                                 const double* __restrict__ inertia_part,
                                                                 int nib,
                                                                 const unsigned long long* __restrict__ n_changed,
                                                                 double* __restrict__ out,
                                                                 const int* __restrict__ done) {
    if (done && *done) return;
    __shared__ double sh[8][32];
    __shared__ double shi[4];
    const int e_loc = threadIdx.x & 31, slice = threadIdx.x >> 5;
    const int e = blockIdx.x * 32 + e_loc;
    double s = 0.0;
    if (e < 4 * k) {
        const int per = (nb + 7) / 8;
        const int b0 = slice * per, b1 = min(nb, b0 + per);
        for (int b = b0; b < b1; ++b) s += partial[(size_t)b * k_alloc * 4 + e];
    }
    sh[slice][e_loc] = s;
    __syncthreads();
    if (slice == 0 && e < 4 * k) {
        double t = 0.0;
#pragma unroll
        for (int q = 0; q < 8; ++q) t += sh[q][e_loc];
        out[e] = t;
    }
    if (blockIdx.x == 0) {
        double v = 0.0;
        for (int b = threadIdx.x; b < nib; b += KM_BLOCK) v += inertia_part[b];
        const double t = block_sum_256(v, shi);
        if (threadIdx.x == 0) {
            out[4 * k] = t;
            out[4 * k + 1] = (double)n_changed[0];
        }
    }
}

__global__ void kmeans_pad_centres_kernel(const double* __restrict__ c3, int k, double* __restrict__ c4,
                                          const int* __restrict__ done) {
    if (done && *done) return;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= k) return;
    c4[4 * j] = c3[3 * j];
    c4[4 * j + 1] = c3[3 * j + 1];
    c4[4 * j + 2] = c3[3 * j + 2];
    c4[4 * j + 3] = 0.0;
}

__global__ void kmeans_reset_labels_kernel(int32_t* __restrict__ labels, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) labels[i] = -1;
}

// Device-side tail of one Lloyd iteration (scikit-learn's _kmeans_single_lloyd, lines after lloyd_iter):
// centres = sums * (1 / count), summed squared centre shift, then the stop rules in scikit-learn's order --
// unchanged labels (strict), shift <= tol, iteration budget.  An empty cluster hands the iteration to the host
// (relocation needs argpartition over all points): needs_host is raised and nothing is updated.
struct KmCtl { int done, it, strict, needs_host; unsigned long long pad; };
__global__ __launch_bounds__(256) void kmeans_update_kernel(const double* __restrict__ out, int k, double tol_abs,
                                                            int max_iter, double* __restrict__ c3,
                                                            unsigned long long* __restrict__ changed,
                                                            KmCtl* __restrict__ ctl) {
    if (ctl->done) return;
    __shared__ double sh[4];
    __shared__ int sh_empty;
    if (threadIdx.x == 0) sh_empty = 0;
    __syncthreads();
    int empty = 0;
    for (int j = threadIdx.x; j < k; j += 256) empty |= out[4 * j + 3] == 0.0;
    if (empty) sh_empty = 1;
    __syncthreads();
    if (sh_empty) {
        if (threadIdx.x == 0) { ctl->needs_host = 1; ctl->done = 1; }
        return;
    }
    double shift = 0.0;
    for (int j = threadIdx.x; j < k; j += 256) {
        const double alpha = 1.0 / out[4 * j + 3];
        const double n0 = out[4 * j] * alpha, n1 = out[4 * j + 1] * alpha, n2 = out[4 * j + 2] * alpha;
        const double d0 = n0 - c3[3 * j], d1 = n1 - c3[3 * j + 1], d2 = n2 - c3[3 * j + 2];
        const double nrm = sqrt((d0 * d0 + d1 * d1) + d2 * d2);
        shift += nrm * nrm;
        c3[3 * j] = n0; c3[3 * j + 1] = n1; c3[3 * j + 2] = n2;
    }
    const double tot = block_sum_256(shift, sh);
    if (threadIdx.x == 0) {
        const int it = ctl->it + 1;
        ctl->it = it;
        if (out[4 * k + 1] == 0.0) { ctl->strict = 1; ctl->done = 1; }
        else if (tot <= tol_abs || it >= max_iter) ctl->done = 1;
        *changed = 0ull;                                   // next iteration's counter
    }
}

static int km_check(hgmm_ctx* c, int k) {
    if (!c->have_f64 || c->n <= 0) return fail(c, HGMM_ERR_STATE, "kmeans: set points first");
    if (k < 1 || k > KM_MAX_K) return fail(c, HGMM_ERR_ARG, "kmeans: k = %d outside 1..%d", k, KM_MAX_K);
    if (c->n > 0x7fffffff - 1024) return fail(c, HGMM_ERR_ARG, "too many points for 32-bit indices");
    return HGMM_OK;
}

}  // namespace hgmm

using namespace hgmm;

// ------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------
// Host helper: column means, variances and the centred copy of an [n,3] cloud in the summation order NumPy uses for
// X.mean(axis=0) / np.var(X, axis=0) on a C-ordered array (a running sum per column, row by row; product and sum
// rounded separately) -- scikit-learn centres by the one and scales tol by the other, and the initialiser has to
// work on the very same doubles.  Three independent add chains per pass: ~1 ns per point instead of NumPy's 25.
#pragma clang fp contract(off)
extern "C" int hgmm_kmeans_center_f64(const double* x, int64_t n, double* mean3, double* var3, double* xc_out) {
    if (!x || n < 1 || !mean3 || !var3 || !xc_out) return HGMM_ERR_ARG;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    for (int64_t i = 0; i < n; ++i) { s0 += x[3 * i]; s1 += x[3 * i + 1]; s2 += x[3 * i + 2]; }
    const double dn = (double)n;
    const double m0 = s0 / dn, m1 = s1 / dn, m2 = s2 / dn;
    double v0 = 0.0, v1 = 0.0, v2 = 0.0;
    for (int64_t i = 0; i < n; ++i) {
        const double d0 = x[3 * i] - m0, d1 = x[3 * i + 1] - m1, d2 = x[3 * i + 2] - m2;
        xc_out[3 * i] = d0; xc_out[3 * i + 1] = d1; xc_out[3 * i + 2] = d2;
        const double q0 = d0 * d0, q1 = d1 * d1, q2 = d2 * d2;
        v0 += q0; v1 += q1; v2 += q2;
    }
    mean3[0] = m0; mean3[1] = m1; mean3[2] = m2;
    var3[0] = v0 / dn; var3[1] = v1 / dn; var3[2] = v2 / dn;
    return HGMM_OK;
}
#pragma clang fp contract(fast)

extern "C" int hgmm_kmeans_plusplus(hgmm_ctx* c, int k, int64_t first_id, const double* rand_vals,
                                    int n_trials, int64_t* ids_out, double* centers_out) {
    HGMM_ENTER(c);
    HGMM_TRY(km_check(c, k));
    if (k > c->n) return fail(c, HGMM_ERR_ARG, "kmeans++: k = %d exceeds the %lld points", k, (long long)c->n);
    if (n_trials < 1 || n_trials > KM_MAX_TRIALS)
        return fail(c, HGMM_ERR_ARG, "kmeans++: n_trials = %d outside 1..%d", n_trials, KM_MAX_TRIALS);
    if (first_id < 0 || first_id >= c->n) return fail(c, HGMM_ERR_ARG, "kmeans++: first_id out of range");
    if (k > 1 && !rand_vals) return fail(c, HGMM_ERR_ARG, "kmeans++: rand_vals is NULL");
    HGMM_HIP(c, hipSetDevice(c->device));
    const int64_t n = c->n, n_pad = c->n_pad;
    const int B = (int)km_nblk(n, KM_BLOCK);
    HGMM_TRY(ensure(c, c->km_closest, sizeof(double) * n_pad));
    const int G = (B + KM_GROUP - 1) / KM_GROUP;
    // (potentials and candidates are double-buffered by centre parity: kmpp_fused_kernel)
    HGMM_TRY(ensure(c, c->km_block, sizeof(double) * ((size_t)2 * B + 2 * (size_t)(B + G) * KM_MAX_TRIALS + 2 * (size_t)G + 8)));
    HGMM_TRY(ensure(c, c->km_centres, sizeof(double) * (7 * (size_t)k + 2 * 3 * KM_MAX_TRIALS)));
    HGMM_TRY(ensure(c, c->km_ids, sizeof(int64_t) * ((size_t)k + 2 * KM_MAX_TRIALS)));
    HGMM_TRY(ensure(c, c->km_rand, sizeof(double) * (size_t)std::max(1, (k - 1) * n_trials)));
    const double* xs = c->x_soa64.as<double>();
    double* closest = c->km_closest.as<double>();
    double* bsum = c->km_block.as<double>();
    double* bprefix = bsum + B;
    double* part = bprefix + B;                               // [2][KM_MAX_TRIALS][B]
    double* pot = part + 2 * (size_t)B * KM_MAX_TRIALS;
    double* part16 = pot + 8;                                 // [2][KM_MAX_TRIALS][G]
    double* g0 = part16 + 2 * (size_t)G * KM_MAX_TRIALS;
    double* gprefix = g0 + G;
    double* centres = c->km_centres.as<double>();
    int64_t* ids = c->km_ids.as<int64_t>();
    int64_t* cand = ids + k;
    double* rand_dev = c->km_rand.as<double>();
    if (k > 1)
        HGMM_HIP(c, hipMemcpyAsync(rand_dev, rand_vals, sizeof(double) * (size_t)(k - 1) * n_trials,
                                   hipMemcpyHostToDevice, c->stream));
    kmpp_first_kernel<<<B, KM_BLOCK, 0, c->stream>>>(xs, n, n_pad, first_id, closest, bsum, centres, ids);
    if (k > 1) {
        // The candidates of centre j, their coordinates and their potentials live in buffer j & 1.
        double* cand_xyz = centres + 7 * (size_t)k;             // [2][T][3], behind the centre tables
        const bool t8 = n_trials <= 8;
        auto cand_b = [&](int j) { return cand + (size_t)(j & 1) * KM_MAX_TRIALS; };
        auto xyz_b = [&](int j) { return cand_xyz + (size_t)(j & 1) * 3 * KM_MAX_TRIALS; };
        auto part_b = [&](int j) { return part + (size_t)(j & 1) * B * KM_MAX_TRIALS; };
        auto part16_b = [&](int j) { return part16 + (size_t)(j & 1) * G * KM_MAX_TRIALS; };
        // tail(jj): accept centre jj among its candidates (select) and / or draw the candidates of centre jj + 1 (draw)
        auto tail = [&](int jj, int select, int draw, const double* rnd) {
            if (t8)
                kmpp_tail_kernel<8><<<1, 1024, 0, c->stream>>>(xs, n, n_pad, closest, part_b(jj), part16_b(jj), G, bsum, g0, B,
                                                               n_trials, jj, select, draw, rnd, cand_b(jj), xyz_b(jj),
                                                               cand_b(jj + 1), xyz_b(jj + 1), centres, ids, gprefix);
            else
                kmpp_tail_kernel<KM_MAX_TRIALS><<<1, 1024, 0, c->stream>>>(xs, n, n_pad, closest, part_b(jj), part16_b(jj), G, bsum,
                                                                           g0, B, n_trials, jj, select, draw, rnd, cand_b(jj),
                                                                           xyz_b(jj), cand_b(jj + 1), xyz_b(jj + 1), centres,
                                                                           ids, gprefix);
        };
        auto step = [&](int j, int fold) {
            if (t8)
                kmpp_step_kernel<8><<<G, KM_STEP_BLOCK, 0, c->stream>>>(xs, n, n_pad, closest, centres, fold, xyz_b(j), n_trials,
                                                                        part_b(j), B, part16_b(j));
            else
                kmpp_step_kernel<KM_MAX_TRIALS><<<G, KM_STEP_BLOCK, 0, c->stream>>>(xs, n, n_pad, closest, centres, fold, xyz_b(j),
                                                                                    n_trials, part_b(j), B, part16_b(j));
        };
        kmpp_group_kernel<<<km_nblk(G, 256), 256, 0, c->stream>>>(bsum, B, G, g0);
        tail(0, 0, 1, rand_dev);
        if (G <= KM_TAIL_LDS_GROUPS && !c->cfg[CFG_KMPP_TWO_LAUNCHES]) {
            // one launch per centre (kmpp_fused_kernel): pass of centre 1, then for every further centre the tail of
            // the previous one inside the pass's own launch, and the last centre's tail on its own
            step(1, -1);
            unsigned long long* dbg = nullptr;               // (phase clocks of thread 0: a debugging aid, off)
            for (int j = 2; j < k; ++j) {
                const double* rnd = rand_dev + (size_t)(j - 1) * n_trials;
                if (t8)
                    kmpp_fused_kernel<8><<<G, KM_STEP_BLOCK, 0, c->stream>>>(
                        xs, n, n_pad, closest, part_b(j - 1), part16_b(j - 1), part_b(j), part16_b(j), G, B, n_trials, j, rnd,
                        cand_b(j - 1), xyz_b(j - 1), cand_b(j), xyz_b(j), centres, ids, dbg);
                else
                    kmpp_fused_kernel<KM_MAX_TRIALS><<<G, KM_STEP_BLOCK, 0, c->stream>>>(
                        xs, n, n_pad, closest, part_b(j - 1), part16_b(j - 1), part_b(j), part16_b(j), G, B, n_trials, j, rnd,
                        cand_b(j - 1), xyz_b(j - 1), cand_b(j), xyz_b(j), centres, ids, dbg);
            }
            tail(k - 1, 1, 0, nullptr);
        } else {
            // two launches per centre: the pass over the points (fold of the previous centre + candidate potentials),
            // then one workgroup that names the winner and draws the next centre's candidates
            for (int j = 1; j < k; ++j) {
                step(j, j >= 2 ? j - 1 : -1);
                tail(j, 1, j + 1 < k ? 1 : 0, rand_dev + (size_t)j * n_trials);
            }
        }
    }
    (void)pot;
    HGMM_HIP(c, hipGetLastError());
    StagedDownloads dl(c);
    dl.add(ids_out, ids, sizeof(int64_t) * k);
    dl.add(centers_out, centres, sizeof(double) * 3 * k);
    HGMM_HIP(c, dl.finish());
    return HGMM_OK;
}

namespace {
struct KmLaunch {
    int nslot, k_alloc, nb_assign, nb_acc;
    bool acc_lds;                                   // per-wave LDS tables (k <= 1024) instead of register slots
    double *c3, *c4, *out, *inertia_part;
    unsigned long long* changed;
    KmCtl* ctl;
};

int km_prepare(hgmm_ctx* c, int k, int reset_labels, KmLaunch& L) {
    HGMM_HIP(c, hipSetDevice(c->device));
    const int64_t n = c->n, n_pad = c->n_pad;
    L.nslot = k <= 256 ? 4 : KM_ACC_SLOTS;
    L.k_alloc = (k + 64 * L.nslot - 1) / (64 * L.nslot) * (64 * L.nslot);
    L.nb_assign = (int)km_nblk(n, 2 * KM_BLOCK);
    L.acc_lds = k <= 1024 && !c->cfg[CFG_KMEANS_ACC_REGS];
    L.nb_acc = (int)std::min<int64_t>((L.acc_lds ? 1 : 2) * (int64_t)c->cus, km_nblk(n, KM_BLOCK));
    HGMM_TRY(ensure(c, c->km_labels, sizeof(int32_t) * n_pad));
    HGMM_TRY(ensure(c, c->km_mind2, sizeof(double) * n_pad));
    HGMM_TRY(ensure(c, c->km_centres, sizeof(double) * (7 * (size_t)k + 3 * KM_MAX_TRIALS)));
    HGMM_TRY(ensure(c, c->km_partial, sizeof(double) * 4 * (size_t)L.k_alloc * L.nb_acc));
    HGMM_TRY(ensure(c, c->km_out, sizeof(double) * (4 * (size_t)k + 8) + sizeof(double) * L.nb_assign + 16));
    if (reset_labels || c->km_labels_n != n) {
        kmeans_reset_labels_kernel<<<km_nblk(n, 256), 256, 0, c->stream>>>(c->km_labels.as<int32_t>(), n);
        c->km_labels_n = n;
    }
    L.c3 = c->km_centres.as<double>();
    L.c4 = L.c3 + 3 * (size_t)k;
    L.out = c->km_out.as<double>();
    L.changed = reinterpret_cast<unsigned long long*>(L.out + 4 * (size_t)k + 2);
    L.ctl = reinterpret_cast<KmCtl*>(L.out + 4 * (size_t)k + 3);                 // 3 doubles
    L.inertia_part = L.out + 4 * (size_t)k + 6;
    return HGMM_OK;
}

// assignment + per-cluster sums + reduction with the centres in L.c3 -> L.out[4k + 2]
int km_enqueue(hgmm_ctx* c, int k, const KmLaunch& L, const int* done) {
    const int64_t n = c->n, n_pad = c->n_pad;
    const double* xs = c->x_soa64.as<double>();
    int32_t* labels = c->km_labels.as<int32_t>();
    kmeans_pad_centres_kernel<<<km_nblk(k, 256), 256, 0, c->stream>>>(L.c3, k, L.c4, done);
    {
        ProfScope prof(c, HGMM_K_KMEANS_ASSIGN);
        kmeans_assign_kernel<<<L.nb_assign, KM_BLOCK, 0, c->stream>>>(xs, n, n_pad, L.c4, k, labels,
                                                                      c->km_mind2.as<double>(), L.inertia_part,
                                                                      L.changed, done);
    }
    {
        ProfScope prof(c, HGMM_K_KMEANS_ACCUM);
        if (L.acc_lds) {
            const size_t lds = sizeof(double) * ((size_t)4 * 4 * L.k_alloc + 4 * 256);
            HGMM_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&kmeans_accum_lds_kernel),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            kmeans_accum_lds_kernel<<<L.nb_acc, KM_BLOCK, lds, c->stream>>>(xs, n, n_pad, labels, L.k_alloc,
                                                                           c->km_partial.as<double>(), done);
        } else
        for (int slot0 = 0; slot0 * 64 < k; slot0 += L.nslot) {
            if (L.nslot == 4)
                kmeans_accum_kernel<4><<<L.nb_acc, KM_BLOCK, 0, c->stream>>>(xs, n, n_pad, labels, slot0, L.k_alloc,
                                                                              c->km_partial.as<double>(), done);
            else
                kmeans_accum_kernel<KM_ACC_SLOTS><<<L.nb_acc, KM_BLOCK, 0, c->stream>>>(
                    xs, n, n_pad, labels, slot0, L.k_alloc, c->km_partial.as<double>(), done);
        }
    }
    kmeans_reduce_kernel<<<km_nblk(4 * (int64_t)k, 32), KM_BLOCK, 0, c->stream>>>(
        c->km_partial.as<double>(), L.nb_acc, L.k_alloc, k, L.inertia_part, L.nb_assign, L.changed, L.out, done);
    HGMM_HIP(c, hipGetLastError());
    return HGMM_OK;
}
}  // namespace

extern "C" int hgmm_kmeans_step(hgmm_ctx* c, int k, const double* centers, int reset_labels,
                                double* sums_out, double* inertia_out, int64_t* n_changed_out) {
    HGMM_ENTER(c);
    HGMM_TRY(km_check(c, k));
    if (!centers) return fail(c, HGMM_ERR_ARG, "kmeans: centers is NULL");
    KmLaunch L;
    HGMM_TRY(km_prepare(c, k, reset_labels, L));
    HGMM_HIP(c, hipMemcpyAsync(L.c3, centers, sizeof(double) * 3 * k, hipMemcpyHostToDevice, c->stream));
    HGMM_HIP(c, hipMemsetAsync(L.changed, 0, sizeof(unsigned long long), c->stream));
    HGMM_TRY(km_enqueue(c, k, L, nullptr));
    if (c->comm_on()) HGMM_TRY(allreduce_f64_dev(c, L.out, 4 * (size_t)k + 2));
    double tail[2] = {0.0, 0.0};
    StagedDownloads dl(c);
    dl.add(sums_out, L.out, sizeof(double) * 4 * k);
    dl.add(tail, L.out + 4 * (size_t)k, sizeof tail);
    HGMM_HIP(c, dl.finish());
    if (inertia_out) *inertia_out = tail[0];
    if (n_changed_out) *n_changed_out = (int64_t)tail[1];
    return HGMM_OK;
}

extern "C" int hgmm_kmeans_lloyd(hgmm_ctx* c, int k, double* centers_inout, int max_iter, double tol_abs,
                                 int reset_labels, int* n_iter_out, int* strict_out, int* needs_host_out,
                                 double* sums_out, int64_t* n_changed_out) {
    HGMM_ENTER(c);
    HGMM_TRY(km_check(c, k));
    if (!centers_inout) return fail(c, HGMM_ERR_ARG, "kmeans: centers is NULL");
    if (c->comm_on()) return fail(c, HGMM_ERR_STATE, "kmeans: the device-resident Lloyd loop is single-rank; "
                                                     "use hgmm_kmeans_step under a communicator");
    if (n_iter_out) *n_iter_out = 0;
    if (strict_out) *strict_out = 0;
    if (needs_host_out) *needs_host_out = 0;
    if (max_iter < 1) return HGMM_OK;
    KmLaunch L;
    HGMM_TRY(km_prepare(c, k, reset_labels, L));
    HGMM_HIP(c, hipMemcpyAsync(L.c3, centers_inout, sizeof(double) * 3 * k, hipMemcpyHostToDevice, c->stream));
    HGMM_HIP(c, hipMemsetAsync(L.changed, 0, sizeof(unsigned long long) + sizeof(KmCtl), c->stream));
    // the stop rule runs on the device; `batch` iterations are enqueued per host synchronisation and the
    // kernels of iterations past the stop return at once (same scheme as the tree build)
    const int batch = 8;
    KmCtl h = {};
    while (!h.done) {
        for (int b = 0; b < batch; ++b) {
            HGMM_TRY(km_enqueue(c, k, L, &L.ctl->done));
            kmeans_update_kernel<<<1, 256, 0, c->stream>>>(L.out, k, tol_abs, max_iter, L.c3, L.changed, L.ctl);
        }
        HGMM_HIP(c, hipGetLastError());
        HGMM_HIP(c, hipMemcpyAsync(&h, L.ctl, sizeof h, hipMemcpyDeviceToHost, c->stream));
        HGMM_HIP(c, ctx_stream_sync(c));
    }
    {
        double tail[2] = {0.0, 0.0};
        StagedDownloads dl(c);
        dl.add(centers_inout, L.c3, sizeof(double) * 3 * k);
        if (h.needs_host) {
            // the iteration that met an empty cluster: its sums / counts / changed labels, centres untouched
            dl.add(sums_out, L.out, sizeof(double) * 4 * k);
            dl.add(tail, L.out + 4 * (size_t)k, sizeof tail);
        }
        HGMM_HIP(c, dl.finish());
        if (h.needs_host && n_changed_out) *n_changed_out = (int64_t)tail[1];
    }
    if (n_iter_out) *n_iter_out = h.it;
    if (strict_out) *strict_out = h.strict;
    if (needs_host_out) *needs_host_out = h.needs_host;
    return HGMM_OK;
}

extern "C" int hgmm_kmeans_labels(hgmm_ctx* c, int32_t* labels_out, double* min_dist2_out) {
    HGMM_ENTER(c);
    if (c->km_labels_n != c->n || c->n <= 0 || !c->km_labels.p)
        return fail(c, HGMM_ERR_STATE, "kmeans: no assignment on the device (call hgmm_kmeans_step first)");
    if (labels_out)
        HGMM_HIP(c, hipMemcpyAsync(labels_out, c->km_labels.p, sizeof(int32_t) * c->n, hipMemcpyDeviceToHost, c->stream));
    if (min_dist2_out)
        HGMM_HIP(c, hipMemcpyAsync(min_dist2_out, c->km_mind2.p, sizeof(double) * c->n, hipMemcpyDeviceToHost, c->stream));
    HGMM_HIP(c, ctx_stream_sync(c));
    return HGMM_OK;
}
