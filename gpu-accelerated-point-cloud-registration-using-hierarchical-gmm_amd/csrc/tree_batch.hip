// Batched HGMM: B independent clouds per launch set ("forest").
//
// The reference's unit of work is ONE scan pair -- registration_gmmtree(source, target): buildGMMTree of the source
// (src/python/hgmm/hgmm_gpu.py:466-548) + GMMTree.registration of the target (hgmm_gpu.py:754-768, E-step 550-577), called
// at hgmm_gpu.py:802-807.  A 40 k-point pair is ~350 launches whose dependent trips to memory leave an MI355X > 95 % idle
// (profiles/r05/kernel_trace_pair.txt); host threads driving several contexts top out near 1000 pairs/s.  Here B pairs
// share every launch:
//
//   hgmm_tree_build_batch      the B source clouds lie back to back in the context's resident cloud.  At level l the forest
//                              has B 8^l parent segments (segment p: cloud p >> 3 l); the partition kernels, the chunk table and
//                              the E-step work on segments and never notice; the moments, the log-likelihood and the stop rule
//                              find their cloud from the segment / the workgroup index and keep ONE state per cloud (stop
//                              flag, loop state, q shares, trace, progress word).  The levels run in lock-step: a cloud
//                              whose level has stopped costs a returned workgroup per launch until the last one stops.
//   hgmm_tree_set_targets_batch / hgmm_tree_register_batch
//                              the B targets back to back, one table entry per pair (transform, fixed-point encoding,
//                              active flag), one E-step launch + one normal-equations launch per iteration for ALL pairs,
//                              the B 6 x 6 solves on the host (north_star keeps the rigid solve there) between them.
//
// Every workgroup runs the arithmetic of the serial call for its cloud -- the device code is the SAME functions
// (csrc/tree_device.h), the chunks, blocks and orders of summation are the serial ones, the host steps are the same inline
// functions -- so trees, iteration counts, q traces and (R, t) are bitwise those of hgmm_tree_build / hgmm_tree_register
// (tests/test_tree_batch_gpu.py).
#include "tree_device.h"

namespace hgmm {

constexpr TreeFollow NO_FOLLOW{nullptr, 0, nullptr, nullptr, nullptr, 0.0, 0, nullptr, 0, nullptr};
constexpr TreeStop NO_STOP{nullptr, 0.0, 0, nullptr, 0};

// ------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------
__global__ void forest_prep_kernel(const double* __restrict__ pi, const double* __restrict__ mu,
                                   const double* __restrict__ cov, int64_t n_nodes, int T, double* __restrict__ prep,
                                   int* __restrict__ flags /*[B]*/) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_nodes) return;
    const double* c = cov + 9 * j;
    prep_node(pi[j], mu[3 * j], mu[3 * j + 1], mu[3 * j + 2], c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7], c[8],
              prep + PREP_N * j, flags + j / T);
}

// (An XCD-aware work order -- XCD x taking the x-th contiguous eighth of the chunks instead of every eighth chunk, so that a
//  few clouds' node parameters stay in one XCD's scalar caches and L2 -- was built and measured in round 6: the build of 32
//  bunny scans went from 8.9 to 13.9 ms, 3650 -> 2740 pairs/s.  Eight widely spaced streams through the point arrays cost
//  more than the scalar loads' ~800-cycle misses; the grid order stays.)
// launch 0 of a level: the E-step of every chunk of every cloud
template <bool HALF>
__global__ __launch_bounds__(CH, 7) void forest_estep_kernel(TreeEstepArgs ea, ForestArgs fa) {
    __shared__ double smem[tree_estep_lds<HALF>()];
    tree_estep_body<HALF, true>((int)blockIdx.x, ea, NO_FOLLOW, smem, &fa);
}

// One wave per child node of the level, all clouds: tree_moments_kernel with the cloud's own stop flag, point count and --
// from the level's second iteration on -- the cloud's own verdict on the previous iteration's q (tree_follow_wave: this
// launch is the one that follows the log-likelihood; the cloud's first wave speaks for it).
__global__ __launch_bounds__(64) void forest_moments_kernel(const double* __restrict__ partials,
                                                            const int* __restrict__ chunk_first, int n_level,
                                                            double* __restrict__ mom, int64_t lb, double ld, double* pi,
                                                            double* mu, double* cov, double* prep, int* __restrict__ flags,
                                                            ForestArgs fa, const double* __restrict__ block_q, int e) {
    const int cg = blockIdx.x;                       // forest-wide child index: cloud b's children are [b n_level, (b + 1) n_level)
    const int b = cg / n_level, cl = cg - b * n_level;
    const int seg = cg >> 3, k = cg & 7;             // (n_level is a multiple of 8: cg >> 3 = b 8^l + (cl >> 3), the parent segment)
    ForestCloud* fc = fa.clouds + b;
    const int stop_flag = fc->done;                  // (the cloud's entry and the chunk range: one trip)
    const int q_first = fc->q_first, q_count = fc->q_count;
    const double n_total = fc->n_total;
    const int c0 = chunk_first[seg], c1 = chunk_first[seg + 1];
    if (e >= 1 && c0 == c1 && cl != 0) return;      // (children of a parent without points: tree_moments_kernel)
    const TreeFollow follow = forest_follow(fa, b, e, block_q, q_first, q_count);
    TreeFollowLoads fl;
    if (e >= 1) fl = tree_follow_wave_load(follow);
    else if (stop_flag) return;
    double acc[NMOM];
    tree_moments_gather(partials, c0, c1, k, acc);
    if (e >= 1 && tree_follow_wave_verdict(follow, fl, stop_flag, cl == 0)) return;
#pragma unroll
    for (int m = 0; m < NMOM; ++m) acc[m] = wave_sum_f64(acc[m]);
    if (threadIdx.x == 0) {
        const int64_t node = (int64_t)b * fa.T + lb + cl;
#pragma unroll
        for (int m = 0; m < NMOM; ++m) mom[(size_t)node * NMOM + m] = acc[m];
        mstep_node(acc, node, n_total, ld, pi, mu, cov, prep, flags + b, /*with_complexity=*/false);
    }
}

// The same for the eight children of one parent per wave (tree_moments_gather8: the same sums bit for bit)
__global__ __launch_bounds__(64) void forest_moments8_kernel(const double* __restrict__ partials,
                                                             const int* __restrict__ chunk_first, int n_level,
                                                             double* __restrict__ mom, int64_t lb, double ld, double* pi,
                                                             double* mu, double* cov, double* prep, int* __restrict__ flags,
                                                             ForestArgs fa, const double* __restrict__ block_q, int e) {
    const int seg = blockIdx.x;                      // forest-wide parent segment; its children are cg = 8 seg + k
    const int b = (8 * seg) / n_level, cl0 = 8 * seg - b * n_level;
    ForestCloud* fc = fa.clouds + b;
    const int stop_flag = fc->done;
    const int q_first = fc->q_first, q_count = fc->q_count;
    const double n_total = fc->n_total;
    const int c0 = chunk_first[seg], c1 = chunk_first[seg + 1];
    if (e >= 1 && c0 == c1 && cl0 != 0) return;      // (children of a parent without points: tree_moments_kernel)
    const TreeFollow follow = forest_follow(fa, b, e, block_q, q_first, q_count);
    TreeFollowLoads fl;
    if (e >= 1) fl = tree_follow_wave_load(follow);
    else if (stop_flag) return;
    double acc[NMOM];
    tree_moments_gather8(partials, c0, c1, acc);
    if (e >= 1 && tree_follow_wave_verdict(follow, fl, stop_flag, cl0 == 0)) return;
    if ((threadIdx.x & 7) == 0) {
        const int64_t node = (int64_t)b * fa.T + lb + cl0 + ((int)threadIdx.x >> 3);
#pragma unroll
        for (int m = 0; m < NMOM; ++m) mom[(size_t)node * NMOM + m] = acc[m];
        mstep_node(acc, node, n_total, ld, pi, mu, cov, prep, flags + b, /*with_complexity=*/false);
    }
}

// Iteration e's log-likelihood of every cloud + (with_estep) iteration e + 1's speculative E-step of every chunk, as in
// tree_ll_estep_kernel.  Workgroups [0, B ll_stride): cloud w / ll_stride, point block w % ll_stride (clouds with fewer
// blocks return); the rest: chunks.
// (float64 pdfs: five waves per SIMD, 96 registers, no spills; 4 / 5 / 6 measured 14.2 / 13.9 / 14.4 ms per build of 32 bunny
//  scans -- the kernel keeps the fp64 pipe ~80 % busy at any of them, profiles/r06/pmc_sq_batch32.txt.  float32 pdfs: the
//  launch is latency chains of E-step workgroups for the larger part -- six waves at 80 registers, still without spills)
// F32: the log-likelihood workgroups evaluate their pdfs in float32 (hgmm_tree_set_precision; tree_loglik_f32_body)
template <bool F32>
__global__ __launch_bounds__(CH, F32 ? 6 : 5) void forest_ll_estep_kernel(const double* __restrict__ xs, int64_t n_pad,
                                                             const double* __restrict__ prep, int64_t lb, int n_level,
                                                             double* __restrict__ block_q, const int* __restrict__ flags,
                                                             ForestArgs fa, int ll_stride, TreeEstepArgs ea, int with_estep) {
    constexpr int LL_LDS = F32 ? tree_loglik_f32_lds() : tree_loglik_lds<false>();
    constexpr int LDS = LL_LDS > tree_estep_lds<true>() ? LL_LDS : tree_estep_lds<true>();
    __shared__ __attribute__((aligned(16))) double smem[LDS];
    const int w = (int)blockIdx.x;
    const int n_ll = fa.B * ll_stride;
    if (w < n_ll) {
        const int b = w / ll_stride, bx = w - b * ll_stride;
        const ForestCloud* fc = fa.clouds + b;
        const int gx = fc->ll_gx, gy = fc->ll_gy, per_chunk = fc->ll_per_chunk, pt_first = fc->pt_first,
                  pt_count = fc->pt_count, q_first = fc->q_first, q_count = fc->q_count, stop_flag = fc->done;
        if (bx >= gx || stop_flag) return;
        const TreeLoglikArgs la{xs, (int64_t)pt_first + pt_count, n_pad, prep, (int64_t)b * fa.T + lb, n_level, per_chunk,
                                nullptr, block_q + q_first, nullptr, nullptr, nullptr, NO_STOP, flags + b, nullptr, nullptr,
                                (int64_t)pt_first, q_count};
        if constexpr (F32) tree_loglik_f32_body<2, true>(bx, 0, gx, gy, la, smem);
        else tree_loglik_body<2, false, true>(bx, 0, gx, gy, la, smem);
    } else if (with_estep) {
        tree_estep_body<true, true>(w - n_ll, ea, NO_FOLLOW, smem, &fa);
    }
}

// behind the budget's last iteration: one workgroup per cloud accounts for its q (tree_close_kernel)
__global__ __launch_bounds__(CH) void forest_close_kernel(ForestArgs fa, const double* __restrict__ block_q, int e) {
    __shared__ double sh4[4];
    const int b = blockIdx.x;
    ForestCloud* fc = fa.clouds + b;
    const int stop_flag = fc->done;
    const TreeFollow f = forest_follow(fa, b, e, block_q, fc->q_first, fc->q_count);
    (void)tree_follow(f, stop_flag, sh4, true);
}

// partition: tree_hist_kernel / tree_scatter_kernel with the assignment buffer of the chunk's OWN cloud (iteration e's
// E-step wrote buffer e & 1; a cloud whose level took `it` iterations keeps its assignment in buffer (it - 1) & 1)
__global__ __launch_bounds__(CH) void forest_hist_kernel(const int* __restrict__ cur0, const int* __restrict__ cur1,
                                                         const int* __restrict__ chunk_desc,
                                                         const int* __restrict__ n_chunks, int* __restrict__ hist,
                                                         ForestArgs fa) {
    const int c = blockIdx.x;
    if (c >= *n_chunks) return;
    const int seg = chunk_desc[3 * c], begin = chunk_desc[3 * c + 1], end = chunk_desc[3 * c + 2];
    const int it = fa.clouds[seg >> fa.shift].final_it;
    const int* __restrict__ cur = ((it - 1) & 1) ? cur1 : cur0;
    const int i = begin + (int)threadIdx.x;
    const int key = (i < end) ? (cur[i] & 7) : -1;
    __shared__ int sh[CH / 64][8];
    const int w = wave_in_block();
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const unsigned long long m = __ballot(key == k);
        if (lane_id() == 0) sh[w][k] = __popcll(m);
    }
    __syncthreads();
    if (threadIdx.x < 8) {
        int t = 0;
        for (int ww = 0; ww < CH / 64; ++ww) t += sh[ww][threadIdx.x];
        hist[c * 8 + threadIdx.x] = t;
    }
}
__global__ __launch_bounds__(CH) void forest_scatter_kernel(const double* __restrict__ xs, int64_t n_pad,
                                                            const int* __restrict__ cur0, const int* __restrict__ cur1,
                                                            const int* __restrict__ chunk_desc,
                                                            const int* __restrict__ n_chunks,
                                                            const int* __restrict__ chunk_off, double* __restrict__ xs_new,
                                                            ForestArgs fa) {
    const int c = blockIdx.x;
    if (c >= *n_chunks) return;
    const int seg = chunk_desc[3 * c], begin = chunk_desc[3 * c + 1], end = chunk_desc[3 * c + 2];
    const int it = fa.clouds[seg >> fa.shift].final_it;
    const int* __restrict__ cur = ((it - 1) & 1) ? cur1 : cur0;
    const int i = begin + (int)threadIdx.x;
    const bool active = i < end;
    const int key = active ? (cur[i] & 7) : -1;
    __shared__ int sh[CH / 64][8];
    const int w = wave_in_block(), lane = lane_id();
    int rank_in_wave = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const unsigned long long m = __ballot(key == k);
        if (key == k) rank_in_wave = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) sh[w][k] = __popcll(m);
    }
    __syncthreads();
    if (!active) return;
    int before = 0;
    for (int ww = 0; ww < w; ++ww) before += sh[ww][key];
    const int dst = chunk_off[c * 8 + key] + before + rank_in_wave;
    xs_new[dst] = xs[i];
    xs_new[n_pad + dst] = xs[n_pad + i];
    xs_new[2 * n_pad + dst] = xs[2 * n_pad + i];
}

// targets: [n,3] rows -> the forest's structure of arrays at `first`, and the largest |x|^2 of the cloud (the same
// products and sums, in the same order, as hgmm_tree_set_target's host loop: no fused operations)
template <class IN>                                        // (double, or float widened here: the same float64 values either way)
__global__ void forest_target_kernel(const IN* __restrict__ aos, int64_t n, int64_t first, int64_t pad,
                                     double* __restrict__ soa, unsigned long long* __restrict__ r2max_bits) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    double r2 = 0.0;
    if (i < n) {
        const double x = (double)aos[3 * i], y = (double)aos[3 * i + 1], z = (double)aos[3 * i + 2];
        soa[first + i] = x;
        soa[pad + first + i] = y;
        soa[2 * pad + first + i] = z;
        r2 = __dadd_rn(__dadd_rn(__dmul_rn(x, x), __dmul_rn(y, y)), __dmul_rn(z, z));
        if (!(r2 >= 0.0)) r2 = 0.0;                  // (NaN coordinates: the host's `r2 > r2max` never takes them either)
    }
    // non-negative doubles order like their bit patterns
    unsigned long long bits = (unsigned long long)__double_as_longlong(r2);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long o = __shfl_xor(bits, off);
        bits = o > bits ? o : bits;
    }
    if (lane_id() == 0 && bits != 0ull) atomicMax(r2max_bits, bits);
}

template <int NMQ>
__global__ __launch_bounds__(CH) void forest_reg_estep_kernel(const double* __restrict__ tg, int64_t tg_pad,
                                                              const ForestRegPair* __restrict__ tab,
                                                              const double* __restrict__ prep, int T, int L,
                                                              double lambda_c, unsigned long long* __restrict__ momq, int gx) {
    __shared__ unsigned long long lds[REG_LDS_NODES * NMQ];
    const int item = (int)blockIdx.x;                  // (a one-dimensional grid of gx x B items)
    const int b = item / gx, bx = item - b * gx;
    const ForestRegPair* pr = tab + b;
    const int active = pr->active, first = pr->tg_first, count = pr->tg_count;
    if (!active || (int64_t)bx * CH >= count) return;
    const Rigid tf = pr->tf;
    const double inv_d = pr->inv_d, fix_scale = pr->fix_scale;
    const int64_t li = (int64_t)bx * CH + threadIdx.x;
    tree_reg_estep_body<NMQ>(first + li, li < count, tg, tg_pad, tf, prep + (size_t)PREP_N * T * b, L, lambda_c, inv_d,
                             fix_scale, momq + (size_t)NMQ * T * b, lds);
}

__global__ __launch_bounds__(256) void forest_reg_normal_kernel(unsigned long long* __restrict__ momq,
                                                                const ForestRegPair* __restrict__ tab,
                                                                const double* __restrict__ prep, int T,
                                                                double* __restrict__ out, double* host_out,
                                                                unsigned long long* host_seq, unsigned long long seq) {
    const int b = blockIdx.x;
    const ForestRegPair* pr = tab + b;
    if (!pr->active) return;
    tree_reg_normal_body(momq + (size_t)4 * T * b, pr->d_ext, pr->inv_scale, prep + (size_t)PREP_N * T * b, T, out + 28 * b,
                         host_out + 28 * b, host_seq + b, seq);
}

// reg_device_solve: the normal equations of pair b, then -- one thread -- the host's part of the iteration (reg_device_step)
// and the pair's progress word for the host: (left the loop << 32) | iterations done
__global__ __launch_bounds__(256) void forest_reg_solve_kernel(unsigned long long* __restrict__ momq, ForestRegPair* tab,
                                                               const double* __restrict__ prep, int T,
                                                               double* __restrict__ out, double tol, int max_iter,
                                                               double* __restrict__ trace, unsigned long long* host_words) {
    const int b = blockIdx.x;
    ForestRegPair* pr = tab + b;
    if (!pr->active) return;
    tree_reg_normal_body(momq + (size_t)4 * T * b, pr->d_ext, pr->inv_scale, prep + (size_t)PREP_N * T * b, T, out + 28 * b,
                         nullptr, nullptr, 0ull);
    __syncthreads();                                         // (the 28 sums are in `out`, written by this workgroup)
    if (threadIdx.x == 0) {
        const int it = pr->it;
        reg_device_step(out + 28 * b, pr, tol, max_iter, trace ? trace + ((size_t)b * max_iter + it) * 13 : nullptr);
        __hip_atomic_store(host_words + b, ((unsigned long long)(pr->active ? 0 : 1) << 32) | (unsigned long long)(unsigned)pr->it,
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
// pinned, coherent host block: [0, 8 B) progress words of the build / sequence words of the registration, then the
// registration's 28 numbers per pair
static int forest_host(hgmm_ctx* c, int B, unsigned long long** words, double** out28) {
    const size_t want = (size_t)B * (8 + 28 * 8) + 256;
    if (!c->forest.host || c->forest.host_cap < want) {
        HGMM_HIP(c, ctx_stream_sync(c));
        if (c->forest.host) HGMM_HIP(c, hipHostFree(c->forest.host));
        c->forest.host = nullptr;
        HGMM_HIP(c, hipHostMalloc(&c->forest.host, want, hipHostMallocMapped | hipHostMallocCoherent));
        c->forest.host_cap = want;
        std::memset(c->forest.host, 0, want);
    }
    *words = static_cast<unsigned long long*>(c->forest.host);
    *out28 = reinterpret_cast<double*>(static_cast<char*>(c->forest.host) + (((size_t)B * 8 + 255) & ~(size_t)255));
    return HGMM_OK;
}

static int64_t forest_T(int L) { return level_first(L); }

// The registration loop of B pairs with the device on its own (reg_device_solve): every iteration is two launches -- the
// E-step of all pairs, then per pair the normal equations + reg_device_step -- enqueued by a host that only follows the
// pairs' progress words and keeps a few iterations ahead of the slowest running pair; launches behind a pair's stop
// return at their first load.  tab_dev: B entries; out_dev: 28 B doubles.  The caller's arrays are filled at the end.
int forest_register_on_device(::hgmm_ctx* c, int B, const double* tg, int64_t tg_pad, const int64_t* tg_first,
                              const int64_t* tg_counts, const double* tg_rmax, const double* mu_rmax, const double* prep, int T,
                              int L, unsigned long long* momq, double* rot, double* t, double scale, double lambda_c,
                              int max_iter, double tol, double* q_prev_inout, int32_t* iters_out, int32_t* status_out,
                              double* trace) {
    for (int b = 0; b < B; ++b) { iters_out[b] = 0; status_out[b] = 0; }
    if (max_iter < 1) return HGMM_OK;
    HGMM_TRY(ensure(c, c->fr_reg, (sizeof(ForestRegPair) + 28 * sizeof(double) + sizeof(unsigned long long)) * (size_t)B + 512));
    const size_t trace_bytes = trace ? sizeof(double) * 13 * (size_t)max_iter * B : 0;
    if (trace) HGMM_TRY(ensure(c, c->fr_trace, trace_bytes));
    ForestRegPair* d_tab = c->fr_reg.as<ForestRegPair>();
    double* d_out = reinterpret_cast<double*>(d_tab + B);
    double* d_trace = trace ? c->fr_trace.as<double>() : nullptr;
    unsigned long long* words = nullptr;
    double* unused28 = nullptr;
    HGMM_TRY(forest_host(c, B, &words, &unused28));
    void* d_words = nullptr;
    HGMM_HIP(c, hipHostGetDevicePointer(&d_words, words, 0));
    std::vector<ForestRegPair> tab(B);
    int64_t longest = 0;
    for (int b = 0; b < B; ++b) {
        ForestRegPair& pr = tab[b];
        std::memset(&pr, 0, sizeof pr);
        pr.active = 1;
        pr.tg_first = (int)tg_first[b];
        pr.tg_count = (int)tg_counts[b];
        for (int i = 0; i < 9; ++i) pr.tf.r[i] = rot[9 * b + i];
        for (int i = 0; i < 3; ++i) pr.tf.t[i] = t[3 * b + i];
        pr.tf.s = scale;
        double D = 1.0;
        int Fb = 0;
        reg_encoding(reg_extent(pr.tf, tg_rmax[b], mu_rmax[b]), (double)tg_counts[b], &D, &Fb);
        pr.inv_d = 1.0 / D;
        pr.fix_scale = std::ldexp(1.0, Fb);
        pr.d_ext = D;
        pr.inv_scale = std::ldexp(1.0, -Fb);
        pr.tg_rmax = tg_rmax[b];
        pr.mu_rmax = mu_rmax[b];
        pr.has_q = (q_prev_inout[b] == q_prev_inout[b]) ? 1 : 0;              // (NaN: no previous q)
        pr.q_prev = pr.has_q ? q_prev_inout[b] : 0.0;
        longest = std::max(longest, tg_counts[b]);
        __atomic_store_n(words + b, 0ull, __ATOMIC_RELAXED);
    }
    {
        void* st = nullptr;
        HGMM_TRY(stage_reserve(c, sizeof(ForestRegPair) * B, &st));
        std::memcpy(st, tab.data(), sizeof(ForestRegPair) * B);
        HGMM_HIP(c, hipMemcpyAsync(d_tab, st, sizeof(ForestRegPair) * B, hipMemcpyHostToDevice, c->stream));
    }
    const int ahead = 3;
    int enq = 0;
    unsigned spins = 0;
    while (true) {
        bool all_done = true;
        int it_min = 0x7fffffff;
        unsigned long long sig = 0;
        for (int b = 0; b < B; ++b) {
            const unsigned long long w = __atomic_load_n(words + b, __ATOMIC_RELAXED);
            sig += w;
            if (w >> 32) continue;
            all_done = false;
            it_min = std::min(it_min, (int)(w & 0xffffffffull));
        }
        if (all_done) break;
        if (enq < max_iter && enq - it_min < ahead) {
            {
                ProfScope prof(c, HGMM_K_TREE_REG);
                forest_reg_estep_kernel<4><<<nblk(longest, CH) * (unsigned)B, CH, 0, c->stream>>>(tg, tg_pad, d_tab, prep, T, L, lambda_c, momq,
                                                                                                  (int)nblk(longest, CH));
            }
            forest_reg_solve_kernel<<<B, 256, 0, c->stream>>>(momq, d_tab, prep, T, d_out, tol, max_iter, d_trace,
                                                             static_cast<unsigned long long*>(d_words));
            HGMM_HIP(c, hipGetLastError());
            ++enq;
            spins = 0;
            continue;
        }
        __builtin_ia32_pause();
        if ((++spins & 0x3fff) == 0) {
            const hipError_t qe = hipStreamQuery(c->stream);
            if (qe != hipSuccess && qe != hipErrorNotReady)
                return fail(c, HGMM_ERR_HIP, "registration (device loop): device error: %s", hipGetErrorString(qe));
            if (qe == hipSuccess) {
                unsigned long long sig2 = 0;
                for (int b = 0; b < B; ++b) sig2 += __atomic_load_n(words + b, __ATOMIC_ACQUIRE);
                if (sig2 == sig) return fail(c, HGMM_ERR_STATE, "registration (device loop): no progress (%d iterations enqueued)", enq);
            }
        }
    }
    std::vector<double> trace_host(trace ? (size_t)13 * max_iter * B : 0);
    {
        StagedDownloads dl(c);
        dl.add(tab.data(), d_tab, sizeof(ForestRegPair) * B);
        if (trace) dl.add(trace_host.data(), d_trace, trace_bytes);
        HGMM_HIP(c, dl.finish());
    }
    for (int b = 0; b < B; ++b) {
        const ForestRegPair& pr = tab[b];
        for (int i = 0; i < 9; ++i) rot[9 * b + i] = pr.tf.r[i];
        for (int i = 0; i < 3; ++i) t[3 * b + i] = pr.tf.t[i];
        iters_out[b] = pr.it;
        status_out[b] = pr.status;
        if (pr.has_q) q_prev_inout[b] = pr.q_prev;
        if (trace) std::memcpy(trace + (size_t)13 * max_iter * b, trace_host.data() + (size_t)13 * max_iter * b, sizeof(double) * 13 * (size_t)pr.it);
    }
    return HGMM_OK;
}

}  // namespace hgmm

using namespace hgmm;

extern "C" int hgmm_tree_build_batch(hgmm_ctx* c, int B, const int64_t* counts, int L, double ls, double ld,
                                     const double* init_mu, double sig2, int max_iters_per_level, double* pi_out,
                                     double* mu_out, double* cov_out, int32_t* iters_out, double* q_trace_out,
                                     int q_capacity, int32_t* q_len_out) {
    HGMM_ENTER(c);
    if (!c->have_f64 || c->n <= 0) return fail(c, HGMM_ERR_STATE, "tree build (batch): set points first");
    if (B < 1 || B > 4096 || !counts) return fail(c, HGMM_ERR_ARG, "tree build (batch): B = %d clouds", B);
    if (L < 1 || L > 6) return fail(c, HGMM_ERR_ARG, "tree levels L = %d outside 1..6", L);
    if (!init_mu) return fail(c, HGMM_ERR_ARG, "init_mu is NULL");
    if (c->comm_on()) return fail(c, HGMM_ERR_STATE, "tree build (batch): independent clouds take no communicator");
    if (c->n > 0x7fffffff - 1024) return fail(c, HGMM_ERR_ARG, "too many points for 32-bit indices");
    if (max_iters_per_level < 1) max_iters_per_level = 1;
    const int64_t n = c->n, n_pad = c->n_pad;
    {
        int64_t sum = 0;
        for (int b = 0; b < B; ++b) {
            if (counts[b] < 1) return fail(c, HGMM_ERR_ARG, "tree build (batch): cloud %d has no points", b);
            // (a cloud of >= 400 000 points fills the chip by itself and takes the serial build's four-points-per-thread
            //  log-likelihood, which this path does not reproduce)
            if (counts[b] >= 400000)
                return fail(c, HGMM_ERR_ARG, "tree build (batch): cloud %d has %lld points; clouds of >= 400000 points "
                            "go through hgmm_tree_build", b, (long long)counts[b]);
            sum += counts[b];
        }
        if (sum != n) return fail(c, HGMM_ERR_ARG, "tree build (batch): the counts add up to %lld, the resident cloud has %lld points",
                                  (long long)sum, (long long)n);
    }
    const int64_t T = forest_T(L);
    const int64_t TT = T * B;
    int64_t P8 = 1;
    for (int i = 0; i < L - 1; ++i) P8 *= 8;                   // parents of one cloud at the last level
    const int64_t maxP = P8 * B;
    if (TT > 0x3fffffff || maxP > (1 << 24)) return fail(c, HGMM_ERR_ARG, "tree build (batch): %d clouds x %d levels is too large", B, L);
    const int64_t max_chunks = n / CH + maxP + 8;
    ForestState& F = c->forest;
    F.nodes_ready = false;
    HGMM_TRY(ensure(c, c->fr_pi, sizeof(double) * TT));
    HGMM_TRY(ensure(c, c->fr_mu, sizeof(double) * 3 * TT));
    HGMM_TRY(ensure(c, c->fr_cov, sizeof(double) * 9 * TT));
    HGMM_TRY(ensure(c, c->fr_prep, sizeof(double) * PREP_N * TT));
    HGMM_TRY(ensure(c, c->fr_mom, sizeof(double) * NMOM * TT));
    HGMM_TRY(ensure(c, c->fr_clouds, sizeof(ForestCloud) * B + sizeof(int) * (B + 1) + 256));
    HGMM_TRY(ensure(c, c->fr_q, sizeof(double) * (size_t)(n / CH + B + 8)));
    const int trace_cap = std::min(max_iters_per_level, 4096);
    HGMM_TRY(ensure(c, c->fr_trace, sizeof(double) * (size_t)trace_cap * L * B));
    HGMM_TRY(ensure(c, c->scratch, sizeof(double) * 3 * TT));
    HGMM_TRY(ensure(c, c->t_current, sizeof(int) * 2 * n_pad));
    HGMM_TRY(ensure(c, c->t_parent, sizeof(double) * 3 * n_pad));
    HGMM_TRY(ensure(c, c->t_seg, sizeof(int) * (2 * (8 * maxP + 2) + 2 * (maxP + 2) + 8)));
    HGMM_TRY(ensure(c, c->t_chunks, sizeof(int) * (size_t)(3 + 8 + 8) * max_chunks));
    HGMM_TRY(ensure(c, c->t_partials, sizeof(double) * (size_t)8 * NMOM * max_chunks));
    double* xs_c = nullptr;
    if (L > 2) {
        HGMM_TRY(ensure(c, c->t_xs3, sizeof(double) * 3 * n_pad));
        xs_c = c->t_xs3.as<double>();
    }
    unsigned long long* words = nullptr;
    double* unused28 = nullptr;
    HGMM_TRY(forest_host(c, B, &words, &unused28));
    void* dp = nullptr;
    HGMM_HIP(c, hipHostGetDevicePointer(&dp, words, 0));
    unsigned long long* words_dev = static_cast<unsigned long long*>(dp);

    double* d_pi = c->fr_pi.as<double>();
    double* d_mu = c->fr_mu.as<double>();
    double* d_cov = c->fr_cov.as<double>();
    double* d_prep = c->fr_prep.as<double>();
    double* d_mom = c->fr_mom.as<double>();
    ForestCloud* d_clouds = c->fr_clouds.as<ForestCloud>();
    int* d_flags = reinterpret_cast<int*>(d_clouds + B);
    double* block_q = c->fr_q.as<double>();
    double* trace_base = c->fr_trace.as<double>();
    double* xs_a = c->x_soa64.as<double>();
    double* xs_b = c->t_parent.as<double>();
    int* cur0 = c->t_current.as<int>();
    int* cur1 = cur0 + n_pad;
    int* seg_a = c->t_seg.as<int>();
    int* seg_b = seg_a + (8 * maxP + 2);
    int* chunk_first = seg_b + (8 * maxP + 2);
    int* n_chunks_dev = chunk_first + (maxP + 2) * 2;
    int* chunk_desc = c->t_chunks.as<int>();
    int* hist = chunk_desc + 3 * max_chunks;
    int* chunk_off = hist + 8 * max_chunks;
    double* partials = c->t_partials.as<double>();

    // node tables: pi = 1/8, mu = given, cov = sig2 I for every tree; per-cloud form flags start at zero
    HGMM_HIP(c, hipMemcpyAsync(c->scratch.p, init_mu, sizeof(double) * 3 * TT, hipMemcpyHostToDevice, c->stream));
    HGMM_HIP(c, hipMemsetAsync(d_flags, 0, sizeof(int) * (B + 1), c->stream));
    tree_init_nodes_kernel<<<nblk(TT, 256), 256, 0, c->stream>>>(c->scratch.as<double>(), sig2, TT, d_pi, d_mu, d_cov);
    forest_prep_kernel<<<nblk(TT, 256), 256, 0, c->stream>>>(d_pi, d_mu, d_cov, TT, (int)T, d_prep, d_flags);
    HGMM_HIP(c, hipGetLastError());
    std::vector<int> first(B + 1, 0);
    for (int b = 0; b < B; ++b) first[b + 1] = first[b] + (int)counts[b];
    {
        void* st = nullptr;
        HGMM_TRY(stage_reserve(c, sizeof(int) * (B + 1), &st));
        std::memcpy(st, first.data(), sizeof(int) * (B + 1));
        HGMM_HIP(c, hipMemcpyAsync(seg_a, st, sizeof(int) * (B + 1), hipMemcpyHostToDevice, c->stream));
    }

    const double* xs_cur = xs_a;
    int* seg_cur = seg_a;
    int P = B;
    std::vector<int> level_iters((size_t)B * L, 0);
    std::vector<ForestCloud> table(B);
    int rc = HGMM_OK;
    const int ahead_iters = 2;                                  // iterations enqueued beyond the slowest running cloud (hgmm_tree_build)
    for (int l = 0; l < L && rc == HGMM_OK; ++l) {
        const int64_t lb = level_first(l), le = level_first(l + 1);
        const int n_level = (int)(le - lb);
        const int64_t parent_first = (l == 0) ? 0 : level_first(l - 1);
        // the clouds' entries for this level: the decomposition the serial build would use for each of them
        int q_at = 0, ll_stride = 1;
        for (int b = 0; b < B; ++b) {
            ForestCloud& fc = table[b];
            std::memset(&fc, 0, sizeof fc);
            fc.pt_first = first[b];
            fc.pt_count = (int)counts[b];
            const int llblocks = (int)nblk(counts[b], CH * 2);
            int chunks = 1, per_chunk = n_level;
            tree_ll_split(llblocks, n_level, c->cus, &chunks, &per_chunk);
            fc.ll_gx = llblocks;
            fc.ll_gy = chunks;
            fc.ll_per_chunk = per_chunk;
            fc.q_first = q_at;
            // (level 0: the E-step stores the shares, one per chunk -- a cloud's chunks are consecutive, TreeEstepArgs::q_shares)
            fc.q_count = (l == 0 || chunks > 1) ? (int)nblk(counts[b], CH) : llblocks;
            q_at += fc.q_count;
            fc.n_total = (double)counts[b];
            ll_stride = std::max(ll_stride, llblocks);
        }
        if (l == 0) ll_stride = 0;                              // no log-likelihood workgroups at level 0
        {
            void* st = nullptr;
            rc = stage_reserve(c, sizeof(ForestCloud) * B, &st);
            if (rc != HGMM_OK) break;
            std::memcpy(st, table.data(), sizeof(ForestCloud) * B);
            if (hipMemcpyAsync(d_clouds, st, sizeof(ForestCloud) * B, hipMemcpyHostToDevice, c->stream) != hipSuccess) {
                rc = fail(c, HGMM_ERR_HIP, "tree build (batch): upload of the cloud table failed");
                break;
            }
        }
        const ForestArgs fa{d_clouds, B, (int)T, 3 * l, ls, max_iters_per_level, trace_base, trace_cap, L, l, words_dev};
        tree_chunks_kernel<<<1, 1024, 0, c->stream>>>(seg_cur, P, chunk_first, chunk_desc, n_chunks_dev);
        const unsigned grid_chunks = (unsigned)(n / CH + P + 1);
        for (int b = 0; b < B; ++b) __atomic_store_n(words + b, 0ull, __ATOMIC_RELAXED);
        auto enqueue_iteration = [&](int e) -> int {
            const TreeEstepArgs ea_now{xs_cur, n_pad, d_prep, chunk_desc, n_chunks_dev, parent_first, l, partials,
                                       (e & 1) ? cur1 : cur0, nullptr};
            if (e == 0) {
                ProfScope prof(c, HGMM_K_TREE_ESTEP);
                forest_estep_kernel<true><<<grid_chunks, CH, 0, c->stream>>>(ea_now, fa);
            }
            if (l > 0)
                forest_moments8_kernel<<<(unsigned)(B * n_level / 8), 64, 0, c->stream>>>(partials, chunk_first, n_level, d_mom, lb, ld,
                                                                                         d_pi, d_mu, d_cov, d_prep, d_flags, fa, block_q, e);
            else                                                    // (level 0: a wave per child, tree_moments8_kernel's note)
                forest_moments_kernel<<<(unsigned)(B * n_level), 64, 0, c->stream>>>(partials, chunk_first, n_level, d_mom, lb, ld,
                                                                                    d_pi, d_mu, d_cov, d_prep, d_flags, fa, block_q, e);
            {
                ProfScope prof(c, HGMM_K_TREE_LOGLIK);
                // (level 0: behind the budget's last iteration the E-step runs for the shares of q alone)
                const int with_estep = (e + 1 < max_iters_per_level || l == 0) ? 1 : 0;
                const TreeEstepArgs ea_next{xs_cur, n_pad, d_prep, chunk_desc, n_chunks_dev, parent_first, l, partials,
                                            ((e + 1) & 1) ? cur1 : cur0, nullptr, l == 0 ? block_q : nullptr};
                const unsigned g = (unsigned)(B * ll_stride) + (with_estep ? grid_chunks : 0u);
                if (l == 0)
                    // (level 0 is E-step workgroups only: the plain E-step kernel -- 72 registers instead of the fused kernel's
                    //  92-96, one more wave per SIMD for a launch that is a chain of trips to memory)
                    forest_estep_kernel<true><<<grid_chunks, CH, 0, c->stream>>>(ea_next, fa);
                else if (c->tree.pdf_f32)
                    forest_ll_estep_kernel<true><<<g, CH, 0, c->stream>>>(xs_cur, n_pad, d_prep, lb, n_level, block_q, d_flags,
                                                                         fa, ll_stride, ea_next, with_estep);
                else
                    forest_ll_estep_kernel<false><<<g, CH, 0, c->stream>>>(xs_cur, n_pad, d_prep, lb, n_level, block_q, d_flags,
                                                                          fa, ll_stride, ea_next, with_estep);
            }
            const hipError_t le = hipGetLastError();
            if (le != hipSuccess) return fail(c, HGMM_ERR_HIP, "tree build (batch): kernel launch failed: %s", hipGetErrorString(le));
            return HGMM_OK;
        };
        // The host keeps `ahead` iterations enqueued beyond the slowest cloud that is still running (every cloud's speaker
        // stores (stopped << 32 | iterations) into its own word of pinned host memory) and leaves the level once every
        // cloud has stopped.
        int enq = 0;
        unsigned spins = 0;
        while (rc == HGMM_OK) {
            bool all_done = true;
            int it_min = 0x7fffffff;
            unsigned long long sig = 0;
            for (int b = 0; b < B; ++b) {
                const unsigned long long w = __atomic_load_n(words + b, __ATOMIC_RELAXED);
                sig += w;
                if (w >> 32) continue;
                all_done = false;
                it_min = std::min(it_min, (int)(w & 0xffffffffull));
            }
            if (all_done) break;
            if (enq < max_iters_per_level && enq - it_min < ahead_iters) {
                rc = enqueue_iteration(enq);
                ++enq;
                if (rc == HGMM_OK && enq == max_iters_per_level) {
                    forest_close_kernel<<<B, CH, 0, c->stream>>>(fa, block_q, enq);
                    if (hipGetLastError() != hipSuccess) rc = fail(c, HGMM_ERR_HIP, "tree build (batch): launch failed");
                }
                spins = 0;
                continue;
            }
            __builtin_ia32_pause();
            if ((++spins & 0x3fff) == 0) {                      // every ~16k polls: is the device still alive?
                const hipError_t qe = hipStreamQuery(c->stream);
                if (qe != hipSuccess && qe != hipErrorNotReady) {
                    rc = fail(c, HGMM_ERR_HIP, "tree build (batch): device error: %s", hipGetErrorString(qe));
                } else if (qe == hipSuccess) {
                    unsigned long long sig2_ = 0;
                    for (int b = 0; b < B; ++b) sig2_ += __atomic_load_n(words + b, __ATOMIC_ACQUIRE);
                    if (sig2_ == sig)
                        rc = fail(c, HGMM_ERR_STATE, "tree build (batch): level %d made no progress (%d iterations enqueued)", l, enq);
                }
            }
        }
        if (rc != HGMM_OK) break;
        for (int b = 0; b < B; ++b) level_iters[(size_t)b * L + l] = (int)(__atomic_load_n(words + b, __ATOMIC_ACQUIRE) & 0xffffffffull);
        if (l + 1 < L) {
            forest_hist_kernel<<<grid_chunks, CH, 0, c->stream>>>(cur0, cur1, chunk_desc, n_chunks_dev, hist, fa);
            int* seg_next = (seg_cur == seg_a) ? seg_b : seg_a;
            tree_offsets_kernel<<<P, OFF_BLOCK, 0, c->stream>>>(hist, chunk_first, seg_cur, P, chunk_off, seg_next);
            double* xs_next = (xs_cur == xs_b) ? xs_c : xs_b;   // A -> B -> C -> B -> ... (the resident cloud is never overwritten)
            forest_scatter_kernel<<<grid_chunks, CH, 0, c->stream>>>(xs_cur, n_pad, cur0, cur1, chunk_desc, n_chunks_dev, chunk_off,
                                                                    xs_next, fa);
            if (hipGetLastError() != hipSuccess) { rc = fail(c, HGMM_ERR_HIP, "tree build (batch): partition launch failed"); break; }
            xs_cur = xs_next;
            seg_cur = seg_next;
            P *= 8;
        }
    }
    if (rc != HGMM_OK) {
        (void)ctx_stream_sync(c);
        return rc;
    }
    // The serial pair goes on through hgmm_tree_set_nodes: the finished tables are prepared once more, by tree_prep_kernel
    // (complexity ratio included).  The SAME kernel runs here over the whole forest, so that the registration reads the very
    // numbers it reads after a serial build (the per-iteration preparation inside the moments kernel is the same function
    // inlined elsewhere -- the compiler need not contract it alike).  Its one flags word is a scratch word behind the clouds'.
    tree_prep_kernel<<<nblk(TT, 256), 256, 0, c->stream>>>(d_pi, d_mu, d_cov, 0, TT, d_prep, d_flags + B);
    HGMM_HIP(c, hipGetLastError());
    // the means always come back (the registration's extent bound needs the largest |mu| per tree); the rest on request
    std::vector<double> mu_host;
    double* mu_dst = mu_out;
    if (!mu_dst) { mu_host.resize((size_t)3 * TT); mu_dst = mu_host.data(); }
    std::vector<double> trace_host;
    if (q_trace_out) trace_host.resize((size_t)trace_cap * L * B);
    {
        StagedDownloads dl(c);
        dl.add(mu_dst, d_mu, sizeof(double) * 3 * TT);
        dl.add(pi_out, d_pi, sizeof(double) * TT);
        dl.add(cov_out, d_cov, sizeof(double) * 9 * TT);
        if (q_trace_out) dl.add(trace_host.data(), trace_base, sizeof(double) * trace_host.size());
        HGMM_HIP(c, dl.finish());
    }
    F.B = B;
    F.L = L;
    F.T = (int)T;
    F.counts.assign(counts, counts + B);
    F.mu_rmax.resize(B);
    for (int b = 0; b < B; ++b) F.mu_rmax[b] = tree_mu_rmax(mu_dst + (size_t)3 * T * b, T);
    for (int b = 0; b < B; ++b) {
        int at = 0;
        for (int l = 0; l < L; ++l) {
            const int it = level_iters[(size_t)b * L + l];
            if (iters_out) iters_out[(size_t)b * L + l] = it;
            if (q_trace_out) {
                const int take = std::min(std::min(it, trace_cap), q_capacity - at);
                if (take > 0)
                    std::memcpy(q_trace_out + (size_t)b * q_capacity + at, trace_host.data() + ((size_t)b * L + l) * trace_cap,
                                sizeof(double) * take);
            }
            at += it;
        }
        if (q_len_out) q_len_out[b] = at < q_capacity ? at : q_capacity;
    }
    F.nodes_ready = true;
    return HGMM_OK;
}

extern "C" int hgmm_tree_get_nodes_batch(hgmm_ctx* c, int b, double* pi_out, double* mu_out, double* cov_out) {
    HGMM_ENTER(c);
    const ForestState& F = c->forest;
    if (!F.nodes_ready) return fail(c, HGMM_ERR_STATE, "no forest (hgmm_tree_build_batch first)");
    if (b < 0 || b >= F.B) return fail(c, HGMM_ERR_ARG, "tree %d of %d", b, F.B);
    const size_t T = (size_t)F.T;
    StagedDownloads dl(c);
    dl.add(pi_out, c->fr_pi.as<double>() + T * b, sizeof(double) * T);
    dl.add(mu_out, c->fr_mu.as<double>() + 3 * T * b, sizeof(double) * 3 * T);
    dl.add(cov_out, c->fr_cov.as<double>() + 9 * T * b, sizeof(double) * 9 * T);
    HGMM_HIP(c, dl.finish());
    return HGMM_OK;
}

template <class IN>
static int set_targets_batch(hgmm_ctx* c, int B, const IN* const* xyz, const int64_t* counts) {
    HGMM_ENTER(c);
    if (B < 1 || B > 4096 || !xyz || !counts) return fail(c, HGMM_ERR_ARG, "targets (batch): B = %d", B);
    ForestState& F = c->forest;
    F.tg_B = 0;
    int64_t total = 0, longest = 0;
    for (int b = 0; b < B; ++b) {
        if (!xyz[b] || counts[b] < 1) return fail(c, HGMM_ERR_ARG, "targets (batch): target %d is empty", b);
        total += counts[b];
        longest = std::max(longest, counts[b]);
    }
    if (total > 0x7fffffff - 1024) return fail(c, HGMM_ERR_ARG, "too many target points for 32-bit indices");
    const int64_t pad = (total + 255) / 256 * 256;
    HGMM_TRY(ensure(c, c->fr_tg, sizeof(double) * 3 * pad));
    HGMM_TRY(ensure(c, c->scratch, sizeof(IN) * 3 * (size_t)total));
    HGMM_TRY(ensure(c, c->fr_reg, (sizeof(ForestRegPair) + 28 * sizeof(double) + sizeof(unsigned long long)) * (size_t)B + 512));
    // layout of fr_reg: [pairs table][28 B doubles][B r2max words]
    unsigned long long* r2bits = reinterpret_cast<unsigned long long*>(c->fr_reg.as<char>() +
                                                                     (sizeof(ForestRegPair) + 28 * sizeof(double)) * (size_t)B);
    HGMM_HIP(c, hipMemsetAsync(r2bits, 0, sizeof(unsigned long long) * B, c->stream));
    F.tg_counts.assign(counts, counts + B);
    F.tg_first.assign(B, 0);
    int64_t at = 0;
    IN* stage = c->scratch.as<IN>();
    for (int b = 0; b < B; ++b) {
        F.tg_first[b] = at;
        HGMM_HIP(c, hipMemcpyAsync(stage + 3 * at, xyz[b], sizeof(IN) * 3 * (size_t)counts[b], hipMemcpyHostToDevice, c->stream));
        forest_target_kernel<IN><<<nblk(counts[b], 256), 256, 0, c->stream>>>(stage + 3 * at, counts[b], at, pad, c->fr_tg.as<double>(),
                                                                             r2bits + b);
        at += counts[b];
    }
    HGMM_HIP(c, hipGetLastError());
    std::vector<unsigned long long> bits(B);
    {
        StagedDownloads dl(c);
        dl.add(bits.data(), r2bits, sizeof(unsigned long long) * B);
        HGMM_HIP(c, dl.finish());
    }
    F.tg_rmax.resize(B);
    for (int b = 0; b < B; ++b) {
        double r2;
        std::memcpy(&r2, &bits[b], sizeof r2);
        F.tg_rmax[b] = std::sqrt(r2);
    }
    F.tg_pad = pad;
    F.tg_B = B;
    return HGMM_OK;
}

extern "C" int hgmm_tree_set_targets_batch(hgmm_ctx* c, int B, const double* const* xyz, const int64_t* counts) {
    return set_targets_batch<double>(c, B, xyz, counts);
}
extern "C" int hgmm_tree_set_targets_batch_f32(hgmm_ctx* c, int B, const float* const* xyz, const int64_t* counts) {
    return set_targets_batch<float>(c, B, xyz, counts);
}

extern "C" int hgmm_tree_register_batch(hgmm_ctx* c, int B, double* rot, double* t, double scale, double lambda_c,
                                        int max_iter, double tol, double* q_prev_inout, int32_t* iters_out,
                                        int32_t* status_out, double* trace) {
    HGMM_ENTER(c);
    if (!rot || !t || !q_prev_inout || !iters_out || !status_out) return fail(c, HGMM_ERR_ARG, "tree_register (batch): NULL argument");
    ForestState& F = c->forest;
    if (!F.nodes_ready) return fail(c, HGMM_ERR_STATE, "registration (batch): no forest (hgmm_tree_build_batch first)");
    if (B != F.B || B != F.tg_B)
        return fail(c, HGMM_ERR_STATE, "registration (batch): %d pairs, but %d trees and %d targets are resident", B, F.B, F.tg_B);
    const int T = F.T, L = F.L;
    const size_t momq_bytes = sizeof(unsigned long long) * 4 * (size_t)T * B;
    if (c->fr_momq.cap < momq_bytes || !c->fr_momq.p) F.momq_clean = false;
    HGMM_TRY(ensure(c, c->fr_momq, momq_bytes));
    if (!F.momq_clean) {
        HGMM_HIP(c, hipMemsetAsync(c->fr_momq.p, 0, c->fr_momq.cap, c->stream));
        F.momq_clean = true;
    }
    if (c->cfg[CFG_REG_DEVICE_SOLVE])
        return forest_register_on_device(c, B, c->fr_tg.as<double>(), F.tg_pad, F.tg_first.data(), F.tg_counts.data(),
                                         F.tg_rmax.data(), F.mu_rmax.data(), c->fr_prep.as<double>(), T, L,
                                         c->fr_momq.as<unsigned long long>(), rot, t, scale, lambda_c, max_iter, tol,
                                         q_prev_inout, iters_out, status_out, trace);
    unsigned long long* words = nullptr;
    double* h_out = nullptr;
    HGMM_TRY(forest_host(c, B, &words, &h_out));
    void *d_words = nullptr, *d_hout = nullptr;
    HGMM_HIP(c, hipHostGetDevicePointer(&d_words, words, 0));
    HGMM_HIP(c, hipHostGetDevicePointer(&d_hout, h_out, 0));
    ForestRegPair* d_tab = c->fr_reg.as<ForestRegPair>();
    double* d_out = reinterpret_cast<double*>(d_tab + B);
    std::vector<ForestRegPair> tab(B);
    std::vector<char> active(B, 1);
    int64_t longest = 0;
    for (int b = 0; b < B; ++b) {
        iters_out[b] = 0;
        status_out[b] = 0;                                        // 0: budget used up, 1: |dq| < tol, 2: host M-step needed
        longest = std::max(longest, F.tg_counts[b]);
    }
    int n_active = B;
    for (int it = 0; it < max_iter && n_active > 0; ++it) {
        for (int b = 0; b < B; ++b) {
            ForestRegPair& pr = tab[b];
            std::memset(&pr, 0, sizeof pr);
            pr.active = active[b];
            pr.tg_first = (int)F.tg_first[b];
            pr.tg_count = (int)F.tg_counts[b];
            if (!active[b]) continue;
            for (int i = 0; i < 9; ++i) pr.tf.r[i] = rot[9 * b + i];
            for (int i = 0; i < 3; ++i) pr.tf.t[i] = t[3 * b + i];
            pr.tf.s = scale;
            double D = 1.0;
            int Fb = 0;
            reg_encoding(reg_extent(pr.tf, F.tg_rmax[b], F.mu_rmax[b]), (double)F.tg_counts[b], &D, &Fb);
            pr.inv_d = 1.0 / D;
            pr.fix_scale = std::ldexp(1.0, Fb);
            pr.d_ext = D;
            pr.inv_scale = std::ldexp(1.0, -Fb);
        }
        void* st = nullptr;
        HGMM_TRY(stage_reserve(c, sizeof(ForestRegPair) * B, &st));
        std::memcpy(st, tab.data(), sizeof(ForestRegPair) * B);
        HGMM_HIP(c, hipMemcpyAsync(d_tab, st, sizeof(ForestRegPair) * B, hipMemcpyHostToDevice, c->stream));
        const unsigned long long seq = ++F.seq;
        {
            ProfScope prof(c, HGMM_K_TREE_REG);
            forest_reg_estep_kernel<4><<<nblk(longest, CH) * (unsigned)B, CH, 0, c->stream>>>(
                c->fr_tg.as<double>(), F.tg_pad, d_tab, c->fr_prep.as<double>(), T, L, lambda_c, c->fr_momq.as<unsigned long long>(),
                (int)nblk(longest, CH));
        }
        forest_reg_normal_kernel<<<B, 256, 0, c->stream>>>(c->fr_momq.as<unsigned long long>(), d_tab, c->fr_prep.as<double>(), T,
                                                          d_out, static_cast<double*>(d_hout),
                                                          static_cast<unsigned long long*>(d_words), seq);
        HGMM_HIP(c, hipGetLastError());
        // every active pair's normal equations arrive with its own sequence word; each is solved as soon as it is there
        std::vector<char> pending(active);
        int n_pending = n_active;
        unsigned spins = 0;
        while (n_pending > 0) {
            bool progressed = false;
            for (int b = 0; b < B; ++b) {
                if (!pending[b] || __atomic_load_n(words + b, __ATOMIC_ACQUIRE) != seq) continue;
                pending[b] = 0;
                --n_pending;
                progressed = true;
                double q = 0.0;
                const int stp = reg_host_step(h_out + 28 * b, rot + 9 * b, t + 3 * b, q_prev_inout + b, tol, &q);
                if (stp == 2) { status_out[b] = 2; active[b] = 0; --n_active; continue; }
                if (trace) {
                    double* tr = trace + ((size_t)b * max_iter + it) * 13;
                    for (int i = 0; i < 9; ++i) tr[i] = rot[9 * b + i];
                    for (int i = 0; i < 3; ++i) tr[9 + i] = t[3 * b + i];
                    tr[12] = q;
                }
                iters_out[b] = it + 1;
                if (stp == 1) { status_out[b] = 1; active[b] = 0; --n_active; }
            }
            if (progressed) { spins = 0; continue; }
            __builtin_ia32_pause();
            if ((++spins & 0x3fff) == 0) {
                const hipError_t qe = hipStreamQuery(c->stream);
                if (qe != hipSuccess && qe != hipErrorNotReady)
                    return fail(c, HGMM_ERR_HIP, "registration (batch): device error: %s", hipGetErrorString(qe));
                if (qe == hipSuccess) {
                    bool still = false;
                    for (int b = 0; b < B; ++b) still = still || (pending[b] && __atomic_load_n(words + b, __ATOMIC_ACQUIRE) != seq);
                    if (still) return fail(c, HGMM_ERR_STATE, "registration (batch): the normal-equation kernel did not report (sequence %llu)", seq);
                }
            }
        }
    }
    return HGMM_OK;
}
