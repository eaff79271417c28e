// Gauss transform of the L2 GMMReg registration cost for gfx950, float64.
//
// The reference evaluates  G_k(y_i) = sum_j w_kj exp(-|y_i - s_j|^2 / h^2)  on the host, one
// np.apply_along_axis pass per weight row (src/python/gmmreg_gpu/transforms.py:43-49), four times per
// cost evaluation (cost_functions.py:29-40: w = phi_t / z and phi_t mu_t / z) and once per BFGS
// function call.  Both point sets are mixture means (J <= a few thousand), so the whole transform
// is J_s x J_t kernel evaluations: one launch, lanes across the evaluation points, the centres
// tiled through LDS, the centre range split over gridDim.y so that even J = 800 fills the chip.
// The per-split partial sums are added on the host in split order (deterministic).
#include "hgmm_ctx.h"
#include "wave_ops.h"

#include <algorithm>
#include <cmath>
#include <vector>

namespace hgmm {

constexpr int GT_BLOCK = 64;         // evaluation points per workgroup (one wave)
constexpr int GT_TILE = 128;         // centres per LDS tile
constexpr int GT_MAX_W = 8;          // weight rows

template <int NW>
__global__ __launch_bounds__(GT_BLOCK) void gauss_transform_kernel(
    const double* __restrict__ centres /*[n_c][3]*/, int n_c, const double* __restrict__ pts /*[n_p][3]*/,
    int n_p, const double* __restrict__ weights /*[NW][n_c]*/, double inv_h2, int per_split,
    double* __restrict__ partial /*[splits][NW][n_p]*/) {
    __shared__ double tile[GT_TILE][3 + NW];
    const int i = blockIdx.x * GT_BLOCK + threadIdx.x;
    const bool live = i < n_p;
    double y0 = 0.0, y1 = 0.0, y2 = 0.0;
    if (live) { y0 = pts[3 * i]; y1 = pts[3 * i + 1]; y2 = pts[3 * i + 2]; }
    double acc[NW];
#pragma unroll
    for (int k = 0; k < NW; ++k) acc[k] = 0.0;
    const int c_begin = blockIdx.y * per_split;
    const int c_end = min(n_c, c_begin + per_split);
    for (int base = c_begin; base < c_end; base += GT_TILE) {
        const int cnt = min(GT_TILE, c_end - base);
        __syncthreads();
        for (int t = threadIdx.x; t < cnt * (3 + NW); t += GT_BLOCK) {
            const int j = t / (3 + NW), f = t % (3 + NW);
            tile[j][f] = f < 3 ? centres[3 * (size_t)(base + j) + f] : weights[(size_t)(f - 3) * n_c + base + j];
        }
        __syncthreads();
        for (int j = 0; j < cnt; ++j) {
            const double d0 = y0 - tile[j][0], d1 = y1 - tile[j][1], d2 = y2 - tile[j][2];
            const double e = exp(-(d0 * d0 + d1 * d1 + d2 * d2) * inv_h2);
#pragma unroll
            for (int k = 0; k < NW; ++k) acc[k] += tile[j][3 + k] * e;
        }
    }
    if (live) {
#pragma unroll
        for (int k = 0; k < NW; ++k) partial[((size_t)blockIdx.y * NW + k) * n_p + i] = acc[k];
    }
}

}  // namespace hgmm

using namespace hgmm;

extern "C" int hgmm_gauss_transform(hgmm_ctx* c, const double* centres, int n_centres, const double* points,
                                    int n_points, const double* weights, int n_weights, double h, double* out) {
    HGMM_ENTER(c);
    if (!centres || !points || !weights || !out) return fail(c, HGMM_ERR_ARG, "gauss transform: NULL argument");
    if (n_centres < 1 || n_points < 1) return fail(c, HGMM_ERR_ARG, "gauss transform: empty point set");
    if (n_weights < 1 || n_weights > GT_MAX_W)
        return fail(c, HGMM_ERR_ARG, "gauss transform: n_weights = %d outside 1..%d", n_weights, GT_MAX_W);
    if (!(h > 0.0)) return fail(c, HGMM_ERR_ARG, "gauss transform: bandwidth must be positive");
    HGMM_HIP(c, hipSetDevice(c->device));
    const int pblocks = (n_points + GT_BLOCK - 1) / GT_BLOCK;
    // split the centre range until there are about two workgroups per CU (but >= one tile per split)
    int splits = std::max(1, std::min((2 * c->cus + pblocks - 1) / pblocks, (n_centres + GT_TILE - 1) / GT_TILE));
    const int per_split = ((n_centres + splits - 1) / splits + GT_TILE - 1) / GT_TILE * GT_TILE;
    splits = (n_centres + per_split - 1) / per_split;
    const size_t in_doubles = 3 * (size_t)n_centres + 3 * (size_t)n_points + (size_t)n_weights * n_centres;
    const size_t part_doubles = (size_t)splits * n_weights * n_points;
    HGMM_TRY(ensure(c, c->gt_buf, sizeof(double) * (in_doubles + part_doubles)));
    double* d_c = c->gt_buf.as<double>();
    double* d_p = d_c + 3 * (size_t)n_centres;
    double* d_w = d_p + 3 * (size_t)n_points;
    double* d_part = d_w + (size_t)n_weights * n_centres;
    HGMM_HIP(c, hipMemcpyAsync(d_c, centres, sizeof(double) * 3 * n_centres, hipMemcpyHostToDevice, c->stream));
    HGMM_HIP(c, hipMemcpyAsync(d_p, points, sizeof(double) * 3 * n_points, hipMemcpyHostToDevice, c->stream));
    HGMM_HIP(c, hipMemcpyAsync(d_w, weights, sizeof(double) * (size_t)n_weights * n_centres, hipMemcpyHostToDevice,
                               c->stream));
    const dim3 grid(pblocks, splits);
    const double inv_h2 = 1.0 / (h * h);
#define GT_CASE(NW)                                                                                         \
    case NW:                                                                                                \
        gauss_transform_kernel<NW><<<grid, GT_BLOCK, 0, c->stream>>>(d_c, n_centres, d_p, n_points, d_w,     \
                                                                     inv_h2, per_split, d_part);            \
        break
    switch (n_weights) {
        GT_CASE(1); GT_CASE(2); GT_CASE(3); GT_CASE(4); GT_CASE(5); GT_CASE(6); GT_CASE(7); GT_CASE(8);
    }
#undef GT_CASE
    HGMM_HIP(c, hipGetLastError());
    std::vector<double> part(part_doubles);
    HGMM_HIP(c, hipMemcpyAsync(part.data(), d_part, sizeof(double) * part_doubles, hipMemcpyDeviceToHost, c->stream));
    HGMM_HIP(c, ctx_stream_sync(c));
    const size_t row = (size_t)n_weights * n_points;
    for (size_t e = 0; e < row; ++e) {
        double s = 0.0;
        for (int sp = 0; sp < splits; ++sp) s += part[(size_t)sp * row + e];
        out[e] = s;
    }
    return HGMM_OK;
}
