/*
 * hgmm.h -- C ABI of the MI355X-native (gfx950) GMM / hierarchical-GMM EM engine.
 *
 * This is the drop-in boundary for the EM hot path of
 * somanshu25/GPU-Accelerated-Point-Cloud-Registration-Using-Hierarchical-GMM.
 * The reference has no FFI seam of its own: its seam is Python duck typing
 * (`cupy.get_array_module(X)` in src/python/gmm_waymo/src/gmm_impl.py:19,31,54,68 and the
 * single host->device hop in src/python/gmm_waymo/src/gmm.py:72-80).  Each entry point
 * below names the reference function(s) it replaces (paths relative to the upstream
 * repository root).  INTEGRATION.md shows the ctypes binding a maintainer adds.
 *
 * Conventions
 *   - plain C types only; every function returns 0 on success or a negative hgmm_status.
 *     hgmm_last_error(ctx) returns a human-readable description of the last failure.
 *   - "host" pointers are borrowed for the duration of the call; "dev" pointers are device
 *     addresses obtained from hgmm_alloc() (or any hipMalloc'd buffer on ctx's device).
 *   - one context per device / per rank, no global state; a context is not re-entrant,
 *     distinct contexts may be driven from distinct threads.
 *   - all kernels run on the context's own HIP stream; calls that return host data
 *     synchronise that stream, all others are asynchronous.
 *   - cov_type: 0 = diag ([J,3] per-axis), 1 = spherical ([J]);
 *     variant : 0 = "W" (gmm_waymo/src/gmm_impl.py), 1 = "G" (gmmreg_gpu/gmm_impl.py)
 *     -- the two flavours differ in eps / regularisation, see SURVEY.md 8(a).
 *   - `inv_std` is what the reference calls `inv_cov` (it is 1/sigma, not 1/sigma^2).
 */
#ifndef HGMM_H
#define HGMM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct hgmm_ctx hgmm_ctx;

enum hgmm_status {
    HGMM_OK = 0,
    HGMM_ERR_HIP = -1,        /* a HIP runtime call failed */
    HGMM_ERR_ARG = -2,        /* invalid argument / unsupported size */
    HGMM_ERR_STATE = -3,      /* call sequence error (e.g. no points set) */
    HGMM_ERR_RCCL = -4,       /* an RCCL call failed */
    HGMM_ERR_NODEVICE = -5    /* no usable gfx950 device */
};

enum { HGMM_COV_DIAG = 0, HGMM_COV_SPHERICAL = 1 };
enum { HGMM_VARIANT_W = 0, HGMM_VARIANT_G = 1 };

/* kernels that the in-library profiler times with hipEvents (hgmm_profile_*) */
enum hgmm_kernel_id {
    HGMM_K_FLAT_ESTEP = 0,    /* materialising E-step (writes log_resp[N,J]) */
    HGMM_K_FLAT_FUSED = 1,    /* fused E+M sufficient-statistics kernel      */
    HGMM_K_FLAT_MSTEP = 2,    /* M-step moments from materialised resp       */
    HGMM_K_TREE_ESTEP = 3,    /* HGMM level E-step (8 children / point)      */
    HGMM_K_TREE_LOGLIK = 4,   /* HGMM level log-likelihood (all level nodes) */
    HGMM_K_TREE_REG = 5,      /* HGMM registration E-step (tree descent)     */
    HGMM_K_UTIL_FILL = 6,     /* hgmm_util_fill_f32 (HBM write-ceiling probe) */
    HGMM_K_FULL_PASS = 7,     /* full-cov flat EM: denominators + arg-max + log-likelihood */
    HGMM_K_FULL_MOMENTS = 8,  /* full-cov flat EM: fp64-MFMA sufficient statistics */
    HGMM_K_KMEANS_ASSIGN = 9, /* KMeans initialiser: nearest-centre assignment */
    HGMM_K_KMEANS_ACCUM = 10, /* KMeans initialiser: per-cluster sums */
    HGMM_K_ALLREDUCE = 11,    /* the sufficient-statistics all-reduce (RCCL / host backend), N > 1 only */
    HGMM_K_FULL_FUSED = 12,   /* full-cov flat EM, one pass: denominators + arg-max + q + fp64-MFMA statistics */
    HGMM_K_COUNT = 13
};

/* ---- lifecycle ------------------------------------------------------------------ */
int hgmm_version(void);
int hgmm_device_count(int* count);
int hgmm_create(int device_id, hgmm_ctx** out);
int hgmm_destroy(hgmm_ctx* ctx);
const char* hgmm_last_error(const hgmm_ctx* ctx);  /* ctx may be NULL: last create() error */
int hgmm_device_info(hgmm_ctx* ctx, char* name, int name_len, int* compute_units,
                     int64_t* hbm_bytes);
int hgmm_synchronize(hgmm_ctx* ctx);

/* ---- device memory (what cupy.asarray / cupy.asnumpy did: gmm.py:73-80, 95) ------ */
int hgmm_alloc(hgmm_ctx* ctx, size_t bytes, void** dev_out);
int hgmm_free(hgmm_ctx* ctx, void* dev);
int hgmm_h2d(hgmm_ctx* ctx, void* dev_dst, const void* host_src, size_t bytes);
int hgmm_d2h(hgmm_ctx* ctx, void* host_dst, const void* dev_src, size_t bytes);
/* device -> device on the context's stream, nothing waits (cupy.ndarray.copy() / astype(copy=True)) */
int hgmm_d2d(hgmm_ctx* ctx, void* dev_dst, const void* dev_src, size_t bytes);
/* Host-visible scalars: `count` doubles (<= 4096) of pinned host memory owned by the context that kernels write
 * directly (dev_out is the address they use, host_out the one the host reads).  With an event recorded behind the
 * producing kernel a value is read without draining the stream: what a 0-d CuPy array is to the reference's loops
 * (`lower_bound = log_prob_norm`, gmm_impl.py:136, read by `abs(change) < tol` while the M-step is still running).
 * The same array is returned on every call (count only grows it before its first use). */
int hgmm_host_scalars(hgmm_ctx* ctx, int count, double** host_out, double** dev_out);
/* 64 event slots on the context's stream: record marks the work enqueued so far, wait blocks the host until that
 * work (and only that) is done. */
#define HGMM_EVENT_SLOTS 64
int hgmm_event_record(hgmm_ctx* ctx, int slot);
int hgmm_event_wait(hgmm_ctx* ctx, int slot);
/* Elementwise float32 arithmetic on device arrays, enqueued on the context's stream (IEEE: bitwise NumPy's results
 * for + - * / sqrt): out[i] = a[i] op (dev_b ? dev_b[i] : scalar); RSUB / RDIV take the operands the other way round;
 * SQRT / EXP / LOG ignore the second operand.  out may alias a or b.  This is the `inv_cov = 1 / (xp.sqrt(covariances
 * + 1e-6) + eps)` of a caller's own EM loop (gmm_impl.py:134) without a trip to the host. */
enum hgmm_elementwise_op {
    HGMM_EW_ADD = 0, HGMM_EW_SUB = 1, HGMM_EW_RSUB = 2, HGMM_EW_MUL = 3, HGMM_EW_DIV = 4, HGMM_EW_RDIV = 5,
    HGMM_EW_SQRT = 6, HGMM_EW_EXP = 7, HGMM_EW_LOG = 8, HGMM_EW_MAX = 9, HGMM_EW_MIN = 10
};
int hgmm_elementwise_f32(hgmm_ctx* ctx, int op, int64_t n, const float* dev_a, const float* dev_b, float scalar,
                         float* dev_out);

/* ---- point cloud ------------------------------------------------------------------
 * Replaces `dev_X = cupy.asarray(X.astype(np.float32))` (gmm_waymo/src/gmm.py:73) and
 * `cuda.to_device(points)` (hgmm/hgmm_gpu.py:513).  xyz is host row-major [n,3]. */
int hgmm_set_points_f32(hgmm_ctx* ctx, const float* xyz, int64_t n);
int hgmm_set_points_f64(hgmm_ctx* ctx, const double* xyz, int64_t n);
int64_t hgmm_num_points(const hgmm_ctx* ctx);
/* Several resident clouds per context -- what any number of `cupy.asarray(frame)` arrays are to the reference (the H2D
 * copy of every frame in gmm_waymo/src/waymoutils.py, `dev_X` in gmm.py:73): a handle owns its device copy (float32
 * rows + float64 structure of arrays, like hgmm_set_points_*), hgmm_points_bind makes it the cloud every following
 * call of the context works on -- a pointer swap: nothing is copied, nothing waits, kernels already enqueued keep the
 * cloud they were launched on.  bind(ctx, NULL) goes back to the cloud of hgmm_set_points_* (the one-cloud shortcut,
 * which stays).  Destroying the bound handle leaves nothing bound.  Handles must be destroyed before their context. */
typedef struct hgmm_points hgmm_points;
int hgmm_points_create_f32(hgmm_ctx* ctx, const float* xyz, int64_t n, hgmm_points** out);
int hgmm_points_create_f64(hgmm_ctx* ctx, const double* xyz, int64_t n, hgmm_points** out);
int hgmm_points_bind(hgmm_ctx* ctx, hgmm_points* points);
int hgmm_points_destroy(hgmm_ctx* ctx, hgmm_points* points);
int64_t hgmm_points_count(const hgmm_points* points);
/* The float32 rows [n,3] of a resident cloud back on the host (`cupy.asnumpy(dev_X)`): what a caller needs to draw the
 * reference's initial parameters (init_gmm_params samples the HOST array, gmm_waymo/src/gmm_impl.py:26-41) for a cloud it
 * only holds as a handle.  points = NULL: the cloud the context is working on.  Synchronises the context's stream. */
int hgmm_points_download_f32(hgmm_ctx* ctx, const hgmm_points* points, float* xyz_out);

/* ---- flat GMM EM (diag / spherical) ------------------------------------------------
 * hgmm_flat_estep   <- e_step()            gmm_waymo gmm_impl.py:105-116, gmmreg_gpu gmm_impl.py:55-61
 *                      (+ estimate_log_prob[_spherical] 53-78)
 * hgmm_flat_predict <- predict()           gmm_waymo gmm_impl.py:147-155, gmmreg_gpu gmm_impl.py:88-91
 * hgmm_flat_mstep   <- m_step()            gmm_waymo gmm_impl.py:90-103 (+81-88), gmmreg_gpu gmm_impl.py:46-52
 * hgmm_flat_train   <- train_gmm()         gmm_waymo gmm_impl.py:118-145, gmmreg_gpu gmm_impl.py:63-85
 *
 * mu [J,3], inv_std / cov [J,3] (diag) or [J] (spherical), w [J]: host float32.
 * dev_log_resp [N,J] row-major, dev_lpn [N], dev_argmax [N]: device buffers or NULL.    */
int hgmm_flat_estep(hgmm_ctx* ctx, int cov_type, int variant, int J,
                    const float* mu, const float* inv_std, const float* w,
                    float* dev_log_resp, float* dev_lpn, int32_t* dev_argmax,
                    double* mean_lpn_out);
/* The same E-step without waiting for it: the kernels are enqueued on the context's stream, the host parameter
 * arrays have been copied to the context's pinned staging ring when the call returns (the caller may reuse them),
 * and the mean log-normaliser is written to the DEVICE double dev_mean_lpn (or NULL) by a one-workgroup kernel
 * behind the E-step.  This is what the reference's e_step() is under CuPy (gmm_impl.py:105-116 returns device
 * arrays, nothing synchronises): a following hgmm_flat_mstep overlaps its host work with this kernel. */
int hgmm_flat_estep_async(hgmm_ctx* ctx, int cov_type, int variant, int J,
                          const float* mu, const float* inv_std, const float* w,
                          float* dev_log_resp, float* dev_lpn, int32_t* dev_argmax, double* dev_mean_lpn);
/* E-step / M-step with the PARAMETERS in device arrays (dense, the host layouts: mu [J,3], inv_std / cov [J,3] or
 * [J], w [J]) -- what the reference's functions are when they are handed CuPy arrays (`xp = cupy.get_array_module(X)`,
 * gmm_impl.py:91, 106): nothing is staged, nothing is downloaded, nothing waits.  hgmm_flat_mstep_dev writes the new
 * parameters to the caller's device arrays; dev_centre_hint may be NULL.  dev_mean_lpn may be a host-visible scalar
 * (hgmm_host_scalars). */
int hgmm_flat_estep_dev(hgmm_ctx* ctx, int cov_type, int variant, int J,
                        const float* dev_mu, const float* dev_inv_std, const float* dev_w,
                        float* dev_log_resp, float* dev_lpn, int32_t* dev_argmax, double* dev_mean_lpn);
int hgmm_flat_mstep_dev(hgmm_ctx* ctx, int cov_type, int variant, int J,
                        const float* dev_resp, int is_log, const float* dev_centre_hint,
                        float* dev_w, float* dev_mu, float* dev_cov);
int hgmm_flat_predict_dev(hgmm_ctx* ctx, int cov_type, int variant, int J,
                          const float* dev_mu, const float* dev_inv_std, const float* dev_w,
                          int32_t* dev_labels);
/* The store pacer of the two N x J writers (hgmm_flat_estep*, hgmm_flat_log_prob): *target_gbs_out = the rate the next
 * paced launch offers its rows at (GB/s; 0 = un-paced), *steps_down_out / *steps_up_out = how often the context's
 * controller has lowered it / raised it for good since the context was created.  The rate starts just below the write
 * path's congestion knee; three launches in a row that run more than 6 % longer than the rate explains lower it by 2 %;
 * after a run of clean launches it is raised by 2 % on probation (one long launch takes that back and caps the rate).
 * Measurement aid, no reference counterpart. */
int hgmm_pace_info(hgmm_ctx* ctx, double* target_gbs_out, int* steps_down_out, int* steps_up_out);
/* Forget what the controller has learnt (rate, ceiling, counters): the next paced launch starts at the initial rate again.
 * A rate that congested once is remembered as a ceiling -- rightly while its cause lasts (another stream writing, a hot
 * chip), wrongly for ever: the controller forgets a ceiling by itself after 10 000 clean launches below it
 * (HGMM_PACE_FORGET=<n>); a caller that knows the cause is gone says so here.  Measurement aid, no reference counterpart. */
int hgmm_pace_reset(hgmm_ctx* ctx);
/* Un-normalised per-pair log-densities log N(x_i; mu_j, diag) -> dev_log_prob [N,J]
 * (estimate_log_prob / estimate_log_prob_spherical, gmm_waymo gmm_impl.py:53-78). */
int hgmm_flat_log_prob(hgmm_ctx* ctx, int cov_type, int J, const float* mu, const float* inv_std,
                       float* dev_log_prob);
int hgmm_flat_predict(hgmm_ctx* ctx, int cov_type, int variant, int J,
                      const float* mu, const float* inv_std, const float* w,
                      int32_t* dev_labels);
/* dev_resp: [N,J] responsibilities (is_log = 0) or log-responsibilities (is_log = 1, the
 * exp() of train_gmm's `m_step(X, xp.exp(log_resp))` is then fused into the read).
 * centre_hint [J,3] (host, may be NULL): moments are accumulated about it to avoid the
 * raw-second-moment cancellation; results are mathematically independent of it.        */
int hgmm_flat_mstep(hgmm_ctx* ctx, int cov_type, int variant, int J,
                    const float* dev_resp, int is_log, const float* centre_hint,
                    float* w_out, float* mu_out, float* cov_out);
/* Whole loop device-resident (fused E+M, no N x J traffic).  mu/cov/w: in = initial,
 * out = final.  lls has room for max_iter floats.                                      */
int hgmm_flat_train(hgmm_ctx* ctx, int cov_type, int variant, int J, int max_iter, float tol,
                    float* mu, float* cov, float* w, float* inv_std_out,
                    float* lls_out, int* n_iter_out, int* converged_out);
/* The same loop split so a caller (bench, streaming refit) can time / interleave steps:
 * begin uploads the initial parameters, step enqueues `iters` EM iterations without
 * synchronising, end synchronises and downloads.                                       */
int hgmm_flat_train_begin(hgmm_ctx* ctx, int cov_type, int variant, int J, float tol,
                          const float* mu, const float* cov, const float* w, int lls_capacity);
int hgmm_flat_train_step(hgmm_ctx* ctx, int iters);
int hgmm_flat_train_end(hgmm_ctx* ctx, float* mu, float* cov, float* w, float* inv_std_out,
                        float* lls_out, int* n_iter_out, int* converged_out);
/* Sufficient statistics of ONE fused E+M pass on this context's points (after the
 * all-reduce when a communicator is attached): stats [J,7] doubles laid out
 * (s0, a_x, a_y, a_z, b_x, b_y, b_z) with a = sum r (x - mu_old), b = sum r (x - mu_old)^2,
 * plus sum_i lpn_i and the (global) point count.                                        */
int hgmm_flat_stats(hgmm_ctx* ctx, int cov_type, int variant, int J,
                    const float* mu, const float* inv_std, const float* w,
                    double* stats_out, double* sum_lpn_out, double* n_points_out);

/* All-reduces this context has enqueued on its communicator so far, and the level-iterations hgmm_tree_build has enqueued
 * behind a level's stop (each costs two all-reduces on unchanged operands): under a communicator every rank must issue
 * the same collectives, so each rank tops its queue up to min(iterations + tree_ahead, budget) per level -- 2 surplus
 * iterations per level by default (round 5: up to 15).  Either pointer may be NULL.                                    */
int hgmm_comm_stats(hgmm_ctx* ctx, unsigned long long* collectives_out, unsigned long long* surplus_tree_iterations_out);

/* ---- per-context options ---------------------------------------------------------------------------------------------
 * The library reads the environment ONCE, in hgmm_create: every option below starts from HGMM_<NAME IN CAPITALS> when
 * that variable is set, else from its default, and can be changed per context afterwards.  Each option selects a path
 * that data also reaches (another J, another cloud size, a failed factorisation) or a documented operating mode; the table
 * is in INTEGRATION.md and tests/test_config_gpu.py runs every option against its golden.  There is no reference
 * counterpart: the reference has no configuration beyond its call arguments.
 *   estep_target_gbs (-1)   store pacing of e_step()'s kernel: -1 controlled, 0 un-paced, > 0 fixed rate in GB/s
 *   pace_start (6600), pace_forget (10000)   the pacing controller's start rate / clean launches until a learnt ceiling is forgotten
 *   predict_single_row (0)  predict() on the general single-row kernel
 *   tree_no_chol (0), tree_rel (0)   symmetric form of the tree pdfs' exponent / relative reach test of the level log-likelihood
 *   tree_ahead (2), tree_tickets (0), tree_overlap (1)   hgmm_tree_build's host look-ahead, where its stop rule runs, E-step
 *                           of iteration e + 1 inside iteration e's log-likelihood launch
 *   fullcov_two_pass (0)    two-kernel form of hgmm_fullcov_fit's iteration (the path of J > 1024)
 *   kmpp_two_launches (0), kmeans_acc_regs (0)   the KMeans initialiser's paths for > 16.7 M points / k > 1024
 *   ipc_timeout_s (20)      seconds the peer exchange waits for a peer before the collective is reported as failed        */
int hgmm_config_count(void);
const char* hgmm_config_name(int index);                      /* NULL beyond hgmm_config_count() - 1 */
int hgmm_config_set(hgmm_ctx* ctx, const char* name, int value);
int hgmm_config_get(hgmm_ctx* ctx, const char* name, int* value_out);

/* ---- hierarchical GMM (8-ary tree, full 3x3 covariance, float64) -------------------
 * hgmm_tree_build     <- buildGMMTree()     hgmm/hgmm_cupy_cpu_working.py:122-160 (CPU twin,
 *                        canonical) == hgmm/hgmm_gpu.py:466-548 (kernels 107-115, 387-426)
 * hgmm_tree_set_nodes <- GMMTree._nodes     hgmm_cupy_cpu_working.py:334,341
 * hgmm_tree_reg_estep <- gmmTreeRegESTep()  hgmm_cupy_cpu_working.py:202-228 == hgmm_gpu.py:550-577
 *
 * T = 8 (8^L - 1) / 7 nodes.  init_mu [T,3] host float64 (RNG stays with the caller).
 * Outputs pi [T], mu [T,3], cov [T,3,3] host float64; leaf_idx [N] = currentIdx after the
 * last level (may be NULL); iters_per_level [L]; q_trace[q_capacity] (may be NULL).     */
int hgmm_tree_build(hgmm_ctx* ctx, int L, double ls, double ld, const double* init_mu,
                    double sig2, int max_iters_per_level,
                    double* pi_out, double* mu_out, double* cov_out, int32_t* leaf_idx_out,
                    int32_t* iters_per_level_out, double* q_trace_out, int q_capacity,
                    int* q_len_out);
int hgmm_tree_set_nodes(hgmm_ctx* ctx, int L, const double* pi, const double* mu,
                        const double* cov);
/* Arithmetic type of the pdf evaluations behind hgmm_tree_build's STOP RULE.  The reference has both: its CPU twin is
 * float64 throughout (hgmm_cupy_cpu_working.py; the default here and the parity reference), its GPU file float32
 * throughout (`points.astype(np.float32)`, float32 node and moment arrays: hgmm/hgmm_gpu.py:472, 478-484).
 * HGMM_PRECISION_F32_PDF: the level log-likelihood q = sum_i log max(sum_j pi_j N(x_i; j), eps) (logLikelihoodValue,
 * hgmm_gpu.py:107-115) evaluates its N x 8^(l+1) Gaussians in packed float32 on coordinates relative to the workgroup's
 * first point (formed in float64), with log() and the sum over the points in float64 -- on clouds of any size and in
 * hgmm_tree_build_batch (round 6; round 5: clouds of >= 400 000 points only).  Level 0 of a small cloud or a forest needs
 * no such kernel in either mode: its q comes out of the float64 E-step's own eight terms.  The E-step, the moments and
 * the M-step stay float64: a level that stops after the same number of iterations yields the float64 tree bit for bit; q
 * itself differs by ~1e-7 relative (|dq| < 1 % of the smallest stop threshold in use on every cloud tried,
 * tests/test_tree_gpu.py, tests/test_tree_batch_gpu.py).
 * The flat FULL-covariance fit (hgmm_fullcov_fit / hgmm_fullcov_estep, J <= 1024) under the same setting takes its
 * float32-TILE kernel (round 6): pdfs from head + tail differences in packed float32, the 16-point tile of un-normalised
 * responsibilities in float32, the statistics' products on v_mfma_f32_16x16x4_f32 about the cloud's centroid (float
 * partials per 256 points, added in float64); row sums, 1 / den, log() and the M-step float64.  1.69 -> 1.21 ms per
 * launch at N = 10^6, J = 800; against the float64 fit after 3 iterations there: pi 2e-6, mu 2e-7, Sigma 7e-6 of a
 * component's variance, 2 of 10^6 labels (tests/test_fullcov_gpu.py).  float64 stays the default and the parity reference.
 * Stays in force for the context until set again. */
enum { HGMM_PRECISION_F64 = 0, HGMM_PRECISION_F32_PDF = 1 };
int hgmm_tree_set_precision(hgmm_ctx* ctx, int precision);
/* Registration target cloud (host [n,3] float64), kept resident across iterations. */
int hgmm_tree_set_target(hgmm_ctx* ctx, const double* xyz, int64_t n);
/* E-step of the registration loop on the resident target transformed on the fly by
 * x' = scale * R x + t (R row-major [3,3]; pass NULL for identity).  Outputs host
 * float64 m0 [T], m1 [T,3], m2 [T,3,3].                                                 */
int hgmm_tree_reg_estep(hgmm_ctx* ctx, const double* rot, const double* t, double scale,
                        double lambda_c, double* m0_out, double* m1_out, double* m2_out);
/* E-step + normal equations of one registration iteration (GMMTree.expectation_step + the least-squares
 * system of GMMTree.maximization_step, hgmm/hgmm_gpu.py:722-752), entirely on the device: out28 (host) =
 * the 21 upper-triangle entries (row-major) of A^T A, the 6 of A^T b, and b^T b of the reference's stacked
 * twist system  [ s_i x n | n ] x = n . (mu_i - s_i); the caller solves the 6 x 6 system and composes the
 * twist (hgmm_gpu.py:620-664), residual q = b^T b - x . A^T b.  Deterministic (fixed-point moment sums). */
int hgmm_tree_reg_normal(hgmm_ctx* ctx, const double* rot, const double* t, double scale,
                         double lambda_c, double* out28);
/* The registration loop of GMMTree.registration (src/python/hgmm/hgmm_gpu.py:754-768) run by the library: per
 * iteration hgmm_tree_reg_normal on the device, then on the host the 6 x 6 solve, q = max(b'b - x.A'b, 0), the
 * composition of the twist with (rot, t) (twist_mul, hgmm_gpu.py:634-664) and the reference's stop rule
 * |q - q_prev| < tol.  rot [9] row-major and t [3] are updated in place; *q_prev_inout: NaN = no previous q.
 * status: 0 = max_iter iterations done, 1 = stopped by tol, 2 = the normal equations of the NEXT iteration are too
 * ill-conditioned (or not finite): the caller runs that iteration with the reference's stacked least squares
 * (eigh + QR on the host) and may call again.  trace (optional): per iteration rot, t, q.
 * Per-context option reg_device_solve = 1 (hgmm_config_set; off by default -- the host solve is the parity reference):
 * the whole loop runs on the device -- Cholesky solve of the 6 x 6 system, twist, stop rule and the fixed-point encoding
 * of the next E-step in one device thread behind the normal equations, the host only follows a progress word.  Same
 * trajectory to ~1e-12 (device sin / cos, fused multiply-adds, Cholesky instead of pivoted elimination); status 2 is
 * decided on the Cholesky pivots (smallest pivot <= 1e-11 largest diagonal entry).  Not under a communicator.      */
int hgmm_tree_register(hgmm_ctx* ctx, double* rot, double* t, double scale, double lambda_c, int max_iter, double tol,
                       double* q_prev_inout, int* iters_out, int* status_out, double* trace);
/* ---- batched HGMM: B independent scan pairs per launch set ("forest") ------------------------------------------------
 * The reference's unit of work is ONE pair -- registration_gmmtree(source, target) (hgmm/hgmm_gpu.py:802-807) =
 * buildGMMTree(source) (hgmm_gpu.py:466-548) + GMMTree.registration(target) (hgmm_gpu.py:754-768, E-step 550-577) -- a
 * chain of ~350 small launches for a 40 k-point scan.  These entries run B such pairs through the SAME launches; every
 * pair's tree, iteration counts, q trace and (R, t) are bitwise those of hgmm_tree_build / hgmm_tree_register on that pair
 * alone (same device functions, same chunks and orders of summation, same host steps).  No communicator.
 *
 * hgmm_set_points_batch_f64   B host arrays [counts[b],3] become ONE resident cloud, cloud after cloud (float64 view only:
 *                             the flat EM entry points refuse it)
 * hgmm_tree_build_batch       <- B x buildGMMTree on the resident cloud's B consecutive pieces (each < 400 000 points);
 *                             init_mu [B][T][3]; outputs (each may be NULL) pi [B][T], mu [B][T][3], cov [B][T][3][3],
 *                             iters [B][L], q_trace [B][q_capacity] + q_len [B] (levels back to back, as hgmm_tree_build).
 *                             The levels run in lock-step: a level lasts as long as its slowest cloud.
 * hgmm_tree_get_nodes_batch   tree b of the resident forest (host tables, as hgmm_tree_build returns them)
 * hgmm_tree_set_targets_batch <- B x hgmm_tree_set_target
 * hgmm_tree_register_batch    <- B x hgmm_tree_register on (tree b, target b): rot [B][9], t [B][3], q_prev [B] (NaN: none)
 *                             updated in place; iters [B]; status [B] as hgmm_tree_register's (a pair that meets status 2
 *                             leaves the batch at that iteration: the caller finishes it through the serial entries);
 *                             trace (optional) [B][max_iter][13].  The 6 x 6 solves stay on the host unless the context's
 *                             reg_device_solve option is set (see hgmm_tree_register): then one device thread per pair.    */
int hgmm_set_points_batch_f64(hgmm_ctx* ctx, int B, const double* const* xyz, const int64_t* counts);
/* ... and from float32 rows (the scans' files, `points.astype(np.float32)` of hgmm/hgmm_gpu.py:472): widened to float64 on
 * the device -- exactly, so every result equals the float64 entry's on the widened arrays -- half the bytes per upload and
 * no conversion pass on the host.  hgmm_tree_set_targets_batch_f32 likewise.                                           */
int hgmm_set_points_batch_f32(hgmm_ctx* ctx, int B, const float* const* xyz, const int64_t* counts);
int hgmm_tree_build_batch(hgmm_ctx* ctx, int B, const int64_t* counts, int L, double ls, double ld,
                          const double* init_mu, double sig2, int max_iters_per_level,
                          double* pi_out, double* mu_out, double* cov_out, int32_t* iters_out,
                          double* q_trace_out, int q_capacity, int32_t* q_len_out);
int hgmm_tree_get_nodes_batch(hgmm_ctx* ctx, int b, double* pi_out, double* mu_out, double* cov_out);
int hgmm_tree_set_targets_batch(hgmm_ctx* ctx, int B, const double* const* xyz, const int64_t* counts);
int hgmm_tree_set_targets_batch_f32(hgmm_ctx* ctx, int B, const float* const* xyz, const int64_t* counts);
int hgmm_tree_register_batch(hgmm_ctx* ctx, int B, double* rot, double* t, double scale, double lambda_c,
                             int max_iter, double tol, double* q_prev_inout, int32_t* iters_out,
                             int32_t* status_out, double* trace);
/* The steps buildGMMTree is made of, one at a time (reference function granularity).  Node tables
 * hold T nodes (any T >= 8, need not be a complete tree).
 * hgmm_tree_estep  <- gmmTreeEStep()       hgmm_cupy_cpu_working.py:162-191: parent_idx[N] arbitrary
 *                     (-1 = root), returns moments of ALL nodes + currentIdx[N]
 * hgmm_tree_mstep  <- gmmTreeMStep()       hgmm_cupy_cpu_working.py:193-198: ML update (with the
 *                     m0 < ld rule) of nodes [j_begin, j_end) in place
 * hgmm_tree_loglik <- logLikelihoodValue() hgmm_cupy_cpu_working.py:72-85 over nodes [j_begin, j_end) */
int hgmm_tree_estep(hgmm_ctx* ctx, int64_t T, const double* pi, const double* mu, const double* cov,
                    const int32_t* parent_idx, double* m0_out, double* m1_out, double* m2_out,
                    int32_t* current_idx_out);
int hgmm_tree_mstep(hgmm_ctx* ctx, int64_t T, const double* m0, const double* m1, const double* m2,
                    int64_t j_begin, int64_t j_end, double n_points, double ld, double* pi_inout,
                    double* mu_inout, double* cov_inout);
int hgmm_tree_loglik(hgmm_ctx* ctx, int64_t T, const double* pi, const double* mu, const double* cov,
                     int64_t j_begin, int64_t j_end, double* q_out);
/* Work actually done by the level log-likelihood kernels since the last hgmm_tree_build / hgmm_tree_set_nodes /
 * stand-alone step started (measurement aid for bench.py, no reference counterpart): *pairs_out = number of
 * (point, node) pairs whose pdf was evaluated -- the reference's logLikelihoodValue (hgmm_gpu.py:107-115) visits all
 * N x 8^(l+1); nodes with pi < eps contribute exactly 0 there and nodes whose pdf underflows to 0.0 for every point
 * of a workgroup are rejected per workgroup here, so the count is smaller --, *flags_out bit 0 = some node's
 * Sigma^-1 was not numerically positive definite (the kernels then evaluate the symmetric quadratic form). */
int hgmm_tree_stats(hgmm_ctx* ctx, unsigned long long* pairs_out, int* flags_out);
/* smallest eigenvalue / trace per node (complexity(), hgmm_cupy_cpu_working.py:87-91) */
int hgmm_tree_node_complexity(hgmm_ctx* ctx, double* cplx_out);

/* ---- flat GMM EM, FULL 3x3 covariance (float64) --------------------------------------
 * The reference's Python has no flat full-covariance EM; its in-scope definition
 * is ONE tree level with branching J: the CPU twin run with its module global n_node = J and
 * maxTreeLevel = 1 (hgmm_cupy_cpu_working.py:30, 62-198): pi0 = 1/J, cov0 = sig2 I,
 * gamma = pi N / sum, gammas < 1e-15 dropped, m0 < ld -> (pi = 0, mu = 0, cov = I), stop when
 * |q - q_prev| < ls.  Statistics are the 10 floats per cluster (m0, m1[3], unique m2[6]) that
 * the multi-GPU all-reduce exchanges; they are contracted on the fp64 matrix cores.
 * hgmm_fullcov_estep returns one E-step's moments in the reference layout m0[J], m1[J,3],
 * m2[J,3,3] + hard labels + the level log-likelihood q of the given parameters.            */
int hgmm_fullcov_fit(hgmm_ctx* ctx, int J, double ls, double ld, const double* init_mu, double sig2,
                     int max_iters, double* pi_out, double* mu_out, double* cov_out,
                     int32_t* labels_out, double* q_trace_out, int q_capacity, int* q_len_out);
int hgmm_fullcov_estep(hgmm_ctx* ctx, int J, const double* pi, const double* mu, const double* cov,
                       double* m0_out, double* m1_out, double* m2_out, int32_t* labels_out,
                       double* q_out);

/* ---- KMeans initialiser (float64, on the resident cloud) -------------------------------
 * Replaces the scikit-learn call the GMMReg flavour seeds its EM with
 * (gmmreg_gpu/gmm_impl.py:18-24: KMeans(n_clusters=k, random_state=1, max_iter=50, n_init=1).fit(X),
 * X = the caller's float64 points, gmm.py:79).  The host side (kmeans.py) draws the random
 * numbers, applies the stop rule and relocates empty clusters exactly as scikit-learn does; the
 * two O(N k) parts run on the device:
 * hgmm_kmeans_plusplus  greedy k-means++ seeding.  first_id = the first centre (host-drawn),
 *                       rand_vals[(k-1) * n_trials] = the uniforms of every later step in draw
 *                       order; per step: candidates = searchsorted(cumsum(closest), rand * pot),
 *                       keep the candidate with the lowest potential (first minimum).
 *                       ids_out[k] = chosen point indices, centers_out[k,3] their coordinates.
 * hgmm_kmeans_step      one Lloyd assignment with the given centres: nearest centre per point
 *                       (first minimum), sums_out[k,4] = (sum x, sum y, sum z, count) per cluster,
 *                       inertia = sum of squared distances to the nearest centre, n_changed =
 *                       labels that differ from the previous step's (reset_labels != 0: from -1).
 *                       With a communicator attached the three results are all-reduced.
 * hgmm_kmeans_labels    labels[n] / squared distance to the assigned centre of the last step. */
/* Host-side (no device work): column means / variances / centred copy of an [n,3] float64 cloud, bit for bit what
 * NumPy's X.mean(axis=0), np.var(X, axis=0), X - mean give on a C-ordered array -- the preprocessing of
 * sklearn.cluster.KMeans.fit behind src/python/gmmreg_gpu/gmm_impl.py:20-21. */
int hgmm_kmeans_center_f64(const double* x_host, int64_t n, double* mean3, double* var3, double* xc_out);
int hgmm_kmeans_plusplus(hgmm_ctx* ctx, int k, int64_t first_id, const double* rand_vals, int n_trials,
                         int64_t* ids_out, double* centers_out);
int hgmm_kmeans_step(hgmm_ctx* ctx, int k, const double* centers, int reset_labels, double* sums_out,
                     double* inertia_out, int64_t* n_changed_out);
int hgmm_kmeans_labels(hgmm_ctx* ctx, int32_t* labels_out, double* min_dist2_out);
/* hgmm_kmeans_lloyd     the whole Lloyd loop on the device (single rank): per iteration assignment, sums,
 *                       centres = sums * (1 / count), summed squared centre shift and scikit-learn's stop
 *                       rules in its order (unchanged labels -> strict; shift <= tol_abs; max_iter), several
 *                       iterations enqueued per host synchronisation.  centers_inout: in = start, out = the
 *                       centres after n_iter completed iterations.  An iteration that leaves a cluster empty
 *                       is handed back unfinished (needs_host = 1, its sums[k,4] / changed-label count in
 *                       sums_out / n_changed_out, centres untouched) for the host's relocation rule.        */
int hgmm_kmeans_lloyd(hgmm_ctx* ctx, int k, double* centers_inout, int max_iter, double tol_abs,
                      int reset_labels, int* n_iter_out, int* strict_out, int* needs_host_out,
                      double* sums_out, int64_t* n_changed_out);

/* ---- L2 GMMReg: Gauss transform (float64) ---------------------------------------------
 * out[k, i] = sum_j weights[k, j] exp(-|points_i - centres_j|^2 / h^2),  k < n_weights <= 8
 * = GaussTransform(centres, h).compute(points, weights) of gmmreg_gpu/transforms.py:43-86, which
 * cost_functions.py:29-40 evaluates with the target means as centres, the transformed source means
 * as points, h = sqrt(2) sigma and the weight rows phi_t / z, phi_t mu_t / z (one call = the value
 * and the gradient of the L2 cost).  Host arrays in and out (mixture sizes).                    */
int hgmm_gauss_transform(hgmm_ctx* ctx, const double* centres, int n_centres, const double* points,
                         int n_points, const double* weights, int n_weights, double h, double* out);

/* ---- multi-GPU: one context per rank, RCCL over xGMI --------------------------------
 * New functionality (the reference is single-GPU).  With a communicator attached,
 * hgmm_flat_train*, hgmm_flat_stats and hgmm_tree_build all-reduce their per-cluster
 * sufficient statistics (and the point count) across ranks before every M-step.        */
int hgmm_comm_unique_id(void* id128_out);                       /* 128 bytes */
int hgmm_comm_init_rank(hgmm_ctx* ctx, int nranks, int rank, const void* id128);
int hgmm_comm_destroy(hgmm_ctx* ctx);
/* Test backend behind the same all-reduce call sites: the ranks are processes of ONE machine that
 * meet in the POSIX shared-memory object `name` (they may share a GPU, which RCCL does not allow), every
 * all-reduce goes device -> host -> summed in rank order -> device.  For exercising the N > 1 path on a
 * single-GPU box; not a performance path.                                                          */
/* `name` (here and in hgmm_comm_init_ipc) must be unique to the JOB -- rank 0 creates the object, the others join the
 * object of that name: give it a token the ranks agreed on beforehand (bench.py: a random token of rank 0, handed out
 * over the TCP star; tests: the pid of the launching process).  Joiners refuse an object whose creator is gone (pid
 * in the same pid namespace) or that is older than 10 minutes in a foreign one, but a crashed earlier job's object
 * under the SAME name whose creator's pid has been recycled cannot be told apart from the live one.                  */
int hgmm_comm_init_host(hgmm_ctx* ctx, int nranks, int rank, const char* name);
/* One-shot exchange backend behind the same all-reduce call sites (the message is 57 KB, latency is
 * everything): the ranks are processes of ONE node, one GPU each.  Every rank owns an exchange buffer in uncached
 * device memory, exported with hipIpcGetMemHandle and mapped by all peers (the handles meet in the POSIX shared-memory
 * object `name`).  An all-reduce is ONE kernel per rank: each workgroup writes its 4 KB piece of the rank's slice
 * into the slot this rank owns in EVERY peer's buffer (plain stores over xGMI), releases a sequence flag per piece and
 * peer, waits for the same piece of every peer in its own buffer and adds the slices up in RANK ORDER -- so all ranks
 * hold bitwise the same sum, with no ring, no tree and no proxy thread.  Slots are double-buffered by the parity of
 * the collective's sequence number (stream order makes that enough).  A peer that never arrives raises an error word
 * after ~20 s instead of hanging the GPU: the next synchronising call fails.  Two ranks may share one GPU (tests). */
int hgmm_comm_init_ipc(hgmm_ctx* ctx, int nranks, int rank, const char* name);
int hgmm_comm_allreduce_f64(hgmm_ctx* ctx, double* host_inout, int n, int op /*0 sum,1 max*/);

/* ---- profiling (hipEvent pairs around the hot kernels, on the context's stream) ---- */
int hgmm_profile_enable(hgmm_ctx* ctx, int on);
int hgmm_profile_reset(hgmm_ctx* ctx);
int hgmm_profile_get(hgmm_ctx* ctx, int kernel_id, double* total_ms_out, int64_t* launches_out);
/* Phase clocks of the one-pass full-covariance kernels (a profiling aid like the event profiler above; no reference
 * counterpart): enable = 1 arms them -- the following hgmm_fullcov_estep / hgmm_fullcov_fit launches record, per wave of
 * workgroup 7, the clock64() cycles spent in phase A (pdfs -> tile), phase B (row sums, arg-max), phase C (statistics on
 * the matrix cores) and waiting at the phases' barriers; enable = 0 disarms them and copies the last launch's
 * clocks_out[8 waves][4] (A, B, C, wait).  profiles/r06/fullcov_phase_clocks.log is made with it.                      */
int hgmm_fullcov_phase_clocks(hgmm_ctx* ctx, int enable, int64_t* clocks_out);
/* Streams `value` into n float32 of a device buffer with 16-byte (optionally non-temporal)
 * stores: the pure-write HBM ceiling the E-step's log_resp stream is compared with. */
int hgmm_util_fill_f32(hgmm_ctx* ctx, float* dev, int64_t n, float value, int nontemporal);

#ifdef __cplusplus
}
#endif
#endif /* HGMM_H */
