// Internal context of libhgmm_hip.so (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <string>
#include <cstring>
#include <vector>

#include "../../include/hgmm.h"

namespace hgmm {

constexpr float FLAT_EPS = 1e-8f;     // gmm_waymo gmm_impl.py:15 / gmmreg_gpu gmm_impl.py:16
constexpr int FLAT_NSTAT = 7;         // s0, a[3], b[3]
constexpr int FLAT_MAX_J = 1024;      // single-pass kernels (all parameters of a lane in registers)
constexpr int FLAT_MAX_J_CHUNKED = 16384;  // chunked path (832-component chunks)
constexpr int FLAT_MAX_BLOCKS = 2048; // persistent grid upper bound (partials buffer)

// A growable device buffer owned by the context.
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

struct EventPair { hipEvent_t a, b; int kernel; };

struct FlatState {
    int cov_type = 0, variant = 0, J = 0, Jpad = 0;
    float tol = 0.f;
    int lls_cap = 0;
    int launched = 0;                 // EM iterations enqueued since train_begin
    bool active = false;
    bool chunked = false;             // J > FLAT_MAX_J: 832-component chunks
    int nchunks = 1;
    int last_kernel = 0;              // what this context enqueued last: 1 materialising E-step, 2 M-step from resp, 3 fused EM
    bool idle_since_launch = true;    // the host has drained the stream since the last of these launches (ctx_stream_sync)
};

// Online control of the store pacer's target rate (flat_kernels.hip, pace_target / pace_observe): the last few paced
// E-step launches are bracketed by event pairs that are looked at -- never waited for -- when the next one is launched.
struct PaceCtl {
    static constexpr int RING = 4;
    double target = 0.0;               // GB/s offered by the next paced launch; 0: not initialised
    int strikes = 0;                   // consecutive launches that ran longer than their target explains
    int J = 0;                         // the row length the state belongs to (another J starts over)
    int steps_down = 0;                // how often the target was lowered (reported by hgmm_pace_info)
    int steps_up = 0;                  // probes upwards that held
    // probing upwards: after `probe_after` clean launches in a row the next ones are offered 2 % more; three clean ones
    // make that the new target, a single long one ends the probe, caps the rate below it and doubles `probe_after`
    double ceiling = 1e30;             // a rate that was seen to congest: not probed again until it is forgotten
    int since_ceiling = 0;             // clean launches since the ceiling was learnt (forgotten after PACE_FORGET_AFTER)
    int ceilings_forgotten = 0;
    double probe_base = 0.0;           // > 0: a probe is running, this is the rate to fall back to
    int clean = 0, probe_after = 0, probe_seen = 0;
    hipEvent_t ev[RING][2] = {};
    double tgt_at[RING] = {}, bytes_at[RING] = {};
    unsigned head = 0, tail = 0;
    bool have_events = false;
    // what a launch is judged by: wall-clock stamps the kernel's workgroups leave in pinned host memory -- [RING][2][STAMP_WG]:
    // start and end per workgroup -- so that nothing the host, the queue or a profiler puts around the launch counts as
    // the memory system's answer (the event pair only says WHEN the stamps are complete; it is the fall-back measure if
    // the buffer could not be had)
    static constexpr int STAMP_WG = 1024;
    unsigned long long* stamps = nullptr;       // host address
    unsigned long long* stamps_dev = nullptr;   // the same memory as the device sees it
    int grid_at[RING] = {};
};

// Per-context options.  Each starts from the environment variable HGMM_<NAME IN CAPITALS>, read ONCE -- in hgmm_create, the
// only getenv of the library -- and can be set per context afterwards (hgmm_config_set / Context.config_set).  Every option
// selects a path that is also reached by data (another J, another cloud size, a failed factorisation) or is a documented
// operating mode; each is listed in INTEGRATION.md and run against its golden by tests/test_config_gpu.py.  Round 5 had
// 46 environment switches read at call time; the tuning knobs among them are fixed at their measured values now and
// the rejected variants are gone with their code.
enum ConfigKey {
    CFG_ESTEP_TARGET_GBS,      // store pacing of the materialising E-step: -1 controlled (default), 0 un-paced, > 0 fixed rate in GB/s
    CFG_PACE_START,            // rate the controller starts from, GB/s
    CFG_PACE_FORGET,           // clean launches after which a ceiling learnt under congestion is forgotten
    CFG_PREDICT_SINGLE_ROW,    // predict() on the general single-row kernel (the path of layouts the 4-row kernel is not built for)
    CFG_TREE_NO_CHOL,          // symmetric instead of triangular form of the pdfs' exponent (the fallback of a failed factorisation)
    CFG_TREE_REL,              // relative reach test of the level log-likelihood (large clouds; gives up the bitwise q for ~3 %)
    CFG_TREE_AHEAD,            // iterations the host keeps enqueued ahead of the device in hgmm_tree_build (0: batches + control-word copy)
    CFG_TREE_TICKETS,          // stop rule in the log-likelihood's last workgroup instead of the next launch
    CFG_TREE_OVERLAP,          // small clouds: iteration e + 1's E-step rides in iteration e's log-likelihood launch
    CFG_FULLCOV_TWO_PASS,      // two-kernel form of the full-covariance iteration (the path of J > 1024)
    CFG_KMPP_TWO_LAUNCHES,     // k-means++ step and tail as two launches (the path of > 16.7 M points)
    CFG_KMEANS_ACC_REGS,       // Lloyd sums in registers instead of LDS tables (the path of k > 1024)
    CFG_IPC_TIMEOUT_S,         // seconds the peer exchange waits for a peer's flag before it reports the collective as failed
    CFG_REG_DEVICE_SOLVE,      // registration loop without the host: 6 x 6 solve, twist, stop rule and the next encoding on the device
    CFG_COUNT
};
struct ConfigSpec { const char* name; int dflt, lo, hi; };
extern const ConfigSpec CONFIG_SPECS[CFG_COUNT];

struct HostComm;                           // hgmm_api.hip
struct IpcComm;                            // hgmm_api.hip: one-shot peer-to-peer exchange over mapped peer memory

struct TreeState {
    int L = 0;
    int T = 0;
    bool nodes_ready = false;
    bool pdf_f32 = false;             // hgmm_tree_set_precision: the level log-likelihood's pdfs in float32 (large clouds)
    double mu_rmax = -1.0;            // largest |mu_j| of the node table (< 0: not known on the host yet)
    bool momq_dirty = true;           // the fixed-point moment words hold sums nobody has cleared yet
    unsigned long long reg_seq = 0;   // sequence number of the last registration system handed over in pinned memory
    unsigned long long surplus_iterations = 0;   // (communicator) level-iterations enqueued behind a level's stop, all builds
};

// A FOREST: B independent clouds whose trees are built / whose targets are registered by the same launches
// (tree_batch.hip: hgmm_tree_build_batch, hgmm_tree_set_targets_batch, hgmm_tree_register_batch)
struct ForestState {
    int B = 0, L = 0, T = 0;
    bool nodes_ready = false;
    std::vector<int64_t> counts;          // points per cloud (they lie back to back in the context's resident cloud)
    std::vector<double> mu_rmax;          // largest |mu_j| per tree (extent bound of the fixed-point moments)
    int tg_B = 0;
    std::vector<int64_t> tg_counts, tg_first;
    std::vector<double> tg_rmax;          // largest |x| per target
    int64_t tg_pad = 0;
    bool momq_clean = false;              // every word of fr_momq is zero
    unsigned long long seq = 0;           // sequence number of the last hand-over through pinned memory
    void* host = nullptr;                 // pinned, coherent: progress words of the build, hand-over of the registration
    size_t host_cap = 0;
};

}  // namespace hgmm

// A point cloud resident in HBM that outlives the calls made on it (hgmm_points_create_*): the float32 row-major copy the
// flat EM reads and the float64 structure-of-arrays copy of the HGMM / KMeans kernels.  Any number per context.
struct hgmm_points {
    hgmm_ctx* ctx = nullptr;
    int64_t n = 0, n_pad = 0;
    hgmm::DevBuf x_aos, x_soa64;
};

struct hgmm_ctx {
    int cfg[hgmm::CFG_COUNT] = {};    // hgmm::ConfigKey -> value (hgmm_create: defaults, then the environment)
    int device = 0;
    int cus = 256;
    int wall_khz = 0;                 // rate of wall_clock64() on this device (StorePacer, flat_kernels.hip)
    hipStream_t stream = nullptr;
    std::string err;

    // ---- points -----------------------------------------------------------------
    int64_t n = 0;                    // local points
    // x_aos / x_soa64 are VIEWS of the cloud the kernels work on: the context's own one (own_points, hgmm_set_points_*) or
    // a caller-held handle (hgmm_points_bind); the memory belongs to whoever `bound` names
    hgmm::DevBuf x_aos;               // float [n,3]  (flat EM; wave-uniform scalar loads)
    hgmm::DevBuf x_soa64;             // double [3][n_pad] (HGMM; lanes across points)
    hgmm_points own_points;           // the one-cloud shortcut's storage
    hgmm_points* bound = nullptr;     // nullptr: nothing resident yet
    bool have_f32 = false, have_f64 = false;
    int64_t n_pad = 0;

    // ---- flat EM ----------------------------------------------------------------
    hgmm::FlatState flat;
    hgmm::PaceCtl pace;
    hgmm::DevBuf f_block;                     // float [10][Jpad]: the allocation behind the four arrays below
    hgmm::DevBuf f_mu, f_cov, f_w, f_inv;     // float model parameters (reference layout): NON-OWNING slices of f_block
    hgmm::DevBuf f_pack;                      // float [PK_ROWS = 8][Jpad] packed E-step params (flat_kernels.hip: mu, g, c, w)
    hgmm::DevBuf f_partials;                  // float [blocks][7][Jpad]
    hgmm::DevBuf f_lpn_partials;              // double [blocks]
    hgmm::DevBuf f_stats;                     // double [7*Jpad + 2]  (+ sum lpn, + n)
    hgmm::DevBuf f_lls;                       // float [cap]
    hgmm::DevBuf f_ctl;                       // int  [4]: done, n_iter, converged, pad ; float prev
    hgmm::DevBuf f_hint;                      // float [3][Jpad] centre hint for m-step
    hgmm::DevBuf f_cm, f_cs, f_ca, f_lpn2;    // chunked path: per-chunk (max, sum, arg-max) [C][n], lpn2 [n]
    hgmm::DevBuf scratch;
    double* h_scalars = nullptr;              // pinned, device-visible scalars (hgmm_host_scalars)
    int h_scalars_n = 0;
    hipEvent_t ev_slots[64] = {};             // hgmm_event_record / hgmm_event_wait
    // pinned host ring for small parameter uploads / result downloads (flat_kernels.hip: stage_*): a pageable
    // hipMemcpyAsync is staged by the runtime behind the stream's pending work, a pinned one is a plain DMA packet
    void* h_stage = nullptr;
    size_t h_stage_cap = 0, h_stage_off = 0;

    // ---- tree -------------------------------------------------------------------
    hgmm::TreeState tree;
    hgmm::DevBuf t_pi, t_mu, t_cov;           // double node tables [T], [T,3], [T,9]
    hgmm::DevBuf t_prep;                      // double [T][12] : inv(6) coef logc pi mu(3) -> see tree_kernels
    hgmm::DevBuf t_cplx;                      // double [T]
    void* tree_hctl = nullptr;                // pinned host copy of the level control words (2 slots)
    hipEvent_t tree_ev[2] = {nullptr, nullptr};
    hgmm::DevBuf exp_tab2;                    // double [2048] 2^(j/2048) for the throughput kernels' exp (tree_kernels.hip)
    hgmm::DevBuf t_tickets;                   // uint: two-level arrival counters of the last-workgroup reductions, 4 KB apart
    hgmm::DevBuf t_flags;                     // int [4] tree flags + uint64 executed-pair counter (tree_kernels.hip)
    hgmm::DevBuf t_mom;                       // double [T][10]
    hgmm::DevBuf t_momq;                      // uint64 [T][10]  registration E-step: fixed-point moments
    hgmm::DevBuf t_parent, t_current;         // int32 [n]
    hgmm::DevBuf t_perm;                      // int32 [n]  points sorted by parent
    hgmm::DevBuf t_xs3;                       // double [3][n_pad] third coordinate buffer (L > 2)
    hgmm::DevBuf t_seg;                       // int32 segment tables
    hgmm::DevBuf t_chunks;                    // int32 chunk descriptors
    hgmm::DevBuf t_partials;                  // double per-chunk partial moments
    hgmm::DevBuf t_q;                         // double per-block q partials + result
    hgmm::DevBuf t_llp;                       // double [node chunks][n_pad] log-likelihood partial sums
    hgmm::DevBuf t_qtrace;                    // double [max iterations per level] q of the running level
    hgmm::DevBuf tgt_soa64;                   // double [3][m_pad] registration target
    int64_t tgt_n = 0, tgt_pad = 0;
    double tgt_rmax = 0.0;                    // largest |x| of the target (extent bound of the fixed-point moments)

    // ---- forest (batched trees: tree_batch.hip) ------------------------------------
    hgmm::ForestState forest;
    hgmm::DevBuf fr_pi, fr_mu, fr_cov, fr_prep, fr_mom;   // node tables [B T], as t_pi ... t_mom
    hgmm::DevBuf fr_clouds;                   // ForestCloud [B] + int flags [B]
    hgmm::DevBuf fr_q;                        // double: the clouds' shares of q
    hgmm::DevBuf fr_trace;                    // double [B][L][trace_cap]
    hgmm::DevBuf fr_tg;                       // double [3][tg_pad] the targets, back to back
    hgmm::DevBuf fr_momq;                     // uint64 [B T][4] registration sums
    hgmm::DevBuf ff_clocks;                   // int64 [8][4] phase clocks of the one-pass full-covariance kernels (armed: ff_clocks_on)
    bool ff_clocks_on = false;
    hgmm::DevBuf ff_origin;                   // double [256][3] partial coordinate sums: origin of the float32 full-covariance statistics
    hgmm::DevBuf fr_reg;                      // per-pair registration table + the 28 numbers per pair + per-pair r2max words

    // ---- KMeans initialiser (float64, on x_soa64) -----------------------------------
    hgmm::DevBuf km_closest;                  // double [n_pad] k-means++: min squared distance so far
    hgmm::DevBuf km_block;                    // double block sums / prefix / candidate partials
    hgmm::DevBuf km_centres;                  // double [k][3] + padded [k][4]
    hgmm::DevBuf km_ids;                      // int64 [k] chosen point ids + candidates
    hgmm::DevBuf km_rand;                     // double [(k-1) * trials] host-drawn uniforms
    hgmm::DevBuf km_labels;                   // int32 [n_pad]
    hgmm::DevBuf km_mind2;                    // double [n_pad]
    hgmm::DevBuf km_partial;                  // double [blocks][k_alloc][4]
    hgmm::DevBuf km_out;                      // double [4k + 2] sums, inertia, changed (+ scratch)
    int64_t km_labels_n = -1;

    // ---- L2 GMMReg Gauss transform ---------------------------------------------------
    hgmm::DevBuf gt_buf;                      // double: centres, points, weights, per-split partial sums

    // ---- multi-GPU --------------------------------------------------------------
    ncclComm_t comm = nullptr;                // RCCL over xGMI (one GPU per rank)
    hgmm::HostComm* hcomm = nullptr;          // host shared-memory communicator (tests: ranks may share a GPU)
    hgmm::IpcComm* icomm = nullptr;           // one-shot exchange: every rank writes its slice into every peer's buffer (xGMI)
    volatile unsigned* icomm_err = nullptr;   // pinned host word an exchange kernel raises when a peer never arrived
    int nranks = 1, rank = 0;
    hgmm::DevBuf comm_buf;
    bool comm_on() const { return comm != nullptr || hcomm != nullptr || icomm != nullptr; }
    unsigned long long collectives = 0;       // all-reduces this context has enqueued on its communicator (hgmm_comm_stats)

    // ---- profiling --------------------------------------------------------------
    bool profiling = false;
    std::vector<hgmm::EventPair> events;
    size_t events_used = 0;
    double prof_ms[HGMM_K_COUNT] = {0};
    int64_t prof_n[HGMM_K_COUNT] = {0};
};

namespace hgmm {
// a region of the context's pinned staging ring (flat_kernels.hip); reused only after a stream synchronisation
int stage_reserve(hgmm_ctx* c, size_t bytes, void** out);
constexpr size_t STAGE_RING_BYTES = 4u << 20;

}  // namespace hgmm

// Every wait for the context's stream goes through here: the E-step's grid policy (flat_kernels.hip, estep_rows_grid)
// wants to know whether the chip has been idle since the last bandwidth-bound launch.
inline hipError_t ctx_stream_sync(hgmm_ctx* c) {
    const hipError_t e = hipStreamSynchronize(c->stream);
    c->flat.idle_since_launch = true;
    // an exchange kernel that waited in vain for a peer's slice has produced garbage: every result read after this
    // synchronisation would be wrong, so the wait itself fails (hipErrorLaunchTimeOut -> HGMM_ERR_HIP at the call site)
    if (e == hipSuccess && c->icomm_err && *c->icomm_err) return hipErrorLaunchTimeOut;
    return e;
}


namespace hgmm {

// Results of an API call on their way to the caller's (pageable) arrays: every add() enqueues ONE asynchronous DMA into
// the pinned ring behind the kernels already on the stream (large arrays, or whatever no longer fits half the ring, go
// straight to their destination), finish() synchronises once and hands the ring's regions out with plain memcpys.
// (Copied straight into pageable memory each array is staged by the runtime and waited for in turn, ~17 us apiece.)
struct StagedDownloads {
    hgmm_ctx* c;
    struct Pending { void* dst; const void* src; size_t bytes; };
    std::vector<Pending> pending;
    size_t staged = 0;
    hipError_t e = hipSuccess;
    explicit StagedDownloads(hgmm_ctx* ctx) : c(ctx) {}
    void add(void* dst, const void* dev_src, size_t bytes) {
        if (e != hipSuccess || !dst || bytes == 0) return;
        void* st = nullptr;
        if (bytes <= (256u << 10) && staged + bytes + 256 <= STAGE_RING_BYTES / 2 && stage_reserve(c, bytes, &st) == 0) {
            staged += (bytes + 255) & ~(size_t)255;
            e = hipMemcpyAsync(st, dev_src, bytes, hipMemcpyDeviceToHost, c->stream);
            pending.push_back({dst, st, bytes});
        } else {
            e = hipMemcpyAsync(dst, dev_src, bytes, hipMemcpyDeviceToHost, c->stream);
        }
    }
    // (hipSuccess, or the first error of a copy / of the synchronisation)
    hipError_t finish() {
        // always drain the stream, also after a failed add(): copies enqueued before the failure may still be in flight
        // into the caller's memory (a stack array in kmeans_kernels.hip), and the ring is only free once they are done
        const hipError_t es = ctx_stream_sync(c);
        if (e == hipSuccess) e = es;
        if (e == hipSuccess)
            for (const Pending& pd : pending) std::memcpy(pd.dst, pd.src, pd.bytes);
        if (es == hipSuccess) c->h_stage_off = 0;          // the stream is idle: every region of the ring is free
        pending.clear();
        return e;
    }
};

inline int fail(hgmm_ctx* c, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->err = buf;
    return code;
}

#define HGMM_HIP(ctx, call)                                                              \
    do {                                                                                 \
        hipError_t e_ = (call);                                                          \
        if (e_ != hipSuccess)                                                            \
            return hgmm::fail((ctx), HGMM_ERR_HIP, "%s failed: %s (%s:%d)", #call,       \
                              hipGetErrorString(e_), __FILE__, __LINE__);                \
    } while (0)

#define HGMM_NCCL(ctx, call)                                                             \
    do {                                                                                 \
        ncclResult_t r_ = (call);                                                        \
        if (r_ != ncclSuccess)                                                           \
            return hgmm::fail((ctx), HGMM_ERR_RCCL, "%s failed: %s (%s:%d)", #call,      \
                              ncclGetErrorString(r_), __FILE__, __LINE__);               \
    } while (0)

// First statement of every extern "C" entry that takes a context: the current HIP device is a property of the calling
// THREAD, so a caller that drives two contexts from one thread, or one context from several, gets the context's own
// device for every allocation, event and launch the entry makes (include/hgmm.h: "distinct contexts may be driven from
// distinct threads").  hipSetDevice on the device that is already current is a thread-local compare.
#define HGMM_ENTER(ctx)                                                                  \
    do {                                                                                 \
        if (!(ctx)) return HGMM_ERR_ARG;                                                 \
        HGMM_HIP((ctx), hipSetDevice((ctx)->device));                                    \
    } while (0)

#define HGMM_TRY(expr)                   \
    do {                                 \
        int rc_ = (expr);                \
        if (rc_ != HGMM_OK) return rc_;  \
    } while (0)

int ensure(hgmm_ctx* c, DevBuf& b, size_t bytes);

// profiling brackets: record an event pair around one kernel launch when enabled
struct ProfScope {
    hgmm_ctx* c;
    EventPair* ev = nullptr;
    ProfScope(hgmm_ctx* ctx, int kernel);
    ~ProfScope();
};
int profile_collect(hgmm_ctx* c);

// sub-system entry points implemented in flat_kernels.hip / tree_kernels.hip
int allreduce_f64_dev(hgmm_ctx* c, double* dev, size_t n);
int allreduce_f64_oop(hgmm_ctx* c, const double* src, double* dst, size_t n);
int allreduce_i64_dev(hgmm_ctx* c, long long* dev, size_t n);      // exact (integer) sum, in place

}  // namespace hgmm
