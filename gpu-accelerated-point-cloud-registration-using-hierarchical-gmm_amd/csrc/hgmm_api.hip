// Context lifecycle, device memory, point upload, RCCL attachment and the hipEvent profiler
// behind the C ABI of include/hgmm.h.
#include "hgmm_ctx.h"
#include "wave_ops.h"

#include <fcntl.h>
#include <signal.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <cctype>
#include <cerrno>
#include <chrono>
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

namespace hgmm {

static std::string g_create_error;
static std::mutex g_mu;

int ensure(hgmm_ctx* c, DevBuf& b, size_t bytes) {
    if (bytes <= b.cap && b.p) return HGMM_OK;
    if (b.p) {
        HGMM_HIP(c, ctx_stream_sync(c));
        HGMM_HIP(c, hipFree(b.p));
        b.p = nullptr;
        b.cap = 0;
    }
    size_t want = bytes < 256 ? 256 : bytes;
    HGMM_HIP(c, hipMalloc(&b.p, want));
    b.cap = want;
    return HGMM_OK;
}

ProfScope::ProfScope(hgmm_ctx* ctx, int kernel) : c(ctx) {
    if (!c->profiling) return;
    if (c->events_used == c->events.size()) {
        EventPair p;
        p.kernel = kernel;
        if (hipEventCreate(&p.a) != hipSuccess || hipEventCreate(&p.b) != hipSuccess) return;
        c->events.push_back(p);
    }
    ev = &c->events[c->events_used++];
    ev->kernel = kernel;
    (void)hipEventRecord(ev->a, c->stream);
}
ProfScope::~ProfScope() {
    if (ev) (void)hipEventRecord(ev->b, c->stream);
}

int profile_collect(hgmm_ctx* c) {
    if (c->events_used == 0) return HGMM_OK;
    HGMM_HIP(c, ctx_stream_sync(c));
    for (size_t i = 0; i < c->events_used; ++i) {
        float ms = 0.f;
        EventPair& p = c->events[i];
        if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            c->prof_ms[p.kernel] += (double)ms;
            c->prof_n[p.kernel] += 1;
        }
    }
    c->events_used = 0;
    return HGMM_OK;
}

// ---- host shared-memory communicator ----------------------------------------------------------
// A second backend behind the same all-reduce call sites, for exercising the N > 1 data path on a box
// with ONE GPU (RCCL refuses two ranks on one device): the ranks are processes of one machine, every
// all-reduce goes device -> POSIX shared memory -> summed in rank order -> device.  Deterministic,
// slow, not a performance path.
// Both shared-memory objects start with {ready, owner_pid}: see shm_attach.
struct HostCommShm {
    std::atomic<int> ready;
    std::atomic<int> owner_pid;
    unsigned long long owner_pidns, created_s;      // (ShmHeader)
    std::atomic<int> count;
    std::atomic<int> generation;
    int nranks;
    size_t slot_doubles;
};
struct HostComm {
    HostCommShm* shm = nullptr;
    double* data = nullptr;          // [nranks][slot_doubles]
    size_t bytes = 0;
    std::string name;
    bool owner = false;
    std::vector<double> tmp;
};
constexpr size_t HOSTCOMM_SLOT = 1u << 18;       // doubles per rank and round
constexpr int HOSTCOMM_TIMEOUT_S = 120;

static int hostcomm_barrier(hgmm_ctx* c) {
    HostCommShm* s = c->hcomm->shm;
    const int gen = s->generation.load(std::memory_order_acquire);
    if (s->count.fetch_add(1, std::memory_order_acq_rel) + 1 == s->nranks) {
        s->count.store(0, std::memory_order_relaxed);
        s->generation.store(gen + 1, std::memory_order_release);
        return HGMM_OK;
    }
    const auto t0 = std::chrono::steady_clock::now();
    while (s->generation.load(std::memory_order_acquire) == gen) {
        sched_yield();
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(HOSTCOMM_TIMEOUT_S))
            return fail(c, HGMM_ERR_STATE, "host communicator: barrier timed out (a peer rank is gone?)");
    }
    return HGMM_OK;
}

// in place on a DEVICE buffer (op 0 sum, 1 max), ranks combined in rank order
static int hostcomm_allreduce_dev(hgmm_ctx* c, double* dev, size_t n, int op) {
    HostComm* h = c->hcomm;
    const int R = h->shm->nranks;
    for (size_t off = 0; off < n; off += HOSTCOMM_SLOT) {
        const size_t cnt = std::min(HOSTCOMM_SLOT, n - off);
        double* mine = h->data + (size_t)c->rank * HOSTCOMM_SLOT;
        HGMM_HIP(c, hipMemcpyAsync(mine, dev + off, sizeof(double) * cnt, hipMemcpyDeviceToHost, c->stream));
        HGMM_HIP(c, ctx_stream_sync(c));
        HGMM_TRY(hostcomm_barrier(c));
        h->tmp.resize(cnt);
        if (op == 2) {                                      // the words are int64: exact sums
            const long long* src = reinterpret_cast<const long long*>(h->data);
            long long* dst = reinterpret_cast<long long*>(h->tmp.data());
            for (size_t i = 0; i < cnt; ++i) {
                long long acc = src[i];
                for (int r = 1; r < R; ++r) acc += src[(size_t)r * HOSTCOMM_SLOT + i];
                dst[i] = acc;
            }
        } else
        for (size_t i = 0; i < cnt; ++i) {
            double acc = h->data[i];
            for (int r = 1; r < R; ++r) {
                const double v = h->data[(size_t)r * HOSTCOMM_SLOT + i];
                acc = op == 1 ? (v > acc ? v : acc) : acc + v;
            }
            h->tmp[i] = acc;
        }
        HGMM_TRY(hostcomm_barrier(c));                     // everybody has read the slots
        HGMM_HIP(c, hipMemcpyAsync(dev + off, h->tmp.data(), sizeof(double) * cnt, hipMemcpyHostToDevice, c->stream));
        HGMM_HIP(c, ctx_stream_sync(c));
    }
    return HGMM_OK;
}

// The ranks of a job meet in a POSIX shared-memory object that rank 0 creates (unlink + O_EXCL) and the others open by
// name.  An object a CRASHED earlier run left under the same name must not be taken for this job's: its header says
// ready = 1 already, and a rank that opened it before rank 0 replaced it would publish into an orphan and sit out the
// barrier's timeout.  So the header carries the creator's pid: a joining rank accepts an object only once it is ready
// AND its creator is alive, and otherwise lets go of it and opens the name again (rank 0's fresh object is zero-filled:
// not ready until rank 0 has initialised it).  The owner clears `ready` before it unlinks.
// (A pid only means something inside its PID namespace: ranks started in separate containers that share /dev/shm see each
//  other's objects but not each other's pids.  The header therefore also carries the creator's namespace and the time of
//  creation: a joiner from ANOTHER namespace cannot ask whether the creator lives and accepts an object younger than
//  SHM_FOREIGN_MAX_AGE_S instead.)
struct ShmHeader { std::atomic<int> ready; std::atomic<int> owner_pid; unsigned long long owner_pidns, created_s; };
constexpr unsigned long long SHM_FOREIGN_MAX_AGE_S = 600;
static unsigned long long my_pidns() {
    struct stat st;
    return stat("/proc/self/ns/pid", &st) == 0 ? (unsigned long long)st.st_ino : 0ull;
}
static unsigned long long now_s() {
    return (unsigned long long)std::chrono::duration_cast<std::chrono::seconds>(std::chrono::system_clock::now().time_since_epoch()).count();
}
static void* shm_attach(const std::string& name, size_t bytes, bool owner, int timeout_s) {
    if (owner) {
        shm_unlink(name.c_str());
        const int fd = shm_open(name.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd < 0) return nullptr;
        if (ftruncate(fd, (off_t)bytes) != 0) { close(fd); shm_unlink(name.c_str()); return nullptr; }
        void* p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        close(fd);
        if (p == MAP_FAILED) { shm_unlink(name.c_str()); return nullptr; }
        ShmHeader* h = static_cast<ShmHeader*>(p);
        h->owner_pid.store((int)getpid(), std::memory_order_relaxed);
        h->owner_pidns = my_pidns();
        h->created_s = now_s();
        return p;                                           // (the caller fills its fields in, then stores ready = 1)
    }
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        const int fd = shm_open(name.c_str(), O_RDWR, 0600);
        if (fd >= 0) {
            struct stat st;
            void* p = MAP_FAILED;
            if (fstat(fd, &st) == 0 && (size_t)st.st_size >= bytes)
                p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
            close(fd);
            if (p != MAP_FAILED) {
                ShmHeader* h = static_cast<ShmHeader*>(p);
                // a fresh object becomes ready within microseconds of its creation: give it a moment before re-opening
                for (int spin = 0; spin < 200 && h->ready.load(std::memory_order_acquire) != 1; ++spin) usleep(100);
                const int pid = h->owner_pid.load(std::memory_order_relaxed);
                const unsigned long long ns = my_pidns();
                const bool same_ns = ns != 0 && ns == h->owner_pidns;
                const bool alive = same_ns ? (pid > 0 && (kill((pid_t)pid, 0) == 0 || errno == EPERM))
                                           : (now_s() - h->created_s <= SHM_FOREIGN_MAX_AGE_S);
                if (h->ready.load(std::memory_order_acquire) == 1 && alive) return p;
                munmap(p, bytes);                           // an orphan, or not initialised yet: look at the name again
            }
        }
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(timeout_s)) return nullptr;
        usleep(1000);
    }
}

static void hostcomm_close(hgmm_ctx* c) {
    HostComm* h = c->hcomm;
    if (!h) return;
    if (h->shm && h->owner) h->shm->ready.store(0, std::memory_order_release);
    if (h->shm) munmap(h->shm, h->bytes);
    if (h->owner) shm_unlink(h->name.c_str());
    delete h;
    c->hcomm = nullptr;
}

// ---- one-shot peer exchange (hgmm_comm_init_ipc) ------------------------------------------------
// The sufficient-statistics all-reduce is 57 KB per EM iteration of 0.38 ms: latency is all that matters, and the
// topology is a full xGMI mesh.  So no ring and no tree: every rank WRITES its slice straight into a slot it owns in
// every peer's exchange buffer and every rank adds the R slices it received, in rank order -- one kernel per rank and
// collective, one trip over the links, bitwise the same sum on all ranks.
//
// Exchange buffer of a rank (uncached device memory, exported with hipIpcGetMemHandle, mapped by all peers):
//   flags [2][IPC_MAX_RANKS][IPC_MAX_CHUNKS] uint64   sequence number of the last collective whose piece has landed
//   slots [2][IPC_MAX_RANKS][IPC_SLOT]       double   the pieces; index = (parity of the sequence number, writer)
// Double buffering by parity is enough: rank A can only write collective k + 2 into B's parity-(k & 1) slots after it
// has consumed k + 1, i.e. after it has seen B's flags of k + 1 -- and B released those from a kernel that its stream
// started after B's kernel of collective k (the reader of those slots) had finished.
constexpr int IPC_MAX_RANKS = 8;
constexpr int IPC_CHUNK = 512;                         // doubles per workgroup: two per thread, 4 KB
constexpr size_t IPC_SLOT = (size_t)1 << 16;           // doubles per slot (512 KB); larger payloads go in pieces
constexpr int IPC_MAX_CHUNKS = (int)(IPC_SLOT / IPC_CHUNK);
constexpr size_t IPC_FLAG_BYTES = sizeof(unsigned long long) * 2 * IPC_MAX_RANKS * IPC_MAX_CHUNKS;
constexpr size_t IPC_BUF_BYTES = IPC_FLAG_BYTES + sizeof(double) * 2 * IPC_MAX_RANKS * IPC_SLOT;


struct IpcShm {
    std::atomic<int> ready;
    std::atomic<int> owner_pid;
    unsigned long long owner_pidns, created_s;      // (ShmHeader)
    std::atomic<int> count;
    std::atomic<int> generation;
    std::atomic<int> failed;                     // some rank could not map a peer: nobody keeps the communicator
    int nranks;
    hipIpcMemHandle_t handle[IPC_MAX_RANKS];
    int device[IPC_MAX_RANKS];
    int pid[IPC_MAX_RANKS];
};
struct IpcPeers {
    unsigned long long* flags[IPC_MAX_RANKS];
    double* slots[IPC_MAX_RANKS];
};
struct IpcComm {
    IpcShm* shm = nullptr;
    size_t shm_bytes = 0;
    std::string name;
    bool owner = false;
    void* local = nullptr;                       // this rank's exchange buffer
    void* mapped[IPC_MAX_RANKS] = {};            // every rank's buffer as this process sees it (mapped[rank] == local)
    IpcPeers peers = {};
    unsigned long long seq = 0;                  // collectives issued so far (the same number on every rank)
    bool in_step = false;                        // initialised together with the peers: teardown meets them in a barrier
    unsigned* err_host = nullptr;
    unsigned* err_dev = nullptr;
    long long timeout_ticks = 0;
};

static int ipc_barrier(hgmm_ctx* c) {
    IpcShm* s = c->icomm->shm;
    const int gen = s->generation.load(std::memory_order_acquire);
    if (s->count.fetch_add(1, std::memory_order_acq_rel) + 1 == s->nranks) {
        s->count.store(0, std::memory_order_relaxed);
        s->generation.store(gen + 1, std::memory_order_release);
        return HGMM_OK;
    }
    const auto t0 = std::chrono::steady_clock::now();
    while (s->generation.load(std::memory_order_acquire) == gen) {
        sched_yield();
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(HOSTCOMM_TIMEOUT_S))
            return fail(c, HGMM_ERR_STATE, "peer exchange: barrier timed out (a peer rank is gone?)");
    }
    return HGMM_OK;
}

__device__ __forceinline__ double ipc_combine(double a, double b, int op) {
    if (op == 1) return b > a ? b : a;
    if (op == 2) return __longlong_as_double(__double_as_longlong(a) + __double_as_longlong(b));   // exact int64 sum
    return a + b;
}

// One collective: blockIdx.x = piece of IPC_CHUNK values.  op: 0 sum, 1 max (float64), 2 sum of int64 words.
__global__ __launch_bounds__(256) void ipc_allreduce_kernel(IpcPeers pp, int R, int rank, const double* __restrict__ src,
                                                            double* __restrict__ dst, int n, int op,
                                                            unsigned long long seq, unsigned* err, long long timeout_ticks) {
    const int c = blockIdx.x, t = threadIdx.x;
    const int i0 = c * IPC_CHUNK + 2 * t;
    const int parity = (int)(seq & 1);
    const double v0 = i0 < n ? src[i0] : 0.0, v1 = i0 + 1 < n ? src[i0 + 1] : 0.0;
    // 1. this rank's piece into the slot it owns in every buffer (its own included: one code path, one summation order)
    const size_t mine = ((size_t)parity * IPC_MAX_RANKS + rank) * IPC_SLOT + i0;
    for (int p = 0; p < R; ++p) {
        typedef double d2 __attribute__((ext_vector_type(2)));
        *reinterpret_cast<d2*>(pp.slots[p] + mine) = d2{v0, v1};
    }
    __threadfence_system();                      // the piece has left for every peer ...
    __syncthreads();
    if (t < R) {
        // 2. ... before its flag does (release, system scope)
        __hip_atomic_store(pp.flags[t] + ((size_t)parity * IPC_MAX_RANKS + rank) * IPC_MAX_CHUNKS + c, seq,
                           __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        // 3. the same piece of peer t, in MY buffer
        const unsigned long long* f = pp.flags[rank] + ((size_t)parity * IPC_MAX_RANKS + t) * IPC_MAX_CHUNKS + c;
        const long long t0 = wall_clock64();
        // (once the error word is up -- an earlier collective waited in vain -- nobody waits again: the collectives
        //  enqueued behind it drain at once instead of sitting out the timeout one after the other)
        const bool broken = __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u;
        while (!broken && __hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < seq) {
            __builtin_amdgcn_s_sleep(2);
            if (wall_clock64() - t0 > timeout_ticks) {             // never hang the GPU: raise the error word, go on
                __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                break;
            }
        }
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");     // system scope: the pieces behind the flags
    // 4. the R slices in rank order
    double a0 = 0.0, a1 = 0.0;
    for (int p = 0; p < R; ++p) {
        const unsigned long long* q = reinterpret_cast<const unsigned long long*>(
            pp.slots[rank] + ((size_t)parity * IPC_MAX_RANKS + p) * IPC_SLOT + i0);
        const double w0 = __longlong_as_double((long long)__hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
        const double w1 = __longlong_as_double((long long)__hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
        a0 = p == 0 ? w0 : ipc_combine(a0, w0, op);
        a1 = p == 0 ? w1 : ipc_combine(a1, w1, op);
    }
    if (i0 < n) dst[i0] = a0;
    if (i0 + 1 < n) dst[i0 + 1] = a1;
}

static int ipc_allreduce(hgmm_ctx* c, const double* src, double* dst, size_t n, int op) {
    IpcComm* ic = c->icomm;
    for (size_t off = 0; off < n; off += IPC_SLOT) {
        const int cnt = (int)std::min(IPC_SLOT, n - off);
        const unsigned long long seq = ++ic->seq;
        ipc_allreduce_kernel<<<(cnt + IPC_CHUNK - 1) / IPC_CHUNK, 256, 0, c->stream>>>(
            ic->peers, c->nranks, c->rank, src + off, dst + off, cnt, op, seq, ic->err_dev, ic->timeout_ticks);
    }
    HGMM_HIP(c, hipGetLastError());
    return HGMM_OK;
}

static void ipc_close(hgmm_ctx* c) {
    IpcComm* ic = c->icomm;
    if (!ic) return;
    for (int p = 0; p < IPC_MAX_RANKS; ++p)
        if (ic->mapped[p] && ic->mapped[p] != ic->local) (void)hipIpcCloseMemHandle(ic->mapped[p]);
    if (ic->shm) {
        // nobody frees its buffer while a peer still has it mapped
        if (ic->in_step) (void)ipc_barrier(c);
        if (ic->owner) ic->shm->ready.store(0, std::memory_order_release);
        munmap(ic->shm, ic->shm_bytes);
    }
    if (ic->local) (void)hipFree(ic->local);
    if (ic->err_host) (void)hipHostFree(ic->err_host);
    if (ic->owner) shm_unlink(ic->name.c_str());
    c->icomm_err = nullptr;
    delete ic;
    c->icomm = nullptr;
}

int allreduce_f64_dev(hgmm_ctx* c, double* dev, size_t n) {
    if (!c->comm_on()) return HGMM_OK;
    c->collectives++;
    ProfScope prof(c, HGMM_K_ALLREDUCE);
    if (c->comm) {
        HGMM_NCCL(c, ncclAllReduce(dev, dev, n, ncclDouble, ncclSum, c->comm, c->stream));
        return HGMM_OK;
    }
    if (c->icomm) return ipc_allreduce(c, dev, dev, n, 0);
    if (c->hcomm) return hostcomm_allreduce_dev(c, dev, n, 0);
    return HGMM_OK;
}

int allreduce_i64_dev(hgmm_ctx* c, long long* dev, size_t n) {
    if (!c->comm_on()) return HGMM_OK;
    c->collectives++;
    ProfScope prof(c, HGMM_K_ALLREDUCE);
    if (c->comm) {
        HGMM_NCCL(c, ncclAllReduce(dev, dev, n, ncclInt64, ncclSum, c->comm, c->stream));
        return HGMM_OK;
    }
    if (c->icomm) return ipc_allreduce(c, reinterpret_cast<double*>(dev), reinterpret_cast<double*>(dev), n, 2);
    return hostcomm_allreduce_dev(c, reinterpret_cast<double*>(dev), n, 2);
}

// out of place: `src` keeps the rank's own values (callers that enqueue iterations past a device-side stop rely
// on that: the skipped kernels leave `src` untouched, so the repeated all-reduce reproduces the same `dst`)
int allreduce_f64_oop(hgmm_ctx* c, const double* src, double* dst, size_t n) {
    if (!c->comm_on()) {
        if (src != dst) HGMM_HIP(c, hipMemcpyAsync(dst, src, sizeof(double) * n, hipMemcpyDeviceToDevice, c->stream));
        return HGMM_OK;
    }
    c->collectives++;
    ProfScope prof(c, HGMM_K_ALLREDUCE);
    if (c->comm) {
        HGMM_NCCL(c, ncclAllReduce(src, dst, n, ncclDouble, ncclSum, c->comm, c->stream));
        return HGMM_OK;
    }
    if (c->icomm) return ipc_allreduce(c, src, dst, n, 0);
    if (src != dst) HGMM_HIP(c, hipMemcpyAsync(dst, src, sizeof(double) * n, hipMemcpyDeviceToDevice, c->stream));
    return hostcomm_allreduce_dev(c, dst, n, 0);
}

__global__ void aos_to_soa64_f32(const float* __restrict__ in, int64_t n, int64_t n_pad,
                                 double* __restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pad) return;
    for (int d = 0; d < 3; ++d) out[d * n_pad + i] = (i < n) ? (double)in[3 * i + d] : 0.0;
}
__global__ void aos_to_soa64_f64(const double* __restrict__ in, int64_t n, int64_t n_pad,
                                 double* __restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pad) return;
    for (int d = 0; d < 3; ++d) out[d * n_pad + i] = (i < n) ? in[3 * i + d] : 0.0;
}
__global__ void f64_to_f32(const double* __restrict__ in, int64_t n, float* __restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (float)in[i];
}

// HBM write-ceiling probe.  MODE 0: grid-stride, one 16-byte store per thread per step (every
// wave writes 1 KiB adjacent to its neighbours').  MODE 1: each workgroup owns one contiguous
// region and streams through it (the E-step's pattern: every wave a private sequential stream).
// MODE 2: like 0 with four independent 16-byte stores per thread per step.
// MODE 6 (paced): the E-step's store pattern without its arithmetic -- every wave writes runs of `run4` float4 (12.8 KB at
// J = 800: four rows), runs dealt round-robin over the waves, each run released by a StorePacer whose period makes all
// waves together offer `pace16`'s rate: what the write path takes when it is offered exactly that much.
template <bool NT>
__global__ __launch_bounds__(256) void util_fill_paced_kernel(float* __restrict__ p, int64_t n4, int run4, int pace16) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    const f4 one = {1.f, 1.f, 1.f, 1.f};
    f4* q = reinterpret_cast<f4*>(p);
    const int64_t nw = (int64_t)gridDim.x * 4, gw = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int64_t nruns = (n4 + run4 - 1) / run4;
    StorePacer pacer(pace16, gw, nw);
    for (int64_t r = gw; r < nruns; r += nw) {
        const int64_t lo = r * run4, hi = (lo + run4 < n4) ? lo + run4 : n4;
        pacer.wait();
        for (int64_t i = lo + lane; i < hi; i += 64) {
            if (NT) __builtin_nontemporal_store(one, q + i);
            else q[i] = one;
        }
    }
}

// MODE 7 (read probe): the M-step's access pattern without its arithmetic -- every wave READS runs of `run4` float4 (a
// row of J = 800: 200 float4), `depth` runs requested before the first is consumed, runs contiguous per wave or dealt
// round-robin; the sums keep the loads alive (one value per wave written at the end).  What the HBM read path delivers.
template <int DEPTH, bool NT>
__global__ __launch_bounds__(256) void util_read_kernel(const float* __restrict__ p, int64_t n4, int run4, int round_robin,
                                                        float* __restrict__ sink) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    const f4* q = reinterpret_cast<const f4*>(p);
    const int64_t nw = (int64_t)gridDim.x * 4, gw = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int64_t nruns = n4 / run4;
    const int64_t per = (nruns + nw - 1) / nw;
    const int64_t r_lo = round_robin ? gw : gw * per, r_step = round_robin ? nw : 1;
    const int64_t cnt = round_robin ? (gw < nruns ? (nruns - gw + nw - 1) / nw : 0)
                                    : (r_lo < nruns ? (r_lo + per < nruns ? per : nruns - r_lo) : 0);
    constexpr int SEG = 4;                                  // float4 per lane and run (run4 <= 256)
    f4 buf[DEPTH][SEG];
    auto load = [&](int64_t it, f4 (&dst)[SEG]) {
        const int64_t r = r_lo + (it < cnt ? it : cnt - 1) * r_step;
#pragma unroll
        for (int s = 0; s < SEG; ++s) {
            const int off = s * 64 + lane;
            const f4* src = q + r * run4 + (off < run4 ? off : run4 - 1);
            dst[s] = NT ? __builtin_nontemporal_load(src) : *src;
        }
    };
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    if (cnt > 0) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) load(d, buf[d]);
        for (int64_t it = 0; it < cnt; it += DEPTH) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                if (it + d < cnt) {
#pragma unroll
                    for (int s = 0; s < SEG; ++s) acc += buf[d][s];
                    load(it + d + DEPTH, buf[d]);
                }
            }
        }
    }
    const float t = acc.x + acc.y + acc.z + acc.w;
    if (t == 12345.678f) sink[gw] = t;                      // (never true for the probe's data: the loads stay, nothing is written)
}

template <bool NT, int MODE>
__global__ __launch_bounds__(256) void util_fill_kernel(float* __restrict__ p, int64_t n4, float v) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    const f4 val = {v, v, v, v};
    f4* q = reinterpret_cast<f4*>(p);
    auto st = [&](int64_t i) {
        if (NT) __builtin_nontemporal_store(val, q + i);
        else q[i] = val;
    };
    if (MODE == 0) {
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) st(i);
    } else if (MODE == 1) {
        const int64_t per = (n4 + gridDim.x - 1) / gridDim.x;
        const int64_t lo = (int64_t)blockIdx.x * per, hi = (lo + per < n4) ? lo + per : n4;
        for (int64_t i = lo + threadIdx.x; i < hi; i += blockDim.x) st(i);
    } else if (MODE == 2) {
        const int64_t stride = (int64_t)gridDim.x * blockDim.x;
        int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
        for (; i + 3 * stride < n4; i += 4 * stride) { st(i); st(i + stride); st(i + 2 * stride); st(i + 3 * stride); }
        for (; i < n4; i += stride) st(i);
    } else if (MODE == 5) {
        // MODE 5: rows of 200 float4 (J = 800); a workgroup takes `chunk` consecutive rows per step, workgroups
        // round-robin; thread t writes float4 t of every row (threads 200.. idle): wave q owns piece q of a row --
        // the store pattern of an E-step whose four waves split the COMPONENTS of a row
        const int rows = (int)v;
        const f4 one = {1.f, 1.f, 1.f, 1.f};
        const int64_t nrows = n4 / 200;
        for (int64_t r0 = (int64_t)blockIdx.x * rows; r0 < nrows; r0 += (int64_t)gridDim.x * rows) {
            for (int r = 0; r < rows; ++r) {
                if (threadIdx.x < 200 && r0 + r < nrows) {
                    if (NT) __builtin_nontemporal_store(one, q + (r0 + r) * 200 + threadIdx.x);
                    else q[(r0 + r) * 200 + threadIdx.x] = one;
                }
            }
        }
    } else {
        // MODE 3: one writer wavefront per workgroup streams `chunk` float4 (a batch of rows) at a
        // time, workgroups take chunks round-robin -- the store pattern of a producer/consumer
        // E-step with a dedicated writer wave per CU
        const int64_t chunk = (int64_t)v;              // chunk length in float4, passed through v
        const f4 one = {1.f, 1.f, 1.f, 1.f};
        f4* qq = reinterpret_cast<f4*>(p);
        const int64_t nchunks = (n4 + chunk - 1) / chunk;
        for (int64_t cidx = blockIdx.x; cidx < nchunks; cidx += gridDim.x) {
            const int64_t lo = cidx * chunk, hi = (lo + chunk < n4) ? lo + chunk : n4;
            for (int64_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
                if (NT) __builtin_nontemporal_store(one, qq + i);
                else qq[i] = one;
            }
        }
    }
}

}  // namespace hgmm

using namespace hgmm;

extern "C" int hgmm_util_fill_f32(hgmm_ctx* c, float* dev, int64_t n, float value, int nontemporal) {
    HGMM_ENTER(c);
    if (!c || !dev || n < 4) return HGMM_ERR_ARG;
    const int64_t n4 = n / 4;
    // bits 8.. of `nontemporal` select the probe variant: mode = (flags >> 8) & 3, grid = cus * ((flags >> 16) or 8)
    const int mode = (nontemporal >> 8) & 7;
    const int gmul = (nontemporal >> 16) ? (nontemporal >> 16) : 8;
    const bool nt = (nontemporal & 1) != 0;
    const int grid = c->cus * gmul;
    {
        ProfScope prof(c, HGMM_K_UTIL_FILL);
#define FILL(NTF, M) util_fill_kernel<NTF, M><<<grid, 256, 0, c->stream>>>(dev, n4, value)
        if (mode == 0) { if (nt) FILL(true, 0); else FILL(false, 0); }
        else if (mode == 1) { if (nt) FILL(true, 1); else FILL(false, 1); }
        else if (mode == 2) { if (nt) FILL(true, 2); else FILL(false, 2); }
        else if (mode == 6) {       // paced runs: value = offered GB/s (0: un-paced), grid = cus * gmul workgroups of 4 waves
            const int run4 = 800;                                   // 12.8 KB: four rows of J = 800
            int khz = c->wall_khz;
            if (khz <= 0) {
                if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, c->device) != hipSuccess || khz <= 0) khz = 100000;
                c->wall_khz = khz;
            }
            int pace16 = 0;
            if (value > 0.f) {
                const double seconds = 16.0 * run4 * (double)grid * 4.0 / ((double)value * 1e9);
                pace16 = (int)(seconds * (double)khz * 1e3 * 16.0 + 0.5);
            }
            if (nt) util_fill_paced_kernel<true><<<grid, 256, 0, c->stream>>>(dev, n4, run4, pace16);
            else util_fill_paced_kernel<false><<<grid, 256, 0, c->stream>>>(dev, n4, run4, pace16);
        }
        else if (mode == 7) {       // read probe: value = 10 * depth + round_robin (depth 1, 2, 3), grid = cus * gmul
            const int depth = ((int)value) / 10, rr = ((int)value) % 10;
            const int run4 = 200;
#define RD(D) do { if (nt) util_read_kernel<D, true><<<grid, 256, 0, c->stream>>>(dev, n4, run4, rr, dev); \
                   else util_read_kernel<D, false><<<grid, 256, 0, c->stream>>>(dev, n4, run4, rr, dev); } while (0)
            if (depth <= 1) RD(1); else if (depth == 2) RD(2); else RD(3);
#undef RD
        }
        else if (mode == 4) {       // mode 3's chunks written by a whole 256-thread workgroup
            if (nt) util_fill_kernel<true, 3><<<grid, 256, 0, c->stream>>>(dev, n4, value);
            else util_fill_kernel<false, 3><<<grid, 256, 0, c->stream>>>(dev, n4, value);
        } else if (mode == 5) {
            if (nt) util_fill_kernel<true, 5><<<grid, 256, 0, c->stream>>>(dev, n4, value);
            else util_fill_kernel<false, 5><<<grid, 256, 0, c->stream>>>(dev, n4, value);
        } else {
            // mode 3: grid = cus * gmul workgroups of ONE wavefront
            if (nt) util_fill_kernel<true, 3><<<grid, 64, 0, c->stream>>>(dev, n4, value);
            else util_fill_kernel<false, 3><<<grid, 64, 0, c->stream>>>(dev, n4, value);
        }
#undef FILL
    }
    HGMM_HIP(c, hipGetLastError());
    return HGMM_OK;
}

extern "C" int hgmm_version(void) { return 100; }

extern "C" int hgmm_device_count(int* count) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (count) *count = (e == hipSuccess) ? n : 0;
    return e == hipSuccess ? HGMM_OK : HGMM_ERR_NODEVICE;
}

extern "C" const char* hgmm_last_error(const hgmm_ctx* c) {
    if (c) return c->err.c_str();
    std::lock_guard<std::mutex> lk(g_mu);
    return g_create_error.c_str();
}

namespace hgmm {
const ConfigSpec CONFIG_SPECS[CFG_COUNT] = {
    {"estep_target_gbs", -1, -1, 20000}, {"pace_start", 6600, 1000, 20000}, {"pace_forget", 10000, 1, 1 << 30},
    {"predict_single_row", 0, 0, 1},     {"tree_no_chol", 0, 0, 1},         {"tree_rel", 0, 0, 1},
    {"tree_ahead", 2, 0, 64},            {"tree_tickets", 0, 0, 1},         {"tree_overlap", 1, 0, 1},
    {"fullcov_two_pass", 0, 0, 1},       {"kmpp_two_launches", 0, 0, 1},    {"kmeans_acc_regs", 0, 0, 1},
    {"ipc_timeout_s", 20, 1, 600},       {"reg_device_solve", 0, 0, 1},
};
static int config_find(const char* name) {
    if (!name) return -1;
    for (int k = 0; k < CFG_COUNT; ++k)
        if (std::strcmp(name, CONFIG_SPECS[k].name) == 0) return k;
    return -1;
}
// defaults, then HGMM_<NAME> from the environment (the library's one getenv)
static void config_init(hgmm_ctx* c) {
    for (int k = 0; k < CFG_COUNT; ++k) {
        const ConfigSpec& sp = CONFIG_SPECS[k];
        c->cfg[k] = sp.dflt;
        std::string var = "HGMM_";
        for (const char* p = sp.name; *p; ++p) var += (char)std::toupper((unsigned char)*p);
        const char* v = std::getenv(var.c_str());
        if (v && *v) c->cfg[k] = std::max(sp.lo, std::min(sp.hi, atoi(v)));
    }
}
}  // namespace hgmm

extern "C" int hgmm_config_count(void) { return CFG_COUNT; }
extern "C" const char* hgmm_config_name(int index) { return (index >= 0 && index < CFG_COUNT) ? CONFIG_SPECS[index].name : nullptr; }
extern "C" int hgmm_config_set(hgmm_ctx* c, const char* name, int value) {
    if (!c) return HGMM_ERR_ARG;
    const int k = config_find(name);
    if (k < 0) return fail(c, HGMM_ERR_ARG, "hgmm_config_set: unknown option '%s'", name ? name : "(null)");
    const ConfigSpec& sp = CONFIG_SPECS[k];
    if (value < sp.lo || value > sp.hi)
        return fail(c, HGMM_ERR_ARG, "hgmm_config_set: %s = %d outside [%d, %d]", sp.name, value, sp.lo, sp.hi);
    c->cfg[k] = value;
    return HGMM_OK;
}
extern "C" int hgmm_config_get(hgmm_ctx* c, const char* name, int* value_out) {
    if (!c || !value_out) return HGMM_ERR_ARG;
    const int k = config_find(name);
    if (k < 0) return fail(c, HGMM_ERR_ARG, "hgmm_config_get: unknown option '%s'", name ? name : "(null)");
    *value_out = c->cfg[k];
    return HGMM_OK;
}

extern "C" int hgmm_create(int device_id, hgmm_ctx** out) {
    if (!out) return HGMM_ERR_ARG;
    *out = nullptr;
    auto set_err = [](const std::string& s) {
        std::lock_guard<std::mutex> lk(g_mu);
        g_create_error = s;
    };
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        set_err(std::string("no HIP device available: ") + hipGetErrorString(e));
        return HGMM_ERR_NODEVICE;
    }
    if (device_id < 0 || device_id >= n) {
        set_err("device_id out of range");
        return HGMM_ERR_ARG;
    }
    e = hipSetDevice(device_id);
    if (e != hipSuccess) {
        set_err(std::string("hipSetDevice failed: ") + hipGetErrorString(e));
        return HGMM_ERR_HIP;
    }
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, device_id);
    if (e != hipSuccess) {
        set_err(std::string("hipGetDeviceProperties failed: ") + hipGetErrorString(e));
        return HGMM_ERR_HIP;
    }
    hgmm_ctx* c = new hgmm_ctx();
    c->device = device_id;
    config_init(c);
    c->cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        set_err(std::string("hipStreamCreate failed: ") + hipGetErrorString(e));
        delete c;
        return HGMM_ERR_HIP;
    }
    *out = c;
    return HGMM_OK;
}

extern "C" int hgmm_destroy(hgmm_ctx* c) {
    if (!c) return HGMM_OK;
    (void)hipSetDevice(c->device);
    (void)ctx_stream_sync(c);
    if (c->comm) { (void)ncclCommDestroy(c->comm); c->comm = nullptr; }
    hostcomm_close(c);
    ipc_close(c);
    // (x_aos / x_soa64 are views; handles a caller still holds are the caller's to destroy -- before the context)
    DevBuf* bufs[] = {&c->own_points.x_aos, &c->own_points.x_soa64, &c->f_block, &c->f_pack,
                      &c->f_partials, &c->f_lpn_partials, &c->f_stats, &c->f_lls, &c->f_ctl, &c->f_hint,
                      &c->scratch, &c->t_pi, &c->t_mu, &c->t_cov, &c->t_prep, &c->t_cplx, &c->t_mom,
                      &c->t_parent, &c->t_current, &c->t_perm, &c->t_seg, &c->t_chunks, &c->t_partials,
                      &c->t_q, &c->tgt_soa64, &c->comm_buf, &c->t_xs3, &c->t_llp, &c->t_qtrace, &c->f_cm, &c->f_cs, &c->f_ca, &c->f_lpn2,
                      &c->km_closest, &c->km_block, &c->km_centres, &c->km_ids, &c->km_rand, &c->km_labels,
                      &c->km_mind2, &c->km_partial, &c->km_out, &c->gt_buf, &c->t_momq, &c->t_flags, &c->exp_tab2, &c->t_tickets,
                      &c->fr_pi, &c->fr_mu, &c->fr_cov, &c->fr_prep, &c->fr_mom, &c->fr_clouds, &c->fr_q, &c->fr_trace, &c->fr_tg,
                      &c->fr_momq, &c->fr_reg, &c->ff_origin, &c->ff_clocks};
    for (DevBuf* b : bufs)
        if (b->p) (void)hipFree(b->p);
    if (c->h_stage) (void)hipHostFree(c->h_stage);
    if (c->forest.host) (void)hipHostFree(c->forest.host);
    if (c->h_scalars) (void)hipHostFree(c->h_scalars);
    for (hipEvent_t& e : c->ev_slots) if (e) (void)hipEventDestroy(e);
    if (c->pace.have_events)
        for (auto& pr : c->pace.ev) { (void)hipEventDestroy(pr[0]); (void)hipEventDestroy(pr[1]); }
    if (c->pace.stamps) (void)hipHostFree(c->pace.stamps);
    if (c->tree_hctl) { (void)hipHostFree(c->tree_hctl); (void)hipEventDestroy(c->tree_ev[0]); (void)hipEventDestroy(c->tree_ev[1]); }
    for (auto& p : c->events) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
    (void)hipStreamDestroy(c->stream);
    delete c;
    return HGMM_OK;
}

extern "C" int hgmm_device_info(hgmm_ctx* c, char* name, int name_len, int* compute_units,
                                int64_t* hbm_bytes) {
    HGMM_ENTER(c);
    hipDeviceProp_t prop;
    HGMM_HIP(c, hipGetDeviceProperties(&prop, c->device));
    if (name && name_len > 0) {
        std::string s = std::string(prop.name) + " (" + prop.gcnArchName + ")";
        strncpy(name, s.c_str(), (size_t)name_len - 1);
        name[name_len - 1] = 0;
    }
    if (compute_units) *compute_units = prop.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = (int64_t)prop.totalGlobalMem;
    return HGMM_OK;
}

extern "C" int hgmm_synchronize(hgmm_ctx* c) {
    HGMM_ENTER(c);
    HGMM_HIP(c, ctx_stream_sync(c));
    return HGMM_OK;
}

extern "C" int hgmm_alloc(hgmm_ctx* c, size_t bytes, void** dev_out) {
    HGMM_ENTER(c);
    if (!c || !dev_out) return HGMM_ERR_ARG;
    HGMM_HIP(c, hipSetDevice(c->device));
    HGMM_HIP(c, hipMalloc(dev_out, bytes ? bytes : 4));
    return HGMM_OK;
}
extern "C" int hgmm_free(hgmm_ctx* c, void* dev) {
    HGMM_ENTER(c);
    if (!dev) return HGMM_OK;
    HGMM_HIP(c, ctx_stream_sync(c));
    HGMM_HIP(c, hipFree(dev));
    return HGMM_OK;
}
extern "C" int hgmm_host_scalars(hgmm_ctx* c, int count, double** host_out, double** dev_out) {
    HGMM_ENTER(c);
    if (!c || !host_out || !dev_out) return c ? fail(c, HGMM_ERR_ARG, "hgmm_host_scalars: NULL output") : HGMM_ERR_ARG;
    if (count < 1 || count > 4096) return fail(c, HGMM_ERR_ARG, "hgmm_host_scalars: count %d outside 1..4096", count);
    HGMM_HIP(c, hipSetDevice(c->device));
    if (!c->h_scalars) {
        void* p = nullptr;
        HGMM_HIP(c, hipHostMalloc(&p, sizeof(double) * (size_t)count, hipHostMallocMapped));
        std::memset(p, 0, sizeof(double) * (size_t)count);
        c->h_scalars = static_cast<double*>(p);
        c->h_scalars_n = count;
    } else if (count > c->h_scalars_n) {
        return fail(c, HGMM_ERR_STATE, "hgmm_host_scalars: the array already exists with %d entries", c->h_scalars_n);
    }
    void* d = nullptr;
    HGMM_HIP(c, hipHostGetDevicePointer(&d, c->h_scalars, 0));
    *host_out = c->h_scalars;
    *dev_out = static_cast<double*>(d);
    return HGMM_OK;
}
extern "C" int hgmm_event_record(hgmm_ctx* c, int slot) {
    HGMM_ENTER(c);
    if (slot < 0 || slot >= HGMM_EVENT_SLOTS) return fail(c, HGMM_ERR_ARG, "event slot %d", slot);
    HGMM_HIP(c, hipSetDevice(c->device));
    if (!c->ev_slots[slot]) HGMM_HIP(c, hipEventCreateWithFlags(&c->ev_slots[slot], hipEventDisableTiming));
    HGMM_HIP(c, hipEventRecord(c->ev_slots[slot], c->stream));
    return HGMM_OK;
}
extern "C" int hgmm_event_wait(hgmm_ctx* c, int slot) {
    HGMM_ENTER(c);
    if (slot < 0 || slot >= HGMM_EVENT_SLOTS) return fail(c, HGMM_ERR_ARG, "event slot %d", slot);
    if (!c->ev_slots[slot]) return fail(c, HGMM_ERR_STATE, "event slot %d was never recorded", slot);
    HGMM_HIP(c, hipEventSynchronize(c->ev_slots[slot]));
    return HGMM_OK;
}
extern "C" int hgmm_h2d(hgmm_ctx* c, void* dev_dst, const void* host_src, size_t bytes) {
    HGMM_ENTER(c);
    HGMM_HIP(c, hipMemcpyAsync(dev_dst, host_src, bytes, hipMemcpyHostToDevice, c->stream));
    HGMM_HIP(c, ctx_stream_sync(c));
    return HGMM_OK;
}
extern "C" int hgmm_d2h(hgmm_ctx* c, void* host_dst, const void* dev_src, size_t bytes) {
    HGMM_ENTER(c);
    HGMM_HIP(c, hipMemcpyAsync(host_dst, dev_src, bytes, hipMemcpyDeviceToHost, c->stream));
    HGMM_HIP(c, ctx_stream_sync(c));
    return HGMM_OK;
}

extern "C" int hgmm_d2d(hgmm_ctx* c, void* dev_dst, const void* dev_src, size_t bytes) {
    HGMM_ENTER(c);
    if (bytes == 0) return HGMM_OK;
    if (!dev_dst || !dev_src) return fail(c, HGMM_ERR_ARG, "hgmm_d2d: NULL pointer");
    HGMM_HIP(c, hipMemcpyAsync(dev_dst, dev_src, bytes, hipMemcpyDeviceToDevice, c->stream));
    return HGMM_OK;
}

extern "C" int64_t hgmm_num_points(const hgmm_ctx* c) { return c ? c->n : 0; }

// the cloud `p` becomes the one every kernel of the context works on
static void bind_points(hgmm_ctx* c, hgmm_points* p) {
    c->bound = p;
    c->x_aos = p ? p->x_aos : DevBuf();
    c->x_soa64 = p ? p->x_soa64 : DevBuf();
    c->n = p ? p->n : 0;
    c->n_pad = p ? p->n_pad : 0;
    c->have_f32 = c->have_f64 = p != nullptr && p->n > 0;
    c->flat.active = false;
    c->tree.nodes_ready = false;
    c->km_labels_n = -1;              // labels / distances of the previous cloud are void
}

static int points_alloc(hgmm_ctx* c, hgmm_points* p, int64_t n) {
    if (n <= 0) return fail(c, HGMM_ERR_ARG, "number of points must be positive");
    HGMM_HIP(c, hipSetDevice(c->device));
    p->ctx = c;
    p->n = n;
    p->n_pad = (n + 255) / 256 * 256;
    HGMM_TRY(ensure(c, p->x_aos, sizeof(float) * 3 * (size_t)n));
    HGMM_TRY(ensure(c, p->x_soa64, sizeof(double) * 3 * (size_t)p->n_pad));
    return HGMM_OK;
}

static int points_upload_f32(hgmm_ctx* c, hgmm_points* p, const float* xyz, int64_t n) {
    HGMM_TRY(points_alloc(c, p, n));
    HGMM_HIP(c, hipMemcpyAsync(p->x_aos.p, xyz, sizeof(float) * 3 * (size_t)n, hipMemcpyHostToDevice, c->stream));
    aos_to_soa64_f32<<<(unsigned)((p->n_pad + 255) / 256), 256, 0, c->stream>>>(
        p->x_aos.as<float>(), n, p->n_pad, p->x_soa64.as<double>());
    HGMM_HIP(c, hipGetLastError());
    HGMM_HIP(c, ctx_stream_sync(c));
    return HGMM_OK;
}

static int points_upload_f64(hgmm_ctx* c, hgmm_points* p, const double* xyz, int64_t n) {
    HGMM_TRY(points_alloc(c, p, n));
    HGMM_TRY(ensure(c, c->scratch, sizeof(double) * 3 * (size_t)n));
    HGMM_HIP(c, hipMemcpyAsync(c->scratch.p, xyz, sizeof(double) * 3 * (size_t)n, hipMemcpyHostToDevice, c->stream));
    aos_to_soa64_f64<<<(unsigned)((p->n_pad + 255) / 256), 256, 0, c->stream>>>(
        c->scratch.as<double>(), n, p->n_pad, p->x_soa64.as<double>());
    f64_to_f32<<<(unsigned)((3 * n + 255) / 256), 256, 0, c->stream>>>(c->scratch.as<double>(), 3 * n,
                                                                      p->x_aos.as<float>());
    HGMM_HIP(c, hipGetLastError());
    HGMM_HIP(c, ctx_stream_sync(c));
    return HGMM_OK;
}

extern "C" int hgmm_set_points_f32(hgmm_ctx* c, const float* xyz, int64_t n) {
    HGMM_ENTER(c);
    if (!c || !xyz) return c ? fail(c, HGMM_ERR_ARG, "xyz is NULL") : HGMM_ERR_ARG;
    bind_points(c, nullptr);                       // (a failed upload leaves nothing bound)
    HGMM_TRY(points_upload_f32(c, &c->own_points, xyz, n));
    bind_points(c, &c->own_points);
    return HGMM_OK;
}

extern "C" int hgmm_set_points_f64(hgmm_ctx* c, const double* xyz, int64_t n) {
    HGMM_ENTER(c);
    if (!c || !xyz) return c ? fail(c, HGMM_ERR_ARG, "xyz is NULL") : HGMM_ERR_ARG;
    bind_points(c, nullptr);
    HGMM_TRY(points_upload_f64(c, &c->own_points, xyz, n));
    bind_points(c, &c->own_points);
    return HGMM_OK;
}

// B clouds back to back as ONE resident cloud (the forest of hgmm_tree_build_batch): every cloud is uploaded from its own
// host array -- no concatenated copy on the host -- into the context's own storage.  Only the float64 structure of arrays is
// filled (the HGMM kernels' view); the flat EM's float32 rows are not, and the flat entry points say so.
// (IN = double or float: float32 rows are widened on the device -- exact --, half the bytes over PCIe and no host pass)
template <class IN>
__global__ void aos_to_soa64_at(const IN* __restrict__ in, int64_t n, int64_t first, int64_t n_pad,
                                double* __restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    for (int d = 0; d < 3; ++d) out[d * n_pad + first + i] = (double)in[3 * i + d];
}
template <class IN>
static int set_points_batch(hgmm_ctx* c, int B, const IN* const* xyz, const int64_t* counts) {
    HGMM_ENTER(c);
    if (B < 1 || !xyz || !counts) return fail(c, HGMM_ERR_ARG, "set_points (batch): B = %d", B);
    int64_t total = 0;
    for (int b = 0; b < B; ++b) {
        if (!xyz[b] || counts[b] < 1) return fail(c, HGMM_ERR_ARG, "set_points (batch): cloud %d is empty", b);
        total += counts[b];
    }
    bind_points(c, nullptr);
    hgmm_points* p = &c->own_points;
    HGMM_TRY(points_alloc(c, p, total));
    HGMM_TRY(ensure(c, c->scratch, sizeof(IN) * 3 * (size_t)total));
    IN* stage = c->scratch.as<IN>();
    if (p->n_pad > total)                                    // the padding rows read as the origin, like hgmm_set_points_f64's
        for (int d = 0; d < 3; ++d)
            HGMM_HIP(c, hipMemsetAsync(p->x_soa64.as<double>() + (size_t)d * p->n_pad + total, 0,
                                       sizeof(double) * (size_t)(p->n_pad - total), c->stream));
    int64_t at = 0;
    for (int b = 0; b < B; ++b) {
        HGMM_HIP(c, hipMemcpyAsync(stage + 3 * at, xyz[b], sizeof(IN) * 3 * (size_t)counts[b], hipMemcpyHostToDevice, c->stream));
        aos_to_soa64_at<IN><<<(unsigned)((counts[b] + 255) / 256), 256, 0, c->stream>>>(stage + 3 * at, counts[b], at, p->n_pad,
                                                                                        p->x_soa64.as<double>());
        at += counts[b];
    }
    HGMM_HIP(c, hipGetLastError());
    HGMM_HIP(c, ctx_stream_sync(c));
    bind_points(c, p);
    c->have_f32 = false;
    return HGMM_OK;
}
extern "C" int hgmm_set_points_batch_f64(hgmm_ctx* c, int B, const double* const* xyz, const int64_t* counts) {
    return set_points_batch<double>(c, B, xyz, counts);
}
extern "C" int hgmm_set_points_batch_f32(hgmm_ctx* c, int B, const float* const* xyz, const int64_t* counts) {
    return set_points_batch<float>(c, B, xyz, counts);
}

extern "C" int hgmm_points_create_f32(hgmm_ctx* c, const float* xyz, int64_t n, hgmm_points** out) {
    HGMM_ENTER(c);
    if (!c || !xyz || !out) return c ? fail(c, HGMM_ERR_ARG, "hgmm_points_create: NULL argument") : HGMM_ERR_ARG;
    *out = nullptr;
    hgmm_points* p = new hgmm_points();
    const int rc = points_upload_f32(c, p, xyz, n);
    if (rc != HGMM_OK) {
        if (p->x_aos.p) (void)hipFree(p->x_aos.p);
        if (p->x_soa64.p) (void)hipFree(p->x_soa64.p);
        delete p;
        return rc;
    }
    *out = p;
    return HGMM_OK;
}

extern "C" int hgmm_points_create_f64(hgmm_ctx* c, const double* xyz, int64_t n, hgmm_points** out) {
    HGMM_ENTER(c);
    if (!c || !xyz || !out) return c ? fail(c, HGMM_ERR_ARG, "hgmm_points_create: NULL argument") : HGMM_ERR_ARG;
    *out = nullptr;
    hgmm_points* p = new hgmm_points();
    const int rc = points_upload_f64(c, p, xyz, n);
    if (rc != HGMM_OK) {
        if (p->x_aos.p) (void)hipFree(p->x_aos.p);
        if (p->x_soa64.p) (void)hipFree(p->x_soa64.p);
        delete p;
        return rc;
    }
    *out = p;
    return HGMM_OK;
}

extern "C" int hgmm_points_bind(hgmm_ctx* c, hgmm_points* p) {
    HGMM_ENTER(c);
    if (!p) p = c->own_points.n > 0 ? &c->own_points : nullptr;      // NULL: back to the cloud of hgmm_set_points_*
    if (p && p->ctx != c) return fail(c, HGMM_ERR_ARG, "hgmm_points_bind: the cloud belongs to another context");
    if (p == c->bound) return HGMM_OK;             // (kernels already enqueued keep the pointers they were launched with)
    bind_points(c, p);
    return HGMM_OK;
}

extern "C" int hgmm_points_destroy(hgmm_ctx* c, hgmm_points* p) {
    HGMM_ENTER(c);
    if (!p) return HGMM_OK;
    if (p->ctx != c || p == &c->own_points) return fail(c, HGMM_ERR_ARG, "hgmm_points_destroy: not a cloud created on this context");
    if (c->bound == p) bind_points(c, nullptr);
    HGMM_HIP(c, hipSetDevice(c->device));
    HGMM_HIP(c, ctx_stream_sync(c));               // kernels that read it may still be running
    if (p->x_aos.p) HGMM_HIP(c, hipFree(p->x_aos.p));
    if (p->x_soa64.p) HGMM_HIP(c, hipFree(p->x_soa64.p));
    delete p;
    return HGMM_OK;
}

extern "C" int64_t hgmm_points_count(const hgmm_points* p) { return p ? p->n : 0; }

extern "C" int hgmm_points_download_f32(hgmm_ctx* c, const hgmm_points* p, float* xyz_out) {
    HGMM_ENTER(c);
    if (!p) p = c->bound;                          // NULL: the cloud the context works on
    if (!p || !xyz_out) return fail(c, HGMM_ERR_ARG, "hgmm_points_download_f32: no cloud / NULL output");
    if (p->ctx != c && p != &c->own_points) return fail(c, HGMM_ERR_ARG, "hgmm_points_download_f32: the cloud belongs to another context");
    HGMM_HIP(c, hipMemcpyAsync(xyz_out, p->x_aos.p, sizeof(float) * 3 * (size_t)p->n, hipMemcpyDeviceToHost, c->stream));
    HGMM_HIP(c, ctx_stream_sync(c));
    return HGMM_OK;
}

// ---- RCCL -----------------------------------------------------------------------------------
extern "C" int hgmm_comm_unique_id(void* id128_out) {
    if (!id128_out) return HGMM_ERR_ARG;
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId size");
    ncclUniqueId id;
    if (ncclGetUniqueId(&id) != ncclSuccess) return HGMM_ERR_RCCL;
    memcpy(id128_out, &id, sizeof id);
    return HGMM_OK;
}

extern "C" int hgmm_comm_init_rank(hgmm_ctx* c, int nranks, int rank, const void* id128) {
    HGMM_ENTER(c);
    if (!c || !id128) return HGMM_ERR_ARG;
    if (nranks < 1 || rank < 0 || rank >= nranks) return fail(c, HGMM_ERR_ARG, "bad rank %d / %d", rank, nranks);
    if (c->comm_on()) return fail(c, HGMM_ERR_STATE, "communicator already attached");
    HGMM_HIP(c, hipSetDevice(c->device));
    ncclUniqueId id;
    memcpy(&id, id128, sizeof id);
    HGMM_NCCL(c, ncclCommInitRank(&c->comm, nranks, id, rank));
    c->nranks = nranks;
    c->rank = rank;
    return HGMM_OK;
}

extern "C" int hgmm_comm_init_host(hgmm_ctx* c, int nranks, int rank, const char* name) {
    HGMM_ENTER(c);
    if (!c || !name || !name[0]) return c ? fail(c, HGMM_ERR_ARG, "host communicator: name is empty") : HGMM_ERR_ARG;
    if (nranks < 1 || rank < 0 || rank >= nranks) return fail(c, HGMM_ERR_ARG, "bad rank %d / %d", rank, nranks);
    if (c->comm_on()) return fail(c, HGMM_ERR_STATE, "communicator already attached");
    HostComm* h = new HostComm;
    h->name = std::string("/") + name;
    h->bytes = sizeof(HostCommShm) + 64 + sizeof(double) * HOSTCOMM_SLOT * (size_t)nranks;
    h->owner = rank == 0;
    void* p = shm_attach(h->name, h->bytes, h->owner, HOSTCOMM_TIMEOUT_S);
    if (!p) { delete h; return fail(c, HGMM_ERR_STATE, "host communicator: cannot %s shared memory %s", rank == 0 ? "create" : "join", name); }
    h->shm = static_cast<HostCommShm*>(p);
    h->data = reinterpret_cast<double*>(static_cast<char*>(p) + ((sizeof(HostCommShm) + 63) / 64) * 64);
    if (rank == 0) {
        h->shm->count.store(0);
        h->shm->generation.store(0);
        h->shm->nranks = nranks;
        h->shm->slot_doubles = HOSTCOMM_SLOT;
        h->shm->ready.store(1, std::memory_order_release);
    } else if (h->shm->nranks != nranks) {
        munmap(p, h->bytes);
        delete h;
        return fail(c, HGMM_ERR_ARG, "host communicator: world size mismatch");
    }
    c->hcomm = h;
    c->nranks = nranks;
    c->rank = rank;
    return hostcomm_barrier(c);
}

extern "C" int hgmm_comm_init_ipc(hgmm_ctx* c, int nranks, int rank, const char* name) {
    HGMM_ENTER(c);
    if (!c || !name || !name[0]) return c ? fail(c, HGMM_ERR_ARG, "peer exchange: name is empty") : HGMM_ERR_ARG;
    if (nranks < 1 || nranks > IPC_MAX_RANKS || rank < 0 || rank >= nranks)
        return fail(c, HGMM_ERR_ARG, "peer exchange: bad rank %d / %d (at most %d ranks: one node)", rank, nranks, IPC_MAX_RANKS);
    if (c->comm_on()) return fail(c, HGMM_ERR_STATE, "communicator already attached");
    HGMM_HIP(c, hipSetDevice(c->device));
    IpcComm* ic = new IpcComm;
    c->icomm = ic;
    c->nranks = nranks;
    c->rank = rank;
    auto bail = [&](int code, const char* what, const char* detail) {
        ipc_close(c);                                       // (in_step is false: no barrier, the peers are not in step)
        c->nranks = 1;
        c->rank = 0;
        return fail(c, code, "peer exchange: %s%s%s", what, detail[0] ? ": " : "", detail);
    };
    // exchange buffer: device memory that peers write while kernels here read it -> uncached (fine-grained as second choice)
    hipError_t e = hipExtMallocWithFlags(&ic->local, IPC_BUF_BYTES, hipDeviceMallocUncached);
    if (e != hipSuccess) { (void)hipGetLastError(); e = hipExtMallocWithFlags(&ic->local, IPC_BUF_BYTES, hipDeviceMallocFinegrained); }
    if (e != hipSuccess) { ic->local = nullptr; return bail(HGMM_ERR_HIP, "no uncached device memory", hipGetErrorString(e)); }
    e = hipMemset(ic->local, 0, IPC_FLAG_BYTES);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    void* eh = nullptr;
    if (e == hipSuccess) e = hipHostMalloc(&eh, 64, hipHostMallocMapped);
    if (e != hipSuccess) return bail(HGMM_ERR_HIP, "setup failed", hipGetErrorString(e));
    ic->err_host = static_cast<unsigned*>(eh);
    *ic->err_host = 0;
    void* ed = nullptr;
    if ((e = hipHostGetDevicePointer(&ed, eh, 0)) != hipSuccess) return bail(HGMM_ERR_HIP, "setup failed", hipGetErrorString(e));
    ic->err_dev = static_cast<unsigned*>(ed);
    int khz = 0;
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, c->device) != hipSuccess || khz <= 0) khz = 100000;
    const double timeout_s = (double)c->cfg[CFG_IPC_TIMEOUT_S];
    ic->timeout_ticks = (long long)(timeout_s * 1000.0 * (double)khz);
    hipIpcMemHandle_t mine;
    if ((e = hipIpcGetMemHandle(&mine, ic->local)) != hipSuccess)
        return bail(HGMM_ERR_HIP, "hipIpcGetMemHandle", hipGetErrorString(e));
    // the handles meet in a POSIX shared-memory object
    ic->name = std::string("/") + name;
    ic->shm_bytes = sizeof(IpcShm);
    ic->owner = rank == 0;
    void* p = shm_attach(ic->name, ic->shm_bytes, ic->owner, HOSTCOMM_TIMEOUT_S);
    if (!p) return bail(HGMM_ERR_STATE, rank == 0 ? "cannot create shared memory" : "rank 0's shared memory never became ready", name);
    ic->shm = static_cast<IpcShm*>(p);
    if (rank == 0) {
        ic->shm->count.store(0);
        ic->shm->generation.store(0);
        ic->shm->failed.store(0);
        ic->shm->nranks = nranks;
        ic->shm->ready.store(1, std::memory_order_release);
    } else if (ic->shm->nranks != nranks) {
        return bail(HGMM_ERR_ARG, "world size mismatch", "");
    }
    ic->shm->handle[rank] = mine;
    ic->shm->device[rank] = c->device;
    ic->shm->pid[rank] = (int)getpid();
    if (ipc_barrier(c) != HGMM_OK) return bail(HGMM_ERR_STATE, "a peer rank never published its buffer", "");
    for (int r = 0; r < nranks; ++r) {
        if (r == rank) { ic->mapped[r] = ic->local; continue; }
        if (ic->shm->pid[r] == (int)getpid()) return bail(HGMM_ERR_ARG, "two ranks in one process", "");
        void* q = nullptr;
        e = hipIpcOpenMemHandle(&q, ic->shm->handle[r], hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) {
            char msg[160];
            snprintf(msg, sizeof msg, "%s (rank %d on device %d -> rank %d on device %d)", hipGetErrorString(e), rank, c->device,
                     r, ic->shm->device[r]);
            ic->shm->failed.store(1, std::memory_order_release);
            (void)ipc_barrier(c);                            // the peers are waiting in the barrier below, and look at `failed`
            return bail(HGMM_ERR_HIP, "hipIpcOpenMemHandle", msg);
        }
        ic->mapped[r] = q;
    }
    for (int r = 0; r < nranks; ++r) {
        ic->peers.flags[r] = static_cast<unsigned long long*>(ic->mapped[r]);
        ic->peers.slots[r] = reinterpret_cast<double*>(static_cast<char*>(ic->mapped[r]) + IPC_FLAG_BYTES);
    }
    if (ipc_barrier(c) != HGMM_OK) return bail(HGMM_ERR_STATE, "a peer rank never finished mapping", "");
    if (ic->shm->failed.load(std::memory_order_acquire)) return bail(HGMM_ERR_STATE, "a peer rank could not map the buffers", "");
    c->icomm_err = ic->err_host;                             // every rank has every buffer mapped
    ic->in_step = true;
    return HGMM_OK;
}

extern "C" int hgmm_comm_destroy(hgmm_ctx* c) {
    HGMM_ENTER(c);
    HGMM_HIP(c, hipSetDevice(c->device));                  // (the current device is per THREAD: a caller may tear down from another one)
    if (c->hcomm) {
        HGMM_HIP(c, ctx_stream_sync(c));
        (void)hostcomm_barrier(c);                          // nobody unlinks while a peer still reduces
        hostcomm_close(c);
        c->nranks = 1;
        c->rank = 0;
    }
    if (c->icomm) {
        const hipError_t e = ctx_stream_sync(c);
        ipc_close(c);                                       // (barrier inside: nobody unmaps while a peer still exchanges)
        c->nranks = 1;
        c->rank = 0;
        if (e != hipSuccess) return fail(c, HGMM_ERR_HIP, "peer exchange: %s", hipGetErrorString(e));
    }
    if (c->comm) {
        HGMM_HIP(c, ctx_stream_sync(c));
        HGMM_NCCL(c, ncclCommDestroy(c->comm));
        c->comm = nullptr;
        c->nranks = 1;
        c->rank = 0;
    }
    return HGMM_OK;
}

extern "C" int hgmm_comm_stats(hgmm_ctx* c, unsigned long long* collectives_out, unsigned long long* surplus_tree_iterations_out) {
    if (!c) return HGMM_ERR_ARG;
    if (collectives_out) *collectives_out = c->collectives;
    if (surplus_tree_iterations_out) *surplus_tree_iterations_out = c->tree.surplus_iterations;
    return HGMM_OK;
}

extern "C" int hgmm_comm_allreduce_f64(hgmm_ctx* c, double* host_inout, int n, int op) {
    HGMM_ENTER(c);
    if (!c || !host_inout || n < 1) return HGMM_ERR_ARG;
    if (!c->comm_on()) return HGMM_OK;   // single rank: identity
    c->collectives++;
    HGMM_TRY(ensure(c, c->comm_buf, sizeof(double) * (size_t)n));
    HGMM_HIP(c, hipMemcpyAsync(c->comm_buf.p, host_inout, sizeof(double) * n, hipMemcpyHostToDevice, c->stream));
    if (c->icomm)
        HGMM_TRY(ipc_allreduce(c, c->comm_buf.as<double>(), c->comm_buf.as<double>(), (size_t)n, op == 1 ? 1 : 0));
    else if (c->hcomm)
        HGMM_TRY(hostcomm_allreduce_dev(c, c->comm_buf.as<double>(), (size_t)n, op));
    else
        HGMM_NCCL(c, ncclAllReduce(c->comm_buf.p, c->comm_buf.p, (size_t)n, ncclDouble,
                                   op == 1 ? ncclMax : ncclSum, c->comm, c->stream));
    HGMM_HIP(c, hipMemcpyAsync(host_inout, c->comm_buf.p, sizeof(double) * n, hipMemcpyDeviceToHost, c->stream));
    HGMM_HIP(c, ctx_stream_sync(c));
    return HGMM_OK;
}

// ---- profiling --------------------------------------------------------------------------------
extern "C" int hgmm_profile_enable(hgmm_ctx* c, int on) {
    HGMM_ENTER(c);
    if (!on) HGMM_TRY(profile_collect(c));
    c->profiling = on != 0;
    return HGMM_OK;
}
extern "C" int hgmm_profile_reset(hgmm_ctx* c) {
    HGMM_ENTER(c);
    HGMM_TRY(profile_collect(c));
    for (int i = 0; i < HGMM_K_COUNT; ++i) { c->prof_ms[i] = 0.0; c->prof_n[i] = 0; }
    return HGMM_OK;
}
extern "C" int hgmm_profile_get(hgmm_ctx* c, int kernel_id, double* total_ms_out, int64_t* launches_out) {
    HGMM_ENTER(c);
    if (!c || kernel_id < 0 || kernel_id >= HGMM_K_COUNT) return HGMM_ERR_ARG;
    HGMM_TRY(profile_collect(c));
    if (total_ms_out) *total_ms_out = c->prof_ms[kernel_id];
    if (launches_out) *launches_out = c->prof_n[kernel_id];
    return HGMM_OK;
}
