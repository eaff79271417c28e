// Grid-barrier cost on gfx950 (MI355X): what ONE synchronisation of a co-resident grid costs, measured -- the figure the
// k-means++ seeding's design turns on (csrc/kmeans_kernels.hip: one launch per centre, ~1.5 us per launch boundary,
// against a persistent kernel that would keep every workgroup's points in registers and pay a grid barrier per centre).
//
// Grid: one 1024-thread workgroup per CU on 245 CUs (the seeding pass's grid at N = 10^6: 4096 points per workgroup),
// all co-resident (cooperative-style; the probe checks that every workgroup arrived at the first barrier).  Per round
// every workgroup publishes 8 doubles (a candidate's block sums: the payload of the real exchange) and then needs
// EVERYBODY's.  Three barrier constructions, 800 rounds each, wall_clock64 of workgroup 0 / rounds:
//   counter   one agent-scope atomic counter per round parity (fetch_add), spin on it
//   tickets   two-level counters, 4 KB apart (the construction of tree_kernels.hip: store_block_q)
//   flags     no read-modify-write at all: workgroup b stores the round number next to its payload (one 128-byte line
//             per workgroup); thread t of every workgroup polls line t until it carries the round number -- the poll IS
//             the read of the payload
// and, for scale, the same 800 rounds as 800 LAUNCHES of a kernel that does the publish + read of the previous round's
// lines (what the seeding does today).
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/gridbarrier tools/gridbarrier.hip && tools/gridbarrier
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int LINE = 16;                   // doubles per workgroup line (128 bytes): [0..7] payload, [8] round number
constexpr int TICKET_STRIDE = 1024;        // unsigned ints between two ticket counters (4 KB)

// every spin gives up after ~1 s (a grid that is not co-resident must not hang the GPU): the run then reports "gave up"
__device__ unsigned g_gave_up;
#define SPIN_WHILE(cond)                                                                         \
    do {                                                                                         \
        const unsigned long long s0_ = wall_clock64();                                           \
        while (cond) {                                                                           \
            __builtin_amdgcn_s_sleep(1);                                                         \
            if (wall_clock64() - s0_ > 100000000ull || __hip_atomic_load(&g_gave_up, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { \
                __hip_atomic_store(&g_gave_up, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  \
                break;                                                                           \
            }                                                                                    \
        }                                                                                        \
    } while (0)

template <int MODE>   // 0 counter, 1 tickets, 2 flags (release / acquire); 3 tickets, 4 flags with RELAXED agent-scope atomics + s_waitcnt
// (csrc/tree_kernels.hip store_block_q: relaxed 8-byte agent-scope atomics on both sides are a complete hand-off; release /
//  acquire at agent scope add an L2 write-back and an L1 invalidate per operation)
__global__ __launch_bounds__(1024) void persistent(double* lines, unsigned* counters, int rounds, double* sink,
                                                   unsigned long long* ticks, int work) {
    const int b = blockIdx.x, G = gridDim.x, tid = threadIdx.x;
    __shared__ double sh[1024];
    __shared__ int go;
    double acc = 0.0;
    unsigned long long t0 = 0;
    for (int r = 1; r <= rounds; ++r) {
        if (r == 2 && b == 0 && tid == 0) t0 = wall_clock64();          // (round 1 absorbs the grid's start-up skew)
        // a little arithmetic per round stands in for the pass (work = fp64 fma per thread)
        double v = (double)(tid + r);
        for (int k = 0; k < work; ++k) v = fma(v, 1.0000001, 1e-9);
        // publish: 8 doubles + (flags mode) the round number, one line per workgroup
        double* mine = lines + (size_t)((r & 1) * G + b) * LINE;
        if (tid < 8) __hip_atomic_store(mine + tid, v + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        constexpr bool FLAGS = MODE == 2 || MODE == 4;
        constexpr bool RLX = MODE >= 3;
        if (FLAGS) {
            __syncthreads();
            if (tid == 0) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (RLX) __hip_atomic_store(mine + 8, (double)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else __hip_atomic_store(mine + 8, (double)r, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            }
            // poll: thread t waits for workgroup t's line of this round
            double got = 0.0;
            if (tid < G) {
                const double* theirs = lines + (size_t)((r & 1) * G + tid) * LINE;
                if (RLX) SPIN_WHILE(__hip_atomic_load(theirs + 8, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (double)r);
                else SPIN_WHILE(__hip_atomic_load(theirs + 8, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != (double)r);
                got = __hip_atomic_load(theirs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            sh[tid] = got;
            __syncthreads();
        } else {
            __syncthreads();
            if (tid == 0) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (MODE == 0) {   // (counter)
                    unsigned* c = counters + (r & 1) * TICKET_STRIDE;
                    __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                    SPIN_WHILE(__hip_atomic_load(c, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(((r + 1) / 2) * G));
                } else {
                    // two levels: 64 group counters, then the top counter; everybody spins on the top counter
                    const int ng = G < 64 ? G : 64;
                    const int g = b % ng;
                    const unsigned members = (unsigned)((G - g + ng - 1) / ng);
                    unsigned* base = counters + (size_t)(r & 1) * 80 * TICKET_STRIDE;
                    unsigned* mine_c = base + (size_t)(1 + g) * TICKET_STRIDE;
                    const unsigned round_no = (unsigned)((r + 1) / 2);
                    if (RLX) {
                        if (__hip_atomic_fetch_add(mine_c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == round_no * members - 1)
                            __hip_atomic_fetch_add(base, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        SPIN_WHILE(__hip_atomic_load(base, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < round_no * (unsigned)ng);
                    } else {
                        if (__hip_atomic_fetch_add(mine_c, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT) == round_no * members - 1)
                            __hip_atomic_fetch_add(base, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                        SPIN_WHILE(__hip_atomic_load(base, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < round_no * (unsigned)ng);
                    }
                }
                go = r;
            }
            __syncthreads();
            double got = 0.0;
            if (tid < G) got = __hip_atomic_load(lines + (size_t)((r & 1) * G + tid) * LINE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            sh[tid] = got;
            __syncthreads();
        }
        acc += sh[(tid * 7 + r) & 1023];
    }
    if (b == 0 && tid == 0) ticks[0] = wall_clock64() - t0;
    if (acc == 123.456) sink[0] = acc;
}

__global__ __launch_bounds__(1024) void one_round(double* lines, int r, double* sink, int work) {
    const int b = blockIdx.x, G = gridDim.x, tid = threadIdx.x;
    __shared__ double sh[1024];
    double got = 0.0;
    if (tid < G) got = lines[(size_t)(((r - 1) & 1) * G + tid) * LINE];       // the previous launch's lines
    double v = (double)(tid + r);
    for (int k = 0; k < work; ++k) v = fma(v, 1.0000001, 1e-9);
    if (tid < 8) lines[(size_t)((r & 1) * G + b) * LINE + tid] = v + tid;
    sh[tid] = got;
    __syncthreads();
    if (sh[(tid * 7 + r) & 1023] == 123.456) sink[0] = v;
}

int main() {
    int dev = 0, cus = 0, khz = 0;
    CHECK(hipSetDevice(dev));
    CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    CHECK(hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev));
    const int G = cus >= 245 ? 245 : cus - 1, rounds = 801;
    double *lines, *sink;
    unsigned* counters;
    unsigned long long* ticks;
    CHECK(hipMalloc(&lines, sizeof(double) * 2 * G * LINE));
    CHECK(hipMalloc(&sink, 64));
    CHECK(hipMalloc(&counters, sizeof(unsigned) * 2 * 80 * TICKET_STRIDE));
    CHECK(hipHostMalloc(&ticks, 64, hipHostMallocMapped));
    printf("grid %d workgroups x 1024 threads on %d CUs, %d timed rounds, wall clock %d kHz\n", G, cus, rounds - 1, khz);
    for (int work : {0, 600}) {
        for (int mode = 0; mode < 5; ++mode) {
            double best = 1e30;
            for (int rep = 0; rep < 3; ++rep) {
                CHECK(hipMemset(lines, 0, sizeof(double) * 2 * G * LINE));
                CHECK(hipMemset(counters, 0, sizeof(unsigned) * 2 * 80 * TICKET_STRIDE));
                ticks[0] = 0;
                if (mode == 0) persistent<0><<<G, 1024>>>(lines, counters, rounds, sink, ticks, work);
                else if (mode == 1) persistent<1><<<G, 1024>>>(lines, counters, rounds, sink, ticks, work);
                else if (mode == 2) persistent<2><<<G, 1024>>>(lines, counters, rounds, sink, ticks, work);
                else if (mode == 3) persistent<3><<<G, 1024>>>(lines, counters, rounds, sink, ticks, work);
                else persistent<4><<<G, 1024>>>(lines, counters, rounds, sink, ticks, work);
                CHECK(hipDeviceSynchronize());
                unsigned gave_up = 0;
                CHECK(hipMemcpyFromSymbol(&gave_up, HIP_SYMBOL(g_gave_up), sizeof gave_up));
                if (gave_up) { printf("mode %d: a spin gave up after 1 s (grid not co-resident?)\n", mode); return 2; }
                const double us = (double)ticks[0] / (double)khz * 1e3 / (rounds - 1);
                if (us < best) best = us;
            }
            printf("persistent, %-17s barrier, %4d fma per thread and round: %7.3f us per round\n",
                   mode == 0 ? "counter" : mode == 1 ? "tickets" : mode == 2 ? "flags" : mode == 3 ? "tickets (relaxed)" : "flags (relaxed)", work, best);
        }
        double best = 1e30;
        for (int rep = 0; rep < 3; ++rep) {
            CHECK(hipDeviceSynchronize());
            const auto t0 = std::chrono::steady_clock::now();
            for (int r = 1; r < rounds; ++r) one_round<<<G, 1024>>>(lines, r, sink, work);
            CHECK(hipDeviceSynchronize());
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / (rounds - 1);
            if (us < best) best = us;
        }
        printf("one launch per round (stream order is the barrier), %4d fma per thread and round: %7.3f us per round\n", work, best);
    }
    return 0;
}
