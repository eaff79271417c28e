// VALU / MFMA issue-rate probe for gfx950 (MI355X): cycles per wave-instruction of the instruction
// classes the flat EM kernels are made of, at 1 and 2 waves per SIMD.  Evidence for the roofline of
// flat_fused_pk_kernel (VALU-bound): what a packed fp32 op, a plain fp32 op, a transcendental and a DPP
// step cost, and whether fp32 MFMA work overlaps VALU work of the same wave.
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/valubench tools/valubench.hip && tools/valubench
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

#define REP8(X) X X X X X X X X
#define REP64(X) REP8(REP8(X))

enum { OP_FMA = 0, OP_PKFMA, OP_PKMUL, OP_PKADD, OP_EXP, OP_RCP, OP_DPPADD, OP_FMA64, OP_MFMA, OP_MFMA_PKFMA,
       OP_MIX_FUSED, OP_PKFMA_DEP, OP_FMA_DEP, OP_ADD, OP_MIX_PK_SP, OP_MIX_SP_EXP, OP_COUNT };
static const char* NAMES[] = {"v_fma_f32 (8 chains)", "v_pk_fma_f32 (8 chains)", "v_pk_mul_f32 (8 chains)",
                              "v_pk_add_f32 (8 chains)", "v_exp_f32 (8 chains)", "v_rcp_f32 (8 chains)",
                              "v_add_f32 dpp row_shr:1 (8 chains)", "v_fma_f64 (8 chains)",
                              "v_mfma_f32_16x16x4_f32 (4 acc)", "mfma_f32_16x16x4 + 8 v_pk_fma_f32 per mfma",
                              "fused-kernel mix: 24 pk + 2 exp per pair", "v_pk_fma_f32 (1 dependent chain)",
                              "v_fma_f32 (1 dependent chain)", "v_add_f32 (8 chains)",
                              "4 v_pk_fma_f32 + 4 v_fma_f32 interleaved", "6 v_fma_f32 + 2 v_exp_f32 interleaved"};
static const int INSTR_PER_BODY[] = {64, 64, 64, 64, 64, 64, 64, 64, 64, 72, 26 * 4, 64, 64, 64, 64, 64};

template <int OP>
__global__ __launch_bounds__(256) void probe(float* out, long long* cyc, int iters) {
    float s0 = threadIdx.x * 1e-3f, s1 = s0 + 1.f, s2 = s0 + 2.f, s3 = s0 + 3.f, s4 = s0 + 4.f, s5 = s0 + 5.f,
          s6 = s0 + 6.f, s7 = s0 + 7.f;
    f2 p0 = {s0, s1}, p1 = {s2, s3}, p2 = {s4, s5}, p3 = {s6, s7}, p4 = {s1, s0}, p5 = {s3, s2}, p6 = {s5, s4},
       p7 = {s7, s6};
    const f2 ka = {0.999f, 1.001f}, kb = {1e-3f, -1e-3f};
    const float fa = 0.999f, fb = 1e-3f;
    double d0 = s0, d1 = s1, d2 = s2, d3 = s3, d4 = s4, d5 = s5, d6 = s6, d7 = s7;
    f4 m0 = {0, 0, 0, 0}, m1 = m0, m2 = m0, m3 = m0;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if (OP == OP_FMA) {
            REP8(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n"
                              "v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n"
                              "v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                              : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3), "+v"(s4), "+v"(s5), "+v"(s6), "+v"(s7)
                              : "v"(fa), "v"(fb));)
        } else if (OP == OP_ADD) {
            REP8(asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n"
                              "v_add_f32 %3, %3, %8\n v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n"
                              "v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n"
                              : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3), "+v"(s4), "+v"(s5), "+v"(s6), "+v"(s7)
                              : "v"(fb));)
        } else if (OP == OP_FMA_DEP) {
            REP64(asm volatile("v_fma_f32 %0, %0, %1, %2\n" : "+v"(s0) : "v"(fa), "v"(fb));)
        } else if (OP == OP_PKFMA_DEP) {
            REP64(asm volatile("v_pk_fma_f32 %0, %0, %1, %2\n" : "+v"(p0) : "v"(ka), "v"(kb));)
        } else if (OP == OP_PKFMA || OP == OP_PKMUL || OP == OP_PKADD) {
#define PK8(INS, TAIL)                                                                                         \
    REP8(asm volatile(INS " %0, %0, %8" TAIL "\n" INS " %1, %1, %8" TAIL "\n" INS " %2, %2, %8" TAIL "\n" INS    \
                          " %3, %3, %8" TAIL "\n" INS " %4, %4, %8" TAIL "\n" INS " %5, %5, %8" TAIL "\n" INS    \
                          " %6, %6, %8" TAIL "\n" INS " %7, %7, %8" TAIL "\n"                                    \
                      : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7)          \
                      : "v"(ka), "v"(kb));)
            if (OP == OP_PKFMA) { PK8("v_pk_fma_f32", ", %9") }
            else if (OP == OP_PKMUL) { PK8("v_pk_mul_f32", "") }
            else { PK8("v_pk_add_f32", "") }
        } else if (OP == OP_EXP || OP == OP_RCP) {
#define TR8(INS)                                                                                               \
    REP8(asm volatile(INS " %0, %0\n" INS " %1, %1\n" INS " %2, %2\n" INS " %3, %3\n" INS " %4, %4\n" INS        \
                          " %5, %5\n" INS " %6, %6\n" INS " %7, %7\n"                                            \
                      : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3), "+v"(s4), "+v"(s5), "+v"(s6), "+v"(s7));)
            if (OP == OP_EXP) { TR8("v_exp_f32") } else { TR8("v_rcp_f32") }
        } else if (OP == OP_DPPADD) {
            REP8(asm volatile("v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                              "v_add_f32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                              "v_add_f32_dpp %2, %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                              "v_add_f32_dpp %3, %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                              "v_add_f32_dpp %4, %4, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                              "v_add_f32_dpp %5, %5, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                              "v_add_f32_dpp %6, %6, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                              "v_add_f32_dpp %7, %7, %7 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                              : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3), "+v"(s4), "+v"(s5), "+v"(s6), "+v"(s7));)
        } else if (OP == OP_FMA64) {
            const double da = 0.999, db = 1e-3;
            REP8(asm volatile("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n"
                              "v_fma_f64 %3, %3, %8, %9\n v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n"
                              "v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n"
                              : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7)
                              : "v"(da), "v"(db));)
        } else if (OP == OP_MFMA) {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                m0 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, m0, 0, 0, 0);
                m1 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, m1, 0, 0, 0);
                m2 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, m2, 0, 0, 0);
                m3 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, m3, 0, 0, 0);
            }
        } else if (OP == OP_MFMA_PKFMA) {
            // 8 mfma, each followed by 8 independent packed fmas: does the VALU work hide under the matrix pipe?
#pragma unroll
            for (int u = 0; u < 2; ++u) {
#define MF(ACC)                                                                                                 \
    ACC = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, ACC, 0, 0, 0);                                           \
    asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n"      \
                 "v_pk_fma_f32 %3, %3, %8, %9\n v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n"      \
                 "v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"                                    \
                 : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7)               \
                 : "v"(ka), "v"(kb));
                MF(m0) MF(m1) MF(m2) MF(m3)
            }
        } else if (OP == OP_MIX_PK_SP) {
            // do packed (64-bit datapath) and plain fp32 operations share one pipe?
            REP8(asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_fma_f32 %4, %4, %10, %11\n v_pk_fma_f32 %1, %1, %8, %9\n"
                              "v_fma_f32 %5, %5, %10, %11\n v_pk_fma_f32 %2, %2, %8, %9\n v_fma_f32 %6, %6, %10, %11\n"
                              "v_pk_fma_f32 %3, %3, %8, %9\n v_fma_f32 %7, %7, %10, %11\n"
                              : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(s4), "+v"(s5), "+v"(s6), "+v"(s7)
                              : "v"(ka), "v"(kb), "v"(fa), "v"(fb));)
        } else if (OP == OP_MIX_SP_EXP) {
            REP8(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n"
                              "v_exp_f32 %6, %6\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n"
                              "v_fma_f32 %5, %5, %8, %9\n v_exp_f32 %7, %7\n"
                              : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3), "+v"(s4), "+v"(s5), "+v"(s6), "+v"(s7)
                              : "v"(fa), "v"(fb));)
        } else if (OP == OP_MIX_FUSED) {
            // the instruction mix of one component pair of flat_fused_pk_kernel, 4 pairs per body
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                asm volatile(
                    "v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n"
                    "v_pk_mul_f32 %3, %0, %8\n v_pk_mul_f32 %4, %1, %8\n v_pk_mul_f32 %5, %2, %8\n"
                    "v_pk_fma_f32 %6, %3, %0, %9\n v_pk_fma_f32 %6, %4, %1, %6\n v_pk_fma_f32 %6, %5, %2, %6\n"
                    "v_exp_f32 %10, %10\n v_exp_f32 %11, %11\n"
                    "v_pk_add_f32 %7, %7, %6\n"
                    "v_pk_mul_f32 %6, %6, %8\n"
                    "v_pk_add_f32 %0, %0, %9\n v_pk_add_f32 %1, %1, %9\n v_pk_add_f32 %2, %2, %9\n"
                    "v_pk_mul_f32 %3, %6, %0\n v_pk_mul_f32 %4, %6, %1\n v_pk_mul_f32 %5, %6, %2\n"
                    "v_pk_add_f32 %7, %7, %6\n"
                    "v_pk_add_f32 %7, %7, %3\n v_pk_add_f32 %7, %7, %4\n v_pk_add_f32 %7, %7, %5\n"
                    "v_pk_fma_f32 %7, %3, %0, %7\n v_pk_fma_f32 %7, %4, %1, %7\n v_pk_fma_f32 %7, %5, %2, %7\n"
                    : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7)
                    : "v"(ka), "v"(kb), "v"(s0), "v"(s1));
            }
        }
    }
    const long long t1 = clock64();
    float r = s0 + s1 + s2 + s3 + s4 + s5 + s6 + s7 + p0.x + p1.x + p2.x + p3.x + p4.y + p5.y + p6.y + p7.y +
              (float)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7) + m0.x + m1.y + m2.z + m3.w;
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int OP>
static void run(int blocks, int iters, float* out, long long* cyc) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    probe<OP><<<blocks, 256>>>(out, cyc, iters);
    hipEventRecord(a);
    probe<OP><<<blocks, 256>>>(out, cyc, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    std::vector<long long> h(blocks * 4);
    hipMemcpy(h.data(), cyc, sizeof(long long) * h.size(), hipMemcpyDeviceToHost);
    double mean = 0;
    for (auto v : h) mean += (double)v;
    mean /= (double)h.size();
    const double instr = (double)INSTR_PER_BODY[OP] * iters;
    // s_memtime ticks at a fixed 100 MHz on gfx9 (REFCLK), so convert through wall time as well
    const double waves_per_simd = blocks / 256.0;
    const double wall_cyc_24 = ms * 1e-3 * 2.4e9;
    printf("%-48s waves/SIMD %.0f  %8.3f ms  clock64/instr %7.3f  wall-cycles@2.4GHz per instr per SIMD %6.2f\n",
           NAMES[OP], waves_per_simd, ms, mean / instr, wall_cyc_24 / (instr * waves_per_simd));
    hipEventDestroy(a);
    hipEventDestroy(b);
}


// ---- cross-wave overlap of the matrix pipe with the vector pipe ----------------------------------------------------
// 512-thread workgroups, one per CU: waves 0-3 (one per SIMD) run stream A, waves 4-7 stream B (or nothing).  If the
// two pipes of a SIMD work concurrently for DIFFERENT waves, T(A and B) ~ max(T(A), T(B)); if they serialise, the sum.
//   PAIR 0: A = v_mfma_f32_4x4x1_16b_f32 (the shape a fused-kernel moment update would use), B = v_pk_fma_f32
//   PAIR 1: A = v_mfma_f32_16x16x4_f32,                                                      B = v_pk_fma_f32
//   PAIR 2: A = v_mfma_f64_16x16x4_f64 (full-covariance statistics),                         B = v_fma_f64
typedef double d4 __attribute__((ext_vector_type(4)));
template <int PAIR>
__global__ __launch_bounds__(512) void probe_split(float* out, long long* cyc, int iters, int run_a, int run_b) {
    const int w = threadIdx.x >> 6;
    float s0 = threadIdx.x * 1e-3f;
    f2 p0 = {s0, s0 + 1.f}, p1 = {s0 + 2.f, s0}, p2 = p0, p3 = p1, p4 = p0, p5 = p1, p6 = p0, p7 = p1;
    const f2 ka = {0.999f, 1.001f}, kb = {1e-3f, -1e-3f};
    double d0 = s0, d1 = s0 + 1, d2 = s0 + 2, d3 = s0 + 3, d4v = s0 + 4, d5 = s0 + 5, d6 = s0 + 6, d7 = s0 + 7;
    f4 m0 = {0, 0, 0, 0}, m1 = m0, m2 = m0, m3 = m0;
    d4 q0 = {0, 0, 0, 0}, q1 = q0, q2 = q0, q3 = q0;
    double r0 = 0, r1 = 0, r2 = 0, r3 = 0;
    const float fa = 0.999f, fb = 1e-3f;
    const double da = 0.999, db = 1e-3;
    const long long t0 = clock64();
    if (w < 4) {
        if (run_a)
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (PAIR == 0) {
                        m0 = __builtin_amdgcn_mfma_f32_4x4x1f32(fa, fb, m0, 0, 0, 0);
                        m1 = __builtin_amdgcn_mfma_f32_4x4x1f32(fa, fb, m1, 0, 0, 0);
                        m2 = __builtin_amdgcn_mfma_f32_4x4x1f32(fa, fb, m2, 0, 0, 0);
                        m3 = __builtin_amdgcn_mfma_f32_4x4x1f32(fa, fb, m3, 0, 0, 0);
                    } else if (PAIR == 1) {
                        m0 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, m0, 0, 0, 0);
                        m1 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, m1, 0, 0, 0);
                        m2 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, m2, 0, 0, 0);
                        m3 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, m3, 0, 0, 0);
                    } else if (PAIR == 3) {
                        r0 = __builtin_amdgcn_mfma_f64_4x4x4f64(da, db, r0, 0, 0, 0);
                        r1 = __builtin_amdgcn_mfma_f64_4x4x4f64(da, db, r1, 0, 0, 0);
                        r2 = __builtin_amdgcn_mfma_f64_4x4x4f64(da, db, r2, 0, 0, 0);
                        r3 = __builtin_amdgcn_mfma_f64_4x4x4f64(da, db, r3, 0, 0, 0);
                    } else {
                        q0 = __builtin_amdgcn_mfma_f64_16x16x4f64(da, db, q0, 0, 0, 0);
                        q1 = __builtin_amdgcn_mfma_f64_16x16x4f64(da, db, q1, 0, 0, 0);
                        q2 = __builtin_amdgcn_mfma_f64_16x16x4f64(da, db, q2, 0, 0, 0);
                        q3 = __builtin_amdgcn_mfma_f64_16x16x4f64(da, db, q3, 0, 0, 0);
                    }
                }
            }
    } else if (run_b) {
        for (int it = 0; it < iters; ++it) {
            if (PAIR < 2) {
                REP8(asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n"
                                  "v_pk_fma_f32 %3, %3, %8, %9\n v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n"
                                  "v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
                                  : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7)
                                  : "v"(ka), "v"(kb));)
            } else {
                REP8(asm volatile("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n"
                                  "v_fma_f64 %3, %3, %8, %9\n v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n"
                                  "v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n"
                                  : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4v), "+v"(d5), "+v"(d6), "+v"(d7)
                                  : "v"(da), "v"(db));)
            }
        }
    }
    const long long t1 = clock64();
    float r = p0.x + p1.x + p2.x + p3.x + p4.y + p5.y + p6.y + p7.y + m0.x + m1.y + m2.z + m3.w +
              (float)(d0 + d1 + d2 + d3 + d4v + d5 + d6 + d7 + q0.x + q1.y + q2.z + q3.w + r0 + r1 + r2 + r3);
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + w] = t1 - t0;
}

template <int PAIR>
static void run_split(float* out, long long* cyc, const char* name_a, int instr_a, const char* name_b) {
    const int iters = 2000;
    float ms[3];
    for (int mode = 0; mode < 3; ++mode) {                 // A alone, B alone, both
        const int ra = mode != 1, rb = mode != 0;
        hipEvent_t a, b;
        hipEventCreate(&a);
        hipEventCreate(&b);
        probe_split<PAIR><<<256, 512>>>(out, cyc, iters, ra, rb);
        hipEventRecord(a);
        probe_split<PAIR><<<256, 512>>>(out, cyc, iters, ra, rb);
        hipEventRecord(b);
        hipEventSynchronize(b);
        hipEventElapsedTime(&ms[mode], a, b);
        hipEventDestroy(a);
        hipEventDestroy(b);
    }
    printf("cross-wave overlap  A = %-28s (%d per wave)  B = %-14s (%d per wave):  A alone %.3f ms  B alone %.3f ms  "
           "A and B on different waves of every SIMD %.3f ms  -> %s (sum %.3f, max %.3f)\n",
           name_a, instr_a * iters, name_b, 64 * iters, ms[0], ms[1], ms[2],
           ms[2] < 0.5 * (ms[0] + ms[1] + (ms[0] > ms[1] ? ms[0] : ms[1])) ? "OVERLAP" : "SERIALISED",
           ms[0] + ms[1], ms[0] > ms[1] ? ms[0] : ms[1]);
}

int main() {
    float* out;
    long long* cyc;
    hipMalloc(&out, sizeof(float) * 1024 * 512);
    hipMalloc(&cyc, sizeof(long long) * 1024 * 8);
    if (getenv("VALUBENCH_SPLIT")) {
        for (int rep = 0; rep < 2; ++rep) {
            run_split<0>(out, cyc, "v_mfma_f32_4x4x1_16b_f32", 16, "v_pk_fma_f32");
            run_split<1>(out, cyc, "v_mfma_f32_16x16x4_f32", 16, "v_pk_fma_f32");
            run_split<2>(out, cyc, "v_mfma_f64_16x16x4_f64", 16, "v_fma_f64");
            run_split<3>(out, cyc, "v_mfma_f64_4x4x4_4b_f64", 16, "v_fma_f64");
        }
        return 0;
    }
    const int iters = 4000;
    if (getenv("VALUBENCH_MIX")) {                 // the pipe-sharing questions only, 1..4 waves per SIMD
        for (int blocks : {256, 512, 768, 1024}) {
            run<OP_FMA>(blocks, iters, out, cyc);
            run<OP_PKFMA>(blocks, iters, out, cyc);
            run<OP_EXP>(blocks, iters, out, cyc);
            run<OP_MIX_PK_SP>(blocks, iters, out, cyc);
            run<OP_MIX_SP_EXP>(blocks, iters, out, cyc);
            run<OP_MIX_FUSED>(blocks, iters, out, cyc);
        }
        return 0;
    }
    for (int rep = 0; rep < 2; ++rep)
        for (int blocks : {256, 512}) {
            run<OP_FMA>(blocks, iters, out, cyc);
            run<OP_ADD>(blocks, iters, out, cyc);
            run<OP_FMA_DEP>(blocks, iters, out, cyc);
            run<OP_PKFMA>(blocks, iters, out, cyc);
            run<OP_PKFMA_DEP>(blocks, iters, out, cyc);
            run<OP_PKMUL>(blocks, iters, out, cyc);
            run<OP_PKADD>(blocks, iters, out, cyc);
            run<OP_EXP>(blocks, iters, out, cyc);
            run<OP_RCP>(blocks, iters, out, cyc);
            run<OP_DPPADD>(blocks, iters, out, cyc);
            run<OP_FMA64>(blocks, iters, out, cyc);
            run<OP_MFMA>(blocks, iters, out, cyc);
            run<OP_MFMA_PKFMA>(blocks, iters, out, cyc);
            run<OP_MIX_FUSED>(blocks, iters, out, cyc);
        }
    return 0;
}
