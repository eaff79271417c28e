// Wave64 cross-lane primitives for gfx950 (CDNA4).  DPP-only reductions: no LDS traffic,
// no ds_bpermute.  A CDNA wavefront is 64 lanes = 4 DPP rows of 16.
#pragma once
#include <hip/hip_runtime.h>

namespace hgmm {

// DPP control words (GFX9/CDNA encoding)
constexpr int DPP_QUAD_XOR1 = 0xB1;       // quad_perm:[1,0,3,2]
constexpr int DPP_QUAD_XOR2 = 0x4E;       // quad_perm:[2,3,0,1]
constexpr int DPP_ROW_HALF_MIRROR = 0x141;
constexpr int DPP_ROW_MIRROR = 0x140;
constexpr int DPP_ROW_BCAST15 = 0x142;    // lane 15 of each row -> next row
constexpr int DPP_ROW_BCAST31 = 0x143;    // lane 31 -> rows 2,3

template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ float dpp_f32(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL,
                                                      ROW_MASK, 0xF, false));
}
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ int dpp_i32(int v) {
    return __builtin_amdgcn_update_dpp(v, v, CTRL, ROW_MASK, 0xF, false);
}

struct OpMax { __device__ __forceinline__ float operator()(float a, float b) const { return fmaxf(a, b); } };
struct OpSum { __device__ __forceinline__ float operator()(float a, float b) const { return a + b; } };
struct OpMinI { __device__ __forceinline__ int operator()(int a, int b) const { return a < b ? a : b; } };

// Reduce over the 64 lanes; result is wave-uniform (returned through an SGPR read of lane 63).
template <class Op>
__device__ __forceinline__ float wave_reduce(float v, Op op) {
    v = op(v, dpp_f32<DPP_QUAD_XOR1>(v));
    v = op(v, dpp_f32<DPP_QUAD_XOR2>(v));
    v = op(v, dpp_f32<DPP_ROW_HALF_MIRROR>(v));
    v = op(v, dpp_f32<DPP_ROW_MIRROR>(v));              // every lane: its row's (16) result
    v = op(v, dpp_f32<DPP_ROW_BCAST15, 0xA>(v));        // rows 1,3 += rows 0,2
    v = op(v, dpp_f32<DPP_ROW_BCAST31, 0xC>(v));        // rows 2,3 += row 1 (which holds 0+1)
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
// ---- single-instruction DPP steps --------------------------------------------------------------
// hipcc lowers `op(v, update_dpp(v))` to v_mov_b32 + v_mov_b32_dpp + v_add/v_max (3 VALU instructions per
// step, 18 per wave reduction).  The ISA folds the lane permutation into the arithmetic instruction
// (v_add_f32_dpp / v_max_f32_dpp: 1 instruction per step); lanes a row_mask excludes keep their value,
// which is exactly the bcast15 / bcast31 tail of the reduction.  Inline asm is opaque to the
// compiler's hazard recogniser, so the wait states the ISA requires are written out: a DPP operand
// written by the previous VALU instruction needs 2 (s_nop 1), v_readlane of a just-written VGPR 1,
// an SGPR written by v_readlane and read by the next VALU instruction 2.
#define HGMM_DPP_CHAIN(INS)                                                                        \
    "s_nop 1\n" INS " %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"                  \
    "s_nop 1\n" INS " %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"                  \
    "s_nop 1\n" INS " %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n"                      \
    "s_nop 1\n" INS " %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n"                           \
    "s_nop 1\n" INS " %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n"                         \
    "s_nop 1\n" INS " %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n"                         \
    "s_nop 0\n v_readlane_b32 %1, %0, 63\n s_nop 1\n"

// same summation / maximum order as wave_reduce (bitwise identical results)
__device__ __forceinline__ float wave_sum_dpp(float v) {
    float r;
    asm volatile(HGMM_DPP_CHAIN("v_add_f32_dpp") : "+v"(v), "=s"(r));
    return r;
}
__device__ __forceinline__ float wave_max_dpp(float v) {
    float r;
    asm volatile(HGMM_DPP_CHAIN("v_max_f32_dpp") : "+v"(v), "=s"(r));
    return r;
}

// four independent reductions interleaved: the three instructions between two dependent steps cover
// the DPP wait states, so a wave that runs alone on its SIMD does not stall on s_nop
#define HGMM_DPP_STEP4(INS, CTRL)                                                                  \
    INS " %0, %0, %0 " CTRL "\n" INS " %1, %1, %1 " CTRL "\n" INS " %2, %2, %2 " CTRL "\n" INS       \
        " %3, %3, %3 " CTRL "\n"
#define HGMM_DPP_CHAIN4(INS)                                                                       \
    "s_nop 1\n" HGMM_DPP_STEP4(INS, "quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")               \
    HGMM_DPP_STEP4(INS, "quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")                           \
    HGMM_DPP_STEP4(INS, "row_half_mirror row_mask:0xf bank_mask:0xf")                               \
    HGMM_DPP_STEP4(INS, "row_mirror row_mask:0xf bank_mask:0xf")                                    \
    HGMM_DPP_STEP4(INS, "row_bcast:15 row_mask:0xa bank_mask:0xf")                                  \
    HGMM_DPP_STEP4(INS, "row_bcast:31 row_mask:0xc bank_mask:0xf")                                  \
    "v_readlane_b32 %4, %0, 63\n v_readlane_b32 %5, %1, 63\n v_readlane_b32 %6, %2, 63\n"            \
    "v_readlane_b32 %7, %3, 63\n s_nop 1\n"
__device__ __forceinline__ void wave_sum4_dpp(float (&v)[4]) {
    float r0, r1, r2, r3;
    asm volatile(HGMM_DPP_CHAIN4("v_add_f32_dpp")
                 : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "=s"(r0), "=s"(r1), "=s"(r2), "=s"(r3));
    v[0] = r0; v[1] = r1; v[2] = r2; v[3] = r3;
}
__device__ __forceinline__ void wave_max4_dpp(float (&v)[4]) {
    float r0, r1, r2, r3;
    asm volatile(HGMM_DPP_CHAIN4("v_max_f32_dpp")
                 : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "=s"(r0), "=s"(r1), "=s"(r2), "=s"(r3));
    v[0] = r0; v[1] = r1; v[2] = r2; v[3] = r3;
}

template <class Op>
__device__ __forceinline__ int wave_reduce_i(int v, Op op) {
    v = op(v, dpp_i32<DPP_QUAD_XOR1>(v));
    v = op(v, dpp_i32<DPP_QUAD_XOR2>(v));
    v = op(v, dpp_i32<DPP_ROW_HALF_MIRROR>(v));
    v = op(v, dpp_i32<DPP_ROW_MIRROR>(v));
    v = op(v, dpp_i32<DPP_ROW_BCAST15, 0xA>(v));
    v = op(v, dpp_i32<DPP_ROW_BCAST31, 0xC>(v));
    return __builtin_amdgcn_readlane(v, 63);
}

// double-precision sum over the wave (two 32-bit DPP moves per step)
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ double dpp_f64(double v) {
    long long b = __double_as_longlong(v);
    int lo = (int)(b & 0xffffffffLL), hi = (int)(b >> 32);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, ROW_MASK, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, ROW_MASK, 0xF, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double wave_sum_f64(double v) {
    v += dpp_f64<DPP_QUAD_XOR1>(v);
    v += dpp_f64<DPP_QUAD_XOR2>(v);
    v += dpp_f64<DPP_ROW_HALF_MIRROR>(v);
    v += dpp_f64<DPP_ROW_MIRROR>(v);
    v += dpp_f64<DPP_ROW_BCAST15, 0xA>(v);
    v += dpp_f64<DPP_ROW_BCAST31, 0xC>(v);
    long long b = __double_as_longlong(v);
    int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffLL), 63);
    int hi = __builtin_amdgcn_readlane((int)(b >> 32), 63);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

__device__ __forceinline__ double wave_max_f64(double v) {
    v = fmax(v, dpp_f64<DPP_QUAD_XOR1>(v));
    v = fmax(v, dpp_f64<DPP_QUAD_XOR2>(v));
    v = fmax(v, dpp_f64<DPP_ROW_HALF_MIRROR>(v));
    v = fmax(v, dpp_f64<DPP_ROW_MIRROR>(v));
    v = fmax(v, dpp_f64<DPP_ROW_BCAST15, 0xA>(v));
    v = fmax(v, dpp_f64<DPP_ROW_BCAST31, 0xC>(v));
    long long b = __double_as_longlong(v);
    int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffLL), 63);
    int hi = __builtin_amdgcn_readlane((int)(b >> 32), 63);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// Reductions over the two 32-lane halves of a wave at once (5 DPP steps instead of 6; lanes 31 and 63 end up
// with their half's result): lo = lanes 0..31, hi = lanes 32..63.  Same combination order in both halves.
__device__ __forceinline__ double readlane_f64(double v, int l) {
    long long b = __double_as_longlong(v);
    int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffLL), l);
    int hi = __builtin_amdgcn_readlane((int)(b >> 32), l);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ void halfwave_sum_f64(double v, double& lo, double& hi) {
    v += dpp_f64<DPP_QUAD_XOR1>(v);
    v += dpp_f64<DPP_QUAD_XOR2>(v);
    v += dpp_f64<DPP_ROW_HALF_MIRROR>(v);
    v += dpp_f64<DPP_ROW_MIRROR>(v);
    v += dpp_f64<DPP_ROW_BCAST15, 0xA>(v);
    lo = readlane_f64(v, 31);
    hi = readlane_f64(v, 63);
}
__device__ __forceinline__ void halfwave_max_f64(double v, double& lo, double& hi) {
    v = fmax(v, dpp_f64<DPP_QUAD_XOR1>(v));
    v = fmax(v, dpp_f64<DPP_QUAD_XOR2>(v));
    v = fmax(v, dpp_f64<DPP_ROW_HALF_MIRROR>(v));
    v = fmax(v, dpp_f64<DPP_ROW_MIRROR>(v));
    v = fmax(v, dpp_f64<DPP_ROW_BCAST15, 0xA>(v));
    lo = readlane_f64(v, 31);
    hi = readlane_f64(v, 63);
}
__device__ __forceinline__ void halfwave_min_i32(int v, int& lo, int& hi) {
    OpMinI op;
    v = op(v, dpp_i32<DPP_QUAD_XOR1>(v));
    v = op(v, dpp_i32<DPP_QUAD_XOR2>(v));
    v = op(v, dpp_i32<DPP_ROW_HALF_MIRROR>(v));
    v = op(v, dpp_i32<DPP_ROW_MIRROR>(v));
    v = op(v, dpp_i32<DPP_ROW_BCAST15, 0xA>(v));
    lo = __builtin_amdgcn_readlane(v, 31);
    hi = __builtin_amdgcn_readlane(v, 63);
}

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63u); }
__device__ __forceinline__ int wave_in_block() {
    return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
}

// Per-wave store schedule on the constant-rate wall clock (flat_kernels.hip, "Store pacing", explains why).
struct StorePacer {
    unsigned long long next;
    unsigned period16, frac;           // period in 1/16 ticks of the wall clock; 0 = no pacing
    // wave `gw` of `nw`: the waves' schedules are staggered evenly over one period -- started in phase, all of them would
    // store at the same moments, a burst per period
    __device__ __forceinline__ StorePacer(int p16, long long gw, long long nw) : next(0), period16((unsigned)p16), frac(0) {
        if (period16) next = wall_clock64() + (unsigned long long)(((unsigned long long)period16 * (unsigned long long)gw / (unsigned long long)nw) >> 4);
    }
    // call right before a group's stores
    __device__ __forceinline__ void wait() {
        if (!period16) return;
        unsigned long long now = wall_clock64();
        while (now < next) {
            __builtin_amdgcn_s_sleep(1);
            now = wall_clock64();
        }
        // a wave that has fallen behind by more than a period (its arithmetic could not keep up: low clocks, a late
        // start) is re-anchored instead of catching up in a burst -- bursts are what the write path punishes
        const unsigned period = period16 >> 4;
        if (now > next + period) next = now;
        frac += period16;
        next += frac >> 4;
        frac &= 15u;
    }
};

}  // namespace hgmm
