// Shared helpers for the libcwfa_hip.so translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "cwfa_hip.h"

void cwfa_set_error(const char* fmt, ...);

#define CWFA_REQUIRE(cond, code, ...)        \
    do {                                     \
        if (!(cond)) {                       \
            cwfa_set_error(__VA_ARGS__);     \
            return (code);                   \
        }                                    \
    } while (0)

#define CWFA_LAUNCH_CHECK(name)                                              \
    do {                                                                     \
        hipError_t e_ = hipGetLastError();                                   \
        if (e_ != hipSuccess) {                                              \
            cwfa_set_error("%s: launch failed: %s", name, hipGetErrorString(e_)); \
            return CWFA_E_HIP;                                               \
        }                                                                    \
    } while (0)

static inline bool cwfa_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// fp32 constant the reference multiplies with: python float 1/math.sqrt(2) rounded to fp32 (INN_utils.py:150,161)
#define CWFA_INV_SQRT2_F 0.70710678118654752440f
// fp32(math.sqrt(2)) used as a divisor in networks.py:671
#define CWFA_SQRT2_F 1.41421356237309504880f

// wave64 sum, result valid in every lane
__device__ __forceinline__ double cwfa_wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float cwfa_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// sum over the 16 lanes of a DPP row (lanes with equal lane >> 4), result valid in all 16: quad xor 1, quad xor 2, mirror of the
// half row, mirror of the row -- four VALU instructions, no LDS
__device__ __forceinline__ float cwfa_row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
    return v;
}

// 16-byte buffer store with an SGPR offset.  gfx950: the store keeps reading its data registers after issue, also in this
// addressing form, which LLVM's hazard rule exempts ("only if soffset is not a register"): a VALU write into the tuple right
// behind it corrupted the stored data.  EVERY 16-byte buffer store of the library goes through
// this helper, which carries the wait states and keeps the scheduler from moving anything across.
typedef unsigned cwfa_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void cwfa_buffer_store_b128(cwfa_u32x4 data, __amdgpu_buffer_rsrc_t rsrc, unsigned voffset, int soffset) {
    __builtin_amdgcn_raw_buffer_store_b128(data, rsrc, voffset, soffset, 0);
    asm volatile("s_nop 1");
    __builtin_amdgcn_sched_barrier(0);
}

// block-wide sum of doubles (blockDim.x multiple of 64, <= 1024); result valid in thread 0.
__device__ __forceinline__ double cwfa_block_sum(double v, double* lds /* >= 16 doubles */) {
    v = cwfa_wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    if (lane == 0) lds[wave] = v;
    __syncthreads();
    double r = 0.0;
    if (threadIdx.x == 0)
        for (int i = 0; i < nw; ++i) r += lds[i];
    __syncthreads();
    return r;
}

// ELU(v) = v > 0 ? v : expm1(v).  ocml's expm1f is ~30 VALU instructions, and on gfx950 every vector instruction of an
// fp32-MFMA kernel is time taken from the matrix pipe (they do not co-execute), so: hardware exp (v_exp_f32) minus one,
// 5 instructions.  Absolute error <= 1.2e-7 (one ulp of exp(v) ~ 1); the relative error of tiny negative results is
// not bounded, which the 1e-4 parity bound (relative to the tensor's max) does not need.
__device__ __forceinline__ float cwfa_elu(float v) {
    const float e = __expf(v) - 1.0f;
    return v > 0.f ? v : e;
}
// atan for the soft clamp of the couplings (coupling_layers.py:52: 0.636 * atan(s / clamp) ...).  ocml's atanf is ~30 vector
// instructions (an IEEE division in its range reduction); the fused chain kernels evaluate it five times per element and
// were VALU-bound on it.  Here: r = |x| or 1/|x| (v_rcp_f32), atan(r) = r * P(r^2) with a degree-7 Chebyshev-interpolated
// P on [0, 1], pi/2 - . for |x| > 1, sign restored: 15 instructions, absolute error <= 1.9e-7 (3 ulp at pi/2) over all x.
__device__ __forceinline__ float cwfa_atan(float x) {
    const float ax = fabsf(x);
    const bool inv = ax > 1.0f;
    const float r = inv ? __builtin_amdgcn_rcpf(ax) : ax;
    const float t = r * r;
    float p = -0.00455979211255908f;
    p = fmaf(p, t, 0.023780519142746925f);
    p = fmaf(p, t, -0.05882975459098816f);
    p = fmaf(p, t, 0.09868865460157394f);
    p = fmaf(p, t, -0.14003290235996246f);
    p = fmaf(p, t, 0.19966961443424225f);
    p = fmaf(p, t, -0.3333181142807007f);
    p = fmaf(p, t, 0.9999998807907104f);
    float a = p * r;
    a = inv ? 1.57079632679489661923f - a : a;
    return copysignf(a, x);
}

// tanh for the TANH soft clamp (AllInOneBlock, all_in_one_block.py:216): 1 - 2 / (exp(2x) + 1) on the hardware exp and reciprocal
// for |x| >= 0.35 (absolute error <= 2.5e-7; exp saturates cleanly: +-1 for |x| > 44), and the odd series x P(x^2) below that:
// the first form cancels for small |x| (relative error ~1e-3 at |x| = 1e-4, where freshly initialised AllInOne blocks sit: their
// summed log-det would carry that floor).  Series truncated after x^9: next term 1382/155925 x^11 < 9e-8 x at 0.35.
__device__ __forceinline__ float cwfa_tanh(float x) {
    const float e = __expf(2.0f * x);
    const float big = 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
    const float t = x * x;
    float p = 62.0f / 2835.0f;
    p = fmaf(p, t, -17.0f / 315.0f);
    p = fmaf(p, t, 2.0f / 15.0f);
    p = fmaf(p, t, -1.0f / 3.0f);
    p = fmaf(p, t, 1.0f);
    return fabsf(x) < 0.35f ? x * p : big;
}

__device__ __forceinline__ float cwfa_gelu(float v) { return 0.5f * v * (1.f + erff(v * 0.70710678118654752440f)); }

__device__ __forceinline__ float cwfa_act(float v, int act, float alpha) {
    switch (act) {
        case CWFA_ACT_ELU: return cwfa_elu(v);
        case CWFA_ACT_PRELU: return v > 0.f ? v : alpha * v;
        case CWFA_ACT_GELU: return cwfa_gelu(v);
        case CWFA_ACT_RELU: return v > 0.f ? v : 0.f;
        default: return v;
    }
}
