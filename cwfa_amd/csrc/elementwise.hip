// HBM-bound streaming kernels of the flow stack: Haar wavelets, index gathers, affine coupling apply,
// and the fused per-step chains.  Compiled with -ffp-contract=off so that the two-op sequences of the
// reference ((a+b)*f, exp(s)*x+t, (x-t)*exp(-s)) round exactly like the reference's separate torch ops.
#include "common.h"

#include <stdarg.h>
#include <stdio.h>

// ------------------------------------------------------------------------------------------------ error plumbing
static thread_local char g_err[512] = "";

void cwfa_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int cwfa_version(void) { return CWFA_VERSION; }
extern "C" const char* cwfa_last_error(void) { return g_err; }

// ------------------------------------------------------------------------------------------------ Haar 1-D (depth)
// One thread = VEC consecutive pixels of one channel PAIR: 2 loads + 2 stores of VEC*4 bytes each, all coalesced.
// Algorithmic traffic: 8 bytes per input element (read once, write once) -- the HBM roofline of DESIGN.md.
template <int VEC>
__global__ __launch_bounds__(256) void haar1d_fwd_kernel(const float* __restrict__ x, float* __restrict__ lo,
                                                         float* __restrict__ hi, int h, int64_t HWv, int64_t HW,
                                                         int64_t x_bs, int64_t lo_bs, int64_t hi_bs) {
    typedef float vec_t __attribute__((ext_vector_type(VEC)));
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= HWv) return;
    const int i = blockIdx.y, b = blockIdx.z;
    const vec_t e = *reinterpret_cast<const vec_t*>(x + b * x_bs + (int64_t)(2 * i) * HW + p * VEC);
    const vec_t o = *reinterpret_cast<const vec_t*>(x + b * x_bs + (int64_t)(2 * i + 1) * HW + p * VEC);
    *reinterpret_cast<vec_t*>(lo + b * lo_bs + (int64_t)i * HW + p * VEC) = (e + o) * CWFA_INV_SQRT2_F;
    *reinterpret_cast<vec_t*>(hi + b * hi_bs + (int64_t)i * HW + p * VEC) = (e - o) * CWFA_INV_SQRT2_F;
}

template <int VEC>
__global__ __launch_bounds__(256) void haar1d_inv_kernel(const float* __restrict__ lo, const float* __restrict__ hi,
                                                         float* __restrict__ x, int h, int64_t HWv, int64_t HW,
                                                         int64_t lo_bs, int64_t hi_bs, int64_t x_bs) {
    typedef float vec_t __attribute__((ext_vector_type(VEC)));
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= HWv) return;
    const int i = blockIdx.y, b = blockIdx.z;
    const vec_t l = *reinterpret_cast<const vec_t*>(lo + b * lo_bs + (int64_t)i * HW + p * VEC);
    vec_t d = (vec_t)(0.f);
    if (hi) d = *reinterpret_cast<const vec_t*>(hi + b * hi_bs + (int64_t)i * HW + p * VEC);
    *reinterpret_cast<vec_t*>(x + b * x_bs + (int64_t)(2 * i) * HW + p * VEC) = (l + d) * CWFA_INV_SQRT2_F;
    *reinterpret_cast<vec_t*>(x + b * x_bs + (int64_t)(2 * i + 1) * HW + p * VEC) = (l - d) * CWFA_INV_SQRT2_F;
}

static int haar1d_check(const char* name, const void* a, const void* b, const void* c, int B, int D, int64_t HW) {
    CWFA_REQUIRE(B >= 0 && D >= 0 && HW >= 0, CWFA_E_INVAL, "%s: negative size", name);
    if (B == 0 || D == 0 || HW == 0) return CWFA_OK;        // empty tensors legitimately carry null data pointers
    CWFA_REQUIRE(a && c, CWFA_E_INVAL, "%s: null pointer", name);
    CWFA_REQUIRE(D % 2 == 0, CWFA_E_SHAPE, "%s: depth %d is odd", name, D);
    CWFA_REQUIRE(D / 2 <= 65535 && B <= 65535, CWFA_E_SHAPE, "%s: grid too large (D/2=%d, B=%d)", name, D / 2, B);
    (void)b;
    return CWFA_OK;
}

extern "C" int cwfa_haar1d_fwd_f32(const float* x, float* lo, float* hi, int B, int D, int64_t HW, int64_t x_bs,
                                   int64_t lo_bs, int64_t hi_bs, void* stream) {
    int rc = haar1d_check("cwfa_haar1d_fwd_f32", x, lo, hi, B, D, HW);
    if (rc) return rc;
    if (B == 0 || D == 0 || HW == 0) return CWFA_OK;
    CWFA_REQUIRE(lo && hi, CWFA_E_INVAL, "cwfa_haar1d_fwd_f32: null output");
    const int h = D / 2;
    const bool v4 = HW % 4 == 0 && x_bs % 4 == 0 && lo_bs % 4 == 0 && hi_bs % 4 == 0 && cwfa_aligned16(x) &&
                    cwfa_aligned16(lo) && cwfa_aligned16(hi);
    const int64_t HWv = v4 ? HW / 4 : HW;
    dim3 grid((unsigned)((HWv + 255) / 256), h, B);
    if (v4)
        hipLaunchKernelGGL(haar1d_fwd_kernel<4>, grid, dim3(256), 0, (hipStream_t)stream, x, lo, hi, h, HWv, HW, x_bs,
                           lo_bs, hi_bs);
    else
        hipLaunchKernelGGL(haar1d_fwd_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, x, lo, hi, h, HWv, HW, x_bs,
                           lo_bs, hi_bs);
    CWFA_LAUNCH_CHECK("cwfa_haar1d_fwd_f32");
    return CWFA_OK;
}

extern "C" int cwfa_haar1d_inv_f32(const float* lo, const float* hi, float* x, int B, int D, int64_t HW, int64_t lo_bs,
                                   int64_t hi_bs, int64_t x_bs, void* stream) {
    int rc = haar1d_check("cwfa_haar1d_inv_f32", lo, hi, x, B, D, HW);
    if (rc) return rc;
    if (B == 0 || D == 0 || HW == 0) return CWFA_OK;
    const int h = D / 2;
    const bool v4 = HW % 4 == 0 && x_bs % 4 == 0 && lo_bs % 4 == 0 && (hi == nullptr || hi_bs % 4 == 0) &&
                    cwfa_aligned16(x) && cwfa_aligned16(lo) && cwfa_aligned16(hi);
    const int64_t HWv = v4 ? HW / 4 : HW;
    dim3 grid((unsigned)((HWv + 255) / 256), h, B);
    if (v4)
        hipLaunchKernelGGL(haar1d_inv_kernel<4>, grid, dim3(256), 0, (hipStream_t)stream, lo, hi, x, h, HWv, HW, lo_bs,
                           hi_bs, x_bs);
    else
        hipLaunchKernelGGL(haar1d_inv_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, lo, hi, x, h, HWv, HW, lo_bs,
                           hi_bs, x_bs);
    CWFA_LAUNCH_CHECK("cwfa_haar1d_inv_f32");
    return CWFA_OK;
}

// ------------------------------------------------------------------------------------------------ Haar 2-D (spatial)
// One thread = one 2x2 input patch of one channel = one output pixel of the 4 wavelet channels.
// The sum order matches a 2x2 conv's natural accumulation (p00, p01, p10, p11) of reshapes.py:282.
__global__ __launch_bounds__(256) void haar2d_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int C, int H,
                                                         int W, int obw, float fac) {
    const int h2 = H / 2, w2 = W / 2;
    const int64_t n = (int64_t)C * h2 * w2;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int b = blockIdx.y;
    const int wx = (int)(i % w2), hy = (int)((i / w2) % h2), c = (int)(i / ((int64_t)w2 * h2));
    const float* px = x + ((int64_t)b * C + c) * H * W + (int64_t)(2 * hy) * W + 2 * wx;
    const float2 r0 = *reinterpret_cast<const float2*>(px);
    const float2 r1 = *reinterpret_cast<const float2*>(px + W);
    const float a = ((r0.x + r0.y) + r1.x) + r1.y;
    const float v1 = ((r0.x - r0.y) + r1.x) - r1.y;
    const float v2 = ((r0.x + r0.y) - r1.x) - r1.y;
    const float d = ((r0.x - r0.y) - r1.x) + r1.y;
    const int64_t plane = (int64_t)h2 * w2, pos = (int64_t)hy * w2 + wx;
    float* py = y + (int64_t)b * 4 * C * plane + pos;
    const int c0 = obw ? c : 4 * c, cs = obw ? C : 1;
    py[(int64_t)(c0)*plane] = a * fac;
    py[(int64_t)(c0 + cs) * plane] = v1 * fac;
    py[(int64_t)(c0 + 2 * cs) * plane] = v2 * fac;
    py[(int64_t)(c0 + 3 * cs) * plane] = d * fac;
}

__global__ __launch_bounds__(256) void haar2d_inv_kernel(const float* __restrict__ y, float* __restrict__ x, int C, int H,
                                                         int W, int obw, float fac) {
    const int h2 = H / 2, w2 = W / 2;
    const int64_t n = (int64_t)C * h2 * w2;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int b = blockIdx.y;
    const int wx = (int)(i % w2), hy = (int)((i / w2) % h2), c = (int)(i / ((int64_t)w2 * h2));
    const int64_t plane = (int64_t)h2 * w2, pos = (int64_t)hy * w2 + wx;
    const float* py = y + (int64_t)b * 4 * C * plane + pos;
    const int c0 = obw ? c : 4 * c, cs = obw ? C : 1;
    const float a = py[(int64_t)(c0)*plane] * fac;
    const float v1 = py[(int64_t)(c0 + cs) * plane] * fac;
    const float v2 = py[(int64_t)(c0 + 2 * cs) * plane] * fac;
    const float d = py[(int64_t)(c0 + 3 * cs) * plane] * fac;
    float* px = x + ((int64_t)b * C + c) * H * W + (int64_t)(2 * hy) * W + 2 * wx;
    *reinterpret_cast<float2*>(px) = make_float2(((a + v1) + v2) + d, ((a - v1) + v2) - d);
    *reinterpret_cast<float2*>(px + W) = make_float2(((a + v1) - v2) - d, ((a - v1) - v2) + d);
}

static int haar2d_launch(bool fwd, const float* a, float* b, int B, int C, int H, int W, int obw, float fac,
                         void* stream) {
    const char* name = fwd ? "cwfa_haar2d_fwd_f32" : "cwfa_haar2d_inv_f32";
    CWFA_REQUIRE(a && b, CWFA_E_INVAL, "%s: null pointer", name);
    CWFA_REQUIRE(B >= 0 && C >= 0 && H >= 0 && W >= 0, CWFA_E_INVAL, "%s: negative size", name);
    CWFA_REQUIRE(H % 2 == 0 && W % 2 == 0, CWFA_E_SHAPE, "%s: H=%d, W=%d must be even", name, H, W);
    CWFA_REQUIRE(B <= 65535, CWFA_E_SHAPE, "%s: batch too large", name);
    if (B == 0 || C == 0 || H == 0 || W == 0) return CWFA_OK;
    const int64_t n = (int64_t)C * (H / 2) * (W / 2);
    dim3 grid((unsigned)((n + 255) / 256), B);
    if (fwd)
        hipLaunchKernelGGL(haar2d_fwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, a, b, C, H, W, obw, fac);
    else
        hipLaunchKernelGGL(haar2d_inv_kernel, grid, dim3(256), 0, (hipStream_t)stream, a, b, C, H, W, obw, fac);
    CWFA_LAUNCH_CHECK(name);
    return CWFA_OK;
}

extern "C" int cwfa_haar2d_fwd_f32(const float* x, float* y, int B, int C, int H, int W, int order_by_wavelet, float fac,
                                   void* stream) {
    return haar2d_launch(true, x, y, B, C, H, W, order_by_wavelet, fac, stream);
}
extern "C" int cwfa_haar2d_inv_f32(const float* y, float* x, int B, int C, int H, int W, int order_by_wavelet, float fac,
                                   void* stream) {
    return haar2d_launch(false, y, x, B, C, H, W, order_by_wavelet, fac, stream);
}

// ------------------------------------------------------------------------------------------------ Haar 3-D (2 x 2 x 2 tiles)
// The depth Haar (INN_utils.py:142-161) followed by the spatial Haar of every band (reshapes.py:273-300) in ONE pass:
// y = haar2d(haar1d(x)), [B,D,H,W] -> [B,4D,H/2,W/2].  One thread = one depth pair x NP adjacent 2x2 patches = NP 2x2x2
// tiles, kept in registers: 4 loads of 8*NP bytes, 8 stores of 4*NP bytes, every element read once and written once
// (8 bytes per element; the two-launch composition moves 16).  The arithmetic is the composition's, operation by
// operation ((e + o) * f, then ((p00 + p01) + p10) + p11 times fac), so the results are bit-identical to it.
template <int NP>
__global__ __launch_bounds__(256) void haar3d_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int D, int H, int W,
                                                         int obw, float fac, int64_t x_bs) {
    typedef float vin_t __attribute__((ext_vector_type(2 * NP)));
    typedef float vout_t __attribute__((ext_vector_type(NP)));
    const int h2 = H / 2, w2 = W / 2, wq = w2 / NP;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)h2 * wq) return;
    const int k = blockIdx.y, b = blockIdx.z, hy = (int)(i / wq), wx = (int)(i % wq) * NP;
    const int64_t HW = (int64_t)H * W;
    const float* pe = x + b * x_bs + (int64_t)(2 * k) * HW + (int64_t)(2 * hy) * W + 2 * wx;
    const vin_t e0 = *reinterpret_cast<const vin_t*>(pe), e1 = *reinterpret_cast<const vin_t*>(pe + W);
    const vin_t o0 = *reinterpret_cast<const vin_t*>(pe + HW), o1 = *reinterpret_cast<const vin_t*>(pe + HW + W);
    const vin_t band[2][2] = {{(e0 + o0) * CWFA_INV_SQRT2_F, (e1 + o1) * CWFA_INV_SQRT2_F},      // low band rows 0, 1
                              {(e0 - o0) * CWFA_INV_SQRT2_F, (e1 - o1) * CWFA_INV_SQRT2_F}};     // detail band
    const int64_t plane = (int64_t)h2 * w2, pos = (int64_t)hy * w2 + wx;
    float* py = y + (int64_t)b * 4 * D * plane + pos;
#pragma unroll
    for (int hb = 0; hb < 2; ++hb) {
        const int c = hb * (D / 2) + k;                      // channel of the band in the depth-Haar layout (lo | hi)
        const int c0 = obw ? c : 4 * c, cs = obw ? D : 1;
        vout_t a, v1, v2, d;
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            const float p00 = band[hb][0][2 * q], p01 = band[hb][0][2 * q + 1], p10 = band[hb][1][2 * q], p11 = band[hb][1][2 * q + 1];
            a[q] = (((p00 + p01) + p10) + p11) * fac;
            v1[q] = (((p00 - p01) + p10) - p11) * fac;
            v2[q] = (((p00 + p01) - p10) - p11) * fac;
            d[q] = (((p00 - p01) - p10) + p11) * fac;
        }
        *reinterpret_cast<vout_t*>(py + (int64_t)(c0)*plane) = a;
        *reinterpret_cast<vout_t*>(py + (int64_t)(c0 + cs) * plane) = v1;
        *reinterpret_cast<vout_t*>(py + (int64_t)(c0 + 2 * cs) * plane) = v2;
        *reinterpret_cast<vout_t*>(py + (int64_t)(c0 + 3 * cs) * plane) = d;
    }
}

template <int NP>
__global__ __launch_bounds__(256) void haar3d_inv_kernel(const float* __restrict__ y, float* __restrict__ x, int D, int H, int W,
                                                         int obw, float fac, int64_t x_bs) {
    typedef float vin_t __attribute__((ext_vector_type(2 * NP)));
    typedef float vout_t __attribute__((ext_vector_type(NP)));
    const int h2 = H / 2, w2 = W / 2, wq = w2 / NP;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)h2 * wq) return;
    const int k = blockIdx.y, b = blockIdx.z, hy = (int)(i / wq), wx = (int)(i % wq) * NP;
    const int64_t HW = (int64_t)H * W, plane = (int64_t)h2 * w2, pos = (int64_t)hy * w2 + wx;
    const float* py = y + (int64_t)b * 4 * D * plane + pos;
    vin_t band[2][2];
#pragma unroll
    for (int hb = 0; hb < 2; ++hb) {
        const int c = hb * (D / 2) + k;
        const int c0 = obw ? c : 4 * c, cs = obw ? D : 1;
        const vout_t a = *reinterpret_cast<const vout_t*>(py + (int64_t)(c0)*plane) * fac;
        const vout_t v1 = *reinterpret_cast<const vout_t*>(py + (int64_t)(c0 + cs) * plane) * fac;
        const vout_t v2 = *reinterpret_cast<const vout_t*>(py + (int64_t)(c0 + 2 * cs) * plane) * fac;
        const vout_t d = *reinterpret_cast<const vout_t*>(py + (int64_t)(c0 + 3 * cs) * plane) * fac;
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            band[hb][0][2 * q] = ((a[q] + v1[q]) + v2[q]) + d[q];
            band[hb][0][2 * q + 1] = ((a[q] - v1[q]) + v2[q]) - d[q];
            band[hb][1][2 * q] = ((a[q] + v1[q]) - v2[q]) - d[q];
            band[hb][1][2 * q + 1] = ((a[q] - v1[q]) - v2[q]) + d[q];
        }
    }
    float* pe = x + b * x_bs + (int64_t)(2 * k) * HW + (int64_t)(2 * hy) * W + 2 * wx;
    *reinterpret_cast<vin_t*>(pe) = (band[0][0] + band[1][0]) * CWFA_INV_SQRT2_F;
    *reinterpret_cast<vin_t*>(pe + W) = (band[0][1] + band[1][1]) * CWFA_INV_SQRT2_F;
    *reinterpret_cast<vin_t*>(pe + HW) = (band[0][0] - band[1][0]) * CWFA_INV_SQRT2_F;
    *reinterpret_cast<vin_t*>(pe + HW + W) = (band[0][1] - band[1][1]) * CWFA_INV_SQRT2_F;
}

static int haar3d_launch(bool fwd, const float* a, float* b, int B, int D, int H, int W, int obw, float fac, int64_t x_bs,
                         void* stream) {
    const char* name = fwd ? "cwfa_haar3d_fwd_f32" : "cwfa_haar3d_inv_f32";
    CWFA_REQUIRE(B >= 0 && D >= 0 && H >= 0 && W >= 0, CWFA_E_INVAL, "%s: negative size", name);
    if (B == 0 || D == 0 || H == 0 || W == 0) return CWFA_OK;
    CWFA_REQUIRE(a && b, CWFA_E_INVAL, "%s: null pointer", name);
    CWFA_REQUIRE(D % 2 == 0 && H % 2 == 0 && W % 2 == 0, CWFA_E_SHAPE, "%s: D=%d, H=%d, W=%d must be even", name, D, H, W);
    CWFA_REQUIRE(D / 2 <= 65535 && B <= 65535, CWFA_E_SHAPE, "%s: grid too large (D/2=%d, B=%d)", name, D / 2, B);
    const float* xin = fwd ? a : b;                         // the [B,D,H,W] side carries the batch stride
    const float* yside = fwd ? b : a;
    const bool np2 = W % 4 == 0 && x_bs % 4 == 0 && cwfa_aligned16(xin) && (reinterpret_cast<uintptr_t>(yside) & 7u) == 0;
    const int64_t n = (int64_t)(H / 2) * (W / 2 / (np2 ? 2 : 1));
    dim3 grid((unsigned)((n + 255) / 256), D / 2, B);
    hipStream_t st = (hipStream_t)stream;
    if (fwd) {
        if (np2) hipLaunchKernelGGL(haar3d_fwd_kernel<2>, grid, dim3(256), 0, st, a, b, D, H, W, obw, fac, x_bs);
        else hipLaunchKernelGGL(haar3d_fwd_kernel<1>, grid, dim3(256), 0, st, a, b, D, H, W, obw, fac, x_bs);
    } else {
        if (np2) hipLaunchKernelGGL(haar3d_inv_kernel<2>, grid, dim3(256), 0, st, a, b, D, H, W, obw, fac, x_bs);
        else hipLaunchKernelGGL(haar3d_inv_kernel<1>, grid, dim3(256), 0, st, a, b, D, H, W, obw, fac, x_bs);
    }
    CWFA_LAUNCH_CHECK(name);
    return CWFA_OK;
}

extern "C" int cwfa_haar3d_fwd_f32(const float* x, float* y, int B, int D, int H, int W, int order_by_wavelet, float fac,
                                   int64_t x_bs, void* stream) {
    return haar3d_launch(true, x, y, B, D, H, W, order_by_wavelet, fac, x_bs, stream);
}
extern "C" int cwfa_haar3d_inv_f32(const float* y, float* x, int B, int D, int H, int W, int order_by_wavelet, float fac,
                                   int64_t x_bs, void* stream) {
    return haar3d_launch(false, y, x, B, D, H, W, order_by_wavelet, fac, x_bs, stream);
}

// ------------------------------------------------------------------------------------------------ gathers
struct Pos {
    int c, h, w;
};

__device__ __forceinline__ Pos gather_pos(Pos p, const int64_t* __restrict__ perm, int axis) {
    if (perm) {
        if (axis == 1)
            p.c = (int)perm[p.c];
        else if (axis == 2)
            p.h = (int)perm[p.h];
        else
            p.w = (int)perm[p.w];
    }
    return p;
}

__global__ __launch_bounds__(256) void gather_kernel(const float* __restrict__ x, const int64_t* __restrict__ perm,
                                                     float* __restrict__ y, int C, int H, int W, int axis, int64_t x_bs,
                                                     int64_t y_bs) {
    const int64_t n = (int64_t)C * H * W;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int b = blockIdx.y;
    Pos p{(int)(i / ((int64_t)H * W)), (int)((i / W) % H), (int)(i % W)};
    const Pos q = gather_pos(p, perm, axis);
    y[b * y_bs + i] = x[b * x_bs + ((int64_t)q.c * H + q.h) * W + q.w];
}

static int check_perm_axis(const char* name, int axis) {
    CWFA_REQUIRE(axis >= 1 && axis <= 3, CWFA_E_INVAL, "%s: axis %d not in 1..3", name, axis);
    return CWFA_OK;
}

extern "C" int cwfa_gather_f32(const float* x, const int64_t* perm, float* y, int B, int C, int H, int W, int axis,
                               int64_t x_bs, int64_t y_bs, void* stream) {
    CWFA_REQUIRE(x && y && perm, CWFA_E_INVAL, "cwfa_gather_f32: null pointer");
    CWFA_REQUIRE(B >= 0 && C >= 0 && H >= 0 && W >= 0, CWFA_E_INVAL, "cwfa_gather_f32: negative size");
    CWFA_REQUIRE(B <= 65535, CWFA_E_SHAPE, "cwfa_gather_f32: batch too large");
    int rc = check_perm_axis("cwfa_gather_f32", axis);
    if (rc) return rc;
    const int64_t n = (int64_t)C * H * W;
    if (B == 0 || n == 0) return CWFA_OK;
    dim3 grid((unsigned)((n + 255) / 256), B);
    hipLaunchKernelGGL(gather_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, perm, y, C, H, W, axis, x_bs, y_bs);
    CWFA_LAUNCH_CHECK("cwfa_gather_f32");
    return CWFA_OK;
}

// ------------------------------------------------------------------------------------------------ affine apply
__device__ __forceinline__ float soft_clamp(float a, int kind, float clamp) {
    switch (kind) {
        case CWFA_CLAMP_ATAN: return clamp * (0.636f * cwfa_atan(a));
        case CWFA_CLAMP_TANH: return clamp * cwfa_tanh(a);
        case CWFA_CLAMP_SIGMOID: return clamp * (2.f * (1.f / (1.f + expf(-a)) - 0.5f));
        default: return clamp * a;
    }
}

// read s (clamped) and t of a stage at linear in-sample offset `off`
__device__ __forceinline__ void stage_st(const cwfa_affine_stage& st, int b, int64_t off, float& s, float& t) {
    s = 0.f;
    t = 0.f;
    if (st.s_raw) s = soft_clamp(st.s_raw[b * st.s_bs + off] * st.pre_scale, st.clamp_kind, st.clamp);
    if (st.t) {
        const float tv = st.t[b * st.t_bs + off];
        t = st.t_neg_div_sqrt2 ? (-tv) / CWFA_SQRT2_F : tv * st.pre_scale;
    }
}

__device__ __forceinline__ float affine_apply(float v, float s, float t, int rev) {
    return rev ? (v - t) * expf(-s) : expf(s) * v + t;
}

__global__ __launch_bounds__(256) void affine_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                     cwfa_affine_stage st, int rev, int C, int H, int W, int64_t x_bs,
                                                     int64_t y_bs, double* __restrict__ logdet,
                                                     double* __restrict__ sumsq) {
    __shared__ double red[16];
    const int64_t n = (int64_t)C * H * W;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    float s = 0.f, out = 0.f;
    if (i < n) {
        Pos p{(int)(i / ((int64_t)H * W)), (int)((i / W) % H), (int)(i % W)};
        const Pos q = gather_pos(p, st.perm, st.perm_axis);
        const float v = x ? x[b * x_bs + ((int64_t)q.c * H + q.h) * W + q.w] : 0.f;
        float t;
        stage_st(st, b, i, s, t);
        out = affine_apply(v, s, t, rev);
        y[b * y_bs + i] = out;
    }
    if (logdet) {
        const double tot = cwfa_block_sum((double)s, red);
        if (threadIdx.x == 0) atomicAdd(&logdet[b], rev ? -tot : tot);
    }
    if (sumsq) {
        const double tot = cwfa_block_sum((double)out * (double)out, red);
        if (threadIdx.x == 0) atomicAdd(sumsq, tot);
    }
}

// GIN variant: the channel mean of s is removed at every pixel (coupling_layers.py:355,372); one thread per pixel.
__global__ __launch_bounds__(256) void affine_gin_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                         cwfa_affine_stage st, int rev, int C, int H, int W,
                                                         int64_t x_bs, int64_t y_bs, double* __restrict__ sumsq) {
    __shared__ double red[16];
    const int64_t HW = (int64_t)H * W;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    double sq = 0.0;
    if (i < HW) {
        float mean = 0.f;
        for (int c = 0; c < C; ++c) {
            float s, t;
            stage_st(st, b, c * HW + i, s, t);
            mean += s;
        }
        mean /= (float)C;
        const int h = (int)(i / W), w = (int)(i % W);
        for (int c = 0; c < C; ++c) {
            float s, t;
            stage_st(st, b, c * HW + i, s, t);
            s -= mean;
            const Pos q = gather_pos(Pos{c, h, w}, st.perm, st.perm_axis);
            const float v = x ? x[b * x_bs + ((int64_t)q.c * H + q.h) * W + q.w] : 0.f;
            const float out = affine_apply(v, s, t, rev);
            y[b * y_bs + c * HW + i] = out;
            sq += (double)out * (double)out;
        }
    }
    if (sumsq) {
        const double tot = cwfa_block_sum(sq, red);
        if (threadIdx.x == 0) atomicAdd(sumsq, tot);
    }
}

static int check_stage(const char* name, const cwfa_affine_stage& st) {
    CWFA_REQUIRE(st.clamp_kind >= CWFA_CLAMP_NONE && st.clamp_kind <= CWFA_CLAMP_SIGMOID, CWFA_E_INVAL,
                 "%s: bad clamp kind %d", name, st.clamp_kind);
    if (st.perm) {
        int rc = check_perm_axis(name, st.perm_axis);
        if (rc) return rc;
    }
    return CWFA_OK;
}

extern "C" int cwfa_affine_f32(const float* x, float* y, const cwfa_affine_stage* st, int rev, int B, int C, int H, int W,
                               int64_t x_bs, int64_t y_bs, double* logdet, double* sumsq, void* stream) {
    CWFA_REQUIRE(y && st, CWFA_E_INVAL, "cwfa_affine_f32: null pointer");
    CWFA_REQUIRE(B >= 0 && C >= 0 && H >= 0 && W >= 0, CWFA_E_INVAL, "cwfa_affine_f32: negative size");
    CWFA_REQUIRE(B <= 65535, CWFA_E_SHAPE, "cwfa_affine_f32: batch too large");
    int rc = check_stage("cwfa_affine_f32", *st);
    if (rc) return rc;
    const int64_t n = (int64_t)C * H * W;
    if (B == 0 || n == 0) return CWFA_OK;
    if (st->gin) {
        CWFA_REQUIRE(st->s_raw, CWFA_E_INVAL, "cwfa_affine_f32: GIN needs s_raw");
        dim3 grid((unsigned)(((int64_t)H * W + 255) / 256), B);
        hipLaunchKernelGGL(affine_gin_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, y, *st, rev, C, H, W, x_bs, y_bs,
                           sumsq);
    } else {
        dim3 grid((unsigned)((n + 255) / 256), B);
        hipLaunchKernelGGL(affine_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, y, *st, rev, C, H, W, x_bs, y_bs,
                           logdet, sumsq);
    }
    CWFA_LAUNCH_CHECK("cwfa_affine_f32");
    return CWFA_OK;
}

// ------------------------------------------------------------------------------------------------ per-channel affine
__global__ __launch_bounds__(256) void channel_affine_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                             const float* __restrict__ scale,
                                                             const float* __restrict__ shift, int mode,
                                                             const int64_t* __restrict__ perm_in,
                                                             const int64_t* __restrict__ perm_out, int C, int64_t HW,
                                                             int64_t x_bs, int64_t y_bs) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= HW) return;
    const int c = blockIdx.y, b = blockIdx.z;
    // channel whose affine parameters apply, and channel read from the input
    const int cp = perm_out ? (int)perm_out[c] : c;         // parameters follow the pre-permutation channel
    const int cr = perm_in ? (int)perm_in[c] : cp;
    const float v = x[b * x_bs + (int64_t)cr * HW + p];
    const float sc = scale ? scale[cp] : 1.f, sh = shift ? shift[cp] : 0.f;
    y[b * y_bs + (int64_t)c * HW + p] = mode == 0 ? v * sc + sh : (v - sh) / sc;
}

extern "C" int cwfa_channel_affine_f32(const float* x, float* y, const float* scale, const float* shift, int mode,
                                       const int64_t* perm_in, const int64_t* perm_out, int B, int C, int64_t HW,
                                       int64_t x_bs, int64_t y_bs, void* stream) {
    CWFA_REQUIRE(x && y, CWFA_E_INVAL, "cwfa_channel_affine_f32: null pointer");
    CWFA_REQUIRE(!(perm_in && perm_out), CWFA_E_INVAL, "cwfa_channel_affine_f32: both perm_in and perm_out given");
    CWFA_REQUIRE(mode == 0 || mode == 1, CWFA_E_INVAL, "cwfa_channel_affine_f32: bad mode");
    CWFA_REQUIRE(B >= 0 && C >= 0 && HW >= 0 && C <= 65535 && B <= 65535, CWFA_E_SHAPE,
                 "cwfa_channel_affine_f32: bad shape");
    if (B == 0 || C == 0 || HW == 0) return CWFA_OK;
    dim3 grid((unsigned)((HW + 255) / 256), C, B);
    hipLaunchKernelGGL(channel_affine_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, y, scale, shift, mode, perm_in,
                       perm_out, C, HW, x_bs, y_bs);
    CWFA_LAUNCH_CHECK("cwfa_channel_affine_f32");
    return CWFA_OK;
}

// ------------------------------------------------------------------------------------------------ fused chains
// Every stage is a pull:  v_{k+1}[p] = A_k(v_k[g_k(p)], s_k[p], t_k[p]).  A thread owns ONE final position, walks the
// gathers backwards to find where its value starts, then applies the affines forwards.  Each (stage, position) pair is
// visited by exactly one thread, so block-reducing the s values gives the exact per-sample log-det.
__device__ __forceinline__ int64_t lin(Pos p, int H, int W) { return ((int64_t)p.c * H + p.h) * W + p.w; }

__global__ __launch_bounds__(256) void chain_inv_kernel(const float* __restrict__ z, const float* __restrict__ low,
                                                        float* __restrict__ x, cwfa_chain ch, int C, int H, int W,
                                                        int64_t z_bs, int64_t low_bs, int64_t x_bs,
                                                        double* __restrict__ logdet) {
    __shared__ double red[16];
    const int64_t HW = (int64_t)H * W, n = (int64_t)C * HW;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    float ssum = 0.f;
    if (i < n) {
        Pos p{(int)(i / HW), (int)((i / W) % H), (int)(i % W)};
        int64_t off[CWFA_CHAIN_MAX];
#pragma unroll
        for (int k = CWFA_CHAIN_MAX - 1; k >= 0; --k) {
            if (k < ch.n_stages) {
                off[k] = lin(p, H, W);
                p = gather_pos(p, ch.stage[k].perm, ch.stage[k].perm_axis);
            }
        }
        float v = z ? z[b * z_bs + lin(p, H, W)] : 0.f;
#pragma unroll
        for (int k = 0; k < CWFA_CHAIN_MAX; ++k) {
            if (k < ch.n_stages) {
                float s, t;
                stage_st(ch.stage[k], b, off[k], s, t);
                v = affine_apply(v, s, t, 1);
                ssum += s;
            }
        }
        const int c = (int)(i / HW);
        const int64_t pix = i - (int64_t)c * HW;
        const float l = low[b * low_bs + i];
        x[b * x_bs + (int64_t)(2 * c) * HW + pix] = (l + v) * CWFA_INV_SQRT2_F;
        x[b * x_bs + (int64_t)(2 * c + 1) * HW + pix] = (l - v) * CWFA_INV_SQRT2_F;
    }
    if (logdet) {
        const double tot = cwfa_block_sum((double)ssum, red);
        if (threadIdx.x == 0) atomicAdd(&logdet[b], -tot);
    }
}

__global__ __launch_bounds__(256) void chain_fwd_kernel(const float* __restrict__ x, float* __restrict__ low,
                                                        float* __restrict__ zout, cwfa_chain ch,
                                                        const int64_t* __restrict__ final_perm, int C, int H, int W,
                                                        int64_t x_bs, int64_t low_bs, int64_t z_bs,
                                                        double* __restrict__ logdet, double* __restrict__ sumsq) {
    __shared__ double red[16];
    const int64_t HW = (int64_t)H * W, n = (int64_t)C * HW;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    float ssum = 0.f, v = 0.f;
    if (i < n) {
        Pos p{(int)(i / HW), (int)((i / W) % H), (int)(i % W)};
        {   // the low-pass half at this thread's own position (Split.out0, networks.py:364)
            const int64_t pix = i - (int64_t)p.c * HW;
            const float e = x[b * x_bs + (int64_t)(2 * p.c) * HW + pix], o = x[b * x_bs + (int64_t)(2 * p.c + 1) * HW + pix];
            low[b * low_bs + i] = (e + o) * CWFA_INV_SQRT2_F;
        }
        p = gather_pos(p, final_perm, 1);
        int64_t off[CWFA_CHAIN_MAX];
#pragma unroll
        for (int k = CWFA_CHAIN_MAX - 1; k >= 0; --k) {
            if (k < ch.n_stages) {
                off[k] = lin(p, H, W);
                p = gather_pos(p, ch.stage[k].perm, ch.stage[k].perm_axis);
            }
        }
        {   // the detail half where this thread's value starts
            const int64_t pix = (int64_t)p.h * W + p.w;
            const float e = x[b * x_bs + (int64_t)(2 * p.c) * HW + pix], o = x[b * x_bs + (int64_t)(2 * p.c + 1) * HW + pix];
            v = (e - o) * CWFA_INV_SQRT2_F;
        }
#pragma unroll
        for (int k = 0; k < CWFA_CHAIN_MAX; ++k) {
            if (k < ch.n_stages) {
                float s, t;
                stage_st(ch.stage[k], b, off[k], s, t);
                v = affine_apply(v, s, t, 0);
                ssum += s;
            }
        }
        zout[b * z_bs + i] = v;
    }
    if (logdet) {
        const double tot = cwfa_block_sum((double)ssum, red);
        if (threadIdx.x == 0) atomicAdd(&logdet[b], tot);
    }
    if (sumsq) {
        const double tot = cwfa_block_sum((double)v * (double)v, red);
        if (threadIdx.x == 0) atomicAdd(sumsq, tot);
    }
}

// ---- row-staged variants (the ones that normally run).  A block owns ONE output row (b, c, h, all w).  Channel and row
// permutations map whole rows to whole rows, so at every stage the block needs exactly one row of s_k and t_k: it is
// loaded COALESCED into LDS (as exp(-+s) and t), and only then read at the column the value actually travels through.
// In the per-element kernels above a column permutation turns every later s/t access into a 64-way gather (measured
// 1.1 TB/s effective); here all global traffic is full-row streaming.
struct RowPos {
    int c, h;
};
__device__ __forceinline__ RowPos row_gather(RowPos p, const int64_t* __restrict__ perm, int axis) {
    if (perm) {
        if (axis == 1) p.c = (int)perm[p.c];
        if (axis == 2) p.h = (int)perm[p.h];
    }
    return p;
}

__global__ __launch_bounds__(256) void chain_inv_rows_kernel(const float* __restrict__ z, const float* __restrict__ low,
                                                             float* __restrict__ x, cwfa_chain ch, int C, int H, int W,
                                                             int64_t z_bs, int64_t low_bs, int64_t x_bs,
                                                             double* __restrict__ logdet) {
    extern __shared__ float rows[];          // [stage][2][W]: e = exp(-s), t      (+ [W] z row)
    __shared__ double red[16];
    const int b = blockIdx.z, c = blockIdx.y, h = blockIdx.x;
    const int64_t HW = (int64_t)H * W;
    // walk the row coordinates backwards through the stages (uniform over the block)
    RowPos q[CWFA_CHAIN_MAX], src = RowPos{c, h};          // q[k]: row where stage k reads s,t; src: row of z
#pragma unroll
    for (int k = CWFA_CHAIN_MAX - 1; k >= 0; --k)
        if (k < ch.n_stages) {
            q[k] = src;
            src = row_gather(src, ch.stage[k].perm, ch.stage[k].perm_axis);
        }
    float ssum = 0.f;
#pragma unroll
    for (int k = 0; k < CWFA_CHAIN_MAX; ++k)
        if (k < ch.n_stages) {
            const int64_t off = ((int64_t)q[k].c * H + q[k].h) * W;
            for (int w = threadIdx.x; w < W; w += blockDim.x) {
                float s, t;
                stage_st(ch.stage[k], b, off + w, s, t);
                ssum += s;
                rows[(2 * k) * W + w] = expf(-s);
                rows[(2 * k + 1) * W + w] = t;
            }
        }
    float* zrow = rows + 2 * ch.n_stages * W;
    if (z) {
        const int64_t off = ((int64_t)src.c * H + src.h) * W;
        for (int w = threadIdx.x; w < W; w += blockDim.x) zrow[w] = z[b * z_bs + off + w];
    }
    __syncthreads();
    for (int w = threadIdx.x; w < W; w += blockDim.x) {
        int wq[CWFA_CHAIN_MAX], w0 = w;                    // wq[k]: column where stage k reads s,t; w0: column of z
#pragma unroll
        for (int k = CWFA_CHAIN_MAX - 1; k >= 0; --k)
            if (k < ch.n_stages) {
                wq[k] = w0;
                if (ch.stage[k].perm && ch.stage[k].perm_axis == 3) w0 = (int)ch.stage[k].perm[w0];
            }
        float v = z ? zrow[w0] : 0.f;
#pragma unroll
        for (int k = 0; k < CWFA_CHAIN_MAX; ++k)
            if (k < ch.n_stages) v = (v - rows[(2 * k + 1) * W + wq[k]]) * rows[(2 * k) * W + wq[k]];
        const int64_t o = (int64_t)h * W + w;
        const float l = low[b * low_bs + (int64_t)c * HW + o];
        x[b * x_bs + (int64_t)(2 * c) * HW + o] = (l + v) * CWFA_INV_SQRT2_F;
        x[b * x_bs + (int64_t)(2 * c + 1) * HW + o] = (l - v) * CWFA_INV_SQRT2_F;
    }
    if (logdet) {
        const double tot = cwfa_block_sum((double)ssum, red);
        if (threadIdx.x == 0) atomicAdd(&logdet[b], -tot);
    }
}

__global__ __launch_bounds__(256) void chain_fwd_rows_kernel(const float* __restrict__ x, float* __restrict__ low,
                                                             float* __restrict__ zout, cwfa_chain ch,
                                                             const int64_t* __restrict__ final_perm, int C, int H, int W,
                                                             int64_t x_bs, int64_t low_bs, int64_t z_bs,
                                                             double* __restrict__ logdet, double* __restrict__ sumsq) {
    extern __shared__ float rows[];          // [stage][2][W]: e = exp(s), t      + [W] detail row at the source
    __shared__ double red[16];
    const int b = blockIdx.z, c = blockIdx.y, h = blockIdx.x;
    const int64_t HW = (int64_t)H * W;
    RowPos q[CWFA_CHAIN_MAX], src = row_gather(RowPos{c, h}, final_perm, 1);
#pragma unroll
    for (int k = CWFA_CHAIN_MAX - 1; k >= 0; --k)
        if (k < ch.n_stages) {
            q[k] = src;
            src = row_gather(src, ch.stage[k].perm, ch.stage[k].perm_axis);
        }
    float ssum = 0.f;
#pragma unroll
    for (int k = 0; k < CWFA_CHAIN_MAX; ++k)
        if (k < ch.n_stages) {
            const int64_t off = ((int64_t)q[k].c * H + q[k].h) * W;
            for (int w = threadIdx.x; w < W; w += blockDim.x) {
                float s, t;
                stage_st(ch.stage[k], b, off + w, s, t);
                ssum += s;
                rows[(2 * k) * W + w] = expf(s);
                rows[(2 * k + 1) * W + w] = t;
            }
        }
    float* drow = rows + 2 * ch.n_stages * W;
    {   // detail (hi) row where this block's values start, and the low-pass row at the block's own position
        const int64_t so = (int64_t)src.h * W, oo = (int64_t)h * W;
        const float* xe = x + b * x_bs + (int64_t)(2 * src.c) * HW + so;
        const float* xs = x + b * x_bs + (int64_t)(2 * c) * HW + oo;
        for (int w = threadIdx.x; w < W; w += blockDim.x) {
            drow[w] = (xe[w] - xe[HW + w]) * CWFA_INV_SQRT2_F;
            low[b * low_bs + (int64_t)c * HW + oo + w] = (xs[w] + xs[HW + w]) * CWFA_INV_SQRT2_F;
        }
    }
    __syncthreads();
    double sq = 0.0;
    for (int w = threadIdx.x; w < W; w += blockDim.x) {
        int wq[CWFA_CHAIN_MAX], w0 = w;
#pragma unroll
        for (int k = CWFA_CHAIN_MAX - 1; k >= 0; --k)
            if (k < ch.n_stages) {
                wq[k] = w0;
                if (ch.stage[k].perm && ch.stage[k].perm_axis == 3) w0 = (int)ch.stage[k].perm[w0];
            }
        float v = drow[w0];
#pragma unroll
        for (int k = 0; k < CWFA_CHAIN_MAX; ++k)
            if (k < ch.n_stages) v = rows[(2 * k) * W + wq[k]] * v + rows[(2 * k + 1) * W + wq[k]];
        zout[b * z_bs + (int64_t)c * HW + (int64_t)h * W + w] = v;
        sq += (double)v * (double)v;
    }
    if (logdet) {
        const double tot = cwfa_block_sum((double)ssum, red);
        if (threadIdx.x == 0) atomicAdd(&logdet[b], tot);
    }
    if (sumsq) {
        const double tot = cwfa_block_sum(sq, red);
        if (threadIdx.x == 0) atomicAdd(sumsq, tot);
    }
}

// ---- the same with 16-byte accesses and every global load of a block in flight at once (the form that normally runs at
// CWFA's sizes: W a multiple of 4 with W/4 dividing 256, 16-byte aligned rows).  A thread owns FOUR consecutive columns of
// one row; a block of 256 threads owns 1024/W rows of one channel.  All row loads of all stages (s, t: 2n float4 per
// thread, plus z / low or the x pair) are issued BEFORE the first use, so a block has its whole 13 - 14 row working set in
// flight instead of one stage at a time (the row-at-a-time form above ran at 0.27 of the HBM peak: latency-bound).  The
// coefficients are always read at the thread's own columns: a column permutation moves the travelling VALUES between the
// threads of a row through an 8 KB LDS exchange (one barrier each), so occupancy is not limited by LDS and a chain without
// column permutations uses none.
typedef float f4 __attribute__((ext_vector_type(4)));

// The coefficient rows, the low band and z are read ONCE by this kernel: non-temporal loads (no L2 allocation) move the inverse
// chain from 4.75 to 5.0 TB/s and the forward chain from 4.6 to 5.0 - 5.3 TB/s at 48 channels x 512 x 512 (tools/chain_time.py);
// non-temporal STORES add another 4 % to the inverse chain (5.25 TB/s) and cost the forward chain 2 %: CH_NT = 2 stores
// non-temporally in the inverse direction only.
#ifndef CH_NT
#define CH_NT 2          // 0: plain accesses, 1: non-temporal loads, 2: + non-temporal stores of the inverse chain's output
#endif
#ifndef CH_SCALAR
#define CH_SCALAR 0      // (tuning) 1: the block's row index through readfirstlane where a wave holds one row (scalar table loads)
#endif
__device__ __forceinline__ f4 ld_stream(const float* p) {
#if CH_NT
    return __builtin_nontemporal_load(reinterpret_cast<const f4*>(p));
#else
    return *reinterpret_cast<const f4*>(p);
#endif
}
template <bool NT>
__device__ __forceinline__ void st_stream(float* p, f4 v) {
    if constexpr (NT && CH_NT > 1) __builtin_nontemporal_store(v, reinterpret_cast<f4*>(p));
    else *reinterpret_cast<f4*>(p) = v;
}

// s (clamped) and t of four columns from the raw float4 rows of a stage
__device__ __forceinline__ void stage_st4(const cwfa_affine_stage& st, const f4& sr, const f4& tr, f4& s, f4& t) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        s[j] = st.s_raw ? soft_clamp(sr[j] * st.pre_scale, st.clamp_kind, st.clamp) : 0.f;
        t[j] = st.t ? (st.t_neg_div_sqrt2 ? (-tr[j]) / CWFA_SQRT2_F : tr[j] * st.pre_scale) : 0.f;
    }
}

#ifndef CH_BLOCK
#define CH_BLOCK 256     // (tuning) threads per block of chain_rows4_kernel: a block covers CH_BLOCK * 4 pixels = whole image rows
#endif
#ifndef CH_WAVES
#define CH_WAVES 0       // (tuning) minimum waves per SIMD asked of the register allocator (0: none)
#endif
// NS: stages the register arrays are sized for (6 covers a CAT step: five conditional affines + the trailing permutation; sized for
// CWFA_CHAIN_MAX = 8 the kernel held 87 registers = five waves per SIMD -- with <= 80 it holds six, i.e. 1536 resident blocks, which the
// 1536 x 2^k blocks of the 6 / 12 / 24 / 48-channel levels at 512 x 512 fill in whole rounds)
template <bool INV, int NS = CWFA_CHAIN_MAX>
#if CH_WAVES
__global__ __launch_bounds__(CH_BLOCK, CH_WAVES) void chain_rows4_kernel(
#else
__global__ __launch_bounds__(CH_BLOCK) void chain_rows4_kernel(
#endif
const float* __restrict__ a0, float* __restrict__ a1, float* __restrict__ a2,
                                                          cwfa_chain ch, const int64_t* __restrict__ final_perm, int C, int H, int W,
                                                          int64_t bs0, int64_t bs1, int64_t bs2, double* __restrict__ logdet,
                                                          double* __restrict__ sumsq) {
    // INV:  a0 = low [C], a1 = x out [2C], a2 = z in (may be null = zeros, const in effect)      (bs0, bs1, bs2 alike)
    // !INV: a0 = x in [2C], a1 = low out [C], a2 = z out [C]
    extern __shared__ float rows[];          // [2][row of the block][W]: exchange buffers of the travelling values (column gathers)
    __shared__ double red[16];
    const int tpr = W >> 2, RB = CH_BLOCK / tpr;
#if CH_SCALAR
    // a wave holds ONE image row when a row takes a multiple of 64 threads: the row index (and every table entry read with it)
    // is then wave-uniform and goes through the scalar unit
    const int r = (tpr & 63) == 0 ? __builtin_amdgcn_readfirstlane((int)threadIdx.x / tpr) : (int)threadIdx.x / tpr;
    const int w4 = ((int)threadIdx.x - r * tpr) * 4;
#else
    const int r = threadIdx.x / tpr, w4 = (threadIdx.x - r * tpr) * 4;
#endif
    const int b = blockIdx.z, c = blockIdx.y, hh = blockIdx.x * RB + r;
    const bool live = hh < H;
    const int h = live ? hh : H - 1;
    const int64_t HW = (int64_t)H * W;
    const int n = ch.n_stages;
    RowPos q[NS], src = INV ? RowPos{c, h} : row_gather(RowPos{c, h}, final_perm, 1);
    if (ch.src_c) {                          // composed by the caller: independent loads instead of a dependent walk
#pragma unroll
        for (int k = 0; k < NS; ++k)
            if (k < n) q[k] = RowPos{ch.src_c[k * C + c], ch.src_h[k * H + h]};
        src = RowPos{ch.src_c[n * C + c], ch.src_h[n * H + h]};
    } else {
#pragma unroll
        for (int k = NS - 1; k >= 0; --k)
            if (k < n) {
                q[k] = src;
                src = row_gather(src, ch.stage[k].perm, ch.stage[k].perm_axis);
            }
    }
    // ---- every global load of this thread, back to back
    f4 sr[NS], tr[NS];
    const f4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        sr[k] = tr[k] = zero;
        if (k < n) {
            const int64_t off = ((int64_t)q[k].c * H + q[k].h) * W + w4;
            if (ch.stage[k].s_raw) sr[k] = ld_stream(ch.stage[k].s_raw + b * ch.stage[k].s_bs + off);
            if (ch.stage[k].t) tr[k] = ld_stream(ch.stage[k].t + b * ch.stage[k].t_bs + off);
        }
    }
    f4 v0 = zero, lo = zero, own0 = zero, own1 = zero;
    const int64_t so = ((int64_t)src.h) * W + w4, oo = (int64_t)h * W + w4;
    if constexpr (INV) {
        if (a2) v0 = ld_stream(a2 + b * bs2 + (int64_t)src.c * HW + so);
        lo = ld_stream(a0 + b * bs0 + (int64_t)c * HW + oo);
    } else {
        const f4 e0 = *reinterpret_cast<const f4*>(a0 + b * bs0 + (int64_t)(2 * src.c) * HW + so);      // (x is read twice: cached)
        const f4 e1 = *reinterpret_cast<const f4*>(a0 + b * bs0 + (int64_t)(2 * src.c + 1) * HW + so);
        own0 = *reinterpret_cast<const f4*>(a0 + b * bs0 + (int64_t)(2 * c) * HW + oo);
        own1 = *reinterpret_cast<const f4*>(a0 + b * bs0 + (int64_t)(2 * c + 1) * HW + oo);
        v0 = (e0 - e1) * CWFA_INV_SQRT2_F;
    }
    // ---- coefficients at the thread's OWN four columns, then the chain.  A column permutation moves the travelling values
    // between the threads of a row (through one LDS row, double-buffered: one barrier per column permutation); channel and
    // row permutations only changed which rows were loaded above.
    f4 ev[NS], tv[NS];
    float ssum = 0.f;
#pragma unroll
    for (int k = 0; k < NS; ++k)
        if (k < n) {
            f4 sv;
            stage_st4(ch.stage[k], sr[k], tr[k], sv, tv[k]);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                ssum += sv[j];
                ev[k][j] = __expf(INV ? -sv[j] : sv[j]);          // v_exp_f32(s * log2 e): |s| <= clamp, relative error ~2e-7
            }
        }
    f4 v = v0;
    int nx = 0;
#pragma unroll
    for (int k = 0; k < NS; ++k)
        if (k < n) {
            if (ch.stage[k].perm && ch.stage[k].perm_axis == 3) {           // uniform over the block
                float* buf = rows + (size_t)((nx & 1) * RB + r) * W;
                ++nx;
                *reinterpret_cast<f4*>(buf + w4) = v;
                __syncthreads();
                const int64_t* pk = ch.stage[k].perm + w4;
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = buf[(int)pk[j]];
            }
            v = INV ? (v - tv[k]) * ev[k] : ev[k] * v + tv[k];
        }
    double sq = 0.0;
    if (live) {
        if constexpr (INV) {
            st_stream<true>(a1 + b * bs1 + (int64_t)(2 * c) * HW + oo, (lo + v) * CWFA_INV_SQRT2_F);
            st_stream<true>(a1 + b * bs1 + (int64_t)(2 * c + 1) * HW + oo, (lo - v) * CWFA_INV_SQRT2_F);
        } else {
            *reinterpret_cast<f4*>(a1 + b * bs1 + (int64_t)c * HW + oo) = (own0 + own1) * CWFA_INV_SQRT2_F;
            *reinterpret_cast<f4*>(a2 + b * bs2 + (int64_t)c * HW + oo) = v;
#pragma unroll
            for (int j = 0; j < 4; ++j) sq += (double)v[j] * (double)v[j];
        }
    } else {
        ssum = 0.f;
    }
    if (logdet) {
        const double tot = cwfa_block_sum((double)ssum, red);
        if (threadIdx.x == 0) atomicAdd(&logdet[b], INV ? -tot : tot);
    }
    if (!INV && sumsq) {
        const double tot = cwfa_block_sum(sq, red);
        if (threadIdx.x == 0) atomicAdd(sumsq, tot);
    }
}

// rows variant usable?  (LDS budget, grid limits)
// Backward of  L = gscale * 0.5 * sum z^2  -  ldscale * sum_b logdet_b  (CWFA.py:970-978: gscale = 1/numel,
// ldscale = 1/(B*numel)) through a whole forward chain in ONE launch and without stored activations: the flow is
// invertible, so the thread that owns a latent position walks the stages backwards, recovering each stage's input from
// its output, u = (v - t) * exp(-s), while it carries the gradient g:
//     dL/dt = g,   dL/ds = g * (v - t) - ldscale,   g <- g * exp(s),   position <- gather_k(position).
// The gathers are bijections, so every (stage, position) pair is visited by exactly one thread: plain stores.
__device__ __forceinline__ float soft_clamp_grad(float a, int kind, float clamp) {      // d soft_clamp / d a
    switch (kind) {
        case CWFA_CLAMP_ATAN: return clamp * 0.636f / (1.f + a * a);
        case CWFA_CLAMP_TANH: {
            const float th = tanhf(a);
            return clamp * (1.f - th * th);
        }
        case CWFA_CLAMP_SIGMOID: {
            const float sg = 1.f / (1.f + expf(-a));
            return clamp * 2.f * sg * (1.f - sg);
        }
        default: return clamp;
    }
}

// dL/ds, dL/dt of a stage -> gradients of the tensors the sub-network produced (through the soft clamp / the scalings)
__device__ __forceinline__ void store_stage_grads(const cwfa_affine_stage& st, const cwfa_chain_grads& gr, int k, int b, int64_t off,
                                                  float ds, float dt, int accumulate) {
    if (gr.ds[k]) {
        const float a = st.s_raw[b * st.s_bs + off] * st.pre_scale;
        const float v = ds * soft_clamp_grad(a, st.clamp_kind, st.clamp) * st.pre_scale;
        float* dst = gr.ds[k] + b * gr.ds_bs[k] + off;
        *dst = accumulate ? *dst + v : v;
    }
    if (gr.dt[k]) {
        const float v = st.t_neg_div_sqrt2 ? (-dt) / CWFA_SQRT2_F : dt * st.pre_scale;
        float* dst = gr.dt[k] + b * gr.dt_bs[k] + off;
        *dst = accumulate ? *dst + v : v;
    }
}

__global__ __launch_bounds__(256) void chain_bwd_kernel(const float* __restrict__ z, const float* __restrict__ gz, cwfa_chain ch,
                                                        cwfa_chain_grads gr, const int64_t* __restrict__ final_perm,
                                                        float* __restrict__ gv0, int C, int H, int W, int64_t z_bs,
                                                        int64_t gz_bs, int64_t gv0_bs, float gscale, float ldscale,
                                                        int accumulate, const float* __restrict__ gld) {
    const int64_t HW = (int64_t)H * W, n = (int64_t)C * HW;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int b = blockIdx.y;
    if (gld) ldscale -= gld[b];              // an upstream gradient dL/d(logdet_b) (autograd): dL/ds += gld[b]
    Pos p{(int)(i / HW), (int)((i / W) % H), (int)(i % W)};
    float v = z[b * z_bs + i];
    float g = (gz ? gz[b * gz_bs + i] : 0.f) + gscale * v;
    p = gather_pos(p, final_perm, 1);
#pragma unroll
    for (int k = CWFA_CHAIN_MAX - 1; k >= 0; --k) {
        if (k < ch.n_stages) {
            const cwfa_affine_stage& st = ch.stage[k];
            const int64_t off = lin(p, H, W);
            float s, t;
            stage_st(st, b, off, s, t);
            const float e = v - t;
            store_stage_grads(st, gr, k, b, off, g * e - ldscale, g, accumulate);
            v = e * expf(-s);
            g = g * expf(s);
            p = gather_pos(p, st.perm, st.perm_axis);
        }
    }
    if (gv0) gv0[b * gv0_bs + lin(p, H, W)] = g;
}

// Backward of a reconstruction loss on the INVERSE pass (CWFA.py:952-959: F.l1_loss / F.mse_loss(curr_gt, upsampled_vol),
// upsampled_vol = graph([z, low], c, rev=True) CWFA.py:911) through the inverse chain, again without stored activations.
// xhat = Haar1D^-1(cat[low, v0]) and the inverse stages are v_k = G_k^-1((v_{k+1} - t_k) e^{-s_k}); with w_k = G_k(v_k):
//     dL/dt_k = -g e^{-s},   dL/ds_k = -g w_k,   g <- g e^{-s},   w_k -> v_{k+1} = e^{s} w_k + t
// where g starts as the detail band of Haar1D(dL/dxhat) (the transform is orthonormal) and both v and g travel along the
// forward chain's pull walk -- the thread owning a position of v_n walks the gathers back to v_0, loads
// v_0 = hi(xhat), g_0 = gscale * hi(loss'(xhat - gt)) there and applies the stages forwards.  `ch` is the FORWARD-order
// chain.  loss_sum (nullable) += sum |xhat - gt|^p over the elements (p = loss_kind: 1 or 2).
__global__ __launch_bounds__(256) void chain_inv_bwd_kernel(const float* __restrict__ xhat, const float* __restrict__ gt,
                                                            cwfa_chain ch, cwfa_chain_grads gr, int C, int H, int W,
                                                            int64_t xhat_bs, int64_t gt_bs, float gscale, int loss_kind,
                                                            int accumulate, double* __restrict__ loss_sum, float* __restrict__ gz_out,
                                                            float* __restrict__ glow_out) {
    __shared__ double red[16];
    const int64_t HW = (int64_t)H * W, n = (int64_t)C * HW;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    double lsum = 0.0;
    if (i < n) {
        Pos p{(int)(i / HW), (int)((i / W) % H), (int)(i % W)};
        int64_t off[CWFA_CHAIN_MAX];
#pragma unroll
        for (int k = CWFA_CHAIN_MAX - 1; k >= 0; --k) {
            if (k < ch.n_stages) {
                off[k] = lin(p, H, W);
                p = gather_pos(p, ch.stage[k].perm, ch.stage[k].perm_axis);
            }
        }
        const int64_t pix = (int64_t)p.h * W + p.w;
        const float x0 = xhat[b * xhat_bs + (int64_t)(2 * p.c) * HW + pix], x1 = xhat[b * xhat_bs + (int64_t)(2 * p.c + 1) * HW + pix];
        const float q0 = gt[b * gt_bs + (int64_t)(2 * p.c) * HW + pix], q1 = gt[b * gt_bs + (int64_t)(2 * p.c + 1) * HW + pix];
        const float d0 = x0 - q0, d1 = x1 - q1;
        float v = (x0 - x1) * CWFA_INV_SQRT2_F, g;
        if (loss_kind == 0) {                 // `gt` IS the upstream gradient dL/dxhat (autograd): taken as loaded (NOT as x - (x - gt):
            const float u0 = q0, u1 = q1;    // that rounds a gradient of 1e-7 to a multiple of ulp(x) ~ 6e-8 -- round 3's first version)
            g = gscale * ((u0 - u1) * CWFA_INV_SQRT2_F);
            if (glow_out) glow_out[b * n + (int64_t)p.c * HW + pix] = gscale * ((u0 + u1) * CWFA_INV_SQRT2_F);
        } else if (loss_kind == 2) {
            g = gscale * ((d0 - d1) * CWFA_INV_SQRT2_F);
            lsum = (double)d0 * d0 + (double)d1 * d1;
        } else {
            const float s0 = d0 > 0.f ? 1.f : (d0 < 0.f ? -1.f : 0.f), s1 = d1 > 0.f ? 1.f : (d1 < 0.f ? -1.f : 0.f);
            g = gscale * ((s0 - s1) * CWFA_INV_SQRT2_F);
            lsum = (double)fabsf(d0) + (double)fabsf(d1);
        }
#pragma unroll
        for (int k = 0; k < CWFA_CHAIN_MAX; ++k) {
            if (k < ch.n_stages) {
                const cwfa_affine_stage& st = ch.stage[k];
                float s, t;
                stage_st(st, b, off[k], s, t);
                const float ge = g * expf(-s);
                store_stage_grads(st, gr, k, b, off[k], -g * v, -ge, accumulate);
                g = ge;
                v = expf(s) * v + t;
            }
        }
        if (gz_out) gz_out[b * n + i] = g;        // dL/d(latent input of the inverse chain), at this thread's own position
    }
    if (loss_sum) {
        const double tot = cwfa_block_sum(lsum, red);
        if (threadIdx.x == 0) atomicAdd(loss_sum, tot);
    }
}

extern "C" int cwfa_chain_bwd_f32(const float* z, const float* gz, const cwfa_chain* ch, const cwfa_chain_grads* grads,
                                  const int64_t* final_perm, float* gv0, int B, int C, int H, int W, int64_t z_bs, int64_t gz_bs,
                                  int64_t gv0_bs, float gscale, float ldscale, int accumulate, const float* gld, void* stream);

static bool chain_rows_ok(const cwfa_chain* ch, int C, int H, int W, int B, size_t* lds) {
    *lds = (size_t)(2 * ch->n_stages + 1) * W * sizeof(float);
    return *lds <= 60 * 1024 && C <= 65535 && B <= 65535 && W >= 64;
}

// 16-byte form usable?  W = 4 * (a divisor of 256), every row base and batch stride on a 16-byte boundary
static bool chain_rows4_ok(const cwfa_chain* ch, int C, int H, int W, int B, size_t* lds, const void* p0, const void* p1,
                           const void* p2, int64_t bs0, int64_t bs1, int64_t bs2) {
    if (W < 64 || (W & 3) || (W >> 2) > CH_BLOCK || CH_BLOCK % (W >> 2) != 0 || C > 65535 || B > 65535) return false;
    if (!cwfa_aligned16(p0) || !cwfa_aligned16(p1) || (p2 && !cwfa_aligned16(p2)) || (bs0 & 3) || (bs1 & 3) || (p2 && (bs2 & 3))) return false;
    for (int k = 0; k < ch->n_stages; ++k) {
        const cwfa_affine_stage& st = ch->stage[k];
        if ((st.s_raw && (!cwfa_aligned16(st.s_raw) || (st.s_bs & 3))) || (st.t && (!cwfa_aligned16(st.t) || (st.t_bs & 3)))) return false;
    }
    bool col = false;
    for (int k = 0; k < ch->n_stages; ++k) col = col || (ch->stage[k].perm && ch->stage[k].perm_axis == 3);
    *lds = col ? (size_t)2 * CH_BLOCK * 4 * sizeof(float) : 0;      // two exchange rows per block row, only for column gathers
    return *lds <= 64 * 1024;
}

static int check_chain(const char* name, const cwfa_chain* ch) {
    CWFA_REQUIRE(ch, CWFA_E_INVAL, "%s: null chain", name);
    CWFA_REQUIRE(ch->n_stages >= 0 && ch->n_stages <= CWFA_CHAIN_MAX, CWFA_E_INVAL, "%s: %d stages (max %d)", name,
                 ch->n_stages, CWFA_CHAIN_MAX);
    for (int k = 0; k < ch->n_stages; ++k) {
        int rc = check_stage(name, ch->stage[k]);
        if (rc) return rc;
        CWFA_REQUIRE(!ch->stage[k].gin, CWFA_E_INVAL, "%s: GIN stages cannot be chained", name);
    }
    return CWFA_OK;
}

extern "C" int cwfa_chain_inv_f32(const float* z, const float* low, float* x, const cwfa_chain* ch, int B, int C, int H,
                                  int W, int64_t z_bs, int64_t low_bs, int64_t x_bs, double* logdet, void* stream) {
    CWFA_REQUIRE(low && x, CWFA_E_INVAL, "cwfa_chain_inv_f32: null pointer");
    CWFA_REQUIRE(B >= 0 && C >= 0 && H >= 0 && W >= 0 && B <= 65535, CWFA_E_SHAPE, "cwfa_chain_inv_f32: bad shape");
    int rc = check_chain("cwfa_chain_inv_f32", ch);
    if (rc) return rc;
    const int64_t n = (int64_t)C * H * W;
    if (B == 0 || n == 0) return CWFA_OK;
    size_t lds;
    if (chain_rows4_ok(ch, C, H, W, B, &lds, low, x, z, low_bs, x_bs, z_bs)) {
        const int RB = CH_BLOCK * 4 / W;
        if (ch->n_stages <= 6)
            hipLaunchKernelGGL((chain_rows4_kernel<true, 6>), dim3((H + RB - 1) / RB, C, B), dim3(CH_BLOCK), lds, (hipStream_t)stream, low, x,
                               const_cast<float*>(z), *ch, (const int64_t*)nullptr, C, H, W, low_bs, x_bs, z_bs, logdet, (double*)nullptr);
        else
            hipLaunchKernelGGL((chain_rows4_kernel<true, CWFA_CHAIN_MAX>), dim3((H + RB - 1) / RB, C, B), dim3(CH_BLOCK), lds, (hipStream_t)stream, low, x,
                               const_cast<float*>(z), *ch, (const int64_t*)nullptr, C, H, W, low_bs, x_bs, z_bs, logdet, (double*)nullptr);
        CWFA_LAUNCH_CHECK("cwfa_chain_inv_f32");
        return CWFA_OK;
    }
    if (chain_rows_ok(ch, C, H, W, B, &lds)) {
        hipLaunchKernelGGL(chain_inv_rows_kernel, dim3(H, C, B), dim3(256), lds, (hipStream_t)stream, z, low, x, *ch, C, H, W,
                           z_bs, low_bs, x_bs, logdet);
        CWFA_LAUNCH_CHECK("cwfa_chain_inv_f32");
        return CWFA_OK;
    }
    dim3 grid((unsigned)((n + 255) / 256), B);
    hipLaunchKernelGGL(chain_inv_kernel, grid, dim3(256), 0, (hipStream_t)stream, z, low, x, *ch, C, H, W, z_bs, low_bs, x_bs,
                       logdet);
    CWFA_LAUNCH_CHECK("cwfa_chain_inv_f32");
    return CWFA_OK;
}

extern "C" int cwfa_chain_fwd_f32(const float* x, float* low, float* z, const cwfa_chain* ch, const int64_t* final_perm,
                                  int B, int C, int H, int W, int64_t x_bs, int64_t low_bs, int64_t z_bs, double* logdet,
                                  double* sumsq, void* stream) {
    CWFA_REQUIRE(x && low && z, CWFA_E_INVAL, "cwfa_chain_fwd_f32: null pointer");
    CWFA_REQUIRE(B >= 0 && C >= 0 && H >= 0 && W >= 0 && B <= 65535, CWFA_E_SHAPE, "cwfa_chain_fwd_f32: bad shape");
    int rc = check_chain("cwfa_chain_fwd_f32", ch);
    if (rc) return rc;
    const int64_t n = (int64_t)C * H * W;
    if (B == 0 || n == 0) return CWFA_OK;
    size_t lds;
    if (chain_rows4_ok(ch, C, H, W, B, &lds, x, low, z, x_bs, low_bs, z_bs)) {
        const int RB = CH_BLOCK * 4 / W;
        if (ch->n_stages <= 6)
            hipLaunchKernelGGL((chain_rows4_kernel<false, 6>), dim3((H + RB - 1) / RB, C, B), dim3(CH_BLOCK), lds, (hipStream_t)stream, x, low, z,
                               *ch, final_perm, C, H, W, x_bs, low_bs, z_bs, logdet, sumsq);
        else
            hipLaunchKernelGGL((chain_rows4_kernel<false, CWFA_CHAIN_MAX>), dim3((H + RB - 1) / RB, C, B), dim3(CH_BLOCK), lds, (hipStream_t)stream, x, low, z,
                               *ch, final_perm, C, H, W, x_bs, low_bs, z_bs, logdet, sumsq);
        CWFA_LAUNCH_CHECK("cwfa_chain_fwd_f32");
        return CWFA_OK;
    }
    if (chain_rows_ok(ch, C, H, W, B, &lds)) {
        hipLaunchKernelGGL(chain_fwd_rows_kernel, dim3(H, C, B), dim3(256), lds, (hipStream_t)stream, x, low, z, *ch, final_perm,
                           C, H, W, x_bs, low_bs, z_bs, logdet, sumsq);
        CWFA_LAUNCH_CHECK("cwfa_chain_fwd_f32");
        return CWFA_OK;
    }
    dim3 grid((unsigned)((n + 255) / 256), B);
    hipLaunchKernelGGL(chain_fwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, low, z, *ch, final_perm, C, H, W, x_bs,
                       low_bs, z_bs, logdet, sumsq);
    CWFA_LAUNCH_CHECK("cwfa_chain_fwd_f32");
    return CWFA_OK;
}

// ------------------------------------------------------------------------------------------------ small helpers
__global__ __launch_bounds__(256) void scale_channels_kernel(const float* __restrict__ x, const float* __restrict__ sc,
                                                             float* __restrict__ y, int64_t HW) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= HW) return;
    const int64_t bc = blockIdx.y;
    y[bc * HW + p] = x[bc * HW + p] * sc[bc];
}

extern "C" int cwfa_scale_channels_f32(const float* x, const float* scale_bc, float* y, int B, int C, int64_t HW,
                                       void* stream) {
    CWFA_REQUIRE(x && y && scale_bc, CWFA_E_INVAL, "cwfa_scale_channels_f32: null pointer");
    CWFA_REQUIRE(B >= 0 && C >= 0 && HW >= 0 && (int64_t)B * C <= 65535, CWFA_E_SHAPE, "cwfa_scale_channels_f32: bad shape");
    if (B == 0 || C == 0 || HW == 0) return CWFA_OK;
    dim3 grid((unsigned)((HW + 255) / 256), B * C);
    hipLaunchKernelGGL(scale_channels_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, scale_bc, y, HW);
    CWFA_LAUNCH_CHECK("cwfa_scale_channels_f32");
    return CWFA_OK;
}

__global__ __launch_bounds__(256) void axpby_kernel(const float* __restrict__ x, const float* __restrict__ z, float a,
                                                    float bb, float* __restrict__ y, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    y[i] = z ? a * x[i] + bb * z[i] : a * x[i];
}

extern "C" int cwfa_axpby_f32(const float* x, const float* z, float a, float b, float* y, int64_t n, void* stream) {
    CWFA_REQUIRE(x && y, CWFA_E_INVAL, "cwfa_axpby_f32: null pointer");
    CWFA_REQUIRE(n >= 0, CWFA_E_INVAL, "cwfa_axpby_f32: negative size");
    if (n == 0) return CWFA_OK;
    hipLaunchKernelGGL(axpby_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, z, a, b, y, n);
    CWFA_LAUNCH_CHECK("cwfa_axpby_f32");
    return CWFA_OK;
}

// ------------------------------------------------------------------------------------------------ lenslet views
// XLFMDatasetFull.extract_views (XLFMDataset.py:212-242) + the normalisation that follows it (CWFA.py:796-797):
// view n is the sh x sw window centred on lenslet n, clipped to the sensor image; the clipped patch sits in the
// BOTTOM-RIGHT corner of a zero-filled view (stacked_views[:, n, -ph:, -pw:] = patch), then (v - mean) / std everywhere.
__global__ __launch_bounds__(256) void extract_views_kernel(const float* __restrict__ img, const int* __restrict__ coords,
                                                            float* __restrict__ out, int Hs, int Ws, int nviews, int sh, int sw,
                                                            float mean, float stdv, int64_t img_bs) {
    const int n = blockIdx.y, b = blockIdx.z;
    const int cy = coords[2 * n], cx = coords[2 * n + 1];
    const int ly = max(cy - sh / 2, 0), lx = max(cx - sw / 2, 0);
    const int uy = min(cy + sh / 2, Hs), ux = min(cx + sw / 2, Ws);
    const int oy = sh - (uy - ly), ox = sw - (ux - lx);          // first output row / column holding image data
    const float* ib = img + (int64_t)b * img_bs;
    float* ob = out + ((int64_t)b * nviews + n) * sh * sw;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < sh * sw; e += gridDim.x * blockDim.x) {
        const int i = e / sw, j = e % sw;
        float v = 0.f;
        if (i >= oy && j >= ox) v = ib[(int64_t)(ly + i - oy) * Ws + (lx + j - ox)];
        ob[e] = (v - mean) / stdv;
    }
}

extern "C" int cwfa_extract_views_f32(const float* image, const int* coords_yx, float* views, int B, int Hs, int Ws, int nviews,
                                      int sh, int sw, float mean, float stdv, int64_t image_bs, void* stream) {
    CWFA_REQUIRE(image && coords_yx && views, CWFA_E_INVAL, "cwfa_extract_views_f32: null pointer");
    CWFA_REQUIRE(B >= 0 && nviews >= 0 && Hs > 0 && Ws > 0 && sh > 0 && sw > 0, CWFA_E_INVAL, "cwfa_extract_views_f32: bad size");
    CWFA_REQUIRE(nviews <= 65535 && B <= 65535, CWFA_E_SHAPE, "cwfa_extract_views_f32: grid too large");
    if (B == 0 || nviews == 0) return CWFA_OK;
    const int per = (sh * sw + 255) / 256;
    hipLaunchKernelGGL(extract_views_kernel, dim3(per < 256 ? per : 256, nviews, B), dim3(256), 0, (hipStream_t)stream, image,
                       coords_yx, views, Hs, Ws, nviews, sh, sw, mean, stdv, image_bs);
    CWFA_LAUNCH_CHECK("cwfa_extract_views_f32");
    return CWFA_OK;
}

// ------------------------------------------------------------------------------------------------ backward of ONE affine stage
// (torch autograd of a coupling block that is not part of a fused chain: GLOW / RNVP / GIN / NICE / one-sided / AllInOne,
//  coupling_layers.py:50-60,124-437; all_in_one_block.py:206-224).  No gather (the caller differentiates its permutation itself).
//   rev = 0:  y = e^s x + t      dL/dx = g e^s,   dL/dt = g,         dL/ds = g e^s x + gld_b
//   rev = 1:  y = (x - t) e^-s   dL/dx = g e^-s,  dL/dt = -g e^-s,   dL/ds = -g y - gld_b         (logdet_b = -+ sum s)
// s = clamp(pre * s_raw) [GIN: minus its channel mean at the pixel, log-det 0], t = pre * t_raw (or -t_raw / sqrt 2).
// One thread per pixel walks the channels (the GIN mean needs them all; the plain form just strides).
__global__ __launch_bounds__(256) void affine_bwd_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                         cwfa_affine_stage st, int rev, int C, int64_t HW, int64_t x_bs, int64_t g_bs,
                                                         const float* __restrict__ gld, float* __restrict__ gx,
                                                         float* __restrict__ gs, float* __restrict__ gt) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= HW) return;
    const int b = blockIdx.y;
    const int64_t n = (int64_t)C * HW;
    const float ld = gld ? gld[b] : 0.f;
    float mean = 0.f, dsm = 0.f;
    if (st.gin) {                                   // two extra sweeps: mean of s, then mean of dL/ds
        for (int c = 0; c < C; ++c) {
            float s, t;
            stage_st(st, b, c * HW + i, s, t);
            mean += s;
        }
        mean /= (float)C;
        for (int c = 0; c < C; ++c) {
            float s, t;
            stage_st(st, b, c * HW + i, s, t);
            s -= mean;
            const float xv = x[b * x_bs + c * HW + i], gv = g[b * g_bs + c * HW + i];
            dsm += rev ? -gv * ((xv - t) * expf(-s)) : gv * expf(s) * xv;
        }
        dsm /= (float)C;
    }
    for (int c = 0; c < C; ++c) {
        const int64_t off = c * HW + i;
        float s, t;
        stage_st(st, b, off, s, t);
        s -= mean;
        const float xv = x[b * x_bs + off], gv = g[b * g_bs + off];
        float dx, dt, ds;
        if (!rev) {
            const float e = expf(s);
            dx = gv * e;
            dt = gv;
            ds = dx * xv + (st.gin ? 0.f : ld);
        } else {
            const float e = expf(-s);
            dx = gv * e;
            dt = -dx;
            ds = -gv * ((xv - t) * e) - (st.gin ? 0.f : ld);
        }
        ds -= dsm;
        if (gx) gx[b * n + off] = dx;
        if (gs && st.s_raw) {
            const float a = st.s_raw[b * st.s_bs + off] * st.pre_scale;
            gs[b * n + off] = ds * soft_clamp_grad(a, st.clamp_kind, st.clamp) * st.pre_scale;
        }
        if (gt && st.t) gt[b * n + off] = st.t_neg_div_sqrt2 ? (-dt) / CWFA_SQRT2_F : dt * st.pre_scale;
    }
}

extern "C" int cwfa_affine_bwd_f32(const float* x, const float* g, const cwfa_affine_stage* st, int rev, int B, int C, int H, int W,
                                   int64_t x_bs, int64_t g_bs, const float* gld, float* gx, float* gs_raw, float* gt_raw, void* stream) {
    CWFA_REQUIRE(x && g && st, CWFA_E_INVAL, "cwfa_affine_bwd_f32: null pointer");
    CWFA_REQUIRE(B >= 0 && C >= 0 && H >= 0 && W >= 0 && B <= 65535, CWFA_E_SHAPE, "cwfa_affine_bwd_f32: bad shape");
    CWFA_REQUIRE(st->clamp_kind >= CWFA_CLAMP_NONE && st->clamp_kind <= CWFA_CLAMP_SIGMOID, CWFA_E_INVAL, "cwfa_affine_bwd_f32: bad clamp kind %d",
                 st->clamp_kind);
    CWFA_REQUIRE(!st->perm, CWFA_E_INVAL, "cwfa_affine_bwd_f32: a stage with a gather is differentiated by its caller (gather first)");
    CWFA_REQUIRE(!st->gin || st->s_raw, CWFA_E_INVAL, "cwfa_affine_bwd_f32: GIN needs s_raw");
    const int64_t HW = (int64_t)H * W;
    if (B == 0 || C == 0 || HW == 0) return CWFA_OK;
    hipLaunchKernelGGL(affine_bwd_kernel, dim3((unsigned)((HW + 255) / 256), B), dim3(256), 0, (hipStream_t)stream, x, g, *st, rev, C, HW,
                       x_bs, g_bs, gld, gx, gs_raw, gt_raw);
    CWFA_LAUNCH_CHECK("cwfa_affine_bwd_f32");
    return CWFA_OK;
}

static int check_chain(const char* name, const cwfa_chain* ch);
extern "C" int cwfa_chain_bwd_f32(const float* z, const float* gz, const cwfa_chain* ch, const cwfa_chain_grads* grads,
                                  const int64_t* final_perm, float* gv0, int B, int C, int H, int W, int64_t z_bs, int64_t gz_bs,
                                  int64_t gv0_bs, float gscale, float ldscale, int accumulate, const float* gld, void* stream) {
    CWFA_REQUIRE(z && grads, CWFA_E_INVAL, "cwfa_chain_bwd_f32: null pointer");
    CWFA_REQUIRE(B >= 0 && C >= 0 && H >= 0 && W >= 0 && B <= 65535, CWFA_E_SHAPE, "cwfa_chain_bwd_f32: bad shape");
    int rc = check_chain("cwfa_chain_bwd_f32", ch);
    if (rc) return rc;
    for (int k = 0; k < ch->n_stages; ++k)
        CWFA_REQUIRE(!grads->ds[k] || ch->stage[k].s_raw, CWFA_E_INVAL, "cwfa_chain_bwd_f32: stage %d has no s but a ds buffer", k);
    const int64_t n = (int64_t)C * H * W;
    if (B == 0 || n == 0) return CWFA_OK;
    dim3 grid((unsigned)((n + 255) / 256), B);
    hipLaunchKernelGGL(chain_bwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, z, gz, *ch, *grads, final_perm, gv0, C, H, W, z_bs,
                       gz_bs, gv0_bs, gscale, ldscale, accumulate, gld);
    CWFA_LAUNCH_CHECK("cwfa_chain_bwd_f32");
    return CWFA_OK;
}

extern "C" int cwfa_chain_inv_bwd_f32(const float* xhat, const float* gt, const cwfa_chain* ch, const cwfa_chain_grads* grads, int B,
                                      int C, int H, int W, int64_t xhat_bs, int64_t gt_bs, float gscale, int loss_kind, int accumulate,
                                      double* loss_sum, float* gz_out, float* glow_out, void* stream) {
    CWFA_REQUIRE(xhat && gt && grads, CWFA_E_INVAL, "cwfa_chain_inv_bwd_f32: null pointer");
    CWFA_REQUIRE(B >= 0 && C >= 0 && H >= 0 && W >= 0 && B <= 65535, CWFA_E_SHAPE, "cwfa_chain_inv_bwd_f32: bad shape");
    CWFA_REQUIRE(loss_kind >= 0 && loss_kind <= 2, CWFA_E_INVAL, "cwfa_chain_inv_bwd_f32: loss_kind %d (0 = upstream gradient, 1 = L1, 2 = L2)", loss_kind);
    int rc = check_chain("cwfa_chain_inv_bwd_f32", ch);
    if (rc) return rc;
    for (int k = 0; k < ch->n_stages; ++k)
        CWFA_REQUIRE(!grads->ds[k] || ch->stage[k].s_raw, CWFA_E_INVAL, "cwfa_chain_inv_bwd_f32: stage %d has no s but a ds buffer", k);
    const int64_t n = (int64_t)C * H * W;
    if (B == 0 || n == 0) return CWFA_OK;
    dim3 grid((unsigned)((n + 255) / 256), B);
    hipLaunchKernelGGL(chain_inv_bwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, xhat, gt, *ch, *grads, C, H, W, xhat_bs, gt_bs,
                       gscale, loss_kind, accumulate, loss_sum, gz_out, glow_out);
    CWFA_LAUNCH_CHECK("cwfa_chain_inv_bwd_f32");
    return CWFA_OK;
}
