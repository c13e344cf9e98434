// Fused condition-net 3-D stage:  Conv3d(1->K, 3^3) -> PReLU -> Conv3d(K->1, 3^3), zero padding 1 on both,
// over the (H, W, D) volume whose depth axis is the CHANNEL axis of the [B,D,H,W] tensor (networks.py:221-225,239).
//
// Unfused, the K=32 hidden volume costs (4 + 128 + 256 + 128 + 4) bytes/voxel of HBM traffic; here the hidden channel
// only ever exists as one LDS tile: algorithmic traffic is 8 bytes/voxel and the kernel is VALU-bound
// (2*27*K FMAs per voxel).  Per block: output tile DT x 8 x 32 voxels; the haloed input tile (+2 each side) is staged
// once, then for every hidden channel k: phase A computes PReLU(conv1_k) on the +1 halo (zeroed outside the volume,
// because conv2's padding pads the HIDDEN volume), phase B accumulates conv2_k into DT register accumulators per
// thread.  Weights are wave-uniform and come through scalar loads.
#include "common.h"

namespace {

template <int DT>
struct C3 {
    static constexpr int HT = 8, WT = 32, NTHREADS = 256;
    static constexpr int XD = DT + 4, XH = HT + 4, XW = WT + 4;
    static constexpr int HD = DT + 2, HH = HT + 2, HWd = WT + 2;
    static constexpr int XS = XD * XH * XW, HS = HD * HH * HWd;
};

template <int DT>
__global__ __launch_bounds__(256) void conv3d_1k1_kernel(const float* __restrict__ x, const float* __restrict__ w1,
                                                         const float* __restrict__ b1, const float* __restrict__ alpha_p,
                                                         const float* __restrict__ w2, const float* __restrict__ b2,
                                                         float* __restrict__ y, int D, int H, int W, int K, int tiles_w,
                                                         int tiles_h) {
    typedef C3<DT> C;
    __shared__ float Xs[C::XS];
    __shared__ float Hs[C::HS];
    const int tid = threadIdx.x;
    const int tw = blockIdx.x % tiles_w, th = blockIdx.x / tiles_w;
    const int d0 = blockIdx.y * DT, h0 = th * C::HT, w0 = tw * C::WT, b = blockIdx.z;
    const int64_t HW = (int64_t)H * W;
    const float* xb = x + (int64_t)b * D * HW;
    const float alpha = *alpha_p;

    for (int e = tid; e < C::XS; e += C::NTHREADS) {
        const int dd = e / (C::XH * C::XW), rem = e % (C::XH * C::XW), hh = rem / C::XW, ww = rem % C::XW;
        const int gd = d0 + dd - 2, gh = h0 + hh - 2, gw = w0 + ww - 2;
        float v = 0.f;
        if (gd >= 0 && gd < D && gh >= 0 && gh < H && gw >= 0 && gw < W) v = xb[(int64_t)gd * HW + (int64_t)gh * W + gw];
        Xs[e] = v;
    }

    const int oh = tid / C::WT, ow = tid % C::WT;      // this thread's output column (8 x 32 threads)
    float out[DT];
#pragma unroll
    for (int i = 0; i < DT; ++i) out[i] = 0.f;

    for (int k = 0; k < K; ++k) {
        __syncthreads();     // Xs ready (k == 0) / previous phase B done reading Hs
        // ---- phase A: hidden channel k on the +1 halo
        const float* wk = w1 + k * 27;
        const float bk = b1[k];
        for (int q = tid; q < C::HH * C::HWd; q += C::NTHREADS) {
            const int hh = q / C::HWd, ww = q % C::HWd;
            float hv[C::HD];
#pragma unroll
            for (int i = 0; i < C::HD; ++i) hv[i] = bk;
#pragma unroll
            for (int dh = 0; dh < 3; ++dh)
#pragma unroll
                for (int dw = 0; dw < 3; ++dw) {
                    float col[C::XD];
#pragma unroll
                    for (int i = 0; i < C::XD; ++i) col[i] = Xs[(i * C::XH + hh + dh) * C::XW + ww + dw];
#pragma unroll
                    for (int kd = 0; kd < 3; ++kd) {
                        const float wv = wk[(dh * 3 + dw) * 3 + kd];
#pragma unroll
                        for (int i = 0; i < C::HD; ++i) hv[i] = fmaf(wv, col[i + kd], hv[i]);
                    }
                }
            const int gh = h0 + hh - 1, gw = w0 + ww - 1;
            const bool in_hw = gh >= 0 && gh < H && gw >= 0 && gw < W;
#pragma unroll
            for (int i = 0; i < C::HD; ++i) {
                const int gd = d0 + i - 1;
                float v = hv[i];
                v = v > 0.f ? v : alpha * v;
                if (!(in_hw && gd >= 0 && gd < D)) v = 0.f;
                Hs[(i * C::HH + hh) * C::HWd + ww] = v;
            }
        }
        __syncthreads();
        // ---- phase B: accumulate conv2 of hidden channel k
        const float* vk = w2 + k * 27;
#pragma unroll
        for (int dh = 0; dh < 3; ++dh)
#pragma unroll
            for (int dw = 0; dw < 3; ++dw) {
                float col[C::HD];
#pragma unroll
                for (int i = 0; i < C::HD; ++i) col[i] = Hs[(i * C::HH + oh + dh) * C::HWd + ow + dw];
#pragma unroll
                for (int kd = 0; kd < 3; ++kd) {
                    const float wv = vk[(dh * 3 + dw) * 3 + kd];
#pragma unroll
                    for (int i = 0; i < DT; ++i) out[i] = fmaf(wv, col[i + kd], out[i]);
                }
            }
    }
    const int gh = h0 + oh, gw = w0 + ow;
    if (gh < H && gw < W) {
        const float bb = b2[0];
        float* yb = y + (int64_t)b * D * HW + (int64_t)gh * W + gw;
#pragma unroll
        for (int i = 0; i < DT; ++i)
            if (d0 + i < D) yb[(int64_t)(d0 + i) * HW] = out[i] + bb;
    }
}

}  // namespace

extern "C" int cwfa_conv3d_1k1_f32(const float* x, const float* w1, const float* b1, const float* alpha, const float* w2,
                                   const float* b2, float* y, int B, int D, int H, int W, int K, void* stream) {
    CWFA_REQUIRE(x && w1 && b1 && alpha && w2 && b2 && y, CWFA_E_INVAL, "cwfa_conv3d_1k1_f32: null pointer");
    CWFA_REQUIRE(B >= 0 && D >= 0 && H >= 0 && W >= 0 && K > 0, CWFA_E_INVAL, "cwfa_conv3d_1k1_f32: bad size");
    CWFA_REQUIRE(x != y, CWFA_E_INVAL, "cwfa_conv3d_1k1_f32: in-place not supported");
    if (B == 0 || D == 0 || H == 0 || W == 0) return CWFA_OK;
    const int tiles_w = (W + 31) / 32, tiles_h = (H + 7) / 8;
    const bool d6 = D % 6 == 0;
    const int DT = d6 ? 6 : 8;
    const int tiles_d = (D + DT - 1) / DT;
    CWFA_REQUIRE(tiles_d <= 65535 && B <= 65535, CWFA_E_SHAPE, "cwfa_conv3d_1k1_f32: grid too large");
    dim3 grid(tiles_w * tiles_h, tiles_d, B);
    if (d6)
        hipLaunchKernelGGL(conv3d_1k1_kernel<6>, grid, dim3(256), 0, (hipStream_t)stream, x, w1, b1, alpha, w2, b2, y, D, H, W,
                           K, tiles_w, tiles_h);
    else
        hipLaunchKernelGGL(conv3d_1k1_kernel<8>, grid, dim3(256), 0, (hipStream_t)stream, x, w1, b1, alpha, w2, b2, y, D, H, W,
                           K, tiles_w, tiles_h);
    CWFA_LAUNCH_CHECK("cwfa_conv3d_1k1_f32");
    return CWFA_OK;
}
