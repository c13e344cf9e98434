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


// ---- matrix-core version (K <= 32) ---------------------------------------------------------------------------------
// Both convolutions as 32x32x2 fp32 MFMA GEMMs over one row of 32 hidden voxels (lanes = w):
//   conv1:  hidden[k][w] = b1[k] + sum_tap w1[k][tap] * x[tap-shifted w]      M = k (32), N = 32 voxels, K = 27 taps (14 MFMAs)
//   conv2:  P[tap][w]    = sum_k w2[k][tap] * hidden[k][w]                    M = tap (27 of 32), K = 32 channels (16 MFMAs;
//           conv1's accumulator registers ARE conv2's B operand: register r of lane (w, kh) holds channel rc(r)+4kh,
//           which is the k-pair layout the instruction wants)
// and P[tap][w] is the contribution of hidden voxel (d',h',w) to output voxel (d'-dd, h'-dh, w-dw): it is scattered by LDS
// read-modify-write into a PADDED per-wave private copy of the output tile (fixed program order => deterministic; LDS
// float atomics were 2x slower; the padding absorbs out-of-tile targets so the scatter needs no validity tests), and every
// output slab is the sum of exactly two private copies.  Block = 6 x 8 x 30 output voxels; the +1 hidden halo makes rows exactly 32 wide.
struct C3M {
    static constexpr int DT = 6, HT = 8, WT = 30, NTHREADS = 256, NWAVES = 4;
    static constexpr int XD = DT + 4, XH = HT + 4, XW = 36;     // staged input (origin -2), 34 used columns
    static constexpr int HD = DT + 2, HH = HT + 2;               // hidden slabs x rows (origin -1), 32 columns
    static_assert(HD == 2 * NWAVES, "every wave owns two hidden slabs");
    static constexpr int XS = XD * XH * XW;
    // per-wave private output copy, padded so that NO scatter target needs a validity test: 4 depth slabs (a wave owns two
    // hidden slabs), HT + 4 rows, 32 + 2 (+2 pad) columns; contributions that fall outside the tile land in the padding
    static constexpr int PS = 4, PH = HT + 4, PW = 36, OS = PS * PH * PW;
};

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4_c3 __attribute__((ext_vector_type(4)));

// batch of tap t in the scatter (found by exhaustive search: 10 of the 12 two-tap registers keep both halves together)
__host__ __device__ constexpr int c3m_batch(int t) {
    constexpr int F[9] = {0, 2, 1, 2, 1, 1, 2, 1, 0};
    return t < 27 ? ((t / 3) % 3 + F[t % 3 + 3 * (t / 9)]) % 3 : -1;
}

__global__ __launch_bounds__(256) void conv3d_1k1_mfma_kernel(const float* __restrict__ x, const float* __restrict__ w1,
                                                              const float* __restrict__ b1,
                                                              const float* __restrict__ alpha_p,
                                                              const float* __restrict__ w2, const float* __restrict__ b2,
                                                              float* __restrict__ y, int D, int H, int W, int K,
                                                              int tiles_w) {
    typedef C3M C;
    __shared__ float Xs[C::XS];
    __shared__ __attribute__((aligned(16))) float Os[C::NWAVES * C::OS + C::NTHREADS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wl = lane & 31, kh = lane >> 5;
    const int tw = blockIdx.x % tiles_w, th = blockIdx.x / tiles_w;
    const int d0 = blockIdx.y * C::DT, h0 = th * C::HT, w0 = tw * C::WT, b = blockIdx.z;
    const int64_t HW = (int64_t)H * W;
    const float* xb = x + (int64_t)b * D * HW;
    const float alpha = *alpha_p;

    {   // all loads of the haloed tile in flight together (clamped address, masked value), then the LDS writes
        constexpr int NI = (C::XS + C::NTHREADS - 1) / C::NTHREADS;
        float v[NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int e = min(tid + i * C::NTHREADS, C::XS - 1);
            const int dd = e / (C::XH * C::XW), rem = e % (C::XH * C::XW), hh = rem / C::XW, ww = rem % C::XW;
            const int gd = d0 + dd - 2, gh = h0 + hh - 2, gw = w0 + ww - 2;
            const bool in = gd >= 0 && gd < D && gh >= 0 && gh < H && gw >= 0 && gw < W;
            const float t = xb[(int64_t)min(max(gd, 0), D - 1) * HW + (int64_t)min(max(gh, 0), H - 1) * W + min(max(gw, 0), W - 1)];
            v[i] = in ? t : 0.f;
        }
#pragma unroll
        for (int i = 0; i < NI; ++i)
            if (tid + i * C::NTHREADS < C::XS) Xs[tid + i * C::NTHREADS] = v[i];
    }
    for (int e = tid; e < C::NWAVES * C::OS / 4; e += C::NTHREADS) reinterpret_cast<f32x4_c3*>(Os)[e] = f32x4_c3{0.f, 0.f, 0.f, 0.f};

    // operand panels, resident in registers for the whole block.  conv1's bias rides on the unused 28th tap
    // (MFMA 13, k-slot 1: A = b1[k], B = 1).  relp[r] = where accumulator register r of this lane scatters to, relative
    // to the hidden row's base cell in the wave's padded private copy.
    float a1[14], a2[16];
    int toff[14], relp[16];
#pragma unroll
    for (int j = 0; j < 14; ++j) {
        const int tap = 2 * j + kh;
        const int kk = min(wl, K - 1);
        const float wv = w1[kk * 27 + min(tap, 26)], bv = b1[kk];
        a1[j] = wl < K ? (tap < 27 ? wv : bv) : 0.f;
        const int t = tap < 27 ? tap : 0;
        toff[j] = ((t % 3) * C::XH + t / 9) * C::XW + (t / 3) % 3;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int k = (r & 3) + 8 * (r >> 2) + 4 * kh;
        const float w2v = w2[min(k, K - 1) * 27 + min(wl, 26)];
        a2[r] = (k < K && wl < 27) ? w2v : 0.f;
        const int tap = k < 27 ? k : 0;                        // same index formula: row of P held by (r, kh)
        const int dd = tap % 3, dw = (tap / 3) % 3, dh = tap / 9;
        relp[r] = (-dd * C::PH - dh) * C::PW + wl - dw;
    }
    __syncthreads();

    const int dummy = C::NWAVES * C::OS + tid;                 // per-lane sink for the lanes that sit out a batch
    const int gw = w0 - 1 + wl;
    const bool in_w = gw >= 0 && gw < W;
    const bool alpha_le1 = alpha >= 0.f && alpha <= 1.f;                     // wave-uniform
    const bool w_edge = w0 - 1 < 0 || w0 - 1 + 31 >= W;                      // block-uniform: some hidden columns are padding
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    f32x16 Pp = zero16;          // conv2 products of the previous row, scattered under the next row's conv1 MFMAs
    int pbase = wave * C::OS + (2 * C::PH + 2) * C::PW + 2;
    // LDS read-modify-write in three alias-free batches: two contributions can only meet in one output voxel when their
    // taps share (dd, dh) and differ in dw, so a batch takes at most one tap of each (dd, dh) triple (c3m_batch) and its
    // reads can all be in flight before its writes.  A lane half that is not in the batch goes to the sink.
    auto scatter = [&](const f32x16& P, int base) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float old[15];
            int at[15];
#pragma unroll
            for (int r = 0; r < 15; ++r) {
                const int t0 = (r & 3) + 8 * (r >> 2);
                const bool h0 = c3m_batch(t0) == c, h1 = c3m_batch(t0 + 4) == c;
                if (!h0 && !h1) continue;
                at[r] = base + relp[r];
                if (!(h0 && h1)) at[r] = (kh ? h1 : h0) ? at[r] : dummy;
                old[r] = Os[at[r]];
            }
#pragma unroll
            for (int r = 0; r < 15; ++r) {
                const int t0 = (r & 3) + 8 * (r >> 2);
                if (c3m_batch(t0) != c && c3m_batch(t0 + 4) != c) continue;
                Os[at[r]] = old[r] + P[r];
            }
        }
    };
    // wave w owns the hidden slabs d' = 2w, 2w+1 (all HH rows each)
    for (int i = 0; i < 2 * C::HH; ++i) {
        const int dl = i / C::HH, hp = i % C::HH, dp = 2 * wave + dl;
        const int gd = d0 - 1 + dp, gh = h0 - 1 + hp;
        if (gd < 0 || gd >= D || gh < 0 || gh >= H) continue;      // hidden row is conv2's zero padding (wave-uniform)
        const float* xr = Xs + (dp * C::XH + hp) * C::XW + wl;
        float xv[14];
#pragma unroll
        for (int j = 0; j < 14; ++j) xv[j] = xr[toff[j]];
        xv[13] = kh ? 1.f : xv[13];
        f32x16 hid = zero16;
#pragma unroll
        for (int j = 0; j < 14; ++j) hid = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], xv[j], hid, 0, 0, 0);
        scatter(Pp, pbase);
        // PReLU: max(v, alpha v) for 0 <= alpha <= 1 (2 instructions); the column mask only where the row leaves the volume
        if (alpha_le1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) hid[r] = fmaxf(hid[r], alpha * hid[r]);
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) hid[r] = hid[r] > 0.f ? hid[r] : alpha * hid[r];
        }
        if (w_edge) {
#pragma unroll
            for (int r = 0; r < 16; ++r) hid[r] = in_w ? hid[r] : 0.f;
        }
        f32x16 P = zero16;
#pragma unroll
        for (int r = 0; r < 16; ++r) P = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[r], hid[r], P, 0, 0, 0);
        Pp = P;
        pbase = wave * C::OS + ((dl + 2) * C::PH + hp + 2) * C::PW + 2;
    }
    scatter(Pp, pbase);
    __syncthreads();
    const float bb = b2[0];
    // output slab od collects exactly two private copies: wave od/2 (its slab (od&1)+2) and wave od/2+1 (its slab od&1)
    for (int e = tid; e < C::DT * C::HT * 32; e += C::NTHREADS) {
        const int ow = e & 31, oh = (e >> 5) % C::HT, od = e / (32 * C::HT);
        const int gdo = d0 + od, gho = h0 + oh, gwo = w0 + ow;
        if (ow < C::WT && gdo < D && gho < H && gwo < W) {
            const int wa = od >> 1, cell = (oh + 2) * C::PW + ow + 2;
            const float va = Os[wa * C::OS + ((od & 1) + 2) * C::PH * C::PW + cell];
            const float vb = Os[(wa + 1) * C::OS + (od & 1) * C::PH * C::PW + cell];
            y[(int64_t)b * D * HW + (int64_t)gdo * HW + (int64_t)gho * W + gwo] = (va + vb) + bb;
        }
    }
}

}  // namespace

extern "C" int cwfa_conv3d_1k1_f32(const float* x, const float* w1, const float* b1, const float* alpha, const float* w2,
                                   const float* b2, float* y, int B, int D, int H, int W, int K, void* stream) {
    CWFA_REQUIRE(x && w1 && b1 && alpha && w2 && b2 && y, CWFA_E_INVAL, "cwfa_conv3d_1k1_f32: null pointer");
    CWFA_REQUIRE(B >= 0 && D >= 0 && H >= 0 && W >= 0 && K > 0, CWFA_E_INVAL, "cwfa_conv3d_1k1_f32: bad size");
    CWFA_REQUIRE(x != y, CWFA_E_INVAL, "cwfa_conv3d_1k1_f32: in-place not supported");
    if (B == 0 || D == 0 || H == 0 || W == 0) return CWFA_OK;
    if (K <= 32) {
        const int tw = (W + C3M::WT - 1) / C3M::WT, th = (H + C3M::HT - 1) / C3M::HT, td = (D + C3M::DT - 1) / C3M::DT;
        CWFA_REQUIRE(td <= 65535 && B <= 65535, CWFA_E_SHAPE, "cwfa_conv3d_1k1_f32: grid too large");
        hipLaunchKernelGGL(conv3d_1k1_mfma_kernel, dim3(tw * th, td, B), dim3(256), 0, (hipStream_t)stream, x, w1, b1, alpha,
                           w2, b2, y, D, H, W, K, tw);
        CWFA_LAUNCH_CHECK("cwfa_conv3d_1k1_f32");
        return CWFA_OK;
    }
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
