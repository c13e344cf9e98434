// Implicit-GEMM 2-D convolution on the fp32 matrix cores of gfx950 (v_mfma_f32_32x32x2_f32), no im2col.
//
// GEMM view (per image):  Y[co][pix] = sum_{tap, ci} Wt[tap][ci][co] * X[ci][pix + tap]
//   M = output channels  -> MFMA rows   (A operand = weights, lane = (co & 31, k = lane >> 5))
//   N = pixels along W   -> MFMA cols   (B operand = input,   lane = (k = lane >> 5, pix & 31))
//   K = (tap, ci) pairs of input channels per MFMA
// so that an accumulator register holds 32 CONSECUTIVE PIXELS of one output channel per half-wave: NCHW stores are
// 128-byte coalesced and the next 1x1 conv / epilogue can consume the accumulators without any lane movement.
//
// Block = WM x WN waves; each wave owns MT x NT sub-tiles of 32 channels x (1 row x 32 pixels).
// Per K-chunk of CK input channels the block stages the haloed input tile [CK][TR+ks-1][32+ks-1] and the weight
// panel [ks*ks][CK][CT] in LDS (global loads of chunk i+1 are issued into registers before the MFMAs of chunk i),
// every LDS operand read is a conflict-free ds_read_b32 with a compile-time immediate offset.
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

template <int KS_, int CK_, int MT_, int NT_, int WM_, int WN_>
struct Cfg {
    static constexpr int KS = KS_, CK = CK_, MT = MT_, NT = NT_, WM = WM_, WN = WN_;
    static constexpr int PAD = KS / 2;
    static constexpr int NTHREADS = 64 * WM * WN;
    static constexpr int CT = 32 * MT * WM;          // output channels per block
    static constexpr int TR = NT * WN;               // output rows per block
    static constexpr int TC = 32;                    // output columns per block
    static constexpr int XR = TR + KS - 1;
    static constexpr int XC = TC + KS - 1;
    static constexpr int XS = CK * XR * XC;          // floats of the input tile
    static constexpr int XS_PAD = (XS + 3) & ~3;
    static constexpr int WS = KS * KS * CK * CT;     // floats of the weight panel (multiple of 4)
    static constexpr int XPT = (XS + NTHREADS - 1) / NTHREADS;
    static constexpr int WPT = (WS / 4 + NTHREADS - 1) / NTHREADS;
    static constexpr int LDS_BYTES = (XS_PAD + WS) * 4;
};

struct ConvParams {
    const float* x;
    const float* wp;
    float* y;
    int B, Cin, H, W, Cout, nchunks, tiles_x, tiles_y;
    int64_t x_bs, y_bs;
    cwfa_conv_opts o;
};

template <class C>
__global__ __launch_bounds__(C::NTHREADS) void conv2d_mfma_kernel(ConvParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Xs = smem;
    float* Ws = smem + C::XS_PAD;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / C::WN, wn = wave % C::WN;
    const int tx = blockIdx.x % p.tiles_x, ty = blockIdx.x / p.tiles_x;
    const int ct = blockIdx.y, b = blockIdx.z;
    const int row0 = ty * C::TR, col0 = tx * C::TC;
    const int64_t HW = (int64_t)p.H * p.W;

    // ---- per-thread staging map for the input tile (identical for every chunk)
    int xoff[C::XPT];
    int xcl[C::XPT];        // local channel, -1 = not loaded
#pragma unroll
    for (int i = 0; i < C::XPT; ++i) {
        const int e = tid + i * C::NTHREADS;
        const int c = e / (C::XR * C::XC), rem = e % (C::XR * C::XC);
        const int r = rem / C::XC, cc = rem % C::XC;
        const int gr = row0 + r - C::PAD, gc = col0 + cc - C::PAD;
        const bool ok = e < C::XS && gr >= 0 && gr < p.H && gc >= 0 && gc < p.W;
        xcl[i] = ok ? c : -1;
        xoff[i] = ok ? (int)(c * HW + (int64_t)gr * p.W + gc) : 0;
    }
    const float* xb = p.x + (int64_t)b * p.x_bs;
    const float* ab = p.o.in_add ? p.o.in_add + (int64_t)b * p.o.in_add_bs : nullptr;
    const float* wb = p.wp + (int64_t)ct * p.nchunks * C::WS;

    float xr[C::XPT];
    float4 wr[C::WPT];

    auto prefetch = [&](int chunk) {
        const int c0 = chunk * C::CK;
        const int64_t cbase = (int64_t)c0 * HW;
#pragma unroll
        for (int i = 0; i < C::XPT; ++i) {
            float v = 0.f;
            if (xcl[i] >= 0 && c0 + xcl[i] < p.Cin) {
                v = xb[cbase + xoff[i]];
                if (p.o.in_scale) {
                    const int ai = b * p.o.in_affine_bs + c0 + xcl[i];
                    v = v * p.o.in_scale[ai] + p.o.in_shift[ai];
                }
                if (ab) v += ab[cbase + xoff[i]];
            }
            xr[i] = v;
        }
        const float4* w4 = reinterpret_cast<const float4*>(wb + (int64_t)chunk * C::WS);
#pragma unroll
        for (int i = 0; i < C::WPT; ++i) {
            const int e = tid + i * C::NTHREADS;
            wr[i] = e < C::WS / 4 ? w4[e] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int i = 0; i < C::XPT; ++i) {
            const int e = tid + i * C::NTHREADS;
            if (e < C::XS) Xs[e] = xr[i];
        }
#pragma unroll
        for (int i = 0; i < C::WPT; ++i) {
            const int e = tid + i * C::NTHREADS;
            if (e < C::WS / 4) reinterpret_cast<float4*>(Ws)[e] = wr[i];
        }
    };

    f32x16 acc[C::MT][C::NT];
#pragma unroll
    for (int m = 0; m < C::MT; ++m)
#pragma unroll
        for (int n = 0; n < C::NT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

    const int kh = lane >> 5, l31 = lane & 31;
    const float* wlane = Ws + kh * C::CT + (wm * C::MT) * 32 + l31;
    const float* xlane = Xs + kh * (C::XR * C::XC) + (wn * C::NT) * C::XC + l31;

    prefetch(0);
    for (int chunk = 0; chunk < p.nchunks; ++chunk) {
        if (chunk) __syncthreads();
        commit();
        __syncthreads();
        if (chunk + 1 < p.nchunks) prefetch(chunk + 1);
#pragma unroll
        for (int tap = 0; tap < C::KS * C::KS; ++tap) {
            const int dy = tap / C::KS, dx = tap % C::KS;
#pragma unroll
            for (int kk = 0; kk < C::CK / 2; ++kk) {
                float a[C::MT], bv[C::NT];
#pragma unroll
                for (int m = 0; m < C::MT; ++m) a[m] = wlane[(tap * C::CK + 2 * kk) * C::CT + m * 32];
#pragma unroll
                for (int n = 0; n < C::NT; ++n) bv[n] = xlane[(2 * kk) * (C::XR * C::XC) + (n + dy) * C::XC + dx];
#pragma unroll
                for (int m = 0; m < C::MT; ++m)
#pragma unroll
                    for (int n = 0; n < C::NT; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m], bv[n], acc[m][n], 0, 0, 0);
            }
        }
    }

    // ---- epilogue: bias, activation, residual, activation, (pixel-shuffled) store
    const float alpha = (p.o.prelu_alpha && (p.o.act == CWFA_ACT_PRELU || p.o.act2 == CWFA_ACT_PRELU)) ? *p.o.prelu_alpha : 0.f;
    const int col = col0 + l31;
    const int Co = p.o.upshuffle2 ? p.Cout / 4 : p.Cout;
    float* yb = p.y + (int64_t)b * p.y_bs;
    const float* rb = p.o.residual ? p.o.residual + (int64_t)b * p.o.res_bs : nullptr;
#pragma unroll
    for (int m = 0; m < C::MT; ++m) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = ct * C::CT + (wm * C::MT + m) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
            if (co >= p.Cout) continue;
            const int cb = p.o.upshuffle2 ? co % Co : co;
            const float bias = p.o.bias ? p.o.bias[cb] : 0.f;
#pragma unroll
            for (int n = 0; n < C::NT; ++n) {
                const int row = row0 + wn * C::NT + n;
                if (row >= p.H || col >= p.W) continue;
                int64_t o;
                if (p.o.upshuffle2) {
                    const int q = co / Co;
                    o = (int64_t)cb * (4 * HW) + (int64_t)(2 * row + (q >> 1)) * (2 * p.W) + 2 * col + (q & 1);
                } else {
                    o = (int64_t)co * HW + (int64_t)row * p.W + col;
                }
                float v = cwfa_act(acc[m][n][r] + bias, p.o.act, alpha);
                if (rb) v += rb[o];
                v = cwfa_act(v, p.o.act2, alpha);
                yb[o] = v;
            }
        }
    }
}

// ---- weight repack: torch [Cout][Cin][ks][ks] -> [cout tile][chunk][tap][ck][CT]   (zeros beyond Cout / Cin)
__global__ __launch_bounds__(256) void pack_kernel(const float* __restrict__ w, float* __restrict__ out, int Cout, int Cin,
                                                   int ks, int CT, int CK, int nchunks, int64_t total, int transposed) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int taps = ks * ks;
    const int col = (int)(i % CT);
    const int ck = (int)((i / CT) % CK);
    const int tap = (int)((i / ((int64_t)CT * CK)) % taps);
    const int chunk = (int)((i / ((int64_t)CT * CK * taps)) % nchunks);
    const int ctile = (int)(i / ((int64_t)CT * CK * taps * nchunks));
    const int co = ctile * CT + col, ci = chunk * CK + ck;
    float v = 0.f;
    if (co < Cout && ci < Cin) {
        if (transposed) {   // ConvTranspose2d [Cin][Co][2][2] seen as a 1x1 conv with 4*Co outputs, co = q*Co + c
            const int Co = Cout / 4, q = co / Co, c = co % Co;
            v = w[((int64_t)ci * Co + c) * 4 + q];
        } else {
            v = w[((int64_t)co * Cin + ci) * taps + tap];
        }
    }
    out[i] = v;
}

// ---- configuration table.  One entry per (ks, Cout class); pack and launch MUST agree, hence one selector.
typedef Cfg<3, 8, 1, 4, 1, 4> C3_32;      // Cout <= 32 : 32 ch x 16 rows x 32 cols, 256 threads
typedef Cfg<3, 8, 2, 2, 1, 8> C3_64;      // Cout <= 64 : 64 ch x 16 rows x 32 cols, 512 threads
typedef Cfg<3, 8, 2, 2, 2, 4> C3_128;     // Cout  > 64 : 128 ch x 8 rows x 32 cols, 512 threads
typedef Cfg<1, 16, 1, 4, 1, 4> C1_32;
typedef Cfg<1, 16, 2, 2, 1, 8> C1_64;
typedef Cfg<1, 16, 2, 2, 2, 4> C1_128;
typedef Cfg<7, 4, 1, 4, 1, 4> C7_32;
typedef Cfg<7, 4, 2, 2, 1, 8> C7_64;

struct Sel {
    int id, CT, CK;
};

Sel select_cfg(int ks, int Cout) {
    const int cls = Cout <= 32 ? 0 : (Cout <= 64 ? 1 : 2);
    if (ks == 3) return cls == 0 ? Sel{0, C3_32::CT, C3_32::CK} : cls == 1 ? Sel{1, C3_64::CT, C3_64::CK} : Sel{2, C3_128::CT, C3_128::CK};
    if (ks == 1) return cls == 0 ? Sel{3, C1_32::CT, C1_32::CK} : cls == 1 ? Sel{4, C1_64::CT, C1_64::CK} : Sel{5, C1_128::CT, C1_128::CK};
    if (ks == 7) return cls == 0 ? Sel{6, C7_32::CT, C7_32::CK} : Sel{7, C7_64::CT, C7_64::CK};
    return Sel{-1, 0, 0};
}

template <class C>
int launch(const ConvParams& p0, hipStream_t stream) {
    ConvParams p = p0;
    p.tiles_x = (p.W + C::TC - 1) / C::TC;
    p.tiles_y = (p.H + C::TR - 1) / C::TR;
    p.nchunks = (p.Cin + C::CK - 1) / C::CK;
    const int ctiles = (p.Cout + C::CT - 1) / C::CT;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv2d_mfma_kernel<C>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
        if (e != hipSuccess) {
            cwfa_set_error("cwfa_conv2d_f32: hipFuncSetAttribute(%d bytes LDS): %s", C::LDS_BYTES, hipGetErrorString(e));
            return CWFA_E_HIP;
        }
        attr_set = true;
    }
    CWFA_REQUIRE((int64_t)p.tiles_x * p.tiles_y < (1ll << 31) && ctiles <= 65535 && p.B <= 65535, CWFA_E_SHAPE,
                 "cwfa_conv2d_f32: grid too large");
    dim3 grid((unsigned)(p.tiles_x * p.tiles_y), ctiles, p.B);
    hipLaunchKernelGGL(conv2d_mfma_kernel<C>, grid, dim3(C::NTHREADS), C::LDS_BYTES, stream, p);
    CWFA_LAUNCH_CHECK("cwfa_conv2d_f32");
    return CWFA_OK;
}

}  // namespace

extern "C" int64_t cwfa_conv2d_packed_floats(int Cout, int Cin, int ks) {
    const Sel s = select_cfg(ks, Cout);
    if (s.id < 0 || Cout <= 0 || Cin <= 0) return -1;
    const int64_t ctiles = (Cout + s.CT - 1) / s.CT, nchunks = (Cin + s.CK - 1) / s.CK;
    return ctiles * nchunks * ks * ks * s.CK * s.CT;
}

extern "C" int cwfa_conv2d_pack_f32(const float* w, float* packed, int Cout, int Cin, int ks, int transposed, void* stream) {
    CWFA_REQUIRE(w && packed, CWFA_E_INVAL, "cwfa_conv2d_pack_f32: null pointer");
    CWFA_REQUIRE(!transposed || (ks == 1 && Cout % 4 == 0), CWFA_E_SHAPE,
                 "cwfa_conv2d_pack_f32: transposed source needs ks=1 (2x2 stride-2 deconv as 1x1) and Cout = 4*Co");
    const int64_t total = cwfa_conv2d_packed_floats(Cout, Cin, ks);
    CWFA_REQUIRE(total > 0, CWFA_E_SHAPE, "cwfa_conv2d_pack_f32: unsupported filter %dx%d, Cout=%d, Cin=%d", ks, ks, Cout, Cin);
    const Sel s = select_cfg(ks, Cout);
    const int nchunks = (Cin + s.CK - 1) / s.CK;
    hipLaunchKernelGGL(pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, packed, Cout,
                       Cin, ks, s.CT, s.CK, nchunks, total, transposed);
    CWFA_LAUNCH_CHECK("cwfa_conv2d_pack_f32");
    return CWFA_OK;
}

extern "C" int cwfa_conv2d_f32(const float* x, const float* w_packed, float* y, int B, int Cin, int H, int W, int Cout, int ks,
                               int64_t x_bs, int64_t y_bs, const cwfa_conv_opts* opts, void* stream) {
    CWFA_REQUIRE(x && w_packed && y, CWFA_E_INVAL, "cwfa_conv2d_f32: null pointer");
    CWFA_REQUIRE(B >= 0 && Cin > 0 && Cout > 0 && H >= 0 && W >= 0, CWFA_E_INVAL, "cwfa_conv2d_f32: bad size");
    CWFA_REQUIRE((int64_t)Cin * H * W < (1ll << 31), CWFA_E_SHAPE, "cwfa_conv2d_f32: Cin*H*W exceeds 32-bit tile offsets");
    const Sel s = select_cfg(ks, Cout);
    CWFA_REQUIRE(s.id >= 0, CWFA_E_SHAPE, "cwfa_conv2d_f32: kernel size %d not in {1,3,7}", ks);
    CWFA_REQUIRE(cwfa_aligned16(w_packed), CWFA_E_ALIGN, "cwfa_conv2d_f32: packed weights must be 16-byte aligned");
    if (B == 0 || H == 0 || W == 0) return CWFA_OK;
    ConvParams p{};
    p.x = x; p.wp = w_packed; p.y = y;
    p.B = B; p.Cin = Cin; p.H = H; p.W = W; p.Cout = Cout;
    p.x_bs = x_bs; p.y_bs = y_bs;
    if (opts) p.o = *opts;
    CWFA_REQUIRE(!p.o.upshuffle2 || (ks == 1 && Cout % 4 == 0), CWFA_E_SHAPE, "cwfa_conv2d_f32: upshuffle2 needs ks=1, Cout=4*Co");
    CWFA_REQUIRE(!(p.o.in_scale && !p.o.in_shift), CWFA_E_INVAL, "cwfa_conv2d_f32: in_scale without in_shift");
    CWFA_REQUIRE(p.o.act >= 0 && p.o.act <= CWFA_ACT_RELU && p.o.act2 >= 0 && p.o.act2 <= CWFA_ACT_RELU, CWFA_E_INVAL,
                 "cwfa_conv2d_f32: bad activation");
    hipStream_t st = (hipStream_t)stream;
    switch (s.id) {
        case 0: return launch<C3_32>(p, st);
        case 1: return launch<C3_64>(p, st);
        case 2: return launch<C3_128>(p, st);
        case 3: return launch<C1_32>(p, st);
        case 4: return launch<C1_64>(p, st);
        case 5: return launch<C1_128>(p, st);
        case 6: return launch<C7_32>(p, st);
        default: return launch<C7_64>(p, st);
    }
}
