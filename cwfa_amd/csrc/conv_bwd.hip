// Backward kernels of the coupling sub-networks (the training step of CWFA.py:966-1027 on the
// flow steps): the weight gradient of a stride-1 "same" convolution on the fp32 matrix cores, and the ELU backward.
// The data gradient of a convolution is a convolution with the transposed, spatially flipped filter bank and runs on
// the forward kernels (conv2d.hip / conv_wino.hip); the bias gradient is cwfa_channel_stats_f32's sum.
//
// Weight gradient:  dW[co][ci][ky][kx] = sum_{b,y,x} dy[b][co][y][x] * x[b][ci][y+ky-p][x+kx-p]        (torch autograd of
// nn.Conv2d, networks.py:621-638) -- a GEMM with M = Cout, N = Cin*taps and K = B*H*W.  A block owns 64 couts x 64 cins
// x all taps and walks strips of 2 rows x 32 pixels (K = 64 per strip).  Both operands sit channel-minor in LDS
// ([pixel][channel], row stride 65 words), so the 32 lanes of an MFMA operand read consecutive words: the A operand is
// dy[co][pixel], the B operand of tap (ky,kx) is the SAME x tile read at a shifted pixel -- no im2col, a tap costs an
// LDS address immediate.  Wave (mt, cit) of the four holds the 32 co x 32 ci x taps accumulators (144 registers for 3x3).
// Strips are double-buffered: the next strip's global loads are issued before the 32 k-steps of the current one and
// stored to the other buffer after them.  Every pixel worker writes its partial filter bank; a second kernel sums the
// partials in a fixed order, so the result is deterministic (no float atomics).
#include <type_traits>
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int N, class F, int I = 0>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<N, F, I + 1>(static_cast<F&&>(f));
    }
}
__device__ __forceinline__ int acc_row(int r, int kh) { return (r & 3) + 8 * (r >> 2) + 4 * kh; }

struct WgParams {
    const float* x;
    const float* dy;
    float* part;
    float* bpart;      // nullable: per-worker partial bias gradients [worker][Cout]
    int B, Cin, H, W, Cout;
    int64_t x_bs, dy_bs;
    int sx, sy;        // strips per row / per column of strips
    int nstrips;       // B * sy * sx
};

constexpr int WG_CHP = 65;       // LDS words per pixel (64 channels + 1: conflict-free transposing stores)

template <int KS>
struct WgCfg {
    static constexpr int TAPS = KS * KS, PADK = KS / 2;
    static constexpr int XR = 2 + 2 * PADK, XC = 32 + 2 * PADK;
    static constexpr int XW = XR * XC * WG_CHP, DW = 64 * WG_CHP;      // words of the x tile / the dy tile
    static constexpr int BUFW = XW + DW;
    static constexpr int LDS_BYTES = 2 * BUFW * 4;
    static constexpr int XLOADS = 64 * XR / 8;                        // core columns: (ci,row) pairs / 8 per pass
    static constexpr int HLOADS = PADK ? 64 * XR * 2 / 256 : 0;         // halo columns
};

template <int KS>
__global__ __launch_bounds__(256, 1) void conv_wgrad_kernel(WgParams p) {
    typedef WgCfg<KS> C;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, kh = lane >> 5;
    const int mt = wave & 1, cit = wave >> 1;
    const int co0 = blockIdx.y * 64, ci0 = blockIdx.z * 64;
    const int64_t HW = (int64_t)p.H * p.W;
    const unsigned HW4 = (unsigned)HW * 4u;
    constexpr unsigned OOB = 0x80000000u;

    // staging roles (see header): core columns -- col = tid & 31, pair = (tid >> 5) + 8 j
    const int scol = tid & 31, sq = tid >> 5;
    float xr[C::XLOADS], hr[C::HLOADS ? C::HLOADS : 1], dr[16];
    float bacc = 0.f;        // running sum of this lane's A operands = sum over pixels of dy[co]: the bias gradient

    auto issue = [&](int strip) {
        const int b = strip / (p.sy * p.sx), rem = strip - b * (p.sy * p.sx);
        const int r0 = (rem / p.sx) * 2, c0 = (rem % p.sx) * 32;
        const auto rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x + (int64_t)b * p.x_bs + (int64_t)ci0 * HW), 0,
                                                          (int)((unsigned)min(64, p.Cin - ci0) * HW4), 0x00020000);
        const auto rd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dy + (int64_t)b * p.dy_bs + (int64_t)co0 * HW), 0,
                                                          (int)((unsigned)min(64, p.Cout - co0) * HW4), 0x00020000);
        {   // x, core columns: pair q = sq + 8 j -> ci = q / XR, row = q % XR
            const int gc = c0 + scol;
#pragma unroll
            for (int j = 0; j < C::XLOADS; ++j) {
                const int q = sq + 8 * j, ci = q / C::XR, r = q % C::XR;
                const int gr = r0 + r - C::PADK;
                const unsigned off = (gr >= 0 && gr < p.H && gc < p.W) ? (unsigned)ci * HW4 + (unsigned)(gr * p.W + gc) * 4u : OOB;
                xr[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, off, 0, 0));
            }
        }
        if constexpr (C::HLOADS > 0) {      // x, the two halo columns: pair q = tid (ci = q / XR, row = q % XR), side j
            const int ci = tid / C::XR, r = tid % C::XR;
            const int gr = r0 + r - C::PADK;
#pragma unroll
            for (int j = 0; j < C::HLOADS; ++j) {
                const int gc = c0 - 1 + 33 * j;
                const unsigned off = (gr >= 0 && gr < p.H && gc >= 0 && gc < p.W) ? (unsigned)ci * HW4 + (unsigned)(gr * p.W + gc) * 4u : OOB;
                hr[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, off, 0, 0));
            }
        }
        {   // dy: pair q = sq + 8 j -> co = q / 2, row = q % 2
            const int gc = c0 + scol;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int q = sq + 8 * j, co = q >> 1, r = q & 1;
                const int gr = r0 + r;
                const unsigned off = (gr < p.H && gc < p.W) ? (unsigned)co * HW4 + (unsigned)(gr * p.W + gc) * 4u : OOB;
                dr[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rd, off, 0, 0));
            }
        }
    };
    auto stash = [&](int buf) {
        float* xs = smem + buf * C::BUFW;
        float* ds = xs + C::XW;
#pragma unroll
        for (int j = 0; j < C::XLOADS; ++j) {
            const int q = sq + 8 * j, ci = q / C::XR, r = q % C::XR;
            xs[(r * C::XC + scol + C::PADK) * WG_CHP + ci] = xr[j];
        }
        if constexpr (C::HLOADS > 0) {
            const int ci = tid / C::XR, r = tid % C::XR;
#pragma unroll
            for (int j = 0; j < C::HLOADS; ++j) xs[(r * C::XC + 33 * j) * WG_CHP + ci] = hr[j];
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int q = sq + 8 * j, co = q >> 1, r = q & 1;
            ds[(r * 32 + scol) * WG_CHP + co] = dr[j];
        }
    };

    f32x16 acc[C::TAPS];
#pragma unroll
    for (int i = 0; i < C::TAPS; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    int strip = blockIdx.x;
    int buf = 0;
    if (strip < p.nstrips) {
        issue(strip);
        stash(0);
    }
    __syncthreads();
    for (; strip < p.nstrips; strip += gridDim.x) {
        const int next = strip + gridDim.x;
        const bool more = next < p.nstrips;
        if (more) issue(next);
        const float* xs = smem + buf * C::BUFW + kh * WG_CHP + cit * 32 + l31;
        const float* ds = smem + buf * C::BUFW + C::XW + kh * WG_CHP + mt * 32 + l31;
        // operand registers one k-step ahead of the MFMAs that consume them (two sets), so that the LDS latency runs under
        // the nine MFMAs of the step before (one wave per SIMD: nobody else would hide it)
        float av[2], bv[2][C::TAPS];
        auto fetch = [&](auto sc) {
            constexpr int s = decltype(sc)::value;
            constexpr int row = s / 16, col = (2 * s) % 32;
            av[s & 1] = ds[(2 * s) * WG_CHP];
#pragma unroll
            for (int i = 0; i < C::TAPS; ++i) bv[s & 1][i] = xs[((row + i / KS) * C::XC + col + i % KS) * WG_CHP];
        };
        fetch(std::integral_constant<int, 0>{});
        static_for<32>([&](auto sc) {
            constexpr int s = decltype(sc)::value;
            if constexpr (s + 1 < 32) fetch(std::integral_constant<int, s + 1>{});
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < C::TAPS; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s & 1], bv[s & 1][i], acc[i], 0, 0, 0);
            bacc += av[s & 1];
            __builtin_amdgcn_sched_barrier(0);
        });
        if (more) stash(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }

    // bias gradient: lane (l31, kh) of an m-tile's wave has summed dy[co0 + mt*32 + l31] over the pixels of parity kh
    {
        const float tot = bacc + __shfl_xor(bacc, 32, 64);
        const int co = co0 + mt * 32 + l31;
        if (p.bpart && blockIdx.z == 0 && cit == 0 && kh == 0 && co < p.Cout) p.bpart[(int64_t)blockIdx.x * p.Cout + co] = tot;
    }

    // partial filter bank of this pixel worker: part[worker][co][ci][tap]
    const int ci = ci0 + cit * 32 + l31;
    float* out = p.part + (int64_t)blockIdx.x * p.Cout * p.Cin * C::TAPS;
    if (ci < p.Cin) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + mt * 32 + acc_row(r, kh);
            if (co < p.Cout) {
#pragma unroll
                for (int i = 0; i < C::TAPS; ++i) out[((int64_t)co * p.Cin + ci) * C::TAPS + i] = acc[i][r];
            }
        }
    }
}

// Sum of the workers' partial filter banks in a fixed order (deterministic).  A block owns 64 consecutive filter taps; its
// four waves each take a contiguous quarter of the workers with four independent running sums (loads in flight), and
// the 16 sums per tap are combined in a fixed order.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, int64_t n,
                                                           int workers, float beta) {
    __shared__ float red[4][64];
    const int li = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * 64 + li;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (i < n) {
        const int per = (workers + 3) / 4, w0 = grp * per, w1 = min(workers, w0 + per);
        const float* src = part + i;
        int w = w0;
        for (; w + 4 <= w1; w += 4) {
            a0 += src[(int64_t)w * n];
            a1 += src[(int64_t)(w + 1) * n];
            a2 += src[(int64_t)(w + 2) * n];
            a3 += src[(int64_t)(w + 3) * n];
        }
        for (; w < w1; ++w) a0 += src[(int64_t)w * n];
    }
    red[grp][li] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (grp == 0 && i < n) {
        const float s = (red[0][li] + red[1][li]) + (red[2][li] + red[3][li]);
        dw[i] = beta != 0.f ? beta * dw[i] + s : s;
    }
}

// ELU backward from the layer's OUTPUT a = ELU(q):  dq = g * (a > 0 ? 1 : a + 1)   (exp(q) = a + 1 for q <= 0), + add
__global__ __launch_bounds__(256) void elu_bwd_kernel(const float* __restrict__ g, const float* __restrict__ a,
                                                      const float* __restrict__ add, float* __restrict__ y, int64_t n,
                                                      int64_t g_bs, int64_t a_bs, int64_t add_bs, int64_t y_bs) {
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= n) return;
    const int b = blockIdx.y;
    const float4 gv = *reinterpret_cast<const float4*>(g + b * g_bs + i);
    const float4 av = *reinterpret_cast<const float4*>(a + b * a_bs + i);
    float4 o;
    o.x = gv.x * (av.x > 0.f ? 1.f : av.x + 1.f);
    o.y = gv.y * (av.y > 0.f ? 1.f : av.y + 1.f);
    o.z = gv.z * (av.z > 0.f ? 1.f : av.z + 1.f);
    o.w = gv.w * (av.w > 0.f ? 1.f : av.w + 1.f);
    if (add) {
        const float4 dv = *reinterpret_cast<const float4*>(add + b * add_bs + i);
        o.x += dv.x; o.y += dv.y; o.z += dv.z; o.w += dv.w;
    }
    *reinterpret_cast<float4*>(y + b * y_bs + i) = o;
}

int wgrad_workers(int B, int H, int W, int Cout, int Cin, int ks) {
    const int64_t strips = (int64_t)B * ((H + 1) / 2) * ((W + 31) / 32);
    const int tiles = ((Cout + 63) / 64) * ((Cin + 63) / 64);
    // 3x3: one block per CU in total (104 KB of LDS each); 1x1: 65 KB each and little arithmetic per byte, so two
    // resident blocks per CU to keep more loads in flight
    int64_t w = ((ks == 1 ? 512 : 256) + tiles - 1) / tiles;
    if (w > strips) w = strips;
    return (int)(w < 1 ? 1 : w);
}

}  // namespace

extern "C" int64_t cwfa_conv2d_wgrad_workspace_bytes(int B, int Cin, int H, int W, int Cout, int ks) {
    if (B <= 0 || Cin <= 0 || H <= 0 || W <= 0 || Cout <= 0 || (ks != 1 && ks != 3)) return 0;
    return (int64_t)wgrad_workers(B, H, W, Cout, Cin, ks) * ((int64_t)Cout * Cin * ks * ks + Cout) * 4;
}

extern "C" int cwfa_conv2d_wgrad_f32(const float* x, const float* dy, float* dw, float* db, void* workspace, int B, int Cin, int H,
                                     int W, int Cout, int ks, int64_t x_bs, int64_t dy_bs, float beta, void* stream) {
    CWFA_REQUIRE(x && dy && dw && workspace, CWFA_E_INVAL, "cwfa_conv2d_wgrad_f32: null pointer");
    CWFA_REQUIRE(ks == 1 || ks == 3, CWFA_E_SHAPE, "cwfa_conv2d_wgrad_f32: kernel size %d (1 and 3 are built)", ks);
    CWFA_REQUIRE(B >= 0 && Cin > 0 && Cout > 0 && H >= 0 && W >= 0, CWFA_E_SHAPE, "cwfa_conv2d_wgrad_f32: bad shape");
    CWFA_REQUIRE((int64_t)64 * H * W * 4 < (1ll << 31), CWFA_E_SHAPE, "cwfa_conv2d_wgrad_f32: image too large for 32-bit offsets");
    const int64_t n = (int64_t)Cout * Cin * ks * ks;
    hipStream_t st = (hipStream_t)stream;
    if (B == 0 || H == 0 || W == 0) {
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((n + 63) / 64)), dim3(256), 0, st,
                           reinterpret_cast<const float*>(workspace), dw, n, 0, beta);
        CWFA_LAUNCH_CHECK("cwfa_conv2d_wgrad_f32");
        if (db) {
            hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((Cout + 63) / 64)), dim3(256), 0, st,
                               reinterpret_cast<const float*>(workspace), db, (int64_t)Cout, 0, beta);
            CWFA_LAUNCH_CHECK("cwfa_conv2d_wgrad_f32");
        }
        return CWFA_OK;
    }
    WgParams p{};
    p.x = x; p.dy = dy; p.part = reinterpret_cast<float*>(workspace);
    p.B = B; p.Cin = Cin; p.H = H; p.W = W; p.Cout = Cout;
    p.x_bs = x_bs; p.dy_bs = dy_bs;
    p.sx = (W + 31) / 32; p.sy = (H + 1) / 2;
    const int64_t strips = (int64_t)B * p.sy * p.sx;
    CWFA_REQUIRE(strips < (1ll << 31), CWFA_E_SHAPE, "cwfa_conv2d_wgrad_f32: too many strips");
    p.nstrips = (int)strips;
    const int workers = wgrad_workers(B, H, W, Cout, Cin, ks);
    p.bpart = db ? p.part + (int64_t)workers * n : nullptr;
    dim3 grid(workers, (Cout + 63) / 64, (Cin + 63) / 64);
    static bool attr1 = false, attr3 = false;
    if (ks == 3) {
        if (!attr3) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_kernel<3>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, WgCfg<3>::LDS_BYTES);
            CWFA_REQUIRE(e == hipSuccess, CWFA_E_HIP, "cwfa_conv2d_wgrad_f32: hipFuncSetAttribute: %s", hipGetErrorString(e));
            attr3 = true;
        }
        hipLaunchKernelGGL(conv_wgrad_kernel<3>, grid, dim3(256), WgCfg<3>::LDS_BYTES, st, p);
    } else {
        if (!attr1) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_kernel<1>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, WgCfg<1>::LDS_BYTES);
            CWFA_REQUIRE(e == hipSuccess, CWFA_E_HIP, "cwfa_conv2d_wgrad_f32: hipFuncSetAttribute: %s", hipGetErrorString(e));
            attr1 = true;
        }
        hipLaunchKernelGGL(conv_wgrad_kernel<1>, grid, dim3(256), WgCfg<1>::LDS_BYTES, st, p);
    }
    CWFA_LAUNCH_CHECK("cwfa_conv2d_wgrad_f32");
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((n + 63) / 64)), dim3(256), 0, st, p.part, dw, n, workers, beta);
    CWFA_LAUNCH_CHECK("cwfa_conv2d_wgrad_f32");
    if (db) {
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((Cout + 63) / 64)), dim3(256), 0, st, p.bpart, db, (int64_t)Cout, workers,
                           beta);
        CWFA_LAUNCH_CHECK("cwfa_conv2d_wgrad_f32");
    }
    return CWFA_OK;
}

extern "C" int cwfa_elu_bwd_f32(const float* g, const float* a, const float* add, float* y, int B, int64_t n, int64_t g_bs,
                                int64_t a_bs, int64_t add_bs, int64_t y_bs, void* stream) {
    CWFA_REQUIRE(g && a && y, CWFA_E_INVAL, "cwfa_elu_bwd_f32: null pointer");
    CWFA_REQUIRE(B >= 0 && n >= 0 && B <= 65535, CWFA_E_SHAPE, "cwfa_elu_bwd_f32: bad shape");
    CWFA_REQUIRE(n % 4 == 0 && g_bs % 4 == 0 && a_bs % 4 == 0 && y_bs % 4 == 0 && (!add || add_bs % 4 == 0), CWFA_E_SHAPE,
                 "cwfa_elu_bwd_f32: sizes and strides must be multiples of 4 elements");
    CWFA_REQUIRE(cwfa_aligned16(g) && cwfa_aligned16(a) && cwfa_aligned16(y) && (!add || cwfa_aligned16(add)), CWFA_E_ALIGN,
                 "cwfa_elu_bwd_f32: pointers must be 16-byte aligned");
    if (B == 0 || n == 0) return CWFA_OK;
    dim3 grid((unsigned)((n / 4 + 255) / 256), B);
    hipLaunchKernelGGL(elu_bwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, g, a, add, y, n, g_bs, a_bs, add_bs, y_bs);
    CWFA_LAUNCH_CHECK("cwfa_elu_bwd_f32");
    return CWFA_OK;
}
