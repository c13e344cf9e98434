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
typedef float f32x4 __attribute__((ext_vector_type(4)));
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
    int row_off;       // added to the input row (the 7x7 gradient runs as seven 1x7 passes over row-shifted inputs)
};

constexpr int WG_CHP = 65;       // LDS words per pixel (64 channels + 1: conflict-free transposing stores)

template <int KH, int KW>
struct WgCfg {
    static constexpr int TAPS = KH * KW, PADH = KH / 2, PADW = KW / 2;
    static constexpr int XR = 2 + 2 * PADH, XC = 32 + 2 * PADW;
    static constexpr int XW = XR * XC * WG_CHP, DW = 64 * WG_CHP;      // words of the x tile / the dy tile
    static constexpr int BUFW = XW + DW;
    static constexpr int LDS_BYTES = 2 * BUFW * 4;
    static constexpr int XLOADS = 64 * XR / 8;                        // core columns: (ci,row) pairs / 8 per pass
    static_assert(8 % XR == 0, "the staging map steps whole channels per pass");
    static constexpr int HLOADS = 64 * XR * 2 * PADW / 256;             // halo columns (2 * PADW of them)
};

template <int KH, int KW>
__global__ __launch_bounds__(256, 1) void conv_wgrad_kernel(WgParams p) {
    typedef WgCfg<KH, KW> C;
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
        {   // x, core columns: pair q = sq + 8 j -> ci = q / XR, row = q % XR.  8 is a multiple of XR, so the row (and with it
            // the validity) is the same for every j and the channel advances by 8 / XR: one compare, then one add per load
            // (the out-of-range marker survives the adds: 64 * HW4 < 2^31)
            const int gc = c0 + scol, r = sq % C::XR;
            const int gr = r0 + r - C::PADH + p.row_off;
            unsigned off = (gr >= 0 && gr < p.H && gc < p.W) ? (unsigned)(sq / C::XR) * HW4 + (unsigned)(gr * p.W + gc) * 4u : OOB;
#pragma unroll
            for (int j = 0; j < C::XLOADS; ++j) {
                xr[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, off, 0, 0));
                off += (unsigned)(8 / C::XR) * HW4;
            }
        }
        if constexpr (C::HLOADS > 0) {      // x, the 2 * PADW halo columns: entry e = tid + 256 j -> (pair = (ci, row), halo column)
#pragma unroll
            for (int j = 0; j < C::HLOADS; ++j) {
                const int e = tid + 256 * j, hc = e % (2 * C::PADW), pair = e / (2 * C::PADW);
                const int ci = pair / C::XR, r = pair % C::XR;
                const int gr = r0 + r - C::PADH + p.row_off;
                const int tc = hc < C::PADW ? hc : 32 + hc;                 // tile column
                const int gc = c0 - C::PADW + tc;
                const unsigned off = (gr >= 0 && gr < p.H && gc >= 0 && gc < p.W) ? (unsigned)ci * HW4 + (unsigned)(gr * p.W + gc) * 4u : OOB;
                hr[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, off, 0, 0));
            }
        }
        {   // dy: pair q = sq + 8 j -> co = q / 2, row = q % 2: same structure, the channel advances by 4
            const int gc = c0 + scol, gr = r0 + (sq & 1);
            unsigned off = (gr < p.H && gc < p.W) ? (unsigned)(sq >> 1) * HW4 + (unsigned)(gr * p.W + gc) * 4u : OOB;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                dr[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rd, off, 0, 0));
                off += 4u * HW4;
            }
        }
    };
    auto stash = [&](int buf) {
        float* xs = smem + buf * C::BUFW;
        float* ds = xs + C::XW;
#pragma unroll
        for (int j = 0; j < C::XLOADS; ++j) {
            const int q = sq + 8 * j, ci = q / C::XR, r = q % C::XR;
            xs[(r * C::XC + scol + C::PADW) * WG_CHP + ci] = xr[j];
        }
        if constexpr (C::HLOADS > 0) {
#pragma unroll
            for (int j = 0; j < C::HLOADS; ++j) {
                const int e = tid + 256 * j, hc = e % (2 * C::PADW), pair = e / (2 * C::PADW);
                const int ci = pair / C::XR, r = pair % C::XR;
                xs[(r * C::XC + (hc < C::PADW ? hc : 32 + hc)) * WG_CHP + ci] = hr[j];
            }
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
            for (int i = 0; i < C::TAPS; ++i) bv[s & 1][i] = xs[((row + i / KW) * C::XC + col + i % KW) * WG_CHP];
        };
        fetch(std::integral_constant<int, 0>{});
        static_for<32>([&](auto sc) {
            constexpr int s = decltype(sc)::value;
            // one operand read of step s + 1 after every MFMA of step s: a burst of reads between two steps would hold the
            // issue port longer than the last MFMA keeps the matrix pipe busy
            constexpr int s1 = s + 1, row1 = s1 / 16, col1 = (2 * s1) % 32;
#pragma unroll
            for (int i = 0; i < C::TAPS; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s & 1], bv[s & 1][i], acc[i], 0, 0, 0);
                if constexpr (s1 < 32) {
                    bv[s1 & 1][i] = xs[((row1 + i / KW) * C::XC + col1 + i % KW) * WG_CHP];
                    if (i == 0) av[s1 & 1] = ds[(2 * s1) * WG_CHP];
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            bacc += av[s & 1];
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

// The filter bank's and the bias' partials in ONE launch: blocks [0, nb) reduce `part`, the rest `bpart` (one launch less per
// weight gradient: ~200 per training iteration)
__global__ __launch_bounds__(256) void wgrad_reduce2_kernel(const float* __restrict__ part, float* __restrict__ dw, int64_t n,
                                                            const float* __restrict__ bpart, float* __restrict__ db, int64_t nb_,
                                                            int nblocks_w, int workers, float beta) {
    __shared__ float red[4][64];
    const bool second = (int)blockIdx.x >= nblocks_w;
    const float* src0 = second ? bpart : part;
    float* dst = second ? db : dw;
    const int64_t len = second ? nb_ : n;
    const int li = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int64_t i = (int64_t)(second ? blockIdx.x - nblocks_w : blockIdx.x) * 64 + li;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (i < len) {
        const int per = (workers + 3) / 4, w0 = grp * per, w1 = min(workers, w0 + per);
        const float* src = src0 + i;
        int w = w0;
        for (; w + 4 <= w1; w += 4) {
            a0 += src[(int64_t)w * len];
            a1 += src[(int64_t)(w + 1) * len];
            a2 += src[(int64_t)(w + 2) * len];
            a3 += src[(int64_t)(w + 3) * len];
        }
        for (; w < w1; ++w) a0 += src[(int64_t)w * len];
    }
    red[grp][li] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (grp == 0 && i < len) {
        const float s_ = (red[0][li] + red[1][li]) + (red[2][li] + red[3][li]);
        dst[i] = beta != 0.f ? beta * dst[i] + s_ : s_;
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

// ---- 3x3 weight gradient, second form (images whose rows are 16-byte aligned: W % 4 == 0): pixel-minor tiles filled by
// LDS-DMA.  Same block / wave / accumulator layout as conv_wgrad_kernel<3,3>, but
//   * the x tile is [64 ci][4 rows][40 px] (image columns c0-4 .. c0+35) + one pad chunk per channel (164 words: 41
//     sixteen-byte chunks, odd, so the 32 lanes of a ds_read_b128 hit 32 different 16-byte bank groups), the dy tile
//     [64 co][2 rows][32 px] + pad (68 words); both arrive by `buffer_load ... lds` (16 bytes per lane, zeros outside the
//     image from the range check): 14 or 15 DMA instructions per wave and strip, no staging registers, no ds_write;
//   * the MFMA k index pairs the strip's two ROWS (kh) and a k-step is one column, so a lane's operands of consecutive
//     k-steps are consecutive words: one ds_read_b128 of dy and three of x (one per tap row; the three taps of a row slide
//     over a 12-pixel register window) feed FOUR k-steps = 36 MFMAs, where the first form issues 28 LDS reads.
namespace wr {
constexpr int XCH = 10, XCI = 4 * XCH + 1, XCHUNKS = 64 * XCI;      // chunks (16 B) per tile row / per channel / per tile
constexpr int DCO = 2 * 8 + 1, DCHUNKS = 64 * DCO;
constexpr int XINSTR = XCHUNKS / 64, DINSTR = DCHUNKS / 64;         // 41 + 17 wave-wide DMA instructions per strip
constexpr int BUFB = (XCHUNKS + DCHUNKS) * 16, LDS_BYTES = 2 * BUFB;
constexpr int XSLOTS = (XINSTR + 3) / 4, DSLOTS = (DINSTR + 3) / 4; // per wave
static_assert(XCHUNKS % 64 == 0 && DCHUNKS % 64 == 0, "whole wave instructions");
}  // namespace wr

__global__ __launch_bounds__(256, 1) void conv_wgrad_rows_kernel(WgParams p) {
    using namespace wr;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* lds = reinterpret_cast<char*>(smem);
    typedef __attribute__((address_space(3))) void* lds_ptr;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, kh = lane >> 5;
    const int mt = wave & 1, cit = wave >> 1;
    const int co0 = blockIdx.y * 64, ci0 = blockIdx.z * 64;
    const int64_t HW = (int64_t)p.H * p.W;
    const unsigned HW4 = (unsigned)HW * 4u;
    constexpr unsigned OOB = 0x80000000u;

    // DMA roles: slot j of this wave is wave instruction q = wave + 4 j; lane -> chunk g = 64 q + lane of the tile
    unsigned xrel[XSLOTS], drel[DSLOTS];       // byte offset relative to the strip origin (row r0, column c0)
    int xrc[XSLOTS], drc[DSLOTS];              // row | column chunk << 4 | (a real chunk, not a pad) << 8
#pragma unroll
    for (int j = 0; j < XSLOTS; ++j) {
        const int g = 64 * (wave + 4 * j) + lane, ci = g / XCI, rem = g % XCI, row = rem / XCH, cc = rem % XCH;
        const bool real = wave + 4 * j < XINSTR && rem < 4 * XCH;
        xrel[j] = (unsigned)ci * HW4 + (unsigned)(((row - 1) * p.W + 4 * cc - 4) * 4);
        xrc[j] = row | (cc << 4) | ((int)real << 8);
    }
#pragma unroll
    for (int j = 0; j < DSLOTS; ++j) {
        const int g = 64 * (wave + 4 * j) + lane, co = g / DCO, rem = g % DCO, row = rem / 8, cc = rem % 8;
        const bool real = wave + 4 * j < DINSTR && rem < 16;
        drel[j] = (unsigned)co * HW4 + (unsigned)((row * p.W + 4 * cc) * 4);
        drc[j] = row | (cc << 4) | ((int)real << 8);
    }
    auto issue = [&](int strip, int buf) {
        const int b = strip / (p.sy * p.sx), rem = strip - b * (p.sy * p.sx);
        const int r0 = (rem / p.sx) * 2, c0 = (rem % p.sx) * 32;
        const auto rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x + (int64_t)b * p.x_bs + (int64_t)ci0 * HW), 0,
                                                          (int)((unsigned)min(64, p.Cin - ci0) * HW4), 0x00020000);
        const auto rd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dy + (int64_t)b * p.dy_bs + (int64_t)co0 * HW), 0,
                                                          (int)((unsigned)min(64, p.Cout - co0) * HW4), 0x00020000);
        const unsigned origin = (unsigned)((r0 * p.W + c0) * 4);
        // tile rows r0-1 .. r0+2 and column chunks c0-4+4cc: which of them lie inside the image
        const int rmin = r0 == 0 ? 1 : 0, rmax = min(3, p.H - r0), cmin = c0 == 0 ? 1 : 0, cmax = min(XCH - 1, (p.W - c0) / 4);
        char* xb = lds + buf * BUFB;
#pragma unroll
        for (int j = 0; j < XSLOTS; ++j) {
            if (wave + 4 * j < XINSTR) {
                const int row = xrc[j] & 15, cc = (xrc[j] >> 4) & 15;
                const bool ok = (xrc[j] >> 8) && row >= rmin && row <= rmax && cc >= cmin && cc <= cmax;
                const unsigned off = ok ? origin + xrel[j] : OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr)(xb + (wave + 4 * j) * 1024), 16, off, 0, 0, 0);
            }
        }
        const int drmax = min(1, p.H - 1 - r0), dcmax = min(7, (p.W - c0) / 4 - 1);
        char* db = xb + XCHUNKS * 16;
#pragma unroll
        for (int j = 0; j < DSLOTS; ++j) {
            if (wave + 4 * j < DINSTR) {
                const int row = drc[j] & 15, cc = (drc[j] >> 4) & 15;
                const bool ok = (drc[j] >> 8) && row <= drmax && cc <= dcmax;
                const unsigned off = ok ? origin + drel[j] : OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rd, (lds_ptr)(db + (wave + 4 * j) * 1024), 16, off, 0, 0, 0);
            }
        }
    };

    f32x16 acc[9];
#pragma unroll
    for (int i = 0; i < 9; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float bacc = 0.f;

    int strip = blockIdx.x, buf = 0;
    if (strip < p.nstrips) issue(strip, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (; strip < p.nstrips; strip += gridDim.x) {
        const int next = strip + gridDim.x;
        if (next < p.nstrips) issue(next, buf ^ 1);
        // this lane's operand rows: A = dy[co][row kh], B(ky) = x[ci][tile row kh + ky]
        const f32x4* ap = reinterpret_cast<const f32x4*>(lds + buf * BUFB + XCHUNKS * 16) + (mt * 32 + l31) * DCO + kh * 8;
        const f32x4* bp = reinterpret_cast<const f32x4*>(lds + buf * BUFB) + (cit * 32 + l31) * XCI + kh * XCH;
        f32x4 win[3][3], av, nwin[3], nav;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int c = 0; c < 3; ++c) win[ky][c] = bp[ky * XCH + c];
        av = ap[0];
        static_for<8>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                // output column 4j + e sits at tile column t = 4(j+1) + e; taps read t-1, t, t+1 of the three tile rows
                const float a = av[e];
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    const float b0 = e == 0 ? win[ky][0][3] : win[ky][1][e - 1];
                    const float b1 = win[ky][1][e];
                    const float b2 = e == 3 ? win[ky][2][0] : win[ky][1][e + 1];
                    acc[ky * 3 + 0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, acc[ky * 3 + 0], 0, 0, 0);
                    acc[ky * 3 + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, acc[ky * 3 + 1], 0, 0, 0);
                    acc[ky * 3 + 2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b2, acc[ky * 3 + 2], 0, 0, 0);
                    if constexpr (j < 7) {          // one read of the next group after every third of a k-step's MFMAs
                        if (e == ky) nwin[ky] = bp[ky * XCH + j + 3];
                        if (e == 3 && ky == 0) nav = ap[j + 1];
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                bacc += a;
            }
            if constexpr (j < 7) {
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    win[ky][0] = win[ky][1];
                    win[ky][1] = win[ky][2];
                    win[ky][2] = nwin[ky];
                }
                av = nav;
            }
        });
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the next strip's tiles have landed (this wave's part)
        __syncthreads();
        buf ^= 1;
    }

    {
        const float tot = bacc + __shfl_xor(bacc, 32, 64);
        const int co = co0 + mt * 32 + l31;
        if (p.bpart && blockIdx.z == 0 && cit == 0 && kh == 0 && co < p.Cout) p.bpart[(int64_t)blockIdx.x * p.Cout + co] = tot;
    }
    const int ci = ci0 + cit * 32 + l31;
    float* out = p.part + (int64_t)blockIdx.x * p.Cout * p.Cin * 9;
    if (ci < p.Cin) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + mt * 32 + acc_row(r, kh);
            if (co < p.Cout) {
#pragma unroll
                for (int i = 0; i < 9; ++i) out[((int64_t)co * p.Cin + ci) * 9 + i] = acc[i][r];
            }
        }
    }
}


// ------------------------------------------------------------------------------------------------------------------------------
// 3x3 weight gradient on the bf16 matrix cores in the fp32-equivalent split arithmetic of the forward kernels (every fp32 operand =
// three bf16 pieces exactly, six partial products, fp32 accumulation; SIX = false: plain bf16 operands).  Training in split /
// bf16 precision.
//   GEMM: M = 64 couts (A = dy), N = 64 cins x 9 taps (B = x), K = pixels -- v_mfma_f32_16x16x32_bf16 takes 32 PIXELS OF ONE IMAGE
//   ROW per instruction; lane group g holds pixels 8g .. 8g+7 of both operands, i.e. one aligned 16-byte LDS read each.
//   A tap's kx shift would misalign the x read by 2 bytes (ds_read_b128 at a 2-byte offset: 11x slower, tools/probe/lds_unaligned),
//   so the shift is put on dy instead and paid at staging time: dy sits in LDS in THREE copies, shifted by +1 / 0 / -1 pixel
//   (a staging thread holds its eight pixels and the two neighbours and packs three windows of the ten); the ky shift is a row
//   of the x ring.  Zero padding = the range check of the buffer loads (x) / explicit out-of-range offsets (dy beyond the row).
//   A block (8 waves) owns 64 couts x 64 cins and walks UNITS = (sample, 32-pixel column, segment of SEG rows) top to bottom,
//   one dy row per step: x rows live in a four-slot LDS ring (rows y-1, y, y+1 in use, y+2 being written), dy rows in two
//   buffers; wave (mt = wave & 3, nh = wave >> 2) holds the 16 co x 32 ci x 9 taps accumulators (72 registers).  Per step and
//   wave: 9 + 18 ds_read_b128, 108 MFMAs, one barrier.  Waves 0..3 stage dy (8 + 2 pixels per thread, the bias gradient is the
//   running sum of what they load), waves 4..7 stage x (8 pixels per thread); the global loads of step y + 1 are issued before
//   the MFMAs of step y and split / stored behind them.
//   Partials per worker + wgrad_reduce_kernel in a fixed order, as for the fp32 forms: deterministic.
namespace ws {
constexpr int SEG = 32;                       // rows per unit
constexpr int DYROW = 64 + 16;                // bytes per (copy, piece, co): 32 bf16 + pad (5 x 16: conflict-free b128 reads)
constexpr int DYPLANE = 64 * DYROW, DYBUF = 9 * DYPLANE;
constexpr int XROW = 4 * 64 + 16;             // bytes per (piece, ci): four ring rows of 32 bf16 + pad (17 x 16)
constexpr int XPLANE = 64 * XROW, XBUF = 3 * XPLANE;
constexpr int LDS_BYTES = 2 * DYBUF + XBUF;   // 144 384
}  // namespace ws

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <bool SIX>
__device__ __forceinline__ void wsplit3(float v, __bf16& a1, __bf16& a2, __bf16& a3) {
    a1 = (__bf16)v;
    a2 = a3 = (__bf16)0.f;
    if constexpr (SIX) {
        const float r1 = v - (float)a1;
        a2 = (__bf16)r1;
        a3 = (__bf16)(r1 - (float)a2);
    }
}

// KS = 3, or 1: the 1x1 weight gradient as the same walk with one tap (one dy copy, one x row in use)
template <bool SIX, int KS = 3>
__global__ __launch_bounds__(512, 1) void conv_wgrad_split_kernel(WgParams p) {
    using namespace ws;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* lds = reinterpret_cast<char*>(smem);
    const int tid = threadIdx.x, lane = tid & 63, c16 = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mt = wave & 3, nh = wave >> 2;
    const int co0 = blockIdx.y * 64, ci0 = blockIdx.z * 64;
    const int HW = p.H * p.W;
    constexpr unsigned OOB = 0x80000000u;
    constexpr int NQ = SIX ? 3 : 1, PAD = KS / 2, TAPS = KS * KS;
    const bool is_dy = wave < 4;                              // (uniform) staging role
    const int ch = (tid & 255) >> 2, grp = tid & 3;           // channel of the tile, group of eight pixels

    f32x4 acc[2][TAPS];
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int t = 0; t < TAPS; ++t) acc[n][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;

    for (int unit = blockIdx.x; unit < p.nstrips; unit += gridDim.x) {
        const int b = unit / (p.sy * p.sx), rem = unit - b * (p.sy * p.sx);
        const int y0 = (rem / p.sx) * SEG, c0 = (rem % p.sx) * 32;
        const int y1 = min(p.H, y0 + SEG);
        // this thread's staging source: dy threads read channel co0 + ch of dy, x threads channel ci0 + ch of x
        const int cvalid = is_dy ? min(64, p.Cout - co0) : min(64, p.Cin - ci0);
        const float* src = is_dy ? p.dy + (int64_t)b * p.dy_bs + (int64_t)co0 * HW : p.x + (int64_t)b * p.x_bs + (int64_t)ci0 * HW;
        const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, cvalid * HW * 4, 0x00020000);
        const int px = c0 + grp * 8;
        float e[10];                                           // [0] = pixel px - 1, [1..8] = px .. px + 7, [9] = px + 8 (dy only)
        auto load_row = [&](int y) {
            const bool rowok = y >= 0 && y < p.H && ch < cvalid;
            const unsigned base = (unsigned)((ch * HW + y * p.W + px) * 4);
            const f32x4 lo = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (rowok && px < p.W) ? base : OOB, 0, 0));
            const f32x4 hi = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (rowok && px + 4 < p.W) ? base + 16 : OOB, 0, 0));
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                e[1 + j] = lo[j];
                e[5 + j] = hi[j];
            }
            if (is_dy && KS > 1) {
                e[0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (rowok && px >= 1 && px - 1 < p.W) ? base - 4 : OOB, 0, 0));
                e[9] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (rowok && px + 8 < p.W) ? base + 32 : OOB, 0, 0));
            }
        };
        // split the loaded row and store it: x -> ring slot `slot`; dy -> buffer `buf`, three shifted copies; `count`: add to the bias sum
        auto store_row = [&](int slot, int buf, bool count) {
            if (is_dy) {
                __bf16 pc[3][10];
#pragma unroll
                for (int i = (KS > 1 ? 0 : 1); i < (KS > 1 ? 10 : 9); ++i) wsplit3<SIX>(e[i], pc[0][i], pc[1][i], pc[2][i]);
                if (count) {
#pragma unroll
                    for (int i = 1; i <= 8; ++i) bsum += e[i];
                }
                char* dst = lds + buf * DYBUF + ch * DYROW + grp * 16;
#pragma unroll
                for (int s = 0; s < KS; ++s)                  // copy s (= kx): element k <-> dy[px0 + k + PAD - s]
#pragma unroll
                    for (int q = 0; q < NQ; ++q) {
                        bf16x8 v;
#pragma unroll
                        for (int j = 0; j < 8; ++j) v[j] = pc[q][j + 1 + PAD - s];
                        *reinterpret_cast<bf16x8*>(dst + (s * 3 + q) * DYPLANE) = v;
                    }
            } else {
                __bf16 pc[3][8];
#pragma unroll
                for (int i = 0; i < 8; ++i) wsplit3<SIX>(e[1 + i], pc[0][i], pc[1][i], pc[2][i]);
                char* dst = lds + 2 * DYBUF + ch * XROW + slot * 64 + grp * 16;
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    bf16x8 v;
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = pc[q][j];
                    *reinterpret_cast<bf16x8*>(dst + q * XPLANE) = v;
                }
            }
        };
        // ---- fill: x rows y0 - PAD .. y0 + PAD -> slots 0 .. KS - 1; dy row y0 -> buffer 0
        __syncthreads();                                       // (the previous unit's last reads)
        if (is_dy) {
            load_row(y0);
            store_row(0, 0, true);
        } else {
#pragma unroll
            for (int i = 0; i < KS; ++i) {
                load_row(y0 - PAD + i);
                store_row(i, 0, false);
            }
        }
        __syncthreads();
        for (int y = y0; y < y1; ++y) {
            const int j = y - y0;
            // next step's rows: dy row y + 1 -> buffer (j + 1) & 1, x row y + PAD + 1 -> slot (j + KS) & 3 (free: last read at step y - 1)
            load_row(is_dy ? y + 1 : y + PAD + 1);
            const char* dyb = lds + (j & 1) * DYBUF + (mt * 16 + c16) * DYROW + g * 16;
            const char* xb = lds + 2 * DYBUF + ((nh * 2) * 16 + c16) * XROW + g * 16;
            bf16x8 A[KS][3];
#pragma unroll
            for (int s = 0; s < KS; ++s)
#pragma unroll
                for (int q = 0; q < NQ; ++q) A[s][q] = *reinterpret_cast<const bf16x8*>(dyb + (s * 3 + q) * DYPLANE);
#pragma unroll
            for (int ky = 0; ky < KS; ++ky) {
                const int slot = (j + ky) & 3;
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    bf16x8 Bq[3];
#pragma unroll
                    for (int q = 0; q < NQ; ++q) Bq[q] = *reinterpret_cast<const bf16x8*>(xb + q * XPLANE + n * 16 * XROW + slot * 64);
#pragma unroll
                    for (int kx = 0; kx < KS; ++kx) {
                        f32x4 c = acc[n][ky * KS + kx];
                        if constexpr (SIX) {
                            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[kx][2], Bq[0], c, 0, 0, 0);
                            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[kx][1], Bq[1], c, 0, 0, 0);
                            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[kx][0], Bq[2], c, 0, 0, 0);
                            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[kx][1], Bq[0], c, 0, 0, 0);
                            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[kx][0], Bq[1], c, 0, 0, 0);
                        }
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[kx][0], Bq[0], c, 0, 0, 0);
                        acc[n][ky * KS + kx] = c;
                    }
                }
            }
            store_row((j + KS) & 3, (j + 1) & 1, y + 1 < y1);
            __syncthreads();
        }
    }

    // ---- bias gradient: the four pixel groups of a dy channel are four adjacent lanes
    if (p.bpart && blockIdx.z == 0) {
        float t = bsum + __shfl_xor(bsum, 1, 64);
        t += __shfl_xor(t, 2, 64);
        if (is_dy && grp == 0 && co0 + ch < p.Cout) p.bpart[(int64_t)blockIdx.x * p.Cout + co0 + ch] = t;
    }
    // accumulator register r of tile (n, tap): co = co0 + 16 mt + 4 g + r, ci = ci0 + 16 (2 nh + n) + c16
    float* out = p.part + (int64_t)blockIdx.x * p.Cout * p.Cin * TAPS;
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const int ci = ci0 + (nh * 2 + n) * 16 + c16;
        if (ci >= p.Cin) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int co = co0 + mt * 16 + 4 * g + r;
            if (co < p.Cout) {
#pragma unroll
                for (int t = 0; t < TAPS; ++t) out[((int64_t)co * p.Cin + ci) * TAPS + t] = acc[n][t][r];
            }
        }
    }
}

}  // namespace
int g_cwfa_wgrad_rows = 1;
int g_cwfa_wgrad_split = 0;  // 3x3: the split-bf16 form on the bf16 matrix cores (option "wgrad_split"; ops.set_precision sets it)
extern int g_cwfa_split_products;   // 3x3: the LDS-DMA / row-paired form when the image rows are 16-byte aligned (option "wgrad_rows")
namespace {

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
    if (B <= 0 || Cin <= 0 || H <= 0 || W <= 0 || Cout <= 0 || (ks != 1 && ks != 3 && ks != 7)) return 0;
    if (ks == 7)      // seven 1x7 passes: per-worker [Cout][Cin][7] partials + bias partials, and the reduced row
        return ((int64_t)wgrad_workers(B, H, W, Cout, Cin, 3) * ((int64_t)Cout * Cin * 7 + Cout) + (int64_t)Cout * Cin * 7) * 4;
    return (int64_t)wgrad_workers(B, H, W, Cout, Cin, ks) * ((int64_t)Cout * Cin * ks * ks + Cout) * 4;
}

namespace {
// dw[co][ci][ky][kx] (7x7) <- beta * dw + row[co][ci][kx]
__global__ __launch_bounds__(256) void wgrad_scatter_row_kernel(const float* __restrict__ row, float* __restrict__ dw, int64_t ncc, int ky,
                                                                float beta) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ncc * 7) return;
    const int64_t cc = i / 7;
    const int kx = (int)(i % 7);
    float* d = dw + (cc * 7 + ky) * 7 + kx;
    *d = beta != 0.f ? beta * *d + row[i] : row[i];
}
}  // namespace

extern "C" int cwfa_conv2d_wgrad_f32(const float* x, const float* dy, float* dw, float* db, void* workspace, int B, int Cin, int H,
                                     int W, int Cout, int ks, int64_t x_bs, int64_t dy_bs, float beta, void* stream) {
    CWFA_REQUIRE(x && dy && dw && workspace, CWFA_E_INVAL, "cwfa_conv2d_wgrad_f32: null pointer");
    CWFA_REQUIRE(ks == 1 || ks == 3 || ks == 7, CWFA_E_SHAPE, "cwfa_conv2d_wgrad_f32: kernel size %d (1, 3 and 7 are built)", ks);
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
    const int workers = wgrad_workers(B, H, W, Cout, Cin, ks == 7 ? 3 : ks);
    if (ks == 7) {
        const int64_t nrow = (int64_t)Cout * Cin * 7;
        float* row = p.part + (int64_t)workers * (nrow + Cout);
        static bool attr7 = false;
        if (!attr7) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_kernel<1, 7>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, WgCfg<1, 7>::LDS_BYTES);
            CWFA_REQUIRE(e == hipSuccess, CWFA_E_HIP, "cwfa_conv2d_wgrad_f32: hipFuncSetAttribute: %s", hipGetErrorString(e));
            attr7 = true;
        }
        dim3 grid7(workers, (Cout + 63) / 64, (Cin + 63) / 64);
        for (int ky = 0; ky < 7; ++ky) {
            p.row_off = ky - 3;
            p.bpart = (db && ky == 0) ? p.part + (int64_t)workers * nrow : nullptr;
            hipLaunchKernelGGL((conv_wgrad_kernel<1, 7>), grid7, dim3(256), (WgCfg<1, 7>::LDS_BYTES), st, p);
            CWFA_LAUNCH_CHECK("cwfa_conv2d_wgrad_f32");
            hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((nrow + 63) / 64)), dim3(256), 0, st, p.part, row, nrow, workers, 0.f);
            hipLaunchKernelGGL(wgrad_scatter_row_kernel, dim3((unsigned)((nrow + 255) / 256)), dim3(256), 0, st, row, dw,
                               (int64_t)Cout * Cin, ky, beta);
            CWFA_LAUNCH_CHECK("cwfa_conv2d_wgrad_f32");
            if (p.bpart) {
                hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((Cout + 63) / 64)), dim3(256), 0, st, p.bpart, db, (int64_t)Cout,
                                   workers, beta);
                CWFA_LAUNCH_CHECK("cwfa_conv2d_wgrad_f32");
            }
        }
        return CWFA_OK;
    }
    p.bpart = db ? p.part + (int64_t)workers * n : nullptr;
    dim3 grid(workers, (Cout + 63) / 64, (Cin + 63) / 64);
    static bool attr1 = false, attr3 = false;
    // (1x1: only banks of one 64 x 64 tile -- the sub-networks' -- where a block stages each operand once: 60 -> 52 us at 512 x 512; with
    //  several tiles every block re-splits its operands for 12 MFMAs per step and the fp32 form is faster, 444 vs 628 us at 256 -> 256)
    if ((ks == 3 || (ks == 1 && Cout <= 64 && Cin <= 64)) && g_cwfa_wgrad_split && W % 4 == 0 && x_bs % 4 == 0 && dy_bs % 4 == 0 && cwfa_aligned16(x) && cwfa_aligned16(dy) &&
        (int64_t)H * W % 4 == 0) {
        // units = (sample, 32-pixel column, segment of ws::SEG rows); the same workers and partial banks as the fp32 forms
        p.sx = (W + 31) / 32;
        p.sy = (H + ws::SEG - 1) / ws::SEG;
        p.nstrips = B * p.sy * p.sx;
        const bool six = g_cwfa_split_products != 1;
        auto kern = ks == 3 ? (six ? &conv_wgrad_split_kernel<true, 3> : &conv_wgrad_split_kernel<false, 3>)
                            : (six ? &conv_wgrad_split_kernel<true, 1> : &conv_wgrad_split_kernel<false, 1>);
        static bool attr_s4[4] = {false, false, false, false};
        bool& attr_ok = attr_s4[(ks == 1 ? 2 : 0) + (six ? 1 : 0)];
        if (!attr_ok) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, ws::LDS_BYTES);
            CWFA_REQUIRE(e == hipSuccess, CWFA_E_HIP, "cwfa_conv2d_wgrad_f32: hipFuncSetAttribute: %s", hipGetErrorString(e));
            attr_ok = true;
        }
        // (one block per CU: 144 KB of LDS; the fp32 1x1 form asks for two per CU)
        const int tiles = (int)(grid.y * grid.z);
        dim3 sgrid(min(min(workers, max(1, (256 + tiles - 1) / tiles)), p.nstrips), grid.y, grid.z);
        hipLaunchKernelGGL(kern, sgrid, dim3(512), ws::LDS_BYTES, st, p);
        CWFA_LAUNCH_CHECK("cwfa_conv2d_wgrad_f32");
        {
            const int nbw = (int)((n + 63) / 64), nbb = db ? (Cout + 63) / 64 : 0;
            hipLaunchKernelGGL(wgrad_reduce2_kernel, dim3((unsigned)(nbw + nbb)), dim3(256), 0, st, p.part, dw, n, p.bpart, db, (int64_t)Cout, nbw,
                               (int)sgrid.x, beta);
            CWFA_LAUNCH_CHECK("cwfa_conv2d_wgrad_f32");
        }
        return CWFA_OK;
    }
    const bool rows_form = ks == 3 && g_cwfa_wgrad_rows && W % 4 == 0 && x_bs % 4 == 0 && dy_bs % 4 == 0 && cwfa_aligned16(x) &&
                           cwfa_aligned16(dy) && (int64_t)H * W % 4 == 0;
    if (rows_form) {
        static bool attr_r = false;
        if (!attr_r) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_rows_kernel),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, wr::LDS_BYTES);
            CWFA_REQUIRE(e == hipSuccess, CWFA_E_HIP, "cwfa_conv2d_wgrad_f32: hipFuncSetAttribute: %s", hipGetErrorString(e));
            attr_r = true;
        }
        hipLaunchKernelGGL(conv_wgrad_rows_kernel, grid, dim3(256), wr::LDS_BYTES, st, p);
    } else if (ks == 3) {
        if (!attr3) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_kernel<3, 3>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, WgCfg<3, 3>::LDS_BYTES);
            CWFA_REQUIRE(e == hipSuccess, CWFA_E_HIP, "cwfa_conv2d_wgrad_f32: hipFuncSetAttribute: %s", hipGetErrorString(e));
            attr3 = true;
        }
        hipLaunchKernelGGL((conv_wgrad_kernel<3, 3>), grid, dim3(256), (WgCfg<3, 3>::LDS_BYTES), st, p);
    } else {
        if (!attr1) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_kernel<1, 1>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, WgCfg<1, 1>::LDS_BYTES);
            CWFA_REQUIRE(e == hipSuccess, CWFA_E_HIP, "cwfa_conv2d_wgrad_f32: hipFuncSetAttribute: %s", hipGetErrorString(e));
            attr1 = true;
        }
        hipLaunchKernelGGL((conv_wgrad_kernel<1, 1>), grid, dim3(256), (WgCfg<1, 1>::LDS_BYTES), st, p);
    }
    CWFA_LAUNCH_CHECK("cwfa_conv2d_wgrad_f32");
    {
        const int nbw = (int)((n + 63) / 64), nbb = db ? (Cout + 63) / 64 : 0;
        hipLaunchKernelGGL(wgrad_reduce2_kernel, dim3((unsigned)(nbw + nbb)), dim3(256), 0, st, p.part, dw, n, p.bpart, db, (int64_t)Cout, nbw, workers,
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

// =====================================================================================================================
// Backward of the condition net's 3-D stage  y = W2 * PReLU(W1 * x + b1) + b2  (Conv3d 1 -> K -> 1 over (H, W, depth),
// networks.py:221-225,239; weight index (ih*3 + iw)*3 + id as in conv3d.hip) and of its 2-D PReLUs -- the autograd of the
// reference's `optimizer_cond` path (CWFA.py:1008-1012).  The K-channel hidden volume is materialised here (q and m,
// B*K*D*H*W floats each: 1.6 GB at 512 x 512 x 48, K = 32 -- small against 288 GB) and the pieces are separate passes:
//   conv3d_hidden_fwd :  q = W1 * x + b1                                   (VALU, 27 K FMAs per voxel)
//   conv3d_hidden_bwd :  m = (W2^T * dy) . PReLU'(q),  dalpha += sum (W2^T * dy) . min(q, 0)
//   conv3d_wgrad      :  dW[k][tap] = sum_p A[k][p] * src[p +- (tap - 1)]     (fp32 MFMA GEMM, M = K <= 32, N = 27 taps + 1
//                        ones column for the bias, reduction over voxels; operands straight from global memory)
//   conv3d_input_bwd  :  dx = sum_k W1[k]^T * m[k]
// =====================================================================================================================
namespace {

struct Vox {
    int d, y, x;
};

// neighbour offsets of the 27 taps around voxel (d,y,x): sign +1 -> p + (tap - 1), sign -1 -> p - (tap - 1); byte offset
// inside one [D][H][W] volume, or the out-of-range marker for taps that fall into the zero padding
__device__ __forceinline__ void tap_offsets(Vox v, int D, int H, int W, int sign, unsigned (&off)[27]) {
#pragma unroll
    for (int t = 0; t < 27; ++t) {
        const int ih = t / 9, iw = (t / 3) % 3, id = t % 3;
        const int y = v.y + sign * (ih - 1), x = v.x + sign * (iw - 1), d = v.d + sign * (id - 1);
        const bool ok = y >= 0 && y < H && x >= 0 && x < W && d >= 0 && d < D;
        off[t] = ok ? (unsigned)(((int64_t)d * H + y) * W + x) * 4u : 0x80000000u;
    }
}

__global__ __launch_bounds__(256) void conv3d_hidden_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w1,
                                                                const float* __restrict__ b1, float* __restrict__ q, int D, int H,
                                                                int W, int K) {
    const int64_t vol = (int64_t)D * H * W;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= vol) return;
    const int b = blockIdx.y;
    const Vox v{(int)(i / ((int64_t)H * W)), (int)((i / W) % H), (int)(i % W)};
    unsigned off[27];
    tap_offsets(v, D, H, W, +1, off);
    const auto rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x + (int64_t)b * vol), 0, (int)(vol * 4), 0x00020000);
    float nb[27];
#pragma unroll
    for (int t = 0; t < 27; ++t) nb[t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, off[t], 0, 0));
    float* qb = q + (int64_t)b * K * vol + i;
    for (int k = 0; k < K; ++k) {
        float acc = b1[k];
#pragma unroll
        for (int t = 0; t < 27; ++t) acc = fmaf(w1[k * 27 + t], nb[t], acc);
        qb[(int64_t)k * vol] = acc;
    }
}

__global__ __launch_bounds__(256) void conv3d_hidden_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ w2,
                                                                const float* __restrict__ q, const float* __restrict__ alpha_p,
                                                                float* __restrict__ m, double* __restrict__ dalpha, int D, int H,
                                                                int W, int K) {
    __shared__ double red[16];
    const int64_t vol = (int64_t)D * H * W;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    const float alpha = *alpha_p;
    float da = 0.f;
    if (i < vol) {
        const Vox v{(int)(i / ((int64_t)H * W)), (int)((i / W) % H), (int)(i % W)};
        unsigned off[27];
        tap_offsets(v, D, H, W, -1, off);
        const auto rd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dy + (int64_t)b * vol), 0, (int)(vol * 4), 0x00020000);
        float nb[27];
#pragma unroll
        for (int t = 0; t < 27; ++t) nb[t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rd, off[t], 0, 0));
        const float* qb = q + (int64_t)b * K * vol + i;
        float* mb = m + (int64_t)b * K * vol + i;
        for (int k = 0; k < K; ++k) {
            float dh = 0.f;
#pragma unroll
            for (int t = 0; t < 27; ++t) dh = fmaf(w2[k * 27 + t], nb[t], dh);
            const float qv = qb[(int64_t)k * vol];
            mb[(int64_t)k * vol] = qv > 0.f ? dh : alpha * dh;
            da += qv > 0.f ? 0.f : dh * qv;
        }
    }
    if (dalpha) {
        const double tot = cwfa_block_sum((double)da, red);
        if (threadIdx.x == 0) atomicAdd(dalpha, tot);
    }
}

// Four voxels per thread for aligned rows (W % 4 == 0): a (depth, row) line of the stencil is one 4-byte, one 16-byte and one
// 4-byte load, and every hidden-channel value moves as a 16-byte word.  SIGN +1: q = W1 * x + b1;  SIGN -1: m = (W2^T * dy).PReLU'(q).
template <int SIGN>
__global__ __launch_bounds__(256) void conv3d_hidden4_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                             const float* __restrict__ b1, const float* __restrict__ q,
                                                             const float* __restrict__ alpha_p, float* __restrict__ out,
                                                             double* __restrict__ dalpha, int D, int H, int W, int K) {
    __shared__ double red[16];
    const int64_t vol = (int64_t)D * H * W, q4 = vol / 4;
    const int64_t i4 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    double da = 0.0;
    if (i4 < q4) {
        const int64_t i = i4 * 4;
        const int x0 = (int)(i % W), y = (int)((i / W) % H), d = (int)(i / ((int64_t)H * W));
        const float* ib = in + (int64_t)b * vol;
        float v[9][6];                               // line (ih, id): voxels x0-1 .. x0+4 of row y + SIGN (ih-1), depth d + SIGN (id-1)
#pragma unroll
        for (int l = 0; l < 9; ++l) {
            const int ih = l / 3, id = l % 3;
            const int sy = y + SIGN * (ih - 1), sd = d + SIGN * (id - 1);
            const bool ok = sy >= 0 && sy < H && sd >= 0 && sd < D;
            const float* lp = ib + ((int64_t)(ok ? sd : 0) * H + (ok ? sy : 0)) * W + x0;
            const f32x4 mid = ok ? *reinterpret_cast<const f32x4*>(lp) : f32x4{0.f, 0.f, 0.f, 0.f};
            v[l][0] = ok && x0 > 0 ? lp[-1] : 0.f;
            v[l][1] = mid[0]; v[l][2] = mid[1]; v[l][3] = mid[2]; v[l][4] = mid[3];
            v[l][5] = ok && x0 + 4 < W ? lp[4] : 0.f;
        }
        const float alpha = SIGN < 0 ? *alpha_p : 0.f;
        for (int k = 0; k < K; ++k) {
            float acc[4];
            const float bias = SIGN > 0 ? b1[k] : 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = bias;
#pragma unroll
            for (int l = 0; l < 9; ++l)
#pragma unroll
                for (int iw = 0; iw < 3; ++iw) {
                    const float wv = w[k * 27 + ((l / 3) * 3 + iw) * 3 + (l % 3)];
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[j] = fmaf(wv, v[l][j + 1 + SIGN * (iw - 1)], acc[j]);
                }
            const int64_t o = ((int64_t)b * K + k) * vol + i;
            if constexpr (SIGN > 0) {
                *reinterpret_cast<f32x4*>(out + o) = f32x4{acc[0], acc[1], acc[2], acc[3]};
            } else {
                const f32x4 qv = *reinterpret_cast<const f32x4*>(q + o);
                f32x4 mv;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    mv[j] = qv[j] > 0.f ? acc[j] : alpha * acc[j];
                    da += qv[j] > 0.f ? 0.0 : (double)acc[j] * (double)qv[j];
                }
                *reinterpret_cast<f32x4*>(out + o) = mv;
            }
        }
    }
    if constexpr (SIGN < 0) {
        if (dalpha) {
            const double tot = cwfa_block_sum(da, red);
            if (threadIdx.x == 0) atomicAdd(dalpha, tot);
        }
    }
}

__global__ __launch_bounds__(256) void conv3d_input_bwd_kernel(const float* __restrict__ m, const float* __restrict__ w1,
                                                               float* __restrict__ dx, int D, int H, int W, int K) {
    const int64_t vol = (int64_t)D * H * W;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= vol) return;
    const int b = blockIdx.y;
    const Vox v{(int)(i / ((int64_t)H * W)), (int)((i / W) % H), (int)(i % W)};
    unsigned off[27];
    tap_offsets(v, D, H, W, -1, off);
    // one descriptor over the K hidden volumes of this sample (< 2 GiB, checked by the host); the channel moves through the
    // scalar offset and the padding marker 0x80000000 is out of range whichever way the offsets are summed
    const auto rm = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(m + (int64_t)b * K * vol), 0, (int)((int64_t)K * vol * 4),
                                                      0x00020000);
    float acc = 0.f;
    for (int k = 0; k < K; ++k) {
        const int so = (int)((int64_t)k * vol * 4);
#pragma unroll
        for (int t = 0; t < 27; ++t)
            acc = fmaf(w1[k * 27 + t], __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rm, off[t], so, 0)), acc);
    }
    dx[(int64_t)b * vol + i] = acc;
}

// The same for images whose rows are 16-byte aligned (W % 4 == 0): a thread owns FOUR consecutive voxels along x, so a
// (depth, row) line of the stencil is six values -- one 4-byte, one 16-byte and one 4-byte load -- feeding 12 FMAs:
// 27 load instructions per hidden channel and four outputs instead of 108.
__global__ __launch_bounds__(256) void conv3d_input_bwd4_kernel(const float* __restrict__ m, const float* __restrict__ w1,
                                                                float* __restrict__ dx, int D, int H, int W, int K) {
    const int64_t vol = (int64_t)D * H * W, q4 = vol / 4;
    const int64_t i4 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i4 >= q4) return;
    const int b = blockIdx.y;
    const int64_t i = i4 * 4;
    const int x0 = (int)(i % W), y = (int)((i / W) % H), d = (int)(i / ((int64_t)H * W));
    constexpr unsigned OOB = 0x80000000u;
    // line (id, ih): source line (d - (id-1), y - (ih-1)); byte offsets of its three pieces, or the out-of-range marker
    unsigned ol[9], om[9], orr[9];
#pragma unroll
    for (int l = 0; l < 9; ++l) {
        const int ih = l / 3, id = l % 3;
        const int sy = y - (ih - 1), sd = d - (id - 1);
        const bool ok = sy >= 0 && sy < H && sd >= 0 && sd < D;
        const unsigned base = (unsigned)((((int64_t)sd * H + sy) * W + x0) * 4);
        om[l] = ok ? base : OOB;
        ol[l] = ok && x0 > 0 ? base - 4u : OOB;
        orr[l] = ok && x0 + 4 < W ? base + 16u : OOB;
    }
    const auto rm = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(m + (int64_t)b * K * vol), 0, (int)((int64_t)K * vol * 4),
                                                      0x00020000);
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < K; ++k) {
        const int so = (int)((int64_t)k * vol * 4);
        const float* wk = w1 + k * 27;
#pragma unroll
        for (int l = 0; l < 9; ++l) {
            const int ih = l / 3, id = l % 3;
            float v[6];
            v[0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rm, ol[l], so, 0));
            const f32x4 mid = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rm, om[l], so, 0));
            v[1] = mid[0]; v[2] = mid[1]; v[3] = mid[2]; v[4] = mid[3];
            v[5] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rm, orr[l], so, 0));
            // dx[x0 + j] += W1[k][ih][iw][id] * m[x0 + j - (iw - 1)]  ->  v index j + 1 - (iw - 1) = j + 2 - iw
#pragma unroll
            for (int iw = 0; iw < 3; ++iw) {
                const float wv = wk[(ih * 3 + iw) * 3 + id];
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = fmaf(wv, v[j + 2 - iw], acc[j]);
            }
        }
    }
    *reinterpret_cast<f32x4*>(dx + (int64_t)b * vol + i) = f32x4{acc[0], acc[1], acc[2], acc[3]};
}

// GEMM over voxels on the matrix cores: D[k][tap] += A[k][p] * src[p + sign*(tap - 1)];  column 27 multiplies ones (the
// bias gradient of the convolution whose output gradient is A).  A wave owns runs of 64 voxels along x (32 k-steps of two
// voxels); lane (l31, kh) feeds row k = l31 of A and column tap = l31 of the shifted source, both straight from global
// memory (consecutive steps walk consecutive addresses: cache hits).  Each wave writes its 32 x 32 partial; the partials are
// summed in a fixed order by wgrad_reduce_kernel.
__global__ __launch_bounds__(256) void conv3d_wgrad_kernel(const float* __restrict__ a, const float* __restrict__ src,
                                                           const float* __restrict__ alpha_p, float* __restrict__ part, int B, int D,
                                                           int H, int W, int K, int sign, int want_bias) {
    const int lane = threadIdx.x & 63, l31 = lane & 31, kh = lane >> 5;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
    const int64_t HW = (int64_t)H * W, vol = (int64_t)D * HW;
    const int chunks = (W + 63) / 64;
    const int64_t items = (int64_t)B * D * H * chunks;
    const float alpha = alpha_p ? *alpha_p : 1.f;
    const int ih = l31 / 9, iw = (l31 / 3) % 3, id = l31 % 3;       // this lane's tap (l31 < 27)
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int64_t it = wave; it < items; it += nwaves) {
        const int ch = (int)(it % chunks);
        const int64_t row = it / chunks;
        const int y = (int)(row % H), d = (int)((row / H) % D), b = (int)(row / ((int64_t)H * D));
        const int x0 = ch * 64;
        const float* ap = a + ((int64_t)b * K + l31) * vol + (int64_t)d * HW + (int64_t)y * W;
        const int sy = y + sign * (ih - 1), sd = d + sign * (id - 1), sxo = sign * (iw - 1);
        const bool row_ok = l31 < 27 && sy >= 0 && sy < H && sd >= 0 && sd < D;
        const float* sp = src + (int64_t)b * vol + (int64_t)(row_ok ? sd : 0) * HW + (int64_t)(row_ok ? sy : 0) * W;
#pragma unroll 8
        for (int s = 0; s < 32; ++s) {
            const int x = x0 + 2 * s + kh;
            float av = 0.f, bv = 0.f;
            if (x < W) {
                if (l31 < K) {
                    av = ap[x];
                    if (alpha_p) av = av > 0.f ? av : alpha * av;
                }
                const int sx = x + sxo;
                if (row_ok && sx >= 0 && sx < W) bv = sp[sx];
                if (l31 == 27 && want_bias) bv = 1.f;
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
        }
    }
    float* out = part + (int64_t)wave * 1024;
#pragma unroll
    for (int r = 0; r < 16; ++r) out[acc_row(r, kh) * 32 + l31] = acc[r];
}

// The same GEMM for 16-byte aligned rows (W % 4 == 0): the MFMA k index pairs the two HALVES of a 64-voxel run (lane half
// kh walks voxels x0 + 32 kh + s), so a lane's operands of consecutive k-steps are consecutive words -- one 16-byte load of
// A per four k-steps, and for the shifted source the aligned chunk plus its neighbour in the lane's shift direction (the
// four operands are picked with selects): 3 load instructions per four k-steps instead of 8.
__global__ __launch_bounds__(256) void conv3d_wgrad4_kernel(const float* __restrict__ a, const float* __restrict__ src,
                                                            const float* __restrict__ alpha_p, float* __restrict__ part, int B, int D,
                                                            int H, int W, int K, int sign, int want_bias) {
    const int lane = threadIdx.x & 63, l31 = lane & 31, kh = lane >> 5;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
    const int64_t HW = (int64_t)H * W, vol = (int64_t)D * HW;
    const int chunks = (W + 63) / 64;
    const int64_t items = (int64_t)B * D * H * chunks;
    const float alpha = alpha_p ? *alpha_p : 1.f;
    const int ih = l31 / 9, iw = (l31 / 3) % 3, id = l31 % 3;
    const int sxo = sign * (iw - 1);                       // this lane's shift along x: -1, 0, +1
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int64_t it = wave; it < items; it += nwaves) {
        const int ch = (int)(it % chunks);
        const int64_t row = it / chunks;
        const int y = (int)(row % H), d = (int)((row / H) % D), b = (int)(row / ((int64_t)H * D));
        const int xb = ch * 64 + 32 * kh;                   // first voxel of this lane's half-run
        const float* ap = a + ((int64_t)b * K + l31) * vol + (int64_t)d * HW + (int64_t)y * W;
        const int sy = y + sign * (ih - 1), sd = d + sign * (id - 1);
        const bool row_ok = l31 < 27 && sy >= 0 && sy < H && sd >= 0 && sd < D;
        const float* sp = src + (int64_t)b * vol + (int64_t)(row_ok ? sd : 0) * HW + (int64_t)(row_ok ? sy : 0) * W;
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const int x = xb + 4 * g;                       // voxels x .. x+3 are this group's four k-steps
            f32x4 av = zero4, cur = zero4, nb = zero4;
            if (x < W) {                                    // W % 4 == 0: a chunk is inside or outside as a whole
                if (l31 < K) {
                    av = *reinterpret_cast<const f32x4*>(ap + x);
                    if (alpha_p) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) av[e] = av[e] > 0.f ? av[e] : alpha * av[e];
                    }
                }
                if (row_ok) {
                    cur = *reinterpret_cast<const f32x4*>(sp + x);
                    const int xn = x + 4 * sxo;              // the neighbouring chunk in the shift direction
                    if (sxo != 0 && xn >= 0 && xn < W) nb = *reinterpret_cast<const f32x4*>(sp + xn);
                }
            }
            // source voxel x + e + sxo
            f32x4 bv;
            bv[0] = sxo < 0 ? nb[3] : sxo > 0 ? cur[1] : cur[0];
            bv[1] = sxo < 0 ? cur[0] : sxo > 0 ? cur[2] : cur[1];
            bv[2] = sxo < 0 ? cur[1] : sxo > 0 ? cur[3] : cur[2];
            bv[3] = sxo < 0 ? cur[2] : sxo > 0 ? nb[0] : cur[3];
            if (l31 == 27 && want_bias && x < W) bv = f32x4{1.f, 1.f, 1.f, 1.f};
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], bv[e], acc, 0, 0, 0);
        }
    }
    float* out = part + (int64_t)wave * 1024;
#pragma unroll
    for (int r = 0; r < 16; ++r) out[acc_row(r, kh) * 32 + l31] = acc[r];
}

// PReLU backward from the layer's OUTPUT o = PReLU(q), single alpha > 0:  y = g * (o > 0 ? 1 : alpha),
// dalpha += sum g * min(q, 0) with q = o / alpha where o < 0
__global__ __launch_bounds__(256) void prelu_bwd_kernel(const float* __restrict__ g, const float* __restrict__ o,
                                                        const float* __restrict__ alpha_p, float* __restrict__ y,
                                                        double* __restrict__ dalpha, int64_t n, int64_t g_bs, int64_t o_bs, int64_t y_bs) {
    __shared__ double red[16];
    const int b = blockIdx.y;
    const float alpha = *alpha_p;
    double da = 0.0;
    // grid-stride: a bounded number of blocks, so that the one atomic per block does not serialise the kernel
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float gv = g[b * g_bs + i], ov = o[b * o_bs + i];
        y[b * y_bs + i] = ov > 0.f ? gv : alpha * gv;
        da += ov > 0.f ? 0.0 : (double)gv * ((double)ov / (double)alpha);
    }
    if (dalpha) {
        const double tot = cwfa_block_sum(da, red);
        if (threadIdx.x == 0) atomicAdd(dalpha, tot);
    }
}

int check_c3(const char* name, int B, int D, int H, int W, int K) {
    CWFA_REQUIRE(B >= 0 && D > 0 && H > 0 && W > 0 && K > 0 && B <= 65535, CWFA_E_SHAPE, "%s: bad shape", name);
    CWFA_REQUIRE((int64_t)D * H * W * 4 < (1ll << 31), CWFA_E_SHAPE, "%s: volume too large for 32-bit offsets", name);
    CWFA_REQUIRE((int64_t)K * D * H * W * 4 < (1ll << 31), CWFA_E_SHAPE, "%s: hidden volume too large for 32-bit offsets", name);
    return CWFA_OK;
}

}  // namespace

extern "C" int cwfa_conv3d_hidden_fwd_f32(const float* x, const float* w1, const float* b1, float* q, int B, int D, int H, int W,
                                          int K, void* stream) {
    CWFA_REQUIRE(x && w1 && b1 && q, CWFA_E_INVAL, "cwfa_conv3d_hidden_fwd_f32: null pointer");
    int rc = check_c3("cwfa_conv3d_hidden_fwd_f32", B, D, H, W, K);
    if (rc) return rc;
    if (B == 0) return CWFA_OK;
    const int64_t vol = (int64_t)D * H * W;
    if (W % 4 == 0 && cwfa_aligned16(x) && cwfa_aligned16(q))
        hipLaunchKernelGGL(conv3d_hidden4_kernel<1>, dim3((unsigned)((vol / 4 + 255) / 256), B), dim3(256), 0, (hipStream_t)stream, x, w1, b1,
                           (const float*)nullptr, (const float*)nullptr, q, (double*)nullptr, D, H, W, K);
    else
        hipLaunchKernelGGL(conv3d_hidden_fwd_kernel, dim3((unsigned)((vol + 255) / 256), B), dim3(256), 0, (hipStream_t)stream, x, w1, b1,
                           q, D, H, W, K);
    CWFA_LAUNCH_CHECK("cwfa_conv3d_hidden_fwd_f32");
    return CWFA_OK;
}

extern "C" int cwfa_conv3d_hidden_bwd_f32(const float* dy, const float* w2, const float* q, const float* alpha, float* m,
                                          double* dalpha, int B, int D, int H, int W, int K, void* stream) {
    CWFA_REQUIRE(dy && w2 && q && alpha && m, CWFA_E_INVAL, "cwfa_conv3d_hidden_bwd_f32: null pointer");
    int rc = check_c3("cwfa_conv3d_hidden_bwd_f32", B, D, H, W, K);
    if (rc) return rc;
    if (B == 0) return CWFA_OK;
    const int64_t vol = (int64_t)D * H * W;
    if (W % 4 == 0 && cwfa_aligned16(dy) && cwfa_aligned16(q) && cwfa_aligned16(m))
        hipLaunchKernelGGL(conv3d_hidden4_kernel<-1>, dim3((unsigned)((vol / 4 + 255) / 256), B), dim3(256), 0, (hipStream_t)stream, dy, w2,
                           (const float*)nullptr, q, alpha, m, dalpha, D, H, W, K);
    else
        hipLaunchKernelGGL(conv3d_hidden_bwd_kernel, dim3((unsigned)((vol + 255) / 256), B), dim3(256), 0, (hipStream_t)stream, dy, w2, q,
                           alpha, m, dalpha, D, H, W, K);
    CWFA_LAUNCH_CHECK("cwfa_conv3d_hidden_bwd_f32");
    return CWFA_OK;
}

extern "C" int cwfa_conv3d_input_bwd_f32(const float* m, const float* w1, float* dx, int B, int D, int H, int W, int K, void* stream) {
    CWFA_REQUIRE(m && w1 && dx, CWFA_E_INVAL, "cwfa_conv3d_input_bwd_f32: null pointer");
    int rc = check_c3("cwfa_conv3d_input_bwd_f32", B, D, H, W, K);
    if (rc) return rc;
    if (B == 0) return CWFA_OK;
    const int64_t vol = (int64_t)D * H * W;
    if (W % 4 == 0 && cwfa_aligned16(m) && cwfa_aligned16(dx))
        hipLaunchKernelGGL(conv3d_input_bwd4_kernel, dim3((unsigned)((vol / 4 + 255) / 256), B), dim3(256), 0, (hipStream_t)stream, m, w1,
                           dx, D, H, W, K);
    else
        hipLaunchKernelGGL(conv3d_input_bwd_kernel, dim3((unsigned)((vol + 255) / 256), B), dim3(256), 0, (hipStream_t)stream, m, w1, dx,
                           D, H, W, K);
    CWFA_LAUNCH_CHECK("cwfa_conv3d_input_bwd_f32");
    return CWFA_OK;
}

static int c3_wgrad_blocks(int B, int D, int H, int W) {
    const int64_t items = (int64_t)B * D * H * ((W + 63) / 64);
    int64_t blocks = (items + 3) / 4;
    if (blocks > 512) blocks = 512;
    return (int)(blocks < 1 ? 1 : blocks);
}

extern "C" int64_t cwfa_conv3d_wgrad_workspace_bytes(int B, int D, int H, int W) {
    if (B <= 0 || D <= 0 || H <= 0 || W <= 0) return 4096;
    return (int64_t)c3_wgrad_blocks(B, D, H, W) * 4 * 1024 * 4;
}

/* a [B,K,D,H,W] (K <= 32), src [B,D,H,W]; out32 [32][32] floats: out32[k][tap] = beta*out32 + sum_p act(a[k][p]) *
 * src[p + sign*(tap-1)] for tap < 27, out32[k][27] = sum_p act(a[k][p]) if want_bias; act = PReLU(alpha) when alpha != NULL */
extern "C" int cwfa_conv3d_wgrad_f32(const float* a, const float* src, const float* alpha, float* out32, void* workspace, int B,
                                     int D, int H, int W, int K, int sign, int want_bias, float beta, void* stream) {
    CWFA_REQUIRE(a && src && out32 && workspace, CWFA_E_INVAL, "cwfa_conv3d_wgrad_f32: null pointer");
    CWFA_REQUIRE(K <= 32, CWFA_E_SHAPE, "cwfa_conv3d_wgrad_f32: K = %d hidden channels (at most 32 are built)", K);
    CWFA_REQUIRE(sign == 1 || sign == -1, CWFA_E_INVAL, "cwfa_conv3d_wgrad_f32: sign must be +1 or -1");
    int rc = check_c3("cwfa_conv3d_wgrad_f32", B, D, H, W, K);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    int nparts = 0;
    if (B > 0) {
        const int blocks = c3_wgrad_blocks(B, D, H, W);
        nparts = blocks * 4;
        if (W % 4 == 0 && cwfa_aligned16(a) && cwfa_aligned16(src))
            hipLaunchKernelGGL(conv3d_wgrad4_kernel, dim3(blocks), dim3(256), 0, st, a, src, alpha, reinterpret_cast<float*>(workspace), B,
                               D, H, W, K, sign, want_bias);
        else
            hipLaunchKernelGGL(conv3d_wgrad_kernel, dim3(blocks), dim3(256), 0, st, a, src, alpha, reinterpret_cast<float*>(workspace), B,
                               D, H, W, K, sign, want_bias);
        CWFA_LAUNCH_CHECK("cwfa_conv3d_wgrad_f32");
    }
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(16), dim3(256), 0, st, reinterpret_cast<const float*>(workspace), out32, (int64_t)1024,
                       nparts, beta);
    CWFA_LAUNCH_CHECK("cwfa_conv3d_wgrad_f32");
    return CWFA_OK;
}

extern "C" int cwfa_prelu_bwd_f32(const float* g, const float* o, const float* alpha, float* y, double* dalpha, int B, int64_t n,
                                  int64_t g_bs, int64_t o_bs, int64_t y_bs, void* stream) {
    CWFA_REQUIRE(g && o && alpha && y, CWFA_E_INVAL, "cwfa_prelu_bwd_f32: null pointer");
    CWFA_REQUIRE(B >= 0 && n >= 0 && B <= 65535, CWFA_E_SHAPE, "cwfa_prelu_bwd_f32: bad shape");
    if (B == 0 || n == 0) return CWFA_OK;
    const int64_t nb = (n + 255) / 256;
    hipLaunchKernelGGL(prelu_bwd_kernel, dim3((unsigned)(nb < 2048 ? nb : 2048), B), dim3(256), 0, (hipStream_t)stream, g, o, alpha, y, dalpha,
                       n, g_bs, o_bs, y_bs);
    CWFA_LAUNCH_CHECK("cwfa_prelu_bwd_f32");
    return CWFA_OK;
}

// =====================================================================================================================
// Backward pieces of the UNet (unet.py:72-113,161-195; train-mode BatchNorm as the LRNN runs, CWFA.py:532):
//   plane_affine  : u = x * scale[(b,)c] + shift[(b,)c] (+ add)      -- materialises a BatchNorm (x dropout mask) output
//   bn_bwd_stats  : S1[c] = sum g*m, S2[c] = sum g*m*y over (B,H,W)   -- the two reductions of BatchNorm's backward
//   bn_act_bwd    : g_pre = (A[(b,)c]*g + Bc[c] + Cc[c]*y) * PReLU'(y), dalpha += sum (...)*min(q,0)
//                   (A, Bc, Cc folded by the host from S1, S2, the batch statistics and gamma; y = PReLU(q) is the conv output)
//   maxpool2_bwd  : gradient of the 2x2 max-pool routed to the first maximum of each window, plus the skip gradient
// =====================================================================================================================
namespace {

__global__ __launch_bounds__(256) void plane_affine_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                           const float* __restrict__ shift, int aff_bs, const float* __restrict__ add,
                                                           float* __restrict__ y, int C, int64_t HW, int64_t x_bs, int64_t add_bs,
                                                           int64_t y_bs) {
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= HW) return;
    const int c = blockIdx.y, b = blockIdx.z;
    const float s = scale ? scale[b * aff_bs + c] : 1.f, t = shift ? shift[b * aff_bs + c] : 0.f;
    const float4 v = *reinterpret_cast<const float4*>(x + b * x_bs + (int64_t)c * HW + i);
    float4 o{v.x * s + t, v.y * s + t, v.z * s + t, v.w * s + t};
    if (add) {
        const float4 a = *reinterpret_cast<const float4*>(add + b * add_bs + (int64_t)c * HW + i);
        o.x += a.x; o.y += a.y; o.z += a.z; o.w += a.w;
    }
    *reinterpret_cast<float4*>(y + b * y_bs + (int64_t)c * HW + i) = o;
}

// grid (splits, C, B)
__global__ __launch_bounds__(256) void bn_bwd_stats_kernel(const float* __restrict__ g, const float* __restrict__ y,
                                                           const float* __restrict__ mask_bc, double* __restrict__ st, int C, int64_t HW,
                                                           int64_t g_bs, int64_t y_bs) {
    __shared__ double red[16];
    const int c = blockIdx.y, b = blockIdx.z;
    const float m = mask_bc ? mask_bc[b * C + c] : 1.f;
    const float* pg = g + b * g_bs + (int64_t)c * HW;
    const float* py = y + b * y_bs + (int64_t)c * HW;
    double s1 = 0.0, s2 = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += (int64_t)gridDim.x * blockDim.x) {
        const float gv = pg[i] * m;
        s1 += gv;
        s2 += (double)gv * py[i];
    }
    s1 = cwfa_block_sum(s1, red);
    s2 = cwfa_block_sum(s2, red);
    if (threadIdx.x == 0) {
        atomicAdd(&st[2 * c], s1);
        atomicAdd(&st[2 * c + 1], s2);
    }
}

// grid (splits, C, B)
__global__ __launch_bounds__(256) void bn_act_bwd_kernel(const float* __restrict__ g, const float* __restrict__ y,
                                                         const float* __restrict__ A, int a_bs, const float* __restrict__ Bc,
                                                         const float* __restrict__ Cc, const float* __restrict__ alpha_p,
                                                         float* __restrict__ out, double* __restrict__ dalpha, int C, int64_t HW,
                                                         int64_t g_bs, int64_t y_bs, int64_t o_bs) {
    __shared__ double red[16];
    const int c = blockIdx.y, b = blockIdx.z;
    const float a = A[b * a_bs + c], bc = Bc[c], cc = Cc[c];
    const float alpha = alpha_p ? *alpha_p : 1.f;
    const float* pg = g + b * g_bs + (int64_t)c * HW;
    const float* py = y + b * y_bs + (int64_t)c * HW;
    float* po = out + b * o_bs + (int64_t)c * HW;
    double da = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += (int64_t)gridDim.x * blockDim.x) {
        const float yv = py[i];
        const float gy = a * pg[i] + bc + cc * yv;
        if (alpha_p) {
            po[i] = yv > 0.f ? gy : alpha * gy;
            da += yv > 0.f ? 0.0 : (double)gy * ((double)yv / (double)alpha);
        } else {
            po[i] = gy;
        }
    }
    if (dalpha && alpha_p) {
        const double tot = cwfa_block_sum(da, red);
        if (threadIdx.x == 0) atomicAdd(dalpha, tot);
    }
}

// thread = one 2x2 window; full [B*C][H][W] (H, W even), g_pool [B*C][H/2][W/2]
__global__ __launch_bounds__(256) void maxpool2_bwd_kernel(const float* __restrict__ full, const float* __restrict__ g_pool,
                                                           const float* __restrict__ g_skip, float* __restrict__ g_full, int H, int W) {
    const int Ho = H / 2, Wo = W / 2;
    const int64_t n = (int64_t)Ho * Wo;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t bc = blockIdx.y;
    const int oy = (int)(i / Wo), ox = (int)(i % Wo);
    const float* pf = full + bc * H * W + (int64_t)(2 * oy) * W + 2 * ox;
    const float v[4] = {pf[0], pf[1], pf[W], pf[W + 1]};
    int arg = 0;
    float m = v[0];
#pragma unroll
    for (int k = 1; k < 4; ++k)
        if (v[k] > m || v[k] != v[k]) {      // first maximum in window order, NaN wins: ATen's max-pool index rule
            m = v[k];
            arg = k;
        }
    const float gp = g_pool[bc * n + i];
    const int64_t base = bc * H * W + (int64_t)(2 * oy) * W + 2 * ox;
    const int64_t offs[4] = {0, 1, W, (int64_t)W + 1};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float o = k == arg ? gp : 0.f;
        if (g_skip) o += g_skip[base + offs[k]];
        g_full[base + offs[k]] = o;
    }
}

}  // namespace

extern "C" int cwfa_plane_affine_f32(const float* x, const float* scale, const float* shift, int per_sample, const float* add, float* y,
                                     int B, int C, int64_t HW, int64_t x_bs, int64_t add_bs, int64_t y_bs, void* stream) {
    CWFA_REQUIRE(x && y, CWFA_E_INVAL, "cwfa_plane_affine_f32: null pointer");
    CWFA_REQUIRE(B >= 0 && C >= 0 && HW >= 0 && B <= 65535 && C <= 65535, CWFA_E_SHAPE, "cwfa_plane_affine_f32: bad shape");
    CWFA_REQUIRE(HW % 4 == 0 && x_bs % 4 == 0 && y_bs % 4 == 0 && (!add || add_bs % 4 == 0), CWFA_E_SHAPE,
                 "cwfa_plane_affine_f32: plane size and strides must be multiples of 4 elements");
    CWFA_REQUIRE(cwfa_aligned16(x) && cwfa_aligned16(y) && (!add || cwfa_aligned16(add)), CWFA_E_ALIGN,
                 "cwfa_plane_affine_f32: pointers must be 16-byte aligned");
    if (B == 0 || C == 0 || HW == 0) return CWFA_OK;
    hipLaunchKernelGGL(plane_affine_kernel, dim3((unsigned)((HW / 4 + 255) / 256), C, B), dim3(256), 0, (hipStream_t)stream, x, scale,
                       shift, per_sample ? C : 0, add, y, C, HW, x_bs, add_bs, y_bs);
    CWFA_LAUNCH_CHECK("cwfa_plane_affine_f32");
    return CWFA_OK;
}

extern "C" int cwfa_bn_bwd_stats_f32(const float* g, const float* y, const float* mask_bc, double* stats, int B, int C, int64_t HW,
                                     int64_t g_bs, int64_t y_bs, void* stream) {
    CWFA_REQUIRE(g && y && stats, CWFA_E_INVAL, "cwfa_bn_bwd_stats_f32: null pointer");
    CWFA_REQUIRE(B >= 0 && C >= 0 && HW >= 0 && B <= 65535 && C <= 65535, CWFA_E_SHAPE, "cwfa_bn_bwd_stats_f32: bad shape");
    if (B == 0 || C == 0 || HW == 0) return CWFA_OK;
    int64_t splits = (HW + 256 * 16 - 1) / (256 * 16);
    const int64_t cap = 2048 / ((int64_t)B * C) + 1;
    if (splits > cap) splits = cap;
    hipLaunchKernelGGL(bn_bwd_stats_kernel, dim3((unsigned)splits, C, B), dim3(256), 0, (hipStream_t)stream, g, y, mask_bc, stats, C, HW,
                       g_bs, y_bs);
    CWFA_LAUNCH_CHECK("cwfa_bn_bwd_stats_f32");
    return CWFA_OK;
}

extern "C" int cwfa_bn_act_bwd_f32(const float* g, const float* y, const float* A, int per_sample, const float* Bc, const float* Cc,
                                   const float* alpha, float* out, double* dalpha, int B, int C, int64_t HW, int64_t g_bs,
                                   int64_t y_bs, int64_t out_bs, void* stream) {
    CWFA_REQUIRE(g && y && A && Bc && Cc && out, CWFA_E_INVAL, "cwfa_bn_act_bwd_f32: null pointer");
    CWFA_REQUIRE(B >= 0 && C >= 0 && HW >= 0 && B <= 65535 && C <= 65535, CWFA_E_SHAPE, "cwfa_bn_act_bwd_f32: bad shape");
    if (B == 0 || C == 0 || HW == 0) return CWFA_OK;
    int64_t splits = (HW + 256 * 16 - 1) / (256 * 16);
    const int64_t cap = 2048 / ((int64_t)B * C) + 1;
    if (splits > cap) splits = cap;
    hipLaunchKernelGGL(bn_act_bwd_kernel, dim3((unsigned)splits, C, B), dim3(256), 0, (hipStream_t)stream, g, y, A, per_sample ? C : 0, Bc,
                       Cc, alpha, out, dalpha, C, HW, g_bs, y_bs, out_bs);
    CWFA_LAUNCH_CHECK("cwfa_bn_act_bwd_f32");
    return CWFA_OK;
}

extern "C" int cwfa_maxpool2_bwd_f32(const float* full, const float* g_pool, const float* g_skip, float* g_full, int B, int C, int H,
                                     int W, void* stream) {
    CWFA_REQUIRE(full && g_pool && g_full, CWFA_E_INVAL, "cwfa_maxpool2_bwd_f32: null pointer");
    CWFA_REQUIRE(B >= 0 && C >= 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0, CWFA_E_SHAPE,
                 "cwfa_maxpool2_bwd_f32: even H and W only (non-overlapping 2x2 windows)");
    CWFA_REQUIRE((int64_t)B * C <= 65535, CWFA_E_SHAPE, "cwfa_maxpool2_bwd_f32: B*C too large");
    if (B == 0 || C == 0) return CWFA_OK;
    const int64_t n = (int64_t)(H / 2) * (W / 2);
    hipLaunchKernelGGL(maxpool2_bwd_kernel, dim3((unsigned)((n + 255) / 256), B * C), dim3(256), 0, (hipStream_t)stream, full, g_pool,
                       g_skip, g_full, H, W);
    CWFA_LAUNCH_CHECK("cwfa_maxpool2_bwd_f32");
    return CWFA_OK;
}

// =====================================================================================================================
// Backward pieces of the LRNN's mean-volume branch (ConvNeXt networks.py:468-503, GlobalAttention :244-262, combine :552-554)
// =====================================================================================================================
namespace {

constexpr int CWFA_ATT_BWD_MAXC = 8;      // attention backward keeps C*C*4 + 2C running sums per thread (the LRNN has C = 6)

__device__ __forceinline__ float gelu_grad(float v) {      // d/dv [0.5 v (1 + erf(v / sqrt 2))]
    const float cdf = 0.5f * (1.f + erff(v * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * expf(-0.5f * v * v);
    return cdf + v * pdf;
}

// mode 0: y = GELU(p) + res (nullable);   mode 1: y = g * GELU'(p)
__global__ __launch_bounds__(256) void gelu_kernel(const float* __restrict__ p, const float* __restrict__ other, float* __restrict__ y,
                                                   int64_t n, int mode) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float v = p[i];
    if (mode == 0)
        y[i] = cwfa_gelu(v) + (other ? other[i] : 0.f);
    else
        y[i] = other[i] * gelu_grad(v);
}

// LayerNorm over (C,H,W) per sample, y = xhat * w + b:  st[2b] += sum g w, st[2b+1] += sum g w xhat      grid (splits, B)
__global__ __launch_bounds__(256) void ln_bwd_stats_kernel(const float* __restrict__ g, const float* __restrict__ v,
                                                           const float* __restrict__ w, const float* __restrict__ mean,
                                                           const float* __restrict__ invstd, double* __restrict__ st, int64_t n) {
    __shared__ double red[16];
    const int b = blockIdx.y;
    const float mu = mean[b], is = invstd[b];
    double s1 = 0.0, s2 = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float gw = g[b * n + i] * w[i];
        s1 += gw;
        s2 += (double)gw * ((v[b * n + i] - mu) * is);
    }
    s1 = cwfa_block_sum(s1, red);
    s2 = cwfa_block_sum(s2, red);
    if (threadIdx.x == 0) {
        atomicAdd(&st[2 * b], s1);
        atomicAdd(&st[2 * b + 1], s2);
    }
}

// gv[b][i] = invstd_b * (g w - S1_b/n - xhat * S2_b/n);   dw[i] += sum_b g xhat;   db[i] += sum_b g
__global__ __launch_bounds__(256) void ln_bwd_apply_kernel(const float* __restrict__ g, const float* __restrict__ v,
                                                           const float* __restrict__ w, const float* __restrict__ mean,
                                                           const float* __restrict__ invstd, const double* __restrict__ st,
                                                           float* __restrict__ gv, float* __restrict__ dw, float* __restrict__ db, int B,
                                                           int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float wi = w[i];
    float aw = 0.f, ab = 0.f;
    for (int b = 0; b < B; ++b) {
        const float is = invstd[b];
        const float xh = (v[b * n + i] - mean[b]) * is;
        const float gi = g[b * n + i];
        aw += gi * xh;
        ab += gi;
        gv[b * n + i] = is * (gi * wi - (float)(st[2 * b] / (double)n) - xh * (float)(st[2 * b + 1] / (double)n));
    }
    dw[i] += aw;
    db[i] += ab;
}

// out = x + 2 m (att - 0.5), att = sigmoid(W2 relu(W1 *3 mean + b1) + b2) along the flattened H*W sequence.  Given g = dL/dout:
// gm = 2 g (att - 0.5); parameter gradients of the attention (pgrad: [w1 C*C*3 | b1 C | w2 C*C | b2 C], doubles, accumulated).
__global__ __launch_bounds__(256) void attention_bwd_kernel(const float* __restrict__ mean, const float* __restrict__ w1,
                                                            const float* __restrict__ b1, const float* __restrict__ w2,
                                                            const float* __restrict__ b2, const float* __restrict__ m,
                                                            const float* __restrict__ g, float* __restrict__ gm,
                                                            double* __restrict__ pgrad, int C, int64_t L) {
    __shared__ double red[16];
    const int b = blockIdx.y;
    const float* pm = mean + (int64_t)b * C * L;
    constexpr int MC = CWFA_ATT_BWD_MAXC;
    float aw1[MC * MC * 3], ab1[MC], aw2[MC * MC], ab2[MC];
    for (int k = 0; k < C * C * 3; ++k) aw1[k] = 0.f;
    for (int k = 0; k < C * C; ++k) aw2[k] = 0.f;
    for (int k = 0; k < C; ++k) ab1[k] = ab2[k] = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < L; i += (int64_t)gridDim.x * blockDim.x) {
        float nb[MC][3], hid[MC], ga[MC], gh[MC];
        for (int c = 0; c < C; ++c) {
            nb[c][0] = i > 0 ? pm[(int64_t)c * L + i - 1] : 0.f;
            nb[c][1] = pm[(int64_t)c * L + i];
            nb[c][2] = i + 1 < L ? pm[(int64_t)c * L + i + 1] : 0.f;
        }
        for (int o = 0; o < C; ++o) {
            float h = b1[o];
            for (int c = 0; c < C; ++c) {
                const float* ww = w1 + ((int64_t)o * C + c) * 3;
                h = fmaf(ww[2], nb[c][2], fmaf(ww[1], nb[c][1], fmaf(ww[0], nb[c][0], h)));
            }
            hid[o] = h > 0.f ? h : 0.f;
        }
        for (int o = 0; o < C; ++o) {
            float a = b2[o];
            for (int c = 0; c < C; ++c) a = fmaf(w2[o * C + c], hid[c], a);
            const float att = 1.f / (1.f + expf(-a));
            const int64_t idx = ((int64_t)b * C + o) * L + i;
            const float gv = g[idx], mv = m[idx];
            gm[idx] = 2.f * gv * (att - 0.5f);
            ga[o] = 2.f * gv * mv * att * (1.f - att);
            ab2[o] += ga[o];
            for (int c = 0; c < C; ++c) aw2[o * C + c] += ga[o] * hid[c];
        }
        for (int c = 0; c < C; ++c) {
            float t = 0.f;
            for (int o = 0; o < C; ++o) t = fmaf(w2[o * C + c], ga[o], t);
            gh[c] = hid[c] > 0.f ? t : 0.f;
            ab1[c] += gh[c];
        }
        for (int o = 0; o < C; ++o)
            for (int c = 0; c < C; ++c)
                for (int k = 0; k < 3; ++k) aw1[(o * C + c) * 3 + k] += gh[o] * nb[c][k];
    }
    const int n1 = C * C * 3, n2 = C * C;
    for (int k = 0; k < n1 + C + n2 + C; ++k) {
        const float val = k < n1 ? aw1[k] : k < n1 + C ? ab1[k - n1] : k < n1 + C + n2 ? aw2[k - n1 - C] : ab2[k - n1 - C - n2];
        const double tot = cwfa_block_sum((double)val, red);
        if (threadIdx.x == 0) atomicAdd(&pgrad[k], tot);
    }
}

}  // namespace

extern "C" int cwfa_gelu_f32(const float* p, const float* other, float* y, int64_t n, int mode, void* stream) {
    CWFA_REQUIRE(p && y && (mode == 0 || other), CWFA_E_INVAL, "cwfa_gelu_f32: null pointer");
    CWFA_REQUIRE(n >= 0 && (mode == 0 || mode == 1), CWFA_E_INVAL, "cwfa_gelu_f32: bad argument");
    if (n == 0) return CWFA_OK;
    hipLaunchKernelGGL(gelu_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p, other, y, n, mode);
    CWFA_LAUNCH_CHECK("cwfa_gelu_f32");
    return CWFA_OK;
}

extern "C" int cwfa_layernorm_bwd_f32(const float* g, const float* v, const float* w, const float* mean, const float* invstd,
                                      double* stats, float* gv, float* dw, float* db, int B, int64_t n, void* stream) {
    CWFA_REQUIRE(g && v && w && mean && invstd && stats && gv && dw && db, CWFA_E_INVAL, "cwfa_layernorm_bwd_f32: null pointer");
    CWFA_REQUIRE(B >= 0 && n >= 0 && B <= 65535, CWFA_E_SHAPE, "cwfa_layernorm_bwd_f32: bad shape");
    if (B == 0 || n == 0) return CWFA_OK;
    int blocks = (int)((n + 256 * 16 - 1) / (256 * 16));
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(ln_bwd_stats_kernel, dim3(blocks, B), dim3(256), 0, (hipStream_t)stream, g, v, w, mean, invstd, stats, n);
    CWFA_LAUNCH_CHECK("cwfa_layernorm_bwd_f32");
    hipLaunchKernelGGL(ln_bwd_apply_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, g, v, w, mean, invstd,
                       stats, gv, dw, db, B, n);
    CWFA_LAUNCH_CHECK("cwfa_layernorm_bwd_f32");
    return CWFA_OK;
}

extern "C" int cwfa_attention_bwd_f32(const float* mean, const float* w1, const float* b1, const float* w2, const float* b2,
                                      const float* m, const float* g, float* gm, double* pgrad, int B, int C, int64_t HW, void* stream) {
    CWFA_REQUIRE(mean && w1 && b1 && w2 && b2 && m && g && gm && pgrad, CWFA_E_INVAL, "cwfa_attention_bwd_f32: null pointer");
    CWFA_REQUIRE(C > 0 && C <= CWFA_ATT_BWD_MAXC, CWFA_E_SHAPE, "cwfa_attention_bwd_f32: C=%d not in 1..%d", C, CWFA_ATT_BWD_MAXC);
    CWFA_REQUIRE(B >= 0 && HW >= 0 && B <= 65535, CWFA_E_SHAPE, "cwfa_attention_bwd_f32: bad shape");
    if (B == 0 || HW == 0) return CWFA_OK;
    int blocks = (int)((HW + 255) / 256);
    if (blocks > 512) blocks = 512;
    hipLaunchKernelGGL(attention_bwd_kernel, dim3(blocks, B), dim3(256), 0, (hipStream_t)stream, mean, w1, b1, w2, b2, m, g, gm, pgrad, C,
                       HW);
    CWFA_LAUNCH_CHECK("cwfa_attention_bwd_f32");
    return CWFA_OK;
}
