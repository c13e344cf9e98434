// Implicit-GEMM 2-D convolution on the fp32 matrix cores of gfx950 (v_mfma_f32_32x32x2_f32), no im2col.
//
// GEMM view (per image):  Y[co][pix] = sum_{tap, ci} Wt[tap][ci][co] * X[ci][pix + tap]
//   M = output channels  -> MFMA rows   (A operand = weights, lane = (co & 31, k = lane >> 5))
//   N = pixels along W   -> MFMA cols   (B operand = input,   lane = (k = lane >> 5, pix & 31))
//   K = (tap, ci) pairs of input channels per MFMA
// so that an accumulator register holds 32 CONSECUTIVE PIXELS of one output channel per half-wave: NCHW stores are
// 128-byte coalesced and a following 1x1 conv can take the accumulators as its B operand with no lane movement
// (subnet_layer_kernel below does exactly that).
//
// Block = WM x WN waves; each wave owns MT x NT sub-tiles of 32 channels x (1 row x 32 pixels).
// Per K-chunk of CK input channels the block stages the haloed input tile [CK][TR+ks-1][32+ks-1] and the weight
// panel [ks*ks][CK][CT] in LDS (global loads of chunk i+1 are issued into registers before the MFMAs of chunk i),
// every LDS operand read is a conflict-free ds_read_b32 with a compile-time immediate offset.
//
// Epilogues are compile-time specialised for the combinations the model uses (a runtime-switched epilogue unrolled
// over 64 accumulators was 19k instructions / 117 KB of code and cost ~5 K-chunks of time per block); anything else
// takes the generic (runtime) epilogue.
#include "conv_internal.h"

#include <string.h>
#include <type_traits>
#include <utility>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// tuning knobs (defaults = shipped configuration; tools/conv_tune.py builds variants with -D overrides)
#ifndef CWFA_MINW
#define CWFA_MINW 1
#endif
#ifndef CWFA_PREFETCH
#define CWFA_PREFETCH 1
#endif
#ifndef CWFA_WN64
#define CWFA_WN64 8
#endif
#ifndef CWFA_WM128
#define CWFA_WM128 2
#endif
#ifndef CWFA_WN128
#define CWFA_WN128 4
#endif
#ifndef CWFA_CK3
#define CWFA_CK3 8
#endif
#ifndef CWFA_CK1
#define CWFA_CK1 16
#endif

namespace {

template <int KS_, int CK_, int MT_, int NT_, int WM_, int WN_, int XV_ = 1>
struct Cfg {
    static constexpr int KS = KS_, CK = CK_, MT = MT_, NT = NT_, WM = WM_, WN = WN_;
    static constexpr int XV = XV_;                   // floats per staged input element (4: 1x1 convs on 16-byte aligned rows)
    static_assert(XV_ == 1 || (XV_ == 4 && KS_ == 1), "vector staging needs halo-free tiles");
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
    static constexpr int XPT = (XS / XV + NTHREADS - 1) / NTHREADS;
    static constexpr int WPT = (WS / 4 + NTHREADS - 1) / NTHREADS;
    static constexpr int BUF = XS_PAD + WS;          // floats per LDS buffer (two buffers, see conv_mainloop)
    static constexpr int LDS_BYTES = 2 * BUF * 4;
};

struct ConvParams {
    const float* x;
    const float* wp;
    float* y;
    int B, Cin, H, W, Cout, nchunks, tiles_x, tiles_y;
    int64_t x_bs, y_bs;
    cwfa_conv_opts o;
    // fused sub-network layer only
    const float* w1x1;      // [32 k-steps][2][64 lanes] panel of the 1x1 conv
    const float* b1x1;
    int products;           // split-bf16 kernels: 6 = fp32-accurate (three pieces per operand), 1 = plain bf16 operands
};

struct Tile {
    int wm, wn, kh, l31, ct, b, row0, col0;
};

template <class C>
__device__ __forceinline__ Tile make_tile(const ConvParams& p) {
    Tile t;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    t.wm = wave / C::WN;
    t.wn = wave % C::WN;
    t.kh = lane >> 5;
    t.l31 = lane & 31;
    const int tx = blockIdx.x % p.tiles_x, ty = blockIdx.x / p.tiles_x;
    t.ct = blockIdx.y;
    t.b = blockIdx.z;
    t.row0 = ty * C::TR;
    t.col0 = tx * C::TC;
    return t;
}

// ------------------------------------------------------------------------------------------------ main loop
// PRO = the load-side prologue (per-channel affine and/or added tensor) is compiled in; kernels without it do not pay
// its staging registers.
//
// Schedule (same reasoning as conv_wino.hip, measured there): the fp32 MFMA and the vector ALU of a SIMD do not
// co-execute, and a wave with an MFMA ready starves the vector instructions of the other waves on its SIMD, so staging
// cannot be hidden behind "another wave's" MFMAs.  Every wave runs ONE stream: the k-steps of chunk c, and between them
// the staging ITEMS of the block's next tiles -- store item k of chunk c+1 into the other LDS buffer (its loads were
// issued a chunk ago), then issue its loads for chunk c+2 into the registers just freed.  Two LDS buffers, one barrier
// per chunk, placed DEPTH k-steps before the end of the chunk so that the first operand reads of chunk c+1 are covered
// by the last MFMAs of chunk c.  Global reads go through buffer descriptors: the per-lane byte offset is computed once,
// the per-chunk part is the scalar soffset, and padding (outside the image, channels >= Cin) is an out-of-range offset
// that the hardware range check returns as 0.0.
template <int K>
using sc_int = std::integral_constant<int, K>;
template <class F, int... S>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, S...>) {
    (f(sc_int<S>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

template <class C, bool PRO>
__device__ __forceinline__ void conv_mainloop(const ConvParams& p, const Tile& t, float* smem, f32x16 (&acc)[C::MT][C::NT]) {
    const int tid = threadIdx.x;
    const int64_t HW = (int64_t)p.H * p.W;
    constexpr unsigned OOB = 0x80000000u;
    constexpr int NP = PRO ? C::XPT : 1;

    // per-thread staging map for the input tile (identical for every chunk); an element is XV consecutive pixels
    typedef float xvec __attribute__((ext_vector_type(C::XV == 1 ? 2 : C::XV)));      // (XV == 1 uses lane 0 only)
    unsigned voff[C::XPT], cl4[NP];
    unsigned long long okm[NP];                       // lane masks: element is inside the image (load-side affine only)
#pragma unroll
    for (int i = 0; i < C::XPT; ++i) {
        const int e = (tid + i * C::NTHREADS) * C::XV;
        const int c = e / (C::XR * C::XC), rem = e % (C::XR * C::XC);
        const int r = rem / C::XC, cc = rem % C::XC;
        const int gr = t.row0 + r - C::PAD, gc = t.col0 + cc - C::PAD;
        const bool ok = e < C::XS && gr >= 0 && gr < p.H && gc >= 0 && gc < p.W;       // XV == 4: W % 4 == 0 (dispatch)
        voff[i] = ok ? (unsigned)((c * HW + (int64_t)gr * p.W + gc) * 4) : OOB;
        if constexpr (PRO) {
            cl4[i] = ok ? c * 4 : OOB;
            okm[i] = __builtin_amdgcn_ballot_w64(ok);
        }
    }
    const bool has_aff = PRO && p.o.in_scale != nullptr, has_add = PRO && p.o.in_add != nullptr;
    // two-source input (cwfa_conv_opts.in_cat: the channel concatenation of a coupling sub-network's input, never materialised):
    // chunks [0, n1) come from x (its channels past in_cat_c1 are out of range = the zero columns the bank was packed with),
    // chunks [n1, nchunks) from in_cat
    const bool cat = !PRO && p.o.in_cat != nullptr;
    const int n1 = cat ? p.o.in_cat_from / C::CK : p.nchunks;
    const int xbytes = (int)((int64_t)(cat ? p.o.in_cat_c1 : p.Cin) * HW * 4);
    const auto rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x + (int64_t)t.b * p.x_bs), 0, xbytes, 0x00020000);
    const auto rx2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(cat ? p.o.in_cat + (int64_t)t.b * p.o.in_cat_bs : p.x), 0,
                                                       cat ? (int)((int64_t)(p.Cin - p.o.in_cat_from) * HW * 4) : 0, 0x00020000);
    const auto ra = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(has_add ? p.o.in_add + (int64_t)t.b * p.o.in_add_bs : p.x), 0, has_add ? xbytes : 0, 0x00020000);
    const auto rsc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(has_aff ? p.o.in_scale + (int64_t)t.b * p.o.in_affine_bs : p.x), 0, has_aff ? p.Cin * 4 : 0, 0x00020000);
    const auto rsh = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(has_aff ? p.o.in_shift + (int64_t)t.b * p.o.in_affine_bs : p.x), 0, has_aff ? p.Cin * 4 : 0, 0x00020000);
    const auto rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wp + (int64_t)t.ct * p.nchunks * C::WS), 0,
                                                      p.nchunks * C::WS * 4, 0x00020000);
    const int chunk_bytes = (int)(C::CK * HW * 4);
    auto ldf = [](decltype(rx) r, unsigned vo, int so) { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, vo, so, 0)); };
    auto ldx = [&](decltype(rx) r, unsigned vo, int so) {
        xvec v;
        if constexpr (C::XV == 4) v = __builtin_bit_cast(xvec, __builtin_amdgcn_raw_buffer_load_b128(r, vo, so, 0));
        else v[0] = ldf(r, vo, so);
        return v;
    };

    xvec xr[C::XPT], ar[NP];
    float sr[NP], hr[NP];
    f32x4 wr[C::WPT];
    auto load_x = [&](int i, int chunk) {
        if (chunk < n1) xr[i] = ldx(rx, voff[i], chunk * chunk_bytes);              // (uniform)
        else xr[i] = ldx(rx2, voff[i], (chunk - n1) * chunk_bytes);
        if constexpr (PRO) {
            if (has_aff) {
                sr[i] = ldf(rsc, cl4[i], chunk * C::CK * 4);
                hr[i] = ldf(rsh, cl4[i], chunk * C::CK * 4);
            }
            if (has_add) ar[i] = ldx(ra, voff[i], chunk * chunk_bytes);
        }
    };
    auto store_x = [&](int i, int buf) {
        xvec v = xr[i];
        if constexpr (PRO) {
#pragma unroll
            for (int j = 0; j < C::XV; ++j) {
                float u = v[j];
                if (has_aff) {
                    u = u * sr[i] + hr[i];                              // zero padding is inserted AFTER the affine
                    asm("v_cndmask_b32 %0, 0, %1, %2" : "=v"(u) : "v"(u), "s"(okm[i]));
                }
                if (has_add) u += ar[i][j];
                v[j] = u;
            }
        }
        const int e = (tid + i * C::NTHREADS) * C::XV;
        if ((C::XS / C::XV) % C::NTHREADS == 0 || e < C::XS) {
            if constexpr (C::XV == 4) *reinterpret_cast<xvec*>(smem + buf * C::BUF + e) = v;
            else smem[buf * C::BUF + e] = v[0];
        }
    };
    // the weight panel of a chunk is WS/4 16-byte pieces; the last piece index of a thread may fall off the end (range check)
    auto load_w = [&](int i, int chunk) {
        wr[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                              rw, (unsigned)(tid + i * C::NTHREADS) < (unsigned)(C::WS / 4) ? (tid + i * C::NTHREADS) * 16u : OOB,
                                              chunk * C::WS * 4, 0));
    };
    auto store_w = [&](int i, int buf) {
        const int e = tid + i * C::NTHREADS;
        if ((C::WS / 4) % C::NTHREADS == 0 || e < C::WS / 4) reinterpret_cast<f32x4*>(smem + buf * C::BUF + C::XS_PAD)[e] = wr[i];
    };
    constexpr int NITEM = C::XPT + C::WPT;
    auto load_item = [&](auto kc, int chunk) {
        constexpr int k = decltype(kc)::value;
        if constexpr (k < C::XPT) load_x(k, chunk);
        else load_w(k - C::XPT, chunk);
    };
    auto store_item = [&](auto kc, int buf) {
        constexpr int k = decltype(kc)::value;
        if constexpr (k < C::XPT) store_x(k, buf);
        else store_w(k - C::XPT, buf);
    };

#pragma unroll
    for (int m = 0; m < C::MT; ++m)
#pragma unroll
        for (int n = 0; n < C::NT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

    const float* wlane0 = smem + C::XS_PAD + t.kh * C::CT + (t.wm * C::MT) * 32 + t.l31;
    const float* xlane0 = smem + t.kh * (C::XR * C::XC) + (t.wn * C::NT) * C::XC + t.l31;

    // operand look-ahead in k-steps (each MT*NT MFMAs), chosen so that the register ring divides the chunk
    constexpr int NSTEP = C::KS * C::KS * (C::CK / 2), DEPTH = C::KS == 1 ? 3 : C::KS == 3 ? 2 : 1, RING = DEPTH + 1;
    static_assert(NSTEP % RING == 0, "operand ring must divide the k-steps of a chunk");
    constexpr int NSLOT = NSTEP - DEPTH;              // k-steps before the barrier: every item is stored in one of them
    static_assert(NSTEP > DEPTH, "chunk shorter than the operand pipeline");
    float aq[RING][C::MT], bq[RING][C::NT];
    auto ld = [&](int buf, int s, int slot) {
        const int tap = s / (C::CK / 2), kk = s % (C::CK / 2), dy = tap / C::KS, dx = tap % C::KS;
#pragma unroll
        for (int m = 0; m < C::MT; ++m) aq[slot][m] = (wlane0 + buf * C::BUF)[(tap * C::CK + 2 * kk) * C::CT + m * 32];
#pragma unroll
        for (int n = 0; n < C::NT; ++n) bq[slot][n] = (xlane0 + buf * C::BUF)[(2 * kk) * (C::XR * C::XC) + (n + dy) * C::XC + dx];
    };
    // MORE / PF (is there a chunk c+1 to store, a chunk c+2 to load) are compile-time: the steady-state body carries no
    // branches; the last two chunks run their own copies
    auto chunk_body = [&](int cur, int chunk, auto morec, auto pfc) {
        constexpr bool more = decltype(morec)::value, pf = decltype(pfc)::value;
        constexpr int ring0 = 0;
        static_for<NSTEP>([&](auto sc) {
            constexpr int s = decltype(sc)::value;
            if constexpr (s == NSLOT && more) __syncthreads();
            if constexpr (s + DEPTH < NSTEP) {
                ld(cur, s + DEPTH, (ring0 + s + DEPTH) % RING);
            } else if constexpr (more) {
                ld(cur ^ 1, s + DEPTH - NSTEP, (ring0 + s + DEPTH) % RING);
            }
#pragma unroll
            for (int m = 0; m < C::MT; ++m)
#pragma unroll
                for (int n = 0; n < C::NT; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[(ring0 + s) % RING][m], bq[(ring0 + s) % RING][n], acc[m][n], 0, 0, 0);
            if constexpr (s < NSLOT) {
                // items [k0, k1) are staged after this step: NITEM items spread evenly over the NSLOT steps
                constexpr int k0 = s * NITEM / NSLOT, k1 = (s + 1) * NITEM / NSLOT;
                static_for<k1 - k0>([&](auto jc) {
                    constexpr int k = k0 + decltype(jc)::value;
                    if constexpr (more) store_item(sc_int<k>{}, cur ^ 1);
                    if constexpr (pf) load_item(sc_int<k>{}, chunk + 2);
                });
            }
            __builtin_amdgcn_sched_barrier(0);
        });
    };

    static_for<NITEM>([&](auto kc) { load_item(kc, 0); });
    static_for<NITEM>([&](auto kc) {
        store_item(kc, 0);
        if (1 < p.nchunks) load_item(kc, 1);
    });
    __syncthreads();
#pragma unroll
    for (int s = 0; s < DEPTH; ++s) ld(0, s, s);
    typedef std::true_type T;
    typedef std::false_type F;
    int chunk = 0;
    for (; chunk + 2 < p.nchunks; ++chunk) chunk_body(chunk & 1, chunk, T{}, T{});
    if (chunk + 1 < p.nchunks) {
        chunk_body(chunk & 1, chunk, T{}, F{});
        ++chunk;
    }
    chunk_body(chunk & 1, chunk, F{}, F{});
}

// scalar base + 32-bit unsigned BYTE offset: lowers to the saddr + voffset form (one VGPR per address instead of two)
// (the empty asm pins the offset in ONE VGPR at the point of use: without it hipcc forms all 64-bit addresses of an
//  unrolled epilogue up front -- two VGPRs each -- and spills)
__device__ __forceinline__ float ldg_off(const float* base, unsigned byte_off) {
    asm volatile("" : "+v"(byte_off));
    return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_off);
}
__device__ __forceinline__ void stg_off(float* base, unsigned byte_off, float v) {
    asm volatile("" : "+v"(byte_off));
    *reinterpret_cast<float*>(reinterpret_cast<char*>(base) + byte_off) = v;
}

// channel held by accumulator register r of a 32x32 tile in lane half kh (C/D layout of the 32x32 MFMAs)
__device__ __forceinline__ int acc_row(int r, int kh) { return (r & 3) + 8 * (r >> 2) + 4 * kh; }

// ------------------------------------------------------------------------------------------------ epilogues
enum { EPI_GENERIC = 0, EPI_NONE, EPI_ELU, EPI_RES_ELU, EPI_PRELU, EPI_RES_PRELU, EPI_GELU_RES, EPI_UP, EPI_NONE_BLK8, EPI_COUNT };

template <int EPI>
struct EpiTraits {
    static constexpr int ACT1 = EPI == EPI_ELU ? CWFA_ACT_ELU : EPI == EPI_PRELU ? CWFA_ACT_PRELU
                                : EPI == EPI_GELU_RES ? CWFA_ACT_GELU : CWFA_ACT_NONE;
    static constexpr bool RES = EPI == EPI_RES_ELU || EPI == EPI_RES_PRELU || EPI == EPI_GELU_RES;
    static constexpr int ACT2 = EPI == EPI_RES_ELU ? CWFA_ACT_ELU : EPI == EPI_RES_PRELU ? CWFA_ACT_PRELU : CWFA_ACT_NONE;
    static constexpr bool UP = EPI == EPI_UP;
};

template <int ACT>
__device__ __forceinline__ float act_ct(float v, float alpha) {
    if constexpr (ACT == CWFA_ACT_ELU) return cwfa_elu(v);
    if constexpr (ACT == CWFA_ACT_PRELU) return v > 0.f ? v : alpha * v;
    if constexpr (ACT == CWFA_ACT_GELU) return cwfa_gelu(v);
    if constexpr (ACT == CWFA_ACT_RELU) return v > 0.f ? v : 0.f;
    return v;
}

template <class C, int EPI>
__device__ __forceinline__ void epilogue(const ConvParams& p, const Tile& t, f32x16 (&acc)[C::MT][C::NT]) {
    typedef EpiTraits<EPI> E;
    const int64_t HW = (int64_t)p.H * p.W;
    const int col = t.col0 + t.l31;
    float* yb = p.y + (int64_t)t.b * p.y_bs;
    if constexpr (EPI == EPI_GENERIC) {
        // runtime-switched path: the accumulators go through LDS so that ONE copy of the code serves all 64 of them
        const float alpha = (p.o.prelu_alpha && (p.o.act == CWFA_ACT_PRELU || p.o.act2 == CWFA_ACT_PRELU)) ? *p.o.prelu_alpha : 0.f;
        const float* rb = p.o.residual ? p.o.residual + (int64_t)t.b * p.o.res_bs : nullptr;
        extern __shared__ __attribute__((aligned(16))) float smem[];
        __syncthreads();                                   // main loop done with the LDS tiles
        float* mine = smem + (threadIdx.x >> 6) * 1024 + (threadIdx.x & 63);     // 16 floats x 64 lanes per wave
        for (int m = 0; m < C::MT; ++m)
            for (int n = 0; n < C::NT; ++n) {
#pragma unroll
                for (int mm = 0; mm < C::MT; ++mm)
#pragma unroll
                    for (int nn = 0; nn < C::NT; ++nn)
                        if (mm == m && nn == n) {
#pragma unroll
                            for (int r = 0; r < 16; ++r) mine[r * 64] = acc[mm][nn][r];
                        }
                const int row = t.row0 + t.wn * C::NT + n;
                for (int r = 0; r < 16; ++r) {
                    const int co = t.ct * C::CT + (t.wm * C::MT + m) * 32 + acc_row(r, t.kh);
                    if (co >= p.Cout || row >= p.H || col >= p.W) continue;
                    const int cb = p.o.upshuffle2 ? co >> 2 : co;
                    int64_t o;
                    if (p.o.upshuffle2) {
                        const int q = co & 3;
                        o = (int64_t)cb * (4 * HW) + (int64_t)(2 * row + (q >> 1)) * (2 * p.W) + 2 * col + (q & 1);
                    } else {
                        o = (int64_t)co * HW + (int64_t)row * p.W + col;
                    }
                    float v = cwfa_act(mine[r * 64] + (p.o.bias ? p.o.bias[cb] : 0.f), p.o.act, alpha);
                    if (rb) v += rb[o];
                    yb[o] = cwfa_act(v, p.o.act2, alpha);
                }
            }
    } else {
        float alpha = 0.f;
        if constexpr (E::ACT1 == CWFA_ACT_PRELU || E::ACT2 == CWFA_ACT_PRELU) alpha = *p.o.prelu_alpha;
        const float* rb = nullptr;
        if constexpr (E::RES) rb = p.o.residual + (int64_t)t.b * p.o.res_bs;
        if constexpr (E::UP) {
            // ConvTranspose2d(k2,s2) as a 1x1 conv whose packed output channel is c*4 + (dy*2+dx): registers 4g..4g+3 of
            // a lane are the 2x2 output patch of ONE channel at this lane's input pixel -> two 8-byte stores per channel,
            // 256 B contiguous per half-wave and row (the lane-strided 4-byte stores of a naive shuffle were 76 TFLOP/s).
            const int W2 = 2 * p.W;
            const bool al = ((W2 | (int)(p.y_bs & 1)) & 1) == 0 && ((reinterpret_cast<uintptr_t>(p.y) & 7) == 0);
#pragma unroll
            for (int m = 0; m < C::MT; ++m)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int co = t.ct * C::CT + (t.wm * C::MT + m) * 32 + 8 * g + 4 * t.kh;      // multiple of 4
                    if (co >= p.Cout) continue;
                    const int c = co >> 2;
                    const float bias = p.o.bias ? p.o.bias[c] : 0.f;
#pragma unroll
                    for (int n = 0; n < C::NT; ++n) {
                        const int row = t.row0 + t.wn * C::NT + n;
                        if (row >= p.H || col >= p.W) continue;
                        float* o = yb + (int64_t)c * (4 * HW) + (int64_t)(2 * row) * W2 + 2 * col;
                        const float v0 = acc[m][n][4 * g] + bias, v1 = acc[m][n][4 * g + 1] + bias;
                        const float v2 = acc[m][n][4 * g + 2] + bias, v3 = acc[m][n][4 * g + 3] + bias;
                        if (al) {
                            *reinterpret_cast<float2*>(o) = make_float2(v0, v1);
                            *reinterpret_cast<float2*>(o + W2) = make_float2(v2, v3);
                        } else {
                            o[0] = v0; o[1] = v1; o[W2] = v2; o[W2 + 1] = v3;
                        }
                    }
                }
        } else if constexpr (EPI == EPI_NONE_BLK8) {
            // bias only, output CHANNEL-BLOCKED [Cout/8][H][W][8] (cwfa_conv_opts.out_blocked8): registers 4q .. 4q+3 of a lane
            // are the channels 8q + 4 kh + {0..3} of its pixel = 16 contiguous bytes of the pixel's entry in block 4 m' + q
#pragma unroll
            for (int m = 0; m < C::MT; ++m)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int co = t.ct * C::CT + (t.wm * C::MT + m) * 32 + 8 * q + 4 * t.kh;       // multiple of 4; Cout % 8 == 0
                    if (co >= p.Cout) continue;
                    const f32x4 b4 = p.o.bias ? *reinterpret_cast<const f32x4*>(p.o.bias + co) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int n = 0; n < C::NT; ++n) {
                        const int row = t.row0 + t.wn * C::NT + n;
                        if (row >= p.H || col >= p.W) continue;
                        f32x4 v;
#pragma unroll
                        for (int i = 0; i < 4; ++i) v[i] = acc[m][n][4 * q + i] + b4[i];
                        *reinterpret_cast<f32x4*>(yb + ((int64_t)(co >> 3) * HW + (int64_t)row * p.W + col) * 8 + (co & 4)) = v;
                    }
                }
        } else {
            // biases of this lane's channels in one batch of unconditional loads (a per-channel "load or 0" branch costs a
            // vmcnt(0) round trip per channel)
            float bias[C::MT][16];
#pragma unroll
            for (int m = 0; m < C::MT; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    int co = t.ct * C::CT + (t.wm * C::MT + m) * 32 + acc_row(r, t.kh);
                    co = co < p.Cout ? co : p.Cout - 1;
                    bias[m][r] = p.o.bias ? p.o.bias[co] : 0.f;
                }
#pragma unroll
            for (int m = 0; m < C::MT; ++m) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = t.ct * C::CT + (t.wm * C::MT + m) * 32 + acc_row(r, t.kh);
                    if (co >= p.Cout) continue;
#pragma unroll
                    for (int n = 0; n < C::NT; ++n) {
                        const int row = t.row0 + t.wn * C::NT + n;
                        if (row >= p.H || col >= p.W) continue;
                        const int64_t o = (int64_t)co * HW + (int64_t)row * p.W + col;
                        float v = act_ct<E::ACT1>(acc[m][n][r] + bias[m][r], alpha);
                        if constexpr (E::RES) v += rb[o];
                        yb[o] = act_ct<E::ACT2>(v, alpha);
                    }
                }
            }
        }
    }
}

template <class C, int EPI, bool PRO>
__global__ __launch_bounds__(C::NTHREADS, CWFA_MINW) void conv2d_mfma_kernel(ConvParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const Tile t = make_tile<C>(p);
    f32x16 acc[C::MT][C::NT];
    conv_mainloop<C, PRO>(p, t, smem, acc);
    epilogue<C, EPI>(p, t, acc);
}

// ------------------------------------------------------------------------------------------------ fused sub-network layer
// One residual layer of wavelet_flow_subnetwork (networks.py:624-631,660-665) in ONE launch, C = 64 channels:
//     y = ELU( W1x1 . ELU( conv3x3(x) + b3 ) + b1 + x )
// The 3x3 accumulators [64 ch][64 px] of a wave are, register by register, the B operand of the 1x1 GEMM
// (k-pair of register r = channels {32m + row(r), +4} in the two lane halves): no LDS round trip, no lane movement;
// the A operand is a pre-permuted [32 k-steps][2 cout sub-tiles][64 lanes] image of the 1x1 weights staged in the LDS
// region the 3x3 weight panel occupied.  The residual x (+ bias) is the initial value of the 1x1 accumulators.
typedef Cfg<3, 8, 2, 2, 1, 8> CL;       // 64 ch x 16 rows x 32 cols, 512 threads

__global__ __launch_bounds__(CL::NTHREADS, 1) void subnet_layer_kernel(ConvParams p) {
    typedef CL C;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ws = smem + C::XS_PAD;
    const Tile t = make_tile<C>(p);
    f32x16 acc[2][2];
    conv_mainloop<C, false>(p, t, smem, acc);

    const int64_t HW = (int64_t)p.H * p.W;
    const int col = t.col0 + t.l31;
    const bool col_ok = col < p.W;
    const float* xb = p.x + (int64_t)t.b * p.x_bs;
    float* yb = p.y + (int64_t)t.b * p.y_bs;

    // 32-bit UNSIGNED element offsets from the (scalar) batch base: hipcc then uses the saddr + voffset addressing
    // form; 64-bit per-element addresses cost two VGPRs each and spilled this epilogue.
    unsigned oo[2];
    bool ok[2];
    const unsigned HW4 = (unsigned)HW * 4u;                    // bytes per channel plane (Cin*H*W*4 < 2^32 checked)
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const int row = t.row0 + t.wn * 2 + n;
        ok[n] = col_ok && row < p.H;
        oo[n] = (ok[n] ? (unsigned)(row * p.W + col) * 4u : 0u) + (unsigned)t.kh * 4u * HW4;   // + lane-half part
    }
    // The 1x1 GEMM runs in four quarter passes q = (mo, n): 32 chained MFMAs into ONE 32x32 accumulator each
    // (dependent latency == issue interval for v_mfma_f32_32x32x2_f32, so a single chain keeps the pipe full).
    // VALU work is slotted under the MFMAs: the ELU of the hidden values in the two mo=0 passes, and bias + residual +
    // ELU + store of quarter q-1 in pass q.  Live registers: acc 64 + 2 quarters x (16 y + 16 residual) -- no spills,
    // which unbounded hoisting of two half passes (64 y + 64 residual) did produce.
    auto load_res = [&](int mo, int n, f32x16& res) {       // residual x + 1x1 bias, accumulator layout, clamped address
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            // channel = (mo*32 + rc) + 4*kh with rc compile-time: scalar K*HW4 + one per-lane base (oo already has 4*kh)
            const unsigned K = (unsigned)(mo * 32 + acc_row(r, 0));
            res[r] = ldg_off(xb, K * HW4 + oo[n]) + ldg_off(p.b1x1, K * 4u + (unsigned)t.kh * 16u);
        }
    };
    f32x16 yq[4], rq[4];
    f32x16 b3v[2];                                             // 3x3 bias of this lane's 32 hidden channels
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) b3v[m][r] = ldg_off(p.o.bias, (unsigned)(m * 32 + acc_row(r, 0)) * 4u + (unsigned)t.kh * 16u);

    // stage the 1x1 panel (4096 floats) where the 3x3 weight panel was
    __syncthreads();
    {
        const f32x4* src = reinterpret_cast<const f32x4*>(p.w1x1);
        f32x4* dst = reinterpret_cast<f32x4*>(Ws);
        for (int e = threadIdx.x; e < 1024; e += C::NTHREADS) dst[e] = src[e];
    }
    __syncthreads();
    const float* wl = Ws + (threadIdx.x & 63);

    // Every k-step is fenced for the instruction scheduler (hipcc otherwise hoists all loads of all passes to the top
    // and spills); the hardware still overlaps: the MFMA of step j executes while the VALU work of step j+1 issues.
    // The A operand of step j+1 is read from LDS during step j.
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int mo = q >> 1, n = q & 1;
        load_res(mo, n, rq[q]);                                    // consumed during pass q+1 (or the tail)
        float a_next = wl[(0 * 2 + mo) * 64];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int j = m * 16 + r;
                const float a = a_next;
                if (j + 1 < 32) a_next = wl[((j + 1) * 2 + mo) * 64];
                if (mo == 0) acc[m][n][r] = cwfa_elu(acc[m][n][r] + b3v[m][r]);
                if (j == 0) {       // first k-step: literal zero accumulator input (no zero-initialised registers)
                    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    yq[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, acc[m][n][r], zero, 0, 0, 0);
                } else {
                    yq[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, acc[m][n][r], yq[q], 0, 0, 0);
                }
                if (q > 0 && (j & 1)) {                                 // element j/2 of the previous quarter
                    const int pq = q > 0 ? q - 1 : 0, pmo = pq >> 1, pn = pq & 1, rr = j >> 1;
                    const unsigned K = (unsigned)(pmo * 32 + acc_row(rr, 0));
                    if (ok[pn]) stg_off(yb, K * HW4 + oo[pn], cwfa_elu(yq[pq][rr] + rq[pq][rr]));
                }
                __builtin_amdgcn_sched_barrier(0);
            }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {                                      // tail: the last quarter (mo = 1, n = 1)
        const unsigned K = (unsigned)(32 + acc_row(r, 0));
        if (ok[1]) stg_off(yb, K * HW4 + oo[1], cwfa_elu(yq[3][r] + rq[3][r]));
    }
}

// panel[(m*16 + r)*2 + mo][lane] = W[32*mo + (lane & 31)][32*m + row(r, lane >> 5)]      (W: [64][64] 1x1 weights)
__global__ __launch_bounds__(256) void pack1x1_kernel(const float* __restrict__ w, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 4096) return;
    const int lane = i & 63, mo = (i >> 6) & 1, kr = i >> 7, m = kr >> 4, r = kr & 15;
    out[i] = w[(32 * mo + (lane & 31)) * 64 + 32 * m + acc_row(r, lane >> 5)];
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
        if (transposed) {   // ConvTranspose2d [Cin][Co][2][2] seen as a 1x1 conv with 4*Co outputs, co = c*4 + (dy*2+dx)
            const int Co = Cout / 4;
            v = w[((int64_t)ci * Co + (co >> 2)) * 4 + (co & 3)];
        } else {
            v = w[((int64_t)co * Cin + ci) * taps + tap];
        }
    }
    out[i] = v;
}

// ---- configuration table.  One entry per (ks, Cout class); pack and launch MUST agree, hence one selector.
typedef Cfg<3, CWFA_CK3, 1, 4, 1, 4> C3_32;                        // Cout <= 32 : 32 ch x 16 rows x 32 cols
typedef Cfg<3, CWFA_CK3, 2, 2, 1, CWFA_WN64> C3_64;                // Cout <= 64 : 64 ch x 2*WN rows x 32 cols
typedef Cfg<3, CWFA_CK3, 2, 2, CWFA_WM128, CWFA_WN128> C3_128;     // Cout  > 64 : 64*WM ch x 2*WN rows x 32 cols
typedef Cfg<1, CWFA_CK1, 1, 4, 1, 4> C1_32;
typedef Cfg<1, CWFA_CK1, 2, 2, 1, CWFA_WN64> C1_64;
typedef Cfg<1, CWFA_CK1, 2, 2, CWFA_WM128, CWFA_WN128> C1_128;
typedef Cfg<1, CWFA_CK1, 1, 4, 1, 4, 4> C1v_32;                           // the same tiles staged 16 bytes per lane
typedef Cfg<1, CWFA_CK1, 2, 2, 1, CWFA_WN64, 4> C1v_64;
typedef Cfg<1, CWFA_CK1, 2, 2, CWFA_WM128, CWFA_WN128, 4> C1v_128;
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

int classify_epilogue(const cwfa_conv_opts& o) {
    const bool res = o.residual != nullptr;
    if (o.out_blocked8) return EPI_NONE_BLK8;                 // (the entry point checked: bias only)
    if (o.upshuffle2) return (!res && o.act == CWFA_ACT_NONE && o.act2 == CWFA_ACT_NONE) ? EPI_UP : EPI_GENERIC;
    if (!res && o.act2 == CWFA_ACT_NONE) {
        if (o.act == CWFA_ACT_NONE) return EPI_NONE;
        if (o.act == CWFA_ACT_ELU) return EPI_ELU;
        if (o.act == CWFA_ACT_PRELU) return EPI_PRELU;
    }
    if (res && o.act == CWFA_ACT_NONE && o.act2 == CWFA_ACT_ELU) return EPI_RES_ELU;
    if (res && o.act == CWFA_ACT_NONE && o.act2 == CWFA_ACT_PRELU) return EPI_RES_PRELU;
    if (res && o.act == CWFA_ACT_GELU && o.act2 == CWFA_ACT_NONE) return EPI_GELU_RES;
    return EPI_GENERIC;
}

template <class C>
int prepare(ConvParams& p, dim3& grid) {
    p.tiles_x = (p.W + C::TC - 1) / C::TC;
    p.tiles_y = (p.H + C::TR - 1) / C::TR;
    p.nchunks = (p.Cin + C::CK - 1) / C::CK;
    const int ctiles = (p.Cout + C::CT - 1) / C::CT;
    CWFA_REQUIRE((int64_t)p.tiles_x * p.tiles_y < (1ll << 31) && ctiles <= 65535 && p.B <= 65535, CWFA_E_SHAPE,
                 "cwfa_conv2d_f32: grid too large");
    CWFA_REQUIRE((int64_t)(p.Cin + C::CK) * p.H * p.W * 4 < (1ll << 31), CWFA_E_SHAPE,
                 "cwfa_conv2d_f32: one sample's input must stay below 2 GiB (32-bit buffer offsets)");
    grid = dim3((unsigned)(p.tiles_x * p.tiles_y), ctiles, p.B);
    return CWFA_OK;
}

template <class C, int EPI, bool PRO>
int launch_epi(ConvParams p, hipStream_t stream) {
    dim3 grid;
    int rc = prepare<C>(p, grid);
    if (rc) return rc;
    // the generic epilogue stages accumulators through 4 KB of LDS per wave
    constexpr int LDS = EPI == EPI_GENERIC && C::LDS_BYTES < C::NTHREADS * 64 ? C::NTHREADS * 64 : C::LDS_BYTES;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv2d_mfma_kernel<C, EPI, PRO>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) {
            cwfa_set_error("cwfa_conv2d_f32: hipFuncSetAttribute(%d bytes LDS): %s", LDS, hipGetErrorString(e));
            return CWFA_E_HIP;
        }
        attr_set = true;
    }
    hipLaunchKernelGGL((conv2d_mfma_kernel<C, EPI, PRO>), grid, dim3(C::NTHREADS), LDS, stream, p);
    CWFA_LAUNCH_CHECK("cwfa_conv2d_f32");
    return CWFA_OK;
}

// the (config, epilogue, prologue) triples the model uses get a specialised kernel; everything else is generic
template <class C, unsigned ALLOWED, bool PRO>
int launch_sel(const ConvParams& p, int epi, hipStream_t st) {
    if (!(ALLOWED & (1u << epi)) && epi != EPI_NONE_BLK8) epi = EPI_GENERIC;
    switch (epi) {
        case EPI_NONE: if constexpr (ALLOWED & (1u << EPI_NONE)) return launch_epi<C, EPI_NONE, PRO>(p, st); break;
        case EPI_ELU: if constexpr (ALLOWED & (1u << EPI_ELU)) return launch_epi<C, EPI_ELU, PRO>(p, st); break;
        case EPI_RES_ELU: if constexpr (ALLOWED & (1u << EPI_RES_ELU)) return launch_epi<C, EPI_RES_ELU, PRO>(p, st); break;
        case EPI_PRELU: if constexpr (ALLOWED & (1u << EPI_PRELU)) return launch_epi<C, EPI_PRELU, PRO>(p, st); break;
        case EPI_RES_PRELU: if constexpr (ALLOWED & (1u << EPI_RES_PRELU)) return launch_epi<C, EPI_RES_PRELU, PRO>(p, st); break;
        case EPI_GELU_RES: if constexpr (ALLOWED & (1u << EPI_GELU_RES)) return launch_epi<C, EPI_GELU_RES, PRO>(p, st); break;
        case EPI_UP: if constexpr (ALLOWED & (1u << EPI_UP)) return launch_epi<C, EPI_UP, PRO>(p, st); break;
        case EPI_NONE_BLK8:
            if constexpr (ALLOWED & (1u << EPI_NONE_BLK8)) return launch_epi<C, EPI_NONE_BLK8, PRO>(p, st);
            cwfa_set_error("cwfa_conv2d_f32: out_blocked8 is built for the 1x1 kernels with more than 32 output channels");
            return CWFA_E_SHAPE;
        default: break;
    }
    return launch_epi<C, EPI_GENERIC, PRO>(p, st);
}

// ALLOWED: specialised epilogues without a load-side prologue; ALLOWED_PRO: with one (BatchNorm-on-load / skip add)
template <class C, unsigned ALLOWED, unsigned ALLOWED_PRO>
int launch(const ConvParams& p, int epi, hipStream_t st) {
    if (p.o.in_scale || p.o.in_add) return launch_sel<C, ALLOWED_PRO, true>(p, epi, st);
    return launch_sel<C, ALLOWED, false>(p, epi, st);
}

}  // namespace
int g_cwfa_split_products = 6;
int g_cwfa_split_xcd_map = 1;
int g_cwfa_split_rows16 = 1;
namespace {

int fill_params(ConvParams& p, const char* name, const float* x, const float* w_packed, float* y, int B, int Cin, int H,
                int W, int Cout, int64_t x_bs, int64_t y_bs, const cwfa_conv_opts* opts) {
    CWFA_REQUIRE(B >= 0 && Cin > 0 && Cout > 0 && H >= 0 && W >= 0, CWFA_E_INVAL, "%s: bad size", name);
    if (B == 0 || H == 0 || W == 0) return 1;                                    // empty: nothing to do
    CWFA_REQUIRE(x && w_packed && y, CWFA_E_INVAL, "%s: null pointer", name);
    CWFA_REQUIRE((int64_t)Cin * H * W < (1ll << 30) && (int64_t)Cout * H * W * (p.o.upshuffle2 ? 1 : 1) < (1ll << 30), CWFA_E_SHAPE, "%s: C*H*W exceeds the 32-bit byte offsets of a tile (2^30 elements per image)", name);
    CWFA_REQUIRE(cwfa_aligned16(w_packed), CWFA_E_ALIGN, "%s: packed weights must be 16-byte aligned", name);
    p.x = x; p.wp = w_packed; p.y = y;
    p.B = B; p.Cin = Cin; p.H = H; p.W = W; p.Cout = Cout;
    p.x_bs = x_bs; p.y_bs = y_bs;
    p.products = g_cwfa_split_products;
    if (opts) p.o = *opts;
    CWFA_REQUIRE(!(p.o.in_scale && !p.o.in_shift), CWFA_E_INVAL, "%s: in_scale without in_shift", name);
    CWFA_REQUIRE(p.o.act >= 0 && p.o.act <= CWFA_ACT_RELU && p.o.act2 >= 0 && p.o.act2 <= CWFA_ACT_RELU, CWFA_E_INVAL,
                 "%s: bad activation", name);
    CWFA_REQUIRE(!((p.o.act == CWFA_ACT_PRELU || p.o.act2 == CWFA_ACT_PRELU) && !p.o.prelu_alpha), CWFA_E_INVAL,
                 "%s: PReLU without prelu_alpha", name);
    return CWFA_OK;
}

// ------------------------------------------------------------------------------------------------ split-bf16 1x1 GEMM
// (ops.set_precision("split_bf16") / "bf16"; the benchmark's arithmetic): 1x1 convolutions / ConvTranspose2d(k2,s2) with >= 128 output
// channels as an fp32-ACCURATE GEMM on the bf16 matrix pipe.  Every fp32 operand is split exactly into three bf16
// pieces (v = v1 + v2 + v3, 24 mantissa bits), the six products with i + j <= 4 are accumulated in fp32 by
// v_mfma_f32_32x32x16_bf16 (16x the fp32-MFMA rate / 6 products = 2.7x), error = fp32-level.
//   * cwfa_split_input_f32: one HBM pass, x (+ load-side affine / added tensor) -> three bf16 planes laid out
//     [piece][channel group of 8][pixel][8], i.e. exactly the B-operand fragments (16 bytes per lane);
//   * weights are split at pack time into [cout tile][chunk of 16 ci][piece][k half][256 cout][8];
//   * the GEMM kernel moves both with LDS-DMA (buffer_load ... lds, no staging registers, two chunks ahead, three LDS
//     buffers), block = 256 cout x 8 rows x 32 px, 8 waves of 2 x 4 accumulator tiles, the epilogues of the fp32 path.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef Cfg<1, 16, 2, 4, 4, 2> CS;     // 256 ch x 8 rows x 32 cols, 512 threads (geometry only: staging is its own)
constexpr int CS_XB = 3 * 2 * 8 * 32 * 16, CS_WB = 3 * 2 * 256 * 16, CS_BUFB = CS_XB + CS_WB;    // bytes per LDS buffer

__device__ __forceinline__ void split3(float v, unsigned short (&o)[3]) {
    const __bf16 a1 = (__bf16)v;
    const float r1 = v - (float)a1;
    const __bf16 a2 = (__bf16)r1;
    const float r2 = r1 - (float)a2;
    const __bf16 a3 = (__bf16)r2;
    o[0] = __builtin_bit_cast(unsigned short, a1);
    o[1] = __builtin_bit_cast(unsigned short, a2);
    o[2] = __builtin_bit_cast(unsigned short, a3);
}

// grid (ceil(HW/256), CG2, B): thread = one pixel of one group of 8 channels
__global__ __launch_bounds__(256) void split_input_kernel(const float* __restrict__ x, uint4* __restrict__ ws, int Cin, int CG2,
                                                          int64_t HW, int64_t x_bs, const float* __restrict__ sc,
                                                          const float* __restrict__ sh, int64_t aff_bs,
                                                          const float* __restrict__ add, int64_t add_bs) {
    const int64_t px = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (px >= HW) return;
    const int cg = blockIdx.y, b = blockIdx.z;
    unsigned short pc[3][8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = cg * 8 + j;
        float v = 0.f;
        if (c < Cin) {
            v = x[(int64_t)b * x_bs + (int64_t)c * HW + px];
            if (sc) v = v * sc[(int64_t)b * aff_bs + c] + sh[(int64_t)b * aff_bs + c];
            if (add) v += add[(int64_t)b * add_bs + (int64_t)c * HW + px];
        }
        unsigned short o[3];
        split3(v, o);
        pc[0][j] = o[0]; pc[1][j] = o[1]; pc[2][j] = o[2];
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        uint4 u;
        u.x = pc[q][0] | ((unsigned)pc[q][1] << 16);
        u.y = pc[q][2] | ((unsigned)pc[q][3] << 16);
        u.z = pc[q][4] | ((unsigned)pc[q][5] << 16);
        u.w = pc[q][6] | ((unsigned)pc[q][7] << 16);
        ws[(((int64_t)b * 3 + q) * CG2 + cg) * HW + px] = u;
    }
}

// one thread per (cout tile, chunk, tap, k half, cout) 8-channel fragment: writes its three pieces
__global__ __launch_bounds__(256) void split_pack_kernel(const float* __restrict__ w, uint4* __restrict__ out, int Cout, int Cin,
                                                         int nchunks, int taps, int transposed, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;       // over [ctile][chunk][tap][h][co 256]
    if (i >= total) return;
    const int col = (int)(i % 256), h = (int)((i / 256) % 2), tap = (int)((i / 512) % taps);
    const int chunk = (int)((i / (512 * (int64_t)taps)) % nchunks);
    const int ctile = (int)(i / ((int64_t)512 * taps * nchunks));
    const int co = ctile * 256 + col;
    unsigned short pc[3][8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int ci = chunk * 16 + h * 8 + j;
        float v = 0.f;
        if (co < Cout && ci < Cin)
            v = transposed ? w[((int64_t)ci * (Cout / 4) + (co >> 2)) * 4 + (co & 3)] : w[((int64_t)co * Cin + ci) * taps + tap];
        unsigned short o[3];
        split3(v, o);
        pc[0][j] = o[0]; pc[1][j] = o[1]; pc[2][j] = o[2];
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        uint4 u;
        u.x = pc[q][0] | ((unsigned)pc[q][1] << 16);
        u.y = pc[q][2] | ((unsigned)pc[q][3] << 16);
        u.z = pc[q][4] | ((unsigned)pc[q][5] << 16);
        u.w = pc[q][6] | ((unsigned)pc[q][7] << 16);
        out[((((int64_t)ctile * nchunks + chunk) * taps + tap) * 3 + q) * 512 + h * 256 + col] = u;
    }
}

struct SplitParams {
    ConvParams c;           // x unused; wp = split weights; the epilogue fields as in the fp32 path
    const void* ws;         // split input planes
    int CG2;
    int64_t ws_bs;          // bytes per sample
};

template <int EPI>
__global__ __launch_bounds__(512, 1) void conv1x1_split_kernel(SplitParams sp) {
    typedef CS C;
    const ConvParams& p = sp.c;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* lds = reinterpret_cast<char*>(smem);
    const Tile t = make_tile<C>(p);
    const bool six = p.products != 1;
    const int tid = threadIdx.x, wave = tid >> 6;
    const int64_t HW = (int64_t)p.H * p.W;
    constexpr unsigned OOB = 0x80000000u;

    // LDS-DMA: entry e = tid + i*512 of a buffer part (16 bytes each); a wave instruction fills 1 KB contiguously
    unsigned xoff[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int e = tid + i * 512;
        const int px = e & 31, row = (e >> 5) & 7, h = (e >> 8) & 1, piece = e >> 9;
        const int gr = t.row0 + row, gc = t.col0 + px;
        xoff[i] = (gr < p.H && gc < p.W) ? (unsigned)((((int64_t)piece * sp.CG2 + h) * HW + (int64_t)gr * p.W + gc) * 16) : OOB;
    }
    const auto rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(sp.ws) + (int64_t)t.b * sp.ws_bs),
                                                      0, (int)sp.ws_bs, 0x00020000);
    const auto rw = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(p.wp) + (int64_t)t.ct * p.nchunks * CS_WB), 0, p.nchunks * CS_WB, 0x00020000);
    const int xchunk = (int)(2 * HW * 16);
    typedef __attribute__((address_space(3))) void* lds_ptr;
    auto dma = [&](int chunk, int buf) {          // chunks past the end: out of range, zeros, never read
        char* base = lds + buf * CS_BUFB + wave * 1024;
#pragma unroll
        for (int i = 0; i < 3; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr)(base + i * 8192), 16, xoff[i], chunk * xchunk, 0, 0);
#pragma unroll
        for (int i = 0; i < 3; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr)(base + CS_XB + i * 8192), 16, (unsigned)(tid + i * 512) * 16u,
                                                     chunk * CS_WB, 0, 0);
    };

    f32x16 acc[2][4];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

    const char* alane = lds + CS_XB + ((t.kh * 256) + t.wm * 64 + t.l31) * 16;        // + (piece*512 + m*32) * 16
    const char* blane = lds + ((t.kh * 8 + t.wn * 4) * 32 + t.l31) * 16;               // + (piece*512 + n*32) * 16

    dma(0, 0);
    dma(1, 1);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    int buf = 0;
    for (int chunk = 0; chunk < p.nchunks; ++chunk) {
        int nb = buf + 2;
        nb = nb >= 3 ? nb - 3 : nb;
        dma(chunk + 2, nb);
        const char* ab = alane + buf * CS_BUFB;
        const char* bb = blane + buf * CS_BUFB;
        bf16x8 A[2][3], Bq[2][3];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int q = 0; q < 3; ++q) A[m][q] = *reinterpret_cast<const bf16x8*>(ab + (q * 512 + m * 32) * 16);
#pragma unroll
        for (int q = 0; q < 3; ++q) Bq[0][q] = *reinterpret_cast<const bf16x8*>(bb + (q * 512) * 16);
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            if (n < 3) {
#pragma unroll
                for (int q = 0; q < 3; ++q) Bq[(n + 1) & 1][q] = *reinterpret_cast<const bf16x8*>(bb + (q * 512 + (n + 1) * 32) * 16);
            }
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                // smallest terms first: (3,1) (2,2) (1,3) (2,1) (1,2) (1,1)
                f32x16 c = acc[m][n];
                if (six) {
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[m][2], Bq[n & 1][0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[m][1], Bq[n & 1][1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[m][0], Bq[n & 1][2], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[m][1], Bq[n & 1][0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[m][0], Bq[n & 1][1], c, 0, 0, 0);
                }
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[m][0], Bq[n & 1][0], c, 0, 0, 0);
                acc[m][n] = c;
            }
        }
        // the DMA of chunk + 1 (issued one chunk ago) must have landed everywhere; the six just issued may stay in flight
        asm volatile("s_waitcnt vmcnt(6)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        buf = buf + 1 == 3 ? 0 : buf + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // drain the zero-fill DMAs before LDS is reused / the block ends
    __syncthreads();
    epilogue<C, EPI>(p, t, acc);
}

template <int EPI>
int launch_split(SplitParams sp, hipStream_t stream) {
    ConvParams& p = sp.c;
    p.tiles_x = (p.W + CS::TC - 1) / CS::TC;
    p.tiles_y = (p.H + CS::TR - 1) / CS::TR;
    const int ctiles = (p.Cout + CS::CT - 1) / CS::CT;
    CWFA_REQUIRE((int64_t)p.tiles_x * p.tiles_y < (1ll << 31) && ctiles <= 65535 && p.B <= 65535, CWFA_E_SHAPE,
                 "cwfa_conv_split_f32: grid too large");
    constexpr int LDS = 3 * CS_BUFB;
    auto kern = &conv1x1_split_kernel<EPI>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) {
            cwfa_set_error("cwfa_conv_split_f32: hipFuncSetAttribute(%d bytes LDS): %s", LDS, hipGetErrorString(e));
            return CWFA_E_HIP;
        }
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)(p.tiles_x * p.tiles_y), ctiles, p.B), dim3(512), LDS, stream, sp);
    CWFA_LAUNCH_CHECK("cwfa_conv_split_f32");
    return CWFA_OK;
}

}  // namespace

extern "C" int64_t cwfa_split_workspace_bytes(int B, int Cin, int64_t HW) {
    if (B < 0 || Cin <= 0 || HW < 0) return -1;
    const int64_t CG2 = 2 * ((Cin + 15) / 16);
    return (int64_t)B * 3 * CG2 * HW * 16;
}

extern "C" int cwfa_split_input_f32(const float* x, void* ws, int B, int Cin, int64_t HW, int64_t x_bs, const float* in_scale,
                                    const float* in_shift, int64_t in_affine_bs, const float* in_add, int64_t in_add_bs,
                                    void* stream) {
    CWFA_REQUIRE(x && ws, CWFA_E_INVAL, "cwfa_split_input_f32: null pointer");
    CWFA_REQUIRE(B >= 0 && Cin > 0 && HW >= 0 && B <= 65535, CWFA_E_SHAPE, "cwfa_split_input_f32: bad shape");
    CWFA_REQUIRE(!in_scale == !in_shift, CWFA_E_INVAL, "cwfa_split_input_f32: in_scale and in_shift go together");
    CWFA_REQUIRE(cwfa_aligned16(ws), CWFA_E_ALIGN, "cwfa_split_input_f32: workspace must be 16-byte aligned");
    if (B == 0 || HW == 0) return CWFA_OK;
    const int CG2 = 2 * ((Cin + 15) / 16);
    CWFA_REQUIRE((int64_t)3 * CG2 * HW * 16 < (1ll << 31), CWFA_E_SHAPE, "cwfa_split_input_f32: one sample's planes must stay below 2 GiB");
    hipLaunchKernelGGL(split_input_kernel, dim3((unsigned)((HW + 255) / 256), CG2, B), dim3(256), 0, (hipStream_t)stream, x,
                       reinterpret_cast<uint4*>(ws), Cin, CG2, HW, x_bs, in_scale, in_shift, in_affine_bs, in_add, in_add_bs);
    CWFA_LAUNCH_CHECK("cwfa_split_input_f32");
    return CWFA_OK;
}

extern "C" int64_t cwfa_conv_split_packed_bytes(int Cout, int Cin, int ks) {
    if (Cout <= 0 || Cin <= 0 || ks != 1) return -1;           // 3x3: cwfa_conv3x3_split_packed_bytes (conv_split3x3.hip)
    return (int64_t)((Cout + 255) / 256) * ((Cin + 15) / 16) * CS_WB;
}

extern "C" int cwfa_conv_split_pack_f32(const float* w, void* packed, int Cout, int Cin, int ks, int transposed, void* stream) {
    CWFA_REQUIRE(w && packed, CWFA_E_INVAL, "cwfa_conv_split_pack_f32: null pointer");
    CWFA_REQUIRE(Cout > 0 && Cin > 0 && ks == 1 && (!transposed || Cout % 4 == 0), CWFA_E_SHAPE,
                 "cwfa_conv_split_pack_f32: bad shape (1x1 / transposed 2x2 only; 3x3: cwfa_conv3x3_split_pack_f32)");
    CWFA_REQUIRE(cwfa_aligned16(packed), CWFA_E_ALIGN, "cwfa_conv_split_pack_f32: packed image must be 16-byte aligned");
    const int nchunks = (Cin + 15) / 16, taps = ks * ks;
    const int64_t total = (int64_t)((Cout + 255) / 256) * nchunks * taps * 512;
    hipLaunchKernelGGL(split_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w,
                       reinterpret_cast<uint4*>(packed), Cout, Cin, nchunks, taps, transposed, total);
    CWFA_LAUNCH_CHECK("cwfa_conv_split_pack_f32");
    return CWFA_OK;
}

extern "C" int cwfa_conv_split_f32(const void* ws, const void* w_packed, float* y, int B, int Cin, int H, int W, int Cout, int ks,
                                   int64_t y_bs, const cwfa_conv_opts* opts, void* stream) {
    CWFA_REQUIRE(!(opts && (opts->in_blocked8 || opts->out_blocked8 || opts->in_cat)), CWFA_E_INVAL,
                 "cwfa_conv_split_f32: blocked layouts / two-source inputs are not built here");
    CWFA_REQUIRE(ws && w_packed && y, CWFA_E_INVAL, "cwfa_conv_split_f32: null pointer");
    CWFA_REQUIRE(ks == 1, CWFA_E_SHAPE, "cwfa_conv_split_f32: kernel size %d (1x1 only; 3x3: cwfa_conv3x3_split_f32)", ks);
    SplitParams sp{};
    int rc = fill_params(sp.c, "cwfa_conv_split_f32", reinterpret_cast<const float*>(ws), reinterpret_cast<const float*>(w_packed), y,
                         B, Cin, H, W, Cout, 0, y_bs, opts);
    if (rc) return rc < 0 ? rc : CWFA_OK;
    ConvParams& p = sp.c;
    CWFA_REQUIRE(!p.o.in_scale && !p.o.in_add, CWFA_E_INVAL, "cwfa_conv_split_f32: the load-side prologue belongs in cwfa_split_input_f32");
    CWFA_REQUIRE(!p.o.upshuffle2 || (ks == 1 && Cout % 4 == 0), CWFA_E_SHAPE, "cwfa_conv_split_f32: upshuffle2 needs ks=1, Cout=4*Co");
    CWFA_REQUIRE(!p.o.out_stats && !p.o.prelu_per_channel, CWFA_E_INVAL, "cwfa_conv_split_f32: out_stats / prelu_per_channel are features of cwfa_conv3x3_split_f32");
    p.nchunks = (Cin + 15) / 16;
    sp.ws = ws;
    sp.CG2 = 2 * p.nchunks;
    sp.ws_bs = (int64_t)3 * sp.CG2 * H * W * 16;
    CWFA_REQUIRE(sp.ws_bs + (int64_t)4 * H * W * 16 < (1ll << 31) && (int64_t)(p.nchunks * ks * ks + 2) * CS_WB < (1ll << 31), CWFA_E_SHAPE,
                 "cwfa_conv_split_f32: one sample's planes / one cout tile's weights must stay below 2 GiB");
    const int epi = classify_epilogue(p.o);
    hipStream_t st = (hipStream_t)stream;
    switch (epi) {
        case EPI_NONE: return launch_split<EPI_NONE>(sp, st);
        case EPI_UP: return launch_split<EPI_UP>(sp, st);
        default: return launch_split<EPI_GENERIC>(sp, st);
    }
}

int g_cwfa_wino_min_cout = 1;
int g_cwfa_wino_2d = 512;     // 2-D F(2x2,3x3) for >= 512 output channels (the UNet's plain convolutions), see conv_internal.h

extern "C" int cwfa_set_option(const char* name, int value) {
    CWFA_REQUIRE(name, CWFA_E_INVAL, "cwfa_set_option: null name");
    if (strcmp(name, "winograd_min_cout") == 0) {
        g_cwfa_wino_min_cout = value;
        return CWFA_OK;
    }
    if (strcmp(name, "winograd_2d") == 0) {
        g_cwfa_wino_2d = value;
        return CWFA_OK;
    }
    if (strcmp(name, "wgrad_split") == 0) {         // 1: the 3x3 weight gradient on the bf16 matrix cores in split arithmetic (conv_bwd.hip)
        g_cwfa_wgrad_split = value != 0;
        return CWFA_OK;
    }
    if (strcmp(name, "wgrad_rows") == 0) {          // 0: the 3x3 weight gradient always takes its first (register-staged) form
        g_cwfa_wgrad_rows = value;
        return CWFA_OK;
    }
    if (strcmp(name, "split3x3_xcd_map") == 0) {    // (ablation) 0: blocks of the split 3x3 kernel in plain (spatial tile, cout tile) order
        g_cwfa_split_xcd_map = value;
        return CWFA_OK;
    }
    if (strcmp(name, "split3x3_rows16") == 0) {     // (ablation) 0: the 64-channel tiling of the split 3x3 kernel always on 8-row tiles
        g_cwfa_split_rows16 = value;
        return CWFA_OK;
    }
    if (strcmp(name, "split_products") == 0) {      // 6: fp32-accurate split; 1: plain bf16 operands (BASELINE configs[4])
        CWFA_REQUIRE(value == 1 || value == 6, CWFA_E_INVAL, "cwfa_set_option: split_products must be 1 or 6");
        g_cwfa_split_products = value;
        return CWFA_OK;
    }
    cwfa_set_error("cwfa_set_option: unknown option '%s'", name);
    return CWFA_E_INVAL;
}

extern "C" int64_t cwfa_conv2d_packed_floats(int Cout, int Cin, int ks) {
    const Sel s = select_cfg(ks, Cout);
    if (s.id < 0 || Cout <= 0 || Cin <= 0) return -1;
    if (cwfa_wino_selected(ks, Cout)) return cwfa_wino_packed_floats(Cout, Cin);
    const int64_t ctiles = (Cout + s.CT - 1) / s.CT, nchunks = (Cin + s.CK - 1) / s.CK;
    return ctiles * nchunks * ks * ks * s.CK * s.CT;
}

extern "C" int cwfa_conv2d_pack_f32(const float* w, float* packed, int Cout, int Cin, int ks, int transposed, void* stream) {
    CWFA_REQUIRE(w && packed, CWFA_E_INVAL, "cwfa_conv2d_pack_f32: null pointer");
    CWFA_REQUIRE(!transposed || (ks == 1 && Cout % 4 == 0), CWFA_E_SHAPE,
                 "cwfa_conv2d_pack_f32: transposed source needs ks=1 (2x2 stride-2 deconv as 1x1) and Cout = 4*Co");
    const int64_t total = cwfa_conv2d_packed_floats(Cout, Cin, ks);
    CWFA_REQUIRE(total > 0, CWFA_E_SHAPE, "cwfa_conv2d_pack_f32: unsupported filter %dx%d, Cout=%d, Cin=%d", ks, ks, Cout, Cin);
    if (cwfa_wino_selected(ks, Cout)) return cwfa_wino_pack(w, packed, Cout, Cin, (hipStream_t)stream);
    const Sel s = select_cfg(ks, Cout);
    const int nchunks = (Cin + s.CK - 1) / s.CK;
    hipLaunchKernelGGL(pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, packed, Cout,
                       Cin, ks, s.CT, s.CK, nchunks, total, transposed);
    CWFA_LAUNCH_CHECK("cwfa_conv2d_pack_f32");
    return CWFA_OK;
}

extern "C" int cwfa_conv2d_f32(const float* x, const float* w_packed, float* y, int B, int Cin, int H, int W, int Cout, int ks,
                               int64_t x_bs, int64_t y_bs, const cwfa_conv_opts* opts, void* stream) {
    CWFA_REQUIRE(!(opts && opts->in_blocked8), CWFA_E_INVAL, "cwfa_conv2d_f32: in_blocked8 is a cwfa_conv3x3_split_f32 feature");
    CWFA_REQUIRE(!(opts && opts->in_cat) || (ks == 1 && Cout <= 64 && !opts->in_scale && !opts->in_add && opts->in_cat_from % 16 == 0 &&
                                             opts->in_cat_from > 0 && opts->in_cat_from < Cin && opts->in_cat_c1 > 0 &&
                                             opts->in_cat_c1 <= opts->in_cat_from),
                 CWFA_E_INVAL, "cwfa_conv2d_f32: in_cat needs a 1x1 bank with <= 64 outputs packed with in_cat_from (a multiple of 16) "
                               "columns for the first source, and no load-side affine / add");
    CWFA_REQUIRE(!(opts && opts->out_blocked8) || (ks == 1 && Cout % 8 == 0 && !opts->residual && opts->act == CWFA_ACT_NONE &&
                                                   opts->act2 == CWFA_ACT_NONE && !opts->upshuffle2 && !opts->in_scale && !opts->in_add &&
                                                   cwfa_aligned16(y) && (y_bs & 3) == 0 && (!opts->bias || cwfa_aligned16(opts->bias))),
                 CWFA_E_INVAL, "cwfa_conv2d_f32: out_blocked8 needs a plain 1x1 bank (bias only), Cout %% 8 == 0 and 16-byte aligned y / bias");
    const Sel s = select_cfg(ks, Cout);
    CWFA_REQUIRE(s.id >= 0, CWFA_E_SHAPE, "cwfa_conv2d_f32: kernel size %d not in {1,3,7}", ks);
    ConvParams p{};
    int rc = fill_params(p, "cwfa_conv2d_f32", x, w_packed, y, B, Cin, H, W, Cout, x_bs, y_bs, opts);
    if (rc) return rc < 0 ? rc : CWFA_OK;
    CWFA_REQUIRE(!p.o.upshuffle2 || (ks == 1 && Cout % 4 == 0), CWFA_E_SHAPE, "cwfa_conv2d_f32: upshuffle2 needs ks=1, Cout=4*Co");
    CWFA_REQUIRE(!p.o.out_stats && !p.o.prelu_per_channel, CWFA_E_INVAL, "cwfa_conv2d_f32: out_stats / prelu_per_channel are features of cwfa_conv3x3_split_f32");
    const int epi = classify_epilogue(p.o);
    hipStream_t st = (hipStream_t)stream;
    if (cwfa_wino_selected(ks, Cout)) return cwfa_wino_conv(x, w_packed, y, B, Cin, H, W, Cout, x_bs, y_bs, p.o, st);
    constexpr unsigned N = 1u << EPI_NONE, E = 1u << EPI_ELU, RE = 1u << EPI_RES_ELU, P = 1u << EPI_PRELU,
                       RP = 1u << EPI_RES_PRELU, G = 1u << EPI_GELU_RES, U = 1u << EPI_UP, K = 1u << EPI_NONE_BLK8;
    // 1x1: rows of 4-pixel groups on 16-byte boundaries (image, skip tensor, batch strides) take the vector-staged kernels
    const bool v4 = ks == 1 && (W & 3) == 0 && (x_bs & 3) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 &&
                    (!p.o.in_add || ((p.o.in_add_bs & 3) == 0 && (reinterpret_cast<uintptr_t>(p.o.in_add) & 15) == 0));
    if (v4) {
        switch (s.id) {
            case 3: return launch<C1v_32, N | P | G, P>(p, epi, st);
            case 4: return launch<C1v_64, N | RE | G | K, 0>(p, epi, st);
            default: return launch<C1v_128, N | U | K, U>(p, epi, st);
        }
    }
    switch (s.id) {
        case 0: return launch<C3_32, N | P | RP, 0>(p, epi, st);
        case 1: return launch<C3_64, N | E | P | RP, 0>(p, epi, st);
        case 2: return launch<C3_128, N | P, P>(p, epi, st);
        case 3: return launch<C1_32, N | P | G, P>(p, epi, st);
        case 4: return launch<C1_64, N | RE | G | K, 0>(p, epi, st);
        case 5: return launch<C1_128, N | U | K, U>(p, epi, st);
        case 6: return launch<C7_32, N, 0>(p, epi, st);
        default: return launch<C7_64, N, 0>(p, epi, st);
    }
}

extern "C" int cwfa_subnet_pack1x1_f32(const float* w, float* panel, void* stream) {
    CWFA_REQUIRE(w && panel, CWFA_E_INVAL, "cwfa_subnet_pack1x1_f32: null pointer");
    hipLaunchKernelGGL(pack1x1_kernel, dim3(16), dim3(256), 0, (hipStream_t)stream, w, panel);
    CWFA_LAUNCH_CHECK("cwfa_subnet_pack1x1_f32");
    return CWFA_OK;
}

extern "C" int cwfa_subnet_layer_tape_f32(const float* x, const float* w3_packed, const float* b3, const float* w1_panel,
                                          const float* b1, float* y, float* hidden, int B, int H, int W, int64_t x_bs, int64_t y_bs,
                                          int64_t hidden_bs, void* stream) {
    ConvParams p{};
    cwfa_conv_opts o{};
    o.bias = b3;
    int rc = fill_params(p, "cwfa_subnet_layer_tape_f32", x, w3_packed, y, B, 64, H, W, 64, x_bs, y_bs, &o);
    if (rc) return rc < 0 ? rc : CWFA_OK;
    CWFA_REQUIRE(b3 && w1_panel && b1 && hidden, CWFA_E_INVAL, "cwfa_subnet_layer_tape_f32: null pointer");
    CWFA_REQUIRE(x != y && x != hidden && y != hidden, CWFA_E_INVAL, "cwfa_subnet_layer_tape_f32: x, y and hidden must be distinct");
    CWFA_REQUIRE(cwfa_aligned16(w1_panel), CWFA_E_ALIGN, "cwfa_subnet_layer_tape_f32: 1x1 panel must be 16-byte aligned");
    CWFA_REQUIRE(cwfa_wino_selected(3, 64), CWFA_E_INVAL,
                 "cwfa_subnet_layer_tape_f32: needs the Winograd packing of the 3x3 bank (option winograd_min_cout <= 64)");
    return cwfa_wino_layer(x, w3_packed, b3, w1_panel, b1, y, B, H, W, x_bs, y_bs, (hipStream_t)stream, hidden, hidden_bs);
}

extern "C" int cwfa_subnet_layer_f32(const float* x, const float* w3_packed, const float* b3, const float* w1_panel,
                                     const float* b1, float* y, int B, int H, int W, int64_t x_bs, int64_t y_bs, void* stream) {
    ConvParams p{};
    cwfa_conv_opts o{};
    o.bias = b3;
    int rc = fill_params(p, "cwfa_subnet_layer_f32", x, w3_packed, y, B, 64, H, W, 64, x_bs, y_bs, &o);
    if (rc) return rc < 0 ? rc : CWFA_OK;
    CWFA_REQUIRE(b3 && w1_panel && b1, CWFA_E_INVAL, "cwfa_subnet_layer_f32: null pointer");
    CWFA_REQUIRE(x != y, CWFA_E_INVAL, "cwfa_subnet_layer_f32: in-place not supported (3x3 halo)");
    CWFA_REQUIRE(cwfa_aligned16(w1_panel), CWFA_E_ALIGN, "cwfa_subnet_layer_f32: 1x1 panel must be 16-byte aligned");
    if (cwfa_wino_selected(3, 64))      // the 3x3 weights were packed in Winograd form
        return cwfa_wino_layer(x, w3_packed, b3, w1_panel, b1, y, B, H, W, x_bs, y_bs, (hipStream_t)stream);
    static_assert(CL::CT == 64 && C3_64::CT == 64 && C3_64::CK == CL::CK, "fused layer shares the 3x3 64-channel packing");
    p.w1x1 = w1_panel;
    p.b1x1 = b1;
    dim3 grid;
    rc = prepare<CL>(p, grid);
    if (rc) return rc;
    hipLaunchKernelGGL(subnet_layer_kernel, grid, dim3(CL::NTHREADS), CL::LDS_BYTES, (hipStream_t)stream, p);
    CWFA_LAUNCH_CHECK("cwfa_subnet_layer_f32");
    return CWFA_OK;
}
