// Fused condition-net 3-D stage  Conv3d(1->K, 3^3) -> PReLU -> Conv3d(K->1, 3^3)  (networks.py:221-225,239) with
// fp32-equivalent arithmetic on the bf16 matrix cores: every fp32 operand is split EXACTLY into three bf16 pieces and the
// six partial products with i + j <= 4 are accumulated in fp32 by v_mfma_f32_16x16x32_bf16 (conv_split3x3.hip has the
// argument; `split_products` = 1: plain bf16 operands, BASELINE.json configs[4]).  K <= 32.
//
// Both convolutions are GEMMs over N = 16 hidden voxels of one image row (lanes n = lane & 15; g = lane >> 4):
//   conv1   hid[k][n] = b1[k] + sum_tap w1[k][tap] x[voxel n + tap]      M = k (2 m-tiles), K = 27 taps (of 32 slots)
//           B operand: the 27-neighbourhood, gathered from an LDS tile that holds each x value as its three pieces in one
//           8-byte entry (8 ds_read_b64 + 12 v_perm per n-tile); the bias rides in the accumulator's initial value.
//   conv2   the accumulators of conv1 ARE its B operand (K = k: slot 8g + j <-> channel 16 (j >> 2) + 4g + (j & 3)), after
//           PReLU and the three-way split.  M = the nine (dh, dw) taps of ONE depth offset dd (rows 4 dh + dw of a 16-row
//           tile), i.e. three A operands; the sum over dd is taken by the matrix cores themselves: a wave walks the depth
//           axis and keeps a ring of three accumulator tiles, output slab o collects hidden slabs o-1, o, o+1 through the C
//           input (the fresh slot starts from C = 0), so nothing is scattered.  A completed tile G'[dh][dw] goes to LDS
//           (9 planes) and one thread per output voxel adds its nine values (+ b2) and stores.
// Block = 512 threads = 8 waves (two per SIMD); tile = 14 x 30 output voxels x DC depth slabs; hidden tile 16 rows x 32
// columns, wave w owns hidden rows 2w, 2w + 1 (4 n-tile columns, 48 ring registers).  x slabs stream through a 4-slot LDS
// ring: slab d' + 2 is written while hidden slab d' is computed, slab d' + 3 is in flight in registers.  One barrier per slab.
#include "common.h"

#include <stdlib.h>
#include <utility>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

#ifndef C3S_SGB
#define C3S_SGB 0
#endif
// measured on one box, all four condition-net sizes (tools/c3s_tune.py): plain 0.954 - 0.979 ms, software pipeline over the columns
// with a 1 MFMA : 2 VALU scheduling pattern 0.951, + waves 4..7 doing the side work mid-step 0.930
#ifndef C3S_STAGGER
#define C3S_STAGGER 1
#endif
#ifndef C3S_PIPE
#define C3S_PIPE 2
#endif
#ifndef C3S_PRIO
#define C3S_PRIO 0
#endif
#ifndef C3S_STAMP
#define C3S_STAMP 0
#endif
#ifndef C3S_ABL
#define C3S_ABL 0          // (timing ablations only, results wrong) 1: no conv1 MFMAs, 2: no conv2 MFMAs, 4: no residual split, 8: no side work, 16: no B1 gather
#endif

extern int g_cwfa_split_products;       // conv2d.hip ("split_products" option: 6 or 1)

#if C3S_STAMP
// diagnostic build only (tools/c3s_tune.py): s_memtime stamps of block 300's waves 0 and 4 at the section borders of steps 10..13;
// never compiled into the library
__device__ unsigned long long g_c3s_stamps[2 * 4 * 16];
extern "C" int cwfa_dbg_c3s_stamps(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_c3s_stamps), sizeof(g_c3s_stamps));
}
#define STAMP(k)                                                                                                     \
    do {                                                                                                             \
        if (blockIdx.x == 300 && blockIdx.y == 0 && (wave == 0 || wave == 4) && lane == 0 && s >= 10 && s < 14)      \
            g_c3s_stamps[((wave >> 2) * 4 + (s - 10)) * 16 + (k)] = __builtin_readcyclecounter();                  \
    } while (0)
#else
#define STAMP(k)
#endif

namespace {

struct G3 {
    static constexpr int NTH = 512, HR = 16, HT = HR - 2, NTW = 2, WT = 16 * NTW - 2, RPW = 2;
    static constexpr int XR = HR + 2, XC = 16 * NTW + 2;          // staged x rows / columns (origin -2)
    static constexpr int ROWB = 384;                                // bytes of one x row in LDS (34 entries x 8 B, padded: = 128 mod 256)
    static constexpr int SLABB = XR * ROWB + 128;                   // one x slab (= 128 mod 256)
    static constexpr int NE = XR * XC, NEK = (NE + NTH - 1) / NTH;  // staged entries per slab / per thread
    static constexpr int GROW = 16 * NTW * 4;                       // bytes of one G' row
    static constexpr int GPS = HR * GROW + 64;                      // one (dh, dw) plane (lane groups 64 B apart modulo 128)
    static constexpr int GB = 9 * GPS;                              // one G' buffer
    static constexpr int OFF_G = 4 * SLABB;
    static constexpr int LDS = OFF_G + 2 * GB;
    static constexpr int NOUT = HT * WT;
    static_assert(ROWB >= XC * 8 && ROWB % 256 == 128 && SLABB % 256 == 128 && NOUT <= NTH && LDS <= 160 * 1024, "geometry");
};

struct P3 {
    const float *x, *w1, *b1, *alpha, *w2, *b2;
    float* y;
    int D, H, W, K, tiles_w, DC;
};

template <int K>
struct ic {
    static constexpr int value = K;
};

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)

// (v0, v1) -> the packed pairs of their three bf16 pieces (round-to-nearest split: v = p0 + p1 + p2 exactly).  The packed
// conversion result passes through an EMPTY asm statement: seeing through `pk << 16` the compiler would otherwise convert the
// low value a second time on its own.  (No instruction is hidden in asm: the hazard recogniser must see every VALU result that
// an MFMA reads -- a v_cvt_pk_bf16_f32 inside an asm statement directly in front of its MFMA gave NaNs.)  This file is built
// with -fno-slp-vectorize: paired into v_pk_add_f32 the residual subtractions are slow beside MFMAs.
__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {
    const bf16x2 v = {(__bf16)a, (__bf16)b};
    unsigned r = __builtin_bit_cast(unsigned, v);
    asm("" : "+v"(r));
    return r;
}
template <bool SIX>
__device__ __forceinline__ void split_pair(float v0, float v1, unsigned (&pk)[3]) {
    pk[0] = cvt_pk_bf16(v0, v1);
    if constexpr (SIX) {
        const float r0 = v0 - __builtin_bit_cast(float, pk[0] << 16), r1 = v1 - __builtin_bit_cast(float, pk[0] & 0xffff0000u);
        pk[1] = cvt_pk_bf16(r0, r1);
        const float s0 = r0 - __builtin_bit_cast(float, pk[1] << 16), s1 = r1 - __builtin_bit_cast(float, pk[1] & 0xffff0000u);
        pk[2] = cvt_pk_bf16(s0, s1);
    }
}

template <bool SIX>
__device__ __forceinline__ void pack8(const float (&v)[8], bf16x8 (&out)[SIX ? 3 : 1]) {
    u32x4 w[3];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        unsigned pk[3];
        split_pair<SIX>(v[2 * i], v[2 * i + 1], pk);
#pragma unroll
        for (int q = 0; q < (SIX ? 3 : 1); ++q) w[q][i] = pk[q];
    }
#pragma unroll
    for (int q = 0; q < (SIX ? 3 : 1); ++q) out[q] = __builtin_bit_cast(bf16x8, w[q]);
}

// six products, those of b's first piece first: they can start as soon as that piece is converted
template <bool SIX>
__device__ __forceinline__ f32x4 mfma6(f32x4 c, const bf16x8 (&a)[SIX ? 3 : 1], const bf16x8 (&b)[SIX ? 3 : 1]) {
    if constexpr (SIX) {
        c = MFMA(a[2], b[0], c);
        c = MFMA(a[1], b[0], c);
    }
    c = MFMA(a[0], b[0], c);
    if constexpr (SIX) {
        c = MFMA(a[1], b[1], c);
        c = MFMA(a[0], b[1], c);
        c = MFMA(a[0], b[2], c);
    }
    return c;
}

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <bool SIX>
__global__ __launch_bounds__(512, 1) void conv3d_split_kernel(P3 p) {
    typedef G3 C;
    constexpr int NQ = SIX ? 3 : 1;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, n = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tw = blockIdx.x % p.tiles_w, th = blockIdx.x / p.tiles_w;
    const int h0 = th * C::HT, w0 = tw * C::WT, d0 = blockIdx.y * p.DC, b = blockIdx.z;
    const int DCe = min(p.DC, p.D - d0), nsteps = DCe + 2;
    const int HW = p.H * p.W, plane = HW * 4;
    const int K = p.K;
    constexpr unsigned OOB = 0x80000000u;
    const float alpha = *p.alpha;
    // PReLU(v) = max(v, alpha v) for alpha <= 1, min(v, alpha v) otherwise = med3(v, alpha v, +-inf): two instructions, exact
    const float prelu_m = alpha <= 1.f ? __builtin_inff() : -__builtin_inff();

    // ---- operand panels, resident in registers for the whole block
    bf16x8 A1[2][NQ], A2[3][NQ];
    f32x4 bias1[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        float v[8];
        const int k = 16 * t + n;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int tap = 8 * g + j;
            const float wv = p.w1[min(k, K - 1) * 27 + min(tap, 26)];
            v[j] = (k < K && tap < 27) ? wv : 0.f;
        }
        pack8<SIX>(v, A1[t]);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int kk = 16 * t + 4 * g + r;
            const float bv = p.b1[min(kk, K - 1)];
            bias1[t][r] = kk < K ? bv : 0.f;
        }
    }
#pragma unroll
    for (int dd = 0; dd < 3; ++dd) {
        float v[8];
        const int dh = n >> 2, dw = n & 3;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = 16 * (j >> 2) + 4 * g + (j & 3);
            const float wv = p.w2[min(k, K - 1) * 27 + min(9 * dh + 3 * dw + dd, 26)];
            v[j] = (k < K && dh < 3 && dw < 3) ? wv : 0.f;
        }
        pack8<SIX>(v, A2[dd]);
    }

    // ---- x staging: entry e = tid + 512 i of the 18 x 34 slab tile
    const auto rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x + (int64_t)b * p.D * HW), 0, p.D * plane, 0x00020000);
    unsigned xgo[C::NEK];
    int xlo[C::NEK];
    bool xin[C::NEK];
#pragma unroll
    for (int i = 0; i < C::NEK; ++i) {
        const int e = tid + C::NTH * i;
        const int row = e / C::XC, col = e % C::XC;
        const int gr = h0 - 2 + row, gc = w0 - 2 + col;
        xin[i] = e < C::NE;
        xgo[i] = (xin[i] && gr >= 0 && gr < p.H && gc >= 0 && gc < p.W) ? (unsigned)((gr * p.W + gc) * 4) : OOB;
        xlo[i] = row * C::ROWB + col * 8;
    }
    float xr[C::NEK];
    auto load_x = [&](int xq) {                  // x slab d0 - 2 + xq -> registers (zeros outside the volume)
        const int d = d0 - 2 + xq;
        const bool ok = d >= 0 && d < p.D;       // uniform
#pragma unroll
        for (int i = 0; i < C::NEK; ++i)
            xr[i] = ok ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, xgo[i], d * plane, 0)) : 0.f;
    };
    auto store_x = [&](int slot) {               // registers -> ring slot, three pieces per 8-byte entry
#pragma unroll
        for (int i = 0; i < C::NEK; ++i) {
            unsigned pk[3] = {0u, 0u, 0u};
            split_pair<SIX>(xr[i], 0.f, pk);
            if (xin[i]) *reinterpret_cast<u32x2*>(lds + slot * C::SLABB + xlo[i]) = u32x2{(pk[0] & 0xffffu) | (pk[1] << 16), pk[2] & 0xffffu};
        }
    };

    // ---- B-operand addresses of conv1: slot 8g + j <-> tap (dh, dw, dd) = (s / 9, (s / 3) % 3, s % 3); the five spare slots
    // (zero weights) re-read tap 26's row / column so that they stay inside the tile
    int laneoff[8], em[3];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int s = min(8 * g + j, 26);
        laneoff[j] = (wave * C::RPW + s / 9) * C::ROWB + (n + (s / 3) % 3) * 8;
    }
#pragma unroll
    for (int m = 0; m < 3; ++m) em[m] = (2 * g + m) % 3;          // dd of slot j = (8g + j) % 3 = em[j % 3]

    bool row_in[C::RPW];
#pragma unroll
    for (int rr = 0; rr < C::RPW; ++rr) {
        const int gh = h0 - 1 + wave * C::RPW + rr;
        row_in[rr] = gh >= 0 && gh < p.H;
    }
    const bool w_edge = w0 - 1 < 0 || w0 - 1 + 16 * C::NTW - 1 >= p.W;     // some hidden columns are conv2's zero padding
    bool in_w[C::NTW];
#pragma unroll
    for (int nt = 0; nt < C::NTW; ++nt) {
        const int gw = w0 - 1 + 16 * nt + n;
        in_w[nt] = gw >= 0 && gw < p.W;
    }

    // ---- output voxel of this thread in the gather phase
    const int oh = tid / C::WT, ow = tid % C::WT;
    const bool o_ok = tid < C::NOUT && h0 + oh < p.H && w0 + ow < p.W;
    const int g_rd = C::OFF_G + oh * C::GROW + ow * 4;
    const unsigned y_off = o_ok ? (unsigned)(((h0 + oh) * p.W + w0 + ow) * 4) : OOB;
    const auto ry = __builtin_amdgcn_make_buffer_rsrc(p.y + (int64_t)b * p.D * HW, 0, p.D * plane, 0x00020000);
    const float bias2 = p.b2[0];
    const int g_wr = C::OFF_G + (g * 3 * C::GPS) + wave * C::RPW * C::GROW + n * 4;

    constexpr int NCOL = C::RPW * C::NTW;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 ring[3][NCOL];
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int c = 0; c < NCOL; ++c) ring[s][c] = zero4;

#if C3S_PRIO
    if (wave >= 4) __builtin_amdgcn_s_setprio(C3S_PRIO);      // the younger SIMD partner loses every arbitration otherwise
#endif
    // ---- prologue: x slabs 0, 1, 2 of the chunk into ring slots 0..2, slab 3 in flight
    load_x(0);
    store_x(0);
    load_x(1);
    store_x(1);
    load_x(2);
    store_x(2);
    load_x(3);
    lds_barrier();

    auto gather = [&](int o, int buf) {          // y(o, .) = b2 + sum_{dh, dw} G'[dh][dw](h + dh, w + dw)
        float acc = bias2;
#pragma unroll
        for (int dh = 0; dh < 3; ++dh)
#pragma unroll
            for (int dw = 0; dw < 3; ++dw)
                acc += *reinterpret_cast<const float*>(lds + g_rd + buf * C::GB + (dh * 3 + dw) * C::GPS + dh * C::GROW + dw * 4);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, acc), ry, y_off, o * plane, 0);
    };

    // pieces of one n-tile column (rr = col / NTW, nt = col % NTW)
    auto rd_b1 = [&](u32x2 (&e)[8], const int (&a)[8], int col) {
#pragma unroll
        for (int j = 0; j < 8; ++j) e[j] = *reinterpret_cast<const u32x2*>(lds + a[j] + (col / C::NTW) * C::ROWB + (col % C::NTW) * 128);
    };
    auto mk_b1 = [&](const u32x2 (&e)[8], bf16x8 (&B1)[NQ]) {
        u32x4 w[3];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            w[0][i] = __builtin_amdgcn_perm(e[2 * i + 1].x, e[2 * i].x, 0x05040100u);
            if constexpr (SIX) {
                w[1][i] = __builtin_amdgcn_perm(e[2 * i + 1].x, e[2 * i].x, 0x07060302u);
                w[2][i] = __builtin_amdgcn_perm(e[2 * i + 1].y, e[2 * i].y, 0x05040100u);
            }
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q) B1[q] = __builtin_bit_cast(bf16x8, w[q]);
    };

    // one hidden slab d' = d0 - 1 + s; PH = s % 3 names the ring slots at compile time.  EDGE = false: interior blocks (every
    // hidden row and column inside the image): straight-line code over the wave's four n-tile columns, all three conv2
    // products always (what a step adds to a slot that is not a live output slab is discarded: a slot is restarted from
    // C = 0 when it becomes the fresh one).  EDGE = true: the generic form with row / column masks.
    auto step = [&](auto ph, auto edge, int s) {
        constexpr int PH = decltype(ph)::value;
        constexpr bool EDGE = decltype(edge)::value != 0;
        constexpr int S0 = PH, S1 = (PH + 2) % 3, S2 = (PH + 1) % 3;     // slots of output slabs d' + 1, d', d' - 1
        const int dp = d0 - 1 + s;
        // the per-step side work (output gather of the slab completed a step ago, x slab into the ring, next slab's loads).  Waves
        // 4..7 -- the SIMD partners of waves 0..3 -- do it in the MIDDLE of their four columns (C3S_STAGGER): partners that run the
        // same stream in phase reach their vector-heavy and their matrix-heavy parts together
        auto side_work = [&]() {
            if (s >= 3 && tid < C::NOUT) gather(d0 + s - 3, (s - 1) & 1);
            store_x((s + 3) & 3);
            load_x(s + 4);
        };
        const bool late = C3S_STAGGER && !EDGE && wave >= 4;
        STAMP(0);
        if (!late && !(C3S_ABL & 8)) side_work();
        STAMP(1);
        const bool do2 = s >= 2;
        const bool slab_in = dp >= 0 && dp < p.D;
        int a[8];
        {
            int V[3];
#pragma unroll
            for (int m = 0; m < 3; ++m) V[m] = ((s + em[m]) & 3) * C::SLABB;
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] = laneoff[j] + V[j % 3];
        }
        if constexpr (!EDGE) {
            if (slab_in) {
#if C3S_PIPE
                // software pipeline over the wave's four columns: region X_c = { PReLU + split of column c (vector) | conv1 of column
                // c + 1 and conv2 of column c - 1 (matrix) }: independent instruction streams the scheduler can interleave
                u32x2 e[2][8];
                bf16x8 B1[NQ], B2[2][NQ];
                f32x4 hd[2][2];
                if (C3S_ABL & 16) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) e[0][j] = e[1][j] = u32x2{(unsigned)a[j], (unsigned)laneoff[j]};
                } else {
                rd_b1(e[0], a, 0);
                rd_b1(e[1], a, 1);
                }
                STAMP(2);
                mk_b1(e[0], B1);
#pragma unroll
                for (int t = 0; t < 2; ++t) hd[0][t] = (C3S_ABL & 1) ? bias1[t] + __builtin_bit_cast(f32x4, B1[0]).xxxx : mfma6<SIX>(bias1[t], A1[t], B1);
                __builtin_amdgcn_sched_barrier(0);
                STAMP(3);
#pragma unroll
                for (int col = 0; col < NCOL; ++col) {
                    if (col == NCOL / 2 && late && !(C3S_ABL & 8)) side_work();
                    float hv[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) hv[j] = __builtin_amdgcn_fmed3f(hd[col & 1][j >> 2][j & 3], alpha * hd[col & 1][j >> 2][j & 3], prelu_m);
                    if (C3S_ABL & 4) {
                        bf16x8 one[1];
                        pack8<false>(hv, one);
#pragma unroll
                        for (int q = 0; q < NQ; ++q) B2[col & 1][q] = one[0];
                    } else {
                        pack8<SIX>(hv, B2[col & 1]);
                    }
                    if (col + 1 < NCOL) {
                        mk_b1(e[(col + 1) & 1], B1);
                        if (col + 2 < NCOL && !(C3S_ABL & 16)) rd_b1(e[col & 1], a, col + 2);
#pragma unroll
                        for (int t = 0; t < 2; ++t) hd[(col + 1) & 1][t] = (C3S_ABL & 1) ? bias1[t] + __builtin_bit_cast(f32x4, B1[0]).xxxx : mfma6<SIX>(bias1[t], A1[t], B1);
                    }
                    if (col > 0) {
                        if (C3S_ABL & 2) {
                            ring[S0][col - 1] = __builtin_bit_cast(f32x4, B2[(col - 1) & 1][0]);
                            ring[S1][col - 1] += __builtin_bit_cast(f32x4, B2[(col - 1) & 1][NQ - 1]);
                        } else {
                        ring[S0][col - 1] = mfma6<SIX>(zero4, A2[0], B2[(col - 1) & 1]);
                        ring[S1][col - 1] = mfma6<SIX>(ring[S1][col - 1], A2[1], B2[(col - 1) & 1]);
                        ring[S2][col - 1] = mfma6<SIX>(ring[S2][col - 1], A2[2], B2[(col - 1) & 1]);
                        }
                    }
#if C3S_PIPE > 1
#pragma unroll
                    for (int i = 0; i < (SIX ? 30 : 5); ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x002, C3S_PIPE, 0);
                    }
#endif
                    __builtin_amdgcn_sched_barrier(0);
                    STAMP(4 + col);
                }
                if (C3S_ABL & 2) {
                    ring[S0][NCOL - 1] = __builtin_bit_cast(f32x4, B2[(NCOL - 1) & 1][0]);
                    ring[S1][NCOL - 1] += __builtin_bit_cast(f32x4, B2[(NCOL - 1) & 1][NQ - 1]);
                } else {
                ring[S0][NCOL - 1] = mfma6<SIX>(zero4, A2[0], B2[(NCOL - 1) & 1]);
                ring[S1][NCOL - 1] = mfma6<SIX>(ring[S1][NCOL - 1], A2[1], B2[(NCOL - 1) & 1]);
                ring[S2][NCOL - 1] = mfma6<SIX>(ring[S2][NCOL - 1], A2[2], B2[(NCOL - 1) & 1]);
                }
#else
                u32x2 e[2][8];
                rd_b1(e[0], a, 0);
#pragma unroll
                for (int col = 0; col < NCOL; ++col) {
                    if (col == NCOL / 2 && late) side_work();
                    if (col + 1 < NCOL) rd_b1(e[(col + 1) & 1], a, col + 1);
                    bf16x8 B1[NQ], B2[NQ];
                    mk_b1(e[col & 1], B1);
                    f32x4 hd[2];
#pragma unroll
                    for (int t = 0; t < 2; ++t) hd[t] = mfma6<SIX>(bias1[t], A1[t], B1);
                    float hv[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) hv[j] = __builtin_amdgcn_fmed3f(hd[j >> 2][j & 3], alpha * hd[j >> 2][j & 3], prelu_m);
                    pack8<SIX>(hv, B2);
                    ring[S0][col] = mfma6<SIX>(zero4, A2[0], B2);
                    ring[S1][col] = mfma6<SIX>(ring[S1][col], A2[1], B2);
                    ring[S2][col] = mfma6<SIX>(ring[S2][col], A2[2], B2);
                }
#endif
            } else {
                if (late) side_work();
#pragma unroll
                for (int col = 0; col < NCOL; ++col) ring[S0][col] = zero4;
            }
        } else {
            const bool do0 = s < DCe, do1 = s >= 1 && s <= DCe;       // which of the three output slabs exist
#pragma unroll
            for (int rr = 0; rr < C::RPW; ++rr) {
                const bool live = slab_in && row_in[rr];                                // uniform
#pragma unroll
                for (int nt = 0; nt < C::NTW; ++nt) {
                    const int col = rr * C::NTW + nt;
                    if (!live) {
                        if (do0) ring[S0][col] = zero4;
                        continue;
                    }
                    u32x2 e[8];
                    rd_b1(e, a, col);
                    bf16x8 B1[NQ], B2[NQ];
                    mk_b1(e, B1);
                    f32x4 hd[2];
#pragma unroll
                    for (int t = 0; t < 2; ++t) hd[t] = mfma6<SIX>(bias1[t], A1[t], B1);
                    float hv[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float v = __builtin_amdgcn_fmed3f(hd[j >> 2][j & 3], alpha * hd[j >> 2][j & 3], prelu_m);
                        hv[j] = in_w[nt] ? v : 0.f;
                    }
                    pack8<SIX>(hv, B2);
                    if (do0) ring[S0][col] = mfma6<SIX>(zero4, A2[0], B2);
                    if (do1) ring[S1][col] = mfma6<SIX>(ring[S1][col], A2[1], B2);
                    if (do2) ring[S2][col] = mfma6<SIX>(ring[S2][col], A2[2], B2);
                }
            }
        }
        STAMP(8);
        if (do2 && g < 3) {                   // output slab d' - 1 is complete: G'[dh = g][dw = r] of this wave's rows
            const int buf = s & 1;
#pragma unroll
            for (int rr = 0; rr < C::RPW; ++rr)
#pragma unroll
                for (int nt = 0; nt < C::NTW; ++nt)
#pragma unroll
                    for (int r = 0; r < 3; ++r)
                        *reinterpret_cast<float*>(lds + g_wr + buf * C::GB + r * C::GPS + rr * C::GROW + nt * 64) = ring[S2][rr * C::NTW + nt][r];
        }
        STAMP(9);
        lds_barrier();
        STAMP(10);
    };

    // interior blocks: every hidden row and column of the tile lies inside the image
    const bool interior = h0 - 1 >= 0 && h0 - 1 + C::HR - 1 < p.H && !w_edge;
    auto run = [&](auto edge) {
        for (int s = 0; s < nsteps;) {
            step(ic<0>{}, edge, s);
            if (++s >= nsteps) break;
            step(ic<1>{}, edge, s);
            if (++s >= nsteps) break;
            step(ic<2>{}, edge, s);
            ++s;
        }
    };
    if (interior) run(ic<0>{});
    else run(ic<1>{});
    if (tid < C::NOUT) gather(d0 + nsteps - 3, (nsteps - 1) & 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the x slab loaded past the end
}

template <bool SIX>
int launch3(const P3& p, int B, int tiles, int chunks, hipStream_t st) {
    auto kern = &conv3d_split_kernel<SIX>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, G3::LDS);
        if (e != hipSuccess) {
            cwfa_set_error("cwfa_conv3d_1k1_split_f32: hipFuncSetAttribute(%d bytes LDS): %s", G3::LDS, hipGetErrorString(e));
            return CWFA_E_HIP;
        }
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)tiles, (unsigned)chunks, (unsigned)B), dim3(512), G3::LDS, st, p);
    return 0;
}

}  // namespace

extern "C" int cwfa_conv3d_1k1_split_f32(const float* x, const float* w1, const float* b1, const float* alpha, const float* w2,
                                         const float* b2, float* y, int B, int D, int H, int W, int K, void* stream) {
    CWFA_REQUIRE(x && w1 && b1 && alpha && w2 && b2 && y, CWFA_E_INVAL, "cwfa_conv3d_1k1_split_f32: null pointer");
    CWFA_REQUIRE(B >= 0 && D >= 0 && H >= 0 && W >= 0 && K > 0, CWFA_E_INVAL, "cwfa_conv3d_1k1_split_f32: bad size");
    CWFA_REQUIRE(K <= 32, CWFA_E_SHAPE, "cwfa_conv3d_1k1_split_f32: K <= 32 (got %d)", K);
    CWFA_REQUIRE(x != y, CWFA_E_INVAL, "cwfa_conv3d_1k1_split_f32: in-place not supported");
    if (B == 0 || D == 0 || H == 0 || W == 0) return CWFA_OK;
    CWFA_REQUIRE((int64_t)(D + 4) * H * W * 4 < (1ll << 31), CWFA_E_SHAPE, "cwfa_conv3d_1k1_split_f32: one sample must stay below 2 GiB");
    const int tw = (W + G3::WT - 1) / G3::WT, th = (H + G3::HT - 1) / G3::HT;
    const int64_t tiles = (int64_t)tw * th;
    CWFA_REQUIRE(tiles < (1ll << 31) && B <= 65535, CWFA_E_SHAPE, "cwfa_conv3d_1k1_split_f32: grid too large");
    // depth chunk: a block walks DC + 2 hidden slabs (+ ~2 slabs' worth of prologue); pick the DC with the least estimated
    // time over whole rounds of the 256 CUs
    int DC = D;
    double best = 1e300;
    for (int dc = D < 4 ? D : 4; dc <= D; ++dc) {
        const int64_t chunks = (D + dc - 1) / dc, blocks = tiles * chunks * B;
        const double rounds = (double)((blocks + 255) / 256);
        const double cost = rounds * (dc + 4.0);
        if (cost < best - 1e-9) {
            best = cost;
            DC = dc;
        }
    }
#ifdef C3S_DC_ENV
    if (const char* e = getenv("CWFA_C3S_DC")) DC = atoi(e) > 0 ? (atoi(e) < D ? atoi(e) : D) : DC;
#endif
    const int chunks = (D + DC - 1) / DC;
    CWFA_REQUIRE(chunks <= 65535, CWFA_E_SHAPE, "cwfa_conv3d_1k1_split_f32: grid too large");
    P3 p{x, w1, b1, alpha, w2, b2, y, D, H, W, K, tw, DC};
    hipStream_t st = (hipStream_t)stream;
    const int rc = g_cwfa_split_products != 1 ? launch3<true>(p, B, (int)tiles, chunks, st) : launch3<false>(p, B, (int)tiles, chunks, st);
    if (rc != CWFA_OK) return rc;
    CWFA_LAUNCH_CHECK("cwfa_conv3d_1k1_split_f32");
    return CWFA_OK;
}
