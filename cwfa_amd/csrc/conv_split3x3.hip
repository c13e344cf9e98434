// 3x3 convolution (stride 1, zero padding 1) with fp32-equivalent arithmetic on the bf16 matrix cores -- the UNet layers of
// the LRNN (unet.py:94-113: 256 / 512 / 1024 channels) and, with the narrow tilings, the output convolutions of the coupling
// sub-networks (networks.py:633-638).
//
// Every fp32 operand is split EXACTLY into three bf16 pieces (24 significand bits) and the six partial products with
// i + j <= 4 are accumulated in fp32 by v_mfma_f32_16x16x32_bf16 (dropped terms <= 2^-24 relative); `split_products` = 1:
// plain bf16 operands (BASELINE.json configs[4]).  The input is the fp32 tensor itself: the load-side prologue of the UNet
// (BatchNorm affine x dropout mask of the producer, + skip tensor) is applied and the value split on the way into LDS.
//
// Block = 512 threads = 8 waves, two per SIMD, tile = CT = 64*MPW output channels x 8 rows x 32 pixels.  Wave (wm, wn) owns
// 16*MPW channels x 4 rows: MPW m-tiles x 8 n-tiles of 16x16 accumulators (MPW = 4: 128 registers); a B fragment (16 bytes
// per lane from LDS) feeds 6*MPW MFMAs and an A fragment 48.  (One wave per SIMD with the whole 256-pixel tile -- 256
// accumulator registers -- was tried first: every issue cost of the single stream, LDS-DMA above all (12 instructions of
// ~60 cycles per step), is then exposed: 225 TF/s against 260 for the two-wave form's predecessor.)
//   K = 32 per MFMA = TWO (16-channel chunk, tap) units of the 9 * ceil(Cin / 16) a 3x3 has; lane group g = lane >> 4
//   reads k-half g & 1 of unit g >> 1.  The haloed input tile of a chunk, [piece 3][k half 2][9 rows][34 px] x 16 B, sits
//   in one of TWO LDS buffers (even / odd chunks); a tap only shifts the B-operand address.  A 9-step period covers an even
//   and an odd chunk: taps (0,1)(2,3)(4,5)(6,7) of the even one, its tap 8 with tap 0 of the odd one, then (1,2)...(7,8).
//   The even buffer is free from step 5 and refilled there (loads in steps 2..4), the odd one in steps 0..2 (loads in steps
//   6..8 of the period before): buffer loads, padding / border / channels >= Cin come back as 0.0 from the range check.
//   Weights: one slice [piece][k group 4][CT][8] per step through a two-slot ring by LDS-DMA, issued at the start of the
//   step before, verified at its end; one barrier per step.
#include "conv_internal.h"

#include <utility>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

extern int g_cwfa_split_products;       // conv2d.hip ("split_products" option: 6 or 1)
extern int g_cwfa_split_xcd_map;        // conv2d.hip ("split3x3_xcd_map" option)
extern int g_cwfa_split_rows16;         // conv2d.hip ("split3x3_rows16" option: 16-row tiles for the 64-channel tiling)

namespace {

constexpr int TR = 8, TC = 32;              // default tile: 8 rows x 32 pixels (four rows per wave); XG<KS, 8>: 16 rows

// input-tile geometry of a KS x KS convolution (KS = 3: the form everything above describes; KS = 7: the ConvNeXt convolution of
// the LRNN, networks.py:488 -- the same kernel with a 3-pixel halo, 49 taps and a 49-step period over two 16-channel chunks)
// RPW = image rows per wave: 4 (tile of 8 rows, 8 n-tiles per wave) or, for the 64-channel tiling of the 3x3, 8 (tile of 16 rows:
// twice the MFMAs per barrier and per A fragment, 18/16 instead of 10/8 halo rows -- the output convolutions of the sub-networks,
// whose K = 9 x 64 is too short to amortise a tile's fill and drain on 8 rows)
template <int KS, int RPW = 4>
struct XG {
    static constexpr int TRW = 2 * RPW, NT = 2 * RPW;   // tile rows; n-tiles per wave (RPW rows x 2 halves of 16 pixels)
    static constexpr int PAD = KS / 2, XR = TRW + 2 * PAD, XC = TC + 2 * PAD, NTAP = KS * KS;
    static constexpr int EPK = XR * XC;                 // entries per k half: 306 (3x3), 612 (3x3, 16 rows), 532 (7x7)
    // bytes of one k-half plane: EPK x 16 padded so that the two k halves differ by 64 bytes modulo 256 (bank phase):
    // 3x3: 4896 -> 5440;  3x3 on 16 rows: 9792 and 7x7: 8512 already are
    static constexpr int KHB = KS == 7 ? 8512 : RPW == 8 ? 9792 : 5440;
    static constexpr int XPB = 2 * KHB;                 // one piece plane
    static constexpr int XB = 3 * XPB;                  // one input buffer (3x3: 32 640 / 58 752; 7x7: 51 072)
    static constexpr bool SPECIAL = KS == 3 && RPW == 4;        // the two-entry staging map with a wave-uniform k half
    static constexpr int NEK = SPECIAL ? 2 : 3;         // staging entries per thread and chunk
    static_assert(EPK * 16 <= KHB && KHB % 256 == 64 && 2 * EPK <= 512 * NEK, "staging entry map");
    static_assert(!SPECIAL || (EPK > 256 && EPK <= 384), "3x3 staging entry map");
    static_assert(KS == 3 || RPW == 4, "7x7: 8-row tiles");
};

template <int MPW, int KS = 3, int RPW = 4, int WM = 4>
struct Geo {
    static constexpr int CT = 16 * MPW * WM;            // output channels per block (WM channel groups of 16 MPW)
    static constexpr int WSL = 3 * 4 * CT * 16;         // bytes of one weight slice (K = 32)
    static constexpr int LDS = 2 * XG<KS, RPW>::XB + 2 * WSL;   // 3x3, MPW = 4: 163 584 ... see the static_assert
    static_assert(LDS <= 160 * 1024, "LDS budget");
};

struct SParams {
    const float* x;
    const void* wp;
    float* y;
    int B, Cin, H, W, Cout, nchunks, nsteps, tiles_x, ntiles, ctiles, xcd_map;
    int nrun;                 // steps that are executed: nsteps, minus the trailing steps of an all-zero odd chunk (3x3 with an odd chunk count)
    int64_t x_bs, y_bs;
    cwfa_conv_opts o;
    cwfa_couple cp;             // EPI_COUPLE only
};

template <int K>
struct ic {
    static constexpr int value = K;
};
template <class F, int... S>
__device__ __forceinline__ void sfor_impl(F&& f, std::integer_sequence<int, S...>) {
    (f(ic<S>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void sfor(F&& f) {
    sfor_impl(f, std::make_integer_sequence<int, N>{});
}

template <bool SIX>
__device__ __forceinline__ void split3(float v, __bf16& a1, __bf16& a2, __bf16& a3) {
    a1 = (__bf16)v;
    if constexpr (SIX) {
        const float r1 = v - (float)a1;
        a2 = (__bf16)r1;
        const float r2 = r1 - (float)a2;
        a3 = (__bf16)r2;
    }
}

template <int ACT>
__device__ __forceinline__ float act_of(float v, float alpha) {
    if constexpr (ACT == CWFA_ACT_ELU) return cwfa_elu(v);
    if constexpr (ACT == CWFA_ACT_PRELU) return v > 0.f ? v : alpha * v;
    if constexpr (ACT == CWFA_ACT_GELU) return cwfa_gelu(v);
    if constexpr (ACT == CWFA_ACT_RELU) return v > 0.f ? v : 0.f;
    return v;
}

#define MFMA(a, b, c) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)
#define FENCE() __builtin_amdgcn_sched_barrier(0)

// byte offset of the B fragments of tap t in buffer `buf`, relative to the lane base
template <int KS, int RPW>
__host__ __device__ constexpr int tap_off(int buf, int tap) { return buf * XG<KS, RPW>::XB + ((tap / KS) * XG<KS, RPW>::XC + tap % KS) * 16; }

enum { EPI_RUNTIME = -1, EPI_COUPLE = -2 };

// the soft clamp of the coupling blocks (coupling_layers.py:50-60; same expressions as csrc/elementwise.hip)
__device__ __forceinline__ float soft_clamp(float a, int kind, float clamp) {
    switch (kind) {
        case CWFA_CLAMP_ATAN: return clamp * (0.636f * cwfa_atan(a));
        case CWFA_CLAMP_TANH: return clamp * cwfa_tanh(a);
        case CWFA_CLAMP_SIGMOID: return clamp * (2.f * (1.f / (1.f + expf(-a)) - 0.5f));
        default: return clamp * a;
    }
}

// ADD: a second tensor is added on load (UNet skip); ACT1: compile-time activation of the common epilogues (bias -> ACT1),
// EPI_RUNTIME = whatever cwfa_conv_opts says (bias -> act -> + residual -> act2)
// WM = channel groups per block (4: the form described above; 2 / 1: the narrow tilings for banks with <= 32 / <= 16 outputs -- the
// condition nets' 2-D convolutions and the output convolutions of the coarse steps' sub-networks, networks.py:212-219,633-638 --
// on the 16-row tile: the eight waves are WM channel groups x 8 / WM row groups of 16 WM / 8 rows each)
template <int MPW, bool SIX, bool ADD, int ACT1, int KS = 3, int RPW = 4, int WM = 4>
__global__ __launch_bounds__(512, 1) void conv3x3_split_kernel(SParams p) {
    typedef XG<KS, RPW> G;
    constexpr int XC = G::XC, EPK = G::EPK, KHB = G::KHB, XPB = G::XPB, XB = G::XB, NTAP = G::NTAP, PAD = G::PAD;
    constexpr int WN = 8 / WM, RW = G::TRW / WN, NT = 2 * RW;      // row groups, image rows and n-tiles per wave
    constexpr bool SPECIAL = G::SPECIAL;
    static_assert(SPECIAL || ((MPW == 1 || WM != 4) && !ADD), "7x7 / 16-row tiles: the 64-channel and narrow tilings, without a skip add");
    static_assert(WM == 4 || (((KS == 3 && RPW == 8) || KS == 7) && MPW <= 3 && !ADD && ACT1 != EPI_COUPLE),
                  "narrow tilings: 16-row tile (7x7: 8-row), <= 3 m-tiles per wave");
    static_assert(MPW != 3 || WM != 4, "three m-tiles per wave: the 48- / 96-channel tilings only");
    constexpr int CT = 16 * MPW * WM;                   // output channels per block
    constexpr int WSL = 3 * 4 * CT * 16;                // bytes of one weight slice (K = 32)
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, c16 = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);          // scalar: everything derived from it stays in SGPRs
    const int wm = wave % WM, wn = wave / WM;                           // channel group / row group of this wave
    const int HW = p.H * p.W;
    const int plane = HW * 4;
    constexpr unsigned OOB = 0x80000000u;
    constexpr int NQ = SIX ? 3 : 1;
    constexpr int OFF_W = 2 * XB;
    typedef __attribute__((address_space(3))) void* lds_ptr;
    // XCD-aware block -> (spatial tile, cout tile) map.  Workgroups are dealt round-robin over the 8 XCDs (blocks L and L + 8 share
    // an L2): XCD x works through its own contiguous band of spatial tiles (row-major: halo rows are re-read from that L2) with the
    // cout tiles of one spatial tile in consecutive slots, so the second .. fourth read of an input tile hits L2, not the fabric.
    const int b = blockIdx.z;
    int sp, ct;
    if (p.xcd_map && (p.ntiles & 7) == 0) {
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        sp = xcd * (p.ntiles >> 3) + slot / p.ctiles;
        ct = slot % p.ctiles;
    } else {
        sp = blockIdx.x / p.ctiles;
        ct = blockIdx.x % p.ctiles;
    }
    const int row0 = (sp / p.tiles_x) * G::TRW, col0 = (sp % p.tiles_x) * TC;

    // ---- staging entries (two per thread): k = 0: k half wave >> 2, entries 0..255; k = 1: entries 256..339 of k half
    // (wave >> 1) & 1 for waves 0..3 (so that the k half, hence the channel, is uniform over a wave)
    // (7x7: three per thread, entry e = tid + 512 k of the 2 x 532; the k half is NOT uniform over a wave there, so it rides in the
    // per-lane offset and the form has no load-side affine)
    constexpr int NEK = G::NEK;
    int ekh[NEK], eidx[NEK];
    bool fin[NEK];
    unsigned fo[NEK];
    if constexpr (SPECIAL) {
        ekh[0] = wave >> 2; eidx[0] = tid & 255; fin[0] = true;
        ekh[1] = (wave >> 1) & 1; eidx[1] = 256 + (tid & 127); fin[1] = wave < 4 && eidx[1] < EPK;
    } else {
#pragma unroll
        for (int k = 0; k < NEK; ++k) {
            const int e = tid + 512 * k;
            ekh[k] = e >= EPK;
            eidx[k] = e - ekh[k] * EPK;
            fin[k] = e < 2 * EPK;
        }
    }
    bool fok[NEK];
#pragma unroll
    for (int k = 0; k < NEK; ++k) {
        const int r = eidx[k] / XC, c = eidx[k] % XC;
        const int gr = row0 + r - PAD, gc = col0 + c - PAD;
        fok[k] = fin[k] && gr >= 0 && gr < p.H && gc >= 0 && gc < p.W;
        // blocked input ([Cin/8][H][W][8], cwfa_conv_opts.in_blocked8): the 32-byte entry of the pixel; its channel block rides
        // in the scalar offset
        fo[k] = !fok[k] ? OOB : p.o.in_blocked8 ? (unsigned)((gr * p.W + gc) * 32) : (unsigned)((gr * p.W + gc) * 4);
        if constexpr (!SPECIAL) fo[k] = !fok[k] ? OOB : fo[k] + (unsigned)(ekh[k] * 8) * (unsigned)plane;    // (either layout: 8 planes)
    }
    const int xbytes = p.Cin * plane;                   // channels >= Cin: out of range, 0.0
    const auto rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x + (int64_t)b * p.x_bs), 0, xbytes, 0x00020000);
    const auto ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ADD ? p.o.in_add + (int64_t)b * p.o.in_add_bs : p.x), 0,
                                                      ADD ? xbytes : 0, 0x00020000);
    const bool has_aff = p.o.in_scale != nullptr;
    // the load-side affine tables are read through the SCALAR cache (constant address space; they were written by an
    // earlier kernel and the channel index is uniform over a wave): as vector loads each cost a full memory round trip
    // in the middle of the step
    typedef const float __attribute__((address_space(4))) cfloat;
    cfloat* sc = has_aff ? (cfloat*)(p.o.in_scale + (int64_t)b * p.o.in_affine_bs) : nullptr;
    cfloat* sh = has_aff ? (cfloat*)(p.o.in_shift + (int64_t)b * p.o.in_affine_bs) : nullptr;

    auto load_entry = [&](auto kc, float (&xv)[8], float (&av)[8], int chunk) {
        constexpr int k = decltype(kc)::value;
        const int ch0 = chunk * 16 + (SPECIAL ? ekh[k] * 8 : 0);        // (otherwise the k half is in the lane's offset)
        if (KS == 3 && !ADD && p.o.in_blocked8) {                  // (uniform) channel block ch0 / 8 = 8 planes' worth of bytes each (the
                                                        // load-side affine is per channel: unchanged)
            const f32x4 lo4 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, fo[k], ch0 * plane, 0));
            const f32x4 hi4 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, fo[k], ch0 * plane + 16, 0));
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                xv[j] = lo4[j];
                xv[4 + j] = hi4[j];
            }
            return;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) xv[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, fo[k], (ch0 + j) * plane, 0));
        if constexpr (ADD) {
#pragma unroll
            for (int j = 0; j < 8; ++j) av[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ra, fo[k], (ch0 + j) * plane, 0));
        }
    };
    auto store_entry = [&](auto kc, const float (&xv)[8], const float (&av)[8], int chunk, int buf) {
        constexpr int k = decltype(kc)::value;
        if (SPECIAL && k == 1 && !fin[1]) return;
        const int ch0 = chunk * 16 + ekh[k] * 8;          // wave-uniform (two-entry map): the affine tables are read through scalar loads
        bf16x8 pc[3];
        float scv[8], shv[8];
        if (has_aff) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int ch = ch0 + j < p.Cin ? ch0 + j : p.Cin - 1;
                scv[j] = ch0 + j < p.Cin ? sc[ch] : 0.f;
                shv[j] = ch0 + j < p.Cin ? sh[ch] : 0.f;
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float v = xv[j];
            if (has_aff) v = fok[k] ? v * scv[j] + shv[j] : 0.f;       // padding stays zero
            if constexpr (ADD) v += av[j];
            __bf16 a1, a2 = (__bf16)0.f, a3 = (__bf16)0.f;
            split3<SIX>(v, a1, a2, a3);
            pc[0][j] = a1; pc[1][j] = a2; pc[2][j] = a3;
        }
        char* dst = lds + buf * XB + ekh[k] * KHB + eidx[k] * 16;
        if (SPECIAL || fin[k]) {
#pragma unroll
            for (int q = 0; q < NQ; ++q) *reinterpret_cast<bf16x8*>(dst + q * XPB) = pc[q];
        }
    };

    // ---- weight slices by LDS-DMA: WSL / 1024 wave instructions per slice, a quarter per wave
    // 1 KB pieces per slice (12 / 24 / 48), dealt round-robin over the 8 waves; plain-bf16 mode reads only the leading piece plane of a
    // slice ([piece][k group][CT][8]: its first third), so only that is fetched
    constexpr int NPC = (SIX ? WSL : WSL / 3) / 1024;
    const int64_t wtile = (int64_t)ct * p.nsteps * WSL;
    const char* wbase = reinterpret_cast<const char*>(p.wp) + wtile;
    const auto rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(static_cast<const void*>(wbase)), 0, p.nsteps * WSL, 0x00020000);
    auto dma_w = [&](int slice, int slot) {              // slices past the end: out of range, zeros, never read
        sfor<(NPC + 7) / 8>([&](auto ic_) {
            constexpr int i = decltype(ic_)::value;
            const int piece = i * 8 + wave;
            if (piece < NPC) {                           // wave-uniform
                char* dst = lds + OFF_W + slot * WSL + piece * 1024;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr)dst, 16, (unsigned)(lane * 16), slice * WSL + piece * 1024, 0, 0);
            }
        });
    };

    const int alane = OFF_W + (g * CT + wm * MPW * 16 + c16) * 16;        // + slot*WSL + (q*4*CT + mt*16)*16
    const int blane = (g & 1) * KHB + (wn * RW * XC + c16) * 16;             // + tap_off + q*XPB + ((nt>>1)*XC + 16*(nt&1))*16
    const bool sel = (g >> 1) != 0;

    f32x4 acc[MPW][NT];
#pragma unroll
    for (int mt = 0; mt < MPW; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    constexpr int MH = MPW < 2 ? MPW : MPW == 3 ? 3 : 2;   // m-tiles whose A fragments are held at a time (register budget)
    bf16x8 A[MH][3], Bq[2][3];
    float xa[NEK][8], xb[NEK][8], aa[NEK][8], ab[NEK][8];    // staged entries of the next even / odd chunk (+ skip tensor)

    auto read_a = [&](int abase, int m0) {
#pragma unroll
        for (int mt = 0; mt < MH; ++mt)
#pragma unroll
            for (int q = 0; q < NQ; ++q) A[mt][q] = *reinterpret_cast<const bf16x8*>(lds + abase + (q * 4 * CT + (m0 + mt) * 16) * 16);
    };
    auto read_b = [&](int set, int bbase, int nt) {
#pragma unroll
        for (int q = 0; q < NQ; ++q)
            Bq[set][q] = *reinterpret_cast<const bf16x8*>(lds + bbase + q * XPB + ((nt >> 1) * XC + 16 * (nt & 1)) * 16);
    };
    auto mfma6 = [&](f32x4& c, const bf16x8 (&a)[3], const bf16x8 (&bb)[3]) {
        if constexpr (SIX) {
            MFMA(a[2], bb[0], c);
            MFMA(a[1], bb[1], c);
            MFMA(a[0], bb[2], c);
            MFMA(a[1], bb[0], c);
            MFMA(a[0], bb[1], c);
        }
        MFMA(a[0], bb[0], c);
    };

    // ---------------------------------------------------------------------------------------------- prologue
    sfor<NEK>([&](auto kc) { load_entry(kc, xa[decltype(kc)::value], aa[decltype(kc)::value], 0); });
    sfor<NEK>([&](auto kc) { load_entry(kc, xb[decltype(kc)::value], ab[decltype(kc)::value], 1); });
    dma_w(0, 0);
    sfor<NEK>([&](auto kc) { store_entry(kc, xa[decltype(kc)::value], aa[decltype(kc)::value], 0, 0); });
    asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    read_b(0, blane + (sel ? tap_off<KS, RPW>(0, 1) : tap_off<KS, RPW>(0, 0)), 0);  // step 0: taps 0 | 1 of chunk 0
    // ONE step as the body of a rolled loop (the accumulators are loop-carried values: unrolling the nine steps of a period
    // makes the register allocator split their live ranges and keep copies); P = position in the 9-step period
    int sl = 0, P = 0, ce = 0;                           // ring slot of this step's slice; period position; even chunk of the period
    for (int step = 0; step < p.nrun; ++step) {
        const int co = ce + 1;                            // odd chunk (past the end: all zeros, costs only time)
        // units of this step: even chunk taps (2P, 2P+1) for P < 4; (tap 8 | odd tap 0) at P = 4; odd taps (2P-9, 2P-8) after
        // (7x7: 49 taps, the same pairing with a 49-step period)
        const int uA = 2 * P, uB = 2 * P + 1;
        const int bufA = uA >= NTAP, tA = uA - NTAP * bufA, bufB = uB >= NTAP, tB = uB - NTAP * bufB;
        const int offA = bufA * XB + ((tA / KS) * XC + tA % KS) * 16, offB = bufB * XB + ((tB / KS) * XC + tB % KS) * 16;
        // staging schedule inside a period: the odd buffer is free from step 0 (stores SB + k of the entries loaded in the period
        // before), the even one right after the step that pairs tap NTAP-1 with the odd chunk's tap 0 (stores SA + k, loads LA + k
        // a few steps earlier); the odd chunk after next is loaded at LB + k
        constexpr int SB = 0, SA = KS == 3 ? 5 : 26, LA = KS == 3 ? 2 : 22, LB = KS == 3 ? 6 : 44;       // (+ k < NEK: all below NTAP)
        static_assert(LB + NEK - 1 < NTAP && LA + NEK - 1 < SA, "staging schedule");
        // -- order matters: in this rolled loop the compiler cannot count the vector-memory operations between a staging load
        // and its use, so it waits for ALL of them (vmcnt(0)) before the split below: that must come BEFORE this step issues
        // its own DMA and loads, when everything older has long landed (after them it cost 2 - 9 thousand cycles per step).
        // the vector-heavy part of staging: split + LDS stores of one entry of the chunk after next
        // (written out per entry: behind a generic lambda the staged-entry arrays stopped being promoted to registers)
        if (P == SB) store_entry(ic<0>{}, xb[0], ab[0], co, 1);
        if (P == SB + 1) store_entry(ic<1>{}, xb[1], ab[1], co, 1);
        if constexpr (NEK > 2) {
            if (P == SB + 2) store_entry(ic<NEK - 1>{}, xb[NEK - 1], ab[NEK - 1], co, 1);
        }
        if (P == SA) store_entry(ic<0>{}, xa[0], aa[0], ce + 2, 0);
        if (P == SA + 1) store_entry(ic<1>{}, xa[1], aa[1], ce + 2, 0);
        if constexpr (NEK > 2) {
            if (P == SA + 2) store_entry(ic<NEK - 1>{}, xa[NEK - 1], aa[NEK - 1], ce + 2, 0);
        }
        FENCE();
        // first A fragments (the step's first B fragments were requested before the barrier of the step before)
        const int abase = alane + sl * WSL;
        const int bbase = blane + (sel ? offB : offA);
        read_a(abase, 0);
        FENCE();
#pragma unroll
        for (int m0 = 0; m0 < MPW; m0 += MH) {
            if (m0 > 0) {
                read_a(abase, m0);
                read_b(0, bbase, 0);
                FENCE();
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                if (nt + 1 < NT) read_b((nt + 1) & 1, bbase, nt + 1);
                FENCE();
#pragma unroll
                for (int mt = 0; mt < MH; ++mt) {
                    mfma6(acc[m0 + mt][nt], A[mt], Bq[nt & 1]);
                    FENCE();                 // keep the six products of a tile back to back (accumulator forwarding)
                }
                // this step's staging loads and the next weight slice are issued from INSIDE the MFMA stream (after the first
                // n-tiles): issued in a burst right behind the barrier, all eight waves stood in their issue cost (~100 cycles per
                // DMA instruction) at once with the matrix pipe idle; here the SIMD partner's MFMAs run meanwhile.  Loads BEFORE
                // the DMA: they overwrite loop-carried registers, so the compiler waits for every older vector-memory
                // operation first -- which must not include a DMA issued a moment ago.
                if (m0 == 0 && nt == 0) {
                    if (P == LA) load_entry(ic<0>{}, xa[0], aa[0], ce + 2);
                    if (P == LA + 1) load_entry(ic<1>{}, xa[1], aa[1], ce + 2);
                    if constexpr (NEK > 2) {
                        if (P == LA + 2) load_entry(ic<NEK - 1>{}, xa[NEK - 1], aa[NEK - 1], ce + 2);
                    }
                    if (P == LB) load_entry(ic<0>{}, xb[0], ab[0], co + 2);
                    if (P == LB + 1) load_entry(ic<1>{}, xb[1], ab[1], co + 2);
                    if constexpr (NEK > 2) {
                        if (P == LB + 2) load_entry(ic<NEK - 1>{}, xb[NEK - 1], ab[NEK - 1], co + 2);
                    }
                    FENCE();
                }
                if (m0 == 0 && nt == 1) {
                    dma_w(step + 1, sl ^ 1);
                    FENCE();
                }
            }
        }
        // first B fragments of the next step: the input tiles it reads were completed at least two barriers ago
        {
            const int nP = P == NTAP - 1 ? 0 : P + 1;
            const int nA = 2 * nP, nB = 2 * nP + 1;
            const int nbufA = nA >= NTAP, ntA = nA - NTAP * nbufA, nbufB = nB >= NTAP, ntB = nB - NTAP * nbufB;
            const int noffA = nbufA * XB + ((ntA / KS) * XC + ntA % KS) * 16, noffB = nbufB * XB + ((ntB / KS) * XC + ntB % KS) * 16;
            read_b(0, blane + (sel ? noffB : noffA), 0);
            FENCE();
        }
        // the next slice (the youngest vector-memory operation of the step) has landed
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // (LDS: everything but the three B-fragment reads just issued is done, the staging stores of this step included)
        if constexpr (SIX) asm volatile("s_waitcnt lgkmcnt(3)" ::: "memory");
        else asm volatile("s_waitcnt lgkmcnt(1)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        sl ^= 1;
        if (++P == NTAP) {
            P = 0;
            ce += 2;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // DMAs / loads past the end

    // ---------------------------------------------------------------------------------------------- epilogue
    // accumulator register r of tile (mt, nt): channel ct*CT + (wm*MPW + mt)*16 + 4g + r, pixel (row0 + 4*wn + nt/2, col0 + 16*(nt&1) + c16).
    // Buffer stores: the descriptor ends at channel Cout and a pixel outside the image carries an out-of-range offset.
    const auto ry = __builtin_amdgcn_make_buffer_rsrc(p.y + (int64_t)b * p.y_bs, 0, p.Cout * plane, 0x00020000);
    const auto rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.o.residual ? p.o.residual + (int64_t)b * p.o.res_bs : p.y), 0,
                                                      p.o.residual ? p.Cout * plane : 0, 0x00020000);
    const auto rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.o.bias ? p.o.bias : p.y), 0, p.o.bias ? p.Cout * 4 : 0, 0x00020000);
    const bool alpha_pc = p.o.prelu_per_channel != 0;          // (uniform) one PReLU slope per output channel
    const float alpha = (p.o.prelu_alpha && !alpha_pc) ? *p.o.prelu_alpha : 0.f;
    const auto rpa = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(alpha_pc ? p.o.prelu_alpha : p.y), 0, alpha_pc ? p.Cout * 4 : 0, 0x00020000);
    const int cwave = ct * CT + wm * MPW * 16;      // uniform: rides in the scalar offset
    const unsigned glane = (unsigned)(4 * g) * (unsigned)plane;                           // the lane group's channel offset
    if constexpr (ACT1 == EPI_COUPLE) {
        // The bank's rows were interleaved at pack time (cwfa_couple_rows) so that this lane holds s_j AND t_j of its pixels:
        //   MPW = 1: pair j = 8 wm + 2 g + h  ->  s in register h, t in register h + 2 of the one m-tile
        //   MPW = 2: pair j = 16 wm + 4 g + r ->  s in m-tile 0, t in m-tile 1, register r
        // x / y descriptors end at channel n: pairs >= n read 0.0 and their stores are dropped (their s is 0: zero rows).
        static_assert(MPW <= 2 && !ADD, "coupling epilogue: narrow tilings only");
        constexpr int NP = MPW == 1 ? 2 : 4;                                       // pairs per lane and n-tile
        const cwfa_couple& cp = p.cp;
        const auto cx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(cp.x + (int64_t)b * cp.x_bs), 0, cp.n * plane, 0x00020000);
        const auto cy = __builtin_amdgcn_make_buffer_rsrc(cp.y + (int64_t)b * cp.y_bs, 0, cp.n * plane, 0x00020000);
        const unsigned jlane = (unsigned)((MPW == 1 ? 2 : 4) * g) * (unsigned)plane;
        const int jwave = wm * (MPW == 1 ? 8 : 16);
        float bs[NP], bt[NP];
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const int rs = k, rt = MPW == 1 ? k + 2 : 16 + k;                        // packed rows of s_j / t_j relative to cwave + 4g
            bs[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rb, (unsigned)(16 * g), (cwave + rs) * 4, 0));
            bt[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rb, (unsigned)(16 * g), (cwave + rt) * 4, 0));
        }
        float ssum = 0.f;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int row = row0 + wn * RW + (nt >> 1), col = col0 + 16 * (nt & 1) + c16;
            const bool ok = row < p.H && col < p.W;
            const unsigned po = ok ? (unsigned)((row * p.W + col) * 4) + jlane : OOB;
            float xv[NP];
#pragma unroll
            for (int k = 0; k < NP; ++k) xv[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(cx, po, (jwave + k) * plane, 0));
#pragma unroll
            for (int k = 0; k < NP; ++k) {
                const float sr = acc[0][nt][k] + bs[k];
                float tr;
                if constexpr (MPW == 1) tr = acc[0][nt][k + 2] + bt[k];
                else tr = acc[MPW - 1][nt][k] + bt[k];
                const float sv = soft_clamp(sr * cp.pre_scale, cp.clamp_kind, cp.clamp);
                const float tv = tr * cp.pre_scale;
                const float yv = cp.rev ? (xv[k] - tv) * __expf(-sv) : __expf(sv) * xv[k] + tv;     // |s| <= clamp: v_exp_f32, rel. error ~2e-7
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, yv), cy, po, (jwave + k) * plane, 0);
                ssum += ok ? sv : 0.f;
            }
            FENCE();
        }
        if (cp.logdet) {                                  // block sum -> one float64 atomic per block
            const double ws = cwfa_wave_sum((double)ssum);
            double* red = reinterpret_cast<double*>(lds);  // the operand buffers are dead (every wave passed the loop's last barrier)
            __builtin_amdgcn_s_barrier();
            if (lane == 0) red[wave] = ws;
            __syncthreads();
            if (tid == 0) {
                double tot = 0.0;
#pragma unroll
                for (int w = 0; w < 8; ++w) tot += red[w];
                atomicAdd(&cp.logdet[b], cp.rev ? -tot : tot);
            }
        }
        return;
    }
    const bool outb = p.o.out_blocked8 != 0;           // (uniform) y channel-blocked [Cout/8][H][W][8]: the lane's four channels = 16 bytes
    // cwfa_conv_opts.out_stats: per-channel (sum, sum of squares) of the output for the BatchNorm that follows (unet.py:99-107):
    // lane sums over its n-tiles, 16-lane DPP sums (pixels of a row), the block's two row halves through LDS (the operand
    // buffers are dead), then one float64 atomic pair per channel and block
    const bool want_stats = ACT1 != EPI_RUNTIME && p.o.out_stats != nullptr;
    float* red = reinterpret_cast<float*>(lds);        // [wn WN][CT][2]
#pragma unroll
    for (int mt = 0; mt < MPW; ++mt) {
        float bias[4];
        float st1[4] = {0.f, 0.f, 0.f, 0.f}, st2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; ++r)
            bias[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rb, (unsigned)(16 * g), (cwave + mt * 16 + r) * 4, 0));
        float al[4] = {alpha, alpha, alpha, alpha};
        if (alpha_pc) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                al[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rpa, (unsigned)(16 * g), (cwave + mt * 16 + r) * 4, 0));
        }
        // run-time epilogue: the residual of n-tile nt + RLA is requested while n-tile nt is finished (one tile at a time -- as the
        // FENCE below orders it -- every n-tile stood in the full latency of its four loads: 160 -> 1xx us for 64 -> 64 at 512 x 512)
        constexpr int RLA = NT < 4 ? NT : 4;
        float rv[ACT1 == EPI_RUNTIME ? NT : 1][4];
        auto load_res = [&](int nt) {
            const int row = row0 + wn * RW + (nt >> 1), col = col0 + 16 * (nt & 1) + c16;
            const unsigned po = (row < p.H && col < p.W) ? (unsigned)((row * p.W + col) * 4) + glane : OOB;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                rv[ACT1 == EPI_RUNTIME ? nt : 0][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rr, po, (cwave + mt * 16 + r) * plane, 0));
        };
        if constexpr (ACT1 == EPI_RUNTIME) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) rv[nt][r] = 0.f;
            if (p.o.residual) {
#pragma unroll
                for (int nt = 0; nt < RLA; ++nt) load_res(nt);
            }
            FENCE();
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int row = row0 + wn * RW + (nt >> 1), col = col0 + 16 * (nt & 1) + c16;
            if constexpr (ACT1 == EPI_RUNTIME) {
                if (nt + RLA < NT && p.o.residual) load_res(nt + RLA);
                FENCE();
            }
            if constexpr (ACT1 != EPI_RUNTIME) {
                if (outb) {
                    f32x4 o4;
#pragma unroll
                    for (int r = 0; r < 4; ++r) o4[r] = act_of<ACT1>(acc[mt][nt][r] + bias[r], al[r]);
                    const unsigned pb = (row < p.H && col < p.W) ? (unsigned)(((g >> 1) * HW + row * p.W + col) * 32 + (g & 1) * 16) : OOB;
                    cwfa_buffer_store_b128(__builtin_bit_cast(cwfa_u32x4, o4), ry, pb, (cwave + mt * 16) * plane);   // (+ wait states: common.h)
                    continue;
                }
            }
            const unsigned po = (row < p.H && col < p.W) ? (unsigned)((row * p.W + col) * 4) + glane : OOB;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int soff = (cwave + mt * 16 + r) * plane;
                float v = acc[mt][nt][r] + bias[r];
                if constexpr (ACT1 == EPI_RUNTIME) {
                    v = cwfa_act(v, p.o.act, al[r]);
                    v += rv[nt][r];
                    v = cwfa_act(v, p.o.act2, al[r]);
                } else {
                    v = act_of<ACT1>(v, al[r]);
                    if (want_stats) {
                        const float vm = po != OOB ? v : 0.f;
                        st1[r] += vm;
                        st2[r] += vm * vm;
                    }
                }
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ry, po, soff, 0);
            }
            FENCE();                 // keep the results from being staged in registers all at once
        }
        if constexpr (ACT1 != EPI_RUNTIME) {
            if (want_stats) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float a1 = cwfa_row16_sum(st1[r]), a2 = cwfa_row16_sum(st2[r]);
                    if (c16 == 0) {
                        const int ch = (wm * MPW + mt) * 16 + 4 * g + r;
                        red[(wn * CT + ch) * 2] = a1;
                        red[(wn * CT + ch) * 2 + 1] = a2;
                    }
                }
            }
        }
    }
    if constexpr (ACT1 != EPI_RUNTIME) {
        if (want_stats) {
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            const int ch = ct * CT + tid;
            if (tid < CT && ch < p.Cout) {
                double s1 = 0.0, s2 = 0.0;
#pragma unroll
                for (int k = 0; k < WN; ++k) {
                    s1 += (double)red[(k * CT + tid) * 2];
                    s2 += (double)red[(k * CT + tid) * 2 + 1];
                }
                atomicAdd(&p.o.out_stats[2 * ch], s1);
                atomicAdd(&p.o.out_stats[2 * ch + 1], s2);
            }
        }
    }
}

// packed image: [cout tile][step][piece 3][k group 4][CT][8] of bf16; group g of step s = unit u = 2s + (g >> 1) =
// (chunk u / 9, tap u % 9), k half g & 1: element j = w[co][chunk*16 + (g&1)*8 + j][tap] (0 beyond Cout / Cin / the last unit)
template <int CT>
__global__ __launch_bounds__(256) void split3x3_pack_kernel(const float* __restrict__ w, uint4* __restrict__ out, int Cout, int Cin,
                                                            int nchunks, int nsteps, int64_t total, int ntap) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;       // over [ctile][step][g 4][co CT]
    if (i >= total) return;
    const int col = (int)(i % CT), g = (int)((i / CT) % 4), s = (int)((i / (4 * CT)) % nsteps), ctile = (int)(i / ((int64_t)4 * CT * nsteps));
    const int u = 2 * s + (g >> 1), chunk = u / ntap, tap = u % ntap, co = ctile * CT + col;
    unsigned short pc[3][8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int ci = chunk * 16 + (g & 1) * 8 + j;
        float v = 0.f;
        if (co < Cout && ci < Cin && chunk < nchunks) v = w[((int64_t)co * Cin + ci) * ntap + tap];
        __bf16 a1, a2, a3;
        split3<true>(v, a1, a2, a3);
        pc[0][j] = __builtin_bit_cast(unsigned short, a1);
        pc[1][j] = __builtin_bit_cast(unsigned short, a2);
        pc[2][j] = __builtin_bit_cast(unsigned short, a3);
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        uint4 v4;
        v4.x = pc[q][0] | ((unsigned)pc[q][1] << 16);
        v4.y = pc[q][2] | ((unsigned)pc[q][3] << 16);
        v4.z = pc[q][4] | ((unsigned)pc[q][5] << 16);
        v4.w = pc[q][6] | ((unsigned)pc[q][7] << 16);
        out[(((int64_t)ctile * nsteps + s) * 3 + q) * 4 * CT + g * CT + col] = v4;
    }
}

// (m-tiles per wave, channel groups per block) by bank size -- packing and launch must agree.  WM = 4: 64 / 128 / 256 channels per
// block on 8-row tiles (64: also 16-row).  The narrow tilings (16-row tile, no load-side prologue) fit the small banks WITHOUT zero
// rows and with more m-tiles per B fragment: 16 = (1,1); 32 = (2,1); 48 = (3,1): the 64 -> 48 output convolutions of the sub-networks
// (on the 64-channel tiling a B fragment fed ONE m-tile and a quarter of the MFMAs ran on zero rows); 96 = (3,2): the 64 -> 96 ones
// (a quarter of the 128-channel tiling was zero rows).
inline int mpw_of(int Cout) { return Cout > 128 ? 4 : Cout > 96 ? 2 : Cout > 64 ? 3 : Cout > 48 ? 1 : Cout > 32 ? 3 : Cout > 16 ? 2 : 1; }
inline int wm_of(int Cout) { return Cout > 96 ? 4 : Cout > 64 ? 2 : Cout > 48 ? 4 : 1; }
inline int ct_of(int Cout) { return 16 * mpw_of(Cout) * wm_of(Cout); }
inline int nsteps_of(int Cin, int ntap = 9) { return ntap * (((Cin + 15) / 16 + 1) / 2); }   // whole periods of two 16-channel chunks

template <int MPW, bool SIX, bool ADD, int ACT1, int KS = 3, int RPW = 4, int WM = 4>
int launch(const SParams& p, hipStream_t stream) {
    typedef Geo<MPW, KS, RPW, WM> G;
    constexpr int TRW = XG<KS, RPW>::TRW;
    auto kern = &conv3x3_split_kernel<MPW, SIX, ADD, ACT1, KS, RPW, WM>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS);
        if (e != hipSuccess) {
            cwfa_set_error("cwfa_conv3x3_split_f32: hipFuncSetAttribute(%d bytes LDS): %s", G::LDS, hipGetErrorString(e));
            return CWFA_E_HIP;
        }
        attr_set = true;
    }
    const int tiles_y = (p.H + TRW - 1) / TRW, ctiles = (p.Cout + G::CT - 1) / G::CT;
    SParams q = p;
    q.ntiles = p.tiles_x * tiles_y;
    q.ctiles = ctiles;
    q.xcd_map = g_cwfa_split_xcd_map;
    if ((int64_t)q.ntiles * ctiles >= (1ll << 31)) {
        cwfa_set_error("cwfa_conv3x3_split_f32: grid too large");
        return CWFA_E_SHAPE;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)(q.ntiles * ctiles), 1, p.B), dim3(512), G::LDS, stream, q);
    CWFA_LAUNCH_CHECK("cwfa_conv3x3_split_f32");
    return CWFA_OK;
}

template <int MPW, bool SIX>
int launch_epi(const SParams& p, hipStream_t st) {
    const bool plain = !p.o.residual && p.o.act2 == CWFA_ACT_NONE;
    if (p.o.in_add) {
        if (plain && p.o.act == CWFA_ACT_PRELU) return launch<MPW, SIX, true, CWFA_ACT_PRELU>(p, st);
        return launch<MPW, SIX, true, EPI_RUNTIME>(p, st);
    }
    if (plain && p.o.act == CWFA_ACT_PRELU) return launch<MPW, SIX, false, CWFA_ACT_PRELU>(p, st);
    if (plain && p.o.act == CWFA_ACT_NONE) return launch<MPW, SIX, false, CWFA_ACT_NONE>(p, st);
    return launch<MPW, SIX, false, EPI_RUNTIME>(p, st);
}

}  // namespace

// packed row r of a coupling bank <- source row of the torch weight [2n, Cin, 3, 3] (s_j = row j, t_j = row n + j), -1 = zero row
static int couple_row_source(int n, int r) {
    int j, which;
    if (n <= 32) {                         // MPW = 1: row = 16 wm + 4 g + (h + 2 which), j = 8 wm + 2 g + h
        const int wm = r >> 4, g = (r >> 2) & 3, q = r & 3;
        j = 8 * wm + 2 * g + (q & 1);
        which = q >> 1;
    } else {                               // MPW = 2: row = 16 (2 wm + which) + 4 g + r4, j = 16 wm + 4 g + r4
        const int wm = r >> 5;
        which = (r >> 4) & 1;
        j = 16 * wm + (r & 15);
    }
    return j < n ? which * n + j : -1;
}

extern "C" int cwfa_couple_rows(int n, int* rows) {
    if (n <= 0 || n > 64) return -1;
    const int total = n <= 32 ? 64 : 128;
    if (rows)
        for (int r = 0; r < total; ++r) rows[r] = couple_row_source(n, r);
    return total;
}

extern "C" int cwfa_conv3x3_split_couple_f32(const float* x, const void* w_packed, const float* bias_rows, int B, int Cin, int H, int W,
                                             int64_t x_bs, const cwfa_couple* cp, void* stream) {
    CWFA_REQUIRE(B >= 0 && Cin > 0 && H >= 0 && W >= 0, CWFA_E_INVAL, "cwfa_conv3x3_split_couple_f32: bad size");
    CWFA_REQUIRE(cp, CWFA_E_INVAL, "cwfa_conv3x3_split_couple_f32: null coupling descriptor");
    CWFA_REQUIRE(cp->n > 0 && cp->n <= 64, CWFA_E_SHAPE, "cwfa_conv3x3_split_couple_f32: 1 <= n <= 64 (got %d)", cp->n);
    if (B == 0 || H == 0 || W == 0) return CWFA_OK;
    CWFA_REQUIRE(x && w_packed && cp->x && cp->y, CWFA_E_INVAL, "cwfa_conv3x3_split_couple_f32: null pointer");
    CWFA_REQUIRE(cwfa_aligned16(w_packed), CWFA_E_ALIGN, "cwfa_conv3x3_split_couple_f32: packed weights must be 16-byte aligned");
    CWFA_REQUIRE(cp->clamp_kind >= CWFA_CLAMP_NONE && cp->clamp_kind <= CWFA_CLAMP_SIGMOID, CWFA_E_INVAL,
                 "cwfa_conv3x3_split_couple_f32: bad clamp kind %d", cp->clamp_kind);
    SParams p{};
    p.x = x; p.wp = w_packed; p.y = cp->y;
    p.B = B; p.Cin = Cin; p.H = H; p.W = W; p.x_bs = x_bs; p.y_bs = cp->y_bs;
    p.Cout = cwfa_couple_rows(cp->n, nullptr);
    p.o.bias = bias_rows;
    p.o.in_blocked8 = cp->in_blocked8;
    CWFA_REQUIRE(!cp->in_blocked8 || (Cin % 8 == 0 && cwfa_aligned16(x) && (x_bs & 3) == 0), CWFA_E_ALIGN,
                 "cwfa_conv3x3_split_couple_f32: blocked input needs Cin %% 8 == 0 and 16-byte alignment");
    p.cp = *cp;
    p.nchunks = (Cin + 15) / 16;
    p.nsteps = nsteps_of(Cin);
    p.nrun = p.nsteps - ((p.nchunks & 1) ? 4 : 0);
    p.tiles_x = (W + TC - 1) / TC;
    CWFA_REQUIRE((int64_t)(Cin + 64) * H * W * 4 < (1ll << 31) && (int64_t)(cp->n + 64) * H * W * 4 < (1ll << 31), CWFA_E_SHAPE,
                 "cwfa_conv3x3_split_couple_f32: one sample's input / active half must stay below 2 GiB");
    CWFA_REQUIRE((int64_t)p.tiles_x * ((H + TR - 1) / TR) < (1ll << 31) && B <= 65535, CWFA_E_SHAPE,
                 "cwfa_conv3x3_split_couple_f32: grid too large");
    hipStream_t st = (hipStream_t)stream;
    const bool six = g_cwfa_split_products != 1;
    if (p.Cout == 64) {
        if (g_cwfa_split_rows16 && H > 8)       // 16-row tiles (no load-side prologue here by construction)
            return six ? launch<1, true, false, EPI_COUPLE, 3, 8>(p, st) : launch<1, false, false, EPI_COUPLE, 3, 8>(p, st);
        return six ? launch<1, true, false, EPI_COUPLE>(p, st) : launch<1, false, false, EPI_COUPLE>(p, st);
    }
    return six ? launch<2, true, false, EPI_COUPLE>(p, st) : launch<2, false, false, EPI_COUPLE>(p, st);
}

extern "C" int64_t cwfa_conv3x3_split_packed_bytes(int Cout, int Cin) {
    if (Cout <= 0 || Cin <= 0) return -1;
    const int ct = ct_of(Cout);
    return (int64_t)((Cout + ct - 1) / ct) * nsteps_of(Cin) * 3 * 4 * ct * 16;
}

extern "C" int cwfa_conv3x3_split_pack_f32(const float* w, void* packed, int Cout, int Cin, void* stream) {
    CWFA_REQUIRE(w && packed, CWFA_E_INVAL, "cwfa_conv3x3_split_pack_f32: null pointer");
    CWFA_REQUIRE(Cout > 0 && Cin > 0, CWFA_E_SHAPE, "cwfa_conv3x3_split_pack_f32: bad shape");
    CWFA_REQUIRE(cwfa_aligned16(packed), CWFA_E_ALIGN, "cwfa_conv3x3_split_pack_f32: packed image must be 16-byte aligned");
    const int mpw = mpw_of(Cout), ct = ct_of(Cout), nchunks = (Cin + 15) / 16, nsteps = nsteps_of(Cin);
    const int64_t total = (int64_t)((Cout + ct - 1) / ct) * nsteps * 4 * ct;
    const dim3 grid((unsigned)((total + 255) / 256));
    uint4* out = reinterpret_cast<uint4*>(packed);
    if (ct == 32) hipLaunchKernelGGL(split3x3_pack_kernel<32>, grid, dim3(256), 0, (hipStream_t)stream, w, out, Cout, Cin, nchunks, nsteps, total, 9);
    else if (ct == 48) hipLaunchKernelGGL(split3x3_pack_kernel<48>, grid, dim3(256), 0, (hipStream_t)stream, w, out, Cout, Cin, nchunks, nsteps, total, 9);
    else if (ct == 96) hipLaunchKernelGGL(split3x3_pack_kernel<96>, grid, dim3(256), 0, (hipStream_t)stream, w, out, Cout, Cin, nchunks, nsteps, total, 9);
    else if (ct == 16) hipLaunchKernelGGL(split3x3_pack_kernel<16>, grid, dim3(256), 0, (hipStream_t)stream, w, out, Cout, Cin, nchunks, nsteps, total, 9);
    else if (mpw == 4) hipLaunchKernelGGL(split3x3_pack_kernel<256>, grid, dim3(256), 0, (hipStream_t)stream, w, out, Cout, Cin, nchunks, nsteps, total, 9);
    else if (mpw == 2) hipLaunchKernelGGL(split3x3_pack_kernel<128>, grid, dim3(256), 0, (hipStream_t)stream, w, out, Cout, Cin, nchunks, nsteps, total, 9);
    else hipLaunchKernelGGL(split3x3_pack_kernel<64>, grid, dim3(256), 0, (hipStream_t)stream, w, out, Cout, Cin, nchunks, nsteps, total, 9);
    CWFA_LAUNCH_CHECK("cwfa_conv3x3_split_pack_f32");
    return CWFA_OK;
}

extern "C" int cwfa_conv3x3_split_f32(const float* x, const void* w_packed, float* y, int B, int Cin, int H, int W, int Cout,
                                      int64_t x_bs, int64_t y_bs, const cwfa_conv_opts* opts, void* stream) {
    CWFA_REQUIRE(B >= 0 && Cin > 0 && Cout > 0 && H >= 0 && W >= 0, CWFA_E_INVAL, "cwfa_conv3x3_split_f32: bad size");
    if (B == 0 || H == 0 || W == 0) return CWFA_OK;
    CWFA_REQUIRE(x && w_packed && y, CWFA_E_INVAL, "cwfa_conv3x3_split_f32: null pointer");
    CWFA_REQUIRE(cwfa_aligned16(w_packed), CWFA_E_ALIGN, "cwfa_conv3x3_split_f32: packed weights must be 16-byte aligned");
    SParams p{};
    p.x = x; p.wp = w_packed; p.y = y;
    p.B = B; p.Cin = Cin; p.H = H; p.W = W; p.Cout = Cout; p.x_bs = x_bs; p.y_bs = y_bs;
    if (opts) p.o = *opts;
    CWFA_REQUIRE(!p.o.upshuffle2 && !p.o.in_cat, CWFA_E_SHAPE, "cwfa_conv3x3_split_f32: upshuffle2 / in_cat are 1x1 features");
    CWFA_REQUIRE(!p.o.in_blocked8 || (Cin % 8 == 0 && cwfa_aligned16(x) && (x_bs & 3) == 0 && !p.o.in_add), CWFA_E_ALIGN,
                 "cwfa_conv3x3_split_f32: blocked input needs Cin %% 8 == 0, 16-byte alignment and no added tensor");
    CWFA_REQUIRE(!p.o.out_blocked8 || (Cout % 8 == 0 && cwfa_aligned16(y) && (y_bs & 3) == 0 && !p.o.residual && p.o.act2 == CWFA_ACT_NONE &&
                                       (p.o.act == CWFA_ACT_NONE || p.o.act == CWFA_ACT_PRELU)),
                 CWFA_E_ALIGN, "cwfa_conv3x3_split_f32: blocked output needs Cout %% 8 == 0, 16-byte alignment and a bias / PReLU epilogue");
    CWFA_REQUIRE(!(p.o.in_scale && !p.o.in_shift), CWFA_E_INVAL, "cwfa_conv3x3_split_f32: in_scale without in_shift");
    CWFA_REQUIRE(!p.o.out_stats || (!p.o.out_blocked8 && !p.o.residual && p.o.act2 == CWFA_ACT_NONE &&
                                    (p.o.act == CWFA_ACT_NONE || p.o.act == CWFA_ACT_PRELU)),
                 CWFA_E_INVAL, "cwfa_conv3x3_split_f32: out_stats needs an NCHW output and a bias / PReLU epilogue");
    CWFA_REQUIRE(p.o.act >= 0 && p.o.act <= CWFA_ACT_RELU && p.o.act2 >= 0 && p.o.act2 <= CWFA_ACT_RELU, CWFA_E_INVAL,
                 "cwfa_conv3x3_split_f32: bad activation");
    CWFA_REQUIRE(!((p.o.act == CWFA_ACT_PRELU || p.o.act2 == CWFA_ACT_PRELU) && !p.o.prelu_alpha), CWFA_E_INVAL,
                 "cwfa_conv3x3_split_f32: PReLU without prelu_alpha");
    p.nchunks = (Cin + 15) / 16;
    p.nsteps = nsteps_of(Cin);
    // an odd number of 16-channel chunks: the last period's odd chunk is all zeros and the steps that pair ONLY its taps (positions 5 .. 8 of
    // the period: odd taps (1,2) (3,4) (5,6) (7,8)) add nothing -- not run (a 6 -> 256 convolution: five steps instead of nine)
    p.nrun = p.nsteps - ((p.nchunks & 1) ? 4 : 0);
    p.tiles_x = (W + TC - 1) / TC;
    const int mpw = mpw_of(Cout);
    CWFA_REQUIRE((int64_t)(Cin + 64) * H * W * 4 < (1ll << 31) && (int64_t)(Cout + 64 * mpw) * H * W * 4 < (1ll << 31) &&
                     (int64_t)(p.nsteps + 2) * 3 * 4 * 64 * mpw * 16 < (1ll << 31), CWFA_E_SHAPE,
                 "cwfa_conv3x3_split_f32: one sample's input / output / one cout tile's weights must stay below 2 GiB");
    CWFA_REQUIRE((int64_t)p.tiles_x * ((H + TR - 1) / TR) < (1ll << 31) && B <= 65535, CWFA_E_SHAPE, "cwfa_conv3x3_split_f32: grid too large");
    hipStream_t st = (hipStream_t)stream;
    const bool six = g_cwfa_split_products != 1;
    const int wm = wm_of(Cout);
    if (wm != 4) {       // narrow tilings (<= 48 or 65 .. 96 outputs): 16-row tile, no load-side prologue, NCHW / blocked input, bias / PReLU / generic epilogue
        CWFA_REQUIRE(!p.o.in_scale && !p.o.in_add && !p.o.out_blocked8, CWFA_E_INVAL,
                     "cwfa_conv3x3_split_f32: banks with <= 48 or 65 .. 96 outputs take no load-side prologue and write NCHW");
        const bool plain = !p.o.residual && p.o.act2 == CWFA_ACT_NONE;
        const int epi = plain && p.o.act == CWFA_ACT_NONE ? 0 : plain && p.o.act == CWFA_ACT_PRELU ? 1 : 2;
#define CWFA_NARROW(M, W_)                                                                                                                   \
    do {                                                                                                                                     \
        if (epi == 0) return six ? launch<M, true, false, CWFA_ACT_NONE, 3, 8, W_>(p, st) : launch<M, false, false, CWFA_ACT_NONE, 3, 8, W_>(p, st);   \
        if (epi == 1) return six ? launch<M, true, false, CWFA_ACT_PRELU, 3, 8, W_>(p, st) : launch<M, false, false, CWFA_ACT_PRELU, 3, 8, W_>(p, st); \
        return six ? launch<M, true, false, EPI_RUNTIME, 3, 8, W_>(p, st) : launch<M, false, false, EPI_RUNTIME, 3, 8, W_>(p, st);                     \
    } while (0)
        if (mpw == 3 && wm == 2) CWFA_NARROW(3, 2);
        if (mpw == 3) CWFA_NARROW(3, 1);
        if (mpw == 2) CWFA_NARROW(2, 1);
        CWFA_NARROW(1, 1);
#undef CWFA_NARROW
    }
    if (mpw == 4) return six ? launch_epi<4, true>(p, st) : launch_epi<4, false>(p, st);
    if (mpw == 2) return six ? launch_epi<2, true>(p, st) : launch_epi<2, false>(p, st);
    // 64-channel tiling: 16-row tiles for the plain bias-only form (the output convolutions of the sub-networks)
    if (g_cwfa_split_rows16 && H > 8 && !p.o.in_scale && !p.o.in_add && !p.o.residual && p.o.act == CWFA_ACT_NONE && p.o.act2 == CWFA_ACT_NONE)
        return six ? launch<2, true, false, CWFA_ACT_NONE, 3, 8, 2>(p, st) : launch<2, false, false, CWFA_ACT_NONE, 3, 8, 2>(p, st);
    // (two m-tiles per wave x two channel groups instead of one x four: the same 64 channels per block and the same packed image, but a
    //  B fragment feeds two m-tiles -- 30 instead of 51 ds_read_b128 per 96 MFMAs)
    // ... and for any other epilogue without a load-side prologue (activation / residual / second activation: the data-gradient and
    // unfused forward convolutions of the sub-networks in training, 64 -> 64): 160 -> ~100 us at 512 x 512
    if (g_cwfa_split_rows16 && H > 8 && !p.o.in_scale && !p.o.in_add)
        return six ? launch<2, true, false, EPI_RUNTIME, 3, 8, 2>(p, st) : launch<2, false, false, EPI_RUNTIME, 3, 8, 2>(p, st);
    return six ? launch_epi<1, true>(p, st) : launch_epi<1, false>(p, st);
}

// ------------------------------------------------------------------------------------------------ 7x7 (ConvNeXt, networks.py:488)
extern "C" int64_t cwfa_conv7x7_split_packed_bytes(int Cout, int Cin) {
    if (Cout <= 0 || Cout > 64 || Cin <= 0) return -1;
    return (int64_t)nsteps_of(Cin, 49) * 3 * 4 * 64 * 16;
}

extern "C" int cwfa_conv7x7_split_pack_f32(const float* w, void* packed, int Cout, int Cin, void* stream) {
    CWFA_REQUIRE(w && packed, CWFA_E_INVAL, "cwfa_conv7x7_split_pack_f32: null pointer");
    CWFA_REQUIRE(Cout > 0 && Cout <= 64 && Cin > 0, CWFA_E_SHAPE, "cwfa_conv7x7_split_pack_f32: 1 <= Cout <= 64");
    CWFA_REQUIRE(cwfa_aligned16(packed), CWFA_E_ALIGN, "cwfa_conv7x7_split_pack_f32: packed image must be 16-byte aligned");
    const int nchunks = (Cin + 15) / 16, nsteps = nsteps_of(Cin, 49);
    const int64_t total = (int64_t)nsteps * 4 * 64;
    hipLaunchKernelGGL(split3x3_pack_kernel<64>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w,
                       reinterpret_cast<uint4*>(packed), Cout, Cin, nchunks, nsteps, total, 49);
    CWFA_LAUNCH_CHECK("cwfa_conv7x7_split_pack_f32");
    return CWFA_OK;
}

extern "C" int cwfa_conv7x7_split_f32(const float* x, const void* w_packed, float* y, int B, int Cin, int H, int W, int Cout,
                                      int64_t x_bs, int64_t y_bs, const cwfa_conv_opts* opts, void* stream) {
    CWFA_REQUIRE(B >= 0 && Cin > 0 && Cout > 0 && Cout <= 64 && H >= 0 && W >= 0, CWFA_E_INVAL, "cwfa_conv7x7_split_f32: bad size (Cout <= 64)");
    if (B == 0 || H == 0 || W == 0) return CWFA_OK;
    CWFA_REQUIRE(x && w_packed && y, CWFA_E_INVAL, "cwfa_conv7x7_split_f32: null pointer");
    CWFA_REQUIRE(cwfa_aligned16(w_packed), CWFA_E_ALIGN, "cwfa_conv7x7_split_f32: packed weights must be 16-byte aligned");
    SParams p{};
    p.x = x; p.wp = w_packed; p.y = y;
    p.B = B; p.Cin = Cin; p.H = H; p.W = W; p.Cout = Cout; p.x_bs = x_bs; p.y_bs = y_bs;
    if (opts) p.o = *opts;
    CWFA_REQUIRE(!p.o.upshuffle2 && !p.o.in_cat && !p.o.in_scale && !p.o.in_add && !p.o.in_blocked8 && !p.o.out_blocked8 && !p.o.residual && !p.o.out_stats &&
                     p.o.act == CWFA_ACT_NONE && p.o.act2 == CWFA_ACT_NONE,
                 CWFA_E_INVAL, "cwfa_conv7x7_split_f32: bias-only epilogue, no load-side prologue, NCHW maps");
    p.nchunks = (Cin + 15) / 16;
    p.nsteps = nsteps_of(Cin, 49);
    p.nrun = p.nsteps;
    p.tiles_x = (W + TC - 1) / TC;
    CWFA_REQUIRE((int64_t)(Cin + 64) * H * W * 4 < (1ll << 31) && (int64_t)(Cout + 64) * H * W * 4 < (1ll << 31), CWFA_E_SHAPE,
                 "cwfa_conv7x7_split_f32: one sample's input / output must stay below 2 GiB");
    CWFA_REQUIRE((int64_t)p.tiles_x * ((H + TR - 1) / TR) < (1ll << 31) && B <= 65535, CWFA_E_SHAPE, "cwfa_conv7x7_split_f32: grid too large");
    hipStream_t st = (hipStream_t)stream;
    // (2 m-tiles per wave x 2 channel groups: a B fragment feeds two m-tiles -- 18 instead of 27 ds_read_b128 per 48 MFMAs)
    return g_cwfa_split_products != 1 ? launch<2, true, false, CWFA_ACT_NONE, 7, 4, 2>(p, st) : launch<2, false, false, CWFA_ACT_NONE, 7, 4, 2>(p, st);
}
