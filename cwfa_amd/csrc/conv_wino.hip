// 3x3 convolution as a 1-D Winograd F(2,3) implicit GEMM on the fp32 matrix cores (v_mfma_f32_32x32x2_f32).
//
// Along W, two adjacent outputs of a 3-tap filter need 4 multiplies instead of 6 (Winograd minimal filtering):
//     input  window d0..d3 (cols 2t-1 .. 2t+2):  V0 = d0-d2, V1 = d1+d2, V2 = d2-d1, V3 = d1-d3
//     filter row g0,g1,g2:                       U0 = g0, U1 = (g0+g1+g2)/2, U2 = (g0-g1+g2)/2, U3 = g2
//     M_i = sum_{ky,ci} U_i[ky][ci][co] * V_i[ci][row+ky][t]
//     out(2t) = M0 + M1 + M2,   out(2t+1) = M1 - M2 - M3
// so per (ky, ci-pair) a wave issues 4 MFMAs (one per Winograd component, lane = tile t = an output PIXEL PAIR) for 64
// output pixels where the direct kernel (conv2d.hip) needs 6: 1.5x fewer matrix instructions for the same result (exact
// in real arithmetic; in fp32 the extra rounding is ~1e-7 relative, tests keep the 1e-4 parity bound).
//
// GEMM roles as in conv2d.hip: M = output channels (A = transformed weights), N = 32 tiles of one image row, K = ci pairs.
// Block = WM x WN waves, wave = MT x 32 channels x (1 row x 64 pixels), 4 accumulator sets.  Per K-chunk the block stages
// V[4][CK][rows+2][32] (input transform applied on the way into LDS, zero padding / load-side affine / skip add included)
// and U[3][4][CK][CT].  The output transform runs on the accumulators in registers; a lane then owns two adjacent pixels
// and stores them as one 8-byte word (256 B contiguous per half-wave).
#include "conv_internal.h"
#include <type_traits>
#include <utility>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#ifndef CWFA_WDEPTH
#define CWFA_WDEPTH 2
#endif

namespace {

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

template <int CK_, int MT_, int WM_, int WN_>
struct WCfg {
    static constexpr int CK = CK_, MT = MT_, WM = WM_, WN = WN_;
    static constexpr int NTHREADS = 64 * WM * WN;
    static constexpr int CT = 32 * MT * WM;          // output channels per block
    static constexpr int TR = WN;                    // output rows per block (one per pixel-wave)
    static constexpr int TCOLS = 64;                 // output columns per block = 32 tiles
    static constexpr int XR = TR + 2;
    static constexpr int VPLANE = CK * XR * 32;      // floats per Winograd component
    static constexpr int VS = 4 * VPLANE;
    static constexpr int US = 12 * CK * CT;
    static constexpr int VPT = VPLANE / NTHREADS;    // (channel,row,tile) positions per thread
    static constexpr int UV = US % (4 * NTHREADS) == 0 ? 4 : 2;   // floats per staged piece of the U panel
    static constexpr int UPT = US / UV / NTHREADS;
    static_assert(US % (UV * NTHREADS) == 0, "U panel is a whole number of pieces per thread");
    static constexpr int BUF = VS + US;              // floats per LDS buffer (two buffers, see wino_mainloop)
    static constexpr int LDS_BYTES = 2 * BUF * 4;
    static_assert(VPLANE % NTHREADS == 0 && NTHREADS % 32 == 0, "staging map assumes whole tile rows per thread stride");
};

struct WParams {
    const float* x;
    const float* wp;
    float* y;
    int B, Cin, H, W, Cout, nchunks, tiles_x, tiles_y;
    int64_t x_bs, y_bs;
    cwfa_conv_opts o;
    const float* w1x1;
    const float* b1x1;
    float* hidden;          // fused layer, training: the hidden map ELU(conv3x3 + b3) is written here as well (NULL: not kept)
    int64_t hidden_bs;
};

#define LSTAMP(k)

struct WTile {
    int wm, wn, kh, l31, ct, b, row0, col0;
};

template <class C>
__device__ __forceinline__ WTile make_wtile(const WParams& p) {
    WTile t;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    t.wm = wave / C::WN;
    t.wn = wave % C::WN;
    t.kh = lane >> 5;
    t.l31 = lane & 31;
    // XCD-aware block -> tile map.  Workgroups are dealt round-robin over the 8 XCDs (blocks b and b+8 share an L2), so
    // block id = 8*slot + xcd: XCD x works through its own contiguous band of spatial tiles (row-major: the halo rows and
    // the column neighbours of a tile are read by the same L2), and the cout tiles of one spatial tile sit in consecutive
    // slots of that XCD, so their re-reads of the same input hit L2 instead of the fabric.
    const int ntiles = p.tiles_x * p.tiles_y, nct = (int)(gridDim.x / ntiles);
    int tile, ct;
    if ((ntiles & 7) == 0) {
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        ct = slot % nct;
        tile = xcd * (ntiles >> 3) + slot / nct;
    } else {
        ct = blockIdx.x % nct;
        tile = blockIdx.x / nct;
    }
    const int tx = tile % p.tiles_x, ty = tile / p.tiles_x;
    t.ct = ct;
    t.b = blockIdx.z;
    t.row0 = ty * C::TR;
    t.col0 = tx * C::TCOLS;
    return t;
}

__device__ __forceinline__ int acc_row(int r, int kh) { return (r & 3) + 8 * (r >> 2) + 4 * kh; }

// ------------------------------------------------------------------------------------------------ main loop
template <class C, bool PRO>
__device__ __forceinline__ void wino_mainloop(const WParams& p, const WTile& t, float* Vs, float* Us,
                                              f32x16 (&acc)[C::MT][4]) {
    const int tid = threadIdx.x;
    const int64_t HW = (int64_t)p.H * p.W;
    constexpr int RSTEP = C::NTHREADS / 32;

    // Staging map: this thread always handles tile column tt = tid % 32; element i covers (channel cloc(i), tile row r(i)).
    // Global reads go through buffer descriptors (one per tensor, per batch sample): the per-lane part of an address is a
    // 32-bit byte offset computed ONCE (voff), the per-chunk part is the scalar soffset, and everything that is padding --
    // rows/columns outside the image, channels >= Cin -- is an out-of-range offset that the hardware range check turns
    // into 0.0 without any vector instruction (cwfa_wino_conv checks the slice is < 2 GiB so OOB + soffset cannot wrap).
    constexpr unsigned OOB = 0x80000000u;
    const int tt = tid & 31;
    unsigned voff[C::VPT][4], vdst[C::VPT], cl4[PRO ? C::VPT : 1];
    unsigned long long rowok[PRO ? C::VPT : 1], colok[PRO ? 4 : 1];     // lane masks (SGPR pairs), load-side affine only
#pragma unroll
    for (int i = 0; i < C::VPT; ++i) {
        const int rc = (tid >> 5) + i * RSTEP;
        const int r = rc % C::XR, c = rc / C::XR;
        const int gr = t.row0 + r - 1;
        const bool rok = gr >= 0 && gr < p.H;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int gc = t.col0 + 2 * tt - 1 + j;
            const bool ok = rok && gc >= 0 && gc < p.W;
            voff[i][j] = ok ? (unsigned)((c * HW + (int64_t)gr * p.W + gc) * 4) : OOB;
            if (PRO && i == 0) colok[j] = __builtin_amdgcn_ballot_w64(gc >= 0 && gc < p.W);
        }
        vdst[i] = (c * C::XR + r) * 32 + tt;
        if constexpr (PRO) {
            cl4[i] = c * 4;
            rowok[i] = __builtin_amdgcn_ballot_w64(rok);
        }
    }
    const bool has_aff = PRO && p.o.in_scale != nullptr, has_add = PRO && p.o.in_add != nullptr;
    const int xbytes = (int)((int64_t)p.Cin * HW * 4);
    const auto rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x + (int64_t)t.b * p.x_bs), 0, xbytes, 0x00020000);
    const auto ra = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(has_add ? p.o.in_add + (int64_t)t.b * p.o.in_add_bs : p.x), 0, has_add ? xbytes : 0, 0x00020000);
    const auto rsc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(has_aff ? p.o.in_scale + (int64_t)t.b * p.o.in_affine_bs : p.x), 0, has_aff ? p.Cin * 4 : 0, 0x00020000);
    const auto rsh = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(has_aff ? p.o.in_shift + (int64_t)t.b * p.o.in_affine_bs : p.x), 0, has_aff ? p.Cin * 4 : 0, 0x00020000);
    const auto rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wp + (int64_t)t.ct * p.nchunks * C::US), 0,
                                                      p.nchunks * C::US * 4, 0x00020000);
    const int chunk_bytes = (int)(C::CK * HW * 4);
    const unsigned uoff = tid * 4 * C::UV;
    auto ldf = [](decltype(rx) r, unsigned vo, int so) { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, vo, so, 0)); };

    constexpr int NP = PRO ? C::VPT : 1;
    f32x4 dr[C::VPT], ar[NP];
    float sr[NP], hr[NP];
    f32x4 ur[C::UPT];

    // Staging is cut into ITEMS (one V element = 4 inputs -> 4 Winograd components, or one 16-byte piece of the U panel).
    // Nothing waits on a load where it is issued; the load-side affine, its masking and the input transform happen when
    // the item is stored to LDS, a whole chunk later.
    auto load_v = [&](int i, int chunk) {
        const int so = chunk * chunk_bytes;
#pragma unroll
        for (int j = 0; j < 4; ++j) dr[i][j] = ldf(rx, voff[i][j], so);
        if constexpr (PRO) {
            if (has_aff) {
                sr[i] = ldf(rsc, cl4[i], chunk * C::CK * 4);
                hr[i] = ldf(rsh, cl4[i], chunk * C::CK * 4);
            }
            if (has_add) {
#pragma unroll
                for (int j = 0; j < 4; ++j) ar[i][j] = ldf(ra, voff[i][j], so);
            }
        }
    };
    auto load_u = [&](int i, int chunk) {
        const int so = (chunk * C::US + i * C::NTHREADS * C::UV) * 4;
        if constexpr (C::UV == 4) {
            ur[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, uoff, so, 0));
        } else {
            const f32x2 v = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rw, uoff, so, 0));
            ur[i][0] = v[0];
            ur[i][1] = v[1];
        }
    };
    auto store_v = [&](int i, int buf) {
        float d[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float v = dr[i][j];
            if constexpr (PRO) {
                if (has_aff) {
                    v = v * sr[i] + hr[i];                              // padding must be zero AFTER the affine
                    asm("v_cndmask_b32 %0, 0, %1, %2" : "=v"(v) : "v"(v), "s"(rowok[i] & colok[j]));
                }
                if (has_add) v += ar[i][j];
            }
            d[j] = v;
        }
        float* dst = Vs + buf * C::BUF + vdst[i];
        dst[0 * C::VPLANE] = d[0] - d[2];
        dst[1 * C::VPLANE] = d[1] + d[2];
        dst[2 * C::VPLANE] = d[2] - d[1];
        dst[3 * C::VPLANE] = d[1] - d[3];
    };
    auto store_u = [&](int i, int buf) {
        if constexpr (C::UV == 4) {
            reinterpret_cast<f32x4*>(Us + buf * C::BUF)[tid + i * C::NTHREADS] = ur[i];
        } else {
            const f32x2 v = {ur[i][0], ur[i][1]};
            reinterpret_cast<f32x2*>(Us + buf * C::BUF)[tid + i * C::NTHREADS] = v;
        }
    };
    constexpr int NITEM = C::VPT + C::UPT;
    auto load_item = [&](auto kc, int chunk) {
        constexpr int k = decltype(kc)::value;
        if constexpr (k < C::VPT) load_v(k, chunk);
        else load_u(k - C::VPT, chunk);
    };
    auto store_item = [&](auto kc, int buf) {
        constexpr int k = decltype(kc)::value;
        if constexpr (k < C::VPT) store_v(k, buf);
        else store_u(k - C::VPT, buf);
    };

#pragma unroll
    for (int m = 0; m < C::MT; ++m)
#pragma unroll
        for (int xi = 0; xi < 4; ++xi)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][xi][r] = 0.f;

    static_assert(C::MT == 1 || C::MT == 2, "U panel layout: one m-tile, or two interleaved");
    const float* ulane0 = Us + t.kh * C::CT + t.wm * 32 * C::MT + C::MT * t.l31;
    const float* vlane0 = Vs + (t.kh * C::XR + t.wn) * 32 + t.l31;

    // Schedule.  Two LDS buffers, ONE barrier per chunk.  On gfx950 the fp32 MFMA and the vector ALU do not co-execute
    // (SQ_VALU_MFMA_COEXEC_CYCLES = 0) and a wave that has its next MFMA ready STARVES its SIMD partner's vector
    // instructions completely (tools/probe/mfma_partner.hip: partner VALU makes no progress until the chain ends), so
    // "one wave stages while the other computes" does not overlap anything: staging VALU only runs in the gaps of the
    // partner's MFMA stream.  Vector work placed INSIDE a wave's own MFMA stream costs its 4 issue cycles and nothing
    // else (tools/probe/mfma_valu.hip: 64 + ~3.2 cycles per VALU per MFMA).  So every wave runs the same stream: the 48
    // k-steps of chunk c, and after every few steps ONE staging item: store item k of chunk c+1 to the other LDS buffer
    // (its loads were issued one chunk ago), then issue its loads for chunk c+2 into the registers just freed.
    // The LDS operands of k-step s+DEPTH are read while the MFMAs of step s issue; every step is fenced for the scheduler.
    // The operand pipeline runs ACROSS the chunk barrier: the barrier sits DEPTH k-steps before the end of a chunk (every
    // store of chunk c+1 is scheduled before it, and a wave's last reads of the current buffer are issued before it), the
    // first reads of chunk c+1 follow it immediately, and the remaining MFMAs of chunk c cover their latency -- with the
    // barrier at the very end both waves of a SIMD sat idle for one LDS round trip per chunk.
    constexpr int NSTEP = 12 * (C::CK / 2), DEPTH = CWFA_WDEPTH, SLOT0 = 1, SLOTD = (NSTEP - DEPTH - 2) / NITEM;
    static_assert(NSTEP % (DEPTH + 1) == 0 && SLOT0 + (NITEM - 1) * SLOTD < NSTEP - DEPTH, "slot ring / staging before the barrier");
    float bq[DEPTH + 1], aq[DEPTH + 1][C::MT];
    auto ld = [&](int buf, int s, int slot) {
        const int kk = s % (C::CK / 2), xi = (s / (C::CK / 2)) % 4, ky = s / (4 * (C::CK / 2));
        bq[slot] = (vlane0 + buf * C::BUF)[xi * C::VPLANE + ((2 * kk) * C::XR + ky) * 32];
        const float* ua = ulane0 + buf * C::BUF + ((ky * 4 + xi) * C::CK + 2 * kk) * C::CT;
        if constexpr (C::MT == 2) {
            const f32x2 a2 = *reinterpret_cast<const f32x2*>(ua);
            aq[slot][0] = a2[0];
            aq[slot][1] = a2[1];
        } else {
            aq[slot][0] = *ua;
        }
    };
#define STAMP(k)
    // MORE / PF (is there a chunk c+1 to store, a chunk c+2 to load) are compile-time: the steady-state body carries no
    // branches, the last two chunks run their own copies
    auto mfmas = [&](int cur, int chunk, auto morec, auto pfc) {
        constexpr bool more = decltype(morec)::value, pf = decltype(pfc)::value;
        // compile-time step index: the staging item (and its register arrays) must resolve statically, whatever the unroller thinks
        static_for<NSTEP>([&](auto sc) {
            constexpr int s = decltype(sc)::value, xi = (s / (C::CK / 2)) % 4;
            if constexpr (s % 6 == 0) STAMP(s / 6);
            if constexpr (s + DEPTH == NSTEP) {
                if constexpr (more) __syncthreads();
                STAMP(8);
            }
            if constexpr (s + DEPTH < NSTEP) {
                ld(cur, s + DEPTH, (s + DEPTH) % (DEPTH + 1));
            } else if constexpr (more) {
                ld(cur ^ 1, s + DEPTH - NSTEP, (s + DEPTH) % (DEPTH + 1));
            }
#pragma unroll
            for (int m = 0; m < C::MT; ++m)
                acc[m][xi] = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[s % (DEPTH + 1)][m], bq[s % (DEPTH + 1)], acc[m][xi], 0, 0, 0);
            if constexpr (s >= SLOT0 && (s - SLOT0) % SLOTD == 0 && (s - SLOT0) / SLOTD < NITEM) {
                constexpr int k = (s - SLOT0) / SLOTD;
                if constexpr (more) store_item(sc_int<k>{}, cur ^ 1);
                if constexpr (pf) load_item(sc_int<k>{}, chunk + 2);
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
    for (; chunk + 2 < p.nchunks; ++chunk) {
        mfmas(chunk & 1, chunk, T{}, T{});
        STAMP(9);
    }
    if (chunk + 1 < p.nchunks) {
        mfmas(chunk & 1, chunk, T{}, F{});
        ++chunk;
    }
    mfmas(chunk & 1, chunk, F{}, F{});
}

// ------------------------------------------------------------------------------------------------ epilogues
enum { WEPI_GENERIC = 0, WEPI_NONE, WEPI_ELU, WEPI_PRELU, WEPI_RES_PRELU };

template <int EPI>
struct WEpi {
    static constexpr int ACT1 = EPI == WEPI_ELU ? CWFA_ACT_ELU : EPI == WEPI_PRELU ? CWFA_ACT_PRELU : CWFA_ACT_NONE;
    static constexpr bool RES = EPI == WEPI_RES_PRELU;
    static constexpr int ACT2 = EPI == WEPI_RES_PRELU ? CWFA_ACT_PRELU : CWFA_ACT_NONE;
};

template <int ACT>
__device__ __forceinline__ float wact(float v, float alpha) {
    if constexpr (ACT == CWFA_ACT_ELU) return cwfa_elu(v);
    if constexpr (ACT == CWFA_ACT_PRELU) return v > 0.f ? v : alpha * v;
    return v;
}

template <class C, int EPI>
__device__ __forceinline__ void wino_epilogue(const WParams& p, const WTile& t, f32x16 (&acc)[C::MT][4], float* smem) {
    const int64_t HW = (int64_t)p.H * p.W;
    const int row = t.row0 + t.wn, col = t.col0 + 2 * t.l31;
    const bool ok0 = row < p.H && col < p.W, ok1 = row < p.H && col + 1 < p.W;
    float* yb = p.y + (int64_t)t.b * p.y_bs + (int64_t)row * p.W + col;
    const bool vec = ok1 && ((HW | p.W | p.y_bs) & 1) == 0 && ((reinterpret_cast<uintptr_t>(p.y) & 7) == 0);
    if constexpr (EPI == WEPI_GENERIC) {
        // runtime-switched activations: the two output pixels go through LDS so ONE copy of the code serves all rows
        const float alpha = (p.o.prelu_alpha && (p.o.act == CWFA_ACT_PRELU || p.o.act2 == CWFA_ACT_PRELU)) ? *p.o.prelu_alpha : 0.f;
        const float* rb = p.o.residual ? p.o.residual + (int64_t)t.b * p.o.res_bs + (int64_t)row * p.W + col : nullptr;
        __syncthreads();
        float* mine = smem + (threadIdx.x >> 6) * 2048 + (threadIdx.x & 63);      // [2][16] floats x 64 lanes per wave
        for (int m = 0; m < C::MT; ++m) {
#pragma unroll
            for (int mm = 0; mm < C::MT; ++mm)
                if (mm == m) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        mine[r * 64] = (acc[mm][0][r] + acc[mm][1][r]) + acc[mm][2][r];
                        mine[(16 + r) * 64] = (acc[mm][1][r] - acc[mm][2][r]) - acc[mm][3][r];
                    }
                }
            for (int r = 0; r < 16; ++r) {
                const int co = t.ct * C::CT + (t.wm * C::MT + m) * 32 + acc_row(r, t.kh);
                if (co >= p.Cout) continue;
                const float bias = p.o.bias ? p.o.bias[co] : 0.f;
                for (int px = 0; px < 2; ++px) {
                    if (!(px ? ok1 : ok0)) continue;
                    float v = cwfa_act(mine[(px * 16 + r) * 64] + bias, p.o.act, alpha);
                    if (rb) v += rb[(int64_t)co * HW + px];
                    yb[(int64_t)co * HW + px] = cwfa_act(v, p.o.act2, alpha);
                }
            }
        }
    } else {
        typedef WEpi<EPI> E;
        float alpha = 0.f;
        if constexpr (E::ACT1 == CWFA_ACT_PRELU || E::ACT2 == CWFA_ACT_PRELU) alpha = *p.o.prelu_alpha;
        const float* rb = nullptr;
        if constexpr (E::RES) rb = p.o.residual + (int64_t)t.b * p.o.res_bs + (int64_t)row * p.W + col;
#pragma unroll
        for (int m = 0; m < C::MT; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = t.ct * C::CT + (t.wm * C::MT + m) * 32 + acc_row(r, t.kh);
                if (co >= p.Cout) continue;
                const float bias = p.o.bias ? p.o.bias[co] : 0.f;
                float e0 = (acc[m][0][r] + acc[m][1][r]) + acc[m][2][r];
                float e1 = (acc[m][1][r] - acc[m][2][r]) - acc[m][3][r];
                e0 = wact<E::ACT1>(e0 + bias, alpha);
                e1 = wact<E::ACT1>(e1 + bias, alpha);
                const int64_t o = (int64_t)co * HW;
                if (vec) {
                    if constexpr (E::RES) {
                        const f32x2 rr = *reinterpret_cast<const f32x2*>(rb + o);
                        e0 += rr[0];
                        e1 += rr[1];
                    }
                    f32x2 out = {wact<E::ACT2>(e0, alpha), wact<E::ACT2>(e1, alpha)};
                    *reinterpret_cast<f32x2*>(yb + o) = out;
                } else {
                    if (ok0) {
                        if constexpr (E::RES) e0 += rb[o];
                        yb[o] = wact<E::ACT2>(e0, alpha);
                    }
                    if (ok1) {
                        if constexpr (E::RES) e1 += rb[o + 1];
                        yb[o + 1] = wact<E::ACT2>(e1, alpha);
                    }
                }
            }
    }
}

template <class C, int EPI, bool PRO>
__global__ __launch_bounds__(C::NTHREADS, 1) void conv3x3_wino_kernel(WParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const WTile t = make_wtile<C>(p);
    f32x16 acc[C::MT][4];
    LSTAMP(0);
    wino_mainloop<C, PRO>(p, t, smem, smem + C::VS, acc);
    LSTAMP(1);
    wino_epilogue<C, EPI>(p, t, acc, smem);
    LSTAMP(2);
}

// ---- weight transform + repack: torch [Cout][Cin][3][3] -> [cout tile][chunk][ky][xi][ck][CT]
__global__ __launch_bounds__(256) void wino_pack_kernel(const float* __restrict__ w, float* __restrict__ out, int Cout, int Cin,
                                                        int CT, int CK, int nchunks, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int col = (int)(i % CT);
    const int ck = (int)((i / CT) % CK);
    const int xi = (int)((i / ((int64_t)CT * CK)) % 4);
    const int ky = (int)((i / ((int64_t)CT * CK * 4)) % 3);
    const int chunk = (int)((i / ((int64_t)CT * CK * 12)) % nchunks);
    const int ctile = (int)(i / ((int64_t)CT * CK * 12 * nchunks));
    // within a panel the two 32-channel m-tiles of a wave are interleaved (position 64g + 2i + m <-> channel 64g + 32m + i)
    // so a lane fetches its A operands for both tiles with ONE ds_read_b64
    const int co = ctile * CT + (CT >= 64 ? (col & ~63) + (col & 1) * 32 + ((col & 63) >> 1) : col), ci = chunk * CK + ck;
    float v = 0.f;
    if (co < Cout && ci < Cin) {
        const float* g = w + (((int64_t)co * Cin + ci) * 3 + ky) * 3;
        const float g0 = g[0], g1 = g[1], g2 = g[2];
        v = xi == 0 ? g0 : xi == 1 ? ((g0 + g1) + g2) * 0.5f : xi == 2 ? ((g0 - g1) + g2) * 0.5f : g2;
    }
    out[i] = v;
}

typedef WCfg<8, 1, 1, 8> W32;       // Cout <= 32: 32 ch x 8 rows x 64 cols, 512 threads (one m-tile per wave)
typedef WCfg<8, 2, 1, 8> W64;       // Cout <= 64: 64 ch x 8 rows x 64 cols, 512 threads
typedef WCfg<8, 2, 2, 4> W128;      // Cout  > 64: 128 ch x 4 rows x 64 cols, 512 threads

// ------------------------------------------------------------------------------------------------ fused sub-network layer
// y = ELU( W1x1 . ELU( conv3x3(x) + b3 ) + b1 + x ), 64 channels, ONE launch (networks.py:624-631,660-665).
// After the Winograd main loop a lane holds, per hidden channel, the four components of ONE pixel pair.  The output
// transform + bias + ELU produce the hidden value of the even pixel (pass 0) and of the odd pixel (pass 1) in place, and
// those registers ARE the B operand of the 1x1 GEMM (k-pair of register r = channels {32m + row(r), +4} in the two lane
// halves, N = 32 even resp. odd pixels).  Four chained-MFMA passes q = (cout half, pixel parity); the VALU work (output
// transform, ELU, residual add, store) is slotted between the MFMAs; both parities of a channel leave as one 8-byte store.
__device__ __forceinline__ f32x2 ld2(const float* base, unsigned byte_off, bool vec, bool ok1) {
    asm volatile("" : "+v"(byte_off));
    const char* q = reinterpret_cast<const char*>(base) + byte_off;
    if (vec) return *reinterpret_cast<const f32x2*>(q);
    f32x2 v;
    v[0] = *reinterpret_cast<const float*>(q);
    v[1] = ok1 ? *reinterpret_cast<const float*>(q + 4) : 0.f;
    return v;
}
__device__ __forceinline__ void st2(float* base, unsigned byte_off, f32x2 v, bool vec, bool ok0, bool ok1) {
    asm volatile("" : "+v"(byte_off));
    char* q = reinterpret_cast<char*>(base) + byte_off;
    if (vec) {
        *reinterpret_cast<f32x2*>(q) = v;
    } else {
        if (ok0) *reinterpret_cast<float*>(q) = v[0];
        if (ok1) *reinterpret_cast<float*>(q + 4) = v[1];
    }
}
__device__ __forceinline__ float ld1(const float* base, unsigned byte_off) {
    asm volatile("" : "+v"(byte_off));
    return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_off);
}

template <bool TAPE>
__global__ __launch_bounds__(W64::NTHREADS, 1) void wino_layer_kernel(WParams p) {
    typedef W64 C;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Us = smem + C::VS;
    const WTile t = make_wtile<C>(p);
    f32x16 acc[2][4];
    LSTAMP(0);
    wino_mainloop<C, false>(p, t, smem, Us, acc);
    LSTAMP(1);

    const int64_t HW = (int64_t)p.H * p.W;
    const int row = t.row0 + t.wn, col = t.col0 + 2 * t.l31;
    const bool ok0 = row < p.H && col < p.W, ok1 = row < p.H && col + 1 < p.W;
    const bool vec = ok1 && ((HW | p.W | p.x_bs | p.y_bs) & 1) == 0 &&
                     (((reinterpret_cast<uintptr_t>(p.x) | reinterpret_cast<uintptr_t>(p.y)) & 7) == 0);
    const float* xb = p.x + (int64_t)t.b * p.x_bs;
    float* yb = p.y + (int64_t)t.b * p.y_bs;
    float* hb = TAPE ? p.hidden + (int64_t)t.b * p.hidden_bs : nullptr;
    const bool vech = TAPE && ok1 && ((HW | p.W | p.hidden_bs) & 1) == 0 && ((reinterpret_cast<uintptr_t>(p.hidden) & 7) == 0);
    const unsigned HW4 = (unsigned)HW * 4u;
    // per-lane byte offset of this lane's pixel pair in channel 4*kh; channel K adds the scalar K*HW4
    const unsigned oo = (ok0 ? (unsigned)(row * p.W + col) * 4u : 0u) + (unsigned)t.kh * 4u * HW4;

    f32x16 b3v[2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) b3v[m][r] = ld1(p.o.bias, (unsigned)(m * 32 + acc_row(r, 0)) * 4u + (unsigned)t.kh * 16u);

    // stage the 1x1 panel (4096 floats) where the transformed 3x3 weights were
    __syncthreads();
    {
        const f32x4* src = reinterpret_cast<const f32x4*>(p.w1x1);
        f32x4* dst = reinterpret_cast<f32x4*>(Us);
        for (int e = threadIdx.x; e < 1024; e += C::NTHREADS) dst[e] = src[e];
    }
    __syncthreads();
    const float* wl = Us + (threadIdx.x & 63);
    LSTAMP(2);

    auto load_res = [&](int mo, f32x2 (&res)[16]) {           // residual x + 1x1 bias of 16 output channels, both pixels
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const unsigned K = (unsigned)(mo * 32 + acc_row(r, 0));
            const float b1 = ld1(p.b1x1, K * 4u + (unsigned)t.kh * 16u);
            res[r] = ld2(xb, K * HW4 + oo, vec, ok1) + b1;
        }
    };
    f32x16 yq[4];
    f32x2 rq0[16], rq1[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int mo = q >> 1, par = q & 1;
        if (q == 2) load_res(0, rq0);
        if (q == 3) load_res(1, rq1);
        float a_next = wl[(0 * 2 + mo) * 64];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int j = m * 16 + r;
                const float a = a_next;
                if (j + 1 < 32) a_next = wl[((j + 1) * 2 + mo) * 64];
                if (q == 0) acc[m][0][r] = cwfa_elu(((acc[m][0][r] + acc[m][1][r]) + acc[m][2][r]) + b3v[m][r]);
                if (q == 1) acc[m][1][r] = cwfa_elu(((acc[m][1][r] - acc[m][2][r]) - acc[m][3][r]) + b3v[m][r]);
                if constexpr (TAPE) {
                    if (q == 1) {                           // both pixels of hidden channel m*32 + row(r) are final: keep them
                        const f32x2 hv = {acc[m][0][r], acc[m][1][r]};
                        if (ok0) st2(hb, (unsigned)(m * 32 + acc_row(r, 0)) * HW4 + oo, hv, vech, ok0, ok1);
                    }
                }
                if (j == 0) {
                    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    yq[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, acc[m][par][r], zero, 0, 0, 0);
                } else {
                    yq[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, acc[m][par][r], yq[q], 0, 0, 0);
                }
                if (q == 3 && (j & 1)) {                    // output channels 0..31 are complete: one per two k-steps
                    const int rr = j >> 1;
                    const unsigned K = (unsigned)acc_row(rr, 0);
                    f32x2 o = {cwfa_elu(yq[0][rr] + rq0[rr][0]), cwfa_elu(yq[1][rr] + rq0[rr][1])};
                    if (ok0) st2(yb, K * HW4 + oo, o, vec, ok0, ok1);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
    }
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) {                       // tail: output channels 32..63
        const unsigned K = (unsigned)(32 + acc_row(rr, 0));
        f32x2 o = {cwfa_elu(yq[2][rr] + rq1[rr][0]), cwfa_elu(yq[3][rr] + rq1[rr][1])};
        if (ok0) st2(yb, K * HW4 + oo, o, vec, ok0, ok1);
    }
    LSTAMP(3);
}

struct WSel {
    int CT, CK;
};
WSel wsel(int Cout) { return Cout <= 32 ? WSel{W32::CT, W32::CK} : Cout <= 64 ? WSel{W64::CT, W64::CK} : WSel{W128::CT, W128::CK}; }

template <class C, int EPI, bool PRO>
int wlaunch(WParams p, hipStream_t stream) {
    p.tiles_x = (p.W + C::TCOLS - 1) / C::TCOLS;
    p.tiles_y = (p.H + C::TR - 1) / C::TR;
    p.nchunks = (p.Cin + C::CK - 1) / C::CK;
    const int ctiles = (p.Cout + C::CT - 1) / C::CT;
    CWFA_REQUIRE((int64_t)p.tiles_x * p.tiles_y * ctiles < (1ll << 31) && p.B <= 65535, CWFA_E_SHAPE,
                 "cwfa_conv2d_f32: grid too large");
    CWFA_REQUIRE((int64_t)(p.Cin + C::CK) * p.H * p.W * 4 < (1ll << 31), CWFA_E_SHAPE,
                 "cwfa_conv2d_f32: one sample's input must stay below 2 GiB (32-bit buffer offsets)");
    constexpr int LDS = EPI == WEPI_GENERIC && C::LDS_BYTES < C::NTHREADS * 128 ? C::NTHREADS * 128 : C::LDS_BYTES;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_wino_kernel<C, EPI, PRO>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) {
            cwfa_set_error("cwfa_conv2d_f32 (winograd): hipFuncSetAttribute(%d bytes LDS): %s", LDS, hipGetErrorString(e));
            return CWFA_E_HIP;
        }
        attr_set = true;
    }
    dim3 grid((unsigned)(p.tiles_x * p.tiles_y * ctiles), 1, p.B);
    hipLaunchKernelGGL((conv3x3_wino_kernel<C, EPI, PRO>), grid, dim3(C::NTHREADS), LDS, stream, p);
    CWFA_LAUNCH_CHECK("cwfa_conv2d_f32 (winograd)");
    return CWFA_OK;
}

int classify(const cwfa_conv_opts& o) {
    const bool res = o.residual != nullptr;
    if (!res && o.act2 == CWFA_ACT_NONE) {
        if (o.act == CWFA_ACT_NONE) return WEPI_NONE;
        if (o.act == CWFA_ACT_ELU) return WEPI_ELU;
        if (o.act == CWFA_ACT_PRELU) return WEPI_PRELU;
    }
    if (res && o.act == CWFA_ACT_NONE && o.act2 == CWFA_ACT_PRELU) return WEPI_RES_PRELU;
    return WEPI_GENERIC;
}

}  // namespace

int64_t cwfa_wino_packed_floats(int Cout, int Cin) {
    if (cwfa_wino2d_selected(Cout)) return cwfa_wino2d_packed_floats(Cout, Cin);
    const WSel s = wsel(Cout);
    const int64_t ctiles = (Cout + s.CT - 1) / s.CT, nchunks = (Cin + s.CK - 1) / s.CK;
    return ctiles * nchunks * 12 * s.CK * s.CT;
}

int cwfa_wino_pack(const float* w, float* packed, int Cout, int Cin, hipStream_t stream) {
    if (cwfa_wino2d_selected(Cout)) return cwfa_wino2d_pack(w, packed, Cout, Cin, stream);
    const WSel s = wsel(Cout);
    const int64_t total = cwfa_wino_packed_floats(Cout, Cin);
    const int nchunks = (Cin + s.CK - 1) / s.CK;
    hipLaunchKernelGGL(wino_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, w, packed, Cout, Cin, s.CT,
                       s.CK, nchunks, total);
    CWFA_LAUNCH_CHECK("cwfa_conv2d_pack_f32 (winograd)");
    return CWFA_OK;
}

int cwfa_wino_conv(const float* x, const float* w_packed, float* y, int B, int Cin, int H, int W, int Cout, int64_t x_bs,
                   int64_t y_bs, const cwfa_conv_opts& o, hipStream_t stream) {
    if (cwfa_wino2d_selected(Cout)) return cwfa_wino2d_conv(x, w_packed, y, B, Cin, H, W, Cout, x_bs, y_bs, o, stream);
    WParams p{};
    p.x = x; p.wp = w_packed; p.y = y;
    p.B = B; p.Cin = Cin; p.H = H; p.W = W; p.Cout = Cout;
    p.x_bs = x_bs; p.y_bs = y_bs;
    p.o = o;
    const bool pro = o.in_scale || o.in_add;
    const int epi = classify(o);
    if (Cout <= 32) {
        if (pro) return wlaunch<W32, WEPI_GENERIC, true>(p, stream);
        switch (epi) {
            case WEPI_NONE: return wlaunch<W32, WEPI_NONE, false>(p, stream);
            case WEPI_PRELU: return wlaunch<W32, WEPI_PRELU, false>(p, stream);
            case WEPI_RES_PRELU: return wlaunch<W32, WEPI_RES_PRELU, false>(p, stream);
            default: return wlaunch<W32, WEPI_GENERIC, false>(p, stream);
        }
    }
    if (Cout <= 64) {
        if (pro) return wlaunch<W64, WEPI_GENERIC, true>(p, stream);
        switch (epi) {
            case WEPI_NONE: return wlaunch<W64, WEPI_NONE, false>(p, stream);
            case WEPI_ELU: return wlaunch<W64, WEPI_ELU, false>(p, stream);
            case WEPI_PRELU: return wlaunch<W64, WEPI_PRELU, false>(p, stream);
            case WEPI_RES_PRELU: return wlaunch<W64, WEPI_RES_PRELU, false>(p, stream);
            default: return wlaunch<W64, WEPI_GENERIC, false>(p, stream);
        }
    }
    if (pro) {
        if (epi == WEPI_PRELU) return wlaunch<W128, WEPI_PRELU, true>(p, stream);
        return wlaunch<W128, WEPI_GENERIC, true>(p, stream);
    }
    switch (epi) {
        case WEPI_NONE: return wlaunch<W128, WEPI_NONE, false>(p, stream);
        case WEPI_PRELU: return wlaunch<W128, WEPI_PRELU, false>(p, stream);
        default: return wlaunch<W128, WEPI_GENERIC, false>(p, stream);
    }
}

int cwfa_wino_layer(const float* x, const float* w3_packed, const float* b3, const float* w1_panel, const float* b1, float* y,
                    int B, int H, int W, int64_t x_bs, int64_t y_bs, hipStream_t stream, float* hidden, int64_t hidden_bs) {
    typedef W64 C;
    WParams p{};
    p.x = x; p.wp = w3_packed; p.y = y;
    p.B = B; p.Cin = 64; p.H = H; p.W = W; p.Cout = 64;
    p.x_bs = x_bs; p.y_bs = y_bs;
    p.o.bias = b3;
    p.w1x1 = w1_panel;
    p.b1x1 = b1;
    p.hidden = hidden;
    p.hidden_bs = hidden_bs;
    p.tiles_x = (W + C::TCOLS - 1) / C::TCOLS;
    p.tiles_y = (H + C::TR - 1) / C::TR;
    p.nchunks = 64 / C::CK;
    CWFA_REQUIRE((int64_t)p.tiles_x * p.tiles_y < (1ll << 31) && B <= 65535, CWFA_E_SHAPE, "cwfa_subnet_layer_f32: grid too large");
    CWFA_REQUIRE((int64_t)(64 + C::CK) * H * W * 4 < (1ll << 31), CWFA_E_SHAPE,
                 "cwfa_subnet_layer_f32: one sample's input must stay below 2 GiB (32-bit buffer offsets)");
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wino_layer_kernel<false>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wino_layer_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    C::LDS_BYTES);
        if (e != hipSuccess) {
            cwfa_set_error("cwfa_subnet_layer_f32: hipFuncSetAttribute(%d bytes LDS): %s", C::LDS_BYTES, hipGetErrorString(e));
            return CWFA_E_HIP;
        }
        attr_set = true;
    }
    dim3 grid((unsigned)(p.tiles_x * p.tiles_y), 1, B);
    if (hidden)
        hipLaunchKernelGGL(wino_layer_kernel<true>, grid, dim3(C::NTHREADS), C::LDS_BYTES, stream, p);
    else
        hipLaunchKernelGGL(wino_layer_kernel<false>, grid, dim3(C::NTHREADS), C::LDS_BYTES, stream, p);
    CWFA_LAUNCH_CHECK("cwfa_subnet_layer_f32 (winograd)");
    return CWFA_OK;
}

