// 3x3 convolution as a 2-D Winograd F(2x2,3x3) implicit GEMM on the fp32 matrix cores (v_mfma_f32_32x32x2_f32), for
// layers with more than 64 output channels (the UNet of the LRNN: 62 % of the FLOPs of the path).
//
//   tile = 2x2 output pixels, 4x4 input patch d, 16 components (xi, nu):
//     V = B^T d B,   U = G g G^T,   M[xi][nu] = sum_ci U[xi][nu][co][ci] * V[xi][nu][ci][tile],   Y = A^T M A
//     row operator T(x0..x3) = (x0-x2, x1+x2, x2-x1, x1-x3) (= B^T),  G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1],
//     output operator O(m0..m3) = (m0+m1+m2, m1-m2-m3) (= A^T)
//   16 MFMAs per (ci pair, 32 cout, 32 tiles = 2 rows x 64 pixels) where the direct form needs 36 and the 1-D F(2,3)
//   kernel (conv_wino.hip) 24.
//
// STATUS: opt-in (cwfa_set_option("winograd_2d", 1)), parity-tested like the 1-D kernels.  Measured on MI355X it reaches
// 205 / 240 / 246 TFLOP/s (algorithmic) on 256->256@512^2 / 512->512@256^2 / 1024->1024@128^2 against 203 / 213 / 216 for
// the 1-D kernel, but is slower once a load-side prologue is compiled in; without any staging the loop runs at 93 % of
// the 2.25 x 157 TF/s ceiling, and the ablation puts the gap on the input-transform item (64 vector instructions +
// 8 ds_write_b128 per 64 MFMAs, un-hidden because vector and LDS-write work never overlap the fp32 MFMA here): a V value
// feeds only CT/32 = 2 MFMAs, and CT cannot grow because the accumulators already fill the register file.
//
// A wave holds 16 accumulator tiles = 256 registers, so the kernel runs ONE wave per SIMD (256 threads, 512-register
// budget).  On gfx950 that costs nothing: the fp32 MFMA does not co-execute with vector instructions and a second wave
// hides none of them, so as in conv_wino.hip every wave runs one stream -- the k-steps of chunk c
// with the staging items of chunk c+1 (LDS stores) and c+2 (global loads) placed between them.
//
// Block = 2 x 2 waves: 64 cout x 4 image rows x 64 columns; K-chunks of 8 input channels; two LDS buffers of
//   V[8 comp pairs][8 ci][2 tile rows][32 tiles][2] + U[8 comp pairs][8 ci][64 cout][2]  (2 x 64 KB),
// both operands of two consecutive MFMAs arrive with one ds_read_b64 each.  A staging thread owns (ci, tile row, two
// adjacent tiles): 4 rows x (one 16-byte + two 4-byte buffer loads, padding by range check), the 2-D input transform of
// both patches (64 vector instructions) and 8 ds_write_b128.
#include "conv_internal.h"

#include <type_traits>
#include <utility>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

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

struct W2 {
    static constexpr int CK = 8, CT = 64, NTHREADS = 256;
    static constexpr int TROWS = 2;                  // tile rows per block (4 image rows)
    static constexpr int TCOLS = 64;                 // image columns per block (32 tiles)
    static constexpr int VS = 8 * CK * TROWS * 32 * 2;
    static constexpr int US = 8 * CK * CT * 2;       // = 16 * CK * CT floats per chunk and cout tile
    static constexpr int BUF = VS + US;
    static constexpr int LDS_BYTES = 2 * BUF * 4;
    static constexpr int UPT = US / 4 / NTHREADS;    // 16-byte pieces of the U panel per thread
    static_assert(US % (4 * NTHREADS) == 0 && CK * TROWS * 16 == NTHREADS, "staging map");
};

struct W2Params {
    const float* x;
    const float* wp;
    float* y;
    int B, Cin, H, W, Cout, nchunks, tiles_x, tiles_y;
    int64_t x_bs, y_bs;
    cwfa_conv_opts o;
};

__device__ __forceinline__ int acc_row(int r, int kh) { return (r & 3) + 8 * (r >> 2) + 4 * kh; }

__device__ __forceinline__ void rowop(float x0, float x1, float x2, float x3, float (&o)[4]) {
    o[0] = x0 - x2;
    o[1] = x1 + x2;
    o[2] = x2 - x1;
    o[3] = x1 - x3;
}

enum { W2EPI_GENERIC = 0, W2EPI_NONE, W2EPI_PRELU };

// ALIGNED: image rows start on 16-byte boundaries (W % 4 == 0, aligned bases) -> the middle four columns of a staging
// item are one 16-byte load; otherwise four 4-byte loads.
template <int EPI, bool PRO, bool ALIGNED>
__global__ __launch_bounds__(W2::NTHREADS) void conv3x3_wino2d_kernel(W2Params p) {
    typedef W2 C;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, kh = lane >> 5, l31 = lane & 31;
    const int64_t HW = (int64_t)p.H * p.W;

    // XCD-aware block -> (spatial tile, cout tile) map, as in conv_wino.hip
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
    const int row0 = (tile / p.tiles_x) * 4, col0 = (tile % p.tiles_x) * C::TCOLS, b = blockIdx.z;

    // ---- staging map: (ci, tile row, tile pair u); columns 4u-1 .. 4u+4 of four rows
    constexpr unsigned OOB = 0x80000000u;
    const int s_ci = tid >> 5, s_tr = (tid >> 4) & 1, s_u = tid & 15;
    unsigned voff[4][ALIGNED ? 3 : 6];
    unsigned long long rowok[PRO ? 4 : 1], colok[PRO ? 6 : 1];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int gr = row0 + 2 * s_tr - 1 + a;
        const bool rok = gr >= 0 && gr < p.H;
        const int64_t rbase = s_ci * HW + (int64_t)gr * p.W;
        if constexpr (PRO) rowok[a] = __builtin_amdgcn_ballot_w64(rok);
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const int gc = col0 + 4 * s_u - 1 + k;
            const bool ok = rok && gc >= 0 && gc < p.W;
            if constexpr (PRO) {
                if (a == 0) colok[k] = __builtin_amdgcn_ballot_w64(gc >= 0 && gc < p.W);
            }
            const unsigned v = ok ? (unsigned)((rbase + gc) * 4) : OOB;
            if constexpr (ALIGNED) {
                if (k == 0) voff[a][0] = v;
                if (k == 1) voff[a][1] = v;          // 16-byte load: columns 4u .. 4u+3 (all in or all out: W % 4 == 0)
                if (k == 5) voff[a][2] = v;
            } else {
                voff[a][k] = v;
            }
        }
    }
    const bool has_aff = PRO && p.o.in_scale != nullptr, has_add = PRO && p.o.in_add != nullptr;
    const int xbytes = (int)((int64_t)p.Cin * HW * 4);
    const auto rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x + (int64_t)b * p.x_bs), 0, xbytes, 0x00020000);
    const auto ra = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(has_add ? p.o.in_add + (int64_t)b * p.o.in_add_bs : p.x), 0, has_add ? xbytes : 0, 0x00020000);
    const auto rsc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(has_aff ? p.o.in_scale + (int64_t)b * p.o.in_affine_bs : p.x), 0, has_aff ? p.Cin * 4 : 0, 0x00020000);
    const auto rsh = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(has_aff ? p.o.in_shift + (int64_t)b * p.o.in_affine_bs : p.x), 0, has_aff ? p.Cin * 4 : 0, 0x00020000);
    const auto rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wp + (int64_t)ct * p.nchunks * C::US), 0,
                                                      p.nchunks * C::US * 4, 0x00020000);
    const int chunk_bytes = (int)(C::CK * HW * 4);
    auto ldf = [](decltype(rx) r, unsigned vo, int so) { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, vo, so, 0)); };
    auto ldrow = [&](decltype(rx) r, int a, int so, float (&d)[6]) {
        if constexpr (ALIGNED) {
            d[0] = ldf(r, voff[a][0], so);
            const f32x4 m = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff[a][1], so, 0));
            d[1] = m[0]; d[2] = m[1]; d[3] = m[2]; d[4] = m[3];
            d[5] = ldf(r, voff[a][2], so);
        } else {
#pragma unroll
            for (int k = 0; k < 6; ++k) d[k] = ldf(r, voff[a][k], so);
        }
    };

    // A chunk is only ~2 us of matrix work here, less than a loaded HBM round trip: the input loads run TWO chunks ahead,
    // into two register sets (set = chunk parity; the chunk loop is unrolled by two so the set is a compile-time index).
    float dr2[2][4][6], ar2[PRO ? 2 : 1][PRO ? 4 : 1][6], sr2[2] = {1.f, 1.f}, hr2[2] = {0.f, 0.f};
    f32x4 ur[C::UPT];
    auto load_v = [&](auto setc, int chunk) {
        constexpr int set = decltype(setc)::value;
        const int so = chunk * chunk_bytes;
#pragma unroll
        for (int a = 0; a < 4; ++a) ldrow(rx, a, so, dr2[set][a]);
        if constexpr (PRO) {
            if (has_aff) {
                sr2[set] = ldf(rsc, s_ci * 4, chunk * C::CK * 4);
                hr2[set] = ldf(rsh, s_ci * 4, chunk * C::CK * 4);
            }
            if (has_add) {
#pragma unroll
                for (int a = 0; a < 4; ++a) ldrow(ra, a, so, ar2[set][a]);
            }
        }
    };
    auto store_v = [&](auto setc, int buf) {
        constexpr int set = decltype(setc)::value;
        auto& dr = dr2[set];
        auto& ar = ar2[PRO ? set : 0];
        const float sr = sr2[set], hr = hr2[set];
        float e[4][8];                 // row-transformed: e[a][nu] for tile 2u (0..3) and tile 2u+1 (4..7)
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            float d[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                float v = dr[a][k];
                if constexpr (PRO) {
                    if (has_aff) {
                        v = v * sr + hr;                                // padding must be zero AFTER the affine
                        asm("v_cndmask_b32 %0, 0, %1, %2" : "=v"(v) : "v"(v), "s"(rowok[a] & colok[k]));
                    }
                    if (has_add) v += ar[a][k];
                }
                d[k] = v;
            }
            float o0[4], o1[4];
            rowop(d[0], d[1], d[2], d[3], o0);
            rowop(d[2], d[3], d[4], d[5], o1);
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                e[a][n] = o0[n];
                e[a][4 + n] = o1[n];
            }
        }
        // column transform over the four rows, then one 16-byte store per component pair (both tiles)
        float v[8][4];                 // v[column][xi]
#pragma unroll
        for (int n = 0; n < 8; ++n) rowop(e[0][n], e[1][n], e[2][n], e[3][n], v[n]);
        float* dst = smem + buf * C::BUF + (((s_ci * 2 + s_tr) * 32 + 2 * s_u) * 2);
#pragma unroll
        for (int xi = 0; xi < 4; ++xi)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const f32x4 q = {v[2 * j][xi], v[2 * j + 1][xi], v[4 + 2 * j][xi], v[4 + 2 * j + 1][xi]};
                *reinterpret_cast<f32x4*>(dst + (2 * xi + j) * (C::CK * 2 * 32 * 2)) = q;
            }
    };
    const unsigned uoff = tid * 16;
    auto load_u = [&](int i, int chunk) {
        ur[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, uoff, (chunk * C::US + i * C::NTHREADS * 4) * 4, 0));
    };
    auto store_u = [&](int i, int buf) { reinterpret_cast<f32x4*>(smem + buf * C::BUF + C::VS)[tid + i * C::NTHREADS] = ur[i]; };
    constexpr int NITEM = 1 + C::UPT;
    f32x16 acc[16];
#pragma unroll
    for (int c = 0; c < 16; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;

    // k-step = one component pair of one ci pair: two MFMAs, operands read DEPTH steps ahead into a register ring
    constexpr int NSTEP = 8 * (C::CK / 2), DEPTH = 3, RING = DEPTH + 1, SLOT0 = 1, SLOTD = 3;
    static_assert(NSTEP % RING == 0 && SLOT0 + (NITEM - 1) * SLOTD < NSTEP - DEPTH, "ring / staging before the barrier");
    const float* vlane0 = smem + ((kh * 2 + wn) * 32 + l31) * 2;
    const float* ulane0 = smem + C::VS + (kh * C::CT + wm * 32 + l31) * 2;
    f32x2 aq[RING], bq[RING];
    auto ld = [&](int buf, int s, int slot) {   // buf, s, slot are compile-time at every call site
        const int kk = s >> 3, cp = s & 7;
        bq[slot] = *reinterpret_cast<const f32x2*>(vlane0 + buf * C::BUF + ((cp * C::CK + 2 * kk) * 2 * 32) * 2);
        aq[slot] = *reinterpret_cast<const f32x2*>(ulane0 + buf * C::BUF + ((cp * C::CK + 2 * kk) * C::CT) * 2);
    };
    // chunk c (LDS buffer and register set parity CUR): k-steps of chunk c; item 0 stores the V tile of chunk c+1 from
    // register set CUR^1 and re-loads that set with chunk c+3; items 1.. store / load the U pieces of chunk c+1 / c+2
    auto chunk_body = [&](auto setc, int chunk, auto morec) {
        constexpr int set = decltype(setc)::value;               // register-set parity (compile-time)
        constexpr bool more = decltype(morec)::value;
        // LDS buffer parity, the same value but kept a RUN-TIME scalar: folded into the addresses it would push the
        // second buffer's operand offsets past the 16-bit ds_read immediate (one v_add per read)
        int cur = set;
        asm volatile("" : "+s"(cur));
        static_for<NSTEP>([&](auto sc) {
            constexpr int s = decltype(sc)::value, cp = s & 7;
            if constexpr (s + DEPTH == NSTEP && more) __syncthreads();
            if constexpr (s + DEPTH < NSTEP) {
                ld(cur, s + DEPTH, (s + DEPTH) % RING);
            } else if constexpr (more) {
                ld(cur ^ 1, s + DEPTH - NSTEP, (s + DEPTH) % RING);
            }
            acc[2 * cp] = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[s % RING][0], bq[s % RING][0], acc[2 * cp], 0, 0, 0);
            acc[2 * cp + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[s % RING][1], bq[s % RING][1], acc[2 * cp + 1], 0, 0, 0);
            if constexpr (more && s >= SLOT0 && (s - SLOT0) % SLOTD == 0 && (s - SLOT0) / SLOTD < NITEM) {
                constexpr int k = (s - SLOT0) / SLOTD;
                if constexpr (k == 0) {
                    store_v(sc_int<set ^ 1>{}, cur ^ 1);
                    load_v(sc_int<set ^ 1>{}, chunk + 3);      // past the last chunk: out of range, reads zeros, never stored
                } else {
                    store_u(k - 1, cur ^ 1);
                    load_u(k - 1, chunk + 2);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        });
    };

    load_v(sc_int<0>{}, 0);
    static_for<C::UPT>([&](auto kc) { load_u(decltype(kc)::value, 0); });
    load_v(sc_int<1>{}, 1);
    store_v(sc_int<0>{}, 0);
    static_for<C::UPT>([&](auto kc) {
        store_u(decltype(kc)::value, 0);
        load_u(decltype(kc)::value, 1);
    });
    load_v(sc_int<0>{}, 2);
    __syncthreads();
#pragma unroll
    for (int s = 0; s < DEPTH; ++s) ld(0, s, s);
    typedef std::true_type T;
    typedef std::false_type F;
    int chunk = 0;
    for (; chunk + 2 < p.nchunks; chunk += 2) {
        chunk_body(sc_int<0>{}, chunk, T{});
        chunk_body(sc_int<1>{}, chunk + 1, T{});
    }
    if (chunk + 1 < p.nchunks) {            // two chunks left
        chunk_body(sc_int<0>{}, chunk, T{});
        chunk_body(sc_int<1>{}, chunk + 1, F{});
    } else {                                // one chunk left
        chunk_body(sc_int<0>{}, chunk, F{});
    }

    // ---- epilogue: Y = A^T M A per accumulator row, bias / activation, 2x2 pixels per lane and channel
    const int orow = row0 + 2 * wn, ocol = col0 + 2 * l31;
    const bool rok[2] = {orow < p.H, orow + 1 < p.H}, c0ok = ocol < p.W, c1ok = ocol + 1 < p.W;
    const bool vec = c1ok && ((HW | p.W | p.y_bs) & 1) == 0 && ((reinterpret_cast<uintptr_t>(p.y) & 7) == 0);
    float* yb = p.y + (int64_t)b * p.y_bs + (int64_t)orow * p.W + ocol;
    auto out_transform = [&](int r, float (&yv)[2][2]) {
        float z[4][2];
#pragma unroll
        for (int xi = 0; xi < 4; ++xi) {
            z[xi][0] = (acc[4 * xi][r] + acc[4 * xi + 1][r]) + acc[4 * xi + 2][r];
            z[xi][1] = (acc[4 * xi + 1][r] - acc[4 * xi + 2][r]) - acc[4 * xi + 3][r];
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            yv[0][q] = (z[0][q] + z[1][q]) + z[2][q];
            yv[1][q] = (z[1][q] - z[2][q]) - z[3][q];
        }
    };
    if constexpr (EPI == W2EPI_GENERIC) {
        // runtime-switched activations / residual: the 2x2 outputs go through LDS so one copy of the code serves all rows
        const float alpha = (p.o.prelu_alpha && (p.o.act == CWFA_ACT_PRELU || p.o.act2 == CWFA_ACT_PRELU)) ? *p.o.prelu_alpha : 0.f;
        const float* rb = p.o.residual ? p.o.residual + (int64_t)b * p.o.res_bs + (int64_t)orow * p.W + ocol : nullptr;
        __syncthreads();
        float* mine = smem + wave * 4096 + lane;                 // [16 rows][4 outputs] x 64 lanes per wave
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float yv[2][2];
            out_transform(r, yv);
#pragma unroll
            for (int k = 0; k < 4; ++k) mine[(r * 4 + k) * 64] = yv[k >> 1][k & 1];
        }
        for (int r = 0; r < 16; ++r) {
            const int co = ct * C::CT + wm * 32 + acc_row(r, kh);
            if (co >= p.Cout) continue;
            const float bias = p.o.bias ? p.o.bias[co] : 0.f;
            for (int k = 0; k < 4; ++k) {
                const int pr = k >> 1, q = k & 1;
                if (!(rok[pr] && (q ? c1ok : c0ok))) continue;
                const int64_t o = (int64_t)co * HW + (int64_t)pr * p.W + q;
                float v = cwfa_act(mine[(r * 4 + k) * 64] + bias, p.o.act, alpha);
                if (rb) v += rb[o];
                yb[o] = cwfa_act(v, p.o.act2, alpha);
            }
        }
    } else {
        float alpha = 0.f;
        if constexpr (EPI == W2EPI_PRELU) alpha = *p.o.prelu_alpha;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = ct * C::CT + wm * 32 + acc_row(r, kh);
            if (co >= p.Cout) continue;
            const float bias = p.o.bias ? p.o.bias[co] : 0.f;
            float yv[2][2];
            out_transform(r, yv);
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
                float e0 = yv[pr][0] + bias, e1 = yv[pr][1] + bias;
                if constexpr (EPI == W2EPI_PRELU) {
                    e0 = e0 > 0.f ? e0 : alpha * e0;
                    e1 = e1 > 0.f ? e1 : alpha * e1;
                }
                float* dst = yb + (int64_t)co * HW + (int64_t)pr * p.W;
                if (!rok[pr]) continue;
                if (vec) {
                    const f32x2 o2 = {e0, e1};
                    *reinterpret_cast<f32x2*>(dst) = o2;
                } else {
                    if (c0ok) dst[0] = e0;
                    if (c1ok) dst[1] = e1;
                }
            }
        }
    }
}

// ---- weight transform + repack: torch [Cout][Cin][3][3] -> [cout tile][chunk][comp pair][ck][64 cout][2]
__global__ __launch_bounds__(256) void wino2d_pack_kernel(const float* __restrict__ w, float* __restrict__ out, int Cout, int Cin,
                                                          int nchunks, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int j2 = (int)(i & 1);
    const int col = (int)((i >> 1) % W2::CT);
    const int ck = (int)((i / (2 * W2::CT)) % W2::CK);
    const int cp = (int)((i / (2 * W2::CT * W2::CK)) % 8);
    const int chunk = (int)((i / W2::US) % nchunks);
    const int ctile = (int)(i / ((int64_t)W2::US * nchunks));
    const int co = ctile * W2::CT + col, ci = chunk * W2::CK + ck;
    const int xi = cp >> 1, nu = (cp & 1) * 2 + j2;
    float v = 0.f;
    if (co < Cout && ci < Cin) {
        const float* g = w + ((int64_t)co * Cin + ci) * 9;
        // G row: (1,0,0), (.5,.5,.5), (.5,-.5,.5), (0,0,1)
        auto gmul = [](int idx, float a, float b, float c) {
            return idx == 0 ? a : idx == 1 ? ((a + b) + c) * 0.5f : idx == 2 ? ((a - b) + c) * 0.5f : c;
        };
        float t[3];                    // (G g)[xi][kx]
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) t[kx] = gmul(xi, g[kx], g[3 + kx], g[6 + kx]);
        v = gmul(nu, t[0], t[1], t[2]);
    }
    out[i] = v;
}

template <int EPI, bool PRO, bool ALIGNED>
int launch2d(W2Params p, hipStream_t stream) {
    typedef W2 C;
    p.tiles_x = (p.W + C::TCOLS - 1) / C::TCOLS;
    p.tiles_y = (p.H + 3) / 4;
    p.nchunks = (p.Cin + C::CK - 1) / C::CK;
    const int ctiles = (p.Cout + C::CT - 1) / C::CT;
    CWFA_REQUIRE((int64_t)p.tiles_x * p.tiles_y * ctiles < (1ll << 31) && p.B <= 65535, CWFA_E_SHAPE,
                 "cwfa_conv2d_f32: grid too large");
    CWFA_REQUIRE((int64_t)(p.Cin + 4 * C::CK) * p.H * p.W * 4 < (1ll << 31), CWFA_E_SHAPE,
                 "cwfa_conv2d_f32: one sample's input must stay below 2 GiB (32-bit buffer offsets)");
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_wino2d_kernel<EPI, PRO, ALIGNED>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
        if (e != hipSuccess) {
            cwfa_set_error("cwfa_conv2d_f32: hipFuncSetAttribute(%d bytes LDS): %s", C::LDS_BYTES, hipGetErrorString(e));
            return CWFA_E_HIP;
        }
        attr_set = true;
    }
    dim3 grid((unsigned)(p.tiles_x * p.tiles_y * ctiles), 1, p.B);
    hipLaunchKernelGGL((conv3x3_wino2d_kernel<EPI, PRO, ALIGNED>), grid, dim3(C::NTHREADS), C::LDS_BYTES, stream, p);
    CWFA_LAUNCH_CHECK("cwfa_conv2d_f32 (winograd 2-D)");
    return CWFA_OK;
}

template <bool ALIGNED>
int dispatch2d(const W2Params& p, hipStream_t stream) {
    const cwfa_conv_opts& o = p.o;
    const bool pro = o.in_scale || o.in_add;
    const bool plain = !o.residual && o.act2 == CWFA_ACT_NONE;
    if (plain && o.act == CWFA_ACT_PRELU) return pro ? launch2d<W2EPI_PRELU, true, ALIGNED>(p, stream) : launch2d<W2EPI_PRELU, false, ALIGNED>(p, stream);
    if (plain && o.act == CWFA_ACT_NONE && !pro) return launch2d<W2EPI_NONE, false, ALIGNED>(p, stream);
    return pro ? launch2d<W2EPI_GENERIC, true, ALIGNED>(p, stream) : launch2d<W2EPI_GENERIC, false, ALIGNED>(p, stream);
}

}  // namespace

int64_t cwfa_wino2d_packed_floats(int Cout, int Cin) {
    const int64_t ctiles = (Cout + W2::CT - 1) / W2::CT, nchunks = (Cin + W2::CK - 1) / W2::CK;
    return ctiles * nchunks * W2::US;
}

int cwfa_wino2d_pack(const float* w, float* packed, int Cout, int Cin, hipStream_t stream) {
    const int64_t total = cwfa_wino2d_packed_floats(Cout, Cin);
    const int nchunks = (Cin + W2::CK - 1) / W2::CK;
    hipLaunchKernelGGL(wino2d_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, w, packed, Cout, Cin,
                       nchunks, total);
    CWFA_LAUNCH_CHECK("cwfa_conv2d_pack_f32 (winograd 2-D)");
    return CWFA_OK;
}

int cwfa_wino2d_conv(const float* x, const float* w_packed, float* y, int B, int Cin, int H, int W, int Cout, int64_t x_bs,
                     int64_t y_bs, const cwfa_conv_opts& o, hipStream_t stream) {
    W2Params p{};
    p.x = x; p.wp = w_packed; p.y = y;
    p.B = B; p.Cin = Cin; p.H = H; p.W = W; p.Cout = Cout;
    p.x_bs = x_bs; p.y_bs = y_bs;
    p.o = o;
    const bool aligned = (W & 3) == 0 && (x_bs & 3) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 &&
                         (!o.in_add || ((o.in_add_bs & 3) == 0 && (reinterpret_cast<uintptr_t>(o.in_add) & 15) == 0));
    return aligned ? dispatch2d<true>(p, stream) : dispatch2d<false>(p, stream);
}
