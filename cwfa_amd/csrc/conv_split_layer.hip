// Fused residual layer of a coupling sub-network on the bf16 matrix cores with fp32-equivalent arithmetic:
//
//     y = ELU( W1 . ELU( conv3x3(x, W3) + b3 ) + b1 + x ),   64 channels      (networks.py:624-631, 660-665)
//
// Every fp32 operand is split EXACTLY into three bf16 pieces (v = v1 + v2 + v3: 3 x 8 = 24 significand bits) and the six
// partial products with i + j <= 4 are accumulated in fp32 by v_mfma_f32_32x32x16_bf16 (dropped terms <= 2^-24 relative).
// With `products == 1` only the leading piece is used: plain bf16 operands (BASELINE.json configs[4]).
//
// One PERSISTENT workgroup per CU (512 threads = 8 waves, two per SIMD) walks tiles of 64 channels x 16 rows x 32 pixels;
// wave w owns image rows 2w, 2w+1 of the tile (two n-tiles) and both 32-channel m-tiles.  A tile is 40 steps of
// 24 MFMAs per wave:
//   steps  0..35  3x3 conv: 4 chunks of 16 input channels x 9 taps.  The haloed input tile of a chunk,
//                 [piece 3][k half 2][18 rows][34 px] x 16 B (8 channels of one pixel = one B fragment), sits in one of
//                 TWO LDS buffers; a tap only shifts the B-operand address by a constant.  The NEXT chunk (or the next
//                 tile's first) is read from the fp32 tensor during the first three steps of a chunk (buffer loads: padding
//                 and the image border come back as 0.0 from the range check), split and stored into the other buffer
//                 in steps 4..6 -- the vector work rides beside the partner wave's MFMAs (bf16 MFMA and VALU are
//                 separate pipes).
//   steps 36..39  1x1 conv on the same cores: the 3x3 accumulators (+b3, ELU, split in registers) ARE its B operand,
//                 register by register (the k order inside a step is permuted accordingly when W1 is packed).
// Weights: 40 slices of 6 KB ([piece][k half][64 cout][8]) stream through a three-buffer LDS ring by LDS-DMA; the
// slice of step s+3 is issued in step s into the buffer whose fragments the wave already holds in registers (the
// A fragments of step s+1 are read during step s), one barrier per step.  Biases ride in the accumulator init.
#include "conv_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

extern int g_cwfa_split_products;       // conv2d.hip ("split_products" option: 6 or 1)

namespace {

constexpr int TR = 16, TC = 32, XR = TR + 2, XC = TC + 2;
constexpr int XE = 2 * XR * XC;             // 1224 entries (k half, row, col) of 16 bytes per piece
constexpr int XPB = XE * 16;                // bytes of one piece plane
constexpr int XB = 3 * XPB;                 // bytes of one input buffer (58 752)
constexpr int WSL = 3 * 2 * 64 * 16;        // bytes of one weight slice (6 144)
constexpr int NSL = 40;                     // 36 slices of the 3x3 bank + 4 of the 1x1 bank
constexpr int OFF_W = 2 * XB;
constexpr int OFF_DUMP = OFF_W + 3 * WSL;   // 2 KB: where waves 6, 7 send their (out-of-range, zero) share of a slice DMA
constexpr int OFF_BIAS = OFF_DUMP + 2048;   // b3[64], b1[64]
constexpr int LDS_BYTES = OFF_BIAS + 512;   // 138 496
static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");

struct LParams {
    const float* x;
    float* y;
    const void* wp;
    const float* b3;
    const float* b1;
    int B, H, W, tiles_x, tiles_y, ntiles;
    int64_t x_bs, y_bs;
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

__device__ __forceinline__ int acc_row(int r, int kh) { return (r & 3) + 8 * (r >> 2) + 4 * kh; }

__device__ __forceinline__ float elu(float v) {
    const float e = __expf(v) - 1.0f;
    return v > 0.f ? v : e;
}

// v = a1 + a2 + a3 exactly (each difference is exact: the subtrahend is the minuend rounded to 8 significant bits)
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

#define MFMA(a, b, c) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)

template <bool SIX>
__global__ __launch_bounds__(512, 1) void split_layer_kernel(LParams p) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kh = lane >> 5, l31 = lane & 31;
    const int HW = p.H * p.W;
    const int plane = HW * 4;
    constexpr unsigned OOB = 0x80000000u;
    typedef __attribute__((address_space(3))) void* lds_ptr;

    // ---- staging entries e = tid + k*512 < XE of the [k half][18 rows][34 px] tile: geometry inside the tile
    int er[3], ec[3], eh[3];
    bool fin[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int e = tid + k * 512;
        fin[k] = e < XE;
        ec[k] = e % XC;
        er[k] = (e / XC) % XR;
        eh[k] = e / (XR * XC);
    }
    const bool stage2 = wave < 4;              // entry 2 exists only for tid < 200 (uniform per wave up to wave 3)

    float xv[3][8];
    unsigned fo[3];
    auto rsrc_of = [&](const float* base) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, 64 * plane, 0x00020000);
    };
    auto tile_coords = [&](int tile, int& b, int& row0, int& col0) {
        const int per = p.tiles_x * p.tiles_y;
        b = tile / per;
        const int rem = tile - b * per;
        const int ty = rem / p.tiles_x;
        row0 = ty * TR;
        col0 = (rem - ty * p.tiles_x) * TC;
    };
    auto set_stage_tile = [&](bool valid, int row0, int col0) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int gr = row0 + er[k] - 1, gc = col0 + ec[k] - 1;
            const bool ok = valid && fin[k] && gr >= 0 && gr < p.H && gc >= 0 && gc < p.W;
            fo[k] = ok ? (unsigned)((eh[k] * 8 * HW + gr * p.W + gc) * 4) : OOB;
        }
    };
    auto load_entry = [&](auto kc, const float* base, int chunk) {
        constexpr int k = decltype(kc)::value;
        const auto rs = rsrc_of(base);
#pragma unroll
        for (int j = 0; j < 8; ++j)
            xv[k][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, fo[k], (chunk * 16 + j) * plane, 0));
    };
    auto store_entry = [&](auto kc, int buf) {
        constexpr int k = decltype(kc)::value;
        if (k == 2 && !stage2) return;
        bf16x8 pc[3];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            __bf16 a1, a2 = (__bf16)0.f, a3 = (__bf16)0.f;
            split3<SIX>(xv[k][j], a1, a2, a3);
            pc[0][j] = a1; pc[1][j] = a2; pc[2][j] = a3;
        }
        if (fin[k]) {
            char* dst = lds + buf * XB + (tid + k * 512) * 16;
#pragma unroll
            for (int q = 0; q < (SIX ? 3 : 1); ++q) *reinterpret_cast<bf16x8*>(dst + q * XPB) = pc[q];
        }
    };

    // ---- weight slices: 384 x 16 B, waves 0..5 carry one KB each; waves 6, 7 issue the same instruction into the dump area
    const auto rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.wp), 0, NSL * WSL, 0x00020000);
    const unsigned dma_off = wave < 6 ? (unsigned)tid * 16u : OOB;
    auto dma_w = [&](int slice, int wb) {
        char* dst = lds + (wave < 6 ? OFF_W + wb * WSL + wave * 1024 : OFF_DUMP + (wave - 6) * 1024);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr)dst, 16, dma_off, slice * WSL, 0, 0);
    };

    const char* alane = lds + OFF_W + (kh * 64 + l31) * 16;                               // + wb*WSL + (q*128 + m*32)*16
    const char* blane = lds + ((kh * XR + 2 * wave) * XC + l31) * 16;                     // + buf*XB + q*XPB + ((n+dy)*XC + dx)*16
    const float* bias = reinterpret_cast<const float*>(lds + OFF_BIAS);

    bf16x8 A[2][2][3], Bq[2][3];
    f32x16 acc[2][2];

    auto read_a = [&](auto setc, int wb) {
        constexpr int set = decltype(setc)::value;
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int q = 0; q < (SIX ? 3 : 1); ++q)
                A[set][m][q] = *reinterpret_cast<const bf16x8*>(alane + wb * WSL + (q * 128 + m * 32) * 16);
    };
    auto read_b = [&](int n, int buf, int dy, int dx) {
#pragma unroll
        for (int q = 0; q < (SIX ? 3 : 1); ++q)
            Bq[n][q] = *reinterpret_cast<const bf16x8*>(blane + buf * XB + q * XPB + ((n + dy) * XC + dx) * 16);
    };
    // six products of one (m, n) pair, smallest terms first; A fragments from register set `set`
    auto mfma6 = [&](f32x16& c, const bf16x8 (&a)[3], const bf16x8 (&b)[3]) {
        if constexpr (SIX) {
            MFMA(a[2], b[0], c);
            MFMA(a[1], b[1], c);
            MFMA(a[0], b[2], c);
            MFMA(a[1], b[0], c);
            MFMA(a[0], b[1], c);
        }
        MFMA(a[0], b[0], c);
    };

    // ---------------------------------------------------------------------------------------------- prologue
    if (tid < 128) reinterpret_cast<float*>(lds + OFF_BIAS)[tid] = tid < 64 ? p.b3[tid] : p.b1[tid - 64];
    int tile = blockIdx.x;
    int tb, row0, col0;
    tile_coords(tile, tb, row0, col0);
    const float* xs_cur = p.x + (int64_t)tb * p.x_bs;         // sample the staging loads of this tile's chunks read
    set_stage_tile(true, row0, col0);
    sfor<3>([&](auto kc) { load_entry(kc, xs_cur, 0); });
    dma_w(0, 0);
    dma_w(1, 1);
    dma_w(2, 2);
    sfor<3>([&](auto kc) { store_entry(kc, 0); });
    asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    read_a(ic<0>{}, 0);
    read_b(0, 0, 0, 0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();           // every wave holds slice 0: step 0 may overwrite its buffer
    int wb = 0;                             // ring buffer of the current step's slice

    for (; tile < p.ntiles; tile += gridDim.x) {
        // next tile (staged during this tile's last chunk)
        const int ntile = tile + gridDim.x;
        const bool has_next = ntile < p.ntiles;
        int nb, nrow0, ncol0;
        tile_coords(has_next ? ntile : tile, nb, nrow0, ncol0);
        const float* xs_next = p.x + (int64_t)nb * p.x_bs;

        // accumulators start from the conv bias
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float bv = bias[m * 32 + acc_row(r, kh)];
                acc[m][0][r] = bv;
                acc[m][1][r] = bv;
            }

        // -------------------------------------------------------------------------------- 3x3: chunks of 16 channels
        auto chunk_steps = [&](auto parc, int chunk) {
            constexpr int PAR = decltype(parc)::value;            // input buffer of this chunk; A register set parity
            const bool last = chunk == 3;
            if (last) set_stage_tile(has_next, nrow0, ncol0);
            const float* xs = last ? xs_next : xs_cur;
            const int nchunk = last ? 0 : chunk + 1;
            sfor<9>([&](auto tc) {
                constexpr int TAP = decltype(tc)::value, dy = TAP / 3, dx = TAP % 3, set = (PAR + TAP) & 1;
                constexpr int NT = (TAP + 1) % 9, ndy = NT / 3, ndx = NT % 3;
                constexpr int S = TAP;                              // slice index inside the chunk
                // staging loads of the next chunk first, then this step's slice DMA (slice s+3 -> the buffer of slice s)
                if constexpr (TAP < 3) load_entry(ic<TAP>{}, xs, nchunk);
                {
                    int sl = chunk * 9 + S + 3;
                    sl = sl >= NSL ? sl - NSL : sl;
                    dma_w(sl, wb);
                }
                const int wb1 = wb == 2 ? 0 : wb + 1;
                read_b(1, PAR, dy, dx);
                __builtin_amdgcn_sched_barrier(0);
                mfma6(acc[0][0], A[set][0], Bq[0]);
                __builtin_amdgcn_sched_barrier(0);
                read_a(ic<set ^ 1>{}, wb1);
                __builtin_amdgcn_sched_barrier(0);
                mfma6(acc[1][0], A[set][1], Bq[0]);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (TAP >= 4 && TAP < 7) store_entry(ic<TAP - 4>{}, PAR ^ 1);
                __builtin_amdgcn_sched_barrier(0);
                mfma6(acc[0][1], A[set][0], Bq[1]);
                __builtin_amdgcn_sched_barrier(0);
                // first B fragment of the next step (tap 8: the next chunk's tile, complete since the barrier of step 6)
                read_b(0, TAP == 8 ? PAR ^ 1 : PAR, ndy, ndx);
                __builtin_amdgcn_sched_barrier(0);
                mfma6(acc[1][1], A[set][1], Bq[1]);
                __builtin_amdgcn_sched_barrier(0);
                // the DMA of the previous step (slice s+2) has landed; this step's own loads may stay in flight
                if constexpr (TAP < 3) asm volatile("s_waitcnt vmcnt(9)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(1)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                wb = wb1;
            });
        };
        chunk_steps(ic<0>{}, 0);
        chunk_steps(ic<1>{}, 1);
        chunk_steps(ic<0>{}, 2);
        chunk_steps(ic<1>{}, 3);

        // -------------------------------------------------------------------------------- 1x1 on the hidden accumulators
        f32x16 acc2[2][2];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float bv = bias[64 + m * 32 + acc_row(r, kh)];
                acc2[m][0][r] = bv;
                acc2[m][1][r] = bv;
            }
        sfor<4>([&](auto kcc) {
            constexpr int KC = decltype(kcc)::value, set = KC & 1, hm = KC >> 1, r0 = 8 * (KC & 1);
            {
                int sl = 36 + KC + 3;
                sl = sl >= NSL ? sl - NSL : sl;
                dma_w(sl, wb);
            }
            const int wb1 = wb == 2 ? 0 : wb + 1;
            bf16x8 Hq[2][3];
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    __bf16 a1, a2 = (__bf16)0.f, a3 = (__bf16)0.f;
                    split3<SIX>(elu(acc[hm][n][r0 + j]), a1, a2, a3);
                    Hq[n][0][j] = a1; Hq[n][1][j] = a2; Hq[n][2][j] = a3;
                }
            __builtin_amdgcn_sched_barrier(0);
            mfma6(acc2[0][0], A[set][0], Hq[0]);
            __builtin_amdgcn_sched_barrier(0);
            read_a(ic<set ^ 1>{}, wb1);
            __builtin_amdgcn_sched_barrier(0);
            mfma6(acc2[1][0], A[set][1], Hq[0]);
            mfma6(acc2[0][1], A[set][0], Hq[1]);
            mfma6(acc2[1][1], A[set][1], Hq[1]);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt vmcnt(1)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            wb = wb1;
        });

        // -------------------------------------------------------------------------------- residual, ELU, store
        {
            // buffer accesses: a lane outside the image carries an out-of-range offset (loads give 0.0, stores are dropped),
            // the channel rides in the scalar offset -- no address arithmetic and no branches per element
            const auto rx = rsrc_of(p.x + (int64_t)tb * p.x_bs);
            const auto ry = __builtin_amdgcn_make_buffer_rsrc(p.y + (int64_t)tb * p.y_bs, 0, 64 * plane, 0x00020000);
            const int col = col0 + l31;
            unsigned oo[2];
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const int row = row0 + 2 * wave + n;
                oo[n] = (col < p.W && row < p.H) ? (unsigned)(row * p.W + col) * 4u + (unsigned)kh * 4u * (unsigned)plane : OOB;
            }
            f32x16 res[2];
            auto load_res = [&](int g, f32x16& rv) {
                const int m = g >> 1, n = g & 1;
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    rv[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, oo[n], (m * 32 + acc_row(r, 0)) * plane, 0));
            };
            load_res(0, res[0]);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int m = g >> 1, n = g & 1;
                if (g < 3) load_res(g + 1, res[(g + 1) & 1]);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v = elu(acc2[m][n][r] + res[g & 1][r]);
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ry, oo[n], (m * 32 + acc_row(r, 0)) * plane, 0);
                }
            }
        }
        tb = nb; row0 = nrow0; col0 = ncol0;
        xs_cur = xs_next;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // slice DMAs issued past the end
}

// packed image: 40 slices [piece 3][k half 2][64 cout][8] of bf16.
//   slices 0..35  (chunk c, tap t): element j of half h = w3[co][c*16 + h*8 + j][t]
//   slices 36..39 (k step kc of the 1x1): element j of half h = w1[co][ch], ch = 16*kc + 8*(j>>2) + 4*h + (j&3)
//                 -- the hidden channel that register 8*(kc&1) + j of accumulator tile kc>>1 holds in lane half h
__global__ __launch_bounds__(256) void split_layer_pack_kernel(const float* __restrict__ w3, const float* __restrict__ w1,
                                                               uint4* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;            // over [slice 40][h 2][co 64]
    if (i >= NSL * 128) return;
    const int co = i % 64, h = (i / 64) % 2, sl = i / 128;
    unsigned short pc[3][8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float v;
        if (sl < 36) {
            const int c = sl / 9, t = sl % 9;
            v = w3[((int64_t)co * 64 + c * 16 + h * 8 + j) * 9 + t];
        } else {
            const int kc = sl - 36;
            v = w1[co * 64 + 16 * kc + 8 * (j >> 2) + 4 * h + (j & 3)];
        }
        __bf16 a1, a2, a3;
        split3<true>(v, a1, a2, a3);
        pc[0][j] = __builtin_bit_cast(unsigned short, a1);
        pc[1][j] = __builtin_bit_cast(unsigned short, a2);
        pc[2][j] = __builtin_bit_cast(unsigned short, a3);
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        uint4 u;
        u.x = pc[q][0] | ((unsigned)pc[q][1] << 16);
        u.y = pc[q][2] | ((unsigned)pc[q][3] << 16);
        u.z = pc[q][4] | ((unsigned)pc[q][5] << 16);
        u.w = pc[q][6] | ((unsigned)pc[q][7] << 16);
        out[(sl * 3 + q) * 128 + h * 64 + co] = u;
    }
}

int g_num_cus = 0;

}  // namespace

extern "C" int64_t cwfa_subnet_layer_split_packed_bytes(void) { return (int64_t)NSL * WSL; }

extern "C" int cwfa_subnet_layer_split_pack_f32(const float* w3, const float* w1, void* packed, void* stream) {
    CWFA_REQUIRE(w3 && w1 && packed, CWFA_E_INVAL, "cwfa_subnet_layer_split_pack_f32: null pointer");
    CWFA_REQUIRE(cwfa_aligned16(packed), CWFA_E_ALIGN, "cwfa_subnet_layer_split_pack_f32: packed image must be 16-byte aligned");
    hipLaunchKernelGGL(split_layer_pack_kernel, dim3((NSL * 128 + 255) / 256), dim3(256), 0, (hipStream_t)stream, w3, w1,
                       reinterpret_cast<uint4*>(packed));
    CWFA_LAUNCH_CHECK("cwfa_subnet_layer_split_pack_f32");
    return CWFA_OK;
}

extern "C" int cwfa_subnet_layer_split_f32(const float* x, const void* packed, const float* b3, const float* b1, float* y, int B,
                                           int H, int W, int64_t x_bs, int64_t y_bs, void* stream) {
    CWFA_REQUIRE(B >= 0 && H >= 0 && W >= 0, CWFA_E_INVAL, "cwfa_subnet_layer_split_f32: bad size");
    if (B == 0 || H == 0 || W == 0) return CWFA_OK;
    CWFA_REQUIRE(x && packed && b3 && b1 && y, CWFA_E_INVAL, "cwfa_subnet_layer_split_f32: null pointer");
    CWFA_REQUIRE(x != y, CWFA_E_INVAL, "cwfa_subnet_layer_split_f32: in-place not supported (3x3 halo)");
    CWFA_REQUIRE(cwfa_aligned16(packed), CWFA_E_ALIGN, "cwfa_subnet_layer_split_f32: packed image must be 16-byte aligned");
    CWFA_REQUIRE((int64_t)64 * H * W * 4 < (1ll << 31), CWFA_E_SHAPE, "cwfa_subnet_layer_split_f32: image too large for 32-bit offsets");
    LParams p{};
    p.x = x; p.y = y; p.wp = packed; p.b3 = b3; p.b1 = b1;
    p.B = B; p.H = H; p.W = W; p.x_bs = x_bs; p.y_bs = y_bs;
    p.tiles_x = (W + TC - 1) / TC;
    p.tiles_y = (H + TR - 1) / TR;
    const int64_t ntiles = (int64_t)p.tiles_x * p.tiles_y * B;
    CWFA_REQUIRE(ntiles < (1ll << 31), CWFA_E_SHAPE, "cwfa_subnet_layer_split_f32: too many tiles");
    p.ntiles = (int)ntiles;
    if (g_num_cus == 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) {
            cwfa_set_error("cwfa_subnet_layer_split_f32: cannot query the CU count");
            return CWFA_E_HIP;
        }
        g_num_cus = n;
    }
    const bool six = g_cwfa_split_products != 1;
    auto kern = six ? &split_layer_kernel<true> : &split_layer_kernel<false>;
    static bool attr_set[2] = {false, false};
    if (!attr_set[six]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        if (e != hipSuccess) {
            cwfa_set_error("cwfa_subnet_layer_split_f32: hipFuncSetAttribute(%d bytes LDS): %s", LDS_BYTES, hipGetErrorString(e));
            return CWFA_E_HIP;
        }
        attr_set[six] = true;
    }
    const int grid = (int)(ntiles < g_num_cus ? ntiles : g_num_cus);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), LDS_BYTES, (hipStream_t)stream, p);
    CWFA_LAUNCH_CHECK("cwfa_subnet_layer_split_f32");
    return CWFA_OK;
}
