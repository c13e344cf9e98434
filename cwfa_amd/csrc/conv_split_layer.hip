// Fused residual layer of a coupling sub-network on the bf16 matrix cores with fp32-equivalent arithmetic:
//
//     y = ELU( W1 . ELU( conv3x3(x, W3) + b3 ) + b1 + x ),   64 channels      (networks.py:624-631, 660-665)
//
// Every fp32 operand is split EXACTLY into three bf16 pieces (v = v1 + v2 + v3: 3 x 8 = 24 significand bits) and the six
// partial products with i + j <= 4 are accumulated in fp32 by v_mfma_f32_16x16x32_bf16 (dropped terms <= 2^-24 relative).
// With `products == 1` only the leading piece is used: plain bf16 operands (BASELINE.json configs[4]).
// (16x16x32, not 32x32x16: under this load the chip is clock-limited by power and holds a 1.2x higher rate on the
//  small shape -- tools/probe/mfma_shape.hip: 2200 vs 1800 TF/s bf16 with every operand re-read from LDS.)
//
// One PERSISTENT workgroup per CU (512 threads = 8 waves, two per SIMD) walks tiles of 64 channels x 16 rows x 32 pixels;
// wave w owns image rows 2w, 2w+1 of the tile = four n-tiles of 16 pixels, and all four 16-channel m-tiles
// (16 accumulator tiles of 4 registers).  A tile is
//   18 conv steps  K = 32 per MFMA = TWO (16-channel chunk, tap) units of the 36 a 3x3 over 64 channels has; lane group
//                  g = lane >> 4 reads k-half g & 1 of unit g >> 1.  The haloed input tile of a chunk,
//                  [piece 3][k half 2][18 rows][34 px] x 16 B (8 channels of one pixel = one B fragment), sits in one of
//                  TWO LDS buffers (even / odd chunks); a tap only shifts the B-operand address.  Steps pair the taps
//                  (0,1)(2,3)(4,5)(6,7) of an even chunk, tap 8 with tap 0 of the following odd chunk (both buffers
//                  live), then (1,2)(3,4)(5,6)(7,8): the even buffer is free from the 6th step of such a 9-step period
//                  and is refilled there (loads three steps earlier: buffer loads, padding and the image border come
//                  back as 0.0 from the range check; split in registers, three 16-byte LDS stores per entry), the odd
//                  buffer in the first three steps.  The vector work rides beside the partner wave's MFMAs.
//   the 1x1 phase  per n-tile: the 3x3 accumulators (+b3 via the accumulator init, ELU, split in registers) ARE the B
//                  operand of the 1x1 GEMM, register by register (the k order inside a step is permuted accordingly
//                  when W1 is packed); its 16 results overwrite the accumulators they came from, and the epilogue
//                  (+x, ELU, store) of n-tile i is slotted between the MFMAs of n-tile i+1.
// Layouts.  NCHW planes cost this kernel one dword instruction per (channel, pixel run): 8 loads per staging entry, 64
// residual loads and 64 stores per thread and tile -- the stores alone held 15 % of the launch.  The maps BETWEEN the layers
// of a sub-network are private to it, so they may be kept channel-blocked, [C/8][H][W][8]: a staging entry (8 channels of a
// pixel) is 32 contiguous bytes (two 16-byte loads), and the four channels 16 mt + 4 g + r a lane holds for a pixel are 16
// contiguous bytes (one 16-byte residual load and one 16-byte store per accumulator tile).
// Weights: 20 slices of 12 KB ([piece][k group 4][64 cout][8]) stream through a three-slot LDS ring by LDS-DMA; slice
// s+2 is issued at the start of step s and verified at its end (one barrier per step, 19 per tile).
#include "conv_internal.h"

#include <utility>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

extern int g_cwfa_split_products;       // conv2d.hip ("split_products" option: 6 or 1)
extern int g_cwfa_split_xcd_map;        // conv2d.hip ("split3x3_xcd_map" option; also the tile walk of this kernel)

namespace {

constexpr int TR = 16, TC = 32, XR = TR + 2, XC = TC + 2;
constexpr int KHB = 9984;                   // bytes of one k-half plane: 18 x 34 x 16 = 9792 padded to a multiple of 256 (bank phase)
constexpr int XPB = 2 * KHB;                // bytes of one piece plane
constexpr int XB = 3 * XPB;                 // bytes of one input buffer (59 904)
constexpr int WSL = 3 * 4 * 64 * 16;        // bytes of one weight slice (12 288)
constexpr int NSL = 20;                     // 18 slices of the 3x3 bank + 2 of the 1x1 bank (NPER = 1: 9 + 2)
constexpr int OFF_W = 2 * XB;
constexpr int OFF_DUMP = OFF_W + 3 * WSL;   // 4 KB: where waves 4..7 send their second (out-of-range, zero) slice DMA
constexpr int MAXP = 5;                     // problems (filter banks) per launch: what the rest of the 160 KB holds in biases
constexpr int OFF_BIAS = OFF_DUMP + 4096;   // [problem][b3[64], b1[64]]
constexpr int LDS_BYTES = OFF_BIAS + MAXP * 512;   // 163 328
static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
constexpr int NE = 2 * XR * XC;             // 1224 staging entries (k half, row, col) per chunk

struct LParams {
    const float* x;
    float* y;
    const void* wp;
    const float* b3;
    const float* b1;
    int B, H, W, tiles_x, tiles_y, ntiles;
    int64_t x_bs, y_bs;
    int in_blocked, out_blocked;      // layout of x / y: 0 = NCHW planes, 1 = [C/8][H][W][8] (see the header comment)
    int xcd_map;
    const float* u;                   // NPER = 1 ("first layer" form): the 3x3 reads THIS tensor (NCHW, u_ch <= 32 channels) with weights composed
    int64_t u_bs;                     // with the 1x1 in front of the layer; x is then only the residual
    int u_ch;
    float* hid;                       // TAPE form (training forward): the hidden map h = ELU(conv3x3(x) + b3) is written here (NCHW), from the
    int64_t hid_bs;                   // registers that hold it as the 1x1's B operand
    int nprob, spp;                   // grouped launch: sample b belongs to problem b / spp, which has its own packed image (wp + problem *
                                      // NSL * WSL) and biases (b3 / b1 + problem * 64); nprob = 1: one bank for the whole batch
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

// ELU = median(v, exp(v) - 1, 0): for v > 0, exp(v) - 1 > v > 0; for v < 0, v <= exp(v) - 1 < 0 -- one v_med3_f32 instead of a
// compare and a select (the 1x1 phase of the layer is bound by exactly these instructions: 192 ELUs per lane and tile)
__device__ __forceinline__ float elu(float v) {
    return __builtin_amdgcn_fmed3f(v, __expf(v) - 1.0f, 0.f);
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

#define MFMA(a, b, c) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)
#define FENCE() __builtin_amdgcn_sched_barrier(0)

// (chunk, tap) unit u = 0..35 of the 3x3: byte offset of its B fragments relative to the lane base
__host__ __device__ constexpr int unit_off(int u) {
    const int chunk = u / 9, tap = u % 9;
    return (chunk & 1) * XB + ((tap / 3) * XC + tap % 3) * 16;
}

// NPER = number of 9-step periods of the 3x3 = pairs of 16-channel input chunks: 2 for the 64-channel layer; 1 for the FIRST layer of
// a sub-network in its composed form: conv3x3(conv1x1(u) + b0) = conv3x3'(u | 1) with W' = W3 o [W0 | b0] over the <= 31 channels of
// the sub-network's input u and a constant-one channel (exact with zero padding: the padded ones carry no bias), K = 9 x 32 instead
// of 9 x 64 -- half the conv steps; the residual x = conv1x1(u) + b0 is still read from memory.
// XF (first-layer form only): the residual x = conv1x1(u) + b0 is not read either -- it is a THIRD k step of the 1x1 phase, [W1 | W0'] .
// [h ; u | 1], with u's values at the tile's pixels loaded straight into the B-fragment layout (8 channels of a pixel per lane, L2-hot:
// the staging just read the same tile) and W0' = [W0 | b0 | 0] as one more slice of the packed image, read from memory (12 KB,
// cache-resident; the LDS is full): the sub-network's first 1x1 launch and its 64-channel map disappear.
// SHORT (with XF): u has <= 16 channels (the coarse steps: 13 / 7 + the ones channel), i.e. ONE 16-channel chunk -- the odd chunk is all
// zeros and the four steps that pair only its taps add nothing: five conv steps per tile.  The odd LDS buffer is zeroed once (its tap 0
// rides in step 4 beside tap 8 against zero weights); the next tile's chunk is requested during steps 0 .. 2 and written into the even
// buffer behind step 4's barrier, at the head of the 1x1 phase (whose closing barrier publishes it).
template <bool SIX, bool INB, bool OUTB, int NPER = 2, bool TAPE = false, bool XF = false, bool SHORT = false>
__global__ __launch_bounds__(512, 1) void split_layer_kernel(LParams p) {
    static_assert(!TAPE || (!INB && !OUTB && NPER == 2), "tape form: NCHW maps, full layer");
    static_assert(!XF || (NPER == 1 && !INB && !TAPE), "fused first map: the composed first-layer form");
    static_assert(!SHORT || XF, "short form: the composed first layer with its fused first map");
    constexpr int NSTEP = SHORT ? 5 : 9 * NPER, NSLK = NSTEP + 2 + (XF ? 1 : 0);     // conv steps per tile; weight slices per problem
    constexpr bool UIN = NPER == 1;                      // the 3x3 input is p.u (NCHW), not p.x
    constexpr bool INS = INB && !UIN;                    // layout of the STAGED tensor
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c16 = lane & 15, g = lane >> 4;
    const int HW = p.H * p.W;
    const int plane = HW * 4;
    constexpr unsigned OOB = 0x80000000u;
    constexpr int NQ = SIX ? 3 : 1;
    typedef __attribute__((address_space(3))) void* lds_ptr;

    // ---- staging entries e = tid + k*512 < NE of the [k half][18 rows][34 px] tile
    // (kept as ONE packed word + the LDS destination per entry: the kernel sits at the 256-register limit)
    int epk[3], edst[3];                       // column | row << 8 | k half << 16;  byte offset of the entry in an input buffer
    bool fin[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int e = tid + k * 512;
        fin[k] = e < NE;
        const int c_ = e % XC, r_ = (e / XC) % XR, h_ = (e / (XR * XC)) & 1;
        epk[k] = c_ | (r_ << 8) | (h_ << 16);
        edst[k] = h_ * KHB + (r_ * XC + c_) * 16;
    }
    const bool stage2 = wave < 4;              // entry 2 exists only for tid < 200: waves 0..3 (wave 3 partly)
    auto entry_offsets = [&](bool valid, int row0, int col0, unsigned (&fo)[3]) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int gr = row0 + ((epk[k] >> 8) & 255) - 1, gc = col0 + (epk[k] & 255) - 1, eh = epk[k] >> 16;
            const bool ok = valid && fin[k] && gr >= 0 && gr < p.H && gc >= 0 && gc < p.W;
            // NCHW: channel 8 eh of the chunk, pixel (gr, gc); blocked: 32-byte entry (gr, gc) of channel block eh of the chunk
            fo[k] = !ok ? OOB : INS ? (unsigned)((eh * HW + gr * p.W + gc) * 32) : (unsigned)((eh * 8 * HW + gr * p.W + gc) * 4);
        }
    };
    auto rsrc_of = [&](const float* base) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, 64 * plane, 0x00020000);
    };
    // XCD-aware walk: workgroup w sits on XCD w & 7 (round-robin dispatch), so the k-th tile of workgroup w -- logical index
    // w + k * gridDim.x -- is taken from XCD (w & 7)'s own contiguous eighth of the tile list: neighbouring tiles (shared halo
    // rows / columns) are then read through ONE L2 instead of eight.
    const bool xmap = p.xcd_map && (p.ntiles & 7) == 0 && (gridDim.x & 7) == 0;
    auto tile_coords = [&](int it, int& b, int& row0, int& col0) {
        const int tile = xmap ? (it & 7) * (p.ntiles >> 3) + (it >> 3) : it;
        const int per = p.tiles_x * p.tiles_y;
        b = tile / per;
        const int rem = tile - b * per;
        const int ty = rem / p.tiles_x;
        row0 = ty * TR;
        col0 = (rem - ty * p.tiles_x) * TC;
    };
    auto load_entry = [&](f32x4 (&xv)[2], const float* base, unsigned fo, int chunk) {
        // (the staged tensor: 64 channels of x, or the u_ch channels of u -- channels past the end read 0.0)
        const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, (UIN ? p.u_ch : 64) * plane, 0x00020000);
        if constexpr (INS) {                          // chunk = channel blocks 2 chunk, 2 chunk + 1: 16 planes further
            xv[0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, fo, chunk * 16 * plane, 0));
            xv[1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, fo, chunk * 16 * plane + 16, 0));   // (+16 in the SCALAR offset)
            return;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j)
            xv[j >> 2][j & 3] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, fo, (chunk * 16 + j) * plane, 0));
    };
    auto store_entry = [&](auto kc, const f32x4 (&xv)[2], int buf) {
        constexpr int k = decltype(kc)::value;
        if (k == 2 && !stage2) return;
        bf16x8 pc[3];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            __bf16 a1, a2 = (__bf16)0.f, a3 = (__bf16)0.f;
            split3<SIX>(xv[j >> 2][j & 3], a1, a2, a3);
            pc[0][j] = a1; pc[1][j] = a2; pc[2][j] = a3;
        }
        if (fin[k]) {
            char* dst = lds + buf * XB + edst[k];
#pragma unroll
            for (int q = 0; q < NQ; ++q) *reinterpret_cast<bf16x8*>(dst + q * XPB) = pc[q];
        }
    };

    // ---- weight slices: 12 x 1 KB; every wave issues two DMA instructions (waves 4..7: the second into the dump area)
    const auto rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.wp), 0, p.nprob * NSLK * WSL, 0x00020000);
    // `wb`: byte offset of the problem's packed image (scalar)
    auto dma_w = [&](int slice, int slot, int wb) {
        if constexpr (!SIX) {                        // plain bf16: only the leading piece plane (the first 4 KB) of a slice is read
            if (wave < 4)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr)(lds + OFF_W + slot * WSL + wave * 1024), 16, (unsigned)tid * 16u,
                                                         wb + slice * WSL, 0, 0);
            return;
        }
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr)(lds + OFF_W + slot * WSL + wave * 1024), 16, (unsigned)tid * 16u,
                                                 wb + slice * WSL, 0, 0);
        char* d2 = lds + (wave < 4 ? OFF_W + slot * WSL + 8192 + wave * 1024 : OFF_DUMP + (wave - 4) * 1024);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr)d2, 16, wave < 4 ? 8192u + (unsigned)tid * 16u : OOB, wb + slice * WSL, 0, 0);
    };
    auto problem_of = [&](int b) { return p.nprob > 1 ? b / p.spp : 0; };
    auto next_slot = [](int s) { return s == 2 ? 0 : s + 1; };

    const int alane = OFF_W + (g * 64 + c16) * 16;                            // + slot*WSL + (q*256 + mt*16)*16
    const int blane = (g & 1) * KHB + ((2 * wave) * XC + c16) * 16;           // + unit_off + q*XPB + ((nt>>1)*XC + 16*(nt&1))*16
    const bool sel = (g >> 1) != 0;                                           // lane groups 2, 3 read the step's second unit
    const f32x4* bias_all = reinterpret_cast<const f32x4*>(lds + OFF_BIAS);  // per problem: [b3 16 x 4][b1 16 x 4]

    f32x4 acc[4][4];
    bf16x8 A[2][3], Bq[4][3];
    f32x4 xa[3][2], xb[3][2];                   // staged entries (8 channels of a pixel) of the next even / odd chunk
    unsigned fo_c[3], fo_n[3];                  // entry offsets in the current / next tile

    auto read_a = [&](int set, int abase, int mt) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) A[set][q] = *reinterpret_cast<const bf16x8*>(lds + abase + (q * 256 + mt * 16) * 16);
    };
    auto read_b = [&](int nt, int bbase) {
#pragma unroll
        for (int q = 0; q < NQ; ++q)
            Bq[nt][q] = *reinterpret_cast<const bf16x8*>(lds + bbase + q * XPB + ((nt >> 1) * XC + 16 * (nt & 1)) * 16);
    };
    auto mfma6 = [&](f32x4& c, const bf16x8 (&a)[3], const bf16x8 (&b)[3]) {
        if constexpr (SIX) {
            MFMA(a[2], b[0], c);
            MFMA(a[1], b[1], c);
            MFMA(a[0], b[2], c);
            MFMA(a[1], b[0], c);
            MFMA(a[0], b[1], c);
        }
        MFMA(a[0], b[0], c);
    };
    auto bbase_of = [&](int offA, int offB) { return blane + (sel ? offB : offA); };

    // ---------------------------------------------------------------------------------------------- prologue
    for (int e = tid; e < p.nprob * 128; e += 512) {
        const int pr = e >> 7, j = e & 127;
        reinterpret_cast<float*>(lds + OFF_BIAS)[e] = j < 64 ? p.b3[pr * 64 + j] : p.b1[pr * 64 + j - 64];
    }
    int tile = blockIdx.x;
    int tb, row0, col0;
    tile_coords(tile, tb, row0, col0);
    const float* xs_cur = UIN ? p.u + (int64_t)tb * p.u_bs : p.x + (int64_t)tb * p.x_bs;
    entry_offsets(true, row0, col0, fo_c);
    sfor<3>([&](auto kc) { load_entry(xa[decltype(kc)::value], xs_cur, fo_c[decltype(kc)::value], 0); });
    if constexpr (!SHORT) {
        sfor<3>([&](auto kc) { load_entry(xb[decltype(kc)::value], xs_cur, fo_c[decltype(kc)::value], 1); });
    } else {                                     // the odd buffer: zeros, once
#pragma unroll
        for (int k = 0; k < 3; ++k)
            if (fin[k] && (k < 2 || stage2)) {
#pragma unroll
                for (int q = 0; q < NQ; ++q) *reinterpret_cast<f32x4*>(lds + XB + edst[k] + q * XPB) = f32x4{0.f, 0.f, 0.f, 0.f};
            }
    }
    int wb_cur = problem_of(tb) * (NSLK * WSL);
    dma_w(0, 0, wb_cur);
    sfor<3>([&](auto kc) { store_entry(kc, xa[decltype(kc)::value], 0); });
    asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    int cs = 0;                                  // ring slot of the current step's slice
    read_a(0, alane, 0);
    {
        const int bb = bbase_of(unit_off(0), unit_off(1));
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) read_b(nt, bb);
    }

    for (; tile < p.ntiles; tile += gridDim.x) {
        const int ntile = tile + gridDim.x;
        const bool has_next = ntile < p.ntiles;
        int nb, nrow0, ncol0;
        tile_coords(has_next ? ntile : tile, nb, nrow0, ncol0);
        const float* xs_next = UIN ? p.u + (int64_t)nb * p.u_bs : p.x + (int64_t)nb * p.x_bs;
        entry_offsets(has_next, nrow0, ncol0, fo_n);

        const int wb_next = problem_of(nb) * (NSLK * WSL);
        const f32x4* bias4 = bias_all + (wb_cur / (NSLK * WSL)) * 32;
        // accumulators start from the conv bias (channel mt*16 + 4g + r)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const f32x4 bv = bias4[mt * 4 + g];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = bv;
        }

        // ------------------------------------------------------------------------------------------ 18 conv steps
        sfor<NSTEP>([&](auto sc) {
            constexpr int S = decltype(sc)::value, P = S % 9, PER = S / 9;
            constexpr bool LASTP = PER == NPER - 1;            // the last period stages the NEXT tile's chunks 0 and 1
            const int s1 = next_slot(cs), s2 = next_slot(s1);
            constexpr bool LOADS = SHORT ? P <= 2 : ((P >= 2 && P <= 4) || P >= 6);
            // the slice DMA(s), then this step's staging loads (which may stay in flight across the barrier) -- issued from
            // INSIDE the MFMA stream, after the first m-tile: in a burst right behind the barrier all eight waves stood in
            // their issue cost (~100 cycles per DMA instruction) at once with the matrix pipe idle
            auto issue_memory = [&]() {
                if constexpr (S == 0) dma_w(1, s1, wb_cur);     // (late by one step: the 1x1 phase counts as one)
                dma_w(S + 2, s2, wb_cur);
                if constexpr (SHORT) {                          // the one chunk of the next tile
                    if constexpr (P <= 2) load_entry(xa[P], xs_next, fo_n[P], 0);
                    return;
                }
                if constexpr (P >= 2 && P <= 4) {               // even chunk 2*PER+2 (last period: chunk 0 of the next tile)
                    constexpr int k = P - 2;
                    if constexpr (!LASTP) load_entry(xa[k], xs_cur, fo_c[k], 2 * PER + 2);
                    else load_entry(xa[k], xs_next, fo_n[k], 0);
                }
                if constexpr (P >= 6) {                         // odd chunk 2*PER+3 (last period: chunk 1 of the next tile)
                    constexpr int k = P - 6;
                    if constexpr (!LASTP) load_entry(xb[k], xs_cur, fo_c[k], 2 * PER + 3);
                    else load_entry(xb[k], xs_next, fo_n[k], 1);
                }
            };
            const int ab = alane + cs * WSL;
            if constexpr (S == 1) read_a(0, ab, 0);             // slice 1 became visible with the barrier of step 0
            constexpr int NS = (S + 1) % NSTEP;                 // next step's B base (last step: unused, the 1x1 phase follows)
            const int bbn = bbase_of(unit_off(2 * NS), unit_off(2 * NS + 1));
            FENCE();
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                if (mt < 3) read_a((mt + 1) & 1, ab, mt + 1);
                else if (S != 0 && S != NSTEP - 1) read_a(0, alane + s1 * WSL, 0);    // first fragments of the next step
                FENCE();
#pragma unroll
                for (int np = 0; np < 4; np += 2) {
                    mfma6(acc[mt][np], A[mt & 1], Bq[np]);
                    mfma6(acc[mt][np + 1], A[mt & 1], Bq[np + 1]);
                    FENCE();
                    if (mt == 3 && S != NSTEP - 1) {
                        read_b(np, bbn);
                        read_b(np + 1, bbn);
                        FENCE();
                    }
                }
                // after the first m-tile: this step's memory traffic, and the vector-heavy part of staging (split + LDS stores
                // of one entry)
                if (mt == 0) {
                    issue_memory();
                    FENCE();
                    if constexpr (!SHORT && P <= 2) store_entry(ic<P>{}, xb[P], 1);
                    if constexpr (!SHORT && P >= 5 && P <= 7) store_entry(ic<P - 5>{}, xa[P - 5], 0);
                    FENCE();
                }
            }
            // this step's slice DMA has landed; its 8 staging loads (issued after it) may stay in flight
            // (LDS: everything but the 12 youngest operations -- the B fragments just requested for the next step -- is done:
            //  this step's staging stores, and every read of the weight slot and input buffer that get overwritten next)
            // (8 four-byte loads per entry from NCHW planes, two 16-byte loads from a channel-blocked map)
            if constexpr (LOADS && !INB) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if constexpr (LOADS) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if constexpr (S != NSTEP - 1 && SIX) asm volatile("s_waitcnt lgkmcnt(12)" ::: "memory");
            else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            cs = s1;
        });

        // ------------------------------------------------------------------------------------------ 1x1 + epilogue
        if constexpr (SHORT) {      // the next tile's chunk into the even buffer: its last reader (step 4) is behind a barrier
            sfor<3>([&](auto kc) { store_entry(kc, xa[decltype(kc)::value], 0); });
        }
        {
            const int sw0 = cs, sw1 = next_slot(cs), sn0 = next_slot(sw1);         // W1 slices, next tile's slice 0
            dma_w(0, sn0, wb_next);
            const auto rx = rsrc_of(XF ? p.y : p.x + (int64_t)tb * p.x_bs);        // (XF: unused)
            const auto ru = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(XF ? p.u + (int64_t)tb * p.u_bs : p.y), 0, XF ? p.u_ch * plane : 0, 0x00020000);
            const auto ry = __builtin_amdgcn_make_buffer_rsrc(p.y + (int64_t)tb * p.y_bs, 0, 64 * plane, 0x00020000);
            const auto rh = __builtin_amdgcn_make_buffer_rsrc(TAPE ? p.hid + (int64_t)tb * p.hid_bs : p.y, 0, TAPE ? 64 * plane : 0, 0x00020000);
            // offsets of this lane's (row, col) in the two layouts: NCHW: channel 4 g (+ 16 mt + r planes in the scalar offset);
            // blocked: 16-byte half g & 1 of the 32-byte entry of channel block g >> 1 (+ 2 mt blocks = 16 mt planes)
            unsigned oo[4], ob[4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const int row = row0 + 2 * wave + (nt >> 1), col = col0 + 16 * (nt & 1) + c16;
                const bool ok = col < p.W && row < p.H;
                if constexpr (!INB || !OUTB) oo[nt] = ok ? (unsigned)(row * p.W + col) * 4u + (unsigned)g * 4u * (unsigned)plane : OOB;
                if constexpr (INB || OUTB) ob[nt] = ok ? (unsigned)((g >> 1) * HW + row * p.W + col) * 32u + (unsigned)(g & 1) * 16u : OOB;
            }
            // XF: u at this lane's pixel of n-tile nt, channels 8 g .. 8 g + 7 (= k slots 8 g + j of the third k step); two sets: the
            // next n-tile's values are requested while this one multiplies
            unsigned uo[4];
            float uv[2][8];
            if constexpr (XF) {
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    const int row = row0 + 2 * wave + (nt >> 1), col = col0 + 16 * (nt & 1) + c16;
                    uo[nt] = (col < p.W && row < p.H) ? (unsigned)(row * p.W + col) * 4u + (unsigned)(8 * g) * (unsigned)plane : OOB;
                }
            }
            auto load_u = [&](int nt, float (&dst)[8]) {
#pragma unroll
                for (int j = 0; j < 8; ++j) dst[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ru, uo[nt], j * plane, 0));
            };
            if constexpr (XF) load_u(0, uv[0]);
            f32x4 res[4];                           // residual of ONE n-tile: loaded at the end of the iteration before its epilogue
            auto load_res = [&](int nt, f32x4 (&rv)[4]) {
                if constexpr (INB) {
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt)
                        rv[mt] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, ob[nt], mt * 16 * plane, 0));
                    return;
                }
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        rv[mt][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, oo[nt], (mt * 16 + r) * plane, 0));
            };
            // +x, ELU, store of two of the 16 results of n-tile nt (piece i = 0..7); blocked output: the four results of an
            // accumulator tile leave as ONE 16-byte store with its second piece
            auto epilogue_piece = [&](int nt, int i, const f32x4 (&rv)[4]) {
                const int mt = i >> 1;
                if constexpr (OUTB) {
                    if (i & 1) {
                        f32x4 o4;
#pragma unroll
                        for (int r = 0; r < 4; ++r) o4[r] = elu(XF ? acc[mt][nt][r] : acc[mt][nt][r] + rv[mt][r]);
                        // (cwfa_buffer_store_b128: the store with the wait states its data registers need on gfx950, common.h)
                        cwfa_buffer_store_b128(__builtin_bit_cast(cwfa_u32x4, o4), ry, ob[nt], mt * 16 * plane);
                    }
                    return;
                }
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int r = (i & 1) * 2 + h;
                    const float v = elu(XF ? acc[mt][nt][r] : acc[mt][nt][r] + rv[mt][r]);
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ry, oo[nt], (mt * 16 + r) * plane, 0);
                }
            };
            const int a0 = alane + sw0 * WSL, a1 = alane + sw1 * WSL;
            // W1 fragments of sub-block i = (m-tile pair i >> 1, k step i & 1): two sets (the next one is read while this one
            // multiplies)
            constexpr int NA1 = 2;
            bf16x8 A1[NA1][2][3];
            auto read_a1 = [&](int i) {
                if (XF && i >= 4) {                 // W0' fragments of m-tile pair i - 4, from the packed image's last slice in memory
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int q = 0; q < NQ; ++q)
                            A1[i & (NA1 - 1)][h][q] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(
                                rw, (unsigned)((g * 64 + c16) * 16 + (q * 256 + ((i - 4) * 2 + h) * 16) * 16), wb_cur + (NSTEP + 2) * WSL, 0));
                    return;
                }
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int q = 0; q < NQ; ++q)
                        A1[i & (NA1 - 1)][h][q] = *reinterpret_cast<const bf16x8*>(lds + ((i & 1) ? a1 : a0) + (q * 256 + ((i >> 1) * 2 + h) * 16) * 16);
            };
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                // hidden map of this n-tile -> B fragments of the two k steps: element j of lane group g = channel
                // 16*(2s + (j>>2)) + 4g + (j&3) = register j&3 of accumulator tile 2s + (j>>2)
                bf16x8 Hq[2][3];
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        __bf16 h1, h2 = (__bf16)0.f, h3 = (__bf16)0.f;
                        const float hv = elu(acc[2 * s + (j >> 2)][nt][j & 3]);
                        if constexpr (TAPE)
                            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, hv), rh, oo[nt], ((2 * s + (j >> 2)) * 16 + (j & 3)) * plane, 0);
                        split3<SIX>(hv, h1, h2, h3);
                        Hq[s][0][j] = h1; Hq[s][1][j] = h2; Hq[s][2][j] = h3;
                    }
                if constexpr (XF) {
                    if (nt + 1 < 4) load_u(nt + 1, uv[(nt + 1) & 1]);
                }
                read_a1(0);
                FENCE();
#pragma unroll
                for (int i = 0; i < 4; ++i) {       // (m-tile pair, k step): 12 MFMAs on two accumulator tiles
                    const int mp = (i >> 1) * 2, s = i & 1;
                    if (NA1 == 2 && i + 1 < (XF ? 6 : 4)) read_a1(i + 1);
                    if (NA1 == 1 && i > 0) read_a1(i);
                    if (s == 0) {
                        acc[mp][nt] = bias4[16 + mp * 4 + g];
                        acc[mp + 1][nt] = bias4[16 + (mp + 1) * 4 + g];
                    }
                    FENCE();
                    mfma6(acc[mp][nt], A1[i & (NA1 - 1)][0], Hq[s]);
                    mfma6(acc[mp + 1][nt], A1[i & (NA1 - 1)][1], Hq[s]);
                    FENCE();
                    if (nt > 0) {
                        epilogue_piece(nt - 1, 2 * i, res);
                        epilogue_piece(nt - 1, 2 * i + 1, res);
                        FENCE();
                    }
                }
                if constexpr (XF) {                 // third k step: + W0' . (u | 1) = the first map, never formed in memory
                    bf16x8 Uq[3];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        __bf16 u1, u2 = (__bf16)0.f, u3 = (__bf16)0.f;
                        split3<SIX>(uv[nt & 1][j], u1, u2, u3);
                        Uq[0][j] = u1; Uq[1][j] = u2; Uq[2][j] = u3;
                    }
#pragma unroll
                    for (int i = 4; i < 6; ++i) {
                        const int mp = (i - 4) * 2;
                        if (i + 1 < 6) read_a1(i + 1);
                        FENCE();
                        mfma6(acc[mp][nt], A1[i & (NA1 - 1)][0], Uq);
                        mfma6(acc[mp + 1][nt], A1[i & (NA1 - 1)][1], Uq);
                        FENCE();
                    }
                } else {
                    load_res(nt, res);              // consumed by this n-tile's epilogue, one iteration later
                }
                FENCE();
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) epilogue_piece(3, i, res);
            // slice 0 of the next tile (issued at the start of this phase) has landed; the 16 youngest stores may stay in flight
            asm volatile("s_waitcnt vmcnt(16)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            cs = sn0;
            read_a(0, alane + cs * WSL, 0);
            {
                const int bb = bbase_of(unit_off(0), unit_off(1));
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) read_b(nt, bb);
            }
        }
        tb = nb; row0 = nrow0; col0 = ncol0;
        wb_cur = wb_next;
        xs_cur = xs_next;
#pragma unroll
        for (int k = 0; k < 3; ++k) fo_c[k] = fo_n[k];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // slice DMAs issued past the end
}

// packed image: 20 slices [piece 3][k group 4][64 cout][8] of bf16.
//   slices 0..17 (conv step s): group g = unit u = 2s + (g >> 1) = (chunk u / 9, tap u % 9), k half g & 1:
//                 element j = w3[co][chunk*16 + (g&1)*8 + j][tap]
//   slices 18, 19 (k step s of the 1x1): element j of group g = w1[co][16*(2s + (j>>2)) + 4g + (j&3)] -- the hidden
//                 channel that register j & 3 of accumulator tile 2s + (j >> 2) holds in lane group g
// (n3 = 18 slices over cin3 = 64 input channels, or the first-layer form: n3 = 9 over cin3 = 32)
//   first-layer form, optional slice n3 + 2 (w0 != NULL): element j of group g = w0[co][8g + j], the [64][32] matrix [W0 | b0 | 0] of the
//                 1x1 in front of the layer (third k step of the 1x1 phase: the fused first map)
__global__ __launch_bounds__(256) void split_layer_pack_kernel(const float* __restrict__ w3, const float* __restrict__ w1,
                                                               uint4* __restrict__ out, int n3, int cin3, const float* __restrict__ w0 = nullptr) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;            // over [slice n3 + 2 (+ 1)][g 4][co 64]
    if (i >= (n3 + 2 + (w0 ? 1 : 0)) * 256) return;
    const int co = i % 64, g = (i / 64) % 4, sl = i / 256;
    unsigned short pc[3][8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float v;
        if (sl < n3) {
            const int u = 2 * sl + (g >> 1), c = u / 9, t = u % 9;
            v = w3[((int64_t)co * cin3 + c * 16 + (g & 1) * 8 + j) * 9 + t];
        } else if (sl < n3 + 2) {
            const int s = sl - n3;
            v = w1[co * 64 + 16 * (2 * s + (j >> 2)) + 4 * g + (j & 3)];
        } else {
            v = w0[co * 32 + 8 * g + j];
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
        out[(sl * 3 + q) * 256 + g * 64 + co] = u;
    }
}

int g_num_cus = 0;

}  // namespace

extern "C" int64_t cwfa_subnet_layer_split_packed_bytes(void) { return (int64_t)NSL * WSL; }

extern "C" int cwfa_subnet_layer_split_pack_f32(const float* w3, const float* w1, void* packed, void* stream) {
    CWFA_REQUIRE(w3 && w1 && packed, CWFA_E_INVAL, "cwfa_subnet_layer_split_pack_f32: null pointer");
    CWFA_REQUIRE(cwfa_aligned16(packed), CWFA_E_ALIGN, "cwfa_subnet_layer_split_pack_f32: packed image must be 16-byte aligned");
    hipLaunchKernelGGL(split_layer_pack_kernel, dim3((NSL * 256 + 255) / 256), dim3(256), 0, (hipStream_t)stream, w3, w1,
                       reinterpret_cast<uint4*>(packed), 18, 64);
    CWFA_LAUNCH_CHECK("cwfa_subnet_layer_split_pack_f32");
    return CWFA_OK;
}

extern "C" int64_t cwfa_subnet_layer_first_packed_bytes(void) { return (int64_t)12 * WSL; }

extern "C" int cwfa_subnet_layer_first_pack_f32(const float* w3c, const float* w1, const float* w0c, int short_form, void* packed, void* stream) {
    CWFA_REQUIRE(w3c && w1 && packed, CWFA_E_INVAL, "cwfa_subnet_layer_first_pack_f32: null pointer");
    CWFA_REQUIRE(cwfa_aligned16(packed), CWFA_E_ALIGN, "cwfa_subnet_layer_first_pack_f32: packed image must be 16-byte aligned");
    CWFA_REQUIRE(!short_form || w0c, CWFA_E_INVAL, "cwfa_subnet_layer_first_pack_f32: the short form is a form of the fused first map (w0c)");
    hipLaunchKernelGGL(split_layer_pack_kernel, dim3((12 * 256 + 255) / 256), dim3(256), 0, (hipStream_t)stream, w3c, w1,
                       reinterpret_cast<uint4*>(packed), short_form ? 5 : 9, 32, w0c);
    CWFA_LAUNCH_CHECK("cwfa_subnet_layer_first_pack_f32");
    return CWFA_OK;
}

static int layer_launch(const float* x, const void* packed, const float* b3, const float* b1, float* y, int B, int H, int W, int64_t x_bs,
                        int64_t y_bs, int layout, int nprob, int spp, void* stream, const float* u = nullptr, int64_t u_bs = 0, int u_ch = 0,
                        float* hid = nullptr, int64_t hid_bs = 0);

extern "C" int cwfa_subnet_layer_split_tape_f32(const float* x, const void* packed, const float* b3, const float* b1, float* y, float* hid, int B,
                                                int H, int W, int64_t x_bs, int64_t y_bs, int64_t hid_bs, void* stream) {
    CWFA_REQUIRE(hid || B == 0 || H == 0 || W == 0, CWFA_E_INVAL, "cwfa_subnet_layer_split_tape_f32: null pointer");
    CWFA_REQUIRE(hid != x && hid != y, CWFA_E_INVAL, "cwfa_subnet_layer_split_tape_f32: the hidden map needs its own buffer");
    return layer_launch(x, packed, b3, b1, y, B, H, W, x_bs, y_bs, 0, 1, B > 0 ? B : 1, stream, nullptr, 0, 0, hid, hid_bs);
}

extern "C" int cwfa_subnet_layer_split_f32(const float* x, const void* packed, const float* b3, const float* b1, float* y, int B,
                                           int H, int W, int64_t x_bs, int64_t y_bs, int layout, void* stream) {
    return layer_launch(x, packed, b3, b1, y, B, H, W, x_bs, y_bs, layout, 1, B > 0 ? B : 1, stream);
}

extern "C" int cwfa_subnet_layer_first_f32(const float* u, const float* x, const void* packed, const float* b3, const float* b1, float* y,
                                           int B, int u_ch, int H, int W, int64_t u_bs, int64_t x_bs, int64_t y_bs, int layout, void* stream) {
    CWFA_REQUIRE(u, CWFA_E_INVAL, "cwfa_subnet_layer_first_f32: null pointer");
    CWFA_REQUIRE(u_ch >= 1 && u_ch <= 32, CWFA_E_SHAPE, "cwfa_subnet_layer_first_f32: 1 <= u_ch <= 32 (got %d)", u_ch);
    CWFA_REQUIRE((int64_t)32 * H * W * 4 < (1ll << 31), CWFA_E_SHAPE, "cwfa_subnet_layer_first_f32: image too large for 32-bit offsets");
    return layer_launch(x, packed, b3, b1, y, B, H, W, x_bs, y_bs, layout, 1, B > 0 ? B : 1, stream, u, u_bs, u_ch);
}

static int layer_launch(const float* x, const void* packed, const float* b3, const float* b1, float* y, int B, int H, int W, int64_t x_bs,
                        int64_t y_bs, int layout, int nprob, int spp, void* stream, const float* u, int64_t u_bs, int u_ch, float* hid, int64_t hid_bs) {
    CWFA_REQUIRE(layout >= 0 && layout <= 7, CWFA_E_INVAL, "cwfa_subnet_layer_split_f32: layout %d not in 0..7", layout);
    const bool shortf = (layout & 4) != 0;                  // (first-layer form only) the packed image is the SHORT one
    layout &= 3;
    CWFA_REQUIRE(!shortf || (u && !x && u_ch >= 1 && u_ch <= 16), CWFA_E_INVAL,
                 "cwfa_subnet_layer_first_f32: the short form needs x == NULL and u_ch <= 16 (got %d)", u_ch);
    CWFA_REQUIRE(!(layout & 1) || cwfa_aligned16(x), CWFA_E_ALIGN, "cwfa_subnet_layer_split_f32: blocked input must be 16-byte aligned");
    CWFA_REQUIRE(!(layout & 2) || cwfa_aligned16(y), CWFA_E_ALIGN, "cwfa_subnet_layer_split_f32: blocked output must be 16-byte aligned");
    CWFA_REQUIRE(!(layout & 1) || (x_bs & 3) == 0, CWFA_E_ALIGN, "cwfa_subnet_layer_split_f32: blocked input batch stride must be a multiple of 4");
    CWFA_REQUIRE(!(layout & 2) || (y_bs & 3) == 0, CWFA_E_ALIGN, "cwfa_subnet_layer_split_f32: blocked output batch stride must be a multiple of 4");
    CWFA_REQUIRE(B >= 0 && H >= 0 && W >= 0, CWFA_E_INVAL, "cwfa_subnet_layer_split_f32: bad size");
    if (B == 0 || H == 0 || W == 0) return CWFA_OK;
    const bool fused_x = u != nullptr && x == nullptr;      // first-layer form with the first map as a third k step (packed with w0c)
    CWFA_REQUIRE((x || fused_x) && packed && b3 && b1 && y, CWFA_E_INVAL, "cwfa_subnet_layer_split_f32: null pointer");
    CWFA_REQUIRE(x != y && u != y, CWFA_E_INVAL, "cwfa_subnet_layer_split_f32: in-place not supported (3x3 halo)");
    if (fused_x) layout &= 2;                               // (no x: its layout bit means nothing)
    CWFA_REQUIRE(cwfa_aligned16(packed), CWFA_E_ALIGN, "cwfa_subnet_layer_split_f32: packed image must be 16-byte aligned");
    CWFA_REQUIRE((int64_t)64 * H * W * 4 < (1ll << 31), CWFA_E_SHAPE, "cwfa_subnet_layer_split_f32: image too large for 32-bit offsets");
    LParams p{};
    p.x = x; p.y = y; p.wp = packed; p.b3 = b3; p.b1 = b1;
    p.B = B; p.H = H; p.W = W; p.x_bs = x_bs; p.y_bs = y_bs;
    p.in_blocked = layout & 1;
    p.out_blocked = (layout >> 1) & 1;
    p.nprob = nprob;
    p.spp = spp;
    p.u = u; p.u_bs = u_bs; p.u_ch = u_ch;
    p.hid = hid; p.hid_bs = hid_bs;
    p.xcd_map = g_cwfa_split_xcd_map;
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
    typedef void (*kern_t)(LParams);
    static const kern_t kerns[2][2][4] = {
        {{&split_layer_kernel<false, false, false>, &split_layer_kernel<false, true, false>, &split_layer_kernel<false, false, true>,
          &split_layer_kernel<false, true, true>},
         {&split_layer_kernel<true, false, false>, &split_layer_kernel<true, true, false>, &split_layer_kernel<true, false, true>,
          &split_layer_kernel<true, true, true>}},
        {{&split_layer_kernel<false, false, false, 1>, &split_layer_kernel<false, true, false, 1>, &split_layer_kernel<false, false, true, 1>,
          &split_layer_kernel<false, true, true, 1>},
         {&split_layer_kernel<true, false, false, 1>, &split_layer_kernel<true, true, false, 1>, &split_layer_kernel<true, false, true, 1>,
          &split_layer_kernel<true, true, true, 1>}}};
    static const kern_t tape_kerns[2] = {&split_layer_kernel<false, false, false, 2, true>, &split_layer_kernel<true, false, false, 2, true>};
    static const kern_t xf_kerns[2][2] = {{&split_layer_kernel<false, false, false, 1, false, true>, &split_layer_kernel<false, false, true, 1, false, true>},
                                          {&split_layer_kernel<true, false, false, 1, false, true>, &split_layer_kernel<true, false, true, 1, false, true>}};
    static const kern_t xs_kerns[2][2] = {{&split_layer_kernel<false, false, false, 1, false, true, true>, &split_layer_kernel<false, false, true, 1, false, true, true>},
                                          {&split_layer_kernel<true, false, false, 1, false, true, true>, &split_layer_kernel<true, false, true, 1, false, true, true>}};
    const int first = shortf ? 4 : fused_x ? 3 : u != nullptr ? 1 : hid != nullptr ? 2 : 0;
    kern_t kern = first == 4 ? xs_kerns[six][layout >> 1] : first == 3 ? xf_kerns[six][layout >> 1] : first == 2 ? tape_kerns[six] : kerns[first][six][layout];
    static bool attr_set_all[5][2][4] = {};
    bool& attr_done = attr_set_all[first][six][layout];
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        if (e != hipSuccess) {
            cwfa_set_error("cwfa_subnet_layer_split_f32: hipFuncSetAttribute(%d bytes LDS): %s", LDS_BYTES, hipGetErrorString(e));
            return CWFA_E_HIP;
        }
        attr_done = true;
    }
    const int grid = (int)(ntiles < g_num_cus ? ntiles : g_num_cus);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), LDS_BYTES, (hipStream_t)stream, p);
    CWFA_LAUNCH_CHECK("cwfa_subnet_layer_split_f32");
    return CWFA_OK;
}
