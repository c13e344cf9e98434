// Which bf16 MFMA shape does the chip run faster under a split-bf16 (six products) load on RANDOM data, operands
// re-read from LDS every k-step?  Same 64 x 64 output tile per wave, same LDS bytes per FLOP, 2 waves per SIMD.
//   32x32x16: 2 x 2 accumulator tiles, k-step 16: A 2 x 3 pieces, B 2 x 3 pieces (ds_read_b128), 24 MFMAs
//   16x16x32: 4 x 4 accumulator tiles, k-step 32: A 4 x 3 pieces, B 4 x 3 pieces, 96 MFMAs
// (MI355X_MICROARCH.md, DVFS give-back item 7: the clock the chip holds depends on the shape.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(512, 1) void k32(const uint4* __restrict__ src, float* out, int reps) {
    __shared__ uint4 lds[4096];                                  // 64 KB of random bf16 bits
    for (int i = threadIdx.x; i < 4096; i += 512) lds[i] = src[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bf16x8* base = reinterpret_cast<const bf16x8*>(lds) + lane + wave * 64;
    f32x16 acc[2][2] = {};
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int r = 0; r < reps; ++r) {
        bf16x8 A[2][3], B[2][3];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                A[m][q] = base[((r * 12 + m * 3 + q) * 512) & 3071];
                B[m][q] = base[((r * 12 + 6 + m * 3 + q) * 512 + 64) & 3071];
            }
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                f32x16 c = acc[m][n];
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[m][2], B[n][0], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[m][1], B[n][1], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[m][0], B[n][2], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[m][1], B[n][0], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[m][0], B[n][1], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[m][0], B[n][0], c, 0, 0, 0);
                acc[m][n] = c;
            }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int m = 0; m < 2; ++m)
        for (int n = 0; n < 2; ++n)
            for (int j = 0; j < 16; ++j) s += acc[m][n][j];
    out[blockIdx.x * 512 + threadIdx.x] = s + (float)(t1 - t0) * 1e-30f;
}

template <int NT, bool BAR, bool PAIR>
__global__ __launch_bounds__(NT, 1) void k16v(const uint4* __restrict__ src, float* out, int reps) {
    __shared__ uint4 lds[4096];
    for (int i = threadIdx.x; i < 4096; i += NT) lds[i] = src[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bf16x8* base = reinterpret_cast<const bf16x8*>(lds) + lane + wave * 64;
    f32x4 acc[4][4] = {};
    for (int r = 0; r < reps; ++r) {
        bf16x8 A[4][3], B[4][3];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                A[m][q] = base[((r * 24 + m * 3 + q) * 512) & 3071];
                B[m][q] = base[((r * 24 + 12 + m * 3 + q) * 512 + 64) & 3071];
            }
        constexpr int qa[6] = {2, 1, 0, 1, 0, 0}, qb[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 4; n += (PAIR ? 2 : 1))
#pragma unroll
                for (int pr = 0; pr < 6; ++pr) {
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[m][qa[pr]], B[n][qb[pr]], acc[m][n], 0, 0, 0);
                    if (PAIR) acc[m][n + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[m][qa[pr]], B[n + 1][qb[pr]], acc[m][n + 1], 0, 0, 0);
                }
        if (BAR) __syncthreads();
    }
    float s = 0;
    for (int m = 0; m < 4; ++m)
        for (int n = 0; n < 4; ++n)
            for (int j = 0; j < 4; ++j) s += acc[m][n][j];
    out[blockIdx.x * NT + threadIdx.x] = s;
}

__global__ __launch_bounds__(512, 1) void k16(const uint4* __restrict__ src, float* out, int reps) {
    __shared__ uint4 lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 512) lds[i] = src[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bf16x8* base = reinterpret_cast<const bf16x8*>(lds) + lane + wave * 64;
    f32x4 acc[4][4] = {};
    for (int r = 0; r < reps; ++r) {
        bf16x8 A[4][3], B[4][3];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                A[m][q] = base[((r * 24 + m * 3 + q) * 512) & 3071];
                B[m][q] = base[((r * 24 + 12 + m * 3 + q) * 512 + 64) & 3071];
            }
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                f32x4 c = acc[m][n];
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[m][2], B[n][0], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[m][1], B[n][1], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[m][0], B[n][2], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[m][1], B[n][0], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[m][0], B[n][1], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[m][0], B[n][0], c, 0, 0, 0);
                acc[m][n] = c;
            }
    }
    float s = 0;
    for (int m = 0; m < 4; ++m)
        for (int n = 0; n < 4; ++n)
            for (int j = 0; j < 4; ++j) s += acc[m][n][j];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

// the same 16x16x32 loop with the six products of a tile NOT back to back on one accumulator (16-cycle issue, dependent
// latency may exceed it): products outer, tiles inner
__global__ __launch_bounds__(512, 1) void k16i(const uint4* __restrict__ src, float* out, int reps) {
    __shared__ uint4 lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 512) lds[i] = src[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bf16x8* base = reinterpret_cast<const bf16x8*>(lds) + lane + wave * 64;
    f32x4 acc[4][4] = {};
    for (int r = 0; r < reps; ++r) {
        bf16x8 A[4][3], B[4][3];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                A[m][q] = base[((r * 24 + m * 3 + q) * 512) & 3071];
                B[m][q] = base[((r * 24 + 12 + m * 3 + q) * 512 + 64) & 3071];
            }
        constexpr int qa[6] = {2, 1, 0, 1, 0, 0}, qb[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
        for (int pr = 0; pr < 6; ++pr)
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[m][qa[pr]], B[n][qb[pr]], acc[m][n], 0, 0, 0);
    }
    float s = 0;
    for (int m = 0; m < 4; ++m)
        for (int n = 0; n < 4; ++n)
            for (int j = 0; j < 4; ++j) s += acc[m][n][j];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <class F>
float timed(F f, int reps) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    f(); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) f();
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}

int main() {
    uint4* src; float* out;
    (void)hipMalloc(&src, 65536); (void)hipMalloc(&out, 256 * 512 * 4);
    unsigned short* h = (unsigned short*)malloc(65536);
    srand(1);
    for (int i = 0; i < 32768; ++i) {                   // random bf16 in +-[0.5, 2): sign, exponent 126..127, 7 mantissa bits
        h[i] = (unsigned short)(((rand() & 1) << 15) | ((126 + (rand() & 1)) << 7) | (rand() & 127));
    }
    (void)hipMemcpy(src, h, 65536, hipMemcpyHostToDevice);
    const int reps32 = 4000, reps16 = 2000;             // a k32 rep = 24 MFMAs of 32x32x16; a k16 rep = 96 of 16x16x32 = 2x the FLOPs
    for (int round = 0; round < 3; ++round) {
        const float t32 = timed([&] { k32<<<256, 512>>>(src, out, reps32); }, 10);
        const float t16 = timed([&] { k16<<<256, 512>>>(src, out, reps16); }, 10);
        const float t16i = timed([&] { k16i<<<256, 512>>>(src, out, reps16); }, 10);
        const double fl = 256.0 * 8 * reps32 * 24 * 2.0 * 32 * 32 * 16;
        printf("round %d: 32x32x16 %.3f ms = %.0f TF/s bf16 (%.0f TF/s fp32-equivalent) | 16x16x32 chained %.3f ms = %.0f TF/s (%.0f) | 16x16x32 interleaved %.3f ms = %.0f TF/s (%.0f)\n",
               round, t32, fl / t32 / 1e9, fl / t32 / 6e9, t16, fl / t16 / 1e9, fl / t16 / 6e9, t16i, fl / t16i / 1e9, fl / t16i / 6e9);
    }
    {
        const double fl2 = 256.0 * 8 * reps16 * 96 * 2.0 * 16 * 16 * 32;
        auto rep = [&](const char* name, float t, double fl) { printf("%-44s %.3f ms = %.0f TF/s bf16\n", name, t, fl / t / 1e9); };
        rep("16x16x32 2 waves/SIMD chained", timed([&] { k16v<512, false, false><<<256, 512>>>(src, out, reps16); }, 10), fl2);
        rep("16x16x32 2 waves/SIMD pair-interleaved", timed([&] { k16v<512, false, true><<<256, 512>>>(src, out, reps16); }, 10), fl2);
        rep("16x16x32 2 waves/SIMD chained + barrier/96", timed([&] { k16v<512, true, false><<<256, 512>>>(src, out, reps16); }, 10), fl2);
        rep("16x16x32 2 waves/SIMD pair + barrier/96", timed([&] { k16v<512, true, true><<<256, 512>>>(src, out, reps16); }, 10), fl2);
        rep("16x16x32 1 wave/SIMD chained", timed([&] { k16v<256, false, false><<<256, 256>>>(src, out, reps16); }, 10), fl2 / 2);
        rep("16x16x32 1 wave/SIMD pair-interleaved", timed([&] { k16v<256, false, true><<<256, 256>>>(src, out, reps16); }, 10), fl2 / 2);
        rep("16x16x32 1 wave/SIMD pair + barrier/96", timed([&] { k16v<256, true, true><<<256, 256>>>(src, out, reps16); }, 10), fl2 / 2);
    }
    return 0;
}
