// Measured peaks on the box: HBM stream copy / read / write, and the fp32 MFMA issue loop.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void k_copy(const f32x4* __restrict__ a, f32x4* __restrict__ b, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) b[i] = a[i];
}
__global__ __launch_bounds__(256) void k_read(const f32x4* __restrict__ a, float* __restrict__ out, long n) {
    f32x4 s = {0, 0, 0, 0};
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) s += a[i];
    if (s[0] + s[1] + s[2] + s[3] == 1234.5f) out[0] = 1.f;
}
__global__ __launch_bounds__(256) void k_write(f32x4* __restrict__ b, long n) {
    const f32x4 v = {1.f, 2.f, 3.f, 4.f};
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) b[i] = v;
}
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k_mfma(float* out, int reps, float a) {
    f32x16 acc[4] = {};
    for (int r = 0; r < reps; ++r)
#pragma unroll
        for (int u = 0; u < 8; ++u) acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, a, acc[u & 3], 0, 0, 0);
    float s = 0;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 16; ++j) s += acc[i][j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
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
    const long bytes = 2L << 30, n = bytes / 16;
    f32x4 *a, *b; float* out;
    (void)hipMalloc(&a, bytes); (void)hipMalloc(&b, bytes); (void)hipMalloc(&out, 256 * 512 * 4);
    (void)hipMemset(a, 0, bytes);
    const float tc = timed([&] { k_copy<<<8192, 256>>>(a, b, n); }, 5);
    const float tr = timed([&] { k_read<<<8192, 256>>>(a, out, n); }, 5);
    const float tw = timed([&] { k_write<<<8192, 256>>>(b, n); }, 5);
    printf("HBM stream (2 GiB buffers, 16 B/lane): copy %.0f GB/s (read+write), read %.0f GB/s, write %.0f GB/s\n",
           2.0 * bytes / tc / 1e6, bytes / tr / 1e6, bytes / tw / 1e6);
    const int reps = 4000;
    const float t1 = timed([&] { k_mfma<4><<<256, 256>>>(out, reps, 1.f); }, 3);
    const float t2 = timed([&] { k_mfma<8><<<256, 512>>>(out, reps, 1.f); }, 3);
    const double fl1 = 256.0 * 4 * reps * 8 * 4096, fl2 = 2 * fl1;
    printf("fp32 MFMA loop (v_mfma_f32_32x32x2_f32, 4 independent accumulators): 1 wave/SIMD %.1f TFLOP/s, 2 waves/SIMD %.1f TFLOP/s\n",
           fl1 / t1 / 1e9, fl2 / t2 / 1e9);
    return 0;
}
