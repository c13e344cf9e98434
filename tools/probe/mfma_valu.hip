// Probe: does VALU work co-execute with v_mfma_f32_32x32x2_f32 on gfx950?  Each wave runs REPS x (one MFMA + NV VALU ops).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/probe/mfma_valu tools/probe/mfma_valu.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NV, int KIND>   // KIND 0: v_add_f32, 1: v_add_u32, 2: v_mfma only via other acc (two chains)
__global__ __launch_bounds__(256) void probe(float* out, int reps, float a, float b) {
    f32x16 acc = {0};
    __shared__ float lds[4096];
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    f32x4 q[2] = {{0}, {0}};
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 q2[4] = {{0}, {0}, {0}, {0}};
    int sv = 0;
    const unsigned ldsa = (unsigned)(size_t)lds + (threadIdx.x & 63) * (KIND == 3 ? 16 : KIND == 6 ? 8 : 4);
    lds[threadIdx.x] = 1.f;
    __syncthreads();
    float v[8];
    int iv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { v[i] = threadIdx.x * 0.5f + i; iv[i] = threadIdx.x + i; }
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                if (KIND == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[i & 7]) : "v"(a));
                if (KIND == 1) asm volatile("v_add_u32 %0, %0, %1" : "+v"(iv[i & 7]) : "v"(iv[(i + 1) & 7]));
                if (KIND == 2) asm volatile("ds_read_b32 %0, %1" : "=v"(v[i & 7]) : "v"(ldsa) : "memory");
                if (KIND == 3) asm volatile("ds_read_b128 %0, %1" : "=v"(q[i & 1]) : "v"(ldsa) : "memory");
                if (KIND == 6) asm volatile("ds_read_b64 %0, %1" : "=v"(q2[i & 3]) : "v"(ldsa) : "memory");
                if (KIND == 5) asm volatile("ds_write_b32 %1, %0" : : "v"(v[i & 7]), "v"(ldsa) : "memory");
            }
        }
        if (KIND == 2 || KIND == 3 || KIND == 6) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    float s = sv + q[0][0] + q[1][1] + q2[0][0] + q2[1][1] + q2[2][0] + q2[3][1];
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i] + iv[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NV, int KIND>
float run(int blocks, int threads, int reps, float* out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    probe<NV, KIND><<<blocks, threads>>>(out, reps, 1.f, 2.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<NV, KIND><<<blocks, threads>>>(out, reps, 1.f, 2.f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    float* out; hipMalloc(&out, 256 * 8 * 1024 * 4);
    const int reps = 2000;
    for (int wps = 1; wps <= 2; ++wps) {       // waves per SIMD: blocks of 256 threads = 1 wave/SIMD; wps blocks per CU
        const int blocks = 256 * wps;
        const double mf = (double)blocks * 4 * reps * 8;   // MFMAs
#define ROW(NV, K) { float ms = run<NV, K>(blocks, 256, reps, out); printf("wps %d kind %d nv %2d  %.3f ms  cyc/MFMA/SIMD@2.4GHz %.1f\n", wps, K, NV, ms, ms * 1e-3 * 2.4e9 / (mf / 1024)); }
        ROW(0, 0) ROW(8, 0) ROW(16, 0)
        ROW(8, 1) ROW(16, 1)
        ROW(1, 2) ROW(2, 2) ROW(4, 2) ROW(8, 2)
        ROW(1, 3) ROW(2, 3) ROW(4, 3)
        ROW(1, 6) ROW(2, 6)
        ROW(1, 5) ROW(2, 5) ROW(4, 5)
    }
    return 0;
}
