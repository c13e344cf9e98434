// Probe 2: one wave per SIMD issues only fp32 MFMAs, its SIMD partner (wave + 4 of a 512-thread block) issues only VALU
// (or LDS-write / global-load) work.  How much does the partner slow the MFMA wave, and how fast does the partner run?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int KIND>   // 0: independent v_add_f32 (8 chains)  1: one dependent chain  2: v_add_u32  3: ds_write_b32  4: global_load_dword
__global__ __launch_bounds__(512) void probe(float* out, long long* ticks, int mfmas, int pinstr, float a, const float* g) {
    __shared__ float lds[8192];
    const int wave = threadIdx.x >> 6;
    f32x16 acc = {0};
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x + i;
    int iv = threadIdx.x;
    const unsigned la = (unsigned)(size_t)lds + threadIdx.x * 4;
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    if (wave < 4) {
        for (int r = 0; r < mfmas / 8; ++r) {
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, a, acc, 0, 0, 0);
        }
    } else {
        for (int r = 0; r < pinstr / 8; ++r) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (KIND == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[u]) : "v"(a));
                if (KIND == 1) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[0]) : "v"(a));
                if (KIND == 2) asm volatile("v_add_u32 %0, %0, %1" : "+v"(iv) : "v"(iv));
                if (KIND == 3) asm volatile("ds_write_b32 %0, %1" : : "v"(la), "v"(v[u]) : "memory");
                if (KIND == 4) asm volatile("global_load_dword %0, %1, off" : "=v"(v[u]) : "v"(g + threadIdx.x + u * 512) : "memory");
            }
            if (KIND == 4) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) ticks[wave] = t1 - t0;
    float s = iv;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND>
void run(int mfmas, int pinstr, float* out, long long* dt, const float* g) {
    probe<KIND><<<256, 512>>>(out, dt, mfmas, pinstr, 1.f, g);
    (void)hipDeviceSynchronize();
    probe<KIND><<<256, 512>>>(out, dt, mfmas, pinstr, 1.f, g);
    (void)hipDeviceSynchronize();
    long long h[8];
    (void)hipMemcpy(h, dt, sizeof(h), hipMemcpyDeviceToHost);
    printf("kind %d  mfmas %5d partner-instr %6d :  mfma-wave %8lld cyc (%.1f/mfma)   partner %8lld cyc (%.1f/instr)\n", KIND, mfmas,
           pinstr, h[0], (double)h[0] / mfmas, h[4], pinstr ? (double)h[4] / pinstr : 0.0);
}

int main() {
    float *out, *g; long long* dt;
    (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&dt, 64); (void)hipMalloc(&g, 1 << 20);
    run<0>(4096, 0, out, dt, g);
    run<0>(4096, 4096, out, dt, g);  run<0>(4096, 16384, out, dt, g);  run<0>(4096, 65536, out, dt, g);
    run<1>(4096, 4096, out, dt, g);  run<1>(4096, 16384, out, dt, g);
    run<2>(4096, 16384, out, dt, g);
    run<3>(4096, 4096, out, dt, g);  run<3>(4096, 16384, out, dt, g);
    run<4>(4096, 4096, out, dt, g);
    run<0>(0, 16384, out, dt, g);    run<1>(0, 16384, out, dt, g);
    return 0;
}
