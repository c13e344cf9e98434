// Probe 3: the Winograd k-step (ds_read_b64 A pair + ds_read_b32 B, two MFMAs, operands 2 steps ahead) with one wave
// per SIMD (256-thread block) and with two (512 threads): cycles per MFMA per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int NT, int DEPTH>
__global__ __launch_bounds__(NT) void probe(float* out, long long* ticks, int reps) {
    __shared__ __attribute__((aligned(16))) float lds[16384];
    for (int i = threadIdx.x; i < 16384; i += NT) lds[i] = 1.f / (1 + i);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const float* ua = lds + 2 * (lane & 31) + (lane >> 5) * 64;
    const float* vb = lds + 8192 + (lane & 31) + (lane >> 5) * 320;
    f32x16 acc[2][4] = {};
    float bq[DEPTH + 1];
    f32x2 aq[DEPTH + 1];
    const long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int s = 0; s < DEPTH; ++s) { aq[s] = *(const f32x2*)(ua + s * 128); bq[s] = vb[s * 64]; }
#pragma unroll
        for (int s = 0; s < 48; ++s) {
            if (s + DEPTH < 48) {
                aq[(s + DEPTH) % (DEPTH + 1)] = *(const f32x2*)(ua + (s + DEPTH) * 128);
                bq[(s + DEPTH) % (DEPTH + 1)] = vb[((s + DEPTH) % 24) * 64 + ((s + DEPTH) / 24) * 2048];
            }
            acc[0][(s / 4) % 4] = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[s % (DEPTH + 1)][0], bq[s % (DEPTH + 1)], acc[0][(s / 4) % 4], 0, 0, 0);
            acc[1][(s / 4) % 4] = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[s % (DEPTH + 1)][1], bq[s % (DEPTH + 1)], acc[1][(s / 4) % 4], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    if (lane == 0 && blockIdx.x == 0) ticks[threadIdx.x >> 6] = t1 - t0;
    float sum = 0;
    for (int m = 0; m < 2; ++m)
        for (int x = 0; x < 4; ++x)
            for (int i = 0; i < 16; ++i) sum += acc[m][x][i];
    out[blockIdx.x * NT + threadIdx.x] = sum;
}

template <int NT, int DEPTH>
void run(float* out, long long* dt) {
    const int reps = 200;
    probe<NT, DEPTH><<<256, NT>>>(out, dt, reps);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    probe<NT, DEPTH><<<256, NT>>>(out, dt, reps);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    long long h[8];
    (void)hipMemcpy(h, dt, sizeof(h), hipMemcpyDeviceToHost);
    const double per = (double)h[0] / (reps * 96.0) / (NT / 256);
    printf("threads %d depth %d: wave0 %lld ticks (wave4 %lld), %.1f ticks per MFMA per SIMD; wall %.3f ms = %.1f ns per MFMA per SIMD\n", NT, DEPTH, h[0], h[4], per, ms, ms * 1e6 / (reps * 96.0 * (NT / 256)));
}

int main() {
    float* out; long long* dt;
    (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&dt, 64);
    run<256, 2>(out, dt); run<512, 2>(out, dt); run<256, 4>(out, dt); run<512, 4>(out, dt);
    return 0;
}
