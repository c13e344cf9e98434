#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
struct __attribute__((packed, aligned(2))) U4 { u32x4 v; };
__global__ void k(unsigned* out, int off) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = (unsigned short)i;
    __syncthreads();
    const char* base = reinterpret_cast<const char*>(lds) + threadIdx.x * 16 + off;
    u32x4 v = reinterpret_cast<const U4*>(base)->v;
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = v[j];
}
__global__ void t(unsigned* out, int off, int reps) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[8192];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = (unsigned short)i;
    __syncthreads();
    u32x4 acc = {0, 0, 0, 0};
    const unsigned base = (unsigned)(size_t)lds + (threadIdx.x & 63) * 16 + off;
    for (int r = 0; r < reps; ++r) {
        u32x4 v0, v1, v2, v3;
        asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:1024\n\tds_read_b128 %2, %4 offset:2048\n\tds_read_b128 %3, %4 offset:3072\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3) : "v"(base) : "memory");
        acc += v0 + v1 + v2 + v3;
    }
    out[threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}
int main() {
    unsigned* d; hipMalloc(&d, 1 << 20);
    unsigned h[256 * 4];
    for (int off : {0, 2, 4, 8, 6}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, off);
        hipMemcpy(h, d, sizeof(unsigned) * 256, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int t_ = 0; t_ < 64; ++t_) for (int j = 0; j < 4; ++j) {
            unsigned e0 = (t_ * 8 + off / 2 + 2 * j) & 0xffff, e1 = (e0 + 1) & 0xffff;
            if (h[t_ * 4 + j] != (e0 | (e1 << 16))) ++bad;
        }
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipLaunchKernelGGL(t, dim3(256), dim3(512), 0, 0, d, off, 1000);
        hipEventRecord(a); hipLaunchKernelGGL(t, dim3(256), dim3(512), 0, 0, d, off, 20000); hipEventRecord(b); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("offset %d: wrong words %d, %.3f ms for 80000 b128 reads per wave\n", off, bad, ms);
    }
    return 0;
}
