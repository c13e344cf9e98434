// Calibration for rocprofv3 FETCH_SIZE on gfx950 (MI355X_MICROARCH.md, HBM section: only 16-B/lane streams are
// calibrated there): stream a buffer larger than the Infinity Cache once with 4-, 8- and 16-byte-per-lane coalesced
// loads; the kernel names carry the width, the byte count is printed.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int W>
__global__ __launch_bounds__(256) void calib_read(const float* __restrict__ x, float* __restrict__ out, long n) {
    typedef float vec __attribute__((ext_vector_type(W)));
    const vec* p = reinterpret_cast<const vec*>(x);
    const long nv = n / W, stride = (long)gridDim.x * blockDim.x;
    float s = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += stride) {
        const vec v = __builtin_nontemporal_load(p + i);
        for (int j = 0; j < W; ++j) s += v[j];
    }
    if (s == 12345.678f) out[0] = s;
}
int main() {
    const long n = 1L << 28;    // 1 GiB of floats
    float *x, *out;
    (void)hipMalloc(&x, n * 4); (void)hipMalloc(&out, 4);
    (void)hipMemset(x, 0, n * 4);
    for (int rep = 0; rep < 2; ++rep) {
        calib_read<1><<<8192, 256>>>(x, out, n);
        calib_read<2><<<8192, 256>>>(x, out, n);
        calib_read<4><<<8192, 256>>>(x, out, n);
    }
    (void)hipDeviceSynchronize();
    printf("bytes_per_launch %ld\n", n * 4);
    return 0;
}
