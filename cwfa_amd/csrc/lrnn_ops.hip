// Small HBM-bound kernels of the LRNN (UNet + ConvNeXt + GlobalAttention): normalisation statistics and applies,
// adaptive max-pool, attention/combine.  The convolutions themselves are in conv2d.hip.
#include "common.h"

// ------------------------------------------------------------------------------------------------ BatchNorm statistics
// grid (splits, C, B): each block sums a contiguous slice of one (sample, channel) plane and adds (sum, sumsq) in double.
// Per thread the partial sums of <= 64 elements stay in fp32 pairs feeding doubles (the products are exact in double).
typedef float cs_f4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void channel_stats_kernel(const float* __restrict__ x, double* __restrict__ stats, int64_t HW,
                                                            int64_t x_bs, int vec_ok) {
    __shared__ double red[16];
    const int c = blockIdx.y;
    const float* p = x + (int64_t)blockIdx.z * x_bs + (int64_t)c * HW;
    const int64_t per = (((HW + gridDim.x - 1) / gridDim.x) + 3) & ~(int64_t)3;      // slices start on 16-byte boundaries
    const int64_t lo = (int64_t)blockIdx.x * per < HW ? (int64_t)blockIdx.x * per : HW, hi = lo + per < HW ? lo + per : HW;
    double s = 0.0, q = 0.0;
    if (vec_ok && (lo & 3) == 0) {
        const int64_t n4 = (hi - lo) >> 2;
        const cs_f4* p4 = reinterpret_cast<const cs_f4*>(p + lo);
        int64_t i = threadIdx.x;
        for (; i + 3 * (int64_t)blockDim.x < n4; i += 4 * (int64_t)blockDim.x) {        // four 16-byte loads in flight per thread
            cs_f4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = p4[i + u * (int64_t)blockDim.x];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                s += ((double)v[u][0] + (double)v[u][1]) + ((double)v[u][2] + (double)v[u][3]);
                q += ((double)v[u][0] * v[u][0] + (double)v[u][1] * v[u][1]) + ((double)v[u][2] * v[u][2] + (double)v[u][3] * v[u][3]);
            }
        }
        for (; i < n4; i += blockDim.x) {
            const cs_f4 v = p4[i];
            s += ((double)v[0] + (double)v[1]) + ((double)v[2] + (double)v[3]);
            q += ((double)v[0] * v[0] + (double)v[1] * v[1]) + ((double)v[2] * v[2] + (double)v[3] * v[3]);
        }
        for (int64_t i = lo + (n4 << 2) + threadIdx.x; i < hi; i += blockDim.x) {
            const float v = p[i];
            s += v;
            q += (double)v * v;
        }
    } else {
        for (int64_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
            const float v = p[i];
            s += v;
            q += (double)v * v;
        }
    }
    s = cwfa_block_sum(s, red);
    q = cwfa_block_sum(q, red);
    if (threadIdx.x == 0) {
        atomicAdd(&stats[2 * c], s);
        atomicAdd(&stats[2 * c + 1], q);
    }
}

extern "C" int cwfa_channel_stats_f32(const float* x, double* stats, int B, int C, int64_t HW, int64_t x_bs, void* stream) {
    CWFA_REQUIRE(x && stats, CWFA_E_INVAL, "cwfa_channel_stats_f32: null pointer");
    CWFA_REQUIRE(B >= 0 && C >= 0 && HW >= 0 && C <= 65535 && B <= 65535, CWFA_E_SHAPE, "cwfa_channel_stats_f32: bad shape");
    if (B == 0 || C == 0 || HW == 0) return CWFA_OK;
    // enough blocks to fill the chip several times over (256 CUs x 8 resident blocks), but as few atomics and as long
    // streams per block as that allows: ~4096 blocks in all, >= 16 elements per thread
    int splits = (int)((HW + 256 * 16 - 1) / (256 * 16));
    const int64_t want = (4096 + (int64_t)C * B - 1) / ((int64_t)C * B);
    if (splits > want) splits = (int)want;
    if (splits < 1) splits = 1;
    if (splits > 64) splits = 64;
    const int vec_ok = (HW & 3) == 0 && (x_bs & 3) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0;
    hipLaunchKernelGGL(channel_stats_kernel, dim3(splits, C, B), dim3(256), 0, (hipStream_t)stream, x, stats, HW, x_bs, vec_ok);
    CWFA_LAUNCH_CHECK("cwfa_channel_stats_f32");
    return CWFA_OK;
}


__global__ void bn_fold_kernel(const double* __restrict__ stats, double count, const float* __restrict__ rm,
                               const float* __restrict__ rv, const float* __restrict__ w, const float* __restrict__ bsh,
                               float eps, const float* __restrict__ mask, int B, float* __restrict__ scale,
                               float* __restrict__ shift, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double sc = 1.0, sh = 0.0;
    if (stats || rm) {
        double mean, var;
        if (stats) {
            mean = stats[2 * c] / count;
            var = stats[2 * c + 1] / count - mean * mean;
            if (var < 0.0) var = 0.0;
        } else {
            mean = rm[c];
            var = rv[c];
        }
        sc = (w ? (double)w[c] : 1.0) / sqrt(var + (double)eps);
        sh = (bsh ? (double)bsh[c] : 0.0) - mean * sc;
    }
    if (mask) {
        for (int b = 0; b < B; ++b) {
            const float m = mask[b * C + c];
            scale[b * C + c] = (float)sc * m;
            shift[b * C + c] = (float)sh * m;
        }
    } else {
        scale[c] = (float)sc;
        shift[c] = (float)sh;
    }
}

// Everything that follows the statistics pass of a train-mode BatchNorm2d in ONE launch (the inference path issued a fill, the
// fold, the running-buffer update and three elementwise kernels for the dropout mask per layer -- ~40 tiny launches per volume):
// (scale, shift) as bn_fold_kernel; running_mean / running_var / num_batches_tracked as bn_running_update_kernel; the dropout
// factor from the raw uniform draw, m = (u >= p) / (1 - p) (F.dropout2d, unet.py:80,86); and the statistics buffer is zeroed
// for its next use.
__global__ void bn_finish_kernel(double* __restrict__ stats, double count, float* __restrict__ rm, float* __restrict__ rv,
                                 long long* __restrict__ nbt, float momentum, int update_running, const float* __restrict__ w,
                                 const float* __restrict__ bsh, float eps, const float* __restrict__ mask,
                                 const float* __restrict__ mask_u, float drop_p, float keep_scale, int B, float* __restrict__ scale,
                                 float* __restrict__ shift, int C, int zero_stats) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c == 0 && nbt && update_running) *nbt += 1;
    if (c >= C) return;
    double sc = 1.0, sh = 0.0;
    if (stats || rm) {
        double mean, var;
        if (stats) {
            const double s1 = stats[2 * c], s2 = stats[2 * c + 1];
            mean = s1 / count;
            var = s2 / count - mean * mean;
            if (var < 0.0) var = 0.0;
            if (update_running) {
                const double var_u = (s2 - s1 * mean) / (count > 1.0 ? count - 1.0 : 1.0);
                rm[c] = rm[c] * (1.f - momentum) + momentum * (float)mean;
                rv[c] = rv[c] * (1.f - momentum) + momentum * (float)var_u;
            }
            if (zero_stats) stats[2 * c] = stats[2 * c + 1] = 0.0;
        } else {
            mean = rm[c];
            var = rv[c];
        }
        sc = (w ? (double)w[c] : 1.0) / sqrt(var + (double)eps);
        sh = (bsh ? (double)bsh[c] : 0.0) - mean * sc;
    }
    if (mask || mask_u) {
        for (int b = 0; b < B; ++b) {
            const float m = mask ? mask[b * C + c] : (mask_u[b * C + c] >= drop_p ? 1.f : 0.f) / keep_scale;
            scale[b * C + c] = (float)sc * m;
            shift[b * C + c] = (float)sh * m;
        }
    } else {
        scale[c] = (float)sc;
        shift[c] = (float)sh;
    }
}

extern "C" int cwfa_bn_finish_f32(double* stats, double count, float* running_mean, float* running_var, long long* num_batches_tracked,
                                  float momentum, int update_running, const float* weight, const float* bias, float eps,
                                  const float* mask_bc, const float* mask_u, float drop_p, float keep_scale, int B, float* scale,
                                  float* shift, int C, int zero_stats, void* stream) {
    CWFA_REQUIRE(scale && shift, CWFA_E_INVAL, "cwfa_bn_finish_f32: null output");
    CWFA_REQUIRE(stats || (running_mean && running_var) || mask_bc || mask_u, CWFA_E_INVAL,
                 "cwfa_bn_finish_f32: neither batch nor running statistics nor a mask");
    CWFA_REQUIRE(!running_mean == !running_var, CWFA_E_INVAL, "cwfa_bn_finish_f32: running_mean and running_var go together");
    CWFA_REQUIRE(!(mask_bc && mask_u), CWFA_E_INVAL, "cwfa_bn_finish_f32: a mask OR a uniform draw");
    CWFA_REQUIRE(!(mask_bc || mask_u) || B > 0, CWFA_E_INVAL, "cwfa_bn_finish_f32: mask needs B > 0");
    CWFA_REQUIRE(!mask_u || (drop_p >= 0.f && drop_p < 1.f), CWFA_E_INVAL, "cwfa_bn_finish_f32: 0 <= drop_p < 1");
    CWFA_REQUIRE(!stats || count > 0, CWFA_E_INVAL, "cwfa_bn_finish_f32: count must be positive");
    CWFA_REQUIRE(!update_running || (stats && running_mean), CWFA_E_INVAL, "cwfa_bn_finish_f32: the running update needs statistics and buffers");
    CWFA_REQUIRE(C >= 0, CWFA_E_INVAL, "cwfa_bn_finish_f32: negative size");
    if (C == 0) return CWFA_OK;
    CWFA_REQUIRE(!mask_u || keep_scale > 0.f, CWFA_E_INVAL, "cwfa_bn_finish_f32: keep_scale = fp32(1 - p) must be positive");
    hipLaunchKernelGGL(bn_finish_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, stats, count, running_mean, running_var,
                       num_batches_tracked, momentum, update_running, weight, bias, eps, mask_bc, mask_u, drop_p, keep_scale, B, scale,
                       shift, C, zero_stats);
    CWFA_LAUNCH_CHECK("cwfa_bn_finish_f32");
    return CWFA_OK;
}

// nn.BatchNorm2d's train-mode buffer bookkeeping in one launch (torch/nn/modules/batchnorm.py via unet.py:101-107):
// running_mean <- (1-m) running_mean + m mean,  running_var <- (1-m) running_var + m var_unbiased,  num_batches_tracked += 1
__global__ void bn_running_update_kernel(const double* __restrict__ stats, double count, float momentum, float* __restrict__ rm,
                                         float* __restrict__ rv, long long* __restrict__ nbt, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c == 0 && nbt) *nbt += 1;
    if (c >= C) return;
    const double mean = stats[2 * c] / count;
    const double var_u = (stats[2 * c + 1] - stats[2 * c] * mean) / (count > 1.0 ? count - 1.0 : 1.0);
    rm[c] = rm[c] * (1.f - momentum) + momentum * (float)mean;
    rv[c] = rv[c] * (1.f - momentum) + momentum * (float)var_u;
}

extern "C" int cwfa_bn_running_update_f32(const double* stats, double count, float momentum, float* running_mean,
                                          float* running_var, long long* num_batches_tracked, int C, void* stream) {
    CWFA_REQUIRE(stats && running_mean && running_var, CWFA_E_INVAL, "cwfa_bn_running_update_f32: null pointer");
    CWFA_REQUIRE(C >= 0 && count > 0.0, CWFA_E_INVAL, "cwfa_bn_running_update_f32: bad size");
    if (C == 0) return CWFA_OK;
    hipLaunchKernelGGL(bn_running_update_kernel, dim3((C + 127) / 128), dim3(128), 0, (hipStream_t)stream, stats, count, momentum,
                       running_mean, running_var, num_batches_tracked, C);
    CWFA_LAUNCH_CHECK("cwfa_bn_running_update_f32");
    return CWFA_OK;
}

extern "C" int cwfa_bn_fold_f32(const double* stats, double count, const float* running_mean, const float* running_var,
                                const float* weight, const float* bias, float eps, const float* mask_bc, int B, float* scale,
                                float* shift, int C, void* stream) {
    CWFA_REQUIRE(scale && shift, CWFA_E_INVAL, "cwfa_bn_fold_f32: null output");
    CWFA_REQUIRE(stats || (running_mean && running_var) || mask_bc, CWFA_E_INVAL,
                 "cwfa_bn_fold_f32: neither batch nor running statistics nor a mask");
    CWFA_REQUIRE(!running_mean == !running_var, CWFA_E_INVAL, "cwfa_bn_fold_f32: running_mean and running_var go together");
    CWFA_REQUIRE(!mask_bc || B > 0, CWFA_E_INVAL, "cwfa_bn_fold_f32: mask needs B > 0");
    CWFA_REQUIRE(!stats || count > 0, CWFA_E_INVAL, "cwfa_bn_fold_f32: count must be positive");
    CWFA_REQUIRE(C >= 0, CWFA_E_INVAL, "cwfa_bn_fold_f32: negative size");
    if (C == 0) return CWFA_OK;
    hipLaunchKernelGGL(bn_fold_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, stats, count, running_mean,
                       running_var, weight, bias, eps, mask_bc, B, scale, shift, C);
    CWFA_LAUNCH_CHECK("cwfa_bn_fold_f32");
    return CWFA_OK;
}

// ------------------------------------------------------------------------------------------------ adaptive max pool
// window of output i along a dimension of size n -> m:  [floor(i*n/m), ceil((i+1)*n/m))   (ATen adaptive pooling)
__global__ __launch_bounds__(256) void maxpool_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                      float* __restrict__ full, const float* __restrict__ scale,
                                                      const float* __restrict__ shift, int C, int H, int W, int Ho, int Wo) {
    const int64_t n = (int64_t)Ho * Wo;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int bc = blockIdx.y, c = bc % C;
    const int oy = (int)(i / Wo), ox = (int)(i % Wo);
    const int y0 = (int)(((int64_t)oy * H) / Ho), y1 = (int)((((int64_t)oy + 1) * H + Ho - 1) / Ho);
    const int x0 = (int)(((int64_t)ox * W) / Wo), x1 = (int)((((int64_t)ox + 1) * W + Wo - 1) / Wo);
    const float sc = scale ? scale[c] : 1.f, sh = shift ? shift[c] : 0.f;
    const float* px = x + (int64_t)bc * H * W;
    float* pf = full ? full + (int64_t)bc * H * W : nullptr;
    float m = -INFINITY;
    for (int yy = y0; yy < y1; ++yy)
        for (int xx = x0; xx < x1; ++xx) {
            float v = px[(int64_t)yy * W + xx];
            if (scale) v = v * sc + sh;
            // overlapping adaptive windows (H % Ho != 0) rewrite the same value: benign
            if (pf) pf[(int64_t)yy * W + xx] = v;
            m = (v > m || v != v) ? v : m;      // NaN propagates like ATen
        }
    y[(int64_t)bc * n + i] = m;
}

// exact 2x2 windows (H = 2 Ho, W = 2 Wo, W % 4 == 0, 16-byte aligned rows): a thread owns two output pixels = 4 x 2 inputs,
// 16-byte loads / stores (the UNet's pooling, unet.py:79, always has this shape)
__global__ __launch_bounds__(256) void maxpool2x2_kernel(const float* __restrict__ x, float* __restrict__ y, float* __restrict__ full,
                                                         const float* __restrict__ scale, const float* __restrict__ shift, int C,
                                                         int H, int W) {
    const int Wo = W >> 1, Ho = H >> 1, wq = W >> 2;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)Ho * wq) return;
    const int bc = blockIdx.y, c = bc % C;
    const int oy = (int)(i / wq), q = (int)(i % wq);
    const float sc = scale ? scale[c] : 1.f, sh = shift ? shift[c] : 0.f;
    const int64_t o = (int64_t)bc * H * W + (int64_t)(2 * oy) * W + 4 * q;
    cs_f4 r0 = *reinterpret_cast<const cs_f4*>(x + o), r1 = *reinterpret_cast<const cs_f4*>(x + o + W);
    if (scale) {
        r0 = r0 * sc + sh;
        r1 = r1 * sc + sh;
    }
    if (full) {
        *reinterpret_cast<cs_f4*>(full + o) = r0;
        *reinterpret_cast<cs_f4*>(full + o + W) = r1;
    }
    float m[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {               // same visiting order as the generic kernel: (0,0) (0,1) (1,0) (1,1); NaN propagates
        float a = -INFINITY;
        const float v4[4] = {r0[2 * k], r0[2 * k + 1], r1[2 * k], r1[2 * k + 1]};
#pragma unroll
        for (int j = 0; j < 4; ++j) a = (v4[j] > a || v4[j] != v4[j]) ? v4[j] : a;
        m[k] = a;
    }
    *reinterpret_cast<float2*>(y + (int64_t)bc * Ho * Wo + (int64_t)oy * Wo + 2 * q) = make_float2(m[0], m[1]);
}

extern "C" int cwfa_maxpool_f32(const float* x, float* y, float* full, const float* scale, const float* shift, int B, int C,
                                int H, int W, int Ho, int Wo, void* stream) {
    CWFA_REQUIRE(x && y, CWFA_E_INVAL, "cwfa_maxpool_f32: null pointer");
    CWFA_REQUIRE(!(scale && !shift), CWFA_E_INVAL, "cwfa_maxpool_f32: scale without shift");
    CWFA_REQUIRE(B >= 0 && C >= 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0 && Ho <= H && Wo <= W, CWFA_E_SHAPE,
                 "cwfa_maxpool_f32: bad shape");
    CWFA_REQUIRE((int64_t)B * C <= 65535, CWFA_E_SHAPE, "cwfa_maxpool_f32: B*C too large");
    if (B == 0 || C == 0) return CWFA_OK;
    const int64_t n = (int64_t)Ho * Wo;
    if (H == 2 * Ho && W == 2 * Wo && (W & 3) == 0 && cwfa_aligned16(x) && (!full || cwfa_aligned16(full)) &&
        (reinterpret_cast<uintptr_t>(y) & 7) == 0) {
        hipLaunchKernelGGL(maxpool2x2_kernel, dim3((unsigned)((n / 2 + 255) / 256), B * C), dim3(256), 0, (hipStream_t)stream, x, y, full,
                           scale, shift, C, H, W);
        CWFA_LAUNCH_CHECK("cwfa_maxpool_f32");
        return CWFA_OK;
    }
    hipLaunchKernelGGL(maxpool_kernel, dim3((unsigned)((n + 255) / 256), B * C), dim3(256), 0, (hipStream_t)stream, x, y, full,
                       scale, shift, C, H, W, Ho, Wo);
    CWFA_LAUNCH_CHECK("cwfa_maxpool_f32");
    return CWFA_OK;
}

// ------------------------------------------------------------------------------------------------ LayerNorm over (C,H,W)
__global__ __launch_bounds__(256) void sample_stats_kernel(const float* __restrict__ x, double* __restrict__ stats,
                                                           int64_t CHW) {
    __shared__ double red[16];
    const int b = blockIdx.y;
    const float* px = x + (int64_t)b * CHW;
    double s = 0.0, q = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < CHW; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = px[i];
        s += v;
        q += (double)v * v;
    }
    s = cwfa_block_sum(s, red);
    q = cwfa_block_sum(q, red);
    if (threadIdx.x == 0) {
        atomicAdd(&stats[2 * b], s);
        atomicAdd(&stats[2 * b + 1], q);
    }
}

extern "C" int cwfa_sample_stats_f32(const float* x, double* stats, int B, int64_t CHW, void* stream) {
    CWFA_REQUIRE(x && stats, CWFA_E_INVAL, "cwfa_sample_stats_f32: null pointer");
    CWFA_REQUIRE(B >= 0 && CHW >= 0 && B <= 65535, CWFA_E_SHAPE, "cwfa_sample_stats_f32: bad shape");
    if (B == 0 || CHW == 0) return CWFA_OK;
    int blocks = (int)((CHW + 256 * 16 - 1) / (256 * 16));
    if (blocks < 1) blocks = 1;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(sample_stats_kernel, dim3(blocks, B), dim3(256), 0, (hipStream_t)stream, x, stats, CHW);
    CWFA_LAUNCH_CHECK("cwfa_sample_stats_f32");
    return CWFA_OK;
}

__global__ __launch_bounds__(256) void layernorm_apply_kernel(const float* __restrict__ x, const double* __restrict__ stats,
                                                              const float* __restrict__ w, const float* __restrict__ bsh,
                                                              float eps, float* __restrict__ y, int64_t CHW) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= CHW) return;
    const int b = blockIdx.y;
    const double mean = stats[2 * b] / (double)CHW;
    double var = stats[2 * b + 1] / (double)CHW - mean * mean;
    if (var < 0.0) var = 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    const float v = (x[(int64_t)b * CHW + i] - (float)mean) * rstd;
    y[(int64_t)b * CHW + i] = v * (w ? w[i] : 1.f) + (bsh ? bsh[i] : 0.f);
}

extern "C" int cwfa_layernorm_apply_f32(const float* x, const double* stats, const float* w, const float* b, float eps,
                                        float* y, int B, int64_t CHW, void* stream) {
    CWFA_REQUIRE(x && stats && y, CWFA_E_INVAL, "cwfa_layernorm_apply_f32: null pointer");
    CWFA_REQUIRE(B >= 0 && CHW >= 0 && B <= 65535, CWFA_E_SHAPE, "cwfa_layernorm_apply_f32: bad shape");
    if (B == 0 || CHW == 0) return CWFA_OK;
    hipLaunchKernelGGL(layernorm_apply_kernel, dim3((unsigned)((CHW + 255) / 256), B), dim3(256), 0, (hipStream_t)stream, x,
                       stats, w, b, eps, y, CHW);
    CWFA_LAUNCH_CHECK("cwfa_layernorm_apply_f32");
    return CWFA_OK;
}

// ------------------------------------------------------------------------------------------------ attention + combine
#define CWFA_ATT_MAXC 16
__global__ __launch_bounds__(256) void attention_combine_kernel(const float* __restrict__ mean, const float* __restrict__ w1,
                                                                const float* __restrict__ b1, const float* __restrict__ w2,
                                                                const float* __restrict__ b2, const float* __restrict__ m,
                                                                const float* __restrict__ x, float* __restrict__ out, int C,
                                                                int64_t L) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= L) return;
    const int b = blockIdx.y;
    const float* pm = mean + (int64_t)b * C * L;
    float hid[CWFA_ATT_MAXC];
    for (int o = 0; o < C; ++o) hid[o] = b1[o];
    for (int c = 0; c < C; ++c) {
        const float vm = i > 0 ? pm[(int64_t)c * L + i - 1] : 0.f;
        const float v0 = pm[(int64_t)c * L + i];
        const float vp = i + 1 < L ? pm[(int64_t)c * L + i + 1] : 0.f;
        for (int o = 0; o < C; ++o) {
            const float* ww = w1 + ((int64_t)o * C + c) * 3;
            hid[o] = fmaf(ww[2], vp, fmaf(ww[1], v0, fmaf(ww[0], vm, hid[o])));
        }
    }
    for (int o = 0; o < C; ++o) hid[o] = hid[o] > 0.f ? hid[o] : 0.f;
    for (int o = 0; o < C; ++o) {
        float a = b2[o];
        for (int c = 0; c < C; ++c) a = fmaf(w2[o * C + c], hid[c], a);
        const float att = 1.f / (1.f + expf(-a));
        const int64_t idx = ((int64_t)b * C + o) * L + i;
        out[idx] = m ? (x ? x[idx] : 0.f) + (m[idx] * 2.f) * (att - 0.5f) : att;
    }
}

extern "C" int cwfa_attention_combine_f32(const float* mean, const float* w1, const float* b1, const float* w2,
                                          const float* b2, const float* m, const float* x, float* out, int B, int C,
                                          int64_t HW, void* stream) {
    CWFA_REQUIRE(mean && w1 && b1 && w2 && b2 && out, CWFA_E_INVAL, "cwfa_attention_combine_f32: null pointer");
    CWFA_REQUIRE(C > 0 && C <= CWFA_ATT_MAXC, CWFA_E_SHAPE, "cwfa_attention_combine_f32: C=%d not in 1..%d", C, CWFA_ATT_MAXC);
    CWFA_REQUIRE(B >= 0 && HW >= 0 && B <= 65535, CWFA_E_SHAPE, "cwfa_attention_combine_f32: bad shape");
    if (B == 0 || HW == 0) return CWFA_OK;
    hipLaunchKernelGGL(attention_combine_kernel, dim3((unsigned)((HW + 255) / 256), B), dim3(256), 0, (hipStream_t)stream, mean,
                       w1, b1, w2, b2, m, x, out, C, HW);
    CWFA_LAUNCH_CHECK("cwfa_attention_combine_f32");
    return CWFA_OK;
}
