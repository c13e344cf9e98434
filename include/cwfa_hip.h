/*
 * cwfa_hip.h -- C ABI of libcwfa_hip.so: hand-written HIP kernels (gfx950 / MI355X) for the
 * CWFA inverse (reconstruction) and forward-NLL hot path.
 *
 * The reference (pvjosue/CWFA) is pure Python on torch ops and has NO native boundary; each entry
 * point below therefore names the reference *Python* op sequence (file:line, relative to the
 * reference root) whose stock ATen kernels it replaces.  The Python host layer (cwfa_amd/) binds
 * these with ctypes; INTEGRATION.md shows the stub.
 *
 * Conventions
 *   - plain C: raw device pointers + sizes, no torch / C++ types.  `stream` is a hipStream_t passed
 *     as void* (NULL = the null stream).
 *   - every launch function ENQUEUES work on `stream` and returns immediately: 0 on success,
 *     <0 on error (CWFA_E_*).  Nothing aborts, nothing synchronises, nothing is allocated.
 *   - cwfa_last_error() returns a thread-local message for the last failing call.
 *   - tensors are fp32, NCHW, innermost (H,W) plane contiguous; where a `*_bs` argument exists it
 *     is the batch stride IN ELEMENTS, the channel stride is H*W (so channel-sliced views of a
 *     larger tensor are passed without a copy: Split / torch.cat never materialise).
 *   - index tables are int64 (torch.LongTensor, as the reference stores them) and are applied
 *     bit-exactly.
 */
#ifndef CWFA_HIP_H
#define CWFA_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CWFA_VERSION 100

enum {
    CWFA_OK = 0,
    CWFA_E_INVAL = -1,  /* null pointer / bad enum / negative size */
    CWFA_E_SHAPE = -2,  /* shape not supported by the kernel */
    CWFA_E_ALIGN = -3,  /* pointer / stride alignment */
    CWFA_E_HIP = -4     /* HIP runtime error at launch */
};

/* soft-clamp of the multiplicative coupling coefficient: s = clamp * f(a)
 * FrEIA/modules/coupling_layers.py:50-60; all_in_one_block.py:216 (TANH) */
enum { CWFA_CLAMP_NONE = 0, CWFA_CLAMP_ATAN = 1, CWFA_CLAMP_TANH = 2, CWFA_CLAMP_SIGMOID = 3 };

/* activations usable in conv epilogues */
enum { CWFA_ACT_NONE = 0, CWFA_ACT_ELU = 1, CWFA_ACT_PRELU = 2, CWFA_ACT_GELU = 3, CWFA_ACT_RELU = 4 };

int cwfa_version(void);
const char* cwfa_last_error(void);
/* process-wide tuning options (set before packing weights; pack and launch consult the same value):
 *   "winograd_min_cout" : 3x3 convolutions with at least this many output channels use the Winograd kernels
 *                         (default 1; a value above every Cout selects the direct kernels everywhere).
 *   "winograd_2d"       : v > 0: layers with more than 64 and at least v output channels use the 2-D F(2x2,3x3) kernel
 *                         (+1..13 % on plain convolutions, slower with a load-side prologue); 0: the 1-D F(2,3) kernel
 *                         as for the narrower layers.
 *   "split_products"    : 6 (default): the split-bf16 kernels form every fp32 product from six bf16 products
 *                         (fp32-equivalent); 1: plain bf16 operands (BASELINE.json configs[4]).
 *   "wgrad_rows"        : 0: the 3x3 weight gradient always takes its first (register-staged) form.
 *   "wgrad_split"       : 1: the 3x3 weight gradient (16-byte aligned rows) runs on the bf16 matrix cores in the split
 *                         arithmetic of "split_products" (training in split / bf16 precision); 0 (default): fp32 MFMA.
 *   "split3x3_xcd_map"  : 0: (ablation) blocks of the split 3x3 kernel in plain (spatial tile, cout tile) order instead of
 *                         the XCD-aware one.
 *   "split3x3_rows16"   : 0: (ablation) the 64-channel tiling of the split 3x3 kernel always on 8-row tiles.
 * returns 0, or CWFA_E_INVAL for an unknown name. */
int cwfa_set_option(const char* name, int value);

/* ------------------------------------------------------------------------------------------------
 * Haar wavelets
 * ---------------------------------------------------------------------------------------------- */

/* HaarTransform1D.forward(rev=False) fused with Split.forward   INN_utils.py:151-161, graph_topology.py:73-80
 * x [B,D,H,W] -> lo [B,D/2,H,W] = (x[2i]+x[2i+1])*f,  hi = (x[2i]-x[2i+1])*f,  f = fp32(1/sqrt2).
 * lo / hi are separate views (pass out+0 and out+(D/2)*HW with the same batch stride to get the
 * reference's single [B,D,H,W] output).  Bit-exact with the reference (same two fp32 ops). */
int cwfa_haar1d_fwd_f32(const float* x, float* lo, float* hi, int B, int D, int64_t HW,
                        int64_t x_bs, int64_t lo_bs, int64_t hi_bs, void* stream);

/* HaarTransform1D.forward(rev=True) fused with Split.forward(rev=True)=torch.cat   INN_utils.py:157-161
 * x[2i] = (lo[i]+hi[i])*f, x[2i+1] = (lo[i]-hi[i])*f.  hi == NULL means hi = 0 (unused by callers today). */
int cwfa_haar1d_inv_f32(const float* lo, const float* hi, float* x, int B, int D, int64_t HW,
                        int64_t lo_bs, int64_t hi_bs, int64_t x_bs, void* stream);

/* HaarDownsampling.forward   FrEIA/modules/reshapes.py:273-300  (2x2 spatial Haar, [B,C,H,W] <-> [B,4C,H/2,W/2])
 * fac = 0.5*rebalance (fwd) or 0.5/rebalance (rev); order_by_wavelet selects channel order j*C+c vs c*4+j. */
int cwfa_haar2d_fwd_f32(const float* x, float* y, int B, int C, int H, int W, int order_by_wavelet, float fac,
                        void* stream);
int cwfa_haar2d_inv_f32(const float* y, float* x, int B, int C, int H, int W, int order_by_wavelet, float fac,
                        void* stream);

/* 3-D Haar over 2 x 2 x 2 tiles in ONE pass: the depth Haar (INN_utils.py:142-161; lo | hi halves on the channel axis)
 * followed by the spatial Haar of every band (reshapes.py:273-300), y = haar2d(haar1d(x)): x [B,D,H,W] (batch stride
 * x_bs) <-> y [B,4D,H/2,W/2] (contiguous).  order_by_wavelet / fac as cwfa_haar2d_*; bit-identical to the two-launch
 * composition; 8 bytes of traffic per element instead of 16. */
int cwfa_haar3d_fwd_f32(const float* x, float* y, int B, int D, int H, int W, int order_by_wavelet, float fac, int64_t x_bs,
                        void* stream);
int cwfa_haar3d_inv_f32(const float* y, float* x, int B, int D, int H, int W, int order_by_wavelet, float fac, int64_t x_bs,
                        void* stream);

/* ------------------------------------------------------------------------------------------------
 * Permutations (index gathers)
 * ---------------------------------------------------------------------------------------------- */

/* y[.., i, ..] = x[.., perm[i], ..] along axis (1 = channels, 2 = rows, 3 = columns).
 * PermuteRandom.forward fixed_transforms.py:37-41; PermuteDim.forward INN_utils.py:73-81;
 * AllInOneBlock hard permutation all_in_one_block.py:191-196 (a 0/1 1x1 conv == channel gather). */
int cwfa_gather_f32(const float* x, const int64_t* perm, float* y, int B, int C, int H, int W, int axis,
                    int64_t x_bs, int64_t y_bs, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Affine coupling apply (+ fused input gather, per-channel affine, log-det, sum of squares)
 * ---------------------------------------------------------------------------------------------- */

typedef struct {
    const float* s_raw;   /* [B,C,H,W] pre-clamp multiplicative coefficient; NULL = no scaling (NICE)      */
    const float* t;       /* [B,C,H,W] additive coefficient; NULL = 0                                      */
    int64_t s_bs, t_bs;   /* batch strides (elements)                                                      */
    int clamp_kind;       /* CWFA_CLAMP_*                                                                  */
    float clamp;          /* alpha (2.0 default)                                                           */
    float pre_scale;      /* multiplies s_raw and t first (AllInOneBlock `a *= 0.1`), 1.0 otherwise        */
    int t_neg_div_sqrt2;  /* 1: t := -t / sqrt(2)  (wavelet_flow_subnetwork2D_first pass-through,
                             networks.py:671) so the torch.cat of that subnet never materialises          */
    const int64_t* perm;  /* optional gather applied to the INPUT of this stage: v_in[p] = v_prev[g(p)]    */
    int perm_axis;        /* 1 / 2 / 3 (see cwfa_gather_f32); ignored when perm == NULL                    */
    int gin;              /* 1: subtract the channel mean of s at every pixel (GIN, coupling_layers.py:355) */
} cwfa_affine_stage;

/* One coupling apply:  fwd y = exp(s)*x' + t,   rev y = (x' - t)*exp(-s),   x' = gather(x),  s = clamp*f(pre*s_raw)
 * coupling_layers.py:209-217,281-289,492-500; all_in_one_block.py:213-225.
 * x == NULL means x = 0 (z at temperature 0, CWFA.py:54-55).
 * logdet (nullable): double[B], ACCUMULATED with += sum_{c,h,w} s (fwd) or -= (rev).
 * sumsq  (nullable): double[1], ACCUMULATED with += sum y^2  (||Z||^2 of CWFA.py:970). */
int cwfa_affine_f32(const float* x, float* y, const cwfa_affine_stage* st, int rev, int B, int C, int H, int W,
                    int64_t x_bs, int64_t y_bs, double* logdet, double* sumsq, void* stream);

/* ActNorm / AllInOneBlock global affine, per channel:   invertible_resnet.py:78-81; all_in_one_block.py:181-196
 * mode 0: y = x*scale[c] + shift[c];   mode 1: y = (x - shift[c]) / scale[c];
 * optional channel gather on the input (perm, applied before the affine) or on the output
 * (perm_out: y[:, i] = v[:, perm_out[i]] of the affine result), at most one of them. */
int cwfa_channel_affine_f32(const float* x, float* y, const float* scale, const float* shift, int mode,
                            const int64_t* perm_in, const int64_t* perm_out, int B, int C, int64_t HW,
                            int64_t x_bs, int64_t y_bs, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Fused flow chain for CAT-type steps (every s,t depends on the conditions only)
 * ---------------------------------------------------------------------------------------------- */
#define CWFA_CHAIN_MAX 8

typedef struct {
    int n_stages;
    cwfa_affine_stage stage[CWFA_CHAIN_MAX];   /* in EXECUTION order for the requested direction */
    /* Optional (both or none; device int32): the stages' channel / row gathers composed by the caller, so that a kernel
     * reads where each stage takes its s,t rows from instead of walking the permutation tables (a chain of dependent
     * loads): src_c [n_stages+1][C], src_h [n_stages+1][H]; row k < n_stages = the channel / image row at which stage k
     * reads its coefficients for OUTPUT channel / row i, row n_stages = where the travelling value starts (for
     * cwfa_chain_fwd_f32 the final permutation is part of the composition).  Used by the 16-byte row kernels of
     * cwfa_chain_inv_f32 / cwfa_chain_fwd_f32; every other entry point ignores them. */
    const int32_t* src_c;
    const int32_t* src_h;
} cwfa_chain;

/* Inverse of one whole conditional step in ONE launch:  GraphINN.forward(rev=True) over
 * [PermuteRandom^-1, (CAT^-1, Permute^-1) x n, CAT_first^-1, Split^-1(cat), HaarTransform1D^-1]
 * graph_inn.py:280-311 over the graph of networks.py:305-366.
 *   v = z (NULL = 0);  for each stage: v <- A_k^-1(gather_k(v));  x = Haar1D^-1(cat[low, v])
 * z [B,C,H,W], low [B,C,H,W] -> x [B,2C,H,W].  logdet nullable (accumulated, rev sign). */
int cwfa_chain_inv_f32(const float* z, const float* low, float* x, const cwfa_chain* ch, int B, int C, int H, int W,
                       int64_t z_bs, int64_t low_bs, int64_t x_bs, double* logdet, void* stream);

/* Forward (NLL direction) of one whole conditional step in ONE launch:
 *   (low, v) = Split(Haar1D(x));  for each stage: v <- A_k(gather_k(v));  z = gather_final(v)
 * final_perm (nullable) is the trailing PermuteRandom (networks.py:353-357).
 * logdet += sum s (per sample), sumsq += sum z^2 (CWFA.py:966-978). */
int cwfa_chain_fwd_f32(const float* x, float* low, float* z, const cwfa_chain* ch, const int64_t* final_perm,
                       int B, int C, int H, int W, int64_t x_bs, int64_t low_bs, int64_t z_bs,
                       double* logdet, double* sumsq, void* stream);

/* ------------------------------------------------------------------------------------------------
 * 2-D convolution on fp32 MFMA (implicit GEMM, no im2col), stride 1, zero padding ks/2, groups 1
 * replaces nn.Conv2d / F.conv2d at networks.py:212-219,488-492,537,621-638; unet.py:99-104,66
 * ---------------------------------------------------------------------------------------------- */

/* number of floats of the packed weight image for a [Cout,Cin,ks,ks] filter bank */
int64_t cwfa_conv2d_packed_floats(int Cout, int Cin, int ks);
/* repack torch-layout weights [Cout,Cin,ks,ks] (device) into the kernel's tiled image (device).
 * transposed != 0: source is ConvTranspose2d layout [Cin,Cout,ks,ks] with ks==2, stride 2 (unet.py:166);
 * the packed image is then the equivalent 1x1 conv with 4*Cout outputs ordered (dy*2+dx)*Cout+co. */
int cwfa_conv2d_pack_f32(const float* w, float* packed, int Cout, int Cin, int ks, int transposed, void* stream);

typedef struct {
    const float* bias;        /* [Cout] nullable                                                          */
    int act;                  /* CWFA_ACT_* applied to (acc + bias)                                       */
    const float* prelu_alpha; /* device scalar (nn.PReLU() has ONE parameter)                             */
    const float* residual;    /* nullable, same shape as y; added AFTER act                               */
    int64_t res_bs;
    int act2;                 /* CWFA_ACT_* applied after the residual add                                */
    const float* in_scale;    /* nullable [Cin]: input is read as x*in_scale[c] + in_shift[c]             */
    const float* in_shift;    /*   (eval-mode BatchNorm of the producer folded into the load; exact with  */
                              /*    zero padding because padding is inserted after the affine)            */
    int in_affine_bs;         /* 0: in_scale/in_shift are [Cin] shared by the batch; else the per-sample
                                 stride ([B,Cin] tables: BatchNorm x dropout2d channel mask, unet.py:80,86)  */
    const float* in_add;      /* nullable, same shape as x: added after the affine (UNet skip add)        */
    int64_t in_add_bs;
    int upshuffle2;           /* 1: the Cout = 4*Co outputs are written pixel-shuffled to [B,Co,2H,2W]
                                 (ConvTranspose2d k=2 s=2 as a 1x1 conv, unet.py:166); residual/bias then
                                 index the shuffled output ([Co])                                         */
    int in_blocked8;          /* cwfa_conv3x3_split_f32 only: x is channel-blocked [B][Cin/8][H][W][8] (what
                                 cwfa_subnet_layer_split_f32 writes with layout bit 1); Cin % 8 == 0, no in_* */
    int out_blocked8;         /* y is written channel-blocked [B][Cout/8][H][W][8]: cwfa_conv2d_f32 for 1x1 banks
                                 with 33..64 outputs, bias only (the first convolution of a coupling sub-network
                                 feeding cwfa_subnet_layer_split_f32 with layout bit 0); cwfa_conv3x3_split_f32
                                 with a bias / PReLU epilogue (consumer: in_blocked8)                         */
    const float* in_cat;      /* cwfa_conv2d_f32, 1x1 banks with <= 64 outputs, no other in_*: the input is the channel
                                 concatenation cat(x, in_cat) WITHOUT materialising it (the input of a coupling
                                 sub-network, cat(half, *conditions): coupling_layers.py:74-87, all_in_one_block.py:
                                 244-247).  x holds in_cat_c1 channels; the bank is packed for Cin = in_cat_from +
                                 channels(in_cat) with zero columns in [in_cat_c1, in_cat_from), in_cat_from % 16 == 0 */
    int64_t in_cat_bs;
    int in_cat_from, in_cat_c1;
    double* out_stats;        /* cwfa_conv3x3_split_f32 only (NCHW output, bias / PReLU epilogue): nullable [2*Cout]; the
                                 launch ADDS (sum y, sum y^2) of its output over (B,H,W) per channel -- the train-mode
                                 BatchNorm statistics of the layer that follows the convolution (unet.py:99-107), taken
                                 from the accumulators instead of a second pass over y (cwfa_channel_stats_f32)       */
    int prelu_per_channel;    /* cwfa_conv3x3_split_f32 only: prelu_alpha points to Cout slopes, one per output channel
                                 (slope 1.0 = no activation on that channel): several filter banks that read the same
                                 input -- conv1 (+ PReLU) and downsample (plain) of the condition nets' ResidualBlocks,
                                 networks.py:212-219,229-233 -- run as ONE convolution                                 */
} cwfa_conv_opts;

int cwfa_conv2d_f32(const float* x, const float* w_packed, float* y, int B, int Cin, int H, int W, int Cout, int ks,
                    int64_t x_bs, int64_t y_bs, const cwfa_conv_opts* opts, void* stream);

/* One residual layer of the coupling sub-network (networks.py:624-631,660-665) fused into ONE launch, 64 channels:
 *     y = ELU( conv1x1( ELU( conv3x3(x) + b3 ) ) + b1 + x )
 * The 3x3 accumulators are consumed in registers as the B operand of the 1x1 MFMAs; the hidden map never exists in
 * memory.  w3_packed: cwfa_conv2d_pack_f32 image of the [64,64,3,3] filter; w1_panel: cwfa_subnet_pack1x1_f32 image
 * (4096 floats) of the [64,64,1,1] filter. */
int cwfa_subnet_pack1x1_f32(const float* w, float* panel, void* stream);
int cwfa_subnet_layer_f32(const float* x, const float* w3_packed, const float* b3, const float* w1_panel, const float* b1,
                          float* y, int B, int H, int W, int64_t x_bs, int64_t y_bs, void* stream);
/* The same layer for the training forward: additionally writes the hidden map h = ELU(conv3x3(x) + b3) [B,64,H,W] (batch
 * stride hidden_bs) that the backward needs, from the accumulators it already holds (one extra store per element). */
int cwfa_subnet_layer_tape_f32(const float* x, const float* w3_packed, const float* b3, const float* w1_panel, const float* b1,
                               float* y, float* hidden, int B, int H, int W, int64_t x_bs, int64_t y_bs, int64_t hidden_bs,
                               void* stream);

/* ------------------------------------------------------------------------------------------------
 * Condition-net 3-D part:  Conv3d(1->K,3^3,pad 1) -> PReLU -> Conv3d(K->1,3^3,pad 1), fused
 * networks.py:221-225,239 on the [B,1,H,W,D] view of a [B,D,H,W] tensor (depth = channel axis).
 * w1 [K,1,3,3,3] (kh,kw,kd), b1 [K], alpha scalar, w2 [1,K,3,3,3], b2 [1].  K <= 32.
 * ---------------------------------------------------------------------------------------------- */
int cwfa_conv3d_1k1_f32(const float* x, const float* w1, const float* b1, const float* alpha, const float* w2,
                        const float* b2, float* y, int B, int D, int H, int W, int K, void* stream);
/* The same stage with fp32-equivalent arithmetic on the bf16 matrix cores (three bf16 pieces per operand, six products, fp32
 * accumulation; "split_products" = 1: plain bf16 operands): both convolutions on v_mfma_f32_16x16x32_bf16, the sum over the
 * depth taps of the second one taken through the accumulator input while a wave walks the depth axis (csrc/conv3d_split.hip).
 * Same arguments and semantics as cwfa_conv3d_1k1_f32; K <= 32. */
int cwfa_conv3d_1k1_split_f32(const float* x, const float* w1, const float* b1, const float* alpha, const float* w2,
                              const float* b2, float* y, int B, int D, int H, int W, int K, void* stream);

/* ------------------------------------------------------------------------------------------------
 * LRNN pieces (unet.py:72-113,161-195; networks.py:244-262,468-555)
 * ---------------------------------------------------------------------------------------------- */

/* per-channel batch statistics for train-mode BatchNorm2d: stats[2*C] += (sum, sumsq) over (B,H,W), double */
int cwfa_channel_stats_f32(const float* x, double* stats, int B, int C, int64_t HW, int64_t x_bs, void* stream);
/* turn (sum,sumsq,count) or running (mean,var) into scale/shift: scale = w*rsqrt(var+eps), shift = b - mean*scale
 * stats != NULL: batch statistics (biased variance);  else running_mean / running_var.
 * mask_bc (nullable, [B*C]): per-(sample,channel) multiplier (F.dropout2d keep-mask / (1-p)); scale and shift are then
 * [B*C] tables scale[b,c] = s_c*m, shift[b,c] = t_c*m; otherwise [C] (B is ignored).
 * weight / bias NULL mean 1 / 0 (so a bare mask can be turned into an input affine). */
int cwfa_bn_fold_f32(const double* stats, double count, const float* running_mean, const float* running_var,
                     const float* weight, const float* bias, float eps, const float* mask_bc, int B, float* scale,
                     float* shift, int C, void* stream);
/* train-mode buffer bookkeeping of nn.BatchNorm2d (unet.py:101-107) from the same statistics: running_mean/var updated
 * with `momentum` (unbiased variance), num_batches_tracked (int64, nullable) incremented. */
/* cwfa_bn_fold_f32 + cwfa_bn_running_update_f32 in one launch; additionally the dropout factor may be given as the raw uniform
 * draw (mask_u [B*C], drop_p, keep_scale = fp32(1 - p)): m = (u >= p) / keep_scale, F.dropout2d unet.py:80,86; zero_stats: the
 * statistics buffer is cleared for its next accumulation (cwfa_channel_stats_f32 adds into it). */
int cwfa_bn_finish_f32(double* stats, double count, float* running_mean, float* running_var, long long* num_batches_tracked,
                       float momentum, int update_running, const float* weight, const float* bias, float eps, const float* mask_bc,
                       const float* mask_u, float drop_p, float keep_scale, int B, float* scale, float* shift, int C, int zero_stats,
                       void* stream);
int cwfa_bn_running_update_f32(const double* stats, double count, float momentum, float* running_mean, float* running_var,
                               long long* num_batches_tracked, int C, void* stream);
/* F.adaptive_max_pool2d(x, (Ho,Wo)) (unet.py:79) with an optional per-channel affine applied BEFORE the max
 * (the producer's BatchNorm); also writes the affine result at full resolution to `full` (nullable; the skip). */
int cwfa_maxpool_f32(const float* x, float* y, float* full, const float* scale, const float* shift, int B, int C,
                     int H, int W, int Ho, int Wo, void* stream);
/* per-sample statistics over (C,H,W) for nn.LayerNorm([C,H,W]): stats[2*B] += (sum, sumsq), double */
int cwfa_sample_stats_f32(const float* x, double* stats, int B, int64_t CHW, void* stream);
/* y = (x - mean_b) * rstd_b * w[chw] + b[chw]   networks.py:490 (eps 1e-5) */
int cwfa_layernorm_apply_f32(const float* x, const double* stats, const float* w, const float* b, float eps,
                             float* y, int B, int64_t CHW, void* stream);
/* GlobalAttention (networks.py:249-262) fused with the LRNN combine (networks.py:552-554):
 *   att = sigmoid(W2 . relu(conv1d_k3(mean over the flattened H*W sequence) ) )
 *   out = x + m * 2 * (att - 0.5)            (m, x nullable -> out = att) */
int cwfa_attention_combine_f32(const float* mean, const float* w1, const float* b1, const float* w2, const float* b2,
                               const float* m, const float* x, float* out, int B, int C, int64_t HW, void* stream);
/* y = x*scale[b*C+c]  (F.dropout2d channel mask / drop_path; mask generated by the caller) */
int cwfa_scale_channels_f32(const float* x, const float* scale_bc, float* y, int B, int C, int64_t HW, void* stream);
/* y = a*x + b*z (elementwise; z nullable) */
int cwfa_axpby_f32(const float* x, const float* z, float a, float b, float* y, int64_t n, void* stream);

/* ------------------------------------------------------------------------------------------------
 * fp32-equivalent 1x1 / 3x3 convolution and ConvTranspose2d(k2,s2) on the bf16 matrix pipe (same reference ops
 * as cwfa_conv2d_f32 with ks = 1 or 3, tiles of 256 output channels: nn.Conv2d 1x1 networks.py:488-492, nn.ConvTranspose2d unet.py:166).  Each fp32 operand is
 * split exactly into three bf16 pieces, six partial products are accumulated in fp32.
 *   cwfa_split_workspace_bytes / cwfa_split_input_f32 : x [B,Cin,HW] (+ per-channel or per-(sample,channel) affine,
 *       + added tensor, as cwfa_conv_opts in_*) -> three bf16 planes in `ws` (16-byte aligned);
 *   cwfa_conv_split_packed_bytes / _pack_f32 : torch weight [Cout,Cin] (or ConvTranspose2d [Cin,Co,2,2]) -> split image;
 *   cwfa_conv_split_f32 : y = epilogue(W . x) with bias / act / residual / act2 / upshuffle2 of `opts` (in_* must be null). */
int64_t cwfa_split_workspace_bytes(int B, int Cin, int64_t HW);
int cwfa_split_input_f32(const float* x, void* ws, int B, int Cin, int64_t HW, int64_t x_bs, const float* in_scale,
                         const float* in_shift, int64_t in_affine_bs, const float* in_add, int64_t in_add_bs, void* stream);
int64_t cwfa_conv_split_packed_bytes(int Cout, int Cin, int ks);
int cwfa_conv_split_pack_f32(const float* w, void* packed, int Cout, int Cin, int ks, int transposed, void* stream);
/* 3x3 (stride 1, zero padding 1) with the same arithmetic straight from the fp32 tensor: the kernel applies the load-side
 * affine / added tensor of `opts` and splits on the way into LDS (no workspace, no extra pass); v_mfma_f32_16x16x32_bf16,
 * one wave per SIMD (csrc/conv_split3x3.hip).  packed = cwfa_conv3x3_split_pack_f32(torch weight [Cout,Cin,3,3]):
 * cwfa_conv3x3_split_packed_bytes(Cout, Cin) bytes, 16-byte aligned.  Epilogue: bias / act / residual / act2 of `opts`. */
int64_t cwfa_conv3x3_split_packed_bytes(int Cout, int Cin);
int cwfa_conv3x3_split_pack_f32(const float* w, void* packed, int Cout, int Cin, void* stream);
int cwfa_conv3x3_split_f32(const float* x, const void* w_packed, float* y, int B, int Cin, int H, int W, int Cout,
                           int64_t x_bs, int64_t y_bs, const cwfa_conv_opts* opts, void* stream);
int cwfa_conv_split_f32(const void* ws, const void* w_packed, float* y, int B, int Cin, int H, int W, int Cout, int ks, int64_t y_bs,
                           const cwfa_conv_opts* opts, void* stream);
/* 7x7 (stride 1, zero padding 3; nn.Conv2d(C, C, 7, 1, 3) of the ConvNeXt block, networks.py:488) in the same arithmetic on the same
 * kernel: 3-pixel halo, 49 taps, a 49-step period over two 16-channel chunks.  Cout <= 64; epilogue: bias only; no in_* prologue. */
int64_t cwfa_conv7x7_split_packed_bytes(int Cout, int Cin);
int cwfa_conv7x7_split_pack_f32(const float* w, void* packed, int Cout, int Cin, void* stream);
int cwfa_conv7x7_split_f32(const float* x, const void* w_packed, float* y, int B, int Cin, int H, int W, int Cout,
                           int64_t x_bs, int64_t y_bs, const cwfa_conv_opts* opts, void* stream);

/* The LAST convolution of a coupling sub-network (3x3, 64 -> 2n channels: [s_raw | t], networks.py:633-638) with the affine
 * coupling applied from its accumulators -- s and t never reach memory:
 *     fwd:  y = x * exp(s) + t            rev:  y = (x - t) * exp(-s)            logdet[b] += (rev ? -1 : +1) * sum s
 *     s = soft_clamp(pre_scale * s_raw), t <- pre_scale * t      (coupling_layers.py:50-60,87-110; all_in_one_block.py:206-224)
 * x / y: the ACTIVE half [B,n,H,W] of the block (planes contiguous, batch strides x_bs / y_bs; may be the same tensor).
 * The filter bank is packed with its rows interleaved so that s_j and t_j of a pixel land in the same lane:
 *     cwfa_couple_rows(n, rows): rows[r] = source row (0..2n-1) of packed row r, or -1 (zero row), r < cwfa_couple_rows(n, NULL)
 *     w' = w[rows] (zeros for -1), bias' likewise;  packed = cwfa_conv3x3_split_pack_f32(w', Cout = that row count)
 * n <= 64.  logdet nullable (float64[B], accumulated). */
typedef struct {
    const float* x;
    float* y;
    int64_t x_bs, y_bs;
    int n;                      /* channels of the active half = half the outputs of the bank                           */
    int clamp_kind;             /* CWFA_CLAMP_*                                                                          */
    float clamp, pre_scale;
    int rev;
    double* logdet;
    int in_blocked8;            /* the convolution's input is channel-blocked (see cwfa_conv_opts.in_blocked8)           */
} cwfa_couple;
int cwfa_couple_rows(int n, int* rows);
int cwfa_conv3x3_split_couple_f32(const float* x, const void* w_packed, const float* bias_rows, int B, int Cin, int H, int W,
                                  int64_t x_bs, const cwfa_couple* cp, void* stream);

/* The fused sub-network layer  y = ELU(W1 . ELU(conv3x3(x, W3) + b3) + b1 + x), 64 channels (networks.py:624-631,660-665)
 * with BOTH convolutions on the split-bf16 core (three bf16 pieces per fp32 operand, six products, fp32 accumulation;
 * `split_products` = 1: plain bf16 operands), one persistent launch (csrc/conv_split_layer.hip).
 *   packed = cwfa_subnet_layer_split_pack_f32(w3 [64,64,3,3], w1 [64,64,1,1]): cwfa_subnet_layer_split_packed_bytes() bytes,
 *   16-byte aligned, 40 slices (36 of the 3x3 bank by (chunk of 16 input channels, tap), 4 of the 1x1 bank). */
int64_t cwfa_subnet_layer_split_packed_bytes(void);
int cwfa_subnet_layer_split_pack_f32(const float* w3, const float* w1, void* packed, void* stream);
int cwfa_subnet_layer_split_f32(const float* x, const void* packed, const float* b3, const float* b1, float* y, int B, int H,
                                int W, int64_t x_bs, int64_t y_bs, int layout, void* stream);
/* The FIRST layer of a sub-network in its composed form.  The 1x1 convolution in front of the three layers (networks.py:621-623,
 * 641-643) and the layer's 3x3 are both linear with nothing in between, so conv3x3(conv1x1(u) + b0) = conv3x3'(u | 1) with
 * W' = W3 o [W0 | b0] over u's channels and a constant-one channel (exact under zero padding: padded ones carry no bias): K = 9 x 32
 * instead of 9 x 64 -- half the convolution steps of the layer.  u: [B, u_ch <= 32, H, W] NCHW INCLUDING the ones channel;
 * x = conv1x1(u) + b0 (the residual, layout bit 0); packed = cwfa_subnet_layer_first_pack_f32(w3c [64,32,3,3] the composed bank,
 * zero in unused input channels; w1 [64,64,1,1]; w0c nullable): cwfa_subnet_layer_first_packed_bytes() bytes.
 * With w0c = [W0 | b0 | 0] ([64,32], the 1x1 bank over u's channels and the ones channel) in the packed image and x == NULL the
 * residual is not read either: the layer's 1x1 phase takes it as a third k step, [W1 | W0c] . [h ; u], from u's values at the tile's
 * pixels -- the sub-network's first 1x1 launch and its 64-channel map never exist.  SHORT form (pack with short_form = 1, launch
 * with layout bit 2 = 4, x == NULL, u_ch <= 16 = one 16-channel chunk): five instead of nine convolution steps per tile -- the steps
 * that pair only taps of the all-zero second chunk are not run. */
int64_t cwfa_subnet_layer_first_packed_bytes(void);
int cwfa_subnet_layer_first_pack_f32(const float* w3c, const float* w1, const float* w0c, int short_form, void* packed, void* stream);
int cwfa_subnet_layer_first_f32(const float* u, const float* x, const void* packed, const float* b3, const float* b1, float* y, int B,
                                int u_ch, int H, int W, int64_t u_bs, int64_t x_bs, int64_t y_bs, int layout, void* stream);
/* Tape form (training forward, SURVEY.md 8f row 1 / CWFA.py:966-1006): the same launch also writes the hidden map
 * h = ELU(conv3x3(x) + b3) [B,64,H,W] the backward needs, from the registers that hold it as the 1x1's operand.  NCHW maps. */
int cwfa_subnet_layer_split_tape_f32(const float* x, const void* packed, const float* b3, const float* b1, float* y, float* hid, int B,
                                     int H, int W, int64_t x_bs, int64_t y_bs, int64_t hid_bs, void* stream);
/* layout: bit 0 = x, bit 1 = y is CHANNEL-BLOCKED, [B][8 blocks][H][W][8 channels] (same size and batch strides as NCHW, 16-byte
 * aligned), instead of NCHW planes.  The maps between the layers of one sub-network are private to it; blocked, a staging entry
 * of the kernel (8 channels of a pixel) is two 16-byte loads instead of eight 4-byte ones and the four channels a lane holds for
 * a pixel leave as one 16-byte store (the 4-byte stores of the NCHW form held 15 % of a launch).  The consumer of a blocked map:
 * this function with bit 0, or cwfa_conv3x3_split_f32 / cwfa_conv3x3_split_couple_f32 with `in_blocked8`. */

/* ------------------------------------------------------------------------------------------------
 * Lenslet views (the step before the path): XLFMDatasetFull.extract_views XLFMDataset.py:212-242 followed by the
 * normalisation of CWFA.py:796-797.  image [B,1,Hs,Ws] (batch stride image_bs), coords_yx int32 [nviews][2] (device),
 * views [B,nviews,sh,sw]: window of sh x sw around each lenslet, clipped to the image, the clipped patch in the
 * bottom-right corner of a zero view, then (v - mean) / stdv (mean 0, stdv 1 = the plain extraction). */
int cwfa_extract_views_f32(const float* image, const int* coords_yx, float* views, int B, int Hs, int Ws, int nviews, int sh,
                           int sw, float mean, float stdv, int64_t image_bs, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Backward of the flow-step training loss (what torch autograd does under
 * `scaler.scale(full_loss).backward()` CWFA.py:1002-1006 for the NLL term of CWFA.py:966-978)
 * ---------------------------------------------------------------------------------------------- */

/* Gradient buffers of a chain: for stage k (same order as the cwfa_chain handed to cwfa_chain_fwd_f32) ds[k] receives
 * dL/d s_raw and dt[k] dL/d t (the tensors the sub-network produced), NULL = not wanted. */
typedef struct {
    float* ds[CWFA_CHAIN_MAX];
    float* dt[CWFA_CHAIN_MAX];
    int64_t ds_bs[CWFA_CHAIN_MAX], dt_bs[CWFA_CHAIN_MAX];
} cwfa_chain_grads;

/* L = gscale * 0.5 * sum z^2 - ldscale * sum_b logdet_b (+ <gz, z> for an upstream gradient gz, nullable), z the output
 * of cwfa_chain_fwd_f32 with the same chain / final_perm.  No stored activations: every stage input is recomputed by
 * inverting the stage.  gv0 (nullable) receives dL/d(detail band entering the chain).  accumulate != 0: the stage
 * gradients are ADDED to the buffers.  gld (nullable, float[B]): an upstream gradient dL/d(logdet_b) per sample, added to the
 * uniform -ldscale (what torch autograd hands the backward of `log_jac_det.mean()`, CWFA.py:978). */
int cwfa_chain_bwd_f32(const float* z, const float* gz, const cwfa_chain* ch, const cwfa_chain_grads* grads,
                       const int64_t* final_perm, float* gv0, int B, int C, int H, int W, int64_t z_bs, int64_t gz_bs,
                       int64_t gv0_bs, float gscale, float ldscale, int accumulate, const float* gld, void* stream);

/* Backward of the reconstruction term (CWFA.py:952-959: F.l1_loss / F.mse_loss(curr_gt, upsampled_vol)) through the
 * INVERSE pass xhat = cwfa_chain_inv_f32(z, low, ...) (CWFA.py:911): L = gscale' * sum |xhat - gt|^p, p = loss_kind
 * (1: gscale = weight/numel, 2: gscale = 2*weight/numel).  `ch` is the FORWARD-order chain (as for cwfa_chain_fwd_f32);
 * the gradients go to the same buffers as cwfa_chain_bwd_f32's (accumulate != 0: added) since both passes use the same
 * coefficients.  loss_sum (nullable, double[1]) += sum |xhat - gt|^p.  No stored activations.
 * loss_kind = 0: `gt` holds an upstream gradient dL/dxhat itself (torch autograd; scaled by gscale).  gz_out / glow_out
 * (nullable, contiguous [B,C,H,W]): dL/d(latent input z) and, for loss_kind 0, dL/d(low) of the inverse pass. */
int cwfa_chain_inv_bwd_f32(const float* xhat, const float* gt, const cwfa_chain* ch, const cwfa_chain_grads* grads, int B, int C,
                           int H, int W, int64_t xhat_bs, int64_t gt_bs, float gscale, int loss_kind, int accumulate,
                           double* loss_sum, float* gz_out, float* glow_out, void* stream);

/* Backward of ONE affine stage y = cwfa_affine_f32(x, st, rev) without a gather (torch autograd of the coupling blocks that are
 * not part of a fused chain: coupling_layers.py:124-437, all_in_one_block.py:206-224): g = dL/dy, gld (nullable, float[B]) =
 * dL/d(logdet_b); writes dL/dx, dL/d s_raw, dL/d t (each nullable, contiguous [B,C,H,W]).  GIN stages included. */
int cwfa_affine_bwd_f32(const float* x, const float* g, const cwfa_affine_stage* st, int rev, int B, int C, int H, int W,
                        int64_t x_bs, int64_t g_bs, const float* gld, float* gx, float* gs_raw, float* gt_raw, void* stream);

/* Weight gradient of a stride-1, zero-padded ("same") convolution, ks = 1 or 3, on the fp32 matrix cores:
 *   dw[co][ci][ky][kx] = beta * dw + sum_{b,y,x} dy[b][co][y][x] * x[b][ci][y+ky-ks/2][x+kx-ks/2]     (torch layout)
 * and, with db != NULL, the bias gradient db[co] = beta * db + sum_{b,y,x} dy[b][co][y][x] from the same pass over dy.
 * x [B,Cin,H,W] (batch stride x_bs), dy [B,Cout,H,W] (dy_bs).  workspace: cwfa_conv2d_wgrad_workspace_bytes() bytes
 * (per-worker partial filter banks, summed in a fixed order: the result is deterministic). */
int64_t cwfa_conv2d_wgrad_workspace_bytes(int B, int Cin, int H, int W, int Cout, int ks);
int cwfa_conv2d_wgrad_f32(const float* x, const float* dy, float* dw, float* db, void* workspace, int B, int Cin, int H, int W,
                          int Cout, int ks, int64_t x_bs, int64_t dy_bs, float beta, void* stream);

/* ELU backward from the layer output a = ELU(q):  y = g * (a > 0 ? 1 : a + 1) (+ add, nullable).  n elements per sample
 * (multiple of 4), batch strides in elements. */
int cwfa_elu_bwd_f32(const float* g, const float* a, const float* add, float* y, int B, int64_t n, int64_t g_bs, int64_t a_bs,
                     int64_t add_bs, int64_t y_bs, void* stream);

/* Backward of the condition net's 3-D stage y = W2 * PReLU(W1 * x + b1) + b2 (Conv3d 1 -> K -> 1 over (H, W, depth),
 * networks.py:221-225,239; x, y, dy: [B,D,H,W] with the depth axis as channel axis; w1 [K,1,3,3,3], w2 [1,K,3,3,3]) in
 * separate passes over a materialised hidden volume q / m [B,K,D,H,W]:
 *   hidden_fwd: q = W1 * x + b1;   hidden_bwd: m = (W2^T * dy) . PReLU'(q), dalpha (nullable, double[1]) += sum (W2^T*dy).min(q,0);
 *   input_bwd:  dx = sum_k W1[k]^T * m[k];
 *   wgrad: out32[k][tap] = beta*out32 + sum_p act(a[k][p]) * src[p + sign*(tap-1)] (tap < 27; column 27 = sum_p act(a[k][p])
 *          when want_bias), act = PReLU(alpha) if alpha != NULL -- dW2 = wgrad(a = q, alpha, src = dy, sign = -1),
 *          dW1 | db1 = wgrad(a = m, src = x, sign = +1, want_bias).  K <= 32; out32 is a [32][32] float buffer. */
int cwfa_conv3d_hidden_fwd_f32(const float* x, const float* w1, const float* b1, float* q, int B, int D, int H, int W, int K,
                               void* stream);
int cwfa_conv3d_hidden_bwd_f32(const float* dy, const float* w2, const float* q, const float* alpha, float* m, double* dalpha,
                               int B, int D, int H, int W, int K, void* stream);
int cwfa_conv3d_input_bwd_f32(const float* m, const float* w1, float* dx, int B, int D, int H, int W, int K, void* stream);
int64_t cwfa_conv3d_wgrad_workspace_bytes(int B, int D, int H, int W);
int cwfa_conv3d_wgrad_f32(const float* a, const float* src, const float* alpha, float* out32, void* workspace, int B, int D, int H,
                          int W, int K, int sign, int want_bias, float beta, void* stream);

/* PReLU backward (single alpha > 0) from the layer output o = PReLU(q):  y = g * (o > 0 ? 1 : alpha),
 * dalpha (nullable, double[1]) += sum g * min(q, 0).  n elements per sample, batch strides in elements. */
int cwfa_prelu_bwd_f32(const float* g, const float* o, const float* alpha, float* y, double* dalpha, int B, int64_t n, int64_t g_bs,
                       int64_t o_bs, int64_t y_bs, void* stream);

/* Backward pieces of the UNet (unet.py:72-113,161-195) with train-mode BatchNorm (CWFA.py:532):
 *   plane_affine: y = x * scale[(b,)c] + shift[(b,)c] (+ add)  (per_sample != 0: [B,C] tables) -- the BatchNorm (x dropout mask)
 *                 output as a tensor (the inference path applies it on load inside the next convolution);
 *   bn_bwd_stats: stats[2c] += sum g*m, stats[2c+1] += sum g*m*y over (B,H,W), m = mask_bc[b][c] or 1 (double[2C], zeroed by caller);
 *   bn_act_bwd:   out = (A[(b,)c]*g + Bc[c] + Cc[c]*y) * PReLU'(y) with y = PReLU(q) the conv output (alpha NULL: no activation),
 *                 dalpha (nullable) += sum (...)*min(q,0) -- A, Bc, Cc folded by the caller from the statistics;
 *   maxpool2_bwd: g_full = (first maximum of each 2x2 window of `full` ? g_pool : 0) + g_skip (nullable). */
int cwfa_plane_affine_f32(const float* x, const float* scale, const float* shift, int per_sample, const float* add, float* y, int B,
                          int C, int64_t HW, int64_t x_bs, int64_t add_bs, int64_t y_bs, void* stream);
int cwfa_bn_bwd_stats_f32(const float* g, const float* y, const float* mask_bc, double* stats, int B, int C, int64_t HW, int64_t g_bs,
                          int64_t y_bs, void* stream);
int cwfa_bn_act_bwd_f32(const float* g, const float* y, const float* A, int per_sample, const float* Bc, const float* Cc,
                        const float* alpha, float* out, double* dalpha, int B, int C, int64_t HW, int64_t g_bs, int64_t y_bs,
                        int64_t out_bs, void* stream);
int cwfa_maxpool2_bwd_f32(const float* full, const float* g_pool, const float* g_skip, float* g_full, int B, int C, int H, int W,
                          void* stream);

/* Backward pieces of the LRNN's mean-volume branch (ConvNeXt networks.py:468-503, GlobalAttention :244-262, combine :552-554):
 *   gelu:          mode 0: y = GELU(p) + other (nullable);  mode 1: y = other * GELU'(p)                       (n elements)
 *   layernorm_bwd: LayerNorm over the n = C*H*W elements of each sample, y = xhat*w + b with xhat = (v - mean[b])*invstd[b]:
 *                  gv = dL/dv, dw += sum_b g*xhat, db += sum_b g (dw, db ACCUMULATED); stats: double[2B] scratch, zeroed by the caller;
 *   attention_bwd: out = x + 2 m (att - 0.5), att = sigmoid(W2 relu(W1 *3 mean + b1) + b2) over the flattened H*W sequence
 *                  (C <= 8): gm = dL/dm, pgrad (double[C*C*3 + C + C*C + C], accumulated) = dL/d[w1 | b1 | w2 | b2]; dL/dx = g. */
int cwfa_gelu_f32(const float* p, const float* other, float* y, int64_t n, int mode, void* stream);
int cwfa_layernorm_bwd_f32(const float* g, const float* v, const float* w, const float* mean, const float* invstd, double* stats,
                           float* gv, float* dw, float* db, int B, int64_t n, void* stream);
int cwfa_attention_bwd_f32(const float* mean, const float* w1, const float* b1, const float* w2, const float* b2, const float* m,
                           const float* g, float* gm, double* pgrad, int B, int C, int64_t HW, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CWFA_HIP_H */
