// Internal (non-ABI) interface between conv2d.hip (dispatch + direct kernels) and conv_wino.hip (Winograd kernels).
#pragma once
#include "common.h"

// 3x3 convolutions with at least g_cwfa_wino_min_cout output channels run the Winograd kernels (1.5x / 2.25x fewer MFMAs).
// The packed weight image differs (G-transformed), so pack and launch must agree on these predicates.
// The threshold is a run-time option ("winograd_min_cout", cwfa_set_option): 1 = default, a huge value = direct only.
extern int g_cwfa_wino_min_cout;
static inline bool cwfa_wino_selected(int ks, int Cout) { return ks == 3 && Cout >= g_cwfa_wino_min_cout; }

// "winograd_2d" = v: layers with more than 64 and at least v output channels take the 2-D F(2x2,3x3) kernel
// (conv_wino2d.hip); 0 = never
extern int g_cwfa_wgrad_rows;    // conv_bwd.hip
extern int g_cwfa_wgrad_split;   // conv_bwd.hip
extern int g_cwfa_wino_2d;
static inline bool cwfa_wino2d_selected(int Cout) { return g_cwfa_wino_2d != 0 && Cout > 64 && Cout >= g_cwfa_wino_2d; }
int64_t cwfa_wino2d_packed_floats(int Cout, int Cin);
int cwfa_wino2d_pack(const float* w, float* packed, int Cout, int Cin, hipStream_t stream);
int cwfa_wino2d_conv(const float* x, const float* w_packed, float* y, int B, int Cin, int H, int W, int Cout, int64_t x_bs,
                     int64_t y_bs, const cwfa_conv_opts& o, hipStream_t stream);

int64_t cwfa_wino_packed_floats(int Cout, int Cin);
int cwfa_wino_pack(const float* w, float* packed, int Cout, int Cin, hipStream_t stream);
int cwfa_wino_conv(const float* x, const float* w_packed, float* y, int B, int Cin, int H, int W, int Cout, int64_t x_bs,
                   int64_t y_bs, const cwfa_conv_opts& o, hipStream_t stream);
// fused sub-network layer, 64 channels: y = ELU(conv1x1(ELU(conv3x3(x) + b3)) + b1 + x)
int cwfa_wino_layer(const float* x, const float* w3_packed, const float* b3, const float* w1_panel, const float* b1, float* y,
                    int B, int H, int W, int64_t x_bs, int64_t y_bs, hipStream_t stream, float* hidden = nullptr,
                    int64_t hidden_bs = 0);
