// Argument structures shared by BOTH precision builds of the kernels (namespace rd = bf16 / fp32, namespace rd_f16 = fp16) and by
// rd_api.cpp.  They live in their own namespace so that the `rd` -> `rd_f16` renaming of the fp16 build does not duplicate the types.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rdt {

struct ConvArgs {
  const void* src1; const void* src2; const void* w; const float* bias;
  void* dst1; void* dst2; float* stats;
  int N, Hin, Win, C1, C2, H1, W1, Cout, KH, KW, stride, pad, dil, OH, OW, act, D1;
  float slope, scale_h, scale_w;
  int M, K, Kpad, ups;
  int s2d;     // data gradient of the same layer on the source: src1 is read space-to-depth, C1 = 4 classes x (C1 / 4) channels (rd_conv_desc.in_s2d)
  int d2s;     // forward of an exact 2x nearest up-sampling 3x3 layer computed on the source: Cout = 4 classes x D1 channels, depth-to-space store (rd_conv_desc.out_d2s)
  int pool2;   // data gradient of an exact 2x nearest up-sampling layer: 2x2 output blocks are summed and stored at half resolution
  const void* add1;   // optional [M][Cout] tensor added to the result before rounding (a gradient's earlier contribution); D1 == Cout only
  // consumer-side BatchNorm apply (round 4): src1 is the PRODUCER's raw convolution output y; the staging gather computes
  // act(in_scale[c] * y + in_shift[c]) per element (rounded to the activation type: bit-identical to rd_affine_act's z) -- the activated
  // tensor is never written.  Padding stays zero; src2 (the skip half of a concat) is read as it is.  nullptr: src1 is read as it is.
  const float* in_scale; const float* in_shift; int in_act; float in_slope;
  // BatchNorm-backward statistics epilogue (round 4, data gradients): dst1 is dz of a BatchNorm-ed producer whose raw output is bn_y
  // ([M][D1], the layout of dst1); `stats` then receives per-block (sum g, sum g * xhat) rows, g = dz * act'(bn_scale*y + bn_shift),
  // xhat = (y - bn_mean) * bn_rstd, over the STORED (rounded) dz -- what rd_bn_act_bwd_recompute's reduce pass would compute.
  const void* bn_y; const float *bn_scale, *bn_shift, *bn_mean, *bn_rstd; int bn_act; float bn_slope;
};

struct WgradArgs {
  const void* src1; const void* src2; const void* dy; float* slab;
  int N, Hin, Win, C1, C2, H1, W1, Cout, KH, KW, stride, pad, OH, OW;
  float scale_h, scale_w;
  int M, K, ups, nsplit, rows_per_split;
  const float* in_scale; const float* in_shift; int in_act; float in_slope;   // consumer-side BatchNorm apply on src1 (see ConvArgs)
};

struct LwgGemm { const void* x1; const void* x2; const void* dy; float* slab; int M, C1, C2, Cout, nsplit, rows_per_split; };

struct LwgReduce { const float* slab; float* dw; int64_t elems; int nsplit, accumulate; };

// deferred split-K reduction of one convolution weight gradient (launch_wgrad with defer) and a batch of them by value in kernel arguments
struct WgradReduceItem { const float* slab; float* dw; int Cout, Cin, KH, KW, nsplit, accumulate; };
static const int WGRAD_BATCH_MAX = 48;      // 48 x (40 + 4 + 4) B + 8 B: under the 4 KiB kernel-argument limit
struct WgradReduceBatch { WgradReduceItem item[WGRAD_BATCH_MAX]; int SL[WGRAD_BATCH_MAX]; int first[WGRAD_BATCH_MAX + 1]; int n; };

// LayerNorm parameter gradients of one (gamma, beta) pair: up to four per-row partial buffers [rows][C][2] = (dbeta, dgamma) terms, summed
// in order (mirrors rd_ln_grad_item)
struct LnGradItem { const float* partial[4]; float* dgamma; float* dbeta; int rows[4]; int C, nparts, accumulate, reserved; };
static const int LN_GRAD_BATCH_MAX = 32;
struct DwWgradItem { const float* partial; float* dw; int rows, C, KK, accumulate; };      // mirrors rd_dw_wgrad_item
static const int DW_WGRAD_BATCH_MAX = 64;
struct ColsumItem { const float* partial; float* out; int rows, C, accumulate, reserved; };      // mirrors rd_colsum_item
static const int COLSUM_BATCH_MAX = 64;      // 64 x 32 B by value in the kernel arguments

static const int LWG_MAX_ITEMS = 64, LWG_MAX_REDS = 96;   // 64 x 56 B and 96 x 32 B: both under the 4 KiB kernel-argument limit

struct LoftrW { const void *wq, *wk, *wv, *wm, *w0, *w2; const float *g1, *b1, *g2, *b2; };

struct LoftrSaved { void *q, *k, *v, *att, *mpre, *msg, *hid, *m2pre; float* stats; };

struct LoftrGrads {
  const void* dout; void *dm2pre, *dhid, *dmpre, *datt, *dq, *dk, *dv, *dx, *dsrc;
  float *lnp1, *lnp2, *dg1, *db1, *dg2, *db2; int accumulate, defer_ln;      // defer_ln = 1: leave the LayerNorm partials, no finalize launches
  int dsrc_accumulate, reserved;                                                    // dsrc_accumulate = 1: dsrc += (cross attention)
};

// ---- routing options (round 5): the kernel-selection switches that used to be read with getenv() at launch time.  One process-wide table in
// rd_api.cpp, written ONLY through the C ABI (rd_set_option / rd_clear_options: names = the lower-case identifiers below, values clamped to
// the ranges listed there), read by the launch code through rd_opt(): the product path consults no environment variable.  Tests use them to
// route small cases to a particular kernel, tools/bench_*.py to A/B block shapes.
enum RdOpt {
  OPT_CONV_PAR, OPT_CONV3X3_MIN_BLOCKS, OPT_CONV_STEM_MIN_M, OPT_WGRAD_BLOCKS, OPT_WGRAD_TINY_MIN_M, OPT_CONV3X3_W8, OPT_PATCH_BN_MAX,
  OPT_CONV3X3_G8, OPT_CONV1X1_MIN_M, OPT_CONV_FEW_MIN_M, OPT_FRAG_V128, OPT_FRAG_V64, OPT_FRAG_V32, OPT_FRAG_SPLIT, OPT_FRAG_SPLIT_BLOCKS,
  OPT_FRAG32_V128, OPT_FRAG32_V64, OPT_FRAG_LIN, OPT_CONV3X3_FRAG, OPT_BN_GEN_PPT, OPT_BN_VEC_PER, OPT_WGRAD_TR_TW, OPT_FRAG_DB, OPT_WGRAD_FIT, OPT_HEAD_NP, OPT_HEAD_CPI, OPT_PW_MIN_M, OPT_PW_KS, OPT_BN_SLAB, OPT_COUNT
};

}  // namespace rdt

extern "C" int rd_opt(int id, int dflt);      // the option's value if it has been set, else dflt
extern "C" int rd_opt_is_set(int id);
