/*
 * riders_hip.h -- C ABI of libriders_hip.so, the MI355X (gfx950) kernel library behind the RIDERS
 * RC-Net / Scale-Map-Learner training hot path.
 *
 * The reference (MMOCKING/RIDERS) has no FFI layer: its seam is torch.nn.Module and every device op is
 * a stock ATen/cuDNN/torchvision call.  Each entry below replaces the ATen dispatch at the cited
 * reference call site (paths relative to the reference root).  INTEGRATION.md shows the ctypes binding
 * a reference maintainer would add.
 *
 * Conventions
 *   - all pointers are DEVICE pointers owned by the caller (PyTorch allocates every tensor, saved-for-
 *     backward buffer and workspace); the library keeps no device memory between calls;
 *   - every entry is an asynchronous enqueue on `stream` (a hipStream_t passed as void*);
 *   - activations are NHWC ("channels_last"), dtype RD_F32, RD_BF16 or RD_F16; parameters, gradients of
 *     parameters, statistics and losses are always fp32 in the reference's layouts (OIHW, [out,in]);
 *   - return value: 0 = ok, negative = argument error (message via rd_last_error_string()),
 *     positive = hipError_t from the launch.  No C++ exception crosses the boundary.
 */
#ifndef RIDERS_HIP_H
#define RIDERS_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define RD_F32 0
#define RD_BF16 1
#define RD_F16 2  /* IEEE half activations (BASELINE.json configs[4]); same kernels, conversions and MFMA opcodes of the fp16 build */

#define RD_ACT_NONE 0
#define RD_ACT_RELU 1
#define RD_ACT_LRELU 2 /* utils/net_utils.py:15  LeakyReLU(0.20) */
#define RD_ACT_RELU6 3

int rd_version(void);
const char* rd_last_error_string(void);

/* ---- convolution family -------------------------------------------------------------------------
 * One descriptor serves forward, data-gradient and weight-gradient.  The logical conv input is the
 * channel concatenation [src1 (C1 ch) | src2 (C2 ch)] of spatial size Hin x Win; with `upsample` set
 * src1/src2 are physically H1 x W1 and read through F.interpolate(mode='nearest') index arithmetic.
 * replaces: utils/net_utils.py:84-91 (Conv2d), :195-198 (UpConv2d), :564-569 (DecoderBlock concat),
 *           RCNet/linear_attention.py:121-131 (nn.Linear as 1x1), modules/midas/blocks.py:28-39,83-88,147,185-191 */
typedef struct rd_conv_desc {
  int32_t dtype;               /* RD_F32 / RD_BF16 / RD_F16 activations                                        */
  int32_t N, Hin, Win;         /* logical input size                                                  */
  int32_t C1, C2;              /* channels of src1 / src2 (C2 = 0: single source)                     */
  int32_t upsample, H1, W1;    /* nearest-upsample the sources from H1 x W1 to Hin x Win              */
  int32_t Cout, KH, KW, stride, pad;
  int32_t in_dilate;           /* 1 for forward; s for the data-gradient of a stride-s convolution    */
  int32_t OH, OW;              /* output spatial size                                                 */
  int32_t act; float slope;    /* epilogue activation for convs without BatchNorm                     */
  int32_t D1;                  /* output channels [0,D1) -> dst1, [D1,Cout) -> dst2 (D1 = Cout: one)  */
  int32_t out_reduce2;         /* 1: sum every 2x2 block of output pixels and store it at (OH/2) x (OW/2): the data gradient of a layer that
                                  up-samples its input by exactly 2 (UpConv2d, utils/net_utils.py:195-198) lands at SOURCE resolution, the
                                  full-resolution gradient is never written.  Only where rd_conv_out_reduce2_ok says so. */
  int32_t out_d2s;             /* 1: the forward of an exact-2x nearest up-sampling 3x3 layer (UpConv2d, utils/net_utils.py:195-198) computed ON THE
                                  SOURCE: the descriptor is the 3x3 / stride-1 convolution of the H1 x W1 source with Cout = 4 x D1 output channels
                                  = (parity class (a, b), channel) and weights packed with mode 2 (per-class 2x2 effective kernels); channel
                                  tile (a, b) of source pixel (i, j) is stored as pixel (2 i + a, 2 j + b) of the (2 OH) x (2 OW) x D1 output
                                  (depth to space), and the structurally zero (tap, class) blocks are skipped.  Statistics rows: Cout columns =
                                  4 rows of D1.  Only where rd_conv_up2_ok says so. */
  int32_t in_s2d;              /* 1: the data gradient of the same layer on the source: src1 is the (2 Hin) x (2 Win) x (C1 / 4) gradient of the
                                  layer's output, read as an Hin x Win x C1 tensor of (parity class, channel) channels (space to depth); weights
                                  packed with mode 3; the structurally zero (tap, class) blocks of the K axis are skipped.  The result is the
                                  gradient at SOURCE resolution (no 2x2 reduction pass).  Only where rd_conv_up2_dgrad_ok says so. */
} rd_conv_desc;

/* elements (of the activation dtype) in a packed weight buffer for `rows` x (KH*KW*C).  K axes that can be 9 taps x whole 128-byte channel
 * chunks get room for a second, MFMA-fragment-ordered copy behind the row-major one (written by the pack calls for 3x3 kernels, read by the
 * register-fed 3x3 kernel of rd_conv_fwd) */
int64_t rd_conv_packed_elems(int32_t rows, int32_t K, int32_t dtype);
/* OIHW fp32 -> packed; mode 0: forward operand, mode 1: data-gradient operand (transposed + flipped), mode 2 (3x3 only): forward operand of the
 * exact-2x up-sampling layer on its source (rd_conv_desc.out_d2s): 4 Cout rows = (parity class, channel), each class's taps pre-summed;
 * mode 3 (3x3 only): its data-gradient operand (rd_conv_desc.in_s2d): Cin rows x 9 taps x 4 Cout (class, channel) columns, transposed + flipped */
int rd_conv_pack_weights(const float* w_oihw, void* packed, int32_t Cout, int32_t Cin, int32_t KH, int32_t KW,
                         int32_t mode, int32_t dtype, void* stream);
/* The same re-layout for MANY weights in one launch (after an optimizer step every cached operand of every conv / linear
   layer is stale: RC-Net has 159 of them, 4.4 us each as separate launches).  `items` is a DEVICE array of n descriptors. */
typedef struct rd_pack_item {
  const float* w_oihw; void* packed;
  int32_t Cout, Cin, KH, KW, mode, dtype;
  int32_t Cin_src;             /* 0 or the real input-channel count of w_oihw when the packed layout is zero-padded to Cin (mode 0) */
  int32_t reserved;            /* 0; 1 = keep the element-wise packing form for this item (A/B against the 16-byte-unit form of round 6; same bytes) */
} rd_pack_item;
/* forward operand with the input channels zero-padded from Cin_src to Cin (the 3-channel stems of utils/net_utils.py's encoders run
   on the 16-byte-vector kernels: pad the image with rd_pad_channels, un-pad the weight gradient with rd_unpad_weight_grad) */
int rd_conv_pack_weights_padded(const float* w_oihw, void* packed, int32_t Cout, int32_t Cin_src, int32_t Cin, int32_t KH, int32_t KW,
                                int32_t dtype, void* stream);
int rd_pad_channels(const void* src, void* dst, int64_t rows, int32_t C, int32_t Cpad, int32_t dtype, void* stream);
int rd_unpad_weight_grad(const float* dw_padded, float* dw, int32_t Cout, int32_t Cin, int32_t Cin_pad, int32_t taps, int32_t accumulate,
                         void* stream);
int rd_conv_pack_weights_batch(const rd_pack_item* items, int32_t n, void* stream);
/* the same for a table whose 16-bit items are fp16: inside the table the item dtype is 0 (fp32) or 1 (the 16-bit type), half_dtype = RD_BF16
   (identical to the call above) or RD_F16 says which 16-bit format that is */
int rd_conv_pack_weights_batch_half(const rd_pack_item* items, int32_t n, int32_t half_dtype, void* stream);
/* the same with a caller-built block map (DEVICE array of `blocks` int32[4] = (item index, block of the item, blocks of the item, 0); every item needs
   at least one block): the launch has exactly the blocks the operands need instead of 256 per item -- RC-Net's ~320 small operands made the
   fixed grid a launch bound by block dispatch (riders_amd.engine.refresh_packed sizes an item at one thread per 16-byte unit group) */
int rd_conv_pack_weights_batch_map(const rd_pack_item* items, int32_t n, int32_t half_dtype, const int32_t* block_map, int32_t blocks, void* stream);
/* Weight gradients of MANY 1x1 / linear layers in one launch + one ordered reduction (reference: autograd of the nn.Linear layers of
   RCNet/linear_attention.py:84-135; 96 products per RC-Net step).  gemm p: slab_p[split][Cout][C1+C2] = partial dY_p^T [X1_p | X2_p]
   over tokens [split*rows_per_split, ...); C1, C2, Cout multiples of the 16-byte vector (4 fp32 / 8 bf16), C1 % 64 == 0 when C2 > 0.  reduce q: dw_q[elems] (+)= sum of nsplit
   consecutive slabs (several gemms that share a weight write consecutive slabs of one reduce item).  Both arrays are HOST memory. */
typedef struct rd_lwg_gemm {
  const void* x1; const void* x2; const void* dy; float* slab;
  int32_t M, C1, C2, Cout, nsplit, rows_per_split;
} rd_lwg_gemm;
typedef struct rd_lwg_reduce {
  const float* slab; float* dw; int64_t elems; int32_t nsplit, accumulate;
} rd_lwg_reduce;
int rd_linear_wgrad_batch(const rd_lwg_gemm* gemms, int32_t n_gemm, const rd_lwg_reduce* reduces, int32_t n_reduce, int32_t dtype,
                          void* stream);
/* ---- fused LoFTR encoder layer (reference RCNet/linear_attention.py:84-135, LoFTREncoderLayer.forward; d_model = 128, nhead = 8,
   attention = 'linear', at most 32 tokens per sequence): x [N][L][128], src [N][S][128] (src == x: self attention), one workgroup per
   sequence.  Weights are rd_conv_pack_weights operands of the six nn.Linear layers (mode 0 for the forward, mode 1 for the backward);
   g1/b1/g2/b2 are norm1 / norm2 weight and bias (fp32).  The forward fills `saved` (activation dtype; stats fp32 [N*L][4] = mean1,
   rstd1, mean2, rstd2); the backward consumes it and writes dx [, dsrc], the LayerNorm parameter gradients, and the output gradients of
   the six linears (dq, dk, dv, dmpre, dhid, dm2pre) whose weight gradients are then one rd_linear_wgrad_batch call:
     Wq: (x, dq)  Wk: (src, dk)  Wv: (src, dv)  Wm: (att, dmpre)  W0: ([x | msg], dhid)  W2: (hid, dm2pre).
   lnp1 / lnp2 are [N][128][2] fp32 scratch. */
typedef struct rd_loftr_weights {
  const void *wq, *wk, *wv, *wm, *w0, *w2;
  const float *g1, *b1, *g2, *b2;
} rd_loftr_weights;
typedef struct rd_loftr_saved {
  void *q, *k, *v, *att, *mpre, *msg, *hid, *m2pre;
  float* stats;
} rd_loftr_saved;
typedef struct rd_loftr_grads {
  const void* dout;
  void *dm2pre, *dhid, *dmpre, *datt, *dq, *dk, *dv, *dx, *dsrc;
  float *lnp1, *lnp2, *dg1, *db1, *dg2, *db2;
  int32_t accumulate;
  int32_t defer_ln;            /* 1: leave the LayerNorm partials lnp1 / lnp2 ([N][128][2] = (dbeta, dgamma) terms per ROI) for rd_ln_grad_batch */
  int32_t dsrc_accumulate;     /* 1 (cross attention only): dsrc holds an earlier gradient contribution of `src`; the kernel stores dsrc + its own,
                                  rounded once (the transformer's second cross call feeds the first one's output: linear_attention.py:174-176) */
  int32_t reserved;
} rd_loftr_grads;
/* LayerNorm parameter gradients of many layer applications in ONE launch: an item is one (gamma, beta) pair with the partial buffers
 * [rows][C][2] of up to four applications (the transformer applies each layer to both token streams: RCNet/linear_attention.py:159-184),
 * added in the order given, each summed as rd_loftr_layer_bwd's own finalize does.  HOST array, passed by value. */
typedef struct rd_ln_grad_item {
  const float* partial[4]; float* dgamma; float* dbeta;
  int32_t rows[4]; int32_t C, nparts, accumulate, reserved;
} rd_ln_grad_item;
int rd_ln_grad_batch(const rd_ln_grad_item* items, int32_t n, void* stream);
int rd_loftr_layer_fwd(const void* x, const void* src, const rd_loftr_weights* w, void* out, const rd_loftr_saved* saved, int32_t N,
                       int32_t L, int32_t S, float eps_attn, float eps_ln, int32_t dtype, void* stream);
int rd_loftr_layer_bwd(const void* x, const void* src, const rd_loftr_weights* w_t, const rd_loftr_saved* saved,
                       const rd_loftr_grads* grads, int32_t N, int32_t L, int32_t S, float eps_attn, int32_t dtype, void* stream);
/* rows of the per-block BatchNorm statistics buffer stats[rows][Cout][2] written by rd_conv_fwd */
int32_t rd_conv_stats_rows(const rd_conv_desc* d);
int rd_conv_fwd(const rd_conv_desc* d, const void* src1, const void* src2, const void* w_packed, const float* bias,
                void* dst1, void* dst2, float* stats, void* stream);
/* Name of the kernel instantiation rd_conv_fwd / rd_conv_wgrad run this shape on, as rocprofv3's kernel trace prints it (without the
 * `void rd::` prefix and the argument list): bench.py groups its per-launch HIP-event timings by it so that its `roofline` object describes
 * the same kernel a `rocprofv3 --kernel-trace --stats` summary of the run ranks first.  Thread-local storage, valid until the next call. */
const char* rd_conv_fwd_kernel_name(const rd_conv_desc* d);
const char* rd_conv_wgrad_kernel_name(const rd_conv_desc* d);
/* 1 when rd_conv_fwd can run this descriptor with out_reduce2 = 1 (even OH / OW, single destination, no statistics, a kernel whose
 * epilogue can pair rows and columns in registers: the narrow-layer 3x3 kernel) */
int32_t rd_conv_out_reduce2_ok(const rd_conv_desc* d);
/* 1 when rd_conv_fwd can run this descriptor with out_d2s = 1 (3x3 / stride 1 / pad 1 on the source, Cout = 4 x D1, a kernel that implements it) */
int32_t rd_conv_up2_ok(const rd_conv_desc* d);
/* 1 when rd_conv_fwd can run this descriptor with in_s2d = 1 (3x3 / stride 1 / pad 1 at source resolution, C1 = 4 x the layer's output channels, C2 = 0) */
int32_t rd_conv_up2_dgrad_ok(const rd_conv_desc* d);
/* dst = round(conv(...) + addend): `addend` is a [N*OH*OW][Cout] tensor of the activation dtype, read once in the epilogue.  The engine's
 * backward hands over a tensor's EARLIER gradient contribution when a second consumer's data gradient arrives (skip connections of
 * utils/net_utils.py:564-569, residual blocks :250-330), instead of writing the second contribution and adding the two in a separate
 * pass.  One destination (D1 == Cout), no statistics; rd_conv_add_ok = 1 where the kernel the descriptor is routed to supports it. */
int32_t rd_conv_add_ok(const rd_conv_desc* d);
int rd_conv_fwd_add(const rd_conv_desc* d, const void* src1, const void* src2, const void* w_packed, const float* bias, const void* addend,
                    void* dst, void* stream);
/* fp32 workspace bytes needed by rd_conv_wgrad */
int64_t rd_conv_wgrad_workspace_bytes(const rd_conv_desc* d);
/* One sizing entry for the caller-owned buffers of a convolution layer (SURVEY 8b `rd_workspace_bytes(op, shape...)`; -1 = bad descriptor / op):
   the weight-gradient slabs, the forward's BatchNorm statistics rows ([rd_conv_stats_rows][Cout][2] floats), the packed forward operand and the
   packed data-gradient operand (bytes in the descriptor's dtype, incl. the fragment-ordered second copies). */
#define RD_WS_CONV_WGRAD 0
#define RD_WS_CONV_STATS 1
#define RD_WS_CONV_PACKED 2
#define RD_WS_CONV_PACKED_DGRAD 3
int64_t rd_workspace_bytes(int32_t op, const rd_conv_desc* d);
/* dw (OIHW fp32) = or += dY^T * gather(X); deterministic two-stage reduction */
int rd_conv_wgrad(const rd_conv_desc* d, const void* src1, const void* src2, const void* dy, float* workspace,
                  float* dw_oihw, int32_t accumulate, void* stream);
/* The same in two steps, so that the split-K reductions of many layers can share ONE launch (30+ launches of ~9 us per RC-Net step
 * otherwise): rd_conv_wgrad_partial writes the partial slabs into `workspace` and fills *item (host memory); rd_wgrad_reduce_batch sums
 * the slabs of every item into its dw_oihw in the fixed order of rd_conv_wgrad (bit-identical result).  The workspaces must stay
 * untouched until the batch has run; one weight must not appear twice in a batch. */
/* 1 when rd_conv_fwd runs this shape on the one-pixel-per-thread streaming kernel (a handful of channels, millions of pixels: the SML
 * `first` 3->3 convolution, the 32->1 head and its data gradient): hand such a layer over WITHOUT zero-padded input channels */
int32_t rd_conv_fwd_streams(const rd_conv_desc* d);
/* 1 when rd_conv_wgrad runs this shape on the register-accumulating streaming kernel (a handful of channels, millions of pixels: the SML
 * `first` 3->3 convolution, the 32->1 head): such layers should be handed over WITHOUT zero-padded input channels */
int32_t rd_conv_wgrad_streams(const rd_conv_desc* d);
typedef struct rd_wgrad_reduce_item { const float* slab; float* dw; int32_t Cout, Cin, KH, KW, nsplit, accumulate; } rd_wgrad_reduce_item;
int rd_conv_wgrad_partial(const rd_conv_desc* d, const void* src1, const void* src2, const void* dy, float* workspace,
                          float* dw_oihw, int32_t accumulate, rd_wgrad_reduce_item* item, void* stream);
int rd_wgrad_reduce_batch(const rd_wgrad_reduce_item* items, int32_t n, void* stream);

/* ---- fusions of the BatchNorm passes into the convolutions either side of them (round 4; utils/net_utils.py:84-91 conv -> BatchNorm2d ->
 * activation, its autograd).  A BatchNorm-ed convolution's raw output y and its per-channel (scale, shift) -- rd_bn_finalize -- stand in
 * for the activated tensor z = act(scale * y + shift), which is then never written:
 *   in_*  (forward / weight gradient of the CONSUMER): src1 is y; the staging gather applies scale / shift / activation per element,
 *         rounded to the activation dtype exactly as rd_affine_act would have stored z; padding stays zero; src2 is read as it is.
 *   bn_*  (data gradient of the CONSUMER, i.e. the kernel that produces dz): besides storing dz into dst1 the epilogue accumulates the
 *         BatchNorm-backward sums of the PRODUCER, (sum g, sum g * xhat) with g = dz * act'(bn_scale * y + bn_shift) and
 *         xhat = (y - bn_mean) * bn_rstd over the stored dz, into `stats` rows of [Cout][2] floats (columns >= D1 unused): the reduce pass
 *         of rd_bn_act_bwd_recompute disappears (rd_bn_bwd_from_partial finishes from these rows).  bn_y has dst1's layout ([pixels][D1]).
 * Null pointers switch a part off.  rd_conv_fusion_ok / rd_conv_wgrad_fusion_ok say whether the kernel a descriptor is routed to supports
 * the requested parts; callers fall back to the separate passes otherwise. */
typedef struct rd_conv_fusion {
  const float* in_scale; const float* in_shift; int32_t in_act; float in_slope;
  const void* bn_y; const float* bn_scale; const float* bn_shift; const float* bn_mean; const float* bn_rstd; int32_t bn_act; float bn_slope;
} rd_conv_fusion;
int32_t rd_conv_fusion_ok(const rd_conv_desc* d, const rd_conv_fusion* f);
/* rd_conv_fwd / rd_conv_fwd_add (addend may be NULL) with the fusions above */
int rd_conv_fwd_fused(const rd_conv_desc* d, const rd_conv_fusion* f, const void* src1, const void* src2, const void* w_packed,
                      const float* bias, const void* addend, void* dst1, void* dst2, float* stats, void* stream);
int32_t rd_conv_wgrad_fusion_ok(const rd_conv_desc* d, const rd_conv_fusion* f);
/* kernel instantiation names of the fused calls (as rd_conv_fwd_kernel_name / rd_conv_wgrad_kernel_name) */
const char* rd_conv_fused_kernel_name(const rd_conv_desc* d, const rd_conv_fusion* f);
const char* rd_conv_wgrad_fused_kernel_name(const rd_conv_desc* d, const rd_conv_fusion* f);
/* rd_conv_wgrad_partial with the in_* part (src1 = the producer's raw output) */
int rd_conv_wgrad_partial_fused(const rd_conv_desc* d, const rd_conv_fusion* f, const void* src1, const void* src2, const void* dy,
                                float* workspace, float* dw_oihw, int32_t accumulate, rd_wgrad_reduce_item* item, void* stream);

/* ---- BatchNorm2d (+ activation, + residual) -- utils/net_utils.py:86-91, :309-323 ----------------- */
int rd_bn_finalize(const float* stats, int32_t rows, int32_t C, double count, const float* gamma, const float* beta,
                   float eps, float momentum, int32_t training, float* running_mean, float* running_var,
                   float* save_mean, float* save_rstd, float* scale, float* shift, void* stream);
/* Wide layers on small maps in ONE launch per direction (csrc/rd_bn_slab.hip: a workgroup owns one 16-byte channel vector of every pixel, so the
 * batch statistics it needs are its own).  rd_bn_slab_ok = 1 for <= 2 816 pixels and >= 64 channel vectors (512 channels of the 16-bit types).
 * rd_bn_finalize_apply = rd_bn_finalize (training mode, from the producer's statistics rows) + rd_affine_act without residual;
 * rd_bn_act_bwd_slab = rd_bn_act_bwd_recompute without dres: no partial rows, no coefficient buffer.  Same arithmetic; the statistics are
 * summed in another (fixed) order.  rd_bn_slab_kernel_name: which 0 = forward, 1 = backward, as rd_conv_fwd_kernel_name. */
int32_t rd_bn_slab_ok(int64_t pixels, int32_t C, int32_t dtype);
int rd_bn_finalize_apply(const float* stats, int32_t rows, const void* y, const float* gamma, const float* beta, float eps, float momentum,
                         float* running_mean, float* running_var, float* save_mean, float* save_rstd, float* scale, float* shift, void* out,
                         int64_t pixels, int32_t C, int32_t act, float slope, int32_t dtype, void* stream);
int rd_bn_act_bwd_slab(const void* dz, const void* y, const float* mean, const float* rstd, const float* scale, const float* shift, float* dgamma,
                       float* dbeta, int32_t accumulate, void* dy, int64_t pixels, int32_t C, int32_t act, float slope, int32_t dtype, void* stream);
const char* rd_bn_slab_kernel_name(int32_t which, int64_t pixels, int32_t dtype, int32_t act);
/* out = act(scale[c]*y + shift[c] + residual); scale/shift/residual may be NULL */
int rd_affine_act(const void* y, const float* scale, const float* shift, const void* residual, void* out, int64_t pixels,
                  int32_t C, int32_t act, float slope, int32_t dtype, void* stream);
/* out = act2(z + residual) with z = act1(scale[c]*y + shift[c]) rounded to the activation dtype as rd_affine_act stores it: the BatchNorm
   apply of a residual block's second convolution fused with the block's add + activation (utils/net_utils.py:309-323); z is not written.
   Channel counts with a 16-byte vector form only (rd_affine_act_add_ok). */
int32_t rd_affine_act_add_ok(int32_t C, int32_t dtype);
int rd_affine_act_add(const void* y, const float* scale, const float* shift, int32_t act1, float slope1, const void* residual, void* out,
                      int64_t pixels, int32_t C, int32_t act2, float slope2, int32_t dtype, void* stream);
int32_t rd_bn_bwd_rows(int64_t pixels, int32_t C);
/* full BN(+act) backward: dy = dBN(dz * act'(z)); dres (optional) = dz * act'(z); dgamma/dbeta fp32 */
int rd_bn_act_bwd(const void* dz, const void* z, const void* y, const float* save_mean, const float* save_rstd,
                  const float* scale, float* partial /* [rows][C][2] */, float* coef /* [2][C] */, float* dgamma,
                  float* dbeta, int32_t accumulate, void* dy, void* dres, int64_t pixels, int32_t C, int32_t act,
                  float slope, int32_t dtype, void* stream);
/* same, for z = act(scale*y + shift) WITHOUT a residual: the activation's argument is recomputed from y (bit-identical to the forward's
   expression), so z is not read (one tensor less in both passes); z is only consulted when C is not a multiple of the 16-byte vector */
int rd_bn_act_bwd_recompute(const void* dz, const void* z, const void* y, const float* save_mean, const float* save_rstd,
                            const float* scale, const float* shift, float* partial, float* coef, float* dgamma, float* dbeta,
                            int32_t accumulate, void* dy, void* dres, int64_t pixels, int32_t C, int32_t act, float slope,
                            int32_t dtype, void* stream);
/* the same call restricted to some of its three launches (bit 0: reduce pass, bit 1: finalize, bit 2: apply pass; 7 = rd_bn_act_bwd_recompute):
   bench.py times the passes one by one so that its roofline object can name the dominant KERNEL of any family */
int rd_bn_act_bwd_recompute_phases(const void* dz, const void* z, const void* y, const float* save_mean, const float* save_rstd,
                                   const float* scale, const float* shift, float* partial, float* coef, float* dgamma, float* dbeta,
                                   int32_t accumulate, void* dy, void* dres, int64_t pixels, int32_t C, int32_t act, float slope,
                                   int32_t dtype, int32_t phases, void* stream);
/* The same backward when the reduce pass has already been done elsewhere: `partial` = rows x row_channels x (sum g, sum g * xhat) written by
 * the data gradient that produced dz (rd_conv_fwd_fused with rd_conv_fusion.bn_y: the statistics rows of that launch, rd_conv_stats_rows of
 * ITS descriptor, row_channels = its Cout >= C).  Finalize + apply only: dz and y are read once. */
int rd_bn_act_bwd_from_partial(const void* dz, const void* y, const float* mean, const float* rstd, const float* scale, const float* shift,
                               const float* partial, int32_t rows, int32_t row_channels, float* coef, float* dgamma, float* dbeta, int32_t accumulate,
                               void* dy, int64_t pixels, int32_t C, int32_t act, float slope, int32_t dtype, void* stream);
/* ---- Decoder head: the last decoder convolution's BatchNorm + activation fused with the one-channel 3x3 output convolution.
 * Replaces, for RCNet/networks.py:773-779 (MultiScaleDecoder.forward: deconv0 -> output0; output0 = net_utils.Conv2d(16 -> 1, 3x3, bias=False,
 * no BatchNorm, linear: utils/net_utils.py:50-91), the chain BatchNorm2d apply + LeakyReLU -> conv2d and its autograd backward (conv2d data +
 * weight gradient, LeakyReLU / BatchNorm2d backward): the activated tensor and its gradient (184 MB each at 5.76 M pixels) are recomputed
 * from the raw convolution output y instead of stored.
 *   y (N,H,W,C) = the producer convolution's raw output; scale / shift / mean / rstd = rd_bn_finalize's outputs for it; w_head fp32 [1][C][3][3]
 *   (rounded to the activation type inside, as the packed operand of rd_conv_fwd is); logits / dlogits (N,H,W,1) in the activation type.
 * rd_bn_head_ok: 1 when the route handles the shape (C == 16).  Backward = rd_bn_head_bwd_reduce (partial: rd_bn_head_rows x 88 x 2 floats)
 * followed by rd_bn_head_bwd_apply (coef: 2 x C floats of scratch; writes the BatchNorm parameter gradients, the head's weight gradient
 * dw_head fp32 [1][C][3][3] and dy = the gradient w.r.t. y; *_accumulate: add to what the gradient buffers hold). */
int32_t rd_bn_head_ok(int32_t N, int32_t H, int32_t W, int32_t C, int32_t dtype);
int32_t rd_bn_head_rows(int32_t N, int32_t H, int32_t W);
int rd_bn_head_fwd(const void* y, const float* scale, const float* shift, int32_t act, float slope, const float* w_head, void* logits, int32_t N,
                   int32_t H, int32_t W, int32_t C, int32_t dtype, void* stream);
int rd_bn_head_bwd_reduce(const void* dlogits, const void* y, const float* mean, const float* rstd, const float* scale, const float* shift,
                          int32_t act, float slope, const float* w_head, float* partial, int32_t N, int32_t H, int32_t W, int32_t C, int32_t dtype,
                          void* stream);
int rd_bn_head_bwd_apply(const void* dlogits, const void* y, const float* mean, const float* rstd, const float* scale, const float* shift,
                         int32_t act, float slope, const float* w_head, const float* partial, int32_t rows, float* coef, float* dgamma, float* dbeta,
                         int32_t bn_accumulate, float* dw_head, int32_t w_accumulate, void* dy, int32_t N, int32_t H, int32_t W, int32_t C,
                         int32_t dtype, void* stream);
/* instantiation name (as rd_conv_fwd_kernel_name) of which = 0: forward, 1: backward reduce, 2: backward apply */
const char* rd_bn_head_kernel_name(int32_t which, int32_t dtype, int32_t act);
/* instantiation name (as rd_conv_fwd_kernel_name) of which = 0: rd_affine_act (flag = residual given), 1: the BatchNorm-backward reduce
   pass, 2: its apply pass (flag = recompute form) for this channel count / dtype / activation */
const char* rd_bn_kernel_name(int32_t which, int32_t C, int32_t dtype, int32_t act, int32_t flag);
int rd_act_bwd(const void* dz, const void* z, void* dx, int64_t n, int32_t act, float slope, int32_t dtype, void* stream);
/* bias gradient: out[c] (+)= sum over rows of x[rows][C] */
int32_t rd_colsum_rows(int64_t rows, int32_t C);
int rd_colsum(const void* x, float* partial, float* out, int32_t accumulate, int64_t rows, int32_t C, int32_t dtype,
              void* stream);
/* the same in two steps for MANY layers: rd_colsum_partial writes a layer's partial rows ([rd_colsum_rows][C][2] floats); one
 * rd_colsum_finalize_batch launch then finishes every pending bias gradient (items is a HOST array, passed by value to the kernel) with
 * the summation order of rd_colsum.  Two items of one batch must not name the same `out`. */
typedef struct rd_colsum_item { const float* partial; float* out; int32_t rows, C, accumulate, reserved; } rd_colsum_item;
int rd_colsum_partial(const void* x, float* partial, int64_t rows, int32_t C, int32_t dtype, void* stream);
int rd_colsum_finalize_batch(const rd_colsum_item* items, int32_t n, void* stream);

/* ---- LayerNorm -- RCNet/linear_attention.py:106-107,125,131-133 (norm1/norm2, x + message) -------- */
int rd_layernorm_fwd(const void* x, const float* gamma, const float* beta, const void* residual, void* out,
                     float* save_mean, float* save_rstd, int64_t rows, int32_t C, float eps, int32_t dtype, void* stream);
int32_t rd_layernorm_bwd_rows(int64_t rows);
int rd_layernorm_bwd(const void* dout, const void* x, const float* gamma, const float* save_mean, const float* save_rstd,
                     void* dx, float* partial, float* dgamma, float* dbeta, int32_t accumulate, int64_t rows, int32_t C,
                     int32_t dtype, void* stream);

/* ---- linear attention -- RCNet/linear_attention.py:18-45 ------------------------------------------
 * q [N*L][ldq], k/v [N*S][ldk/ldv], out [N*L][ldo]; head h uses columns h*16 .. h*16+15 (D = 16), L,S <= 32 */
int rd_linear_attention_fwd(const void* q, const void* k, const void* v, void* out, int32_t N, int32_t L, int32_t S,
                            int32_t H, int32_t ldq, int32_t ldk, int32_t ldv, int32_t ldo, float eps, int32_t dtype,
                            void* stream);
int rd_linear_attention_bwd(const void* q, const void* k, const void* v, const void* dout, void* dq, void* dk, void* dv,
                            int32_t N, int32_t L, int32_t S, int32_t H, int32_t ldq, int32_t ldk, int32_t ldv,
                            int32_t ldo, float eps, int32_t dtype, void* stream);

/* ---- pooling -- RCNet/networks.py:73-76 (MaxPool2d) and :418-433 (torchvision.ops.roi_pool) -------- */
int rd_maxpool_fwd(const void* x, void* out, uint8_t* argmax, int32_t N, int32_t H, int32_t W, int32_t C, int32_t OH,
                   int32_t OW, int32_t k, int32_t s, int32_t p, int32_t dtype, void* stream);
int rd_maxpool_bwd(const void* dout, const uint8_t* argmax, void* dx, int32_t N, int32_t H, int32_t W, int32_t C,
                   int32_t OH, int32_t OW, int32_t k, int32_t s, int32_t p, int32_t dtype, void* stream);
/* rois [R][5] fp32 = (batch index, x1, y1, x2, y2); out [R][PH][PW][C]; argmax int32 = h*W + w or -1 */
int rd_roi_pool_fwd(const void* x, const float* rois, void* out, int32_t* argmax, int32_t R, int32_t N, int32_t H,
                    int32_t W, int32_t C, int32_t PH, int32_t PW, float spatial_scale, int32_t dtype, void* stream);
/* dx_f32 [N][H][W][C] fp32 is zeroed then scatter-added */
int rd_roi_pool_bwd(const void* dout, const float* rois, const int32_t* argmax, float* dx_f32, int32_t R, int32_t N,
                    int32_t H, int32_t W, int32_t C, int32_t PH, int32_t PW, int32_t dtype, void* stream);

/* gather form: dx [N][H][W][C] in the activation dtype, every element written once (no atomics, fixed summation order: ascending roi,
   then bin); needs C % (16 / sizeof(element)) == 0 and the forward's spatial_scale */
/* tile-accumulate form (default when C % 32 == 0): fp32 LDS accumulators per 16 x 16 pixel x 32 channel tile, every gradient element
   written once in the activation dtype; no global atomics */
int rd_roi_pool_bwd_tile(const void* dout, const float* rois, const int32_t* argmax, void* dx, int32_t R, int32_t N, int32_t H,
                         int32_t W, int32_t C, int32_t PH, int32_t PW, float spatial_scale, int32_t dtype, void* stream);
int rd_roi_pool_bwd_gather(const void* dout, const float* rois, const int32_t* argmax, void* dx, int32_t R, int32_t N, int32_t H,
                           int32_t W, int32_t C, int32_t PH, int32_t PW, float spatial_scale, int32_t dtype, void* stream);

/* Compact arg-max (round 4): ONE byte per pooled element instead of the int32 pixel index -- the arg-max's offset inside its bin window,
   (h - hs) << 4 | (w - ws), 0xFF for an empty bin (RC-Net's bins are ~1.02 pixels; the int32 was 4 of the 6 bytes per element either pass
   moves).  C must be a multiple of the 16-byte vector.  A bin window wider or taller than 15 pixels cannot be encoded: the forward then
   raises *overflow_flag (device int32, zero-initialised by the caller, sticky) and both backward forms write NaN gradients.
   rd_roi_pool_bwd_u8 = the fp32 scatter form (rd_roi_pool_bwd), rd_roi_pool_bwd_gather_u8 = the pixel-owner gather (rd_roi_pool_bwd_gather). */
int rd_roi_pool_fwd_u8(const void* x, const float* rois, void* out, uint8_t* argmax, int32_t* overflow_flag, int32_t R, int32_t N, int32_t H,
                       int32_t W, int32_t C, int32_t PH, int32_t PW, float spatial_scale, int32_t dtype, void* stream);
int rd_roi_pool_bwd_u8(const void* dout, const float* rois, const uint8_t* argmax, const int32_t* overflow_flag, float* dx_f32, int32_t R,
                       int32_t N, int32_t H, int32_t W, int32_t C, int32_t PH, int32_t PW, float spatial_scale, int32_t dtype, void* stream);
int rd_roi_pool_bwd_gather_u8(const void* dout, const float* rois, const uint8_t* argmax, const int32_t* overflow_flag, void* dx, int32_t R,
                              int32_t N, int32_t H, int32_t W, int32_t C, int32_t PH, int32_t PW, float spatial_scale, int32_t dtype,
                              void* stream);

/* ---- layout / resampling helpers ------------------------------------------------------------------------- */
int rd_cast(const void* src, void* dst, int64_t n, int32_t src_dtype, int32_t dst_dtype, float scale, void* stream);
int rd_add(const void* a, const void* b, void* out, int64_t n, int32_t dtype, void* stream);
int rd_nchw_to_nhwc(const void* src, void* dst, int32_t N, int32_t C, int32_t H, int32_t W, int32_t src_dtype,
                    int32_t dst_dtype, float scale, void* stream);
int rd_nhwc_to_nchw(const void* src, void* dst, int32_t N, int32_t C, int32_t H, int32_t W, int32_t src_dtype,
                    int32_t dst_dtype, void* stream);
int rd_transpose_last2(const void* src, void* dst, int64_t B, int32_t R, int32_t Ccols, int32_t dtype, void* stream);
int rd_concat2(const void* a, const void* b, void* out, int64_t rows, int32_t Ca, int32_t Cb, int32_t dtype, void* stream);
int rd_split2(const void* in, void* a, void* b, int64_t rows, int32_t Ca, int32_t Cb, int32_t dtype, void* stream);
/* utils/net_utils.py:196 F.interpolate(nearest): standalone forward and its backward (sum over replicas) */
int rd_upsample_nearest_fwd(const void* x, void* y, int32_t N, int32_t Hs, int32_t Ws, int32_t Hv, int32_t Wv, int32_t C,
                            int32_t dtype, void* stream);
int rd_upsample_nearest_bwd(const void* dy, void* dx, int32_t N, int32_t Hs, int32_t Ws, int32_t Hv, int32_t Wv, int32_t C,
                            int32_t dtype, void* stream);

/* ---- RC-Net labels / loss / inference scatter ---------------------------------------------------------- */
/* RCNet/rcnet_main.py:308-332 */
int rd_rcnet_labels(const float* gt, const float* points /* [R][3] */, float* label, float* valid, int32_t R, int32_t HW,
                    float max_distance, int32_t all_valid, void* stream);
/* RCNet/rcnet_model.py:152-160; sums[2] = (sum valid*bce, sum valid) is saved for the backward */
int32_t rd_bce_rows(int64_t n);
int rd_bce_masked_fwd(const void* logits, const float* label, const float* valid, float pos_weight, float* partial,
                      float* loss, float* sums, int64_t n, int32_t dtype, void* stream);
int rd_bce_masked_bwd(const void* logits, const float* label, const float* valid, float pos_weight, const float* sums,
                      const float* dloss, void* dlogits, int64_t n, int32_t dtype, void* stream);
int rd_sigmoid(const void* x, void* y, int64_t n, int32_t dtype, void* stream);
/* RCNet/rcnet_main.py:460-485 (forward_output) */
int rd_scatter_crops(const void* crops, const float* points, float* depth, float* response, int32_t Ncrop, int32_t PH,
                     int32_t PW, int32_t H, int32_t W, float response_thr, int32_t dtype, void* stream);

/* RCNet/run_rcnet_zju.py:221-234: radar points (N,3) -> padded-image coordinates (+pad) and RoI rows (batch, x-pad_x, y-pad_y, x+pad_x, y+pad_y) */
int rd_points_to_rois(const float* points_in, float* points_out, float* rois, int32_t N, float pad_x, float pad_y, int32_t batch_index,
                      void* stream);
/* torchvision convert_boxes_to_roi_format as used by roi_pool at RCNet/networks.py:418-433: boxes (B,K,4) -> rois (B*K,5), image-major;
   first_image offsets the batch index (per-image box lists are converted one image at a time) */
int rd_boxes_to_rois(const float* boxes, float* rois, int32_t B, int32_t K, int32_t first_image, void* stream);
/* data/data_utils.py:128-143 save_depth: uint16 = clamp(trunc(z * multiplier), 0, 65535) (what PIL stores for np.uint32(z * multiplier)) */
int rd_depth_quantize_u16(const float* z, uint16_t* out, int64_t n, float multiplier, void* stream);
/* RCNet/run_rcnet_zju.py:253 np.sum(output_depth) == 0 test of the threshold-retry loop: double-precision sum of n floats */
int rd_sum_f32(const float* x, int64_t n, double* out, void* stream);

/* ---- batch augmentation on the device -- RCNet/rcnet_transforms.py:58-240 as configured by train_rcnet_zju.py:52-59 ---------------------
   params: [B][8] floats = (do_brightness, factor, do_contrast, factor, do_saturation, factor, do_hflip, do_vflip), drawn by the host in the
   reference's order.  image (B,3,H,W) float 0..255.  Step 1 sums the gray values the contrast blend needs ([B][32] int64 partials, exact). */
int rd_augment_gray_partials(const float* image, int32_t B, int32_t H, int32_t W, const float* params, int64_t* partial, void* stream);
/* brightness -> contrast -> saturation (torchvision _blend on the int image) -> v * scale + shift (normalize_images :243-272) -> hflip -> vflip;
   out (B,H,W,3) in `dtype` */
int rd_augment_image(const float* image, int32_t B, int32_t H, int32_t W, const float* params, const int64_t* partial, void* out_nhwc,
                     int32_t dtype, float scale, float shift, void* stream);
/* :163-197: horizontal flip of the ground-truth crops (B,K,1,ph,pw) (out of place) and of the boxes (B,K,4) x1' = n_width - x2 (in place,
   may be NULL); radar points are left as they are, as the reference does */
int rd_augment_flip_labels(const float* labels_in, float* labels_out, int32_t B, int32_t K, int32_t ph, int32_t pw, float* boxes,
                           const float* params, float n_width, void* stream);
/* :199-217: the vertical flip (params column 7) of image and crops rides in rd_augment_image / rd_augment_flip_labels; the reference's box
   update for a vertically flipped sample is boxes[b][1][:] = n_height - boxes[b][3][:], boxes[b][3][:] = n_height - old boxes[b][1][:]
   (rows 1 and 3 of the sample, not y1 / y2 of each box -- kept as written), in place, after the horizontal update; K >= 4 */
int rd_augment_vflip_boxes(float* boxes, int32_t B, int32_t K, const float* params, float n_height, void* stream);
/* data/datasets.py:254-272: ground-truth crops around the (padded-coordinate) radar points from the zero-padded dense map (B,1,Hp,Wp) */
int rd_crop_patches(const float* gt_padded, const float* points, float* crops, int32_t B, int32_t K, int32_t Hp, int32_t Wp, int32_t ph,
                    int32_t pw, void* stream);

/* ---- offline projection + scatter -- data/preprocess/project_transform.py:67-97, pointcloud_project_zju.py:57-76,81-103 ------------------
   points (n, stride >= 3) float32 xyz; t_camera_pcl, projection: 4x4 row-major float64 (device); depth_map (H,W) = max(depth, 1) of the
   NEAREST point projecting to each pixel, 0 elsewhere (bit-exact, order independent); kept_points (n,3) = (u, v, depth) of the points
   that pass canvas_crop and the depth range, compacted in arbitrary order, n_kept their count (both may be NULL) */
int rd_project_scatter(const float* points, int32_t n, int32_t stride, const double* t_camera_pcl, const double* projection, int32_t H, int32_t W,
                       double min_depth, double max_depth, float* depth_map, float* kept_points, int32_t* n_kept, void* stream);

/* ---- lidar interpolation -- data/data_utils.py:231-275 interpolate_depth, :333-367 interpolate_depth_delft (scipy LinearNDInterpolator) ----
   barycentric evaluation of a Delaunay triangulation at every pixel: simplices (M,3) indices into the data points, whose integer pixel
   coordinates are point_row / point_col and whose (possibly log) depths are `values` (float64, as scipy computes); out (H,W) float64 =
   interpolated value inside the convex hull, fill_value outside; owner_workspace (H,W) int32.  The triangulation itself (Qhull) is the
   caller's, as in the reference. */
int rd_tri_raster(const int32_t* simplices, const int32_t* point_row, const int32_t* point_col, const double* values, int32_t n_simplices,
                  int32_t H, int32_t W, double fill_value, int32_t* owner_workspace, double* out, void* stream);

/* modules/interpolator.py:7-18 interpolate_knots with interpolate = 'nearest' (scipy griddata -> NearestNDInterpolator): out (H,W) float64 =
 * value of the closest of n_points knots given as integer (row, col); equidistant knots: the lowest index; no knots: fill_value.  The
 * 'linear' method of the same function is rd_tri_raster with fill_value 1.0. */
int rd_nearest_knot(const int32_t* point_row, const int32_t* point_col, const double* values, int32_t n_points, int32_t H, int32_t W,
                    double fill_value, double* out, void* stream);
/* HOST helper (no GPU work): reverse the per-scanline PNG filters (None / Sub / Up / Average / Paeth) of an inflated 8- or 16-bit grayscale
 * image: raw = h rows of (1 filter byte + row_bytes), bpp = bytes per pixel, out = h * row_bytes.  The depth-map reader of
 * data/data_utils.py:94-125 (PIL there) uses it for the rows whose recurrences do not vectorise. */
int rd_png_unfilter_host(const uint8_t* raw, int32_t h, int32_t row_bytes, int32_t bpp, uint8_t* out);

/* ---- optimizer -- RCNet/rcnet_main.py:233-238,357-359; train_zju.py:205-211,390-392 ------------------------ */
int rd_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1,
                 float beta2, float eps, float weight_decay, int64_t step, float grad_scale, void* stream);
/* fp16 mode with a static loss scale (the reference trains in fp32; BASELINE configs[4] asks for fp16): an inf / NaN in the scaled gradients
 * must not reach the moments.  rd_grad_finite_check makes flag[0] non-zero (device int32[2]) if any of the n gradients is not finite;
 * rd_adam_step_guarded is rd_adam_step that returns at once while flag[0] is set (parameters and moments untouched);
 * rd_adam_skip_count, called once after the step's Adam launches, moves a raised flag into the skipped-steps counter flag[1]. */
int rd_grad_finite_check(const float* grad, int64_t n, int32_t* flag, void* stream);
int rd_adam_step_guarded(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1,
                         float beta2, float eps, float weight_decay, int64_t step, float grad_scale, const int32_t* skip_flag, void* stream);
int rd_adam_skip_count(int32_t* flag, void* stream);

/* ---- routing options: kernel-selection switches for tests and A/B tools (the product path reads no environment variable).  Names and
 * ranges: kOpts in csrc/rd_api.cpp (e.g. "conv3x3_min_blocks", "frag_v128", "frag32_v128", "wgrad_tr_tw"); values are clamped; an unknown
 * name returns -1.  rd_clear_option(NULL) / rd_clear_options() restore every default. */
int rd_set_option(const char* name, int32_t value);
int rd_clear_option(const char* name);
int rd_clear_options(void);

/* ---- gradient exchange over RCCL / xGMI -- replaces RCNet/rcnet_model.py:259-265 (torch.nn.DataParallel) -------------------------
 * One process per GPU.  rd_comm_unique_id: rank 0 draws the 128-byte rendezvous id (ncclGetUniqueId), the caller hands it to the other
 * ranks by any host channel; rd_comm_init: ncclCommInitRank on the CURRENT device + a library-owned communication stream.  RCCL is bound at
 * run time (the copy already in the process, $RIDERS_RCCL_LIB, /opt/rocm/lib/librccl.so): no link-time dependency.
 * rd_allreduce_bucket: in-place SUM of buf[0..n) (fp32) over the ranks, enqueued on the communication stream behind everything queued so
 * far on `compute_stream` -- it returns at once and overlaps whatever the caller enqueues next; mode 0 = all-reduce (RCCL picks the
 * algorithm), 1 = reduce-scatter + all-gather in place.  rd_comm_broadcast: buf of rank `root` to every rank, same ordering.
 * rd_comm_join: `compute_stream` waits for everything issued since the last join.  All four are stream operations and may be captured
 * into a hipGraph (fork / join of the communication stream become graph edges).  rd_comm_pending: collectives issued and not yet joined.
 * rd_comm_available: 0 when an RCCL build can be bound (a copy already mapped into the process is the only candidate then -- two RCCL builds
 * are never mixed); every rank calls it and the ranks agree on the result BEFORE rd_comm_init, which blocks until all of them have entered. */
int rd_comm_available(void);
int rd_comm_unique_id(void* id128);
int rd_comm_init(int32_t rank, int32_t world, const void* id128, void** comm);
int rd_comm_destroy(void* comm);
int rd_allreduce_bucket(void* comm, float* buf, int64_t n, int32_t mode, void* compute_stream);
int rd_comm_broadcast(void* comm, float* buf, int64_t n, int32_t root, void* compute_stream);
int rd_comm_join(void* comm, void* compute_stream);
int64_t rd_comm_pending(void* comm);

/* ==== Scale Map Learner (MiDaS-small) ================================================================================ */
/* depthwise convolution of the tf_efficientnet_lite3 backbone (modules/midas/blocks.py:44-64; torch.hub, third-party):
 * w = OIHW fp32 with I = 1; p = TF-"SAME" leading pad; partial [rd_dw_rows][C][k*k] */
int32_t rd_dw_rows(int64_t pixels, int32_t C);
int rd_dwconv_fwd(const void* x, const float* w, void* y, int32_t N, int32_t H, int32_t W, int32_t C, int32_t OH, int32_t OW, int32_t k,
                  int32_t s, int32_t p, int32_t dtype, void* stream);
/* forward with the following BatchNorm's batch statistics fused into the epilogue: stats [rd_dwconv_stats_rows][C][2] = (sum, sum^2) of
 * the stored outputs, ready for rd_bn_finalize; rd_dwconv_stats_rows is 0 when the shape has no fused form (then rd_dwconv_fwd + rd_bn_stats) */
int32_t rd_dwconv_stats_rows(int32_t N, int32_t OH, int32_t OW, int32_t C, int32_t k, int32_t s);
int rd_dwconv_fwd_stats(const void* x, const float* w, void* y, float* stats, int32_t N, int32_t H, int32_t W, int32_t C, int32_t OH,
                        int32_t OW, int32_t k, int32_t s, int32_t p, int32_t dtype, void* stream);
int rd_dwconv_dgrad(const void* dy, const float* w, void* dx, int32_t N, int32_t H, int32_t W, int32_t C, int32_t OH, int32_t OW, int32_t k,
                    int32_t s, int32_t p, int32_t dtype, void* stream);
int rd_dwconv_wgrad(const void* x, const void* dy, float* partial, float* dw, int32_t accumulate, int32_t N, int32_t H, int32_t W,
                    int32_t C, int32_t OH, int32_t OW, int32_t k, int32_t s, int32_t p, int32_t dtype, void* stream);
/* the same in two steps for MANY layers (EfficientNet-Lite3 has 27 depthwise layers): rd_dwconv_wgrad_partial leaves the layer's partial
 * rows and fills *item (item->rows = 0: the gradient was finished at once -- channel counts off the vector path); one
 * rd_dw_wgrad_finalize_batch launch then finishes every pending item (HOST array, passed by value) with the summation order of
 * rd_dwconv_wgrad.  Two items of one batch must not name the same dw. */
typedef struct rd_dw_wgrad_item { const float* partial; float* dw; int32_t rows, C, KK, accumulate; } rd_dw_wgrad_item;
int rd_dwconv_wgrad_partial(const void* x, const void* dy, float* partial, float* dw, int32_t accumulate, int32_t N, int32_t H, int32_t W,
                            int32_t C, int32_t OH, int32_t OW, int32_t k, int32_t s, int32_t p, int32_t dtype, rd_dw_wgrad_item* item,
                            void* stream);
int rd_dw_wgrad_finalize_batch(const rd_dw_wgrad_item* items, int32_t n, void* stream);
/* BatchNorm (sum, sum^2) partials [rd_dw_rows][C][2] for producers without a fused statistics epilogue */
int rd_bn_stats(const void* y, float* partial, int64_t pixels, int32_t C, int32_t dtype, void* stream);
/* modules/midas/blocks.py:168-170 (align_corners=1) and :187 nn.Upsample (align_corners=0) */
int rd_bilinear_fwd(const void* x, void* y, int32_t N, int32_t H, int32_t W, int32_t C, int32_t OH, int32_t OW, int32_t align_corners,
                    int32_t dtype, void* stream);
int rd_bilinear_bwd(const void* dy, void* dx, int32_t N, int32_t H, int32_t W, int32_t C, int32_t OH, int32_t OW, int32_t align_corners,
                    int32_t dtype, void* stream);
/* modules/midas/midas_net_custom.py:121-133: pred = d*relu(1+out), clamps (hi = 1/min_pred, lo = 1/max_pred, <= 0 disables) */
int rd_sml_head_fwd(const void* out, const float* d, float* pred, int64_t n, float hi, float lo, int32_t dtype, void* stream);
int rd_sml_head_bwd(const void* out, const float* d, const float* dpred, void* dout, int64_t n, float hi, float lo, int32_t dtype,
                    void* stream);
/* train_zju.py:355-356: out = 1/x (dy == NULL) or dx = -dy/x^2 */
int rd_reciprocal(const float* x, const float* dy, float* out, int64_t n, void* stream);
/* train_zju.py:246-343 + modules/estimator.py:146-176: per-sample bounded L1 scale fit of mono inverse depth to radar */
int rd_sml_scale_align(const float* mono, const float* sparse_depth, int32_t B, int32_t HW, float min_depth, float max_depth, float lo,
                       float hi, float* scale, int32_t* nvalid, void* stream);
/* train_zju.py:278-287 ('st') + modules/estimator.py:5-29,90-118: per-sample closed-form least-squares scale AND shift */
int rd_sml_scale_shift_ls(const float* mono, const float* sparse_depth, int32_t B, int32_t HW, float min_depth, float max_depth, float* scale,
                          float* shift, int32_t* nvalid, void* stream);
/* int_depth = clamp(scale * mono + shift) (shift may be NULL: 's' alignment) / int_scales / min-max normalise / nearest resize /
   mean-std normalise / gray -> x (B,h,w,3), d (B,h,w); mm [B][3] scratch.  train_zju.py:289-337 */
int rd_sml_build_inputs(const float* image_nchw, const float* mono, const float* sparse_depth, const float* rcnet_depth, const float* scale,
                        const float* shift, float* mm, int32_t B, int32_t H, int32_t W, int32_t h, int32_t w, float min_depth,
                        float max_depth, float clamp_hi, float clamp_lo, int32_t use_rcnet, float mean_depth, float std_depth,
                        float mean_scales, float std_scales, float* x, float* d, void* stream);
/* utils/net_utils.py:591-638 */
int32_t rd_outlier_parts(int64_t n);
int rd_outlier_removal(const float* depth, float* partial, float* out, int32_t N, int32_t H, int32_t W, int32_t kernel_size, float threshold,
                       void* stream);
/* utils/loss.py:5-135,187-274 ('l1'); info[7] = loss, supervised, lidar, smoothness, edge, n_interp, n_lidar; partial = doubles [rows][8] */
int32_t rd_sml_loss_rows(int64_t n);
int rd_sml_loss_fwd(const float* pred, const float* image, const float* gt_interp, const float* gt_sparse, const float* weights, int32_t N,
                    int32_t H, int32_t W, int32_t filter_size, float w_lidar, float w_smooth, float w_edge, float* gfx, float* gfy,
                    double* partial, float* info, void* stream);
int rd_sml_loss_bwd(const float* pred, const float* gt_interp, const float* gt_sparse, const float* gfx, const float* gfy, const float* info,
                    const float* dloss, int32_t N, int32_t H, int32_t W, int32_t filter_size, float w_lidar, float w_smooth, float* dpred,
                    void* stream);
/* the same for the reference's other supervised terms (utils/loss.py:55-100): loss_kind 0 = 'l1', 1 = 'l2' (mse), 2 = 'smoothl1' (beta 1); with
   w_edge > 0 (needs w_smooth > 0) the saved gradient fields gfx / gfy also carry the edge-matching term (utils/loss.py:241-249), so the backward
   below differentiates it too.  rd_sml_loss_fwd / _bwd are loss_kind 0. */
int rd_sml_loss_fwd_kind(const float* pred, const float* image, const float* gt_interp, const float* gt_sparse, const float* weights, int32_t N,
                         int32_t H, int32_t W, int32_t filter_size, int32_t loss_kind, float w_lidar, float w_smooth, float w_edge, float* gfx,
                         float* gfy, double* partial, float* info, void* stream);
int rd_sml_loss_bwd_kind(const float* pred, const float* gt_interp, const float* gt_sparse, const float* gfx, const float* gfy, const float* info,
                         const float* dloss, int32_t N, int32_t H, int32_t W, int32_t filter_size, int32_t loss_kind, float w_lidar, float w_smooth,
                         float* dpred, void* stream);
/* val_zju.py:200-206 bicubic (A=-0.75, align_corners=False); :212-231 + utils/eval_utils.py metric sums, res = doubles [N][8] */
int rd_bicubic_resize(const float* x, float* y, int32_t N, int32_t H, int32_t W, int32_t OH, int32_t OW, void* stream);
int rd_depth_metrics(const float* out, const float* gt, int32_t N, int32_t HW, float min_depth, float max_depth, double* res, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RIDERS_HIP_H */
