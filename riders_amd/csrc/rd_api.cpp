// extern "C" surface of libriders_hip.so (see include/riders_hip.h).  Argument validation + launch only.
#include "../../include/riders_hip.h"
#include "rd_kernels.h"
// the fp16 build of the same kernels (compiled with -DRD_HALF_F16, namespace rd_f16): same declarations, second namespace
#define rd rd_f16
#include "rd_kernels_decl.h"
#undef rd
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
// (declared here rather than in rd_kernels_decl.h, which every kernel unit includes)
#define RD_SLAB_DECL(NS) namespace NS { \
  bool bn_slab_ok(int64_t pixels, int C, int dtype); \
  void launch_bn_bwd_slab(const void* dz, const void* y, const float* mean, const float* rstd, const float* scale, const float* shift, float* dgamma, float* dbeta, \
                          int accumulate, void* dy, int64_t pixels, int C, int act, float slope, int dtype, hipStream_t st); \
  void launch_bn_fwd_slab(const float* stats, int rows, const void* y, const float* gamma, const float* beta, float eps, float momentum, float* running_mean, \
                          float* running_var, float* mean, float* rstd, float* scale, float* shift, void* out, int64_t pixels, int C, int act, float slope, int dtype, \
                          hipStream_t st); \
  const char* bn_slab_kernel_name(int which, int64_t pixels, int dtype, int act); }
RD_SLAB_DECL(rd)
RD_SLAB_DECL(rd_f16)
#undef RD_SLAB_DECL
namespace rd { void launch_augment_vflip_boxes(float* boxes, int B, int K, const float* params, float n_height, hipStream_t st); }
namespace rd { void launch_pack_weights_batch_map(const void* items, int n, const void* map, int blocks, hipStream_t st); }
namespace rd_f16 { void launch_pack_weights_batch_map(const void* items, int n, const void* map, int blocks, hipStream_t st); }

namespace {
thread_local char g_err[512] = "";
int fail(const char* fmt, ...) {
  va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap);
  return -1;
}
int done(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e)); return (int)e; }
  return 0;
}
inline hipStream_t S(void* s) { return (hipStream_t)s; }
inline bool dt_ok(int d) { return d == RD_F32 || d == RD_BF16 || d == RD_F16; }
// Precision dispatch: RD_F16 selects the fp16 build of the kernels; inside a build the 16-bit activation type always has code 1.
#define RD_DT(dt) ((dt) == RD_F16 ? 1 : (dt))
#define RD_NS(dt, fn) ((dt) == RD_F16 ? rd_f16::fn : rd::fn)

int check_desc(const rd_conv_desc* d) {
  if (!d) return fail("conv: null descriptor");
  if (!dt_ok(d->dtype)) return fail("conv: bad dtype %d", d->dtype);
  if (d->N <= 0 || d->Hin <= 0 || d->Win <= 0 || d->C1 <= 0 || d->C2 < 0 || d->Cout <= 0) return fail("conv: bad sizes");
  if (d->KH <= 0 || d->KW <= 0 || d->stride <= 0 || d->pad < 0 || d->in_dilate <= 0) return fail("conv: bad kernel geometry");
  if (d->OH <= 0 || d->OW <= 0) return fail("conv: bad output size");
  if (d->upsample && (d->H1 <= 0 || d->W1 <= 0)) return fail("conv: upsample needs H1/W1");
  if (d->D1 <= 0 || d->D1 > d->Cout) return fail("conv: bad D1");
  if (d->out_d2s && (d->Cout != 4 * d->D1 || d->out_reduce2 || d->upsample)) return fail("conv: out_d2s needs Cout = 4 x D1, no upsample flag, no out_reduce2");
  if (d->in_s2d && ((d->C1 & 3) || d->C2 || d->upsample || d->out_d2s || d->out_reduce2)) return fail("conv: in_s2d needs C1 = 4 x channels, one source, no upsample / out_d2s / out_reduce2");
  if ((int64_t)d->N * d->OH * d->OW >= (int64_t)1 << 31) return fail("conv: too many output pixels");
  return 0;
}
void fill_args(const rd_conv_desc* d, rd::ConvArgs& a) {
  memset(&a, 0, sizeof(a));
  a.N = d->N; a.Hin = d->Hin; a.Win = d->Win; a.C1 = d->C1; a.C2 = d->C2;
  a.ups = d->upsample ? 1 : 0;
  a.H1 = a.ups ? d->H1 : d->Hin; a.W1 = a.ups ? d->W1 : d->Win;
  a.Cout = d->Cout; a.KH = d->KH; a.KW = d->KW; a.stride = d->stride; a.pad = d->pad; a.dil = d->in_dilate;
  a.OH = d->OH; a.OW = d->OW; a.act = d->act; a.slope = d->slope; a.D1 = d->D1;
  // ATen nearest: scale = (float)in / out
  a.scale_h = (float)a.H1 / (float)d->Hin; a.scale_w = (float)a.W1 / (float)d->Win;
  a.M = d->N * d->OH * d->OW;
  a.K = d->KH * d->KW * (d->C1 + d->C2);
  a.Kpad = RD_NS(d->dtype, conv_kpad)(a.K, RD_DT(d->dtype));
  a.pool2 = d->out_reduce2 ? 1 : 0;
  a.d2s = d->out_d2s ? 1 : 0;
  a.s2d = d->in_s2d ? 1 : 0;
}
}  // namespace

namespace {
struct OptDef { const char* name; int lo, hi; };
// index = rdt::RdOpt
const OptDef kOpts[rdt::OPT_COUNT] = {
  {"conv_par", 0, 1}, {"conv3x3_min_blocks", 0, 1 << 30}, {"conv_stem_min_m", 0, 1 << 30}, {"wgrad_blocks", 8, 1 << 16}, {"wgrad_tiny_min_m", 0, 1 << 30},
  {"conv3x3_w8", 0, 1}, {"patch_bn_max", 32, 128}, {"conv3x3_g8", 1, 4096}, {"conv1x1_min_m", 0, 1 << 30}, {"conv_few_min_m", 0, 1 << 30},
  {"frag_v128", 0, 6}, {"frag_v64", 0, 6}, {"frag_v32", 0, 6}, {"frag_split", 0, 1}, {"frag_split_blocks", 0, 1 << 30},
  {"frag32_v128", 0, 3}, {"frag32_v64", 0, 3}, {"frag_lin", 0, 1}, {"conv3x3_frag", 0, 1}, {"bn_gen_ppt", 2, 64}, {"bn_vec_per", 0, 64},
  {"wgrad_tr_tw", 8, 32}, {"frag_db", 0, 1}, {"wgrad_fit", 0, 1}, {"head_np", 256, 1800}, {"head_cpi", 0, 0x888}, {"pw_min_m", 0, 1 << 30}, {"pw_ks", 0, 8}, {"bn_slab", 0, 2},
};
int g_opt_val[rdt::OPT_COUNT];
bool g_opt_set[rdt::OPT_COUNT];
}  // namespace

extern "C" {

int rd_opt(int id, int dflt) { return (id >= 0 && id < rdt::OPT_COUNT && g_opt_set[id]) ? g_opt_val[id] : dflt; }
int rd_opt_is_set(int id) { return id >= 0 && id < rdt::OPT_COUNT && g_opt_set[id]; }
int rd_set_option(const char* name, int32_t value) {
  if (!name) return fail("set_option: null name");
  for (int i = 0; i < rdt::OPT_COUNT; i++)
    if (!strcmp(name, kOpts[i].name)) {
      g_opt_val[i] = value < kOpts[i].lo ? kOpts[i].lo : (value > kOpts[i].hi ? kOpts[i].hi : value);
      g_opt_set[i] = true;
      return 0;
    }
  return fail("set_option: unknown option '%s'", name);
}
int rd_clear_options(void) { for (int i = 0; i < rdt::OPT_COUNT; i++) g_opt_set[i] = false; return 0; }
int rd_clear_option(const char* name) {
  if (!name) return rd_clear_options();
  for (int i = 0; i < rdt::OPT_COUNT; i++) if (!strcmp(name, kOpts[i].name)) { g_opt_set[i] = false; return 0; }
  return fail("clear_option: unknown option '%s'", name);
}

int rd_version(void) { return 100; }
const char* rd_last_error_string(void) { return g_err; }
void rd_set_last_error(const char* msg) { snprintf(g_err, sizeof(g_err), "%s", msg ? msg : ""); }      // for the library's other host units (rd_comm.cpp)

int64_t rd_conv_packed_elems(int32_t rows, int32_t K, int32_t dtype) {
  return RD_NS(dtype, conv_packed_elems)(rows, K, RD_DT(dtype));
}
int rd_conv_pack_weights(const float* w, void* packed, int32_t Cout, int32_t Cin, int32_t KH, int32_t KW, int32_t mode,
                         int32_t dtype, void* stream) {
  if (!w || !packed) return fail("pack_weights: null pointer");
  if (!dt_ok(dtype) || mode < 0 || mode > 3 || (mode >= 2 && (KH != 3 || KW != 3))) return fail("pack_weights: bad dtype/mode");
  RD_NS(dtype, launch_pack_weights)(w, packed, Cout, Cin, KH, KW, mode, RD_DT(dtype), S(stream), 0);
  return done("rd_conv_pack_weights");
}
int rd_conv_pack_weights_padded(const float* w, void* packed, int32_t Cout, int32_t Cin_src, int32_t Cin, int32_t KH, int32_t KW,
                                int32_t dtype, void* stream) {
  if (!w || !packed) return fail("pack_weights_padded: null pointer");
  if (!dt_ok(dtype) || Cin_src <= 0 || Cin < Cin_src) return fail("pack_weights_padded: bad dtype / channel counts");
  RD_NS(dtype, launch_pack_weights)(w, packed, Cout, Cin, KH, KW, 0, RD_DT(dtype), S(stream), Cin_src);
  return done("rd_conv_pack_weights_padded");
}
int rd_pad_channels(const void* src, void* dst, int64_t rows, int32_t C, int32_t Cpad, int32_t dtype, void* stream) {
  if (!src || !dst) return fail("pad_channels: null pointer");
  if (!dt_ok(dtype) || C <= 0 || Cpad < C || rows < 0) return fail("pad_channels: bad arguments");
  if (rows == 0) return 0;
  RD_NS(dtype, launch_pad_channels)(src, dst, rows, C, Cpad, RD_DT(dtype), S(stream));
  return done("rd_pad_channels");
}
int rd_unpad_weight_grad(const float* dwp, float* dw, int32_t Cout, int32_t Cin, int32_t Cin_pad, int32_t taps, int32_t accumulate,
                         void* stream) {
  if (!dwp || !dw) return fail("unpad_weight_grad: null pointer");
  if (Cout <= 0 || Cin <= 0 || Cin_pad < Cin || taps <= 0) return fail("unpad_weight_grad: bad arguments");
  rd::launch_unpad_weight_grad(dwp, dw, Cout, Cin, Cin_pad, taps, accumulate, S(stream));
  return done("rd_unpad_weight_grad");
}
int rd_conv_pack_weights_batch(const rd_pack_item* items, int32_t n, void* stream) {
  if (n <= 0) return 0;
  if (!items) return fail("pack_weights_batch: null table");
  if (n > 65535) return fail("pack_weights_batch: too many items (%d)", n);
  rd::launch_pack_weights_batch(items, n, S(stream));
  return done("rd_conv_pack_weights_batch");
}
int rd_conv_pack_weights_batch_half(const rd_pack_item* items, int32_t n, int32_t half_dtype, void* stream) {
  if (n <= 0) return 0;
  if (!items) return fail("pack_weights_batch_half: null table");
  if (n > 65535) return fail("pack_weights_batch_half: too many items (%d)", n);
  if (half_dtype != RD_BF16 && half_dtype != RD_F16) return fail("pack_weights_batch_half: half_dtype must be RD_BF16 or RD_F16");
  RD_NS(half_dtype, launch_pack_weights_batch)(items, n, S(stream));
  return done("rd_conv_pack_weights_batch_half");
}
int rd_conv_pack_weights_batch_map(const rd_pack_item* items, int32_t n, int32_t half_dtype, const int32_t* block_map, int32_t blocks, void* stream) {
  if (n <= 0 || blocks <= 0) return 0;
  if (!items || !block_map) return fail("pack_weights_batch_map: null table");
  if (half_dtype != RD_BF16 && half_dtype != RD_F16) return fail("pack_weights_batch_map: half_dtype must be RD_BF16 or RD_F16");
  RD_NS(half_dtype, launch_pack_weights_batch_map)(items, n, block_map, blocks, S(stream));
  return done("rd_conv_pack_weights_batch_map");
}
int rd_linear_wgrad_batch(const rd_lwg_gemm* gemms, int32_t n_gemm, const rd_lwg_reduce* reduces, int32_t n_reduce, int32_t dtype,
                          void* stream) {
  static_assert(sizeof(rd_lwg_gemm) == sizeof(rd::LwgGemm) && sizeof(rd_lwg_reduce) == sizeof(rd::LwgReduce), "ABI mirrors");
  if (n_gemm <= 0) return 0;
  if (!gemms || !reduces || n_reduce <= 0) return fail("linear_wgrad_batch: null table");
  if (!dt_ok(dtype)) return fail("linear_wgrad_batch: bad dtype %d", dtype);
  for (int i = 0; i < n_gemm; i++) {
    const rd_lwg_gemm& g = gemms[i];
    if (!g.x1 || !g.dy || !g.slab || (g.C2 > 0 && !g.x2)) return fail("linear_wgrad_batch: item %d has a null pointer", i);
    const int ve = dtype == RD_F32 ? 4 : 8;
    if (g.M <= 0 || g.C1 <= 0 || g.Cout <= 0 || g.C2 < 0 || (g.C1 % ve) || (g.C2 % ve) || (g.Cout % ve) || (g.C2 > 0 && (g.C1 & 63)))
      return fail("linear_wgrad_batch: item %d: channels must be multiples of %d (and C1 of 64 when C2 > 0)", i, ve);
    if (g.nsplit <= 0 || g.rows_per_split <= 0 || (int64_t)g.nsplit * g.rows_per_split < g.M) return fail("linear_wgrad_batch: item %d: splits do not cover M", i);
  }
  for (int i = 0; i < n_reduce; i++)
    if (!reduces[i].slab || !reduces[i].dw || reduces[i].elems <= 0 || reduces[i].nsplit <= 0) return fail("linear_wgrad_batch: bad reduce item %d", i);
  RD_NS(dtype, launch_linear_wgrad_batch)(reinterpret_cast<const rd::LwgGemm*>(gemms), n_gemm, reinterpret_cast<const rd::LwgReduce*>(reduces), n_reduce, RD_DT(dtype), S(stream));
  return done("rd_linear_wgrad_batch");
}
int rd_loftr_layer_fwd(const void* x, const void* src, const rd_loftr_weights* w, void* out, const rd_loftr_saved* sv, int32_t N,
                       int32_t L, int32_t S, float eps_attn, float eps_ln, int32_t dtype, void* stream) {
  static_assert(sizeof(rd_loftr_weights) == sizeof(rd::LoftrW) && sizeof(rd_loftr_saved) == sizeof(rd::LoftrSaved) &&
                sizeof(rd_loftr_grads) == sizeof(rd::LoftrGrads), "ABI mirrors");
  if (!x || !src || !w || !out || !sv) return fail("loftr_layer_fwd: null pointer");
  if (!dt_ok(dtype)) return fail("loftr_layer_fwd: bad dtype %d", dtype);
  if (N < 0 || L <= 0 || S <= 0 || L > 32 || S > 32) return fail("loftr_layer_fwd: needs 1..32 tokens per sequence (L=%d S=%d)", L, S);
  if (!w->wq || !w->wk || !w->wv || !w->wm || !w->w0 || !w->w2 || !w->g1 || !w->b1 || !w->g2 || !w->b2) return fail("loftr_layer_fwd: null weight");
  if (!sv->q || !sv->k || !sv->v || !sv->att || !sv->mpre || !sv->msg || !sv->hid || !sv->m2pre || !sv->stats) return fail("loftr_layer_fwd: null saved buffer");
  RD_NS(dtype, launch_loftr_layer_fwd)(x, src, *reinterpret_cast<const rd::LoftrW*>(w), out, *reinterpret_cast<const rd::LoftrSaved*>(sv), N, L, S,
                             eps_attn, eps_ln, RD_DT(dtype), (hipStream_t)stream);
  return done("rd_loftr_layer_fwd");
}
int rd_loftr_layer_bwd(const void* x, const void* src, const rd_loftr_weights* w, const rd_loftr_saved* sv, const rd_loftr_grads* g,
                       int32_t N, int32_t L, int32_t S, float eps_attn, int32_t dtype, void* stream) {
  if (!x || !src || !w || !sv || !g) return fail("loftr_layer_bwd: null pointer");
  if (!dt_ok(dtype)) return fail("loftr_layer_bwd: bad dtype %d", dtype);
  if (N < 0 || L <= 0 || S <= 0 || L > 32 || S > 32) return fail("loftr_layer_bwd: needs 1..32 tokens per sequence (L=%d S=%d)", L, S);
  if (!g->dout || !g->dm2pre || !g->dhid || !g->dmpre || !g->datt || !g->dq || !g->dk || !g->dv || !g->dx || (src != x && !g->dsrc))
    return fail("loftr_layer_bwd: null gradient buffer");
  if (g->dsrc_accumulate && src == x) return fail("loftr_layer_bwd: dsrc_accumulate is for cross attention (src != x)");
  if (!g->lnp1 || !g->lnp2 || !g->dg1 || !g->db1 || !g->dg2 || !g->db2) return fail("loftr_layer_bwd: null LayerNorm gradient buffer");
  RD_NS(dtype, launch_loftr_layer_bwd)(x, src, *reinterpret_cast<const rd::LoftrW*>(w), *reinterpret_cast<const rd::LoftrSaved*>(sv),
                             *reinterpret_cast<const rd::LoftrGrads*>(g), N, L, S, eps_attn, RD_DT(dtype), (hipStream_t)stream);
  return done("rd_loftr_layer_bwd");
}
int32_t rd_conv_stats_rows(const rd_conv_desc* d) {
  rd::ConvArgs a; fill_args(d, a);
  return (int32_t)RD_NS(d->dtype, conv_stats_rows)(a, RD_DT(d->dtype));
}

int32_t rd_conv_out_reduce2_ok(const rd_conv_desc* d) {
  if (!d || check_desc(d)) return 0;
  if ((d->OH & 1) || (d->OW & 1) || d->D1 != d->Cout) return 0;
  rd::ConvArgs a; fill_args(d, a);
  a.pool2 = 1;
  return RD_NS(d->dtype, conv_pool2_ok)(a, RD_DT(d->dtype)) ? 1 : 0;
}

int32_t rd_conv_up2_ok(const rd_conv_desc* d) {
  if (!d || check_desc(d) || d->Cout != 4 * d->D1 || d->upsample || d->out_reduce2) return 0;
  rd::ConvArgs a; fill_args(d, a);
  a.d2s = 1;
  return RD_NS(d->dtype, conv_d2s_ok)(a, RD_DT(d->dtype)) ? 1 : 0;
}

int32_t rd_conv_up2_dgrad_ok(const rd_conv_desc* d) {
  if (!d || check_desc(d) || (d->C1 & 3) || d->C2 || d->upsample || d->out_d2s || d->out_reduce2 || d->D1 != d->Cout) return 0;
  rd::ConvArgs a; fill_args(d, a);
  a.s2d = 1;
  return RD_NS(d->dtype, conv_s2d_ok)(a, RD_DT(d->dtype)) ? 1 : 0;
}

int32_t rd_conv_fwd_streams(const rd_conv_desc* d) {
  if (!d || check_desc(d)) return 0;
  rd::ConvArgs a; fill_args(d, a);
  return RD_NS(d->dtype, conv_few_ok)(a) ? 1 : 0;
}

int rd_conv_fwd(const rd_conv_desc* d, const void* src1, const void* src2, const void* w_packed, const float* bias, void* dst1,
                void* dst2, float* stats, void* stream) {
  if (int e = check_desc(d)) return e;
  if (!src1 || !w_packed || !dst1) return fail("conv_fwd: null pointer");
  if (d->C2 > 0 && !src2) return fail("conv_fwd: C2 > 0 but src2 is null");
  if (d->D1 < d->Cout && !dst2 && !d->out_d2s) return fail("conv_fwd: D1 < Cout but dst2 is null");
  rd::ConvArgs a; fill_args(d, a);
  a.src1 = src1; a.src2 = src2; a.w = w_packed; a.bias = bias; a.dst1 = dst1; a.dst2 = dst2; a.stats = stats;
  if (a.s2d && (dst2 || bias || stats || d->act != RD_ACT_NONE || !rd_conv_up2_dgrad_ok(d)))
    return fail("conv_fwd: in_s2d is not available for this descriptor / with a bias, an activation, statistics or a second destination (see rd_conv_up2_dgrad_ok)");
  if (a.d2s && (dst2 || bias || d->act != RD_ACT_NONE || !rd_conv_up2_ok(d)))
    return fail("conv_fwd: out_d2s is not available for this descriptor / with a bias, an activation or a second destination (see rd_conv_up2_ok)");
  if (a.pool2 && (stats || bias || d->act != RD_ACT_NONE || !rd_conv_out_reduce2_ok(d)))      // (a bias would be added once to the 2x2 sum instead of four times)
    return fail("conv_fwd: out_reduce2 is not available for this descriptor / with a bias or an activation (see rd_conv_out_reduce2_ok)");
  RD_NS(d->dtype, launch_conv)(a, RD_DT(d->dtype), S(stream));
  return done("rd_conv_fwd");
}
int32_t rd_conv_add_ok(const rd_conv_desc* d) {
  if (!d || check_desc(d) || d->D1 != d->Cout || d->out_reduce2 || d->out_d2s || d->in_s2d) return 0;
  rd::ConvArgs a; fill_args(d, a);
  return RD_NS(d->dtype, conv_add_ok)(a, RD_DT(d->dtype)) ? 1 : 0;
}
int rd_conv_fwd_add(const rd_conv_desc* d, const void* src1, const void* src2, const void* w_packed, const float* bias, const void* addend,
                    void* dst, void* stream) {
  if (int e = check_desc(d)) return e;
  if (!src1 || !w_packed || !dst || !addend) return fail("conv_fwd_add: null pointer");
  if (d->C2 > 0 && !src2) return fail("conv_fwd_add: C2 > 0 but src2 is null");
  if (!rd_conv_add_ok(d)) return fail("conv_fwd_add: not available for this descriptor (see rd_conv_add_ok)");
  rd::ConvArgs a; fill_args(d, a);
  a.src1 = src1; a.src2 = src2; a.w = w_packed; a.bias = bias; a.dst1 = dst; a.dst2 = nullptr; a.stats = nullptr; a.add1 = addend;
  RD_NS(d->dtype, launch_conv)(a, RD_DT(d->dtype), S(stream));
  return done("rd_conv_fwd_add");
}
namespace {
// fusion request -> kernel arguments; returns false when a part is asked for that the routed kernel cannot do
bool apply_fusion(const rd_conv_desc* d, const rd_conv_fusion* f, rd::ConvArgs& a) {
  if (!f) return true;
  const bool want_in = f->in_scale != nullptr, want_bn = f->bn_y != nullptr;
  if (want_in && !f->in_shift) return false;
  if (want_bn && (!f->bn_scale || !f->bn_shift || !f->bn_mean || !f->bn_rstd)) return false;
  if (want_in) { a.in_scale = f->in_scale; a.in_shift = f->in_shift; a.in_act = f->in_act; a.in_slope = f->in_slope; }
  if (want_bn) { a.bn_y = f->bn_y; a.bn_scale = f->bn_scale; a.bn_shift = f->bn_shift; a.bn_mean = f->bn_mean; a.bn_rstd = f->bn_rstd;
                 a.bn_act = f->bn_act; a.bn_slope = f->bn_slope; }
  if (want_in && !RD_NS(d->dtype, conv_in_affine_ok)(a, RD_DT(d->dtype))) return false;
  if (want_bn && !RD_NS(d->dtype, conv_bn_bwd_ok)(a, RD_DT(d->dtype))) return false;
  return true;
}
}  // namespace
int32_t rd_conv_fusion_ok(const rd_conv_desc* d, const rd_conv_fusion* f) {
  if (!d || check_desc(d)) return 0;
  rd::ConvArgs a; fill_args(d, a);
  return apply_fusion(d, f, a) ? 1 : 0;
}
int rd_conv_fwd_fused(const rd_conv_desc* d, const rd_conv_fusion* f, const void* src1, const void* src2, const void* w_packed,
                      const float* bias, const void* addend, void* dst1, void* dst2, float* stats, void* stream) {
  if (int e = check_desc(d)) return e;
  if (!src1 || !w_packed || !dst1) return fail("conv_fwd_fused: null pointer");
  if (d->C2 > 0 && !src2) return fail("conv_fwd_fused: C2 > 0 but src2 is null");
  if (d->out_d2s || d->in_s2d) return fail("conv_fwd_fused: out_d2s / in_s2d have no fused form");
  if (d->D1 < d->Cout && !dst2) return fail("conv_fwd_fused: D1 < Cout but dst2 is null");
  if (addend && !rd_conv_add_ok(d)) return fail("conv_fwd_fused: an addend is not available for this descriptor (see rd_conv_add_ok)");
  rd::ConvArgs a; fill_args(d, a);
  a.src1 = src1; a.src2 = src2; a.w = w_packed; a.bias = bias; a.dst1 = dst1; a.dst2 = dst2; a.stats = stats; a.add1 = addend;
  if (a.pool2 && (bias || d->act != RD_ACT_NONE || !rd_conv_out_reduce2_ok(d)))
    return fail("conv_fwd_fused: out_reduce2 is not available for this descriptor / with a bias or an activation (see rd_conv_out_reduce2_ok)");
  if (!apply_fusion(d, f, a)) return fail("conv_fwd_fused: the requested fusion is not available for this descriptor (see rd_conv_fusion_ok)");
  if (a.bn_y && !stats) return fail("conv_fwd_fused: the BatchNorm-backward sums need a stats buffer");
  if (!a.bn_y && a.pool2 && stats) return fail("conv_fwd_fused: out_reduce2 has no forward statistics");
  RD_NS(d->dtype, launch_conv)(a, RD_DT(d->dtype), S(stream));
  return done("rd_conv_fwd_fused");
}
const char* rd_conv_fwd_kernel_name(const rd_conv_desc* d) {
  if (!d || check_desc(d)) return "";
  rd::ConvArgs a; fill_args(d, a);
  return RD_NS(d->dtype, conv_kernel_name)(a, RD_DT(d->dtype));
}
const char* rd_conv_fused_kernel_name(const rd_conv_desc* d, const rd_conv_fusion* f) {
  if (!d || check_desc(d)) return "";
  rd::ConvArgs a; fill_args(d, a);
  if (!apply_fusion(d, f, a)) return "";
  return RD_NS(d->dtype, conv_kernel_name)(a, RD_DT(d->dtype));
}
static void fill_wgrad_args(const rd_conv_desc* d, rd::WgradArgs& a);
const char* rd_conv_wgrad_kernel_name(const rd_conv_desc* d) {
  if (!d || check_desc(d) || d->in_dilate != 1) return "";
  rd::WgradArgs a; fill_wgrad_args(d, a);
  return RD_NS(d->dtype, wgrad_kernel_name)(a, RD_DT(d->dtype));
}
const char* rd_conv_wgrad_fused_kernel_name(const rd_conv_desc* d, const rd_conv_fusion* f) {
  if (!d || !rd_conv_wgrad_fusion_ok(d, f)) return "";
  rd::WgradArgs a; fill_wgrad_args(d, a);
  if (f && f->in_scale) { a.in_scale = f->in_scale; a.in_shift = f->in_shift; }
  return RD_NS(d->dtype, wgrad_kernel_name)(a, RD_DT(d->dtype));
}
static void fill_wgrad_args(const rd_conv_desc* d, rd::WgradArgs& a) {
  memset(&a, 0, sizeof(a));
  a.N = d->N; a.Hin = d->Hin; a.Win = d->Win; a.C1 = d->C1; a.C2 = d->C2;
  a.ups = d->upsample ? 1 : 0; a.H1 = a.ups ? d->H1 : d->Hin; a.W1 = a.ups ? d->W1 : d->Win;
  a.Cout = d->Cout; a.KH = d->KH; a.KW = d->KW; a.stride = d->stride; a.pad = d->pad; a.OH = d->OH; a.OW = d->OW;
  a.scale_h = (float)a.H1 / (float)d->Hin; a.scale_w = (float)a.W1 / (float)d->Win;
  a.M = d->N * d->OH * d->OW;
  a.K = d->KH * d->KW * (d->C1 + d->C2);
}
int64_t rd_conv_wgrad_workspace_bytes(const rd_conv_desc* d) {
  rd::WgradArgs a; fill_wgrad_args(d, a);
  int ns = rd::wgrad_slabs(a.M, a.K, d->Cout);
  if (RD_NS(d->dtype, wgrad3x3_tr_ok)(a, RD_DT(d->dtype))) ns = std::max(ns, rd::wgrad3x3_tr_blocks(a));
  // the same shape with a consumer-side BatchNorm apply runs on the 8 x TW kernel even where the map-fitted one takes the plain launch
  static const float kDummy = 0.f;
  a.in_scale = &kDummy;
  if (RD_NS(d->dtype, wgrad3x3_tr_ok)(a, RD_DT(d->dtype))) ns = std::max(ns, rd::wgrad3x3_tr_blocks(a));
  a.in_scale = nullptr;
  return (int64_t)(ns + 1) * d->Cout * a.K * (int64_t)sizeof(float);
}
int64_t rd_workspace_bytes(int32_t op, const rd_conv_desc* d) {
  // SURVEY 8(b)'s one sizing entry for the caller-owned buffers of a convolution layer (the per-op helpers stay: this dispatches to them)
  if (!d || check_desc(d)) return -1;
  const int es = d->dtype == RD_F32 ? 4 : 2;
  switch (op) {
    case RD_WS_CONV_WGRAD: return rd_conv_wgrad_workspace_bytes(d);
    case RD_WS_CONV_STATS: return (int64_t)rd_conv_stats_rows(d) * d->Cout * 2 * (int64_t)sizeof(float);
    case RD_WS_CONV_PACKED: return rd_conv_packed_elems(d->Cout, d->KH * d->KW * (d->C1 + d->C2), d->dtype) * es;
    case RD_WS_CONV_PACKED_DGRAD: return rd_conv_packed_elems(d->C1 + d->C2, d->KH * d->KW * d->Cout, d->dtype) * es;
    default: fail("workspace_bytes: unknown op %d", op); return -1;
  }
}
int rd_conv_wgrad(const rd_conv_desc* d, const void* src1, const void* src2, const void* dy, float* workspace, float* dw,
                  int32_t accumulate, void* stream) {
  if (int e = check_desc(d)) return e;
  if (!src1 || !dy || !workspace || !dw) return fail("conv_wgrad: null pointer");
  if (d->in_dilate != 1) return fail("conv_wgrad: in_dilate must be 1");
  rd::WgradArgs a; memset(&a, 0, sizeof(a));
  a.src1 = src1; a.src2 = src2; a.dy = dy; a.slab = workspace;
  a.N = d->N; a.Hin = d->Hin; a.Win = d->Win; a.C1 = d->C1; a.C2 = d->C2;
  a.ups = d->upsample ? 1 : 0; a.H1 = a.ups ? d->H1 : d->Hin; a.W1 = a.ups ? d->W1 : d->Win;
  a.Cout = d->Cout; a.KH = d->KH; a.KW = d->KW; a.stride = d->stride; a.pad = d->pad; a.OH = d->OH; a.OW = d->OW;
  a.scale_h = (float)a.H1 / (float)d->Hin; a.scale_w = (float)a.W1 / (float)d->Win;
  a.M = d->N * d->OH * d->OW; a.K = d->KH * d->KW * (d->C1 + d->C2);
  RD_NS(d->dtype, launch_wgrad)(a, RD_DT(d->dtype), dw, accumulate, S(stream), nullptr);
  return done("rd_conv_wgrad");
}
int32_t rd_conv_wgrad_streams(const rd_conv_desc* d) {
  if (!d || check_desc(d) || d->in_dilate != 1) return 0;
  rd::WgradArgs a; fill_wgrad_args(d, a);
  a.C1 = d->C1; a.C2 = d->C2; a.ups = d->upsample ? 1 : 0; a.KH = d->KH; a.KW = d->KW; a.stride = d->stride; a.Cout = d->Cout;
  return RD_NS(d->dtype, wgrad_streams)(a) ? 1 : 0;
}
static_assert(sizeof(rd_wgrad_reduce_item) == sizeof(rdt::WgradReduceItem), "rd_wgrad_reduce_item layout");
int rd_conv_wgrad_partial(const rd_conv_desc* d, const void* src1, const void* src2, const void* dy, float* workspace, float* dw,
                          int32_t accumulate, rd_wgrad_reduce_item* item, void* stream) {
  if (int e = check_desc(d)) return e;
  if (!src1 || !dy || !workspace || !dw || !item) return fail("conv_wgrad_partial: null pointer");
  if (d->in_dilate != 1) return fail("conv_wgrad_partial: in_dilate must be 1");
  rd::WgradArgs a; memset(&a, 0, sizeof(a));
  a.src1 = src1; a.src2 = src2; a.dy = dy; a.slab = workspace;
  a.N = d->N; a.Hin = d->Hin; a.Win = d->Win; a.C1 = d->C1; a.C2 = d->C2;
  a.ups = d->upsample ? 1 : 0; a.H1 = a.ups ? d->H1 : d->Hin; a.W1 = a.ups ? d->W1 : d->Win;
  a.Cout = d->Cout; a.KH = d->KH; a.KW = d->KW; a.stride = d->stride; a.pad = d->pad; a.OH = d->OH; a.OW = d->OW;
  a.scale_h = (float)a.H1 / (float)d->Hin; a.scale_w = (float)a.W1 / (float)d->Win;
  a.M = d->N * d->OH * d->OW; a.K = d->KH * d->KW * (d->C1 + d->C2);
  RD_NS(d->dtype, launch_wgrad)(a, RD_DT(d->dtype), dw, accumulate, S(stream), reinterpret_cast<rdt::WgradReduceItem*>(item));
  return done("rd_conv_wgrad_partial");
}
int32_t rd_conv_wgrad_fusion_ok(const rd_conv_desc* d, const rd_conv_fusion* f) {
  if (!d || check_desc(d) || d->in_dilate != 1) return 0;
  if (!f || !f->in_scale) return 1;
  if (!f->in_shift || f->bn_y) return 0;
  rd::WgradArgs a; fill_wgrad_args(d, a);
  return RD_NS(d->dtype, wgrad_in_affine_ok)(a, RD_DT(d->dtype)) ? 1 : 0;
}
int rd_conv_wgrad_partial_fused(const rd_conv_desc* d, const rd_conv_fusion* f, const void* src1, const void* src2, const void* dy,
                                float* workspace, float* dw, int32_t accumulate, rd_wgrad_reduce_item* item, void* stream) {
  if (int e = check_desc(d)) return e;
  if (!src1 || !dy || !workspace || !dw || !item) return fail("conv_wgrad_partial_fused: null pointer");
  if (d->in_dilate != 1) return fail("conv_wgrad_partial_fused: in_dilate must be 1");
  if (!rd_conv_wgrad_fusion_ok(d, f)) return fail("conv_wgrad_partial_fused: the requested fusion is not available for this descriptor (see rd_conv_wgrad_fusion_ok)");
  rd::WgradArgs a; fill_wgrad_args(d, a);
  a.src1 = src1; a.src2 = src2; a.dy = dy; a.slab = workspace;
  if (f && f->in_scale) { a.in_scale = f->in_scale; a.in_shift = f->in_shift; a.in_act = f->in_act; a.in_slope = f->in_slope; }
  RD_NS(d->dtype, launch_wgrad)(a, RD_DT(d->dtype), dw, accumulate, S(stream), reinterpret_cast<rdt::WgradReduceItem*>(item));
  return done("rd_conv_wgrad_partial_fused");
}
int rd_wgrad_reduce_batch(const rd_wgrad_reduce_item* items, int32_t n, void* stream) {
  if (n < 0 || (n > 0 && !items)) return fail("wgrad_reduce_batch: bad args");
  for (int i = 0; i < n; i++)
    if (!items[i].slab || !items[i].dw || items[i].nsplit < 1) return fail("wgrad_reduce_batch: item not filled by rd_conv_wgrad_partial");
  if (n) rd::launch_wgrad_reduce_batch(reinterpret_cast<const rdt::WgradReduceItem*>(items), n, S(stream));
  return done("rd_wgrad_reduce_batch");
}

int rd_bn_finalize(const float* stats, int32_t rows, int32_t C, double count, const float* gamma, const float* beta, float eps,
                   float momentum, int32_t training, float* running_mean, float* running_var, float* save_mean,
                   float* save_rstd, float* scale, float* shift, void* stream) {
  if (!scale || !shift) return fail("bn_finalize: null scale/shift");
  if (training && !stats) return fail("bn_finalize: training needs stats");
  if (!training && (!running_mean || !running_var)) return fail("bn_finalize: eval needs running stats");
  rd::launch_bn_finalize(stats, rows, C, count, gamma, beta, eps, momentum, training, running_mean, running_var, save_mean,
                         save_rstd, scale, shift, S(stream));
  return done("rd_bn_finalize");
}
int rd_affine_act(const void* y, const float* scale, const float* shift, const void* residual, void* out, int64_t pixels,
                  int32_t C, int32_t act, float slope, int32_t dtype, void* stream) {
  if (!y || !out || !dt_ok(dtype)) return fail("affine_act: bad args");
  if (pixels * C == 0) return 0;
  RD_NS(dtype, launch_affine_act)(y, scale, shift, residual, out, pixels, C, act, slope, RD_DT(dtype), S(stream));
  return done("rd_affine_act");
}
int32_t rd_affine_act_add_ok(int32_t C, int32_t dtype) { return dt_ok(dtype) && rd::affine_act_add_ok(C, RD_DT(dtype)) ? 1 : 0; }
int rd_affine_act_add(const void* y, const float* scale, const float* shift, int32_t act1, float slope1, const void* residual, void* out,
                      int64_t pixels, int32_t C, int32_t act2, float slope2, int32_t dtype, void* stream) {
  if (!y || !scale || !shift || !residual || !out || !dt_ok(dtype)) return fail("affine_act_add: bad args");
  if (!rd::affine_act_add_ok(C, RD_DT(dtype))) return fail("affine_act_add: channel count %d has no vector form (see rd_affine_act_add_ok)", C);
  if (pixels * C == 0) return 0;
  RD_NS(dtype, launch_affine_act_add)(y, scale, shift, act1, slope1, residual, out, pixels, C, act2, slope2, RD_DT(dtype), S(stream));
  return done("rd_affine_act_add");
}
int32_t rd_bn_slab_ok(int64_t pixels, int32_t C, int32_t dtype) { return dt_ok(dtype) && rd::bn_slab_ok(pixels, C, RD_DT(dtype)) ? 1 : 0; }
int rd_bn_finalize_apply(const float* stats, int32_t rows, const void* y, const float* gamma, const float* beta, float eps, float momentum,
                         float* running_mean, float* running_var, float* save_mean, float* save_rstd, float* scale, float* shift, void* out, int64_t pixels,
                         int32_t C, int32_t act, float slope, int32_t dtype, void* stream) {
  if (!stats || rows <= 0 || !y || !scale || !shift || !out || !dt_ok(dtype)) return fail("bn_finalize_apply: bad args");
  if (!rd::bn_slab_ok(pixels, C, RD_DT(dtype))) return fail("bn_finalize_apply: %lld pixels x %d channels has no one-launch form (see rd_bn_slab_ok)", (long long)pixels, C);
  RD_NS(dtype, launch_bn_fwd_slab)(stats, rows, y, gamma, beta, eps, momentum, running_mean, running_var, save_mean, save_rstd, scale, shift, out, pixels, C, act, slope,
                                   RD_DT(dtype), S(stream));
  return done("rd_bn_finalize_apply");
}
int rd_bn_act_bwd_slab(const void* dz, const void* y, const float* mean, const float* rstd, const float* scale, const float* shift, float* dgamma, float* dbeta,
                       int32_t accumulate, void* dy, int64_t pixels, int32_t C, int32_t act, float slope, int32_t dtype, void* stream) {
  if (!dz || !y || !mean || !rstd || !scale || !shift || !dy || !dt_ok(dtype)) return fail("bn_act_bwd_slab: bad args");
  if (!rd::bn_slab_ok(pixels, C, RD_DT(dtype))) return fail("bn_act_bwd_slab: %lld pixels x %d channels has no one-launch form (see rd_bn_slab_ok)", (long long)pixels, C);
  RD_NS(dtype, launch_bn_bwd_slab)(dz, y, mean, rstd, scale, shift, dgamma, dbeta, accumulate, dy, pixels, C, act, slope, RD_DT(dtype), S(stream));
  return done("rd_bn_act_bwd_slab");
}
const char* rd_bn_slab_kernel_name(int32_t which, int64_t pixels, int32_t dtype, int32_t act) {
  if (!dt_ok(dtype) || which < 0 || which > 1) return "";
  return RD_NS(dtype, bn_slab_kernel_name)(which, pixels, RD_DT(dtype), act);
}
int32_t rd_bn_bwd_rows(int64_t pixels, int32_t C) { return rd::bn_bwd_rows(pixels, C); }
int rd_bn_act_bwd(const void* dz, const void* z, const void* y, const float* mean, const float* rstd, const float* scale,
                  float* partial, float* coef, float* dgamma, float* dbeta, int32_t accumulate, void* dy, void* dres,
                  int64_t pixels, int32_t C, int32_t act, float slope, int32_t dtype, void* stream) {
  if (!dz || !y || !mean || !rstd || !scale || !partial || !coef || !dy || !dt_ok(dtype)) return fail("bn_act_bwd: bad args");
  if (act != RD_ACT_NONE && !z) return fail("bn_act_bwd: activation backward needs z");
  int rows = rd::bn_bwd_rows(pixels, C);
  RD_NS(dtype, launch_bn_bwd_reduce)(dz, z, y, mean, rstd, partial, pixels, C, act, slope, RD_DT(dtype), S(stream), nullptr, nullptr);
  rd::launch_bn_bwd_finalize(partial, rows, C, (double)pixels, dgamma, dbeta, accumulate, coef, coef + C, S(stream));
  RD_NS(dtype, launch_bn_bwd_apply)(dz, z, y, mean, rstd, scale, coef, coef + C, dy, dres, pixels, C, act, slope, RD_DT(dtype), S(stream), nullptr);
  return done("rd_bn_act_bwd");
}
int rd_bn_act_bwd_recompute_phases(const void* dz, const void* z, const void* y, const float* mean, const float* rstd, const float* scale,
                                   const float* shift, float* partial, float* coef, float* dgamma, float* dbeta, int32_t accumulate, void* dy,
                                   void* dres, int64_t pixels, int32_t C, int32_t act, float slope, int32_t dtype, int32_t phases, void* stream) {
  if (!dz || !y || !mean || !rstd || !scale || !shift || !partial || !coef || !dy || !dt_ok(dtype)) return fail("bn_act_bwd_recompute: bad args");
  if (act != RD_ACT_NONE && !z && (C % (dtype == RD_F32 ? 4 : 8))) return fail("bn_act_bwd_recompute: this channel count needs z");
  int rows = rd::bn_bwd_rows(pixels, C);
  if (phases & 1) RD_NS(dtype, launch_bn_bwd_reduce)(dz, z, y, mean, rstd, partial, pixels, C, act, slope, RD_DT(dtype), S(stream), scale, shift);
  if (phases & 2) rd::launch_bn_bwd_finalize(partial, rows, C, (double)pixels, dgamma, dbeta, accumulate, coef, coef + C, S(stream));
  if (phases & 4) RD_NS(dtype, launch_bn_bwd_apply)(dz, z, y, mean, rstd, scale, coef, coef + C, dy, dres, pixels, C, act, slope, RD_DT(dtype), S(stream), shift);
  return done("rd_bn_act_bwd_recompute");
}
int rd_bn_act_bwd_from_partial(const void* dz, const void* y, const float* mean, const float* rstd, const float* scale, const float* shift,
                               const float* partial, int32_t rows, int32_t row_channels, float* coef, float* dgamma, float* dbeta, int32_t accumulate,
                               void* dy, int64_t pixels, int32_t C, int32_t act, float slope, int32_t dtype, void* stream) {
  if (!dz || !y || !mean || !rstd || !scale || !shift || !partial || !coef || !dy || !dt_ok(dtype)) return fail("bn_act_bwd_from_partial: bad args");
  if (rows <= 0 || row_channels < C || C <= 0) return fail("bn_act_bwd_from_partial: bad partial geometry (rows %d, row channels %d, C %d)", rows, row_channels, C);
  if (act != RD_ACT_NONE && (C % (dtype == RD_F32 ? 4 : 8))) return fail("bn_act_bwd_from_partial: this channel count needs z");
  rd::launch_bn_bwd_finalize(partial, rows, C, (double)pixels, dgamma, dbeta, accumulate, coef, coef + C, S(stream), row_channels);
  RD_NS(dtype, launch_bn_bwd_apply)(dz, nullptr, y, mean, rstd, scale, coef, coef + C, dy, nullptr, pixels, C, act, slope, RD_DT(dtype), S(stream), shift);
  return done("rd_bn_act_bwd_from_partial");
}
int rd_bn_act_bwd_recompute(const void* dz, const void* z, const void* y, const float* mean, const float* rstd, const float* scale,
                            const float* shift, float* partial, float* coef, float* dgamma, float* dbeta, int32_t accumulate, void* dy,
                            void* dres, int64_t pixels, int32_t C, int32_t act, float slope, int32_t dtype, void* stream) {
  return rd_bn_act_bwd_recompute_phases(dz, z, y, mean, rstd, scale, shift, partial, coef, dgamma, dbeta, accumulate, dy, dres, pixels, C, act, slope,
                                        dtype, 7, stream);
}
/* ---- decoder head (rd_head.hip): BatchNorm + activation of the last decoder convolution fused with the one-channel 3x3 output convolution */
int32_t rd_bn_head_ok(int32_t N, int32_t H, int32_t W, int32_t C, int32_t dtype) {
  return dt_ok(dtype) && RD_NS(dtype, bn_head_ok)(N, H, W, C, RD_DT(dtype)) ? 1 : 0;
}
int32_t rd_bn_head_rows(int32_t N, int32_t H, int32_t W) { return rd::bn_head_rows(N, H, W); }
int rd_bn_head_fwd(const void* y, const float* scale, const float* shift, int32_t act, float slope, const float* w_head, void* logits, int32_t N,
                   int32_t H, int32_t W, int32_t C, int32_t dtype, void* stream) {
  if (!y || !scale || !shift || !w_head || !logits || !dt_ok(dtype)) return fail("bn_head_fwd: bad args");
  if (!rd_bn_head_ok(N, H, W, C, dtype)) return fail("bn_head_fwd: unsupported shape (N %d, %d x %d, C %d)", N, H, W, C);
  RD_NS(dtype, launch_bn_head_fwd)(y, scale, shift, act, slope, w_head, logits, N, H, W, RD_DT(dtype), S(stream));
  return done("rd_bn_head_fwd");
}
int rd_bn_head_bwd_reduce(const void* dlogits, const void* y, const float* mean, const float* rstd, const float* scale, const float* shift,
                          int32_t act, float slope, const float* w_head, float* partial, int32_t N, int32_t H, int32_t W, int32_t C, int32_t dtype,
                          void* stream) {
  if (!dlogits || !y || !mean || !rstd || !scale || !shift || !w_head || !partial || !dt_ok(dtype)) return fail("bn_head_bwd_reduce: bad args");
  if (!rd_bn_head_ok(N, H, W, C, dtype)) return fail("bn_head_bwd_reduce: unsupported shape (N %d, %d x %d, C %d)", N, H, W, C);
  RD_NS(dtype, launch_bn_head_bwd_reduce)(dlogits, y, mean, rstd, scale, shift, act, slope, w_head, partial, N, H, W, RD_DT(dtype), S(stream));
  return done("rd_bn_head_bwd_reduce");
}
int rd_bn_head_bwd_apply(const void* dlogits, const void* y, const float* mean, const float* rstd, const float* scale, const float* shift,
                         int32_t act, float slope, const float* w_head, const float* partial, int32_t rows, float* coef, float* dgamma, float* dbeta,
                         int32_t bn_accumulate, float* dw_head, int32_t w_accumulate, void* dy, int32_t N, int32_t H, int32_t W, int32_t C,
                         int32_t dtype, void* stream) {
  if (!dlogits || !y || !mean || !rstd || !scale || !shift || !w_head || !partial || !coef || !dy || !dt_ok(dtype)) return fail("bn_head_bwd_apply: bad args");
  if (!rd_bn_head_ok(N, H, W, C, dtype)) return fail("bn_head_bwd_apply: unsupported shape (N %d, %d x %d, C %d)", N, H, W, C);
  if (rows <= 0 || rows != rd_bn_head_rows(N, H, W)) return fail("bn_head_bwd_apply: rows %d, expected %d", rows, rd_bn_head_rows(N, H, W));
  RD_NS(dtype, launch_bn_head_bwd_apply)(dlogits, y, mean, rstd, scale, shift, act, slope, w_head, partial, rows, coef, dgamma, dbeta, bn_accumulate,
                                         dw_head, w_accumulate, dy, N, H, W, RD_DT(dtype), S(stream));
  return done("rd_bn_head_bwd_apply");
}
const char* rd_bn_head_kernel_name(int32_t which, int32_t dtype, int32_t act) {
  if (!dt_ok(dtype) || which < 0 || which > 2) return "";
  return RD_NS(dtype, bn_head_kernel_name)(which, RD_DT(dtype), act);
}
const char* rd_bn_kernel_name(int32_t which, int32_t C, int32_t dtype, int32_t act, int32_t flag) {
  if (!dt_ok(dtype) || which < 0 || which > 2 || C <= 0) return "";
  return RD_NS(dtype, bn_kernel_name)(which, C, RD_DT(dtype), act, flag);
}
int rd_act_bwd(const void* dz, const void* z, void* dx, int64_t n, int32_t act, float slope, int32_t dtype, void* stream) {
  if (!dz || !z || !dx || !dt_ok(dtype)) return fail("act_bwd: bad args");
  if (n == 0) return 0;
  RD_NS(dtype, launch_act_bwd)(dz, z, dx, n, act, slope, RD_DT(dtype), S(stream));
  return done("rd_act_bwd");
}
int32_t rd_colsum_rows(int64_t rows, int32_t C) { return rd::colsum_rows(rows, C); }
int rd_colsum(const void* x, float* partial, float* out, int32_t accumulate, int64_t rows, int32_t C, int32_t dtype, void* stream) {
  if (!x || !partial || !out || !dt_ok(dtype)) return fail("colsum: bad args");
  RD_NS(dtype, launch_colsum)(x, partial, out, accumulate, rows, C, RD_DT(dtype), S(stream));
  return done("rd_colsum");
}
static_assert(sizeof(rd_ln_grad_item) == sizeof(rdt::LnGradItem), "rd_ln_grad_item layout");
int rd_ln_grad_batch(const rd_ln_grad_item* items, int32_t n, void* stream) {
  if (n <= 0) return 0;
  if (!items) return fail("ln_grad_batch: bad args");
  for (int i = 0; i < n; i++) {
    const rd_ln_grad_item& it = items[i];
    if (!it.dgamma || !it.dbeta || it.C <= 0 || it.nparts < 1 || it.nparts > 4) return fail("ln_grad_batch: bad item");
    for (int p = 0; p < it.nparts; p++) if (!it.partial[p] || it.rows[p] <= 0) return fail("ln_grad_batch: bad partial");
  }
  rd::launch_ln_grad_batch(reinterpret_cast<const rdt::LnGradItem*>(items), n, S(stream));
  return done("rd_ln_grad_batch");
}
static_assert(sizeof(rd_colsum_item) == sizeof(rdt::ColsumItem), "rd_colsum_item layout");
int rd_colsum_partial(const void* x, float* partial, int64_t rows, int32_t C, int32_t dtype, void* stream) {
  if (!x || !partial || !dt_ok(dtype)) return fail("colsum_partial: bad args");
  RD_NS(dtype, launch_colsum)(x, partial, nullptr, 0, rows, C, RD_DT(dtype), S(stream));
  return done("rd_colsum_partial");
}
int rd_colsum_finalize_batch(const rd_colsum_item* items, int32_t n, void* stream) {
  if (n <= 0) return 0;
  if (!items) return fail("colsum_finalize_batch: bad args");
  for (int i = 0; i < n; i++)
    if (!items[i].partial || !items[i].out || items[i].rows <= 0 || items[i].C <= 0) return fail("colsum_finalize_batch: bad item");
  rd::launch_colsum_finalize_batch(reinterpret_cast<const rdt::ColsumItem*>(items), n, S(stream));
  return done("rd_colsum_finalize_batch");
}

int rd_layernorm_fwd(const void* x, const float* gamma, const float* beta, const void* residual, void* out, float* mean,
                     float* rstd, int64_t rows, int32_t C, float eps, int32_t dtype, void* stream) {
  if (!x || !gamma || !beta || !out || !mean || !rstd || !dt_ok(dtype)) return fail("layernorm_fwd: bad args");
  if (C % 64 || C > 512) return fail("layernorm: C must be a multiple of 64 and <= 512 (got %d)", C);
  if (rows == 0) return 0;
  RD_NS(dtype, launch_layernorm_fwd)(x, gamma, beta, residual, out, mean, rstd, rows, C, eps, RD_DT(dtype), S(stream));
  return done("rd_layernorm_fwd");
}
int32_t rd_layernorm_bwd_rows(int64_t rows) { return rd::layernorm_bwd_rows(rows); }
int rd_layernorm_bwd(const void* dout, const void* x, const float* gamma, const float* mean, const float* rstd, void* dx,
                     float* partial, float* dgamma, float* dbeta, int32_t accumulate, int64_t rows, int32_t C, int32_t dtype,
                     void* stream) {
  if (!dout || !x || !gamma || !mean || !rstd || !dx || !partial || !dt_ok(dtype)) return fail("layernorm_bwd: bad args");
  if (C % 64 || C > 512) return fail("layernorm: C must be a multiple of 64 and <= 512 (got %d)", C);
  RD_NS(dtype, launch_layernorm_bwd)(dout, x, gamma, mean, rstd, dx, partial, dgamma, dbeta, accumulate, rows, C, RD_DT(dtype), S(stream));
  return done("rd_layernorm_bwd");
}

int rd_linear_attention_fwd(const void* q, const void* k, const void* v, void* out, int32_t N, int32_t L, int32_t Sx, int32_t H,
                            int32_t ldq, int32_t ldk, int32_t ldv, int32_t ldo, float eps, int32_t dtype, void* stream) {
  if (!q || !k || !v || !out || !dt_ok(dtype)) return fail("linear_attention_fwd: bad args");
  if (L <= 0 || Sx <= 0 || L > 32 || Sx > 32) return fail("linear_attention: L,S must be in 1..32 (got %d,%d)", L, Sx);
  RD_NS(dtype, launch_linear_attention_fwd)(q, k, v, out, N, L, Sx, H, ldq, ldk, ldv, ldo, eps, RD_DT(dtype), S(stream));
  return done("rd_linear_attention_fwd");
}
int rd_linear_attention_bwd(const void* q, const void* k, const void* v, const void* dout, void* dq, void* dk, void* dv,
                            int32_t N, int32_t L, int32_t Sx, int32_t H, int32_t ldq, int32_t ldk, int32_t ldv, int32_t ldo,
                            float eps, int32_t dtype, void* stream) {
  if (!q || !k || !v || !dout || !dq || !dk || !dv || !dt_ok(dtype)) return fail("linear_attention_bwd: bad args");
  if (L <= 0 || Sx <= 0 || L > 32 || Sx > 32) return fail("linear_attention: L,S must be in 1..32 (got %d,%d)", L, Sx);
  RD_NS(dtype, launch_linear_attention_bwd)(q, k, v, dout, dq, dk, dv, N, L, Sx, H, ldq, ldk, ldv, ldo, eps, RD_DT(dtype), S(stream));
  return done("rd_linear_attention_bwd");
}

int rd_maxpool_fwd(const void* x, void* out, uint8_t* arg, int32_t N, int32_t H, int32_t W, int32_t C, int32_t OH, int32_t OW,
                   int32_t k, int32_t s, int32_t p, int32_t dtype, void* stream) {
  if (!x || !out || !arg || !dt_ok(dtype) || k * k > 255) return fail("maxpool_fwd: bad args");
  RD_NS(dtype, launch_maxpool_fwd)(x, out, arg, N, H, W, C, OH, OW, k, s, p, RD_DT(dtype), S(stream));
  return done("rd_maxpool_fwd");
}
int rd_maxpool_bwd(const void* dout, const uint8_t* arg, void* dx, int32_t N, int32_t H, int32_t W, int32_t C, int32_t OH,
                   int32_t OW, int32_t k, int32_t s, int32_t p, int32_t dtype, void* stream) {
  if (!dout || !dx || !arg || !dt_ok(dtype)) return fail("maxpool_bwd: bad args");
  RD_NS(dtype, launch_maxpool_bwd)(dout, arg, dx, N, H, W, C, OH, OW, k, s, p, RD_DT(dtype), S(stream));
  return done("rd_maxpool_bwd");
}
int rd_roi_pool_fwd(const void* x, const float* rois, void* out, int32_t* argmax, int32_t R, int32_t N, int32_t H, int32_t W,
                    int32_t C, int32_t PH, int32_t PW, float scale, int32_t dtype, void* stream) {
  if (R == 0) return 0;
  if (!x || !rois || !out || !argmax || !dt_ok(dtype)) return fail("roi_pool_fwd: bad args");
  RD_NS(dtype, launch_roi_pool_fwd)(x, rois, out, argmax, R, N, H, W, C, PH, PW, scale, RD_DT(dtype), S(stream));
  return done("rd_roi_pool_fwd");
}
int rd_roi_pool_bwd(const void* dout, const float* rois, const int32_t* argmax, float* dx, int32_t R, int32_t N, int32_t H,
                    int32_t W, int32_t C, int32_t PH, int32_t PW, int32_t dtype, void* stream) {
  if (!dx || !dt_ok(dtype)) return fail("roi_pool_bwd: bad args");
  if (R > 0 && (!dout || !rois || !argmax)) return fail("roi_pool_bwd: null pointer");
  RD_NS(dtype, launch_roi_pool_bwd)(dout, rois, argmax, dx, R, N, H, W, C, PH, PW, RD_DT(dtype), S(stream));
  return done("rd_roi_pool_bwd");
}

int rd_roi_pool_bwd_tile(const void* dout, const float* rois, const int32_t* argmax, void* dx, int32_t R, int32_t N, int32_t H,
                         int32_t W, int32_t C, int32_t PH, int32_t PW, float scale, int32_t dtype, void* stream) {
  if (!dx || !dt_ok(dtype)) return fail("roi_pool_bwd_tile: bad args");
  if (R > 0 && (!dout || !rois || !argmax)) return fail("roi_pool_bwd_tile: null pointer");
  if (C % 32) return fail("roi_pool_bwd_tile: C must be a multiple of 32");
  if (N <= 0 || H <= 0 || W <= 0 || PH <= 0 || PW <= 0 || (int64_t)H * W >= (1 << 24)) return fail("roi_pool_bwd_tile: bad sizes");
  RD_NS(dtype, launch_roi_pool_bwd_tile)(dout, rois, argmax, dx, R, N, H, W, C, PH, PW, scale, RD_DT(dtype), S(stream));
  return done("rd_roi_pool_bwd_tile");
}
int rd_roi_pool_bwd_gather(const void* dout, const float* rois, const int32_t* argmax, void* dx, int32_t R, int32_t N, int32_t H,
                           int32_t W, int32_t C, int32_t PH, int32_t PW, float scale, int32_t dtype, void* stream) {
  if (!dx || !dt_ok(dtype)) return fail("roi_pool_bwd_gather: bad args");
  if (R > 0 && (!dout || !rois || !argmax)) return fail("roi_pool_bwd_gather: null pointer");
  if (C % (dtype == RD_F32 ? 4 : 8)) return fail("roi_pool_bwd_gather: C must be a multiple of the 16-byte vector");
  if (N <= 0 || H <= 0 || W <= 0 || PH <= 0 || PW <= 0 || PH >= (1 << 19) || PW >= (1 << 19)) return fail("roi_pool_bwd_gather: bad sizes");
  RD_NS(dtype, launch_roi_pool_bwd_gather)(dout, rois, argmax, dx, R, N, H, W, C, PH, PW, scale, RD_DT(dtype), S(stream));
  return done("rd_roi_pool_bwd_gather");
}
int rd_roi_pool_fwd_u8(const void* x, const float* rois, void* out, uint8_t* argmax, int32_t* overflow_flag, int32_t R, int32_t N, int32_t H,
                       int32_t W, int32_t C, int32_t PH, int32_t PW, float scale, int32_t dtype, void* stream) {
  if (R == 0) return 0;
  if (!x || !rois || !out || !argmax || !overflow_flag || !dt_ok(dtype)) return fail("roi_pool_fwd_u8: bad args");
  if (C % (dtype == RD_F32 ? 4 : 8)) return fail("roi_pool_fwd_u8: C must be a multiple of the 16-byte vector (use rd_roi_pool_fwd)");
  RD_NS(dtype, launch_roi_pool_fwd_u8)(x, rois, out, argmax, overflow_flag, R, N, H, W, C, PH, PW, scale, RD_DT(dtype), S(stream));
  return done("rd_roi_pool_fwd_u8");
}
int rd_roi_pool_bwd_u8(const void* dout, const float* rois, const uint8_t* argmax, const int32_t* overflow_flag, float* dx, int32_t R, int32_t N,
                       int32_t H, int32_t W, int32_t C, int32_t PH, int32_t PW, float scale, int32_t dtype, void* stream) {
  if (!dx || !overflow_flag || !dt_ok(dtype)) return fail("roi_pool_bwd_u8: bad args");
  if (R > 0 && (!dout || !rois || !argmax)) return fail("roi_pool_bwd_u8: null pointer");
  RD_NS(dtype, launch_roi_pool_bwd_u8)(dout, rois, argmax, overflow_flag, dx, R, N, H, W, C, PH, PW, scale, RD_DT(dtype), S(stream));
  return done("rd_roi_pool_bwd_u8");
}
int rd_roi_pool_bwd_gather_u8(const void* dout, const float* rois, const uint8_t* argmax, const int32_t* overflow_flag, void* dx, int32_t R,
                              int32_t N, int32_t H, int32_t W, int32_t C, int32_t PH, int32_t PW, float scale, int32_t dtype, void* stream) {
  if (!dx || !overflow_flag || !dt_ok(dtype)) return fail("roi_pool_bwd_gather_u8: bad args");
  if (R > 0 && (!dout || !rois || !argmax)) return fail("roi_pool_bwd_gather_u8: null pointer");
  if (C % (dtype == RD_F32 ? 4 : 8)) return fail("roi_pool_bwd_gather_u8: C must be a multiple of the 16-byte vector");
  if (N <= 0 || H <= 0 || W <= 0 || PH <= 0 || PW <= 0 || PH >= (1 << 19) || PW >= (1 << 19)) return fail("roi_pool_bwd_gather_u8: bad sizes");
  RD_NS(dtype, launch_roi_pool_bwd_gather_u8)(dout, rois, argmax, overflow_flag, dx, R, N, H, W, C, PH, PW, scale, RD_DT(dtype), S(stream));
  return done("rd_roi_pool_bwd_gather_u8");
}
int rd_cast(const void* src, void* dst, int64_t n, int32_t sd, int32_t dd, float scale, void* stream) {
  if (!src || !dst || !dt_ok(sd) || !dt_ok(dd)) return fail("cast: bad args");
  if ((sd == RD_BF16 && dd == RD_F16) || (sd == RD_F16 && dd == RD_BF16)) return fail("cast: bf16 <-> fp16 is not supported (the two 16-bit types live in separate builds): convert through fp32");
  if (n == 0) return 0;
  RD_NS(((sd == RD_F16 || dd == RD_F16) ? RD_F16 : RD_F32), launch_cast)(src, dst, n, RD_DT(sd), RD_DT(dd), scale, S(stream));
  return done("rd_cast");
}
int rd_add(const void* a, const void* b, void* out, int64_t n, int32_t dtype, void* stream) {
  if (!a || !b || !out || !dt_ok(dtype)) return fail("add: bad args");
  if (n == 0) return 0;
  RD_NS(dtype, launch_add)(a, b, out, n, RD_DT(dtype), S(stream));
  return done("rd_add");
}
int rd_nchw_to_nhwc(const void* src, void* dst, int32_t N, int32_t C, int32_t H, int32_t W, int32_t sd, int32_t dd, float scale,
                    void* stream) {
  if (!src || !dst || !dt_ok(sd) || !dt_ok(dd)) return fail("nchw_to_nhwc: bad args");
  if ((sd == RD_BF16 && dd == RD_F16) || (sd == RD_F16 && dd == RD_BF16)) return fail("nchw_to_nhwc: bf16 <-> fp16 is not supported (the two 16-bit types live in separate builds): convert through fp32");
  RD_NS(((sd == RD_F16 || dd == RD_F16) ? RD_F16 : RD_F32), launch_nchw_to_nhwc)(src, dst, N, C, H, W, RD_DT(sd), RD_DT(dd), scale, S(stream));
  return done("rd_nchw_to_nhwc");
}
int rd_nhwc_to_nchw(const void* src, void* dst, int32_t N, int32_t C, int32_t H, int32_t W, int32_t sd, int32_t dd, void* stream) {
  if (!src || !dst || !dt_ok(sd) || !dt_ok(dd)) return fail("nhwc_to_nchw: bad args");
  if ((sd == RD_BF16 && dd == RD_F16) || (sd == RD_F16 && dd == RD_BF16)) return fail("nhwc_to_nchw: bf16 <-> fp16 is not supported (the two 16-bit types live in separate builds): convert through fp32");
  RD_NS(((sd == RD_F16 || dd == RD_F16) ? RD_F16 : RD_F32), launch_nhwc_to_nchw)(src, dst, N, C, H, W, RD_DT(sd), RD_DT(dd), S(stream));
  return done("rd_nhwc_to_nchw");
}
int rd_transpose_last2(const void* src, void* dst, int64_t B, int32_t R, int32_t Cc, int32_t dtype, void* stream) {
  if (!src || !dst || !dt_ok(dtype)) return fail("transpose_last2: bad args");
  RD_NS(dtype, launch_transpose_last2)(src, dst, B, R, Cc, RD_DT(dtype), S(stream));
  return done("rd_transpose_last2");
}
int rd_concat2(const void* a, const void* b, void* out, int64_t rows, int32_t Ca, int32_t Cb, int32_t dtype, void* stream) {
  if (!a || !b || !out || !dt_ok(dtype)) return fail("concat2: bad args");
  RD_NS(dtype, launch_concat2)(a, b, out, rows, Ca, Cb, RD_DT(dtype), S(stream));
  return done("rd_concat2");
}
int rd_split2(const void* in, void* a, void* b, int64_t rows, int32_t Ca, int32_t Cb, int32_t dtype, void* stream) {
  if (!a || !b || !in || !dt_ok(dtype)) return fail("split2: bad args");
  RD_NS(dtype, launch_split2)(in, a, b, rows, Ca, Cb, RD_DT(dtype), S(stream));
  return done("rd_split2");
}
int rd_upsample_nearest_fwd(const void* x, void* y, int32_t N, int32_t Hs, int32_t Ws, int32_t Hv, int32_t Wv, int32_t C,
                            int32_t dtype, void* stream) {
  if (!x || !y || !dt_ok(dtype)) return fail("upsample_nearest_fwd: bad args");
  RD_NS(dtype, launch_upsample_nearest_fwd)(x, y, N, Hs, Ws, Hv, Wv, C, RD_DT(dtype), S(stream));
  return done("rd_upsample_nearest_fwd");
}
int rd_upsample_nearest_bwd(const void* dy, void* dx, int32_t N, int32_t Hs, int32_t Ws, int32_t Hv, int32_t Wv, int32_t C,
                            int32_t dtype, void* stream) {
  if (!dy || !dx || !dt_ok(dtype)) return fail("upsample_nearest_bwd: bad args");
  RD_NS(dtype, launch_upsample_nearest_bwd)(dy, dx, N, Hs, Ws, Hv, Wv, C, RD_DT(dtype), S(stream));
  return done("rd_upsample_nearest_bwd");
}

int rd_rcnet_labels(const float* gt, const float* points, float* label, float* valid, int32_t R, int32_t HW, float thr,
                    int32_t all_valid, void* stream) {
  if (!gt || !points || !label || !valid) return fail("rcnet_labels: null pointer");
  rd::launch_rcnet_labels(gt, points, label, valid, R, HW, thr, all_valid, S(stream));
  return done("rd_rcnet_labels");
}
int32_t rd_bce_rows(int64_t n) { return rd::bce_rows(n); }
int rd_bce_masked_fwd(const void* logits, const float* label, const float* valid, float pw, float* partial, float* loss,
                      float* sums, int64_t n, int32_t dtype, void* stream) {
  if (!logits || !label || !valid || !partial || !loss || !sums || !dt_ok(dtype)) return fail("bce_fwd: bad args");
  RD_NS(dtype, launch_bce_fwd)(logits, label, valid, pw, partial, loss, sums, n, RD_DT(dtype), S(stream));
  return done("rd_bce_masked_fwd");
}
int rd_bce_masked_bwd(const void* logits, const float* label, const float* valid, float pw, const float* sums, const float* dloss,
                      void* dlogits, int64_t n, int32_t dtype, void* stream) {
  if (!logits || !label || !valid || !sums || !dloss || !dlogits || !dt_ok(dtype)) return fail("bce_bwd: bad args");
  RD_NS(dtype, launch_bce_bwd)(logits, label, valid, pw, sums, dloss, dlogits, n, RD_DT(dtype), S(stream));
  return done("rd_bce_masked_bwd");
}
int rd_sigmoid(const void* x, void* y, int64_t n, int32_t dtype, void* stream) {
  if (!x || !y || !dt_ok(dtype)) return fail("sigmoid: bad args");
  if (n == 0) return 0;
  RD_NS(dtype, launch_sigmoid)(x, y, n, RD_DT(dtype), S(stream));
  return done("rd_sigmoid");
}
int rd_scatter_crops(const void* crops, const float* points, float* depth, float* response, int32_t Ncrop, int32_t PH, int32_t PW,
                     int32_t H, int32_t W, float thr, int32_t dtype, void* stream) {
  if (!depth || !response || !dt_ok(dtype)) return fail("scatter_crops: bad args");
  if (Ncrop > 0 && (!crops || !points)) return fail("scatter_crops: null pointer");
  if ((PH & 1) || (PW & 1)) return fail("scatter_crops: patch size must be even (reference uses patch//2 on both sides)");
  RD_NS(dtype, launch_scatter_crops)(crops, points, depth, response, Ncrop, PH, PW, H, W, thr, RD_DT(dtype), S(stream));
  return done("rd_scatter_crops");
}

int rd_points_to_rois(const float* points_in, float* points_out, float* rois, int32_t N, float pad_x, float pad_y, int32_t batch_index, void* stream) {
  if (N < 0) return fail("points_to_rois: negative count");
  if (N > 0 && (!points_in || !points_out || !rois)) return fail("points_to_rois: null pointer");
  rd::launch_points_to_rois(points_in, points_out, rois, N, pad_x, pad_y, batch_index, S(stream));
  return done("rd_points_to_rois");
}
int rd_boxes_to_rois(const float* boxes, float* rois, int32_t B, int32_t K, int32_t first_image, void* stream) {
  if (B < 0 || K < 0) return fail("boxes_to_rois: negative count");
  if ((int64_t)B * K > 0 && (!boxes || !rois)) return fail("boxes_to_rois: null pointer");
  rd::launch_boxes_to_rois(boxes, rois, B, K, first_image, S(stream));
  return done("rd_boxes_to_rois");
}
int rd_depth_quantize_u16(const float* z, uint16_t* out, int64_t n, float multiplier, void* stream) {
  if (n < 0) return fail("depth_quantize_u16: negative count");
  if (n > 0 && (!z || !out)) return fail("depth_quantize_u16: null pointer");
  rd::launch_depth_quantize_u16(z, out, n, multiplier, S(stream));
  return done("rd_depth_quantize_u16");
}
int rd_sum_f32(const float* x, int64_t n, double* out, void* stream) {
  if (!out || (n > 0 && !x) || n < 0) return fail("sum_f32: bad args");
  rd::launch_sum_f32(x, n, out, S(stream));
  return done("rd_sum_f32");
}

int rd_augment_gray_partials(const float* image, int32_t B, int32_t H, int32_t W, const float* params, int64_t* partial, void* stream) {
  if (!image || !params || !partial || B <= 0 || H <= 0 || W <= 0) return fail("augment_gray_partials: bad args");
  rd::launch_augment_gray_partials(image, B, H, W, params, (long long*)partial, S(stream));
  return done("rd_augment_gray_partials");
}
int rd_augment_image(const float* image, int32_t B, int32_t H, int32_t W, const float* params, const int64_t* partial, void* out_nhwc, int32_t dtype,
                     float scale, float shift, void* stream) {
  if (!image || !params || !partial || !out_nhwc || !dt_ok(dtype) || B <= 0 || H <= 0 || W <= 0) return fail("augment_image: bad args");
  RD_NS(dtype, launch_augment_image)(image, B, H, W, params, (const long long*)partial, out_nhwc, RD_DT(dtype), scale, shift, S(stream));
  return done("rd_augment_image");
}
int rd_augment_flip_labels(const float* labels_in, float* labels_out, int32_t B, int32_t K, int32_t ph, int32_t pw, float* boxes, const float* params,
                           float n_width, void* stream) {
  if (!labels_in || !labels_out || !params || labels_in == labels_out || B <= 0 || K <= 0) return fail("augment_flip_labels: bad args");
  rd::launch_augment_flip_labels(labels_in, labels_out, B, K, ph, pw, boxes, params, n_width, S(stream));
  return done("rd_augment_flip_labels");
}
int rd_augment_vflip_boxes(float* boxes, int32_t B, int32_t K, const float* params, float n_height, void* stream) {
  if (!boxes || !params || B <= 0) return fail("augment_vflip_boxes: bad args");
  if (K < 4) return fail("augment_vflip_boxes: the reference indexes boxes 1 and 3 of each sample (rcnet_transforms.py:213-217): K >= 4");
  rd::launch_augment_vflip_boxes(boxes, B, K, params, n_height, S(stream));
  return done("rd_augment_vflip_boxes");
}
int rd_crop_patches(const float* gt_padded, const float* points, float* crops, int32_t B, int32_t K, int32_t Hp, int32_t Wp, int32_t ph, int32_t pw,
                    void* stream) {
  if (!gt_padded || !points || !crops || B <= 0 || K <= 0 || (ph & 1) || (pw & 1)) return fail("crop_patches: bad args");
  rd::launch_crop_patches(gt_padded, points, crops, B, K, Hp, Wp, ph, pw, S(stream));
  return done("rd_crop_patches");
}

int rd_project_scatter(const float* points, int32_t n, int32_t stride, const double* t_camera_pcl, const double* projection, int32_t H, int32_t W,
                       double min_depth, double max_depth, float* depth_map, float* kept_points, int32_t* n_kept, void* stream) {
  if (n < 0 || stride < 3 || H <= 0 || W <= 0 || !t_camera_pcl || !projection || !depth_map) return fail("project_scatter: bad args");
  if (n > 0 && !points) return fail("project_scatter: null points");
  if ((kept_points != nullptr) != (n_kept != nullptr)) return fail("project_scatter: kept_points and n_kept go together");
  rd::launch_project_scatter(points, n, stride, t_camera_pcl, projection, H, W, min_depth, max_depth, depth_map, kept_points, n_kept, S(stream));
  return done("rd_project_scatter");
}

int rd_tri_raster(const int32_t* simplices, const int32_t* point_row, const int32_t* point_col, const double* values, int32_t n_simplices,
                  int32_t H, int32_t W, double fill_value, int32_t* owner_workspace, double* out, void* stream) {
  if (n_simplices < 0 || H <= 0 || W <= 0 || !owner_workspace || !out) return fail("tri_raster: bad args");
  if (n_simplices > 0 && (!simplices || !point_row || !point_col || !values)) return fail("tri_raster: null pointer");
  rd::launch_tri_raster(simplices, point_row, point_col, values, n_simplices, H, W, fill_value, owner_workspace, out, S(stream));
  return done("rd_tri_raster");
}

int rd_nearest_knot(const int32_t* point_row, const int32_t* point_col, const double* values, int32_t n_points, int32_t H, int32_t W,
                    double fill_value, double* out, void* stream) {
  if (n_points < 0 || H <= 0 || W <= 0 || !out) return fail("nearest_knot: bad args");
  if (n_points > 0 && (!point_row || !point_col || !values)) return fail("nearest_knot: null pointer");
  rd::launch_nearest_knot(point_row, point_col, values, n_points, H, W, fill_value, out, S(stream));
  return done("rd_nearest_knot");
}
int rd_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1, float b2, float eps, float wd,
                 int64_t step, float gscale, void* stream) {
  if (!p || !g || !m || !v) return fail("adam: null pointer");
  if (step < 1) return fail("adam: step must be >= 1");
  if (n == 0) return 0;
  double bc1 = 1.0 - pow((double)b1, (double)step), bc2 = 1.0 - pow((double)b2, (double)step);
  rd::launch_adam(p, g, m, v, n, lr, b1, b2, eps, wd, (float)bc1, (float)sqrt(bc2), gscale, S(stream));
  return done("rd_adam_step");
}
int rd_grad_finite_check(const float* g, int64_t n, int32_t* flag, void* stream) {
  if (!g || !flag) return fail("grad_finite_check: null pointer");
  if (n > 0) rd::launch_grad_finite(g, n, flag, S(stream));
  return done("rd_grad_finite_check");
}
int rd_adam_step_guarded(float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1, float b2, float eps, float wd,
                         int64_t step, float gscale, const int32_t* skip_flag, void* stream) {
  if (!p || !g || !m || !v || !skip_flag) return fail("adam_guarded: null pointer");
  if (step < 1) return fail("adam_guarded: step must be >= 1");
  if (n == 0) return 0;
  double bc1 = 1.0 - pow((double)b1, (double)step), bc2 = 1.0 - pow((double)b2, (double)step);
  rd::launch_adam(p, g, m, v, n, lr, b1, b2, eps, wd, (float)bc1, (float)sqrt(bc2), gscale, S(stream), skip_flag);
  return done("rd_adam_step_guarded");
}
int rd_adam_skip_count(int32_t* flag, void* stream) {
  if (!flag) return fail("adam_skip_count: null pointer");
  rd::launch_adam_skip_count(flag, S(stream));
  return done("rd_adam_skip_count");
}


// ---- Scale Map Learner ------------------------------------------------------------------------------------------------
int32_t rd_dw_rows(int64_t pixels, int32_t C) { return rd::dw_rows(pixels, C); }
int rd_dwconv_fwd(const void* x, const float* w, void* y, int32_t N, int32_t H, int32_t W, int32_t C, int32_t OH, int32_t OW, int32_t k,
                  int32_t s, int32_t p, int32_t dtype, void* stream) {
  if (!x || !w || !y || !dt_ok(dtype) || k > 5 || k < 1) return fail("dwconv_fwd: bad args");
  RD_NS(dtype, launch_dwconv_fwd)(x, w, y, N, H, W, C, OH, OW, k, s, p, RD_DT(dtype), S(stream), nullptr);
  return done("rd_dwconv_fwd");
}
int32_t rd_dwconv_stats_rows(int32_t N, int32_t OH, int32_t OW, int32_t C, int32_t k, int32_t s) { return rd::dwconv_stats_rows(N, OH, OW, C, k, s); }
int rd_dwconv_fwd_stats(const void* x, const float* w, void* y, float* stats, int32_t N, int32_t H, int32_t W, int32_t C, int32_t OH, int32_t OW,
                        int32_t k, int32_t s, int32_t p, int32_t dtype, void* stream) {
  if (!x || !w || !y || !stats || !dt_ok(dtype) || rd::dwconv_stats_rows(N, OH, OW, C, k, s) <= 0) return fail("dwconv_fwd_stats: bad args");
  RD_NS(dtype, launch_dwconv_fwd)(x, w, y, N, H, W, C, OH, OW, k, s, p, RD_DT(dtype), S(stream), stats);
  return done("rd_dwconv_fwd_stats");
}
int rd_dwconv_dgrad(const void* dy, const float* w, void* dx, int32_t N, int32_t H, int32_t W, int32_t C, int32_t OH, int32_t OW, int32_t k,
                    int32_t s, int32_t p, int32_t dtype, void* stream) {
  if (!dy || !w || !dx || !dt_ok(dtype) || k > 5 || k < 1) return fail("dwconv_dgrad: bad args");
  RD_NS(dtype, launch_dwconv_dgrad)(dy, w, dx, N, H, W, C, OH, OW, k, s, p, RD_DT(dtype), S(stream));
  return done("rd_dwconv_dgrad");
}
int rd_dwconv_wgrad(const void* x, const void* dy, float* partial, float* dw, int32_t accumulate, int32_t N, int32_t H, int32_t W, int32_t C,
                    int32_t OH, int32_t OW, int32_t k, int32_t s, int32_t p, int32_t dtype, void* stream) {
  if (!x || !dy || !partial || !dw || !dt_ok(dtype) || k > 5 || k < 1) return fail("dwconv_wgrad: bad args");
  RD_NS(dtype, launch_dwconv_wgrad)(x, dy, partial, dw, accumulate, N, H, W, C, OH, OW, k, s, p, RD_DT(dtype), S(stream), nullptr);
  return done("rd_dwconv_wgrad");
}
static_assert(sizeof(rd_dw_wgrad_item) == sizeof(rdt::DwWgradItem), "rd_dw_wgrad_item layout");
int rd_dwconv_wgrad_partial(const void* x, const void* dy, float* partial, float* dw, int32_t accumulate, int32_t N, int32_t H, int32_t W,
                            int32_t C, int32_t OH, int32_t OW, int32_t k, int32_t s, int32_t p, int32_t dtype, rd_dw_wgrad_item* item,
                            void* stream) {
  if (!x || !dy || !partial || !dw || !item || !dt_ok(dtype) || k > 5 || k < 1) return fail("dwconv_wgrad_partial: bad args");
  RD_NS(dtype, launch_dwconv_wgrad)(x, dy, partial, dw, accumulate, N, H, W, C, OH, OW, k, s, p, RD_DT(dtype), S(stream),
                                    reinterpret_cast<rdt::DwWgradItem*>(item));
  return done("rd_dwconv_wgrad_partial");
}
int rd_dw_wgrad_finalize_batch(const rd_dw_wgrad_item* items, int32_t n, void* stream) {
  if (n <= 0) return 0;
  if (!items) return fail("dw_wgrad_finalize_batch: bad args");
  for (int i = 0; i < n; i++)
    if (!items[i].partial || !items[i].dw || items[i].rows <= 0 || items[i].C <= 0 || items[i].KK <= 0) return fail("dw_wgrad_finalize_batch: item not filled by rd_dwconv_wgrad_partial");
  rd::launch_dw_wgrad_finalize_batch(reinterpret_cast<const rdt::DwWgradItem*>(items), n, S(stream));
  return done("rd_dw_wgrad_finalize_batch");
}
int rd_bn_stats(const void* y, float* partial, int64_t pixels, int32_t C, int32_t dtype, void* stream) {
  if (!y || !partial || !dt_ok(dtype)) return fail("bn_stats: bad args");
  RD_NS(dtype, launch_bn_stats)(y, partial, pixels, C, RD_DT(dtype), S(stream));
  return done("rd_bn_stats");
}
int rd_bilinear_fwd(const void* x, void* y, int32_t N, int32_t H, int32_t W, int32_t C, int32_t OH, int32_t OW, int32_t align, int32_t dtype,
                    void* stream) {
  if (!x || !y || !dt_ok(dtype)) return fail("bilinear_fwd: bad args");
  RD_NS(dtype, launch_bilinear)(x, y, N, H, W, C, OH, OW, align, 0, RD_DT(dtype), S(stream));
  return done("rd_bilinear_fwd");
}
int rd_bilinear_bwd(const void* dy, void* dx, int32_t N, int32_t H, int32_t W, int32_t C, int32_t OH, int32_t OW, int32_t align, int32_t dtype,
                    void* stream) {
  if (!dy || !dx || !dt_ok(dtype)) return fail("bilinear_bwd: bad args");
  RD_NS(dtype, launch_bilinear)(dy, dx, N, H, W, C, OH, OW, align, 1, RD_DT(dtype), S(stream));
  return done("rd_bilinear_bwd");
}
int rd_sml_head_fwd(const void* out, const float* d, float* pred, int64_t n, float hi, float lo, int32_t dtype, void* stream) {
  if (!out || !d || !pred || !dt_ok(dtype)) return fail("sml_head_fwd: bad args");
  RD_NS(dtype, launch_sml_head_fwd)(out, d, pred, n, hi, lo, RD_DT(dtype), S(stream));
  return done("rd_sml_head_fwd");
}
int rd_sml_head_bwd(const void* out, const float* d, const float* dpred, void* dout, int64_t n, float hi, float lo, int32_t dtype, void* stream) {
  if (!out || !d || !dpred || !dout || !dt_ok(dtype)) return fail("sml_head_bwd: bad args");
  RD_NS(dtype, launch_sml_head_bwd)(out, d, dpred, dout, n, hi, lo, RD_DT(dtype), S(stream));
  return done("rd_sml_head_bwd");
}
int rd_reciprocal(const float* x, const float* dy, float* out, int64_t n, void* stream) {
  if (!x || !out) return fail("reciprocal: null pointer");
  if (n == 0) return 0;
  rd::launch_reciprocal(x, dy, out, n, S(stream));
  return done("rd_reciprocal");
}
int rd_sml_scale_align(const float* mono, const float* sparse, int32_t B, int32_t HW, float dmin, float dmax, float lo, float hi, float* scale,
                       int32_t* nvalid, void* stream) {
  if (!mono || !sparse || !scale || !nvalid) return fail("sml_scale_align: null pointer");
  rd::launch_sml_scale_align(mono, sparse, B, HW, dmin, dmax, lo, hi, scale, nvalid, S(stream));
  return done("rd_sml_scale_align");
}
int rd_sml_scale_shift_ls(const float* mono, const float* sparse, int32_t B, int32_t HW, float dmin, float dmax, float* scale, float* shift,
                          int32_t* nvalid, void* stream) {
  if (!mono || !sparse || !scale || !shift || !nvalid) return fail("sml_scale_shift_ls: null pointer");
  rd::launch_sml_scale_shift_ls(mono, sparse, B, HW, dmin, dmax, scale, shift, nvalid, S(stream));
  return done("rd_sml_scale_shift_ls");
}
int rd_sml_build_inputs(const float* image, const float* mono, const float* sparse, const float* rcnet, const float* scale, const float* shift,
                        float* mm, int32_t B, int32_t H, int32_t W, int32_t h, int32_t w, float dmin, float dmax, float hi, float lo,
                        int32_t use_rcnet, float m0, float s0, float m1, float s1, float* x, float* d, void* stream) {
  if (!image || !mono || !sparse || !scale || !mm || !x || !d) return fail("sml_build_inputs: null pointer");
  if (use_rcnet && !rcnet) return fail("sml_build_inputs: use_rcnet without rcnet depth");
  rd::launch_sml_build_inputs(image, mono, sparse, rcnet, scale, shift, mm, B, H, W, h, w, dmin, dmax, hi, lo, use_rcnet, m0, s0, m1, s1, x, d,
                              S(stream));
  return done("rd_sml_build_inputs");
}
int32_t rd_outlier_parts(int64_t n) { return rd::outlier_parts(n); }
int rd_outlier_removal(const float* depth, float* partial, float* out, int32_t N, int32_t H, int32_t W, int32_t k, float thr, void* stream) {
  if (!depth || !partial || !out) return fail("outlier_removal: null pointer");
  rd::launch_outlier_removal(depth, partial, out, N, H, W, k, thr, S(stream));
  return done("rd_outlier_removal");
}
int32_t rd_sml_loss_rows(int64_t n) { return rd::sml_loss_rows(n); }
int rd_sml_loss_fwd(const float* pred, const float* image, const float* gi, const float* gs, const float* weights, int32_t N, int32_t H, int32_t W,
                    int32_t fs, float w_lidar, float w_smooth, float w_edge, float* gfx, float* gfy, double* partial, float* info, void* stream) {
  if (!pred || !image || !gi || !gs || !gfx || !gfy || !partial || !info) return fail("sml_loss_fwd: null pointer");
  if (fs < 3 || fs > 9 || !(fs & 1)) return fail("sml_loss: filter size must be odd in 3..9");
  rd::launch_sml_loss_fwd(pred, image, gi, gs, weights, N, H, W, fs, w_lidar > 0.f ? 1 : 0, w_lidar, w_smooth, w_edge, gfx, gfy, partial, info, S(stream));
  return done("rd_sml_loss_fwd");
}
int rd_sml_loss_bwd(const float* pred, const float* gi, const float* gs, const float* gfx, const float* gfy, const float* info, const float* dloss,
                    int32_t N, int32_t H, int32_t W, int32_t fs, float w_lidar, float w_smooth, float* dpred, void* stream) {
  if (!pred || !gi || !gs || !gfx || !gfy || !info || !dloss || !dpred) return fail("sml_loss_bwd: null pointer");
  rd::launch_sml_loss_bwd(pred, gi, gs, gfx, gfy, info, dloss, N, H, W, fs, w_lidar > 0.f ? 1 : 0, w_lidar, w_smooth, dpred, S(stream));
  return done("rd_sml_loss_bwd");
}
int rd_sml_loss_fwd_kind(const float* pred, const float* image, const float* gi, const float* gs, const float* weights, int32_t N, int32_t H, int32_t W,
                         int32_t fs, int32_t loss_kind, float w_lidar, float w_smooth, float w_edge, float* gfx, float* gfy, double* partial, float* info,
                         void* stream) {
  if (!pred || !image || !gi || !gs || !gfx || !gfy || !partial || !info) return fail("sml_loss_fwd: null pointer");
  if (fs < 3 || fs > 9 || !(fs & 1)) return fail("sml_loss: filter size must be odd in 3..9");
  if (loss_kind < 0 || loss_kind > 2) return fail("sml_loss: loss_kind must be 0 ('l1'), 1 ('l2') or 2 ('smoothl1')");
  if (w_edge > 0.f && !(w_smooth > 0.f)) return fail("sml_loss: w_edge > 0 needs w_smoothness > 0 (the gradient fields are stored per unit of w_smoothness)");
  rd::launch_sml_loss_fwd(pred, image, gi, gs, weights, N, H, W, fs, (w_lidar > 0.f ? 1 : 0) | (loss_kind << 1), w_lidar, w_smooth, w_edge, gfx, gfy, partial, info, S(stream));
  return done("rd_sml_loss_fwd_kind");
}
int rd_sml_loss_bwd_kind(const float* pred, const float* gi, const float* gs, const float* gfx, const float* gfy, const float* info, const float* dloss,
                         int32_t N, int32_t H, int32_t W, int32_t fs, int32_t loss_kind, float w_lidar, float w_smooth, float* dpred, void* stream) {
  if (!pred || !gi || !gs || !gfx || !gfy || !info || !dloss || !dpred) return fail("sml_loss_bwd: null pointer");
  if (loss_kind < 0 || loss_kind > 2) return fail("sml_loss: loss_kind must be 0 ('l1'), 1 ('l2') or 2 ('smoothl1')");
  rd::launch_sml_loss_bwd(pred, gi, gs, gfx, gfy, info, dloss, N, H, W, fs, (w_lidar > 0.f ? 1 : 0) | (loss_kind << 1), w_lidar, w_smooth, dpred, S(stream));
  return done("rd_sml_loss_bwd_kind");
}
int rd_bicubic_resize(const float* x, float* y, int32_t N, int32_t H, int32_t W, int32_t OH, int32_t OW, void* stream) {
  if (!x || !y) return fail("bicubic_resize: null pointer");
  rd::launch_bicubic(x, y, N, H, W, OH, OW, S(stream));
  return done("rd_bicubic_resize");
}
int rd_depth_metrics(const float* out, const float* gt, int32_t N, int32_t HW, float dmin, float dmax, double* res, void* stream) {
  if (!out || !gt || !res) return fail("depth_metrics: null pointer");
  rd::launch_depth_metrics(out, gt, N, HW, dmin, dmax, res, S(stream));
  return done("rd_depth_metrics");
}

}  // extern "C"
