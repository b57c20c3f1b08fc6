// Launcher declarations (NO include guard: rd_api.cpp includes this file twice, once per precision namespace -- see rd_kernels.h).
namespace rd {
using namespace rdt;


// rd_conv.hip
int conv_rows_pad(int rows);
int conv_kpad(int K, int dtype);
int64_t conv_packed_elems(int rows, int K, int dtype);
int conv_block_pixels(int M, int Cout);
int wgrad_nsplit(int M, int K, int Cout);
int wgrad_slabs(int M, int K, int Cout);
int conv_stats_rows(const ConvArgs& a, int dtype);
void launch_conv(const ConvArgs& a, int dtype, hipStream_t st);
const char* conv_kernel_name(const ConvArgs& a, int dtype);
bool conv_pool2_ok(const ConvArgs& a, int dtype);
bool conv_d2s_ok(const ConvArgs& a, int dtype);
bool conv_s2d_ok(const ConvArgs& a, int dtype);
bool conv_add_ok(const ConvArgs& a, int dtype);
bool conv_in_affine_ok(const ConvArgs& a, int dtype);
bool conv_bn_bwd_ok(const ConvArgs& a, int dtype);
bool wgrad_in_affine_ok(const WgradArgs& a, int dtype);
const char* wgrad_kernel_name(const WgradArgs& a, int dtype);
// rd_conv3x3.hip
bool conv3x3_ok(const ConvArgs& a, int dtype);
int conv3x3_tiles(const ConvArgs& a);
void launch_conv3x3(const ConvArgs& a, int dtype, hipStream_t st);
bool conv3x3_small_ok(const ConvArgs& a, int dtype);
void launch_conv3x3_small(const ConvArgs& a, int dtype, hipStream_t st);
const char* conv3x3_small_name(const ConvArgs& a, int dtype);
const char* conv3x3_patch_name(const ConvArgs& a, int dtype);
int conv3x3_small_blocks(const ConvArgs& a, int dtype);
bool conv1x1_direct_ok(const ConvArgs& a, int dtype);
int conv1x1_direct_rows(const ConvArgs& a);
void launch_conv1x1_direct(const ConvArgs& a, int dtype, hipStream_t st);
// rd_conv3x3_frag.hip
bool conv3x3_frag_ok(const ConvArgs& a, int dtype);
int conv3x3_frag_tiles(const ConvArgs& a, int dtype);
bool conv3x3_frag_is32(const ConvArgs& a, int dtype);
bool conv3x3_frag_d2s_ok(const ConvArgs& a, int dtype);
bool conv3x3_frag_s2d_ok(const ConvArgs& a, int dtype);
int conv3x3_frag_blocks(const ConvArgs& a, int dtype);
void launch_conv3x3_frag(const ConvArgs& a, int dtype, hipStream_t st);
const char* conv3x3_frag_name(const ConvArgs& a, int dtype);
bool conv_few_ok(const ConvArgs& a);
int conv_few_blocks(const ConvArgs& a);
void launch_conv_few(const ConvArgs& a, int dtype, hipStream_t st);
bool conv3x3_c1_ok(const ConvArgs& a);
void launch_conv3x3_c1(const ConvArgs& a, int dtype, hipStream_t st);
void launch_pack_weights(const float* w, void* out, int Cout, int Cin, int KH, int KW, int mode, int dtype, hipStream_t st, int CinSrc = 0);
void launch_pack_weights_batch(const void* items, int n, hipStream_t st);
void launch_wgrad(WgradArgs a, int dtype, float* dw, int accumulate, hipStream_t st, WgradReduceItem* defer);
bool wgrad_streams(const WgradArgs& a);
void launch_wgrad_reduce_batch(const WgradReduceItem* items, int n, hipStream_t st);

// rd_wgrad3x3.hip
bool wgrad3x3_tr_ok(const WgradArgs& a, int dtype);
bool wgrad3x3_tr_affine_ok(const WgradArgs& a, int dtype);
int wgrad3x3_tr_blocks(const WgradArgs& a);
void launch_wgrad3x3_tr(const WgradArgs& a, hipStream_t st);
const char* wgrad3x3_tr_name(const WgradArgs& a);

// rd_linear_wgrad.hip (descriptors mirror rd_lwg_gemm / rd_lwg_reduce of the C ABI; passed BY VALUE in kernel arguments)
void launch_linear_wgrad_batch(const LwgGemm* gemms, int n_gemm, const LwgReduce* reds, int n_red, int dtype, hipStream_t st);

// rd_loftr.hip (structs mirror rd_loftr_weights / rd_loftr_saved / rd_loftr_grads of the C ABI)
void launch_loftr_layer_fwd(const void* x, const void* src, const LoftrW& w, void* out, const LoftrSaved& sv, int N, int L, int S,
                            float eps_attn, float eps_ln, int dtype, hipStream_t st);
void launch_loftr_layer_bwd(const void* x, const void* src, const LoftrW& w, const LoftrSaved& sv, const LoftrGrads& gr, int N, int L,
                            int S, float eps_attn, int dtype, hipStream_t st);

// rd_norm.hip
void launch_bn_finalize(const float* partial, int rows, int C, double count, const float* gamma, const float* beta,
                        float eps, float momentum, int training, float* running_mean, float* running_var,
                        float* mean, float* rstd, float* scale, float* shift, hipStream_t st);
void launch_affine_act(const void* y, const float* scale, const float* shift, const void* res, void* out, int64_t pixels,
                       int C, int act, float slope, int dtype, hipStream_t st);
bool affine_act_add_ok(int C, int dtype);
void launch_affine_act_add(const void* y, const float* scale, const float* shift, int act1, float slope1, const void* res, void* out,
                           int64_t pixels, int C, int act2, float slope2, int dtype, hipStream_t st);
const char* bn_kernel_name(int which, int C, int dtype, int act, int flag);
int bn_bwd_rows(int64_t pixels, int C);
void launch_bn_bwd_reduce(const void* dz, const void* z, const void* y, const float* mean, const float* rstd,
                          float* partial, int64_t pixels, int C, int act, float slope, int dtype, hipStream_t st,
                          const float* scale = nullptr, const float* shift = nullptr);
void launch_bn_bwd_finalize(const float* partial, int rows, int C, double count, float* dgamma, float* dbeta,
                            int accumulate, float* c1, float* c2, hipStream_t st, int row_pitch = 0);
void launch_bn_bwd_apply(const void* dz, const void* z, const void* y, const float* mean, const float* rstd,
                         const float* scale, const float* c1, const float* c2, void* dy, void* dres, int64_t pixels,
                         int C, int act, float slope, int dtype, hipStream_t st, const float* shift = nullptr);
void launch_act_bwd(const void* dz, const void* z, void* dx, int64_t n, int act, float slope, int dtype, hipStream_t st);
void launch_colsum(const void* x, float* partial, float* out, int accumulate, int64_t rows, int C, int dtype, hipStream_t st);
int colsum_rows(int64_t rows, int C);
void launch_colsum_finalize_batch(const ColsumItem* items, int n, hipStream_t st);
void launch_ln_grad_batch(const LnGradItem* items, int n, hipStream_t st);
void launch_layernorm_fwd(const void* x, const float* gamma, const float* beta, const void* res, void* out, float* mean,
                          float* rstd, int64_t rows, int C, float eps, int dtype, hipStream_t st);
int layernorm_bwd_rows(int64_t rows);
void launch_layernorm_bwd(const void* dout, const void* x, const float* gamma, const float* mean, const float* rstd,
                          void* dx, float* partial, float* dgamma, float* dbeta, int accumulate, int64_t rows, int C,
                          int dtype, hipStream_t st);

// rd_head.hip (decoder head: BatchNorm + activation of the last decoder convolution fused with the one-channel 3x3 output convolution)
bool bn_head_ok(int N, int H, int W, int C, int dtype);
int bn_head_rows(int N, int H, int W);
void launch_bn_head_fwd(const void* y, const float* scale, const float* shift, int act, float slope, const float* w, void* logits, int N, int H, int W,
                        int dtype, hipStream_t st);
void launch_bn_head_bwd_reduce(const void* dl, const void* y, const float* mean, const float* rstd, const float* scale, const float* shift, int act,
                               float slope, const float* w, float* partial, int N, int H, int W, int dtype, hipStream_t st);
void launch_bn_head_bwd_apply(const void* dl, const void* y, const float* mean, const float* rstd, const float* scale, const float* shift, int act,
                              float slope, const float* w, const float* partial, int rows, float* coef, float* dgamma, float* dbeta, int bn_acc,
                              float* dw, int w_acc, void* dy, int N, int H, int W, int dtype, hipStream_t st);
const char* bn_head_kernel_name(int which, int dtype, int act);

// rd_pool.hip
void launch_maxpool_fwd(const void* x, void* out, unsigned char* arg, int N, int H, int W, int C, int OH, int OW, int k,
                        int s, int p, int dtype, hipStream_t st);
void launch_maxpool_bwd(const void* dout, const unsigned char* arg, void* dx, int N, int H, int W, int C, int OH, int OW,
                        int k, int s, int p, int dtype, hipStream_t st);
void launch_roi_pool_fwd(const void* x, const float* rois, void* out, int* argmax, int R, int N, int H, int W, int C,
                         int PH, int PW, float scale, int dtype, hipStream_t st);
void launch_roi_pool_bwd(const void* dout, const float* rois, const int* argmax, float* dx_f32, int R, int N, int H, int W,
                         int C, int PH, int PW, int dtype, hipStream_t st);

void launch_roi_pool_bwd_tile(const void* dout, const float* rois, const int* argmax, void* dx, int R, int N, int H, int W, int C,
                              int PH, int PW, float scale, int dtype, hipStream_t st);
void launch_roi_pool_bwd_gather(const void* dout, const float* rois, const int* argmax, void* dx, int R, int N, int H, int W, int C,
                                int PH, int PW, float scale, int dtype, hipStream_t st);

void launch_roi_pool_fwd_u8(const void* x, const float* rois, void* out, unsigned char* argmax, int* flag, int R, int N, int H, int W, int C,
                            int PH, int PW, float scale, int dtype, hipStream_t st);
void launch_roi_pool_bwd_u8(const void* dout, const float* rois, const unsigned char* argmax, const int* flag, float* dx_f32, int R, int N, int H,
                            int W, int C, int PH, int PW, float scale, int dtype, hipStream_t st);
void launch_roi_pool_bwd_gather_u8(const void* dout, const float* rois, const unsigned char* argmax, const int* flag, void* dx, int R, int N, int H,
                                   int W, int C, int PH, int PW, float scale, int dtype, hipStream_t st);

// rd_elementwise.hip
void launch_pad_channels(const void* src, void* dst, int64_t rows, int C, int Cpad, int dtype, hipStream_t st);
void launch_unpad_weight_grad(const float* dwp, float* dw, int Cout, int Cin, int CinPad, int taps, int accumulate, hipStream_t st);
void launch_cast(const void* src, void* dst, int64_t n, int src_dtype, int dst_dtype, float scale, hipStream_t st);
void launch_add(const void* a, const void* b, void* out, int64_t n, int dtype, hipStream_t st);
void launch_nchw_to_nhwc(const void* src, void* dst, int N, int C, int H, int W, int src_dtype, int dst_dtype, float scale, hipStream_t st);
void launch_nhwc_to_nchw(const void* src, void* dst, int N, int C, int H, int W, int src_dtype, int dst_dtype, hipStream_t st);
void launch_transpose_last2(const void* src, void* dst, int64_t B, int R, int Ccols, int dtype, hipStream_t st);
void launch_concat2(const void* a, const void* b, void* out, int64_t rows, int Ca, int Cb, int dtype, hipStream_t st);
void launch_split2(const void* in, void* a, void* b, int64_t rows, int Ca, int Cb, int dtype, hipStream_t st);
void launch_upsample_nearest_bwd(const void* dy, void* dx, int N, int Hs, int Ws, int Hv, int Wv, int C, int dtype, hipStream_t st);
void launch_upsample_nearest_fwd(const void* x, void* y, int N, int Hs, int Ws, int Hv, int Wv, int C, int dtype, hipStream_t st);

// rd_attention.hip
void launch_linear_attention_fwd(const void* q, const void* k, const void* v, void* out, int N, int L, int S, int H,
                                 int ldq, int ldk, int ldv, int ldo, float eps, int dtype, hipStream_t st);
void launch_linear_attention_bwd(const void* q, const void* k, const void* v, const void* dout, void* dq, void* dk,
                                 void* dv, int N, int L, int S, int H, int ldq, int ldk, int ldv, int ldo, float eps,
                                 int dtype, hipStream_t st);

// rd_loss.hip
void launch_rcnet_labels(const float* gt, const float* points, float* label, float* valid, int R, int HW, float thr,
                         int all_valid, hipStream_t st);
int bce_rows(int64_t n);
void launch_bce_fwd(const void* logits, const float* label, const float* valid, float pos_weight, float* partial,
                    float* loss, float* sums, int64_t n, int dtype, hipStream_t st);
void launch_bce_bwd(const void* logits, const float* label, const float* valid, float pos_weight, const float* sums,
                    const float* dloss, void* dlogits, int64_t n, int dtype, hipStream_t st);
void launch_sigmoid(const void* x, void* y, int64_t n, int dtype, hipStream_t st);
void launch_scatter_crops(const void* crops, const float* points, float* depth, float* response, int Ncrop, int PH, int PW,
                          int H, int W, float thr, int dtype, hipStream_t st);

void launch_points_to_rois(const float* pin, float* pout, float* rois, int N, float pad_x, float pad_y, int batch_index, hipStream_t st);
void launch_boxes_to_rois(const float* boxes, float* rois, int B, int K, int first_image, hipStream_t st);
void launch_depth_quantize_u16(const float* z, unsigned short* out, int64_t n, float multiplier, hipStream_t st);
void launch_sum_f32(const float* x, int64_t n, double* out, hipStream_t st);

// rd_augment.hip
void launch_augment_gray_partials(const float* image, int B, int H, int W, const float* params, long long* partial, hipStream_t st);
void launch_augment_image(const float* image, int B, int H, int W, const float* params, const long long* partial, void* out, int dtype, float scale, float shift, hipStream_t st);
void launch_augment_flip_labels(const float* lin, float* lout, int B, int K, int ph, int pw, float* boxes, const float* params, float n_width, hipStream_t st);
void launch_crop_patches(const float* gt, const float* points, float* crops, int B, int K, int Hp, int Wp, int ph, int pw, hipStream_t st);

void launch_tri_raster(const int* tri, const int* prow, const int* pcol, const double* values, int M, int H, int W, double fill, int* owner, double* out, hipStream_t st);
void launch_nearest_knot(const int* prow, const int* pcol, const double* values, int K, int H, int W, double fill, double* out, hipStream_t st);
void launch_project_scatter(const float* pts, int n, int stride, const double* T, const double* P, int H, int W, double dmin, double dmax, float* depth_map, float* kept, int* nkept, hipStream_t st);

// rd_optim.hip
void launch_adam(float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1, float b2, float eps,
                 float wd, float bc1, float bc2, float gscale, hipStream_t st, const int* skip = nullptr);
void launch_grad_finite(const float* g, int64_t n, int* flag, hipStream_t st);
void launch_adam_skip_count(int* flag, hipStream_t st);


// rd_dwconv.hip
int dw_rows(int64_t pixels, int C);
int dwconv_stats_rows(int N, int OH, int OW, int C, int k, int s);
void launch_dwconv_fwd(const void* x, const float* w, void* y, int N, int H, int W, int C, int OH, int OW, int k, int s, int p, int dtype, hipStream_t st, float* stats);
void launch_dwconv_dgrad(const void* dy, const float* w, void* dx, int N, int H, int W, int C, int OH, int OW, int k, int s, int p, int dtype, hipStream_t st);
void launch_dwconv_wgrad(const void* x, const void* dy, float* partial, float* dw, int accumulate, int N, int H, int W, int C, int OH, int OW, int k, int s, int p, int dtype, hipStream_t st, DwWgradItem* item = nullptr);
void launch_dw_wgrad_finalize_batch(const DwWgradItem* items, int n, hipStream_t st);
void launch_bn_stats(const void* y, float* partial, int64_t pixels, int C, int dtype, hipStream_t st);
void launch_bilinear(const void* x, void* y, int N, int H, int W, int C, int OH, int OW, int align, int backward, int dtype, hipStream_t st);
void launch_sml_head_fwd(const void* out, const float* d, float* pred, int64_t n, float hi, float lo, int dtype, hipStream_t st);
void launch_sml_head_bwd(const void* out, const float* d, const float* dpred, void* dout, int64_t n, float hi, float lo, int dtype, hipStream_t st);
void launch_reciprocal(const float* x, const float* dy, float* out, int64_t n, hipStream_t st);

// rd_sml.hip
void launch_sml_scale_align(const float* mono, const float* sparse, int B, int HW, float dmin, float dmax, float lo, float hi, float* scale, int* nvalid, hipStream_t st);
void launch_sml_scale_shift_ls(const float* mono, const float* sparse, int B, int HW, float dmin, float dmax, float* scale, float* shift, int* nvalid, hipStream_t st);
void launch_sml_build_inputs(const float* image, const float* mono, const float* sparse, const float* rcnet, const float* scale, const float* shift, float* mm, int B, int H, int W, int h, int w, float dmin, float dmax, float hi, float lo, int use_rcnet, float m0, float s0, float m1, float s1, float* x, float* d, hipStream_t st);
int outlier_parts(int64_t n);
void launch_outlier_removal(const float* depth, float* partial, float* out, int N, int H, int W, int k, float thr, hipStream_t st);
int sml_loss_rows(int64_t n);
void launch_sml_loss_fwd(const float* pred, const float* image, const float* gi, const float* gs, const float* weights, int N, int H, int W, int fs, int mask_interp, float w_lidar, float w_smooth, float w_edge, float* gfx, float* gfy, double* partial, float* info, hipStream_t st);
void launch_sml_loss_bwd(const float* pred, const float* gi, const float* gs, const float* gfx, const float* gfy, const float* info, const float* dloss, int N, int H, int W, int fs, int mask_interp, float w_lidar, float w_smooth, float* dpred, hipStream_t st);
void launch_bicubic(const float* x, float* y, int N, int H, int W, int OH, int OW, hipStream_t st);
void launch_depth_metrics(const float* out, const float* gt, int N, int HW, float dmin, float dmax, double* res, hipStream_t st);

}  // namespace rd
