/* ORACLE (test infrastructure only -- never imported by riders_amd; see oracle/README.md).
 *
 * CPU restatement of torchvision.ops.roi_pool as called at RCNet/networks.py:418-433.  torchvision
 * (pinned 0.14.0 in the reference's environment.yaml:9) is NOT part of /root/reference, so this follows
 * the published algorithm of torchvision/csrc/ops/cpu/roi_pool_kernel.cpp (roi_pool_forward_kernel_impl /
 * roi_pool_backward_kernel_impl), restated in SURVEY.md Appendix A:
 *   round() (C, half away from zero) of the scaled corners, +1 extents, floor/ceil bin edges, clamp to the
 *   map, empty bin -> 0 and argmax -1, strict '>' scan in row-major order (first maximum wins).
 * "parity unpinned" against upstream binaries: the reference holds no test for this op; the contract is the
 * hand-derived known-answer vectors in tests/golden/roi_pool_kat.json (tests/test_oracle_golden.py).
 *
 * Layout here is NCHW (as the reference sees it); rois are (R,5) = (batch, x1, y1, x2, y2).
 */
#include <float.h>
#include <math.h>
#include <stdint.h>

void oracle_roi_pool_fwd(const float* in, const float* rois, float* out, int32_t* argmax, int R, int N, int C, int H,
                         int W, int PH, int PW, float scale) {
  for (int r = 0; r < R; r++) {
    const float* roi = rois + 5 * r;
    int b = (int)roi[0];
    int sw = (int)roundf(roi[1] * scale), sh = (int)roundf(roi[2] * scale);
    int ew = (int)roundf(roi[3] * scale), eh = (int)roundf(roi[4] * scale);
    int rw = ew - sw + 1 > 1 ? ew - sw + 1 : 1;
    int rh = eh - sh + 1 > 1 ? eh - sh + 1 : 1;
    float bh = (float)rh / (float)PH, bw = (float)rw / (float)PW;
    for (int ph = 0; ph < PH; ph++)
      for (int pw = 0; pw < PW; pw++) {
        int hs = (int)floorf((float)ph * bh), he = (int)ceilf((float)(ph + 1) * bh);
        int ws = (int)floorf((float)pw * bw), we = (int)ceilf((float)(pw + 1) * bw);
        hs = hs + sh < 0 ? 0 : (hs + sh > H ? H : hs + sh);
        he = he + sh < 0 ? 0 : (he + sh > H ? H : he + sh);
        ws = ws + sw < 0 ? 0 : (ws + sw > W ? W : ws + sw);
        we = we + sw < 0 ? 0 : (we + sw > W ? W : we + sw);
        int empty = (he <= hs) || (we <= ws);
        for (int c = 0; c < C; c++) {
          float best = empty ? 0.f : -FLT_MAX;
          int bi = -1;
          const float* p = in + ((int64_t)b * C + c) * H * W;
          for (int h = hs; h < he; h++)
            for (int w = ws; w < we; w++) {
              float v = p[h * W + w];
              if (v > best) { best = v; bi = h * W + w; }
            }
          int64_t o = (((int64_t)r * C + c) * PH + ph) * PW + pw;
          out[o] = best;
          argmax[o] = bi;
        }
      }
  }
}

void oracle_roi_pool_bwd(const float* dout, const float* rois, const int32_t* argmax, float* din, int R, int N, int C,
                         int H, int W, int PH, int PW) {
  for (int64_t i = 0; i < (int64_t)N * C * H * W; i++) din[i] = 0.f;
  for (int r = 0; r < R; r++) {
    int b = (int)rois[5 * r];
    for (int c = 0; c < C; c++)
      for (int k = 0; k < PH * PW; k++) {
        int64_t o = ((int64_t)r * C + c) * PH * PW + k;
        int a = argmax[o];
        if (a >= 0) din[((int64_t)b * C + c) * H * W + a] += dout[o];
      }
  }
}
