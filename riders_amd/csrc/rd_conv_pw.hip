// Pointwise (1x1, stride 1) convolutions on a few thousand pixels: a plain GEMM  out[M][Cout] = X[M][Cin] . W[Cout][Cin]^T  whose two
// operands are both k-contiguous, on launches that are either a latency chain or a partial-line store stream in the implicit-GEMM kernel.
//
// Reference: the expansion / projection convolutions of EfficientNet-Lite3's inverted-residual blocks in the Scale Map Learner's backbone
// (modules/midas/midas_net_custom.py:84-111 via geffnet `tf_efficientnet_lite3`, hubconf) on its /16 and /32 stages -- 136 -> 816 -> 136
// on 18x36 maps, 232 -> 1392 -> 232 on 9x18 at batch 16 -- the 1x1 `out_conv`s of the fusion blocks (modules/midas/blocks.py:168-172), and
// their data gradients (the same GEMM with the transposed operand, + the residual branch's earlier gradient as addend).
//
// What was measured on them (tools/bench_pw.py, round 6): M = 2592, 1392 -> 232 took 19.6 us for 8.4 MB and 1.7 GFLOP -- 22 dependent
// 128-byte stages, each a global -> register -> LDS -> barrier round trip, on 164 blocks; M = 10368, 136 -> 816 took 22.5 us (34.5 with the
// addend) for a 17-MB output written as 8-byte pieces 32 bytes wide per pixel row and instruction (conv_epilogue_store_at: a lane owns 4
// channels of one pixel), the addend read the same way.
//
// Here (an experiment that pays on the long-K projections only, see "Where it is routed" below): NO staging and NO barrier in the main loop.  A wave owns a 64-pixel x 64-channel tile over a range of the K axis and loads its MFMA
// operands straight from global memory in fragment order (lane = 16 * k-group + row: 16 bytes of one row, both halves of the row's
// 128-byte line back to back), next chunk in flight while the current one is multiplied.  A block is KS x NT waves: KS waves split the K
// axis of one tile (long K, few tiles: the projections), NT waves take neighbouring channel tiles of the same pixels (short K, wide output:
// the expansions).  The epilogue restages the fp32 tile(s) through LDS so that 8 lanes hold one pixel's 64 consecutive channels: the K-split
// partials are summed in wave order (fixed: reproducible), bias / activation / addend applied, the result rounded ONCE and stored -- and
// the addend loaded -- as whole 128-byte lines; the BatchNorm (sum, sum^2) partials are taken over the stored values, one row per pixel tile.
// Blocks that share a pixel tile sit on the same XCD (its L2 serves the re-read of X).
#include "rd_conv_common.h"
#include <algorithm>

namespace rd {

template <typename T, int KS, int NT, int NB>
__global__ __launch_bounds__(64 * KS * NT) void pw_gemm_kernel(ConvArgs a, int mtiles, int ngroups) {
  constexpr int VE = Elem<T>::VE;
  constexpr int CK = 8 * VE;               // k elements per chunk: 128 bytes of a row
  constexpr int NW = KS * NT;
  constexpr int RS = 68;                   // floats per restaged pixel row (64 + 4: the 16-byte fragment writes spread evenly over the banks)
  constexpr int GT = 64 * KS;              // threads that share one channel tile in the epilogue
  constexpr int IT = 512 / GT;             // (pixel, 8-channel group) items per thread
  __shared__ float red[NW][64][RS];
  __shared__ float sred[NW][64][2];

  const int t = threadIdx.x, lane = t & 63, wv = RD_WAVE_UNIFORM(t >> 6);
  const int fr = lane & 15, fg = lane >> 4;
  // blocks of one pixel tile on one XCD (the dispatcher places block b on XCD b % 8): speed only
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const int ng = idx % ngroups, mt = (idx / ngroups) * 8 + xcd;
  if (mt >= mtiles) return;
  const int ks = wv % KS, nt = wv / KS;
  const int m0 = mt * 64, n0 = (ng * NT + nt) * 64;
  const int K = a.C1, M = a.M, Cout = a.Cout;
  const bool active = n0 < Cout;
  const int nch = (K + CK - 1) / CK;
  const int c0 = (int)((int64_t)ks * nch / KS), c1 = (int)((int64_t)(ks + 1) * nch / KS);

  f32x4 acc[4][4];      // [channel tile][pixel tile]
#pragma unroll
  for (int c = 0; c < 4; c++)
#pragma unroll
    for (int p = 0; p < 4; p++) acc[c][p] = f32x4{0, 0, 0, 0};

  if (active && c0 < c1) {
    // row bases: clamped into the tensors (rows past the end compute values that are never stored)
    // (32-bit element offsets from the two tensor bases: M <= 16384 rows here, and eight 64-bit row pointers cost the K-split-by-8 form its
    // 256-register budget)
    const T* const xbase = (const T*)a.src1; const T* const wbase = (const T*)a.w;
    int xr[4], wr[4];
#pragma unroll
    for (int p = 0; p < 4; p++) xr[p] = min(m0 + p * 16 + fr, M - 1) * K;
#pragma unroll
    for (int c = 0; c < 4; c++) wr[c] = min(n0 + c * 16 + fr, Cout - 1) * a.Kpad;
    auto load = [&](int kc, uint4 (&px)[4][2], uint4 (&pw)[4][2]) RD_INLINE_LAMBDA {
      // a request past the wave's range (issued unconditionally: the loop below has ONE exit, so the accumulators stay in place) re-reads the
      // last chunk and multiplies zeros
      const int k0 = min(kc, c1 - 1) * CK + fg * VE;
#pragma unroll
      for (int h = 0; h < 2; h++) {
        const int kk = k0 + h * 4 * VE;
        const bool ok = kk < K && kc < c1;      // the weights' K axis is padded with zeros up to the chunk; X is not: zero what lies past its row
        const int kx = ok ? kk : 0;
#pragma unroll
        for (int p = 0; p < 4; p++) {
          const uint4 v = *reinterpret_cast<const uint4*>(xbase + (xr[p] + kx));
          px[p][h].x = ok ? v.x : 0u; px[p][h].y = ok ? v.y : 0u; px[p][h].z = ok ? v.z : 0u; px[p][h].w = ok ? v.w : 0u;
        }
#pragma unroll
        for (int c = 0; c < 4; c++) { const uint4 v = *reinterpret_cast<const uint4*>(wbase + (wr[c] + kk)); pw[c][h] = v; }
      }
    };
    auto compute = [&](const uint4 (&px)[4][2], const uint4 (&pw)[4][2]) RD_INLINE_LAMBDA {
#pragma unroll
      for (int h = 0; h < 2; h++)
#pragma unroll
        for (int c = 0; c < 4; c++)
#pragma unroll
          for (int p = 0; p < 4; p++) {
            if (sizeof(T) == 4) {
              acc[c][p] = mfma_16x16x4_f32(__uint_as_float(pw[c][h].x), __uint_as_float(px[p][h].x), acc[c][p]);
              acc[c][p] = mfma_16x16x4_f32(__uint_as_float(pw[c][h].y), __uint_as_float(px[p][h].y), acc[c][p]);
              acc[c][p] = mfma_16x16x4_f32(__uint_as_float(pw[c][h].z), __uint_as_float(px[p][h].z), acc[c][p]);
              acc[c][p] = mfma_16x16x4_f32(__uint_as_float(pw[c][h].w), __uint_as_float(px[p][h].w), acc[c][p]);
            } else {
              s16x8 wa, pb;
              __builtin_memcpy(&wa, &pw[c][h], 16);
              __builtin_memcpy(&pb, &px[p][h], 16);
              acc[c][p] = mfma_16x16x32_bf16(wa, pb, acc[c][p]);
            }
          }
    };
    // NB register sets: chunks kc + 1 .. kc + NB - 1 are in flight while chunk kc is multiplied (a wave's range is a chain of memory round
    // trips, ~2 us each under load: with one block of four waves per CU nothing else hides them).  The fences keep each block of loads where
    // it is written: without them the scheduler sinks the weight loads of the NEXT chunk to just in front of their first use (seen in the ISA:
    // eight loads, then vmcnt(7) ... vmcnt(0) between the MFMAs).
    uint4 bx[NB][4][2], bw[NB][4][2];
#pragma unroll
    for (int u = 0; u < NB - 1; u++) load(c0 + u, bx[u], bw[u]);
    sched_fence();
    for (int kc = c0; kc < c1; kc += NB) {
#pragma unroll
      for (int u = 0; u < NB; u++) {
        load(kc + u + NB - 1, bx[(u + NB - 1) % NB], bw[(u + NB - 1) % NB]);
        sched_fence();
        compute(bx[u], bw[u]);
        sched_fence();
      }
    }
  }

  // ---- the wave's fp32 tile -> LDS, pixel-major: lane (fr, fg) holds channels c * 16 + 4 fg .. + 3 of pixel p * 16 + fr
#pragma unroll
  for (int p = 0; p < 4; p++)
#pragma unroll
    for (int c = 0; c < 4; c++)
      *reinterpret_cast<float4*>(&red[wv][p * 16 + fr][c * 16 + fg * 4]) = make_float4(acc[c][p][0], acc[c][p][1], acc[c][p][2], acc[c][p][3]);
  __syncthreads();

  // ---- epilogue: thread tl of the tile's GT threads takes pixels tl / 8 + (GT / 8) j and the 8 channels 8 (tl % 8) .. + 7
  const int tl = t % GT, g = tl & 7;
  const int co = n0 + g * 8;
  const bool cok = co < Cout;      // (Cout is a multiple of 8: a group is inside or outside as a whole)
  float bv[8];
#pragma unroll
  for (int e = 0; e < 8; e++) bv[e] = (a.bias && cok) ? a.bias[co + e] : 0.f;
  const bool plain = a.bias == nullptr && a.act == ACT_NONE;
  float ssum[8], ssq[8];
#pragma unroll
  for (int e = 0; e < 8; e++) { ssum[e] = 0.f; ssq[e] = 0.f; }
#pragma unroll
  for (int j = 0; j < IT; j++) {
    const int pl = (tl >> 3) + (GT / 8) * j;
    const int m = m0 + pl;
    float x[8];
    {
      const float4 u0 = *reinterpret_cast<const float4*>(&red[nt * KS][pl][g * 8]), u1 = *reinterpret_cast<const float4*>(&red[nt * KS][pl][g * 8 + 4]);
      x[0] = u0.x; x[1] = u0.y; x[2] = u0.z; x[3] = u0.w; x[4] = u1.x; x[5] = u1.y; x[6] = u1.z; x[7] = u1.w;
    }
#pragma unroll
    for (int s = 1; s < KS; s++) {      // the K-split partials, in wave order
      const float4 u0 = *reinterpret_cast<const float4*>(&red[nt * KS + s][pl][g * 8]), u1 = *reinterpret_cast<const float4*>(&red[nt * KS + s][pl][g * 8 + 4]);
      x[0] += u0.x; x[1] += u0.y; x[2] += u0.z; x[3] += u0.w; x[4] += u1.x; x[5] += u1.y; x[6] += u1.z; x[7] += u1.w;
    }
    if (!(cok && m < M)) continue;
    if (!plain) {
#pragma unroll
      for (int e = 0; e < 8; e++) x[e] = act_fwd(x[e] + bv[e], a.act, a.slope);
    }
    const int64_t o = (int64_t)m * Cout + co;
    if (a.add1) {      // the tensor's earlier gradient contribution: cur + this, rounded once
      float av[8];
      if (sizeof(T) == 4) {
        float a0[4], a1[4];
        ld4((const float*)a.add1 + o, a0); ld4((const float*)a.add1 + o + 4, a1);
#pragma unroll
        for (int e = 0; e < 4; e++) { av[e] = a0[e]; av[4 + e] = a1[e]; }
      } else {
        const uint4 r = *reinterpret_cast<const uint4*>((const bf16_t*)a.add1 + o);
        raw16_to_f32((const bf16_t*)nullptr, r, av);
      }
#pragma unroll
      for (int e = 0; e < 8; e++) x[e] += av[e];
    }
    float xr_[8];
    if (sizeof(T) == 4) {
      float* d = (float*)a.dst1 + o;
      *reinterpret_cast<float4*>(d) = make_float4(x[0], x[1], x[2], x[3]);
      *reinterpret_cast<float4*>(d + 4) = make_float4(x[4], x[5], x[6], x[7]);
#pragma unroll
      for (int e = 0; e < 8; e++) xr_[e] = x[e];
    } else {
      uint4 u;
      u.x = pack_bf16x2(x[0], x[1]); u.y = pack_bf16x2(x[2], x[3]); u.z = pack_bf16x2(x[4], x[5]); u.w = pack_bf16x2(x[6], x[7]);
      *reinterpret_cast<uint4*>((bf16_t*)a.dst1 + o) = u;
      raw16_to_f32((const bf16_t*)nullptr, u, xr_);
    }
#pragma unroll
    for (int e = 0; e < 8; e++) { ssum[e] += xr_[e]; ssq[e] += xr_[e] * xr_[e]; }
  }
  if (!a.stats) return;      // (uniform)
  // BatchNorm partials of the STORED values: over the wave's pixels (lanes 8 apart share a channel group), then over the tile's waves in order
#pragma unroll
  for (int e = 0; e < 8; e++) {
#pragma unroll
    for (int o = 8; o < 64; o <<= 1) { ssum[e] += __shfl_xor(ssum[e], o); ssq[e] += __shfl_xor(ssq[e], o); }
  }
  if (lane < 8) {
#pragma unroll
    for (int e = 0; e < 8; e++) { sred[wv][lane * 8 + e][0] = ssum[e]; sred[wv][lane * 8 + e][1] = ssq[e]; }
  }
  __syncthreads();
  // wave w of the block holds the pixels of ... tile w / KS (its threads' tl >> 6 = w % KS part): sum the KS waves of each tile in order
  for (int i = t; i < NT * 64; i += 64 * NW) {
    const int tn = i >> 6, ch = i & 63, cc = (ng * NT + tn) * 64 + ch;
    if (cc < Cout) {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int s = 0; s < KS; s++) { s1 += sred[tn * KS + s][ch][0]; s2 += sred[tn * KS + s][ch][1]; }
      a.stats[((int64_t)mt * Cout + cc) * 2 + 0] = s1;
      a.stats[((int64_t)mt * Cout + cc) * 2 + 1] = s2;
    }
  }
}

// ---- routing ---------------------------------------------------------------------------------------------------------------------------
// Where it is routed (tools/bench_pw.py on MI355X, profiles/r06_microbench/pw_gemm.txt): the fragment-ordered loads are half-line requests,
// 16 lines per 16 lanes, and cost the L1 four times the lookups of a coalesced row load -- the main loop runs at ~1.8 us per 16-KB chunk and
// wave.  That still beats the staged kernel's barrier-per-stage chain where the K axis is long and the tiles are few (M = 2592: 1392 -> 232
// 19.4 -> 11.7 us, 816 -> 232 13.7 -> 9.3 us, both with the K axis split over four waves) and loses everywhere else (136 -> 816 at
// 10 368 pixels: 22 -> 26 us; a third and fourth register set made it slower, not faster: the loop is load-issue bound, not latency
// bound).  Default: those shapes only; "pw_min_m" = 0 (tests) sends every eligible pointwise layer here, "pw_ks" forces a block shape.
static int pw_min_m() { return rd_opt(OPT_PW_MIN_M, -1); }      // test hook (rd_set_option "pw_min_m": 0 forces, 1 << 30 disables)
// the shape alone (ConvArgs::bn_y / add1 / stats are per call)
bool conv_pw_shape(const ConvArgs& a, int dtype) {
  const int ve = dtype == 0 ? 4 : 8;
  const bool form = a.KH == 1 && a.KW == 1 && a.stride == 1 && a.pad == 0 && a.dil == 1 && a.C2 == 0 && !a.ups && a.OH == a.Hin && a.OW == a.Win &&
                    a.K == a.C1 && (a.C1 % ve) == 0 && (a.Cout % 8) == 0 && a.D1 == a.Cout && !a.pool2 && !a.d2s && !a.s2d && a.M <= 16384;
  if (!form) return false;
  const int mm = pw_min_m();
  if (mm >= 0) return a.M >= mm;
  return dtype != 0 && a.M >= 2048 && a.M <= 4096 && a.C1 >= 768 && a.Cout <= 256;
}
bool conv_pw_ok(const ConvArgs& a, int dtype) { return conv_pw_shape(a, dtype) && !a.in_scale && !a.bn_y; }
static void pw_config(const ConvArgs& a, int dtype, int& ks, int& nt) {
  const int ck = dtype == 0 ? 32 : 64, nch = (int)cdiv(a.C1, ck);
  nt = 1;
  if (nch >= 5 || (nch >= 2 && cdiv(a.M, 64) * cdiv(a.Cout, 256) < 200)) ks = 4;
  else { ks = 1; nt = 4; }
  const int force = rd_opt(OPT_PW_KS, 0);
  if (force == 8 || force == 4) { ks = force; nt = 1; }
  else if (force == 1) { ks = 1; nt = 4; }
}
int conv_pw_rows(const ConvArgs& a) { return (int)cdiv(a.M, 64); }
const char* conv_pw_name(const ConvArgs& a, int dtype) {
  static thread_local char buf[96];
  int ks, nt;
  pw_config(a, dtype, ks, nt);
  snprintf(buf, sizeof(buf), "pw_gemm_kernel<%s, %d, %d>", dtype == 0 ? "float" : RD_T16_NAME, ks, nt);
  return buf;
}
void launch_conv_pw(const ConvArgs& a, int dtype, hipStream_t st) {
  int ks, nt;
  pw_config(a, dtype, ks, nt);
  const int mtiles = (int)cdiv(a.M, 64), ngroups = (int)cdiv(a.Cout, 64 * nt);
  const dim3 grid((unsigned)(8 * cdiv(mtiles, 8) * ngroups));
#define RD_PW(TT, KSV, NTV) hipLaunchKernelGGL((pw_gemm_kernel<TT, KSV, NTV, 2>), grid, dim3(64 * KSV * NTV), 0, st, a, mtiles, ngroups)
#define RD_PW_T(TT) { if (ks == 8) RD_PW(TT, 8, 1); else if (ks == 4) RD_PW(TT, 4, 1); else RD_PW(TT, 1, 4); }
  if (dtype == 0) RD_PW_T(float) else RD_PW_T(bf16_t)
#undef RD_PW_T
#undef RD_PW
}

}  // namespace rd
