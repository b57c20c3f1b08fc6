// bf16 weight gradient of 3x3 / stride-1 layers on the bf16 MFMA, fed by the gfx950 LDS transpose read: narrow layers (Cin in {16, 32,
// 64}, Cout <= 32) as one slice, wide layers (Cin % 64 == 0) as (64 input channels) x (32 output channels) slices on blockIdx.y.
//
//   dW[co][tap][ci] = sum over pixels p of dY[p][co] * X[p + tap][ci]          (reference: autograd of utils/net_utils.py:84-91,195-198)
//
// The contraction axis is PIXELS, but activations are NHWC: a lane of v_mfma_f32_16x16x32_bf16 must hold 8 consecutive pixels of ONE
// channel.  The earlier kernels either converted to fp32 and used the 16x16x4 fp32 MFMA (157 TF peak: the halo kernel, 0.56-0.93 ms on
// RC-Net's ROI-resolution layers) or re-packed pixel pairs into 32-bit words with shifts/masks on the way into LDS.  Here the input
// patch (rows+2) x (cols+2) x Cin and the dY tile are copied into LDS as they are (16-byte vectors, upsample / concat folded into the
// gather) and ds_read_b64_tr_b16 delivers the transposed fragments: within a 16-lane group, lane m supplies the address of pixel
// (m >> 2), channels 4*(m & 3)..+3, and lane i receives channel i of pixels 0..3 (mapping measured on the GPU, see rd_common.h).
// Blocks are persistent over 8 x TW pixel tiles (next tile prefetched into registers, double-buffered LDS, one barrier per tile), keep
// the fp32 accumulators of all nine taps in registers, and write ONE slab each; wgrad_reduce_kernel sums the slabs in a fixed order.
#include "rd_conv_common.h"
#include <type_traits>
#include <stdio.h>

namespace rd {

#ifndef RD_WGRAD_DEEP_MAX
#define RD_WGRAD_DEEP_MAX 12
#endif
#ifndef RD_WGRAD_SHARE      // A/B hook: 0 = one pair of transpose reads per column tile and k-step, 1 = shared rows (TW = 16), 2 = + next k-step's reads
                            // issued before this one's MFMAs, 3 = also for TW = 8
#define RD_WGRAD_SHARE 2
#endif

// LDS layout.  Round 1 (profiles/r01_wgrad_tr_lds_swizzle.txt): with plain [pixel][channel] images the 32-lane transpose reads of the
// 64-channel slices hit 2..4 bank groups (SQ_LDS_BANK_CONFLICT = 66 % of SQ_LDS_IDX_ACTIVE); an XOR swizzle removed every conflict and made
// the kernel SLOWER, because per-read address arithmetic replaced immediate offsets.  Round 3: both at once -- the images are stored as
// PLANES of 16 channels, [plane][pixel][32 bytes] (the layout of rd_conv3x3_frag.hip), and the k index of a step is mapped to pixels so
// that the 32 lanes of one transpose read (lane groups 2g, 2g+1) cover 8 CONSECUTIVE pixels of one plane = 256 contiguous bytes = every
// bank once; every read address stays `base + immediate`.  Plane strides are 32 or 64 bytes mod 256 so that the 8-lane groups of the
// 16-byte staging stores are conflict free too.
// NWV waves per block: 4, or 8 for the wide slices (each wave then owns 4-5 instead of 9 column tiles: 40 instead of 72 accumulator
// registers and half the staging registers -> four waves per SIMD instead of two for a kernel that is issue bound per wave)
// AFF: source 1 is a BatchNorm-ed producer's RAW output; scale / shift + activation are applied when the patch is written to LDS
// (WgradArgs::in_scale) -- the x operand of the gradient is the activated tensor the forward convolution saw, which is never stored.
template <int CTI, int RT, int TW, int NWV = 4, bool AFF = false>
__global__ __launch_bounds__(64 * NWV, NWV == 8 ? (RT >= 4 ? 2 : 4) : ((CTI * RT <= 2) ? 4 : 2)) void conv3x3_wgrad_tr_kernel(WgradArgs a, int tilesH, int tilesW, int nci) {
  constexpr int NT = 64 * NWV;
  typedef bf16_t T;
  constexpr int CIN = CTI * 16, COP = RT * 16;
  constexpr int TH = 8, WT = TW + 2, HT = TH + 2, NPX = HT * WT, NPY = TH * TW;
  constexpr int XS = CIN / 8, YS = COP / 8;           // 16-byte slots per pixel
  constexpr int NXS = NPX * XS, NYS = NPY * YS;
  constexpr int XIT = (NXS + NT - 1) / NT, YIT = (NYS + NT - 1) / NT;
  constexpr int KSTEPS = NPY / 32;                     // 32 pixels per MFMA k-step
  constexpr int NCT = 9 * CTI, NCW = (NCT + NWV - 1) / NWV;    // (tap, cin-tile) column tiles, per wave
  // plane strides in bytes: >= 32 bytes per pixel, = 32 (eight slots per pixel) or 64 (four) mod 256 when there is more than one plane
  constexpr int XPSB = XS <= 2 ? NPX * 32 : ((NPX * 32 + 255) / 256) * 256 + (XS == 8 ? 32 : 64);
  constexpr int YPSB = YS <= 2 ? NPY * 32 : ((NPY * 32 + 255) / 256) * 256 + (YS == 8 ? 32 : 64);
  constexpr int XPS = XPSB / 2, YPS = YPSB / 2;        // ... in 16-bit elements
  constexpr int ASTEP = 32 * 16;                       // elements between k-steps in a dY plane (32 consecutive pixels)
  constexpr int BSTEP = (32 / TW) * WT * 16;           // ... in a patch plane (four rows of 8, two rows of 16 or one row of 32)
  // second transpose read of a fragment: the next 16 pixels of the k-step (one row of 16 / two rows of 8 / the second half of a row of 32)
  constexpr int A2 = (TW == 16 ? TW : (TW == 8 ? 2 * TW : 16)) * 16, B2 = (TW == 16 ? WT : (TW == 8 ? 2 * WT : 16)) * 16;
  __shared__ __attribute__((aligned(16))) unsigned char sX[2][(XS / 2 > 0 ? XS / 2 : 1) * XPSB];
  __shared__ __attribute__((aligned(16))) unsigned char sY[2][(YS / 2 > 0 ? YS / 2 : 1) * YPSB];

  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  // wide layers: blockIdx.y picks a (CIN input channels) x (COP output channels) slice of the gradient; every slice walks all tiles
  const int ci0 = ((int)blockIdx.y % nci) * CIN, co0 = ((int)blockIdx.y / nci) * COP;
  const int CinT = a.C1 + a.C2;
  __shared__ __attribute__((aligned(16))) float sAff[AFF ? 2 * CIN : 1];      // (scale, shift) of this slice's input channels
  if (AFF) affine_fill(sAff, a.in_scale, a.in_shift, ci0, CIN, a.C1, t, NT);
  const bool aff_lane = AFF && ci0 + (t % XS) * 8 < a.C1;      // NT % XS == 0: all x slots of a thread carry the same eight channels

  // persistent tile walk: XCD x owns a contiguous range of tiles (halo pixels shared by neighbouring tiles stay in its L2)
  const int ntiles = a.N * tilesH * tilesW;
  const int T8 = (ntiles + 7) >> 3, G8 = gridDim.x >> 3;
  const int xcd = blockIdx.x & 7;
  const int tend = min(ntiles, (xcd + 1) * T8);
  int tile = xcd * T8 + (blockIdx.x >> 3);

  // transpose-read roles.  A k-step is 32 pixels; one read covers 16 of them: lane group fg supplies pixels 4 fg .. 4 fg + 3 of those 16
  // (this lane: 4 fg + (fr >> 2), channels 4 (fr & 3) .. +3 of whichever 16-channel plane is read), the second read the other 16.
  const int j16 = fg * 4 + (fr >> 2);
  const int yl = TW == 8 ? (j16 >> 3) : 0, xl = TW == 8 ? (j16 & 7) : j16;
  const int aoff = (yl * TW + xl) * 16 + (fr & 3) * 4;
  const int boff = (yl * WT + xl) * 16 + (fr & 3) * 4;
  int coloff[NCW]; bool jv[NCW];
#pragma unroll
  for (int j = 0; j < NCW; j++) {
    int idx = wv + NWV * j;
    jv[j] = idx < NCT;
    if (!jv[j]) idx = 0;
    const int tap = idx / CTI, ct = idx - tap * CTI;
    coloff[j] = ((tap / 3) * WT + (tap % 3)) * 16 + ct * XPS;
  }

  f32x4 acc[RT][NCW];
#pragma unroll
  for (int i = 0; i < RT; i++)
#pragma unroll
    for (int j = 0; j < NCW; j++) acc[i][j] = f32x4{0, 0, 0, 0};

  const bool yvec = (a.Cout & 7) == 0;
  // tile coordinates advance by G8 tiles with carries (no per-tile runtime divisions, see rd_conv3x3.hip)
  struct TC { int n, th, tw; };
  const TC tstep = {(G8 / tilesW) / tilesH, (G8 / tilesW) % tilesH, G8 % tilesW};
  auto decode = [&](int tl) RD_INLINE_LAMBDA { TC c; c.tw = tl % tilesW; const int q_ = tl / tilesW; c.th = q_ % tilesH; c.n = q_ / tilesH; return c; };
  auto advance = [&](TC c) RD_INLINE_LAMBDA {
    c.tw += tstep.tw; if (c.tw >= tilesW) { c.tw -= tilesW; c.th++; }
    c.th += tstep.th; if (c.th >= tilesH) { c.th -= tilesH; c.n++; }
    c.n += tstep.n;
    return c;
  };
  // This thread's x / dy slots are the same for every tile: patch pixel, channel offset and source tensor are decoded once and packed
  // into two registers per slot; per tile a slot costs a bounds test and one 64-bit multiply-add (the general gather re-derived all
  // of it per load, ~45 VALU instructions, in a kernel that is issue bound).  Pixel indices fit 32 bits (wgrad3x3_tr_ok).
  int xpk[XIT], xco[XIT];   // (py-1 + 1) << 8 | (px-1 + 1) | source-2 flag << 16 | valid << 17;  channel offset inside the source
#pragma unroll
  for (int i = 0; i < XIT; i++) {
    const int idx = t + NT * i;
    const int pp = idx / XS, sl = idx - pp * XS;
    const int py = pp / WT, px = pp - py * WT;
    const int ci = ci0 + sl * 8;
    const bool s2 = ci >= a.C1;
    xpk[i] = (py << 8) | px | (s2 ? 1 << 16 : 0) | (idx < NXS ? 1 << 17 : 0);
    xco[i] = s2 ? ci - a.C1 : ci;
  }
  int ypk[YIT];             // py << 8 | px | valid << 16 (valid = slot exists and its 8 channels start inside Cout)
#pragma unroll
  for (int i = 0; i < YIT; i++) {
    const int idx = t + NT * i;
    const int pp = idx / YS, sl = idx - pp * YS;
    const int py = pp / TW, px = pp - py * TW;
    ypk[i] = (py << 8) | px | ((idx < NYS && co0 + sl * 8 < a.Cout) ? 1 << 16 : 0) | (sl << 20);
  }
  const int Hp = a.ups ? a.H1 : a.Hin, Wp = a.ups ? a.W1 : a.Win;
  auto fetch = [&](TC tc, uint4 (&rx)[XIT], uint4 (&ry)[YIT], unsigned& okm) RD_INLINE_LAMBDA {
    const int n = tc.n;
    const int oh0 = tc.th * TH, ow0 = tc.tw * TW;
    okm = 0;
#pragma unroll
    for (int i = 0; i < XIT; i++) {
      uint4 v = make_uint4(0, 0, 0, 0);
      const int ih = oh0 - 1 + ((xpk[i] >> 8) & 0xff), iw = ow0 - 1 + (xpk[i] & 0xff);
      if ((xpk[i] >> 17) && (unsigned)ih < (unsigned)a.Hin && (unsigned)iw < (unsigned)a.Win) {
        if (AFF) okm |= 1u << i;
        int hs = ih, ws = iw;
        if (a.ups) {  // F.interpolate(mode='nearest') source index, ATen float formula (as conv_src_ptr)
          hs = min((int)floorf((float)ih * a.scale_h), a.H1 - 1);
          ws = min((int)floorf((float)iw * a.scale_w), a.W1 - 1);
        }
        const int pix = (n * Hp + hs) * Wp + ws;
        const bool s2 = (xpk[i] >> 16) & 1;
        const T* sb = s2 ? (const T*)a.src2 : (const T*)a.src1;
        v = *reinterpret_cast<const uint4*>(sb + (int64_t)pix * (s2 ? a.C2 : a.C1) + xco[i]);
      }
      rx[i] = v;
    }
#pragma unroll
    for (int i = 0; i < YIT; i++) {
      uint4 v = make_uint4(0, 0, 0, 0);
      const int oh = oh0 + ((ypk[i] >> 8) & 0xff), ow = ow0 + (ypk[i] & 0xff);
      if (((ypk[i] >> 16) & 1) && oh < a.OH && ow < a.OW) {
        const int sl = ypk[i] >> 20;
        const T* p = (const T*)a.dy + (int64_t)((n * a.OH + oh) * a.OW + ow) * a.Cout + co0 + sl * 8;
        if (yvec) v = *reinterpret_cast<const uint4*>(p);
        else {  // Cout not a multiple of 8 (the 1-channel head): element-wise, zero padded
          unsigned short e[8];
#pragma unroll
          for (int q = 0; q < 8; q++) e[q] = (co0 + sl * 8 + q < a.Cout) ? p[q].v : (unsigned short)0;
          v.x = e[0] | ((unsigned)e[1] << 16); v.y = e[2] | ((unsigned)e[3] << 16);
          v.z = e[4] | ((unsigned)e[5] << 16); v.w = e[6] | ((unsigned)e[7] << 16);
        }
      }
      ry[i] = v;
    }
  };
  auto stash = [&](int buf, const uint4 (&rx)[XIT], const uint4 (&ry)[YIT], unsigned okm) RD_INLINE_LAMBDA {
    float sc[8], sh[8];
    if (AFF) {
#pragma unroll
      for (int e = 0; e < 8; e++) { sc[e] = sAff[(t % XS) * 8 + e]; sh[e] = sAff[CIN + (t % XS) * 8 + e]; }
    }
#pragma unroll
    for (int i = 0; i < XIT; i++) {
      const int idx = t + NT * i, pp = idx / XS, sl = idx - pp * XS;
      if (idx < NXS) {
        uint4 v = rx[i];
        if (AFF) { const uint4 z = affine16((const T*)nullptr, v, sc, sh, a.in_act, a.in_slope); if (aff_lane && ((okm >> i) & 1u)) v = z; }
        *reinterpret_cast<uint4*>(&sX[buf][(sl >> 1) * XPSB + pp * 32 + (sl & 1) * 16]) = v;
      }
    }
#pragma unroll
    for (int i = 0; i < YIT; i++) {
      const int idx = t + NT * i, pp = idx / YS, sl = idx - pp * YS;
      if (idx < NYS) *reinterpret_cast<uint4*>(&sY[buf][(sl >> 1) * YPSB + pp * 32 + (sl & 1) * 16]) = ry[i];
    }
  };
  auto compute = [&](int buf, auto njc) RD_INLINE_LAMBDA {
    constexpr int NJ = decltype(njc)::value;
    const unsigned short* bx = reinterpret_cast<const unsigned short*>(&sX[buf][0]) + boff;
    const unsigned short* by = reinterpret_cast<const unsigned short*>(&sY[buf][0]) + aoff;
    // 64-channel slices in four waves: wave wv owns input-channel plane wv and ALL nine taps (column tile j is tap j).  The halves of its
    // fragments are then shared between taps and between k-steps -- a transpose read covers one patch row (TW = 16) or one pair of rows
    // (TW = 8) at column offset kw, and tap kh of k-step s wants rows 2s+kh, 2s+kh+1 (pairs 4s+kh, 4s+kh+2): every row / pair is read
    // ONCE per tile and kw and kept in registers across the taps and the next k-step.  Transpose reads per wave and tile: 88 -> 46
    // (TW = 16), 44 -> 35 (TW = 8).  Measured (tools/ab_wgrad.sh, tools/pmc_wgrad.sh; 360000 pixels 128 -> 64): SQ_INSTS_LDS -44 %,
    // SQ_WAIT_INST_LDS -45 %, wave cycles -3.7 %, 0.090 -> 0.086 ms -- the reads were NOT what bound the kernel; TW = 8 came out 1-2 %
    // slower (the operand copies cost more than its nine saved reads) and keeps one read pair per column tile.  A start offset between
    // the two blocks of a CU (0.5 - 8 K cycles) changed nothing either.
    if constexpr (CTI == 4 && NWV == 4 && NJ == 9 && (TW == 16 || (TW == 8 && RD_WGRAD_SHARE >= 3)) && RD_WGRAD_SHARE) {
      constexpr int NP = TW == 16 ? 4 : 5, CARRY = TW == 16 ? 2 : 1, RSTEP = TW == 16 ? 2 : 4, HI = TW == 16 ? 1 : 2, NN = NP - CARRY;
      const unsigned short* bw = bx + wv * XPS;
      uint2 R[3][NP], Rn[3][NN];
      s16x8 ya[RT], yn[RT];
      auto read_a = [&](int s, s16x8 (&dst)[RT]) RD_INLINE_LAMBDA {
#pragma unroll
        for (int i = 0; i < RT; i++) {
          uint2 lo = lds_read_tr16_b64(by + s * ASTEP + i * YPS), hi = lds_read_tr16_b64(by + s * ASTEP + i * YPS + A2);
          uint4 v = make_uint4(lo.x, lo.y, hi.x, hi.y);
          __builtin_memcpy(&dst[i], &v, 16);
        }
      };
#pragma unroll
      for (int kw = 0; kw < 3; kw++)
#pragma unroll
        for (int q = 0; q < NP; q++) R[kw][q] = lds_read_tr16_b64(bw + (q * WT + kw) * 16);
      read_a(0, ya);
      sched_fence();      // the first k-step's reads stay in front of the second's (the scheduler mixed them: lgkmcnt(2) before the first MFMA)
#pragma unroll
      for (int s = 0; s < KSTEPS; s++) {
        // the NEXT k-step's reads are issued before this one's MFMAs (LDS returns in order: the wait in front of the first MFMA leaves
        // them outstanding); with reads, wait, MFMAs per k-step every wave exposed one LDS round trip per k-step -- waves were parked
        // at s_waitcnt / the barrier for 45 % of their cycles (profiles/r03_pmc_patch/wgrad_256to128_planes.txt)
        if (RD_WGRAD_SHARE >= 2 && s + 1 < KSTEPS) {
          read_a(s + 1, yn);
#pragma unroll
          for (int kw = 0; kw < 3; kw++)
#pragma unroll
            for (int q = 0; q < NN; q++) Rn[kw][q] = lds_read_tr16_b64(bw + (((s + 1) * RSTEP + CARRY + q) * WT + kw) * 16);
        }
        sched_fence();
#pragma unroll
        for (int kh = 0; kh < 3; kh++)
#pragma unroll
          for (int kw = 0; kw < 3; kw++) {
            const uint4 v = make_uint4(R[kw][kh].x, R[kw][kh].y, R[kw][kh + HI].x, R[kw][kh + HI].y);
            s16x8 xb;
            __builtin_memcpy(&xb, &v, 16);
#pragma unroll
            for (int i = 0; i < RT; i++) acc[i][kh * 3 + kw] = mfma_16x16x32_bf16(ya[i], xb, acc[i][kh * 3 + kw]);
          }
        sched_fence();
        if (s + 1 < KSTEPS) {
          if (RD_WGRAD_SHARE < 2) {
            read_a(s + 1, yn);
#pragma unroll
            for (int kw = 0; kw < 3; kw++)
#pragma unroll
              for (int q = 0; q < NN; q++) Rn[kw][q] = lds_read_tr16_b64(bw + (((s + 1) * RSTEP + CARRY + q) * WT + kw) * 16);
          }
#pragma unroll
          for (int kw = 0; kw < 3; kw++) {
#pragma unroll
            for (int q = 0; q < CARRY; q++) R[kw][q] = R[kw][NN + q];
#pragma unroll
            for (int q = 0; q < NN; q++) R[kw][CARRY + q] = Rn[kw][q];
          }
#pragma unroll
          for (int i = 0; i < RT; i++) ya[i] = yn[i];
        }
      }
      return;
    }
#pragma unroll
    for (int s = 0; s < KSTEPS; s++) {
      s16x8 ya[RT];
#pragma unroll
      for (int i = 0; i < RT; i++) {
        uint2 lo = lds_read_tr16_b64(by + s * ASTEP + i * YPS), hi = lds_read_tr16_b64(by + s * ASTEP + i * YPS + A2);
        uint4 v = make_uint4(lo.x, lo.y, hi.x, hi.y);
        __builtin_memcpy(&ya[i], &v, 16);
      }
      // ALL fragment reads of the k-step first, then its MFMAs: left to itself the compiler issues two transpose reads, waits for them
      // (lgkmcnt(0)) and runs two MFMAs, nine times per k-step -- one exposed LDS round trip per 32 MFMA cycles (ISA, round 3)
      s16x8 xb[NJ];
#pragma unroll
      for (int j = 0; j < NJ; j++) {
        uint2 lo = lds_read_tr16_b64(bx + s * BSTEP + coloff[j]), hi = lds_read_tr16_b64(bx + s * BSTEP + coloff[j] + B2);
        uint4 v = make_uint4(lo.x, lo.y, hi.x, hi.y);
        __builtin_memcpy(&xb[j], &v, 16);
      }
      sched_fence();
#pragma unroll
      for (int j = 0; j < NJ; j++)
#pragma unroll
        for (int i = 0; i < RT; i++) acc[i][j] = mfma_16x16x32_bf16(ya[i], xb[j], acc[i][j]);
      sched_fence();
    }
  };

  // two fully unrolled bodies (a wave owns NCW or NCW-1 column tiles): a per-MFMA branch makes the compiler shuttle accumulators
  auto tile_body = [&](int buf) RD_INLINE_LAMBDA {
    if (jv[NCW - 1]) compute(buf, std::integral_constant<int, NCW>{});
    else compute(buf, std::integral_constant<int, NCW - 1>{});
  };
  // prefetch distance two tiles where the register budget allows (every variant at <= 12 staged vectors per thread since round 3:
  // 0.831 -> 0.811 ms over the eleven wgrad shapes of tools/bench_conv.py), one otherwise
#ifndef RD_WGRAD_DEEP16
#define RD_WGRAD_DEEP16 0
#endif
  // (the pipelined fragment reads of the 64-channel / 16-wide variant need the second staging set's 32 registers: 256 + 31 spilled with both)
  constexpr bool DEEP = (XIT + YIT) <= RD_WGRAD_DEEP_MAX && (RD_WGRAD_DEEP16 || !(CTI == 4 && NWV == 4 && TW == 16 && RD_WGRAD_SHARE >= 2));
  uint4 xa[XIT], ya_[YIT], xb_[DEEP ? XIT : 1], yb_[DEEP ? YIT : 1];
  int buf = 0;
  unsigned ma = 0, mb = 0;      // validity bits of the staged x slots (AFF: padding stays zero behind the affine map)
  if (DEEP) {
    int t0 = tile, t1 = tile + G8;
    TC c1 = advance(decode(t0));
    if (t0 < tend) fetch(decode(t0), xa, ya_, ma);
    if (t1 < tend) fetch(c1, reinterpret_cast<uint4 (&)[XIT]>(xb_), reinterpret_cast<uint4 (&)[YIT]>(yb_), mb);
    if (AFF) __syncthreads();      // the coefficient copy is complete
    while (t0 < tend) {
      stash(buf, xa, ya_, ma);
      __syncthreads();
      { const int t2 = t1 + G8; c1 = advance(c1); if (t2 < tend) fetch(c1, xa, ya_, ma); tile_body(buf); t0 = t1; t1 = t2; buf ^= 1; }
      if (t0 >= tend) break;
      stash(buf, reinterpret_cast<const uint4 (&)[XIT]>(xb_), reinterpret_cast<const uint4 (&)[YIT]>(yb_), mb);
      __syncthreads();
      { const int t2 = t1 + G8; c1 = advance(c1); if (t2 < tend) fetch(c1, reinterpret_cast<uint4 (&)[XIT]>(xb_), reinterpret_cast<uint4 (&)[YIT]>(yb_), mb); tile_body(buf); t0 = t1; t1 = t2; buf ^= 1; }
    }
  } else {
    TC c0 = decode(tile);
    if (tile < tend) fetch(c0, xa, ya_, ma);
    if (AFF) __syncthreads();      // the coefficient copy is complete
    while (tile < tend) {
      stash(buf, xa, ya_, ma);
      __syncthreads();
      const int next = tile + G8;
      c0 = advance(c0);
      if (next < tend) fetch(c0, xa, ya_, ma);
      tile_body(buf);
      tile = next;
      buf ^= 1;
    }
  }

  float* slab = a.slab + (int64_t)blockIdx.x * a.Cout * a.K;
#pragma unroll
  for (int i = 0; i < RT; i++)
#pragma unroll
    for (int j = 0; j < NCW; j++)
      if (jv[j]) {
        const int idx = wv + NWV * j, tap = idx / CTI, ct = idx - tap * CTI;
        const int k = tap * CinT + ci0 + ct * 16 + fr;
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int co = co0 + i * 16 + fg * 4 + r;
          if (co < a.Cout) slab[(int64_t)co * a.K + k] = acc[i][j][r];
        }
      }
}


// ---- round 5: map-fitted tiles, 64 x 64 output slices, v_mfma_f32_32x32x16 (weight-gradient half) -------------------------
// The kernel above tiles a map with 8 x 8 / 8 x 16 pixel tiles whatever its width: on RC-Net's RoI maps (15x6, 30x12, 60x25) 27-30 % of the
// issued MFMAs multiply padding, and a 64-channel input slice is staged once per 32 output channels (PMC round 4: 36 % padded MFMAs, 1.71 x
// the algorithmic HBM traffic).  Here
//   * a tile is TH FULL-WIDTH rows of one map (TH chosen by the host so that TH x OW <= 128 pixels wastes the least of its last 16-pixel
//     k-step: 15x6 -> the whole map, 90 of 96; 30x12 -> 10 rows, 120 of 128; 60x25 -> 4 rows, 100 of 112): the transpose reads take PER-LANE
//     pixel offsets (k index -> (row, column) of the tile, worked out per k-step with a float reciprocal) instead of the immediates that
//     need power-of-two tile widths; no column padding, the halo columns are the image border;
//   * the MFMA is 32x32x16: A = dY^T (32 output channels x 16 pixels), B = X (16 pixels x 32 input channels), one per tap; a wave owns 32 x 32
//     x nine taps (144 accumulator registers), the four waves of a block a 64 x 64 slice: every staged input slice feeds 64 output channels;
//   * planes of 16 channels x [pixel][32 bytes] as above with plane strides = 128 (mod 256) bytes: the two 16-lane groups a transpose read
//     serves together (channels 0-15 / 16-31 of the same four pixels) fall into different bank halves; staging stores go four pixels x two
//     16-byte halves of ONE plane per 8-lane group (128 contiguous bytes).
struct FitGeom { int TH, tilesH, npx, tpx, ks, nci; float rOW, rWT; };
__device__ __attribute__((aligned(16))) unsigned g_fit_zero[8];      // the source of every staged element that does not exist (halo, padded k, ragged channels)

// Staging is LDS-DMA (global_load_lds_dwordx4, rd_common.h dma16_to_lds): a wave instruction lands 32 pixels x 32 bytes of ONE 16-channel plane
// as a contiguous kilobyte; wave w stages plane w of both operands.  No staging registers and no ds_write pass: the 144 accumulator registers
// leave room for the read pipeline below (the first version of this kernel staged through registers, sat at the 256-register cap and ran
// every k-step's twenty transpose reads in front of its nine MFMAs: 585 against 584-639 TFLOP/s for the 8 x TW kernel).
template <int NWV>
__global__ __launch_bounds__(64 * NWV, 2) void conv3x3_wgrad_fit_kernel(WgradArgs a, FitGeom g) {
  static_assert(NWV == 4, "four waves: (output half) x (input half); wave w stages channel plane w");
  typedef bf16_t T;
  constexpr int NPXMAX = 160, TPXMAX = 128;
  constexpr int XPSB = NPXMAX * 32 + 128, YPSB = TPXMAX * 32 + 128;      // plane strides in bytes (= 128 mod 256)
  constexpr int XG = NPXMAX / 32, YG = TPXMAX / 32;                       // 32-pixel groups per plane = DMA instructions per wave and tile
  __shared__ __attribute__((aligned(16))) unsigned char sX[2][4 * XPSB];
  __shared__ __attribute__((aligned(16))) unsigned char sY[2][4 * YPSB];

  const int t = threadIdx.x, lane = t & 63, wv = RD_WAVE_UNIFORM(t >> 6);
  const int hco = wv & 1, hci = wv >> 1;
  const int ci0 = ((int)blockIdx.y % g.nci) * 64, co0 = ((int)blockIdx.y / g.nci) * 64;
  const int CinT = a.C1 + a.C2;
  const int OW = a.OW, WT = OW + 2, TH = g.TH;
  const int Hp = a.ups ? a.H1 : a.Hin, Wp = a.ups ? a.W1 : a.Win;

  const int ntiles = a.N * g.tilesH;
  const int T8 = (ntiles + 7) >> 3, G8 = gridDim.x >> 3;
  const int xcd = blockIdx.x & 7;
  const int tend = min(ntiles, (xcd + 1) * T8);
  int tile = xcd * T8 + (blockIdx.x >> 3);

  // ---- staging role of this lane: pixel (lane >> 1) of every 32-pixel group, 16-byte half lane & 1 of plane wv ------------------------------
  const int sch = wv * 16 + (lane & 1) * 8;               // channel offset inside the 64-channel slices
  const bool x2nd = ci0 + sch >= a.C1;                    // this lane's input channels come from the second source (concat)
  const T* const xsrc = x2nd ? (const T*)a.src2 + (ci0 + sch - a.C1) : (const T*)a.src1 + (ci0 + sch);
  const int xcs = x2nd ? a.C2 : a.C1;
  const bool yok = co0 + sch < a.Cout;
  const T* const ysrc = (const T*)a.dy + co0 + (yok ? sch : 0);
  const T* const zero = reinterpret_cast<const T*>(g_fit_zero);
  int xm[XG], ym[YG];      // patch (row + 1) << 8 | (column + 1), tile row << 8 | column; -1: the pixel does not exist in this geometry
#pragma unroll
  for (int i = 0; i < XG; i++) {
    const int pp = i * 32 + (lane >> 1);
    int px;
    const int py = fdiv_small(min(pp, g.npx - 1), WT, g.rWT, px);
    xm[i] = (pp < g.npx && px >= 1 && px <= OW) ? (py << 8) | px : -1;      // the halo columns are the image border: always zero
  }
#pragma unroll
  for (int i = 0; i < YG; i++) {
    const int kk = i * 32 + (lane >> 1);
    int kx;
    const int ky = fdiv_small(min(kk, g.tpx - 1), OW, g.rOW, kx);
    ym[i] = (kk < g.tpx && yok) ? (ky << 8) | kx : -1;
  }
  auto issue = [&](int tl, int buf) RD_INLINE_LAMBDA {
    const int n = tl / g.tilesH, oh0 = (tl - n * g.tilesH) * TH;
#pragma unroll
    for (int i = 0; i < XG; i++) {
      const int ih = oh0 - 1 + (xm[i] >> 8), iw = (xm[i] & 0xff) - 1;
      const bool ok = xm[i] >= 0 && (unsigned)ih < (unsigned)a.Hin;
      int hs = min(max(ih, 0), a.Hin - 1), ws = min(max(iw, 0), a.Win - 1);
      if (a.ups) {
        hs = min((int)floorf((float)hs * a.scale_h), a.H1 - 1);
        ws = min((int)floorf((float)ws * a.scale_w), a.W1 - 1);
      }
      const T* p = xsrc + (int64_t)((n * Hp + hs) * Wp + ws) * xcs;
      dma16_to_lds(ok ? p : zero, &sX[buf][wv * XPSB + i * 1024]);
    }
#pragma unroll
    for (int i = 0; i < YG; i++) {
      const int oh = oh0 + (ym[i] >> 8), ow = ym[i] & 0xff;
      const bool ok = ym[i] >= 0 && oh < a.OH;
      const T* p = ysrc + (int64_t)((n * a.OH + min(oh, a.OH - 1)) * OW + ow) * a.Cout;
      dma16_to_lds(ok ? p : zero, &sY[buf][wv * YPSB + i * 1024]);
    }
  };

  // ---- transpose-read roles: 16-lane group G = lane >> 4 reads plane (G & 1) of its half, k-group G >> 1; lane m of the group supplies
  // pixel (m >> 2) of the read's four pixels, channels 4 (m & 3) .. + 3 -------------------------------------------------------------------
  const int G = lane >> 4, m = lane & 15;
  const int kq = 8 * (G >> 1) + (m >> 2);                                         // this lane's pixel inside a k-step, first read (second: + 4)
  const int abase = (2 * hco + (G & 1)) * YPSB + kq * 32 + (m & 3) * 8;
  const int bbase = (2 * hci + (G & 1)) * XPSB + (m & 3) * 8;

  f32x16 acc[9];
#pragma unroll
  for (int j = 0; j < 9; j++)
#pragma unroll
    for (int v = 0; v < 16; v++) acc[j][v] = 0.f;

  auto compute = [&](int buf) RD_INLINE_LAMBDA {
    const unsigned char* const by = &sY[buf][0] + abase;
    const unsigned char* const bx = &sX[buf][0] + bbase;
    // patch offsets of this lane's two pixel quads of k-step s: k -> (row, column) of the tile; padded k (>= TH OW) reads pixel 0 (its dY is zero)
    auto offs = [&](int s_, int (&o)[2]) RD_INLINE_LAMBDA {
#pragma unroll
      for (int r = 0; r < 2; r++) {
        int kk = s_ * 16 + kq + 4 * r, kx;
        kk = kk < g.tpx ? kk : 0;
        const int ky = fdiv_small(kk, OW, g.rOW, kx);
        o[r] = (ky * WT + kx) * 32;
      }
    };
    auto read_a = [&](int s_) RD_INLINE_LAMBDA {
      const uint2 lo = lds_read_tr16_b64(by + s_ * 512), hi = lds_read_tr16_b64(by + s_ * 512 + 128);
      const uint4 v = make_uint4(lo.x, lo.y, hi.x, hi.y);
      s16x8 f;
      __builtin_memcpy(&f, &v, 16);
      return f;
    };
    auto read_b = [&](const int (&o)[2], int d) RD_INLINE_LAMBDA {
      const uint2 lo = lds_read_tr16_b64(bx + o[0] + d), hi = lds_read_tr16_b64(bx + o[1] + d);
      const uint4 v = make_uint4(lo.x, lo.y, hi.x, hi.y);
      s16x8 f;
      __builtin_memcpy(&f, &v, 16);
      return f;
    };
    // read pipeline: the operand of MFMA n + 1 is requested before MFMA n is issued (LDS returns in order: the wait in front of an MFMA leaves
    // the younger request outstanding), across taps and across k-steps; the last k-step re-reads its own first operands
    int o[2];
    offs(0, o);
    s16x8 a_cur = read_a(0), b_cur = read_b(o, 0);
    for (int s_ = 0; s_ < g.ks; s_++) {
      const int sn = s_ + 1 < g.ks ? s_ + 1 : s_;
      int on[2];
      offs(sn, on);
      const s16x8 a_nxt = read_a(sn);
#pragma unroll
      for (int tap = 0; tap < 9; tap++) {
        const s16x8 b_nxt = tap < 8 ? read_b(o, (((tap + 1) / 3) * WT + (tap + 1) % 3) * 32) : read_b(on, 0);
        acc[tap] = mfma_32x32x16_bf16(a_cur, b_cur, acc[tap]);
        b_cur = b_nxt;
      }
      a_cur = a_nxt; o[0] = on[0]; o[1] = on[1];
    }
  };

  int buf = 0;
  if (tile < tend) issue(tile, 0);
  dma_wait_all();
  __syncthreads();
  while (tile < tend) {
    const int next = tile + G8;
    if (next < tend) issue(next, buf ^ 1);      // every wave left buffer buf ^ 1 before the barrier it just passed
    compute(buf);
    dma_wait_all();
    __syncthreads();
    tile = next;
    buf ^= 1;
  }

  // ---- slab: register v of a tap's tile = output channel (v & 3) + 8 (v >> 2) + 4 (lane >> 5), input channel lane & 31 -----------------------
  float* slab = a.slab + (int64_t)blockIdx.x * a.Cout * a.K;
  const int ci = ci0 + hci * 32 + (lane & 31);
#pragma unroll
  for (int j = 0; j < 9; j++) {
    const int k = j * CinT + ci;
#pragma unroll
    for (int v = 0; v < 16; v++) {
      const int co = co0 + hco * 32 + (v & 3) + 8 * (v >> 2) + 4 * (lane >> 5);
      if (co < a.Cout) slab[(int64_t)co * a.K + k] = acc[j][v];
    }
  }
}

// ---- host side -------------------------------------------------------------------------------------------------------------------
// narrow layers (Cin in {16, 32, 64}, Cout <= 32): one slice.  Wide layers (Cin % 64 == 0): (Cin/64) x ceil(Cout/32) slices, used when the
// 8 x 16 tiles cover the feature map well enough (the MFMA work is spent on whole tiles).
static bool tr_geom(const WgradArgs& a) {
  return a.KH == 3 && a.KW == 3 && a.stride == 1 && a.pad == 1 && (a.C1 % 8 == 0) && a.OH == a.Hin && a.OW == a.Win;
}
static bool tr_narrow(const WgradArgs& a) { const int Cin = a.C1 + a.C2; return (Cin == 16 || Cin == 32 || Cin == 64) && a.Cout <= 32; }
static int tr_tw(const WgradArgs& a) {    // tile width with the smallest padded area (8 x TW tiles); 32 only where the LDS budget allows
  const int Cin = a.C1 + a.C2;
  int best = 16;
  int64_t area = cdiv(a.OW, 16) * 16;
  if (cdiv(a.OW, 8) * 8 < area) { best = 8; area = cdiv(a.OW, 8) * 8; }
  if (Cin < 64 && cdiv(a.OW, 32) * 32 <= area) best = 32;
  if (rd_opt_is_set(OPT_WGRAD_TR_TW)) { const int v = rd_opt(OPT_WGRAD_TR_TW, 16); if (v == 8 || v == 16 || (v == 32 && Cin < 64)) best = v; }   // experiment hook (rd_set_option)
  return best;
}
// map-fitted kernel (conv3x3_wgrad_fit_kernel): wide layers on narrow maps; option wgrad_fit: 0 never, 1 wherever it fits, unset = where its
// tiles waste clearly less than the 8 x TW tiles above
struct FitPlan { int TH, tilesH, npx, tpx, ks, nci, nco; double eff; };
static bool fit_plan(const WgradArgs& a, int dtype, FitPlan& p) {
  const int Cin = a.C1 + a.C2;
  if (dtype != 1 || !tr_geom(a) || a.in_scale || Cin % 64 != 0 || a.Cout % 8 != 0 || a.Cout < 64 || a.OW > 50) return false;
  if ((int64_t)a.N * a.Hin * a.Win >= (int64_t)1 << 31) return false;
  const int force = rd_opt(OPT_WGRAD_FIT, 0);
  if (force == 0) return false;
  int best = 0; double beff = 0.0;
  for (int th = 1; th <= a.OH && th * a.OW <= 128; th++) {
    if ((th + 2) * (a.OW + 2) > 160) break;
    const int ks = (int)cdiv(th * a.OW, 16);
    const double eff = (double)a.OH * a.OW / ((double)cdiv(a.OH, th) * ks * 16);
    if (eff > beff + 1e-9) { beff = eff; best = th; }
  }
  if (!best) return false;
  p.TH = best; p.tilesH = (int)cdiv(a.OH, best); p.npx = (best + 2) * (a.OW + 2); p.tpx = best * a.OW; p.ks = (int)cdiv(p.tpx, 16);
  p.nci = Cin / 64; p.nco = (int)cdiv(a.Cout, 64); p.eff = beff;
  // Measured on MI355X (round 5, tools/r05_wgradfit.sh -> profiles/r05_microbench/wgrad_fit_ab.txt), bit-exact in both versions:
  //   v1 (register staging, 256 registers, every k-step's twenty transpose reads in front of its nine MFMAs): 585-588 TFLOP/s on 86 400 pixels
  //      256 -> 128 against 584-639 for the 8 x TW kernel, the RC-Net step 1037.8 against 1045.0 img/s;
  //   v2 (LDS-DMA staging, 206 registers, operand reads pipelined one MFMA ahead -- the ISA shows 2-5 reads outstanding at every wait):
  //      636 / 604-620 against 587-649 / 575-630 on 86 400 pixels, 424-434 against 392-432 on 21 600 pixels 384 -> 256, 673-685 against
  //      622-694 on 360 000 pixels 128 -> 64; the step 1077.9 against 1084.3 img/s.
  // A quarter fewer MFMAs, a third less staging and a pipelined inner loop land where the old kernel is.  What the two have in common is the
  // split over pixels: 512 persistent blocks each write a (64 x 576 or 32 x 576) fp32 slab slice at the same moment at the end -- 75 MB (here)
  // / 38 MB (8 x TW kernel) per launch for 51 GFLOP (= 51 us at 1 PFLOP/s) -- and the reduction pass reads them again: at these sizes the
  // slab traffic is a first-order term that no change inside the tile loop touches.  Kept as option wgrad_fit = 1 (tests: wgrad_fit_cases).
  return force == 1;
}
static int fit_blocks(const WgradArgs& a, const FitPlan& p) {
  const int64_t ntiles = (int64_t)a.N * p.tilesH;
  int cap = std::max(1, 64 / (p.nci * p.nco));      // 512 resident blocks (two per CU) over the slices, in units of 8 (one per XCD)
  if (rd_opt_is_set(OPT_CONV3X3_G8)) cap = rd_opt(OPT_CONV3X3_G8, 1);
  else cap = (int)std::max<int64_t>(1, std::min<int64_t>(cap, ntiles / 32));      // >= 4 tiles per block: every block writes a slab slice
  return 8 * (int)std::min<int64_t>(cdiv(ntiles, 8), cap);
}
bool wgrad3x3_tr_ok(const WgradArgs& a, int dtype) {
  { FitPlan fp; if (fit_plan(a, dtype, fp)) return true; }
  if (dtype != 1 || !tr_geom(a)) return false;
  if ((int64_t)a.N * a.Hin * a.Win >= (int64_t)1 << 31) return false;   // the kernel keeps pixel indices in 32 bits
  if (tr_narrow(a)) return true;
  const int Cin = a.C1 + a.C2;
  if (Cin % 64 != 0 || a.Cout % 8 != 0) return false;
  const int tw = tr_tw(a);
  const double eff = (double)a.OH * a.OW / (double)(cdiv(a.OH, 8) * 8 * cdiv(a.OW, tw) * tw);
  return eff >= 0.6;
}
bool wgrad3x3_tr_affine_ok(const WgradArgs& a, int dtype) {      // (asked WITHOUT in_scale set: the fitted kernel has no affine form, the 8 x TW kernel must take the shape)
  if (dtype != 1 || !tr_geom(a)) return false;
  if ((int64_t)a.N * a.Hin * a.Win >= (int64_t)1 << 31) return false;
  if (tr_narrow(a)) return true;
  const int Cin = a.C1 + a.C2;
  if (Cin % 64 != 0 || a.Cout % 8 != 0) return false;
  const int tw = tr_tw(a);
  return (double)a.OH * a.OW / (double)(cdiv(a.OH, 8) * 8 * cdiv(a.OW, tw) * tw) >= 0.6;
}
static void tr_slices(const WgradArgs& a, int& cti, int& rt, int& nci, int& nco) {
  const int Cin = a.C1 + a.C2;
  cti = Cin >= 64 ? 4 : Cin / 16;
  rt = a.Cout <= 16 ? 1 : 2;
  // Measured and not kept (tools/bench_conv.py, eleven RC-Net wgrad shapes, 0.773 ms with the kernel as it is): 64-channel output
  // slices in four waves (RT = 4, 144 accumulator registers, spills) 0.938 ms; eight-wave blocks (NWV = 8, four waves per SIMD) 0.807 ms;
  // both together 0.843 ms; linear tiling (tools/probes/wgrad3x3_lin_kernel.hip, removed in round 4) 0.893 ms.  Probes: with the MFMA phase removed a launch takes 55 % of its time, with the
  // global fetches removed 91 % -- the LDS phase (stash, barrier, 88 transpose reads and 72 MFMAs per wave and tile) is what binds,
  // at ~620 TFLOP/s whatever the register tile.
  nci = Cin >= 64 ? Cin / 64 : 1;
  nco = (int)cdiv(a.Cout, rt * 16);
}
int wgrad3x3_tr_blocks(const WgradArgs& a) {   // persistent blocks per slice = slabs to reduce
  { FitPlan fp; if (fit_plan(a, 1, fp)) return fit_blocks(a, fp); }
  int cti, rt, nci, nco;
  tr_slices(a, cti, rt, nci, nco);
  const int tw = tr_tw(a);
  const int64_t ntiles = (int64_t)a.N * cdiv(a.OH, 8) * cdiv(a.OW, tw);
  const bool e = rd_opt_is_set(OPT_CONV3X3_G8);  // test hook shared with the forward kernel (rd_set_option "conv3x3_g8")
  // persistent grid = resident capacity: 4 blocks per CU for the light variants, 2 where registers (launch bounds) or LDS allow only two
  // (measured: 512 instead of 1024 blocks is 8-13 % faster on the 64-channel slices and halves the slab traffic)
  const int Cin = a.C1 + a.C2;
  const int lds = 2 * 16 * ((8 + 2) * (tw + 2) * (cti * 16 / 8) + 8 * tw * (rt * 16 / 8));
  const int per_cu = (cti * rt <= 2 && lds * 4 <= 160 * 1024) ? 4 : 2;
  (void)Cin;
  int cap = e ? rd_opt(OPT_CONV3X3_G8, 1) : std::max(1, 32 * per_cu / (nci * nco));
  // every block writes (and the reduction re-reads) a Cout x K slab slice: wide layers with few tiles keep >= 4 tiles per block
  if (!e && nci * nco > 1) cap = (int)std::max<int64_t>(1, std::min<int64_t>(cap, ntiles / 32));
  return 8 * (int)std::min<int64_t>(cdiv(ntiles, 8), cap);
}
const char* wgrad3x3_tr_name(const WgradArgs& a) {
  static thread_local char buf[64];
  { FitPlan fp; if (fit_plan(a, 1, fp)) return "conv3x3_wgrad_fit_kernel<4>"; }
  int cti, rt, nci, nco;
  tr_slices(a, cti, rt, nci, nco);
  snprintf(buf, sizeof(buf), "conv3x3_wgrad_tr_kernel<%d, %d, %d, 4, %s>", cti, rt, tr_tw(a), a.in_scale ? "true" : "false");
  return buf;
}
void launch_wgrad3x3_tr(const WgradArgs& a, hipStream_t st) {
  { FitPlan fp;
    if (fit_plan(a, 1, fp)) {
      FitGeom g;
      g.TH = fp.TH; g.tilesH = fp.tilesH; g.npx = fp.npx; g.tpx = fp.tpx; g.ks = fp.ks; g.nci = fp.nci;
      g.rOW = 1.0f / (float)a.OW; g.rWT = 1.0f / (float)(a.OW + 2);
      const dim3 grid((unsigned)fit_blocks(a, fp), (unsigned)(fp.nci * fp.nco));
      hipLaunchKernelGGL((conv3x3_wgrad_fit_kernel<4>), grid, dim3(256), 0, st, a, g);
      return;
    } }
  int cti, rt, nci, nco;
  tr_slices(a, cti, rt, nci, nco);
  const dim3 grid((unsigned)wgrad3x3_tr_blocks(a), (unsigned)(nci * nco));
  const int tw = tr_tw(a);
  const int tilesH = (int)cdiv(a.OH, 8), tilesW = (int)cdiv(a.OW, tw);
  const bool aff = a.in_scale != nullptr;
#define RD_TR(CTIV, RTV, TWV) \
  if (cti == CTIV && rt == RTV && tw == TWV) { \
    if (aff) hipLaunchKernelGGL((conv3x3_wgrad_tr_kernel<CTIV, RTV, TWV, 4, true>), grid, dim3(256), 0, st, a, tilesH, tilesW, nci); \
    else hipLaunchKernelGGL((conv3x3_wgrad_tr_kernel<CTIV, RTV, TWV>), grid, dim3(256), 0, st, a, tilesH, tilesW, nci); }
  RD_TR(1, 1, 8) RD_TR(1, 1, 16) RD_TR(1, 1, 32) RD_TR(1, 2, 8) RD_TR(1, 2, 16) RD_TR(1, 2, 32)
  RD_TR(2, 1, 8) RD_TR(2, 1, 16) RD_TR(2, 1, 32) RD_TR(2, 2, 8) RD_TR(2, 2, 16) RD_TR(2, 2, 32)
  RD_TR(4, 1, 8) RD_TR(4, 1, 16) RD_TR(4, 2, 8) RD_TR(4, 2, 16)
#undef RD_TR
}

}  // namespace rd
