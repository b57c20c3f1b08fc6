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

// LDS layout note (measured, profiles/r01_wgrad_tr_lds_swizzle.txt): with the plain [pixel][channel] images the 32-lane transpose reads
// of the 64-channel slices hit 2..4 bank groups (SQ_LDS_BANK_CONFLICT = 66 % of SQ_LDS_IDX_ACTIVE).  An XOR swizzle of the 16-channel tile
// position with pixel bits 1 and 3 removed every conflict (counter = 0, LDS cycles / 3) and made the kernel SLOWER (0.210 -> 0.224 ms
// wide, 0.102 -> 0.140 ms narrow): the per-read address arithmetic replaces immediate offsets and the kernel is issue / latency bound
// with one wave per SIMD, not LDS bound.  The affine layout stays.
template <int CTI, int RT, int TW>
__global__ __launch_bounds__(256, (CTI * RT <= 2) ? 4 : 2) void conv3x3_wgrad_tr_kernel(WgradArgs a, int tilesH, int tilesW, int nci) {
  typedef bf16_t T;
  constexpr int CIN = CTI * 16, COP = RT * 16;
  constexpr int TH = 8, WT = TW + 2, HT = TH + 2, NPX = HT * WT, NPY = TH * TW;
  constexpr int XS = CIN / 8, YS = COP / 8;           // 16-byte slots per pixel
  constexpr int NXS = NPX * XS, NYS = NPY * YS;
  constexpr int XIT = (NXS + 255) / 256, YIT = (NYS + 255) / 256;
  constexpr int KSTEPS = NPY / 32;                     // 32 pixels per MFMA k-step
  constexpr int NCT = 9 * CTI, NCW = (NCT + 3) / 4;    // (tap, cin-tile) column tiles, per wave
  constexpr int ASTEP = 32 * COP;                      // elements between k-steps in the dY tile (32 consecutive pixels)
  constexpr int BSTEP = (32 / TW) * WT * CIN;          // ... in the patch (four rows of 8, two rows of 16 or one row of 32)
  __shared__ uint4 sX[2][NXS];
  __shared__ uint4 sY[2][NYS];

  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  // wide layers: blockIdx.y picks a (CIN input channels) x (COP output channels) slice of the gradient; every slice walks all tiles
  const int ci0 = ((int)blockIdx.y % nci) * CIN, co0 = ((int)blockIdx.y / nci) * COP;
  const int CinT = a.C1 + a.C2;

  // persistent tile walk: XCD x owns a contiguous range of tiles (halo pixels shared by neighbouring tiles stay in its L2)
  const int ntiles = a.N * tilesH * tilesW;
  const int T8 = (ntiles + 7) >> 3, G8 = gridDim.x >> 3;
  const int xcd = blockIdx.x & 7;
  const int tend = min(ntiles, (xcd + 1) * T8);
  int tile = xcd * T8 + (blockIdx.x >> 3);

  // transpose-read roles.  k-step s, lane group fg: pixels (y, x0 .. x0+7) of the tile; this lane supplies pixel x0 + (fr >> 2) (+4
  // for the second read) and channels 4*(fr & 3)..+3 of whichever 16-channel tile is being read.
  constexpr int GPR = TW / 8;                          // lane groups (8 pixels each) per tile row
  const int yl = fg / GPR, x0 = (fg % GPR) * 8;
  const int aoff = ((yl * TW) + x0 + (fr >> 2)) * COP + (fr & 3) * 4;
  const int boff = ((yl * WT) + x0 + (fr >> 2)) * CIN + (fr & 3) * 4;
  int coloff[NCW]; bool jv[NCW]; int jk[NCW];
#pragma unroll
  for (int j = 0; j < NCW; j++) {
    int idx = wv + 4 * j;
    jv[j] = idx < NCT;
    if (!jv[j]) idx = 0;
    const int tap = idx / CTI, ct = idx - tap * CTI;
    coloff[j] = ((tap / 3) * WT + (tap % 3)) * CIN + ct * 16;
    jk[j] = tap * CinT + ci0 + ct * 16;
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
    const int idx = t + 256 * i;
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
    const int idx = t + 256 * i;
    const int pp = idx / YS, sl = idx - pp * YS;
    const int py = pp / TW, px = pp - py * TW;
    ypk[i] = (py << 8) | px | ((idx < NYS && co0 + sl * 8 < a.Cout) ? 1 << 16 : 0) | (sl << 20);
  }
  const int Hp = a.ups ? a.H1 : a.Hin, Wp = a.ups ? a.W1 : a.Win;
  auto fetch = [&](TC tc, uint4 (&rx)[XIT], uint4 (&ry)[YIT]) RD_INLINE_LAMBDA {
    const int n = tc.n;
    const int oh0 = tc.th * TH, ow0 = tc.tw * TW;
#pragma unroll
    for (int i = 0; i < XIT; i++) {
      uint4 v = make_uint4(0, 0, 0, 0);
      const int ih = oh0 - 1 + ((xpk[i] >> 8) & 0xff), iw = ow0 - 1 + (xpk[i] & 0xff);
      if ((xpk[i] >> 17) && (unsigned)ih < (unsigned)a.Hin && (unsigned)iw < (unsigned)a.Win) {
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
  auto stash = [&](int buf, const uint4 (&rx)[XIT], const uint4 (&ry)[YIT]) RD_INLINE_LAMBDA {
#pragma unroll
    for (int i = 0; i < XIT; i++) { const int idx = t + 256 * i; if (idx < NXS) sX[buf][idx] = rx[i]; }
#pragma unroll
    for (int i = 0; i < YIT; i++) { const int idx = t + 256 * i; if (idx < NYS) sY[buf][idx] = ry[i]; }
  };
  auto compute = [&](int buf, auto njc) RD_INLINE_LAMBDA {
    constexpr int NJ = decltype(njc)::value;
    const unsigned short* bx = reinterpret_cast<const unsigned short*>(&sX[buf][0]) + boff;
    const unsigned short* by = reinterpret_cast<const unsigned short*>(&sY[buf][0]) + aoff;
#pragma unroll
    for (int s = 0; s < KSTEPS; s++) {
      s16x8 ya[RT];
#pragma unroll
      for (int i = 0; i < RT; i++) {
        uint2 lo = lds_read_tr16_b64(by + s * ASTEP + i * 16), hi = lds_read_tr16_b64(by + s * ASTEP + i * 16 + 4 * COP);
        uint4 v = make_uint4(lo.x, lo.y, hi.x, hi.y);
        __builtin_memcpy(&ya[i], &v, 16);
      }
#pragma unroll
      for (int j = 0; j < NJ; j++) {
        uint2 lo = lds_read_tr16_b64(bx + s * BSTEP + coloff[j]), hi = lds_read_tr16_b64(bx + s * BSTEP + coloff[j] + 4 * CIN);
        uint4 v = make_uint4(lo.x, lo.y, hi.x, hi.y);
        s16x8 xb;
        __builtin_memcpy(&xb, &v, 16);
#pragma unroll
        for (int i = 0; i < RT; i++) acc[i][j] = mfma_16x16x32_bf16(ya[i], xb, acc[i][j]);
      }
    }
  };

  // two fully unrolled bodies (a wave owns NCW or NCW-1 column tiles): a per-MFMA branch makes the compiler shuttle accumulators
  auto tile_body = [&](int buf) RD_INLINE_LAMBDA {
    if (jv[NCW - 1]) compute(buf, std::integral_constant<int, NCW>{});
    else compute(buf, std::integral_constant<int, NCW - 1>{});
  };
  // prefetch distance two tiles where the register budget allows (narrow slices), one otherwise
  constexpr bool DEEP = (XIT + YIT) <= 5;
  uint4 xa[XIT], ya_[YIT], xb_[DEEP ? XIT : 1], yb_[DEEP ? YIT : 1];
  int buf = 0;
  if (DEEP) {
    int t0 = tile, t1 = tile + G8;
    TC c1 = advance(decode(t0));
    if (t0 < tend) fetch(decode(t0), xa, ya_);
    if (t1 < tend) fetch(c1, reinterpret_cast<uint4 (&)[XIT]>(xb_), reinterpret_cast<uint4 (&)[YIT]>(yb_));
    while (t0 < tend) {
      stash(buf, xa, ya_);
      __syncthreads();
      { const int t2 = t1 + G8; c1 = advance(c1); if (t2 < tend) fetch(c1, xa, ya_); tile_body(buf); t0 = t1; t1 = t2; buf ^= 1; }
      if (t0 >= tend) break;
      stash(buf, reinterpret_cast<const uint4 (&)[XIT]>(xb_), reinterpret_cast<const uint4 (&)[YIT]>(yb_));
      __syncthreads();
      { const int t2 = t1 + G8; c1 = advance(c1); if (t2 < tend) fetch(c1, reinterpret_cast<uint4 (&)[XIT]>(xb_), reinterpret_cast<uint4 (&)[YIT]>(yb_)); tile_body(buf); t0 = t1; t1 = t2; buf ^= 1; }
    }
  } else {
    TC c0 = decode(tile);
    if (tile < tend) fetch(c0, xa, ya_);
    while (tile < tend) {
      stash(buf, xa, ya_);
      __syncthreads();
      const int next = tile + G8;
      c0 = advance(c0);
      if (next < tend) fetch(c0, xa, ya_);
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
        const int k = jk[j] + fr;
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int co = co0 + i * 16 + fg * 4 + r;
          if (co < a.Cout) slab[(int64_t)co * a.K + k] = acc[i][j][r];
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
  if (const char* e = getenv("RD_WGRAD_TR_TW")) { const int v = atoi(e); if (v == 8 || v == 16 || (v == 32 && Cin < 64)) best = v; }   // experiment hook
  return best;
}
bool wgrad3x3_tr_ok(const WgradArgs& a, int dtype) {
  if (dtype != 1 || !tr_geom(a)) return false;
  if ((int64_t)a.N * a.Hin * a.Win >= (int64_t)1 << 31) return false;   // the kernel keeps pixel indices in 32 bits
  if (tr_narrow(a)) return true;
  const int Cin = a.C1 + a.C2;
  if (Cin % 64 != 0 || a.Cout % 8 != 0) return false;
  const int tw = tr_tw(a);
  const double eff = (double)a.OH * a.OW / (double)(cdiv(a.OH, 8) * 8 * cdiv(a.OW, tw) * tw);
  return eff >= 0.6;
}
static void tr_slices(const WgradArgs& a, int& cti, int& rt, int& nci, int& nco) {
  const int Cin = a.C1 + a.C2;
  cti = Cin >= 64 ? 4 : Cin / 16;
  rt = a.Cout <= 16 ? 1 : 2;
  // wide layers, 64-channel output slices (RD_WGRAD_TR_RT4, A/B): 26 instead of 22 transpose reads per k-step feed 36 instead of 18 MFMAs, and
  // the input patch is staged for half as many slices
  static const int rt4 = getenv("RD_WGRAD_TR_RT4") ? atoi(getenv("RD_WGRAD_TR_RT4")) : 0;
  if (rt4 && Cin >= 64 && (Cin % 64) == 0 && a.Cout >= 64) rt = 4;
  nci = Cin >= 64 ? Cin / 64 : 1;
  nco = (int)cdiv(a.Cout, rt * 16);
}
int wgrad3x3_tr_blocks(const WgradArgs& a) {   // persistent blocks per slice = slabs to reduce
  int cti, rt, nci, nco;
  tr_slices(a, cti, rt, nci, nco);
  const int tw = tr_tw(a);
  const int64_t ntiles = (int64_t)a.N * cdiv(a.OH, 8) * cdiv(a.OW, tw);
  const char* e = getenv("RD_CONV3X3_G8");  // test hook shared with the forward kernel
  // persistent grid = resident capacity: 4 blocks per CU for the light variants, 2 where registers (launch bounds) or LDS allow only two
  // (measured: 512 instead of 1024 blocks is 8-13 % faster on the 64-channel slices and halves the slab traffic)
  const int Cin = a.C1 + a.C2;
  const int lds = 2 * 16 * ((8 + 2) * (tw + 2) * (cti * 16 / 8) + 8 * tw * (rt * 16 / 8));
  const int per_cu = (cti * rt <= 2 && lds * 4 <= 160 * 1024) ? 4 : 2;
  (void)Cin;
  int cap = e ? atoi(e) : std::max(1, 32 * per_cu / (nci * nco));
  // every block writes (and the reduction re-reads) a Cout x K slab slice: wide layers with few tiles keep >= 4 tiles per block
  if (!e && nci * nco > 1) cap = (int)std::max<int64_t>(1, std::min<int64_t>(cap, ntiles / 32));
  return 8 * (int)std::min<int64_t>(cdiv(ntiles, 8), cap);
}
const char* wgrad3x3_tr_name(const WgradArgs& a) {
  static thread_local char buf[64];
  int cti, rt, nci, nco;
  tr_slices(a, cti, rt, nci, nco);
  snprintf(buf, sizeof(buf), "conv3x3_wgrad_tr_kernel<%d, %d, %d>", cti, rt, tr_tw(a));
  return buf;
}
void launch_wgrad3x3_tr(const WgradArgs& a, hipStream_t st) {
  int cti, rt, nci, nco;
  tr_slices(a, cti, rt, nci, nco);
  const int tw = tr_tw(a);
  const int tilesH = (int)cdiv(a.OH, 8), tilesW = (int)cdiv(a.OW, tw);
  const dim3 grid((unsigned)wgrad3x3_tr_blocks(a), (unsigned)(nci * nco));
#define RD_TR(CTIV, RTV, TWV) \
  if (cti == CTIV && rt == RTV && tw == TWV) hipLaunchKernelGGL((conv3x3_wgrad_tr_kernel<CTIV, RTV, TWV>), grid, dim3(256), 0, st, a, tilesH, tilesW, nci);
  RD_TR(1, 1, 8) RD_TR(1, 1, 16) RD_TR(1, 1, 32) RD_TR(1, 2, 8) RD_TR(1, 2, 16) RD_TR(1, 2, 32)
  RD_TR(2, 1, 8) RD_TR(2, 1, 16) RD_TR(2, 1, 32) RD_TR(2, 2, 8) RD_TR(2, 2, 16) RD_TR(2, 2, 32)
  RD_TR(4, 1, 8) RD_TR(4, 1, 16) RD_TR(4, 2, 8) RD_TR(4, 2, 16) RD_TR(4, 4, 8) RD_TR(4, 4, 16)
#undef RD_TR
}

}  // namespace rd
