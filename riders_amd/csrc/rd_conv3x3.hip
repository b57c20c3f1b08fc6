// Patch-staged 3x3 / stride-1 convolution (forward and data gradient) for gfx950.
//
// The implicit-GEMM kernel in rd_conv.hip re-gathers every input pixel once per filter tap; rocprofv3 showed those nine passes are
// NOT absorbed by L1/L2 at RC-Net's sizes (per-CU load rate ~12 B/clk = the HBM-bound rate, MFMA busy 12 %), so 3x3 layers ran at the
// HBM roofline of 9x their algorithmic bytes.  Here a block owns a 128-pixel output tile (16x8 or 8x16 pixels), stages the
// (rows+2) x (cols+2) input patch ONCE per 128-byte channel chunk in LDS (nearest-upsample and two-source concat folded into the
// staging gather, zero fill = padding) and walks the nine taps out of LDS; only the packed weights stream per tap (double-buffered).
// MFMA roles, 16-byte XOR-swizzled LDS slots, epilogue (bias / activation / dual destination / BatchNorm partials) are the ones of
// the implicit-GEMM kernel.  Replaces the same reference call sites (utils/net_utils.py:84-91,195-198,564-569) for k=3, s=1.
#include "rd_conv_common.h"
#include <stdio.h>

namespace rd {

template <typename T, int BN, bool W8>
__global__ __launch_bounds__(256) void conv3x3_patch_kernel(ConvArgs a, int tilesH, int tilesW) {
  constexpr int VE = Elem<T>::VE;
  constexpr int CKE = STAGE_BYTES / (int)sizeof(T);     // channels per chunk (128 bytes per pixel)
  constexpr int TW = W8 ? 8 : 16, TH = W8 ? 16 : 8;      // 128 output pixels
  constexpr int WT = TW + 2, HT = TH + 2, NP = HT * WT; // patch pixels (180)
  constexpr int PIT = (NP * 8 + 255) / 256;             // patch vectors per thread (6)
  constexpr int CT = BN / 16;
  constexpr int BITER = (BN * 8 + 255) / 256;
  __shared__ uint4 sP[NP * 8];
  __shared__ uint4 sB[2][BN * 8];

  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  int bx = blockIdx.x;
  {  // XCD-aware order (see rd_conv.hip)
    const int nb = gridDim.x, q = nb >> 3, r = nb & 7, xcd = bx & 7, idx = bx >> 3;
    bx = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int n0 = blockIdx.y * BN;
  const int Cin = a.C1 + a.C2;
  const int tw_ = bx % tilesW; const int q_ = bx / tilesW; const int th_ = q_ % tilesH; const int n = q_ / tilesH;
  const int oh0 = th_ * TH, ow0 = tw_ * TW;
  const int nchunk = Cin / CKE;

  // patch staging role: idx = t + 256*i -> pixel pp = idx >> 3, 16-byte slot = idx & 7 (= t & 7 for every i).  A block owns ONE tile, so
  // the source pixel of each of this thread's slots is decoded once (bounds, nearest-upsample index; -1 = padding) and a chunk only
  // picks the source tensor for its channel offset: one 64-bit multiply-add per load instead of the general gather (conv_src_ptr,
  // ~45 VALU instructions per load -- a quarter of the issue cycles of this kernel next to its MFMAs).  dil == 1 (geom3x3); pixel
  // indices fit 32 bits (conv3x3_ok).
  uint4 rp[PIT], rb[BITER];
  int spix[PIT];
  {
    const int Hp = a.ups ? a.H1 : a.Hin, Wp = a.ups ? a.W1 : a.Win;
#pragma unroll
    for (int i = 0; i < PIT; i++) {
      const int pp = (t + 256 * i) >> 3;
      const int py = pp / WT, px = pp - py * WT;
      const int ih = oh0 - 1 + py, iw = ow0 - 1 + px;
      int pix = -1;
      if (pp < NP && (unsigned)ih < (unsigned)a.Hin && (unsigned)iw < (unsigned)a.Win) {
        int hs = ih, ws = iw;
        if (a.ups) {  // F.interpolate(mode='nearest') source index, ATen float formula (as conv_src_ptr)
          hs = min((int)floorf((float)ih * a.scale_h), a.H1 - 1);
          ws = min((int)floorf((float)iw * a.scale_w), a.W1 - 1);
        }
        pix = (n * Hp + hs) * Wp + ws;
      }
      spix[i] = pix;
    }
  }
  auto load_patch = [&](int chunk) RD_INLINE_LAMBDA {
    const int ci = chunk * CKE + (t & 7) * VE;
    const T* cb; int cs;
    if (ci < a.C1) { cb = (const T*)a.src1 + ci; cs = a.C1; } else { cb = (const T*)a.src2 + (ci - a.C1); cs = a.C2; }
#pragma unroll
    for (int i = 0; i < PIT; i++) {
      uint4 v = make_uint4(0, 0, 0, 0);
      if (spix[i] >= 0) v = *reinterpret_cast<const uint4*>(cb + (int64_t)spix[i] * cs);
      rp[i] = v;
    }
  };
  auto store_patch = [&]() RD_INLINE_LAMBDA {
#pragma unroll
    for (int i = 0; i < PIT; i++) {
      int idx = t + 256 * i;
      int pp = idx >> 3, sl = idx & 7;
      if (pp < NP) sP[lds_slot(pp, sl)] = rp[i];
    }
  };
  const uint4* wp = reinterpret_cast<const uint4*>(a.w);
  const int kslots = a.Kpad / VE;
  auto load_w = [&](int chunk, int tap) RD_INLINE_LAMBDA {  // packed row = [tap][Cin]: 128 contiguous bytes per output channel
    const int k0 = (tap * Cin + chunk * CKE) / VE;
#pragma unroll
    for (int i = 0; i < BITER; i++) {
      int idx = t + 256 * i;
      if (BN * 8 % 256 == 0 || idx < BN * 8) {
        const uint4 v = wp[(int64_t)(n0 + (idx >> 3)) * kslots + k0 + (idx & 7)];  // via a value (see rd_conv.hip: avoids a scratch alloca)
        rb[i] = v;
      }
    }
  };
  auto store_w = [&](int buf) RD_INLINE_LAMBDA {
#pragma unroll
    for (int i = 0; i < BITER; i++) {
      int idx = t + 256 * i;
      if (BN * 8 % 256 == 0 || idx < BN * 8) sB[buf][lds_slot(idx >> 3, idx & 7)] = rb[i];
    }
  };

  // wave tiling: 4 waves x (32 pixels x BN channels), or for BN = 128 a square 2 x 2 arrangement of (64 pixels x 64 channels) per wave:
  // 4 + 4 instead of 2 + 8 fragment reads per 16 MFMAs (the kernel is LDS-read bound: measured ~10 ds_read_b128 per MFMA pair)
  constexpr bool SQ = (BN == 128);
  constexpr int NPT = SQ ? 4 : 2, CTW = SQ ? CT / 2 : CT;
  const int wpx = SQ ? (wv & 1) : wv, wc = SQ ? (wv >> 1) : 0;
  // this lane's output pixels (one per MFMA pixel tile) inside the tile, as patch coordinates of tap (0,0)
  int ppix[NPT];
#pragma unroll
  for (int pt = 0; pt < NPT; pt++) {
    const int ptile = SQ ? wpx * 4 + pt : wv * 2 + pt;
    int py = W8 ? (ptile * 2 + (fr >> 3)) : ptile;
    int px = W8 ? (fr & 7) : fr;
    ppix[pt] = py * WT + px;
  }
  // LDS fragment addresses without per-tap swizzle arithmetic (a PMC pass counted 787 VALU instructions per wave and tile next to 72
  // MFMAs on the 64->32 layer; the XOR-swizzled slot of every fragment read was recomputed per tap): the weight-fragment slots do not
  // depend on the tap and are formed once; a pixel's nine XOR terms ((row + tap offset) >> 1) & 7 are packed three bits each into one
  // register, so a patch fragment address is a bit-field extract, an XOR and an add.
  int wfa[2][CTW];
#pragma unroll
  for (int ch = 0; ch < 2; ch++)
#pragma unroll
    for (int c = 0; c < CTW; c++) wfa[ch][c] = lds_slot((wc * CTW + c) * 16 + fr, ch * 4 + fg);
  unsigned pxor[NPT];
#pragma unroll
  for (int pt = 0; pt < NPT; pt++) {
    unsigned pk = 0;
#pragma unroll
    for (int tp = 0; tp < 9; tp++) pk |= (unsigned)(((ppix[pt] + (tp / 3) * WT + (tp % 3)) >> 1) & 7) << (3 * tp);
    pxor[pt] = pk;
  }

  f32x4 acc[CTW][NPT];
#pragma unroll
  for (int c = 0; c < CTW; c++)
#pragma unroll
    for (int pt = 0; pt < NPT; pt++) acc[c][pt] = f32x4{0, 0, 0, 0};

  load_patch(0);
  load_w(0, 0);
  int wbuf = 0;
  for (int chunk = 0; chunk < nchunk; chunk++) {
    __syncthreads();            // every wave is done with the previous patch chunk and weight buffers
    store_patch();
    store_w(wbuf);
    __syncthreads();
    if (chunk + 1 < nchunk) load_patch(chunk + 1);   // next chunk's patch travels while the nine taps run
#pragma unroll 1
    for (int tap = 0; tap < 9; tap++) {
      const bool more = (tap < 8) || (chunk + 1 < nchunk);
      if (more) load_w(tap < 8 ? chunk : chunk + 1, tap < 8 ? tap + 1 : 0);
      const int kh = tap / 3, kw = tap - kh * 3;
      const int toff = kh * WT + kw;
#pragma unroll
      for (int ch = 0; ch < 2; ch++) {
        uint4 pf[NPT];
#pragma unroll
        for (int pt = 0; pt < NPT; pt++) pf[pt] = sP[(ppix[pt] + toff) * 8 + ((ch * 4 + fg) ^ (int)((pxor[pt] >> (3 * tap)) & 7u))];   // = lds_slot(ppix + toff, ch*4 + fg)
#pragma unroll
        for (int c = 0; c < CTW; c++) {
          uint4 wf = sB[wbuf][wfa[ch][c]];
#pragma unroll
          for (int pt = 0; pt < NPT; pt++) {
            if (sizeof(T) == 4) {
              acc[c][pt] = mfma_16x16x4_f32(__uint_as_float(wf.x), __uint_as_float(pf[pt].x), acc[c][pt]);
              acc[c][pt] = mfma_16x16x4_f32(__uint_as_float(wf.y), __uint_as_float(pf[pt].y), acc[c][pt]);
              acc[c][pt] = mfma_16x16x4_f32(__uint_as_float(wf.z), __uint_as_float(pf[pt].z), acc[c][pt]);
              acc[c][pt] = mfma_16x16x4_f32(__uint_as_float(wf.w), __uint_as_float(pf[pt].w), acc[c][pt]);
            } else {
              s16x8 wa, pb;
              __builtin_memcpy(&wa, &wf, 16);
              __builtin_memcpy(&pb, &pf[pt], 16);
              acc[c][pt] = mfma_16x16x32_bf16(wa, pb, acc[c][pt]);
            }
          }
        }
      }
      if (tap < 8) {           // publish the next tap's weights (the last tap's successor is published with the next patch)
        store_w(wbuf ^ 1);
        __syncthreads();
        wbuf ^= 1;
      }
    }
    wbuf ^= 1;
  }

  __syncthreads();  // all waves finished reading the patch before it is reused as reduction scratch
  float ssum[CTW][4], ssq[CTW][4];
#pragma unroll
  for (int c = 0; c < CTW; c++)
#pragma unroll
    for (int r = 0; r < 4; r++) { ssum[c][r] = 0.f; ssq[c][r] = 0.f; }
#pragma unroll
  for (int pp = 0; pp < NPT / 2; pp++) {   // the store routine takes two pixel tiles at a time
    int64_t mm[2]; bool mvv[2]; f32x4 a2[CTW][2];
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const int pt = pp * 2 + h;
      const int ptile = SQ ? wpx * 4 + pt : wv * 2 + pt;
      int py = W8 ? (ptile * 2 + (fr >> 3)) : ptile;
      int px = W8 ? (fr & 7) : fr;
      int oh = oh0 + py, ow = ow0 + px;
      mvv[h] = oh < a.OH && ow < a.OW;
      mm[h] = ((int64_t)n * a.OH + oh) * a.OW + ow;
#pragma unroll
      for (int c = 0; c < CTW; c++) a2[c][h] = acc[c][pt];
    }
    conv_epilogue_store<T, CTW>(a, a2, mm, mvv, n0, wc, fr, fg, ssum, ssq);
  }
  conv_epilogue_stats<CTW, BN, SQ ? 2 : 4>(a, ssum, ssq, n0, wc, wpx, fr, fg, t, bx, reinterpret_cast<float*>(&sP[0]));
}

// ---- narrow layers (Cin * sizeof(T) = 32 / 64 / 128 bytes, Cout <= 32): persistent blocks, weights in registers ---------------------
// RC-Net's decoder tail runs at ROI resolution (5.76 M output pixels for the ZJU config) with 16..32 channels: the layers are HBM-bound
// (algorithmic bytes / 8 TB/s = 50..70 us) and the implicit-GEMM kernel spent 0.3..0.5 ms on them (9x re-gather).  Here the whole K axis
// (9 taps x Cin) of a 128-pixel tile is one LDS patch; a lane keeps its weight fragments for all taps in registers, blocks are
// persistent (tile loop, next patch prefetched into registers, double-buffered LDS, one barrier per tile) and every XCD walks a
// contiguous range of tiles so that halo pixels shared by neighbouring tiles are L2 hits.
// occupancy target of a variant: four waves per SIMD while four blocks fit the LDS (patch double buffer + LDS-resident weights), else two
constexpr int small_min_waves(int spp, int bn) {
  const int steps = (9 * spp + 3) / 4, ct = bn / 16;
  const int lds = 2 * 180 * spp * 16 + (ct * steps > 5 ? ct * steps * 1024 : 0) + 4 * bn * 8;
  // most blocks per CU whose LDS fits; 64-channel tiles keep 64 accumulator + epilogue registers, i.e. at most three waves (168 VGPRs)
  for (int k = (bn >= 64 ? 3 : 4); k > 2; k--)
    if (lds * k <= 160 * 1024) return k;
  return 2;
}
template <int SPP>
__device__ __forceinline__ int patch_slot(int p, int j) { return p * SPP + (j ^ ((p / (16 / SPP)) % SPP)); }

// AFF: source 1 is a BatchNorm-ed producer's RAW output; the staging applies scale / shift + activation (ConvArgs::in_scale) on the way into
// LDS, so the activated tensor never exists in HBM.  A thread's slots all carry the same VE channels (256 % SPP == 0): the coefficients
// come from a block copy in LDS, padding slots stay zero (validity bits travel with each prefetched register set).
// D2S (SPP = 4, BN = 64 only): the forward of an exact-2x up-sampling layer on its SOURCE (ConvArgs::d2s).  The four 16-channel output tiles are
// the four parity classes (a, b) of the 2x2 block of output pixels a source pixel expands to; tap (kh', kw') feeds class (a, b) only if
// kh' - a and kw' - b are 0 or 1 (the class's 2x2 effective kernel, pack mode 2), so 16 of the 36 (tap, class) MFMA pairs remain; every class
// tile is stored at its own output pixel (depth to space), statistics per (class, channel) column.
constexpr bool d2s_used(int c, int s) { return ((s / 3) - (c >> 1) == 0 || (s / 3) - (c >> 1) == 1) && ((s % 3) - (c & 1) == 0 || (s % 3) - (c & 1) == 1); }
constexpr int d2s_slot(int c, int s) {      // rank of (class tile c, tap s) among the 16 used pairs, class-major: only those fragments are kept in LDS
  int n = 0;
  for (int cc = 0; cc < 4; cc++)
    for (int ss = 0; ss < 9; ss++) {
      if (cc == c && ss == s) return n;
      if (d2s_used(cc, ss)) n++;
    }
  return n;
}
constexpr int small_min_waves_d2s() { return (2 * 180 * 4 * 16 + 16 * 1024 + 4 * 64 * 8) * 3 <= 160 * 1024 ? 3 : 2; }
template <typename T, int SPP, int BN, bool W8, bool AFF = false, bool D2S = false>
__global__ __launch_bounds__(256, D2S ? small_min_waves_d2s() : small_min_waves(SPP, BN)) void conv3x3_small_kernel(ConvArgs a, int tilesH, int tilesW) {
  static_assert(!D2S || (SPP == 4 && BN == 64 && !AFF), "D2S: 64-byte pixels, four class tiles");
  constexpr int VE = Elem<T>::VE;
  constexpr int TW = W8 ? 8 : 16, TH = W8 ? 16 : 8;
  constexpr int WT = TW + 2, HT = TH + 2, NP = HT * WT;
  constexpr int NSLOT = NP * SPP;
  constexpr int PIT = (NSLOT + 255) / 256;
  constexpr int CT = BN / 16;
  constexpr int STEPS = (9 * SPP + 3) / 4;  // one step = 4 x 16 bytes of K (one bf16 MFMA / four fp32 MFMAs)
  __shared__ uint4 sP[2][NSLOT];
  __shared__ float red[4 * BN * 2];

  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  const int ntiles = a.N * tilesH * tilesW;
  // XCD x (blocks with blockIdx % 8 == x) owns tiles [x*T8, (x+1)*T8)
  const int T8 = (ntiles + 7) >> 3, G8 = gridDim.x >> 3;
  const int xcd = blockIdx.x & 7;
  const int tend = min(ntiles, (xcd + 1) * T8);
  int tile = xcd * T8 + (blockIdx.x >> 3);

  // weight fragments: row = cout (c*16 + fr), 16-byte K slot = s*4 + fg.  Kept in registers, except for the variants where those 36-40
  // VGPRs cost a wave of occupancy (these kernels are latency bound: 2 -> 3 -> 4 waves per SIMD each measured faster): there every
  // wave reads its fragments from a lane-major LDS copy ([c][s][lane], conflict free).
  constexpr bool WLDS = (CT * STEPS > 5);
  __shared__ uint4 sW[WLDS ? (D2S ? 16 : CT * STEPS) * 64 : 1];
  __shared__ __attribute__((aligned(16))) float sAff[AFF ? 2 * SPP * VE : 1];
  if (AFF) affine_fill(sAff, a.in_scale, a.in_shift, 0, SPP * VE, a.C1, t, 256);
  const bool aff_lane = AFF && (t % SPP) * VE < a.C1;
  uint4 wr[WLDS ? 1 : CT][WLDS ? 1 : STEPS];
  {
    const uint4* wp = reinterpret_cast<const uint4*>(a.w);
    const int kslots = a.Kpad / VE;
#pragma unroll
    for (int c = 0; c < CT; c++)
#pragma unroll
      for (int s = 0; s < STEPS; s++) {
        if (D2S && !d2s_used(c, s)) continue;
        const uint4 v = wp[(int64_t)(c * 16 + fr) * kslots + s * 4 + fg];
        if (WLDS) { if (wv == 0) sW[(D2S ? d2s_slot(c, s) : c * STEPS + s) * 64 + lane] = v; }
        else wr[WLDS ? 0 : c][WLDS ? 0 : s] = v;
      }
  }
  // tile coordinates (image, tile row, tile column) advance by G8 tiles with carries: the runtime divisions of a per-tile decode were
  // ~200 of the ~215 scalar instructions per tile (SQ_INSTS_SALU = SQ_INSTS_VALU in the PMC pass)
  struct TC { int n, th, tw; };
  const TC step = {(G8 / tilesW) / tilesH, (G8 / tilesW) % tilesH, G8 % tilesW};
  auto decode = [&](int tl) RD_INLINE_LAMBDA { TC c; c.tw = tl % tilesW; const int q_ = tl / tilesW; c.th = q_ % tilesH; c.n = q_ / tilesH; return c; };
  auto advance = [&](TC c) RD_INLINE_LAMBDA {
    c.tw += step.tw; if (c.tw >= tilesW) { c.tw -= tilesW; c.th++; }
    c.th += step.th; if (c.th >= tilesH) { c.th -= tilesH; c.n++; }
    if (c.th >= tilesH) { c.th -= tilesH; c.n++; }
    c.n += step.n;
    return c;
  };
  // This thread's patch slots are the same for every tile: pixel (py, px) of the patch, source tensor and channel offset are decoded
  // once; per tile only the bounds test and one 64-bit multiply-add remain (the general gather, conv_src_ptr, re-derived all of it per
  // load -- ~75 VALU instructions each in a VALU-bound kernel).  dil == 1 here (geom3x3); pixel indices fit 32 bits (conv3x3_small_ok).
  int spy[PIT], spx[PIT], sC[PIT]; const T* sbase[PIT]; bool sok[PIT];
#pragma unroll
  for (int i = 0; i < PIT; i++) {
    const int idx = t + 256 * i;
    sok[i] = idx < NSLOT;
    const int pp = idx / SPP, sl = idx - pp * SPP;
    spy[i] = pp / WT - 1; spx[i] = pp - (pp / WT) * WT - 1;
    const int ci = sl * VE;
    if (ci < a.C1) { sbase[i] = (const T*)a.src1 + ci; sC[i] = a.C1; }
    else { sbase[i] = (const T*)a.src2 + (ci - a.C1); sC[i] = a.C2; }
  }
  const int Hp = a.ups ? a.H1 : a.Hin, Wp = a.ups ? a.W1 : a.Win;
  auto load_patch = [&](TC tc, uint4 (&rp)[PIT], unsigned& okm) RD_INLINE_LAMBDA {
    const int oh0 = tc.th * TH, ow0 = tc.tw * TW, nb = tc.n * Hp;
    okm = 0;
#pragma unroll
    for (int i = 0; i < PIT; i++) {
      uint4 v = make_uint4(0, 0, 0, 0);
      const int ih = oh0 + spy[i], iw = ow0 + spx[i];
      if (sok[i] && (unsigned)ih < (unsigned)a.Hin && (unsigned)iw < (unsigned)a.Win) {
        if (AFF) okm |= 1u << i;
        int hs = ih, ws = iw;
        if (a.ups) {  // F.interpolate(mode='nearest') source index, ATen float formula (as conv_src_ptr)
          hs = min((int)floorf((float)ih * a.scale_h), a.H1 - 1);
          ws = min((int)floorf((float)iw * a.scale_w), a.W1 - 1);
        }
        const int pix = (nb + hs) * Wp + ws;
        v = *reinterpret_cast<const uint4*>(sbase[i] + (int64_t)pix * sC[i]);
      }
      rp[i] = v;
    }
  };
  auto store_patch = [&](int buf, const uint4 (&rp)[PIT], unsigned okm) RD_INLINE_LAMBDA {
    float sc[VE], sh[VE];
    if (AFF) {
#pragma unroll
      for (int e = 0; e < VE; e++) { sc[e] = sAff[(t % SPP) * VE + e]; sh[e] = sAff[SPP * VE + (t % SPP) * VE + e]; }
    }
#pragma unroll
    for (int i = 0; i < PIT; i++) {
      int idx = t + 256 * i;
      if (idx < NSLOT) {
        int pp = idx / SPP;
        uint4 v = rp[i];
        if (AFF) { const uint4 z = affine16((const T*)nullptr, v, sc, sh, a.in_act, a.in_slope); if (aff_lane && ((okm >> i) & 1u)) v = z; }
        sP[buf][patch_slot<SPP>(pp, idx - pp * SPP)] = v;
      }
    }
  };
  int ppix[2], lpy[2], lpx[2];
#pragma unroll
  for (int pt = 0; pt < 2; pt++) {
    lpy[pt] = W8 ? (wv * 4 + pt * 2 + (fr >> 3)) : (wv * 2 + pt);
    lpx[pt] = W8 ? (fr & 7) : fr;
    ppix[pt] = lpy[pt] * WT + lpx[pt];
  }

  // one tile: MFMA over the LDS patch in `buf`, then the shared epilogue
  // BatchNorm partial sums of ALL tiles of this block stay in registers; one statistics row per block at the end
  float ssum[CT][4], ssq[CT][4];
#pragma unroll
  for (int c = 0; c < CT; c++)
#pragma unroll
    for (int r = 0; r < 4; r++) { ssum[c][r] = 0.f; ssq[c][r] = 0.f; }
  auto tile_body = [&](int tile, TC tc, int buf) RD_INLINE_LAMBDA {
    f32x4 acc[CT][2];
#pragma unroll
    for (int c = 0; c < CT; c++) { acc[c][0] = f32x4{0, 0, 0, 0}; acc[c][1] = f32x4{0, 0, 0, 0}; }
#pragma unroll
    for (int s = 0; s < STEPS; s++) {
      // K slot g = s*4 + fg  ->  tap g / SPP (clamped: the packed weights are zero past tap 8), slot-in-pixel g % SPP
      int tap, j;
      if (SPP >= 4) { tap = (s * 4) / SPP; j = (s * 4) % SPP + fg; }
      else { tap = min(s * 2 + (fg >> 1), 8); j = fg & 1; }
      const int toff = (tap / 3) * WT + (tap % 3);
      uint4 pf[2];
#pragma unroll
      for (int pt = 0; pt < 2; pt++) pf[pt] = sP[buf][patch_slot<SPP>(ppix[pt] + toff, j)];
#pragma unroll
      for (int c = 0; c < CT; c++) {
        if (D2S && !d2s_used(c, s)) continue;      // structurally zero block
        const uint4 wf = WLDS ? sW[(D2S ? d2s_slot(c, s) : c * STEPS + s) * 64 + lane] : wr[WLDS ? 0 : c][WLDS ? 0 : s];
#pragma unroll
        for (int pt = 0; pt < 2; pt++) {
          if (sizeof(T) == 4) {
            acc[c][pt] = mfma_16x16x4_f32(__uint_as_float(wf.x), __uint_as_float(pf[pt].x), acc[c][pt]);
            acc[c][pt] = mfma_16x16x4_f32(__uint_as_float(wf.y), __uint_as_float(pf[pt].y), acc[c][pt]);
            acc[c][pt] = mfma_16x16x4_f32(__uint_as_float(wf.z), __uint_as_float(pf[pt].z), acc[c][pt]);
            acc[c][pt] = mfma_16x16x4_f32(__uint_as_float(wf.w), __uint_as_float(pf[pt].w), acc[c][pt]);
          } else {
            s16x8 wa, pb;
            __builtin_memcpy(&wa, &wf, 16);
            __builtin_memcpy(&pb, &pf[pt], 16);
            acc[c][pt] = mfma_16x16x32_bf16(wa, pb, acc[c][pt]);
          }
        }
      }
    }
    {
      int64_t mm[2]; bool mvv[2];
#pragma unroll
      for (int pt = 0; pt < 2; pt++) {
        int oh = tc.th * TH + lpy[pt], ow = tc.tw * TW + lpx[pt];
        mvv[pt] = oh < a.OH && ow < a.OW;
        mm[pt] = ((int64_t)tc.n * a.OH + oh) * a.OW + ow;
      }
      if (a.pool2) {
        // Data gradient of a layer that up-samples its input by exactly 2 (nearest): the gradient of a source pixel is the sum of its 2x2
        // block of output pixels.  Tiles start on even rows / columns, so the block lies inside the tile: columns are neighbouring lanes
        // (fr ^ 1); rows are the two pixel tiles of the lane (16-wide tiles: rows 2 wv, 2 wv + 1) or lanes fr ^ 8 (8-wide tiles: a pixel
        // tile holds two rows).  The full-resolution gradient is never written (round 2: 0.19 ms of upsample_nearest_bwd per RC-Net step
        // reading what this kernel had just written, and 4x the store traffic here).  Summed in fp32, rounded once.
#pragma unroll
        for (int c = 0; c < CT; c++) {
          if (!W8) {
#pragma unroll
            for (int r = 0; r < 4; r++) acc[c][0][r] += acc[c][1][r];
          }
#pragma unroll
          for (int pt = 0; pt < (W8 ? 2 : 1); pt++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
              float v = acc[c][pt][r];
              if (W8) v += __shfl_xor(v, 8);
              v += __shfl_xor(v, 1);
              acc[c][pt][r] = v;
            }
        }
        const int OH2 = a.OH >> 1, OW2 = a.OW >> 1;
#pragma unroll
        for (int pt = 0; pt < 2; pt++) {
          const int oh = tc.th * TH + lpy[pt], ow = tc.tw * TW + lpx[pt];
          const bool rep = !(fr & 1) && (W8 ? !(fr & 8) : pt == 0);      // the lane (and pixel tile) that holds the block's sum
          mvv[pt] = mvv[pt] && rep;
          mm[pt] = ((int64_t)tc.n * OH2 + (oh >> 1)) * OW2 + (ow >> 1);
        }
      }
      if (D2S) {
        // class tile c = (a, b) of source pixel (oh, ow) -> output pixel (2 oh + a, 2 ow + b) of the (2 OH) x (2 OW) x D1 tensor
#pragma unroll
        for (int c = 0; c < CT; c++) {
          int64_t mc[2];
#pragma unroll
          for (int pt = 0; pt < 2; pt++) {
            const int oh = tc.th * TH + lpy[pt], ow = tc.tw * TW + lpx[pt];
            mc[pt] = ((int64_t)tc.n * 2 * a.OH + 2 * oh + (c >> 1)) * (2 * a.OW) + 2 * ow + (c & 1);
          }
          conv_epilogue_store_at<T, 1, false>(a, *reinterpret_cast<f32x4 (*)[1][2]>(&acc[c]), mc, mvv, fg * 4, *reinterpret_cast<float (*)[1][4]>(&ssum[c]),
                                              *reinterpret_cast<float (*)[1][4]>(&ssq[c]));
        }
      } else
      conv_epilogue_store<T, CT, false>(a, acc, mm, mvv, 0, 0, fr, fg, ssum, ssq);      // (no addend: these variants sit at their register caps)
    }
  };

  // prefetch distance TWO tiles (two register sets, loop unrolled by two so no set is ever copied): a tile's MFMA + epilogue is far
  // shorter than the HBM latency, with distance one every block stalled on its next patch (ROI-resolution layers ran at 17-30 % of
  // the HBM roof with four blocks per CU)
  uint4 ra[PIT], rb[PIT];
  unsigned ma = 0, mb = 0;      // validity bits of the two register sets (AFF: padding must stay zero behind the affine map)
  int t0 = tile, t1 = tile + G8;
  TC c0 = decode(t0), c1 = advance(c0);
  if (t0 < tend) load_patch(c0, ra, ma);
  if (t1 < tend) load_patch(c1, rb, mb);
  if (AFF) __syncthreads();     // the coefficient copy is complete
  int buf = 0;
  while (t0 < tend) {
    store_patch(buf, ra, ma);
    __syncthreads();
    { const int t2 = t1 + G8; const TC c2 = advance(c1); if (t2 < tend) load_patch(c2, ra, ma); tile_body(t0, c0, buf); t0 = t1; t1 = t2; c0 = c1; c1 = c2; buf ^= 1; }
    if (t0 >= tend) break;
    store_patch(buf, rb, mb);
    __syncthreads();
    { const int t2 = t1 + G8; const TC c2 = advance(c1); if (t2 < tend) load_patch(c2, rb, mb); tile_body(t0, c0, buf); t0 = t1; t1 = t2; c0 = c1; c1 = c2; buf ^= 1; }
  }
  conv_epilogue_stats<CT, BN, 4>(a, ssum, ssq, 0, 0, wv, fr, fg, t, blockIdx.x, red);   // every block writes its row (zeros if it had no tile)
}

// ---- pointwise (1x1, stride 1) layers with a short K axis: K = Cin fits one or two MFMA k-steps, so there is nothing to stage -- a lane
// loads its own pixel fragment (16 bytes of one pixel's channels) straight from HBM, the weight fragments of all output tiles live in
// registers, and the shared epilogue writes the tile.  EfficientNet-Lite3's expansion convolutions (24 -> 144 at 442 K pixels, 32 -> 192,
// 48 -> 288 ...) ran at 0.6 TB/s through the implicit-GEMM kernel (12x their HBM time); reference: SML's MiDaS-small backbone blocks.
template <typename T, int BN, int STEPS>
__global__ __launch_bounds__(256) void conv1x1_direct_kernel(ConvArgs a) {
  constexpr int VE = Elem<T>::VE;
  constexpr int CT = BN / 16;
  __shared__ float red[4 * BN * 2];
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  const int n0 = blockIdx.y * BN;
  const int Cin = a.C1;
  uint4 wr[CT][STEPS];
  {
    const uint4* wp = reinterpret_cast<const uint4*>(a.w);
    const int kslots = a.Kpad / VE;
#pragma unroll
    for (int c = 0; c < CT; c++)
#pragma unroll
      for (int s = 0; s < STEPS; s++) { const uint4 v = wp[(int64_t)(n0 + c * 16 + fr) * kslots + s * 4 + fg]; wr[c][s] = v; }
  }
  // PERSISTENT over 128-pixel tiles (blockIdx.x, + gridDim.x, ...): the weight fragments are loaded once, the BatchNorm partial sums of all
  // the block's tiles stay in registers and go through ONE block reduction at the end, and the next tile's pixel fragments are requested
  // before the current tile is multiplied and stored.  As one tile per block the kernel spent ~775 vector + ~200 LDS instructions per wave
  // (16 MFMAs) almost entirely in the per-tile statistics reduction and ran at 1.0-1.1 TB/s on 0.2-GB outputs (PMC, round 2).
  const int ntiles = (int)cdiv(a.M, 128);
  float ssum[CT][4], ssq[CT][4];
#pragma unroll
  for (int c = 0; c < CT; c++)
#pragma unroll
    for (int r = 0; r < 4; r++) { ssum[c][r] = 0.f; ssq[c][r] = 0.f; }
  auto fetch = [&](int tile, uint4 (&pf)[2][STEPS]) RD_INLINE_LAMBDA {
    const int64_t m0 = (int64_t)tile * 128 + wv * 32;
#pragma unroll
    for (int pt = 0; pt < 2; pt++) {
      const int64_t m = m0 + pt * 16 + fr;
      const int64_t mc = m < a.M ? m : a.M - 1;
#pragma unroll
      for (int s = 0; s < STEPS; s++) {
        const int ci = (s * 4 + fg) * VE;
        const bool ok = m < a.M && ci < Cin;
        // unconditional load from a clamped address, zeroed afterwards (the k axis is padded to the MFMA depth with zeros)
        const uint4 v = *reinterpret_cast<const uint4*>((const T*)a.src1 + mc * Cin + (ci < Cin ? ci : 0));
        pf[pt][s] = ok ? v : make_uint4(0, 0, 0, 0);
      }
    }
  };
  uint4 pa[2][STEPS], pb[2][STEPS];
  int tile = blockIdx.x;
  if (tile < ntiles) fetch(tile, pa);
  while (tile < ntiles) {
    const int nxt = tile + gridDim.x;
    if (nxt < ntiles) fetch(nxt, pb);
    int64_t mm[2]; bool mvv[2];
#pragma unroll
    for (int pt = 0; pt < 2; pt++) { mm[pt] = (int64_t)tile * 128 + wv * 32 + pt * 16 + fr; mvv[pt] = mm[pt] < a.M; }
    f32x4 acc[CT][2];
#pragma unroll
    for (int c = 0; c < CT; c++) { acc[c][0] = f32x4{0, 0, 0, 0}; acc[c][1] = f32x4{0, 0, 0, 0}; }
#pragma unroll
    for (int s = 0; s < STEPS; s++)
#pragma unroll
      for (int c = 0; c < CT; c++)
#pragma unroll
        for (int pt = 0; pt < 2; pt++) {
          if (sizeof(T) == 4) {
            acc[c][pt] = mfma_16x16x4_f32(__uint_as_float(wr[c][s].x), __uint_as_float(pa[pt][s].x), acc[c][pt]);
            acc[c][pt] = mfma_16x16x4_f32(__uint_as_float(wr[c][s].y), __uint_as_float(pa[pt][s].y), acc[c][pt]);
            acc[c][pt] = mfma_16x16x4_f32(__uint_as_float(wr[c][s].z), __uint_as_float(pa[pt][s].z), acc[c][pt]);
            acc[c][pt] = mfma_16x16x4_f32(__uint_as_float(wr[c][s].w), __uint_as_float(pa[pt][s].w), acc[c][pt]);
          } else {
            s16x8 wa, pbv;
            __builtin_memcpy(&wa, &wr[c][s], 16);
            __builtin_memcpy(&pbv, &pa[pt][s], 16);
            acc[c][pt] = mfma_16x16x32_bf16(wa, pbv, acc[c][pt]);
          }
        }
    conv_epilogue_store<T, CT>(a, acc, mm, mvv, n0, 0, fr, fg, ssum, ssq);
    tile = nxt;
#pragma unroll
    for (int pt = 0; pt < 2; pt++)
#pragma unroll
      for (int s = 0; s < STEPS; s++) pa[pt][s] = pb[pt][s];
  }
  conv_epilogue_stats<CT, BN, 4>(a, ssum, ssq, n0, 0, wv, fr, fg, t, blockIdx.x, red);
}

// ---- single input channel (data gradient of a Cout = 1 head): direct form, one pixel per thread ---------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void conv3x3_c1_kernel(ConvArgs a) {
  constexpr int VE = Elem<T>::VE;
  __shared__ float sw[9 * 32];  // [tap][cout]
  for (int i = threadIdx.x; i < 9 * a.Cout; i += 256) {
    int tap = i / a.Cout, co = i - tap * a.Cout;
    sw[i] = Elem<T>::ld((const T*)a.w + (int64_t)co * a.Kpad + tap);
  }
  __syncthreads();
  const T* x = (const T*)a.src1;
  const bool vec = (a.Cout % VE) == 0;
  for (int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x; m < a.M; m += (int64_t)gridDim.x * 256) {
    int ow = (int)(m % a.OW); int64_t q = m / a.OW; int oh = (int)(q % a.OH); int n = (int)(q / a.OH);
    float xv[9];
#pragma unroll
    for (int kh = 0; kh < 3; kh++)
#pragma unroll
      for (int kw = 0; kw < 3; kw++) {
        int ih = oh - 1 + kh, iw = ow - 1 + kw;
        bool ok = (unsigned)ih < (unsigned)a.Hin && (unsigned)iw < (unsigned)a.Win;
        // unconditional load from a clamped address, zero selected afterwards: nine requests in flight instead of nine guarded round trips
        const float xl = Elem<T>::ld(x + ((int64_t)n * a.Hin + min(max(ih, 0), a.Hin - 1)) * a.Win + min(max(iw, 0), a.Win - 1));
        xv[kh * 3 + kw] = ok ? xl : 0.f;
      }
    T* d = (T*)a.dst1 + m * a.Cout;
    for (int c0 = 0; c0 < a.Cout; c0 += 4) {
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; r++) {
        int co = c0 + r;
        float s = 0.f;
        if (co < a.Cout) {
#pragma unroll
          for (int tp = 0; tp < 9; tp++) s += sw[tp * a.Cout + co] * xv[tp];   // ascending-k order, as the MFMA path's oracle
          if (a.bias) s += a.bias[co];
          s = act_fwd(s, a.act, a.slope);
        }
        v[r] = s;
      }
      if (vec || c0 + 3 < a.Cout) { if ((a.Cout & 3) == 0) st4(d + c0, v); else { for (int r = 0; r < 4; r++) Elem<T>::st(d + c0 + r, v[r]); } }
      else for (int r = 0; r < 4 && c0 + r < a.Cout; r++) Elem<T>::st(d + c0 + r, v[r]);
    }
  }
}

// ---- host side -------------------------------------------------------------------------------------------------------------------
static int pick_bn3(int cout) { return cout <= 16 ? 16 : (cout <= 32 ? 32 : (cout <= 64 ? 64 : 128)); }

bool conv3x3_ok(const ConvArgs& a, int dtype) {
  const int Cin = a.C1 + a.C2;
  const int ve = dtype == 0 ? 4 : 8, cke = dtype == 0 ? 32 : 64;
  if ((int64_t)a.N * a.Hin * a.Win >= (int64_t)1 << 31) return false;   // the kernel keeps source pixel indices in 32 bits
  return a.KH == 3 && a.KW == 3 && a.stride == 1 && a.pad == 1 && a.dil == 1 && a.OH == a.Hin && a.OW == a.Win && (Cin % cke == 0) &&
         (a.C1 % ve == 0);
}
static bool use_w8(const ConvArgs& a) {  // tile shape with the smaller padded area
  if (rd_opt_is_set(OPT_CONV3X3_W8)) return rd_opt(OPT_CONV3X3_W8, 0) != 0;   // experiment hook (rd_set_option)
  int64_t a16 = cdiv(a.OH, 8) * 8 * cdiv(a.OW, 16) * 16, a8 = cdiv(a.OH, 16) * 16 * cdiv(a.OW, 8) * 8;
  return a8 < a16;
}
int conv3x3_tiles(const ConvArgs& a) {
  bool w8 = use_w8(a);
  return a.N * (int)cdiv(a.OH, w8 ? 16 : 8) * (int)cdiv(a.OW, w8 ? 8 : 16);
}

template <typename T>
static void launch3_t(const ConvArgs& a, hipStream_t st) {
  const bool w8 = use_w8(a);
  const int tilesH = (int)cdiv(a.OH, w8 ? 16 : 8), tilesW = (int)cdiv(a.OW, w8 ? 8 : 16);
  // at most 64 output channels per block by default: the 128-channel tile (32 KB of double-buffered weights + the 23 KB patch) leaves two
  // blocks per CU, the 64-channel one four -- measured 901 -> 908 img/s on the RC-Net step although every patch is staged twice as often
  // (32-channel tiles: 884).  RD_PATCH_BN_MAX=128 restores the wide tile (A/B).
  const int bn_max = rd_opt(OPT_PATCH_BN_MAX, 64);
  const int bn = std::min(bn_max, pick_bn3(a.Cout));
  dim3 grid((unsigned)(a.N * tilesH * tilesW), (unsigned)cdiv(a.Cout, bn));
#define RD_C3(BNV)                                                                                                   \
  if (bn == BNV) {                                                                                                   \
    if (w8) hipLaunchKernelGGL((conv3x3_patch_kernel<T, BNV, true>), grid, dim3(256), 0, st, a, tilesH, tilesW);     \
    else hipLaunchKernelGGL((conv3x3_patch_kernel<T, BNV, false>), grid, dim3(256), 0, st, a, tilesH, tilesW);       \
  }
  RD_C3(16) RD_C3(32) RD_C3(64) RD_C3(128)
#undef RD_C3
}
void launch_conv3x3(const ConvArgs& a, int dtype, hipStream_t st) {
  if (dtype == 0) launch3_t<float>(a, st);
  else launch3_t<bf16_t>(a, st);
}

static bool geom3x3(const ConvArgs& a) {
  return a.KH == 3 && a.KW == 3 && a.stride == 1 && a.pad == 1 && a.dil == 1 && a.OH == a.Hin && a.OW == a.Win;
}
// narrow layers: Cin bytes per pixel in {32, 64} with Cout <= 64, or 128 bytes with Cout <= 16
bool conv3x3_small_ok(const ConvArgs& a, int dtype) {
  const int Cin = a.C1 + a.C2, es = dtype == 0 ? 4 : 2, ve = 16 / es;
  const int cb = Cin * es;
  if (!geom3x3(a) || (cb != 32 && cb != 64 && cb != 128) || (a.C1 % ve) != 0) return false;
  if ((int64_t)a.N * a.Hin * a.Win >= (int64_t)1 << 31) return false;   // the kernel keeps pixel indices in 32 bits
  return cb == 128 ? a.Cout <= 16 : a.Cout <= 64;
}
int conv3x3_small_blocks(const ConvArgs& a, int dtype) {   // persistent blocks = BatchNorm statistics rows of this path
  const bool w8 = use_w8(a);
  const int64_t ntiles = (int64_t)a.N * cdiv(a.OH, w8 ? 16 : 8) * cdiv(a.OW, w8 ? 8 : 16);
  // (test hook, rd_set_option "conv3x3_g8": few persistent blocks -> several tiles per block on small cases)
  // persistent grid = resident capacity: 8 XCDs x 32 CUs x (4 or 2 blocks per CU, see small_min_waves)
  const int cb_slots = ((a.C1 + a.C2) * (dtype == 0 ? 4 : 2)) / 16;
  const int per_cu = a.d2s ? small_min_waves_d2s() : small_min_waves(cb_slots, pick_bn3(a.Cout));
  return 8 * (int)std::min<int64_t>(cdiv(ntiles, 8), rd_opt(OPT_CONV3X3_G8, 32 * per_cu));
}
template <typename T>
static void launch_small_t(const ConvArgs& a, hipStream_t st) {
  const bool w8 = use_w8(a);
  const int tilesH = (int)cdiv(a.OH, w8 ? 16 : 8), tilesW = (int)cdiv(a.OW, w8 ? 8 : 16);
  const int ntiles = a.N * tilesH * tilesW;
  const int spp = (a.C1 + a.C2) * (int)sizeof(T) / 16, bn = pick_bn3(a.Cout);
  dim3 grid((unsigned)conv3x3_small_blocks(a, (int)sizeof(T) == 4 ? 0 : 1));
  const bool aff = a.in_scale != nullptr;
  if (a.d2s) {      // conv_d2s_ok: 64-byte pixels, 4 x 16 output channels
    if (sizeof(T) == 2) {
      if (w8) hipLaunchKernelGGL((conv3x3_small_kernel<bf16_t, 4, 64, true, false, true>), grid, dim3(256), 0, st, a, tilesH, tilesW);
      else hipLaunchKernelGGL((conv3x3_small_kernel<bf16_t, 4, 64, false, false, true>), grid, dim3(256), 0, st, a, tilesH, tilesW);
    }
    return;
  }
#define RD_S3(SPPV, BNV)                                                                                                  \
  if (spp == SPPV && bn == BNV) {                                                                                         \
    if (aff) {                                                                                                            \
      if (w8) hipLaunchKernelGGL((conv3x3_small_kernel<T, SPPV, BNV, true, true>), grid, dim3(256), 0, st, a, tilesH, tilesW);   \
      else hipLaunchKernelGGL((conv3x3_small_kernel<T, SPPV, BNV, false, true>), grid, dim3(256), 0, st, a, tilesH, tilesW);     \
    } else if (w8) hipLaunchKernelGGL((conv3x3_small_kernel<T, SPPV, BNV, true>), grid, dim3(256), 0, st, a, tilesH, tilesW);    \
    else hipLaunchKernelGGL((conv3x3_small_kernel<T, SPPV, BNV, false>), grid, dim3(256), 0, st, a, tilesH, tilesW);      \
  }
  RD_S3(2, 16) RD_S3(4, 16) RD_S3(8, 16) RD_S3(2, 32) RD_S3(4, 32) RD_S3(2, 64) RD_S3(4, 64)
#undef RD_S3
}
// instantiation names as rocprofv3 prints them (bench.py groups launches by kernel)
const char* conv3x3_small_name(const ConvArgs& a, int dtype) {
  static thread_local char buf[96];
  const int es = dtype == 0 ? 4 : 2;
  if (a.d2s) snprintf(buf, sizeof(buf), "conv3x3_small_kernel<%s, 4, 64, %s, false, true>", RD_T16_NAME, use_w8(a) ? "true" : "false");
  else
  snprintf(buf, sizeof(buf), "conv3x3_small_kernel<%s, %d, %d, %s, %s>", dtype == 0 ? "float" : RD_T16_NAME, (a.C1 + a.C2) * es / 16, pick_bn3(a.Cout),
           use_w8(a) ? "true" : "false", a.in_scale ? "true" : "false");
  return buf;
}
const char* conv3x3_patch_name(const ConvArgs& a, int dtype) {
  static thread_local char buf[96];
  const int bn_max = rd_opt(OPT_PATCH_BN_MAX, 64);
  snprintf(buf, sizeof(buf), "conv3x3_patch_kernel<%s, %d, %s>", dtype == 0 ? "float" : RD_T16_NAME, std::min(bn_max, pick_bn3(a.Cout)),
           use_w8(a) ? "true" : "false");
  return buf;
}
void launch_conv3x3_small(const ConvArgs& a, int dtype, hipStream_t st) {
  if (dtype == 0) launch_small_t<float>(a, st);
  else launch_small_t<bf16_t>(a, st);
}

static int conv1x1_min_m() { return rd_opt(OPT_CONV1X1_MIN_M, 8192); }   // test hook: 0 forces the kernel
// 1x1 / stride 1 / single source, K within two k-steps (Cin * sizeof <= 128 bytes), Cin a multiple of the 16-byte vector
bool conv1x1_direct_ok(const ConvArgs& a, int dtype) {
  const int es = dtype == 0 ? 4 : 2, ve = 16 / es;
  return a.KH == 1 && a.KW == 1 && a.stride == 1 && a.pad == 0 && a.dil == 1 && a.C2 == 0 && !a.ups && a.OH == a.Hin && a.OW == a.Win &&
         (a.C1 % ve) == 0 && a.C1 * es <= 128 && a.M >= conv1x1_min_m();
}
int conv1x1_direct_rows(const ConvArgs& a) { return (int)std::min<int64_t>(cdiv(a.M, 128), 1024); }   // persistent blocks = statistics rows
void launch_conv1x1_direct(const ConvArgs& a, int dtype, hipStream_t st) {
  // at most 64 output channels per block: the persistent form keeps 2 x 4 x BN/16 statistics registers next to the accumulators and the
  // weight fragments (BN = 128: 232-304 VGPRs, one or two waves per SIMD); the input re-read per channel block is the small operand here
  const int bn = std::min(64, pick_bn3(a.Cout)), es = dtype == 0 ? 4 : 2;
  const int steps = a.C1 * es <= 64 ? 1 : 2;
  dim3 grid((unsigned)conv1x1_direct_rows(a), (unsigned)cdiv(a.Cout, bn));
#define RD_P1(TT, BNV, SV) hipLaunchKernelGGL((conv1x1_direct_kernel<TT, BNV, SV>), grid, dim3(256), 0, st, a)
#define RD_P1B(TT, SV) { if (bn == 16) RD_P1(TT, 16, SV); else if (bn == 32) RD_P1(TT, 32, SV); else RD_P1(TT, 64, SV); }
  if (dtype == 0) { if (steps == 1) RD_P1B(float, 1) else RD_P1B(float, 2) }
  else { if (steps == 1) RD_P1B(bf16_t, 1) else RD_P1B(bf16_t, 2) }
#undef RD_P1B
#undef RD_P1
}

// ---- layers with a handful of channels and millions of pixels (SML: the 3 -> 3 `first` convolution, the 32 -> 1 head and its data gradient,
// the data gradient of the 3 -> 32 stride-2 stem):
// on the MFMA kernels they fill 2-4 % of a tile and run 10-15x over their HBM time.  One output pixel per thread, weights in LDS, the
// taps fetched with unconditional clamped loads, BatchNorm partial sums in registers with one block reduction at the end (rows = blocks).
template <typename T, int KH, int CIN, int CO, int DIL>
__global__ __launch_bounds__(256) void conv_few_kernel(ConvArgs a) {
  constexpr int K = KH * KH * CIN, VE = Elem<T>::VE;
  __shared__ float sw[K * CO];      // [k][cout]
  __shared__ float red[4][CO * 2];
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  for (int i = t; i < K * CO; i += 256) {
    const int k = i / CO, co = i - k * CO;
    sw[i] = co < a.Cout ? Elem<T>::ld((const T*)a.w + (int64_t)co * a.Kpad + k) : 0.f;
  }
  __syncthreads();
  const T* x = (const T*)a.src1;
  float s1[CO], s2[CO];
#pragma unroll
  for (int c = 0; c < CO; c++) { s1[c] = 0.f; s2[c] = 0.f; }
  float bv[CO];
#pragma unroll
  for (int c = 0; c < CO; c++) { const float b = a.bias ? a.bias[c < a.Cout ? c : 0] : 0.f; bv[c] = c < a.Cout ? b : 0.f; }
  const bool vec = (a.Cout & 3) == 0;
  // DIL = 2 (data gradient of a stride-2 layer): a pixel only has the taps whose row / column parity matches its own -- 1, 2, 2 or 4 of
  // the nine.  With an even row length the 64 lanes of a wave take pixels of ONE column parity (waves 2k / 2k+1 of a block share a run of
  // 128 pixels: even / odd columns), so a tap is valid for the whole wave or for none of it and the invalid ones are skipped by a ballot:
  // 2.25 instead of 9 taps of 128 weight reads and FMAs per pixel (the LDS weight reads bound the kernel: 0.17 ms for a 58 MB layer).
  const bool split = DIL > 1 && (a.OW & 1) == 0;
  const int64_t Mr = split ? ((a.M + 127) & ~(int64_t)127) : a.M;
  for (int64_t mm = (int64_t)blockIdx.x * 256 + t; mm < Mr; mm += (int64_t)gridDim.x * 256) {
    const int64_t m = split ? ((mm & ~(int64_t)127) | ((mm & 63) << 1) | ((mm >> 6) & 1)) : mm;
    const bool live = m < a.M;
    const int ow = (int)(m % a.OW); const int64_t q = m / a.OW; const int oh = (int)(q % a.OH); const int n = live ? (int)(q / a.OH) : 0;
    float acc[CO];
#pragma unroll
    for (int c = 0; c < CO; c++) acc[c] = 0.f;
#ifndef RD_EMU
    // the weights are loop invariant: with many of them (288 x 4 for the stem's data gradient) the compiler hoists every LDS read out of
    // the pixel loop into registers and spills ~900 of them (measured: 11 ms instead of ~0.05); a compiler-level memory barrier keeps the
    // reads inside the loop
    if constexpr (K * CO > 256) asm volatile("" ::: "memory");
#endif
    // taps in ascending k order; every tap's loads are unconditional (clamped address), a tap that does not exist contributes zeros.
    // DIL = 2: the input is the stride-2 layer's output gradient, read where (oh - pad + kh, ow - pad + kw) is even (data gradient).
    // With many weights the tap loops stay rolled (one tap's 128 LDS values at a time), otherwise the scheduler front-loads all of them.
    constexpr int UNR = (K * CO > 256) ? 1 : KH;
#pragma unroll UNR
    for (int kh = 0; kh < KH; kh++)
#pragma unroll UNR
      for (int kw = 0; kw < KH; kw++) {
        int ih = oh - a.pad + kh, iw = ow - a.pad + kw;
        bool ok = ih >= 0 && iw >= 0;
        if (DIL > 1) { ok = ok && live && (((ih | iw) & (DIL - 1)) == 0); ih >>= 1; iw >>= 1; }      // DIL is 1 or 2
        ok = ok && ih < a.Hin && iw < a.Win;
        if (DIL > 1) { if (__ballot(ok) == 0ull) continue; }      // no lane of the wave has this tap
        const T* px = x + (((int64_t)n * a.Hin + min(max(ih, 0), a.Hin - 1)) * a.Win + min(max(iw, 0), a.Win - 1)) * CIN;
        float xv[CIN];
        if constexpr (CIN % VE == 0) {
#pragma unroll
          for (int c = 0; c < CIN; c += VE) {
            float v[VE];
            ldv(px + c, v);
#pragma unroll
            for (int e = 0; e < VE; e++) xv[c + e] = ok ? v[e] : 0.f;
          }
        } else {
#pragma unroll
          for (int c = 0; c < CIN; c++) { const float v = Elem<T>::ld(px + c); xv[c] = ok ? v : 0.f; }
        }
#pragma unroll
        for (int c = 0; c < CIN; c++)
#pragma unroll
          for (int co = 0; co < CO; co++) acc[co] = fmaf(sw[((kh * KH + kw) * CIN + c) * CO + co], xv[c], acc[co]);
      }
    if (!live) continue;
    T* d = (T*)a.dst1 + m * a.Cout;
#pragma unroll
    for (int c0 = 0; c0 < CO; c0 += 4) {
      float v[4], vr[4];
#pragma unroll
      for (int r = 0; r < 4; r++) v[r] = act_fwd(acc[c0 + r] + bv[c0 + r], a.act, a.slope);
      if (vec && c0 + 3 < a.Cout) round_store4(d + c0, v, vr);
      else {
#pragma unroll
        for (int r = 0; r < 4; r++) { vr[r] = Elem<T>::rnd(v[r]); if (c0 + r < a.Cout) Elem<T>::st(d + c0 + r, vr[r]); }
      }
#pragma unroll
      for (int r = 0; r < 4; r++) { s1[c0 + r] += vr[r]; s2[c0 + r] += vr[r] * vr[r]; }
    }
  }
  if (!a.stats) return;
#pragma unroll
  for (int c = 0; c < CO; c++) {
    const float u1 = wave_sum(s1[c]), u2 = wave_sum(s2[c]);
    if (lane == 0) { red[wv][c * 2] = u1; red[wv][c * 2 + 1] = u2; }
  }
  __syncthreads();
  for (int i = t; i < a.Cout * 2; i += 256)
    a.stats[(int64_t)blockIdx.x * a.Cout * 2 + i] = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
}
// ---- 1 -> Cout pointwise expansion (the data gradient of a Cout = 1 head over a 1x1 kernel: SML's 32 -> 1 output convolution on 2.65 M pixels):
// Cout / VE threads per pixel, each with its VE weights in registers and ONE 16-byte store, so that a wave's store instruction covers 1 KiB
// of consecutive addresses.  (In conv_few_kernel a thread owned a pixel and wrote its 64 bytes as eight 8-byte stores, 64 bytes apart from
// its neighbour's: 106 us for 170 MB.)  Same product and sum per element: results are those of conv_few_kernel bit for bit.
template <typename T>
__global__ __launch_bounds__(256) void pointwise_expand_kernel(ConvArgs a) {
  constexpr int VE = Elem<T>::VE;
  const int G = a.Cout / VE;
  const int64_t total = (int64_t)a.M * G, stride = (int64_t)gridDim.x * 256;
  int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int co0 = (int)(g % G) * VE;      // loop invariant: the grid stride is a multiple of G (256 % G == 0)
  float w[VE], bv[VE];
#pragma unroll
  for (int e = 0; e < VE; e++) { w[e] = Elem<T>::ld((const T*)a.w + (int64_t)(co0 + e) * a.Kpad); bv[e] = a.bias ? a.bias[co0 + e] : 0.f; }
  const T* x = (const T*)a.src1;
  for (; g < total; g += stride) {
    const int64_t m = g / G;
    const float xv = Elem<T>::ld(x + m);
    float v[VE];
#pragma unroll
    for (int e = 0; e < VE; e++) v[e] = act_fwd(fmaf(w[e], xv, 0.f) + bv[e], a.act, a.slope);
    stv((T*)a.dst1 + m * a.Cout + co0, v);
  }
}

static int conv_few_min_m() { return rd_opt(OPT_CONV_FEW_MIN_M, 1 << 16); }   // test hook: 0 forces the kernel
bool conv_few_ok(const ConvArgs& a) {
  if (a.KH != a.KW || a.stride != 1 || a.ups || a.C2 || a.D1 != a.Cout || a.M < conv_few_min_m()) return false;
  if (a.dil == 2)      // data gradient of a stride-2 layer with 3 input channels (the 3 -> 32 stem): 32-channel dy, 3 outputs
    return a.KH == 3 && a.C1 == 32 && a.Cout <= 4;
  if (a.dil != 1 || a.OH != a.Hin || a.OW != a.Win || a.pad != a.KH / 2) return false;
  return (a.KH == 3 && a.C1 == 3 && a.Cout <= 4) || (a.KH == 1 && a.C1 == 1 && a.Cout == 32) || (a.KH == 1 && a.C1 == 32 && a.Cout <= 4);
}
int conv_few_blocks(const ConvArgs& a) { return (int)std::min<int64_t>(cdiv(a.M, 256 * 4), 2048); }
void launch_conv_few(const ConvArgs& a, int dtype, hipStream_t st) {
  if (a.KH == 1 && a.C1 == 1 && a.dil == 1 && !a.stats && !a.add1 && a.Cout % (dtype == 0 ? 4 : 8) == 0 && 256 % (a.Cout / (dtype == 0 ? 4 : 8)) == 0) {
    const int G = a.Cout / (dtype == 0 ? 4 : 8);
    const dim3 ge((unsigned)std::min<int64_t>(cdiv((int64_t)a.M * G, 256 * 4), 4096));
    if (dtype == 0) hipLaunchKernelGGL((pointwise_expand_kernel<float>), ge, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((pointwise_expand_kernel<bf16_t>), ge, dim3(256), 0, st, a);
    return;
  }
  const dim3 grid((unsigned)conv_few_blocks(a));
#define RD_FEW(TT) { if (a.dil == 2) hipLaunchKernelGGL((conv_few_kernel<TT, 3, 32, 4, 2>), grid, dim3(256), 0, st, a);        \
                     else if (a.KH == 3) hipLaunchKernelGGL((conv_few_kernel<TT, 3, 3, 4, 1>), grid, dim3(256), 0, st, a);       \
                     else if (a.C1 == 1) hipLaunchKernelGGL((conv_few_kernel<TT, 1, 1, 32, 1>), grid, dim3(256), 0, st, a);      \
                     else hipLaunchKernelGGL((conv_few_kernel<TT, 1, 32, 4, 1>), grid, dim3(256), 0, st, a); }
  if (dtype == 0) RD_FEW(float) else RD_FEW(bf16_t)
#undef RD_FEW
}

bool conv3x3_c1_ok(const ConvArgs& a) {
  return geom3x3(a) && a.C1 == 1 && a.C2 == 0 && !a.ups && !a.stats && a.D1 == a.Cout && a.Cout <= 32;
}
void launch_conv3x3_c1(const ConvArgs& a, int dtype, hipStream_t st) {
  unsigned grid = (unsigned)std::min<int64_t>(cdiv(a.M, 256), 256 * 32);
  if (dtype == 0) hipLaunchKernelGGL((conv3x3_c1_kernel<float>), dim3(grid), dim3(256), 0, st, a);
  else hipLaunchKernelGGL((conv3x3_c1_kernel<bf16_t>), dim3(grid), dim3(256), 0, st, a);
}

}  // namespace rd
