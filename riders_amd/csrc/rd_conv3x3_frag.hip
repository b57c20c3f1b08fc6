// Register-fed 3x3 / stride-1 convolution (forward and data gradient) for gfx950: the wide layers of RC-Net (Cout > 16 with whole 128-byte
// channel chunks; reference utils/net_utils.py:84-91,195-198,564-569) and of the SML scratch decoder (modules/midas/blocks.py:99-174).
//
// Why a second patch kernel.  conv3x3_patch_kernel (rd_conv3x3.hip) shares the weights of ONE tap at a time through LDS: every tap costs a
// global -> VGPR -> LDS -> VGPR round trip behind a block barrier (9 barriers per 128-byte channel chunk, 16 MFMAs per wave between two of
// them), its wave tile of 32 pixels x 64 channels needs 6 ds_read_b128 per 8 MFMAs, and every fragment address is XOR-swizzle arithmetic.
// PMC (round 2): a third of the wave cycles parked at those barriers, another third stalled on issue, 26 % of the dense bf16 MFMA peak; the
// first version of this file (weights from L2, XOR-swizzled patch) turned out VALU-bound instead: 5-8 vector instructions per MFMA, 45 % of
// its LDS cycles bank conflicts on tiles that wrap around image rows (profiles/r03_pmc_frag_v1.txt).  Now:
//   * only the PIXEL patch lives in LDS (staged once per chunk, nine taps walk it);
//   * the weights never touch LDS: they are packed in MFMA FRAGMENT ORDER ([chunk][tap][16-channel tile][k half][lane] x 16 bytes, written
//     next to the row-major operand by the pack kernels) so a wave fetches a fragment with ONE fully coalesced 1-KiB load, a tap ahead of
//     its use, straight from L2 into the registers the MFMA reads -- no barrier inside a chunk;
//   * a wave owns 32 output channels x (64 or 128) pixels: one ds_read_b128 feeds two MFMAs;
//   * the patch is stored as FOUR PLANES of [pixel][32 bytes] (plane h = 16-byte channel slots 2h, 2h+1; plane stride = 32 mod 256 bytes):
//     a fragment read of 16 consecutive patch pixels is bank-conflict free WITHOUT a swizzle (b128 lane groups mix k-groups g and g+1 of
//     all 16 pixels: even / odd 16-byte columns of one plane, or planes h / h+1 an even number of columns apart), the staging stores of a
//     pixel's eight slots are too, and every fragment address of the tap loop is `base register + immediate`;
//   * narrow maps are tiled LINEARLY: the whole tensor is one strip of virtual pixels, rows of OW + 1 (one shared zero column) and one
//     shared zero row between images; a tile is TP consecutive virtual pixels, its patch TP + 2 (OW + 1) + 2 consecutive ones, the tap
//     offsets kh (OW + 1) + kw.  Virtual zero pixels compute and are not stored: 3-20 % of the MFMAs on the 60x25 ... 15x6 RoI maps against
//     27-30 % with 2-D tiles, and tiles cross image borders.  Wide maps keep 2-D tiles (TP/16 rows x 16 columns);
//   * every patch load is unconditional (clamped address, zero selected afterwards) and the next chunk's patch is requested piecewise
//     BEHIND the fragment loads of the taps (a wave's loads return in order).
#include "rd_conv_common.h"
#include <stdio.h>

namespace rd {

struct FragGeom {
  int lin;        // 1: linear tiles of the virtual pixel strip; 0: 2-D tiles (TP/16 rows x 16 columns of one image)
  int WT;         // patch row stride in pixels (2-D: 18, linear: OW + 1)
  int tilesH, tilesW, ntiles, ncb;
  int np;         // patch pixels
  int ps;         // plane stride in bytes (>= 32 np, = 32 mod 256)
  int ctall;      // 16-channel tiles in the packed operand (rows_pad / 16)
  int wfrag;      // element offset of the fragment-ordered copy inside the packed operand (rows_pad * Kpad)
  float rWT, rH1; // 1 / WT, 1 / (OH + 1): the strip is decoded with float reciprocals + one correction step (host: strip < 2^22 pixels)
  int aff;        // byte offset of the [2][Cin] coefficient copy of the consumer-side BatchNorm apply inside the dynamic LDS
  int db;         // byte offset of the SECOND patch buffer (0: one buffer).  Round 5, multi-chunk layers: chunk c is staged into buffer c & 1 as
                  // soon as a wave has finished chunk c - 1's taps, so a chunk costs ONE block barrier (store -> barrier -> taps) instead of two
                  // (barrier -> store -> barrier -> taps): buffer c & 1 was last read during chunk c - 2, which every wave left before it
                  // passed chunk c - 1's barrier
};

constexpr int frag_pmax(int tp) { return tp <= 128 ? 256 : (tp <= 256 ? 384 : 640); }   // patch pixels a block may stage (32 / 48 / 80 KB)

// NPT 16-pixel tiles per wave, WPX x WCH waves (pixels x 32-channel groups): block tile = (16 NPT WPX) pixels x (32 WCH) channels.
// LIN: linear tiles; MULTI: more than one channel chunk (the next chunk's patch is prefetched).
// AFF: source 1 is a BatchNorm-ed producer's RAW output; scale / shift + activation are applied while the patch is written to LDS
// (ConvArgs::in_scale; coefficients in an LDS copy behind the planes at byte offset g.aff), padding pixels stay zero.
// D2S: the forward of an exact-2x up-sampling layer on its SOURCE (ConvArgs::d2s, pack mode 2; see conv3x3_small_kernel): the wave's 32 output
// channels belong to ONE parity class (a, b) (D1 % 32 == 0), so the wave runs only that class's four taps (a + {0, 1}, b + {0, 1}) per chunk -- their
// fragments alone are fetched -- and stores its tile at output pixel (2 i + a, 2 j + b), channels relative to the class.
// S2D: the data gradient of the same layer on the source (ConvArgs::s2d, pack mode 3): src1 is the layer's output gradient at (2 Hin) x (2 Win) x Cr,
// staged as Hin x Win x 4 Cr = (parity class, channel) channels (space to depth: a 16-byte slot's class picks one of the pixel's 2x2 block); the
// 32-channel k-halves whose classes do not use a tap are skipped (class (a, b) uses the flipped taps kr in {1 - a, 2 - a}, kc in {1 - b, 2 - b}).
template <typename T, int NPT, int WPX, int WCH, bool LIN, bool MULTI, bool AFF = false, bool D2S = false, bool S2D = false>
__global__ __launch_bounds__(64 * WPX * WCH, (WPX * WCH == 8 && WCH == 4) ? 4 : ((WPX * WCH == 8 || NPT == 8 || (MULTI && WCH == 1)) ? 2 : 3)) void conv3x3_frag_kernel(ConvArgs a, FragGeom g) {
  static_assert(!D2S || !AFF, "D2S has no consumer-side BatchNorm apply");
  static_assert(!S2D || (!AFF && !D2S && sizeof(T) == 2), "S2D: 16-bit activations, plain source");
  constexpr int NW = WPX * WCH, NT = 64 * NW;
  constexpr int VE = Elem<T>::VE;
  constexpr int CKE = STAGE_BYTES / (int)sizeof(T);   // channels per chunk (128 bytes per pixel)
  constexpr int TP = 16 * NPT * WPX, BN = 32 * WCH;
  constexpr int PMAX = frag_pmax(TP);
  constexpr int PIT = (PMAX * 8 + NT - 1) / NT;
  RD_DYN_SMEM(smem);

  const int t = threadIdx.x, lane = t & 63, wv = RD_WAVE_UNIFORM(t >> 6);
  const int fr = lane & 15, fg = lane >> 4;
  const int wpx = wv % WPX, wc = wv / WPX;
  const int idx = xcd_contiguous((int)blockIdx.x, (int)gridDim.x);   // an XCD walks neighbouring tiles; the channel blocks of a tile sit together
  const int cb = idx % g.ncb, tile = idx / g.ncb;
  const int n0 = cb * BN;
  const int Cin = a.C1 + a.C2;
  const int nchunk = Cin / CKE;
  const int WT = g.WT, np = g.np, PS = g.ps;
  const int Hp = a.ups ? a.H1 : a.Hin, Wp = a.ups ? a.W1 : a.Win;
  const int H1 = a.OH + 1;
  int ca = 0, cb2 = 0, cls0 = 0;      // D2S: the wave's parity class (row, column) and its first channel
  if (D2S) { const int cls = RD_WAVE_UNIFORM((n0 + wc * 32) / a.D1); ca = cls >> 1; cb2 = cls & 1; cls0 = cls * a.D1; }

  // ---- tile origin ------------------------------------------------------------------------------------------------------------------
  // linear: first virtual pixel of the tile (the strip starts with a zero row, which no tile computes); 2-D: image, first row, first column
  int u0 = 0, tn = 0, toh0 = 0, tow0 = 0;
  if (LIN) u0 = WT + tile * TP;
  else {
    const int tw_ = tile % g.tilesW, q_ = tile / g.tilesW, th_ = q_ % g.tilesH;
    tn = q_ / g.tilesH; toh0 = th_ * (TP / 16); tow0 = tw_ * 16;
  }
  // virtual pixel -> (image, row, column) of the tensor; false for a zero row / zero column / a pixel outside the strip
  auto strip_pixel = [&](int u, int& n, int& ih, int& iw) RD_INLINE_LAMBDA {
    int c, hh;
    const int vrow = fdiv_small(max(u, 0), WT, g.rWT, c);
    n = fdiv_small(vrow, H1, g.rH1, hh);
    ih = hh - 1; iw = c - 1;
    return u >= 0 && c >= 1 && hh >= 1 && n < a.N;
  };

  // ---- patch staging role: slot id = t + NT i -> patch pixel id >> 3, 16-byte channel slot t & 7.  The source pixel of every slot is
  // decoded once (-1 = zero); a chunk only picks the source tensor for its channel offset.
  int spix[PIT];
#pragma unroll
  for (int i = 0; i < PIT; i++) {
    const int pp = (t + NT * i) >> 3;
    int n, ih, iw; bool ok;
    if (LIN) ok = strip_pixel(u0 - WT - 1 + pp, n, ih, iw);
    else {
      const int py = pp / 18, px = pp - py * 18;
      n = tn; ih = toh0 - 1 + py; iw = tow0 - 1 + px;
      ok = (unsigned)ih < (unsigned)a.Hin && (unsigned)iw < (unsigned)a.Win;
    }
    int pix = -1;
    if (ok && pp < np) {
      int hs = ih, ws = iw;
      if (a.ups) {  // F.interpolate(mode='nearest') source index, ATen float formula (as conv_src_ptr)
        hs = min((int)floorf((float)ih * a.scale_h), a.H1 - 1);
        ws = min((int)floorf((float)iw * a.scale_w), a.W1 - 1);
      }
      pix = S2D ? (n * 2 * Hp + 2 * hs) * (2 * Wp) + 2 * ws : (n * Hp + hs) * Wp + ws;      // S2D: pixel (2 i, 2 j) of the (2 H) x (2 W) tensor
    }
    spix[i] = pix;
  }
  uint4 rp[PIT];
  // one 16-byte piece of a chunk's patch: unconditional load from a clamped address, zero selected afterwards (a guarded load is one memory
  // round trip per load).  `last`: there is no next chunk -- every lane re-reads pixel 0 (one cache line per load, nobody waits for it), so
  // the prefetch stays branch-free and the compiler keeps COUNTING the loads in flight instead of waiting vmcnt(0) at the next join.
  auto load_piece = [&](int chunk, int i, bool last) RD_INLINE_LAMBDA {
    const int ci = chunk * CKE + (t & 7) * VE;
    const bool first = ci < a.C1;
    const T* cbp = first ? (const T*)a.src1 + ci : (const T*)a.src2 + (ci - a.C1);
    const int cs = first ? a.C1 : a.C2;
    const int px = last ? 0 : max(spix[i], 0);
    uint4 v;
    if (S2D) {      // channel slot ci = (class, channel): the class selects the pixel of the 2x2 block, the tensor has cr = C1 / 4 channels
      const int cr = a.C1 >> 2, cls = ci / cr, co = ci - cls * cr;
      v = *reinterpret_cast<const uint4*>((const T*)a.src1 + (last ? 0 : (int64_t)(px + (cls >> 1) * 2 * Wp + (cls & 1)) * cr + co));
    } else
    v = *reinterpret_cast<const uint4*>((last ? (const T*)a.src1 : cbp) + (int64_t)px * cs);
    rp[i] = spix[i] < 0 ? make_uint4(0, 0, 0, 0) : v;
  };
  const int st_base = ((t & 7) >> 1) * PS + (t & 1) * 16 + (t >> 3) * 32;    // plane (slot >> 1), 16-byte column (slot & 1), pixel t >> 3
  const float* const aff = reinterpret_cast<const float*>(smem + g.aff);
  if (AFF) affine_fill(const_cast<float*>(aff), a.in_scale, a.in_shift, 0, Cin, a.C1, t, NT);      // visible after the first chunk's barrier
  auto store_patch = [&](int chunk, int boff) RD_INLINE_LAMBDA {
    float sc[VE], sh[VE];
    const int ci = chunk * CKE + (t & 7) * VE;
    if (AFF) {
#pragma unroll
      for (int e = 0; e < VE; e++) { sc[e] = aff[ci + e]; sh[e] = aff[Cin + ci + e]; }
    }
#pragma unroll
    for (int i = 0; i < PIT; i++)
      if (((t + NT * i) >> 3) < np) {
        uint4 v = rp[i];
        if (AFF) { const uint4 z = affine16((const T*)nullptr, v, sc, sh, a.in_act, a.in_slope); if (ci < a.C1 && spix[i] >= 0) v = z; }
        *reinterpret_cast<uint4*>(smem + boff + st_base + i * (NT / 8) * 32) = v;
      }
  };

  // ---- this lane's output pixels (one per MFMA pixel tile): flattened output index or -1 -------------------------------------------------
  int pm[NPT];
#pragma unroll
  for (int pt = 0; pt < NPT; pt++) {
    const int q = wpx * NPT + pt;
    if (LIN) {
      int n, ih, iw;
      const bool ok = strip_pixel(u0 + q * 16 + fr, n, ih, iw);
      pm[pt] = ok ? (D2S ? (n * 2 * a.OH + 2 * ih + ca) * (2 * a.OW) + 2 * iw + cb2 : (n * a.OH + ih) * a.OW + iw) : -1;
    } else {
      const int oh = toh0 + q, ow = tow0 + fr;
      pm[pt] = (oh < a.OH && ow < a.OW) ? (D2S ? (tn * 2 * a.OH + 2 * oh + ca) * (2 * a.OW) + 2 * ow + cb2 : (tn * a.OH + oh) * a.OW + ow) : -1;
    }
  }
  // fragment read of (pixel tile pt, tap (kr, kc), k half kh): plane kh*2 + (fg >> 1), column fg & 1, patch pixel p0 + fr + tap offset where
  // p0 = 16 (wpx NPT + pt) [linear] or 18 (wpx NPT + pt) [2-D]: one base register per (kernel row, k half) [linear: the row stride is a
  // run-time value] or per k half [2-D], everything else an immediate offset.
  const int lb = (fg >> 1) * PS + (fg & 1) * 16 + (wpx * NPT * (LIN ? 16 : 18) + fr) * 32;
  int lbase[LIN ? 3 : 1][2];
#pragma unroll
  for (int kr = 0; kr < (LIN ? 3 : 1); kr++)
#pragma unroll
    for (int kh = 0; kh < 2; kh++) lbase[kr][kh] = lb + kh * 2 * PS + kr * WT * 32;

  // ---- weight fragments: [chunk][tap][16-channel tile][k half][lane], this wave's two channel tiles ------------------------------------
  const uint4* const wfr = reinterpret_cast<const uint4*>((const T*)a.w + g.wfrag) + ((n0 >> 4) + wc * 2) * 128 + lane;
  const int wstep = g.ctall * 128;               // uint4 per (chunk, tap)
  auto load_w = [&](int ct, uint4 (&w)[2][2]) RD_INLINE_LAMBDA {   // ct = chunk * 9 + tap
    const uint4* p = wfr + (int64_t)ct * wstep;
#pragma unroll
    for (int c = 0; c < 2; c++)
#pragma unroll
      for (int kh = 0; kh < 2; kh++) { const uint4 v = p[(c * 2 + kh) * 64]; w[c][kh] = v; }
  };

  f32x4 acc[2][NPT];
#pragma unroll
  for (int c = 0; c < 2; c++)
#pragma unroll
    for (int pt = 0; pt < NPT; pt++) acc[c][pt] = f32x4{0, 0, 0, 0};

  // One tap: the NEXT tap's fragments are requested first (unconditionally: the last tap of the layer re-reads its own), then -- in layers
  // with several chunks -- a couple of 16-byte pieces of the NEXT chunk's patch, then this tap's 2 x NPT fragment reads and 4 x NPT MFMAs.
  // A wave's loads return IN ORDER, so waiting for tap t+1's fragments also waits for every piece requested before them: the pieces go
  // BEHIND the fragment loads of their tap and get two taps of MFMA work to arrive (all of them in front of tap 0 would park every wave
  // for one HBM round trip per chunk).  Scheduling fences keep each tap's requests at its top and its reads inside it.
  constexpr int PPT = (PIT + 7) / 8;   // pieces per tap, taps 0..7
  constexpr int PPT4 = (PIT + 3) / 4;  // D2S: four taps per chunk
  int boff = 0;      // byte offset of the patch buffer the current chunk reads
  // tap: this tap (a compile-time constant at the nine call sites of the plain kernel, a wave-uniform value in D2S); ct_next: the fragment set
  // (chunk * 9 + tap) to request for the next tap_body; pieces [p0, p1) of chunk_next's patch ride behind it
  // S2D: does k-half kh (32 channels from chunk * 64 + 32 kh) of this chunk hold a class that uses flipped tap (kr, kc)?  (block uniform)
  auto s2d_used = [&](int chunk, int kh, int kr, int kc) RD_INLINE_LAMBDA {
    const int cr = a.C1 >> 2, k0 = chunk * CKE + kh * (CKE / 2), c0 = k0 / cr, c1 = (k0 + CKE / 2 - 1) / cr;
    bool u = false;
    for (int c = c0; c <= c1; c++) {
      const int ca_ = c >> 1, cb_ = c & 1;
      u = u || ((kr == 1 - ca_ || kr == 2 - ca_) && (kc == 1 - cb_ || kc == 2 - cb_));
    }
    return u;
  };
  auto tap_body = [&](int tap, int ct_next, int chunk_next, int p0, int p1, uint4 (&wcur)[2][2], uint4 (&wnxt)[2][2]) RD_INLINE_LAMBDA {
    load_w(ct_next, wnxt);
    if (MULTI) {
      const bool last = chunk_next >= nchunk;
#pragma unroll
      for (int k = 0; k < (D2S ? PPT4 : PPT); k++)
        if (p0 + k < p1 && p0 + k < PIT) load_piece(min(chunk_next, nchunk - 1), p0 + k, last);
    }
    sched_fence();
    const int kr = tap / 3, kc = tap - kr * 3;
    const int toff = D2S ? (kr * (LIN ? WT : 18) + kc) * 32 : 0;      // (run-time tap: one scalar offset instead of immediates)
#pragma unroll
    for (int kh = 0; kh < 2; kh++) {
      if (S2D && !s2d_used(chunk_next - 1, kh, kr, kc)) continue;      // structurally zero block of the K axis
      uint4 pf[NPT];
#pragma unroll
      for (int pt = 0; pt < NPT; pt++)
        pf[pt] = D2S ? *reinterpret_cast<const uint4*>(smem + boff + lb + kh * 2 * PS + toff + (LIN ? pt * 16 : pt * 18) * 32)
                     : *reinterpret_cast<const uint4*>(smem + boff + lbase[LIN ? kr : 0][kh] + (LIN ? pt * 16 + kc : (pt + kr) * 18 + kc) * 32);
#pragma unroll
      for (int c = 0; c < 2; c++) {
        const uint4 wf = wcur[c][kh];
#pragma unroll
        for (int pt = 0; pt < NPT; pt++) {
          if (sizeof(T) == 4) {
            acc[c][pt] = mfma_16x16x4_f32(__uint_as_float(wf.x), __uint_as_float(pf[pt].x), acc[c][pt]);
            acc[c][pt] = mfma_16x16x4_f32(__uint_as_float(wf.y), __uint_as_float(pf[pt].y), acc[c][pt]);
            acc[c][pt] = mfma_16x16x4_f32(__uint_as_float(wf.z), __uint_as_float(pf[pt].z), acc[c][pt]);
            acc[c][pt] = mfma_16x16x4_f32(__uint_as_float(wf.w), __uint_as_float(pf[pt].w), acc[c][pt]);
          } else {
            s16x8 wv8, pb;
            __builtin_memcpy(&wv8, &wf, 16);
            __builtin_memcpy(&pb, &pf[pt], 16);
            acc[c][pt] = mfma_16x16x32_bf16(wv8, pb, acc[c][pt]);
          }
        }
      }
    }
    sched_fence();
  };
  // plain kernel: tap t requests tap t + 1's fragments (the layer's last tap re-reads its own) and, for t < 8, its share of the next chunk's patch
#define RD_TAP(chunk, t, wc_, wn_) tap_body(t, min((chunk) * 9 + (t) + 1, nchunk * 9 - 1), (chunk) + 1, (t) * PPT, (t) < 8 ? (t) * PPT + PPT : (t) * PPT, wc_, wn_)
  auto d2s_tap = [&](int i) RD_INLINE_LAMBDA { return (ca + (i >> 1)) * 3 + cb2 + (i & 1); };      // the class's i-th tap

  uint4 wa[2][2], wb[2][2];
#pragma unroll
  for (int i = 0; i < PIT; i++) load_piece(0, i, false);
  load_w(D2S ? d2s_tap(0) : 0, wa);
  for (int chunk = 0; chunk < nchunk; chunk++) {
    boff = (MULTI && (chunk & 1)) ? g.db : 0;
    if (!(MULTI && g.db)) __syncthreads();            // one buffer: every wave is done with the previous chunk's patch
    store_patch(chunk, boff);
    __syncthreads();
    if (D2S) {      // four taps: an even number of register swaps, the next chunk's first fragments land in wa
      const int cn = min(chunk + 1, nchunk - 1);
      tap_body(d2s_tap(0), chunk * 9 + d2s_tap(1), chunk + 1, 0, PPT4, wa, wb);
      tap_body(d2s_tap(1), chunk * 9 + d2s_tap(2), chunk + 1, PPT4, 2 * PPT4, wb, wa);
      tap_body(d2s_tap(2), chunk * 9 + d2s_tap(3), chunk + 1, 2 * PPT4, 3 * PPT4, wa, wb);
      tap_body(d2s_tap(3), cn * 9 + d2s_tap(0), chunk + 1, 3 * PPT4, 4 * PPT4, wb, wa);
      continue;
    }
    RD_TAP(chunk, 0, wa, wb); RD_TAP(chunk, 1, wb, wa); RD_TAP(chunk, 2, wa, wb);
    RD_TAP(chunk, 3, wb, wa); RD_TAP(chunk, 4, wa, wb); RD_TAP(chunk, 5, wb, wa);
    RD_TAP(chunk, 6, wa, wb); RD_TAP(chunk, 7, wb, wa); RD_TAP(chunk, 8, wa, wb);
    if (MULTI) {     // nine taps = an odd number of register swaps: the next chunk's first fragments landed in wb
#pragma unroll
      for (int c = 0; c < 2; c++)
#pragma unroll
        for (int kh = 0; kh < 2; kh++) wa[c][kh] = wb[c][kh];
    }
  }
#undef RD_TAP

  __syncthreads();  // all waves finished reading the patch before it is reused as reduction scratch
  float ssum[2][4], ssq[2][4];
#pragma unroll
  for (int c = 0; c < 2; c++)
#pragma unroll
    for (int r = 0; r < 4; r++) { ssum[c][r] = 0.f; ssq[c][r] = 0.f; }
#pragma unroll
  for (int pp = 0; pp < NPT / 2; pp++) {   // the store routine takes two pixel tiles at a time
    int64_t mm[2]; bool mvv[2]; f32x4 a2[2][2];
#pragma unroll
    for (int h = 0; h < 2; h++) {
      mm[h] = pm[pp * 2 + h]; mvv[h] = pm[pp * 2 + h] >= 0;
#pragma unroll
      for (int c = 0; c < 2; c++) a2[c][h] = acc[c][pp * 2 + h];
    }
    if (D2S) conv_epilogue_store_at<T, 2, false, 16, false>(a, a2, mm, mvv, n0 + wc * 32 - cls0 + fg * 4, ssum, ssq);      // channels relative to the class
    else conv_epilogue_store<T, 2, true, true>(a, a2, mm, mvv, n0, wc, fr, fg, ssum, ssq);
  }
  conv_epilogue_stats<2, BN, WPX, NT>(a, ssum, ssq, n0, wc, wpx, fr, fg, t, tile, reinterpret_cast<float*>(smem));
}


// ---- the same kernel on v_mfma_f32_32x32x16 (round 5) -------------------------------------------------------------------------------
// One MFMA = 32 channels x 32 pixels x 16 input channels: twice the FLOPs of the 16x16x32 form at the same operand bytes per lane, half
// the matrix instructions per FLOP, and a higher ceiling (2.38 against 2.08 PFLOP/s measured, MI355X_MICROARCH.md).  What changes:
//   * a pixel fragment is 32 CONSECUTIVE patch pixels x one 16-byte channel slot per lane half, so the patch is stored as EIGHT planes of
//     [pixel][16 bytes] (plane = 16-byte channel slot): the 16 lanes a ds_read_b128 serves together (MI355X_MICROARCH.md, LDS table) then
//     cover 16 distinct 16-byte columns of one 512-byte run -- conflict free with no swizzle; the plane stride is a compile-time constant
//     = 16 (mod 128) bytes, which also spreads the eight slots of a staged pixel (one ds_write_b128 lane group) over all 32 banks, and every
//     fragment address is `one of three base registers + immediate`;
//   * the weight fragment of (32-channel tile, 16-wide k-step) is read out of the EXISTING fragment-ordered operand copy
//     ([chunk][tap][16-row tile][k half][lane] x 16 B) through a per-lane offset: lanes 0-15 / 16-31 take the tile's two 16-row halves,
//     lanes 32-63 the next 8 input channels -- four contiguous 256-byte runs per wave load, no second packing;
//   * a wave owns CW 32-channel tiles x NQ 32-pixel tiles (CW = 2: one pixel-fragment read feeds two MFMAs).
// Linear tiles only (the RoI-resolution decoder layers, where this kernel's time is); bf16 / fp16 builds only.
constexpr int frag32_ps(int pmax) { return 16 * pmax + 16; }      // plane stride in bytes: >= 16 pixels' worth, = 16 (mod 128)

template <typename T, int NQ, int WPX, int WCH, int CW, bool MULTI>
__global__ __launch_bounds__(64 * WPX * WCH, (WPX * WCH == 8) ? (CW * NQ <= 2 ? 4 : 2) : (CW * NQ <= 2 ? 3 : (CW * NQ <= 4 ? 2 : 1)))
void conv3x3_frag32_kernel(ConvArgs a, FragGeom g) {
  static_assert(sizeof(T) == 2, "16-bit activations only");
  constexpr int NW = WPX * WCH, NT = 64 * NW;
  constexpr int VE = 8, CKE = 64;                       // elements per 16-byte slot, channels per chunk
  constexpr int TP = 32 * NQ * WPX, BN = 32 * CW * WCH;
  constexpr int PMAX = frag_pmax(TP);
  constexpr int PS = frag32_ps(PMAX);
  constexpr int PIT = (PMAX * 8 + NT - 1) / NT;
  RD_DYN_SMEM(smem);

  const int t = threadIdx.x, lane = t & 63, wv = RD_WAVE_UNIFORM(t >> 6);
  const int lp = lane & 31, lh = lane >> 5;
  const int wpx = wv % WPX, wc = wv / WPX;
  const int idx = xcd_contiguous((int)blockIdx.x, (int)gridDim.x);
  const int cb = idx % g.ncb, tile = idx / g.ncb;
  const int n0 = cb * BN;
  const int Cin = a.C1 + a.C2;
  const int nchunk = Cin / CKE;
  const int WT = g.WT, np = g.np;
  const int Hp = a.ups ? a.H1 : a.Hin, Wp = a.ups ? a.W1 : a.Win;
  const int H1 = a.OH + 1;
  const int u0 = WT + tile * TP;
  auto strip_pixel = [&](int u, int& n, int& ih, int& iw) RD_INLINE_LAMBDA {
    int c, hh;
    const int vrow = fdiv_small(max(u, 0), WT, g.rWT, c);
    n = fdiv_small(vrow, H1, g.rH1, hh);
    ih = hh - 1; iw = c - 1;
    return u >= 0 && c >= 1 && hh >= 1 && n < a.N;
  };

  // ---- patch staging: slot id = t + NT i -> patch pixel id >> 3, 16-byte channel slot t & 7 (as conv3x3_frag_kernel) ----------------------
  int spix[PIT];
#pragma unroll
  for (int i = 0; i < PIT; i++) {
    const int pp = (t + NT * i) >> 3;
    int n, ih, iw;
    const bool ok = strip_pixel(u0 - WT - 1 + pp, n, ih, iw);
    int pix = -1;
    if (ok && pp < np) {
      int hs = ih, ws = iw;
      if (a.ups) {
        hs = min((int)floorf((float)ih * a.scale_h), a.H1 - 1);
        ws = min((int)floorf((float)iw * a.scale_w), a.W1 - 1);
      }
      pix = (n * Hp + hs) * Wp + ws;
    }
    spix[i] = pix;
  }
  uint4 rp[PIT];
  auto load_piece = [&](int chunk, int i, bool last) RD_INLINE_LAMBDA {
    const int ci = chunk * CKE + (t & 7) * VE;
    const bool first = ci < a.C1;
    const T* cbp = first ? (const T*)a.src1 + ci : (const T*)a.src2 + (ci - a.C1);
    const int cs = first ? a.C1 : a.C2;
    const int px = last ? 0 : max(spix[i], 0);
    const uint4 v = *reinterpret_cast<const uint4*>((last ? (const T*)a.src1 : cbp) + (int64_t)px * cs);
    rp[i] = spix[i] < 0 ? make_uint4(0, 0, 0, 0) : v;
  };
  const int st_base = (t & 7) * PS + (t >> 3) * 16;      // plane = slot, pixel t >> 3
  auto store_patch = [&]() RD_INLINE_LAMBDA {
#pragma unroll
    for (int i = 0; i < PIT; i++)
      if (((t + NT * i) >> 3) < np) *reinterpret_cast<uint4*>(smem + st_base + i * (NT / 8) * 16) = rp[i];
  };

  // ---- this lane's output pixels (one per 32-pixel MFMA tile; both lane halves hold the same pixel, different channels) ------------------
  int pm[NQ];
#pragma unroll
  for (int q = 0; q < NQ; q++) {
    int n, ih, iw;
    const bool ok = strip_pixel(u0 + (wpx * NQ + q) * 32 + lp, n, ih, iw);
    pm[q] = ok ? (n * a.OH + ih) * a.OW + iw : -1;
  }
  // fragment read of (pixel tile q, tap (kr, kc), k-step s): plane 2 s + lh, patch pixel 32 (wpx NQ + q) + lp + kr WT + kc
  int lbase[3];
#pragma unroll
  for (int kr = 0; kr < 3; kr++) lbase[kr] = lh * PS + (wpx * NQ * 32 + lp + kr * WT) * 16;

  // ---- weight fragments out of the 16x16x32-ordered copy: [chunk][tap][16-row tile][k half][lane] x 16 B --------------------------------
  // (32-row tile T, k-step s), lane l: 16-row tile 2 T + (lp >> 4), k half s >> 1, old lane ((s & 1) 2 + lh) 16 + (lp & 15)
  const uint4* const wfr = reinterpret_cast<const uint4*>((const T*)a.w + g.wfrag) + ((n0 >> 4) + wc * CW * 2 + (lp >> 4)) * 128 + lh * 16 + (lp & 15);
  const int wstep = g.ctall * 128;               // uint4 per (chunk, tap)
  auto load_w = [&](int ct, uint4 (&w)[CW][4]) RD_INLINE_LAMBDA {   // ct = chunk * 9 + tap
    const uint4* p = wfr + (int64_t)ct * wstep;
#pragma unroll
    for (int c = 0; c < CW; c++)
#pragma unroll
      for (int s_ = 0; s_ < 4; s_++) { const uint4 v = p[c * 256 + (s_ >> 1) * 64 + (s_ & 1) * 32]; w[c][s_] = v; }
  };

  f32x16 acc[CW][NQ];
#pragma unroll
  for (int c = 0; c < CW; c++)
#pragma unroll
    for (int q = 0; q < NQ; q++)
#pragma unroll
      for (int v = 0; v < 16; v++) acc[c][q][v] = 0.f;

  constexpr int PPT = (PIT + 7) / 8;   // next-chunk patch pieces per tap, taps 0..7
  auto tap_body = [&](int chunk, int tap, uint4 (&wcur)[CW][4], uint4 (&wnxt)[CW][4]) RD_INLINE_LAMBDA {
    load_w(min(chunk * 9 + tap + 1, nchunk * 9 - 1), wnxt);
    if (MULTI && tap < 8) {
      const bool last = chunk + 1 >= nchunk;
#pragma unroll
      for (int k = 0; k < PPT; k++)
        if (tap * PPT + k < PIT) load_piece(min(chunk + 1, nchunk - 1), tap * PPT + k, last);
    }
    sched_fence();
    const int kr = tap / 3, kc = tap % 3;
#pragma unroll
    for (int s_ = 0; s_ < 4; s_++) {
      uint4 pf[NQ];
#pragma unroll
      for (int q = 0; q < NQ; q++) pf[q] = *reinterpret_cast<const uint4*>(smem + lbase[kr] + s_ * 2 * PS + (q * 32 + kc) * 16);
#pragma unroll
      for (int c = 0; c < CW; c++) {
        s16x8 wv8;
        __builtin_memcpy(&wv8, &wcur[c][s_], 16);
#pragma unroll
        for (int q = 0; q < NQ; q++) {
          s16x8 pb;
          __builtin_memcpy(&pb, &pf[q], 16);
          acc[c][q] = mfma_32x32x16_bf16(wv8, pb, acc[c][q]);
        }
      }
    }
    sched_fence();
  };

  uint4 wa[CW][4], wb[CW][4];
#pragma unroll
  for (int i = 0; i < PIT; i++) load_piece(0, i, false);
  load_w(0, wa);
  for (int chunk = 0; chunk < nchunk; chunk++) {
    __syncthreads();
    store_patch();
    __syncthreads();
    tap_body(chunk, 0, wa, wb); tap_body(chunk, 1, wb, wa); tap_body(chunk, 2, wa, wb);
    tap_body(chunk, 3, wb, wa); tap_body(chunk, 4, wa, wb); tap_body(chunk, 5, wb, wa);
    tap_body(chunk, 6, wa, wb); tap_body(chunk, 7, wb, wa); tap_body(chunk, 8, wa, wb);
    if (MULTI) {
#pragma unroll
      for (int c = 0; c < CW; c++)
#pragma unroll
        for (int s_ = 0; s_ < 4; s_++) wa[c][s_] = wb[c][s_];
    }
  }

  __syncthreads();  // all waves finished reading the patch before it is reused as reduction scratch
  // ---- epilogue: register v of a 32x32 tile = channel (v & 3) + 8 (v >> 2) + 4 lh of pixel lp: four groups of 4 consecutive channels, 8 apart --
  float* const red = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int c = 0; c < CW; c++) {
    float ssum[4][4], ssq[4][4];
#pragma unroll
    for (int gq = 0; gq < 4; gq++)
#pragma unroll
      for (int r = 0; r < 4; r++) { ssum[gq][r] = 0.f; ssq[gq][r] = 0.f; }
    const int cbase = n0 + (wc * CW + c) * 32 + lh * 4;
#pragma unroll
    for (int q0 = 0; q0 < NQ; q0 += 2) {   // the store routine takes two pixel tiles at a time
      int64_t mm[2]; bool mvv[2]; f32x4 a2[4][2];
#pragma unroll
      for (int h = 0; h < 2; h++) {
        const int q = q0 + h < NQ ? q0 + h : q0;
        mm[h] = pm[q]; mvv[h] = (q0 + h < NQ) && pm[q] >= 0;
#pragma unroll
        for (int gq = 0; gq < 4; gq++)
#pragma unroll
          for (int r = 0; r < 4; r++) a2[gq][h][r] = acc[c][q][gq * 4 + r];
      }
      conv_epilogue_store_at<T, 4, true, 8>(a, a2, mm, mvv, cbase, ssum, ssq);
    }
    if (a.stats) {      // (sum, sum^2) over the wave's pixels: 32 pixel lanes per half, the half picks the channels
#pragma unroll
      for (int gq = 0; gq < 4; gq++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const float s1 = row32_sum(ssum[gq][r]), s2 = row32_sum(ssq[gq][r]);
          if (lp == 31) {
            const int col = (wc * CW + c) * 32 + gq * 8 + lh * 4 + r;
            red[(wpx * BN + col) * 2 + 0] = s1;
            red[(wpx * BN + col) * 2 + 1] = s2;
          }
        }
    }
  }
  if (a.stats) {
    __syncthreads();
    for (int col = t; col < BN; col += NT) {
      const int co = n0 + col;
      if (co < a.Cout) {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int w = 0; w < WPX; w++) { s1 += red[(w * BN + col) * 2]; s2 += red[(w * BN + col) * 2 + 1]; }
        a.stats[((int64_t)tile * a.Cout + co) * 2 + 0] = s1;
        a.stats[((int64_t)tile * a.Cout + co) * 2 + 1] = s2;
      }
    }
  }
}

// ---- host side -------------------------------------------------------------------------------------------------------------------
// block shapes (pixels x channels, waves): the table is the A/B surface of tools/bench_conv.py (RD_FRAG_V128 / _V64 / _V32)
struct FragVariant { int tp, bn, nw; };
static const FragVariant kFragVariants[] = {
  {128, 128, 4},   // 0: waves 1 x 4, 128 pixels each
  {128, 128, 8},   // 1: waves 2 x 4,  64 pixels each
  {128,  64, 4},   // 2: waves 2 x 2,  64 pixels each
  {256,  64, 4},   // 3: waves 2 x 2, 128 pixels each
  {256,  64, 8},   // 4: waves 4 x 2,  64 pixels each
  {256,  32, 4},   // 5: waves 4 x 1,  64 pixels each
  {512,  32, 4},   // 6: waves 4 x 1, 128 pixels each
};
struct FragPlan { int variant, tp, bn, nw, lin, WT, tilesH, tilesW, ntiles, ncb, np, ps, v32; };
// 32x32x16 variants (conv3x3_frag32_kernel): pixels x channels per block, waves, (NQ, WPX, WCH, CW)
struct Frag32Variant { int tp, bn, nw, nq, wpx, wch, cw; };
static const Frag32Variant kFrag32Variants[] = {
  {0, 0, 0, 0, 0, 0, 0},
  {128, 128, 8, 2, 2, 4, 1},   // 1: eight waves of 64 pixels x 32 channels (the 16x16x32 kernel's tile)
  {128, 128, 4, 4, 1, 4, 1},   // 2: four waves of 128 x 32
  {128,  64, 4, 2, 2, 2, 1},   // 3: 64-channel blocks, four waves of 64 x 32
};
// Measured on MI355X (profiles/r05_microbench/frag32_variants.txt, RC-Net's > 64-channel decoder shapes, tools/bench_conv.py): variant 1 -- the
// same tile, only the MFMA shape changed -- is 5 % SLOWER than the 16x16x32 kernel (0.914-0.932 vs 0.866-0.871 ms over the twelve shapes),
// variant 2 equal within noise (0.886-0.896), wave tiles of 64 x 64 and 128 x 64 (two MFMAs per pixel-fragment read; removed again) 10-20 %
// slower: with two or four accumulators per wave the 64-cycle latency of a 32x32x16 MFMA is exposed between dependent instructions, and
// the kernel was not bound by matrix-instruction issue in the first place.  The default stays the 16x16x32 kernel; option frag32_v128 /
// frag32_v64 (rd_set_option) selects these variants for A/B.

// A/B hooks of tools/bench_conv.py and the tests (rd_set_option, never the environment).  Variant indices are validated against the block width they may be used for
// (a 64-channel block on a <= 32-channel operand would read weight-fragment tiles past the packed rows), everything else is a plain switch.
static int frag_variant_opt(int id, int dflt, int bn_want) {
  const int v = rd_opt(id, dflt);
  return (v >= 0 && v < 7 && kFragVariants[v].bn == bn_want) ? v : dflt;
}

static bool frag_plan(const ConvArgs& a, int dtype, FragPlan& p) {
  const int Cin = a.C1 + a.C2;
  const int ve = dtype == 0 ? 4 : 8, cke = dtype == 0 ? 32 : 64;
  if (!(a.KH == 3 && a.KW == 3 && a.stride == 1 && a.pad == 1 && a.dil == 1 && a.OH == a.Hin && a.OW == a.Win)) return false;
  if ((Cin % cke) || (a.C1 % ve) || a.Cout <= 16) return false;
  if ((int64_t)a.N * a.Hin * a.Win >= (int64_t)1 << 31) return false;   // the kernel keeps source pixel indices in 32 bits
  // > 64 channels: eight-wave blocks (4 waves per SIMD at 126 VGPRs) when there are several chunks -- 0.847 -> 0.803 ms over the
  // seventeen RC-Net shapes of tools/bench_conv.py -- four 128-pixel waves for one-chunk layers (64 -> 128 data gradient: 0.066 vs 0.069)
  const bool multi = Cin > cke;
  p.variant = a.Cout > 64 ? frag_variant_opt(OPT_FRAG_V128, multi ? 1 : 0, 128) : (a.Cout > 32 ? frag_variant_opt(OPT_FRAG_V64, 2, 64) : frag_variant_opt(OPT_FRAG_V32, 5, 32));
  // fewer 128 x 128 blocks than CUs (the deep encoder stages: 76 and 20 tiles): 64-channel blocks double the grid
  if (a.Cout > 64 && !rd_opt_is_set(OPT_FRAG_V128) && rd_opt(OPT_FRAG_SPLIT, 1) &&
      cdiv((int64_t)a.M, 128) * cdiv(a.Cout, 128) < rd_opt(OPT_FRAG_SPLIT_BLOCKS, 256)) p.variant = 2;
  const FragVariant& v = kFragVariants[p.variant];
  p.tp = v.tp; p.bn = v.bn; p.nw = v.nw;
  // the 32x32x16 form: 16-bit builds, linear tiles, no consumer-side BatchNorm apply (those shapes keep the 16x16x32 kernel)
  p.v32 = 0;
  if (dtype != 0 && !a.in_scale) {
    int want = a.Cout > 64 ? rd_opt(OPT_FRAG32_V128, 0) : (a.Cout > 32 ? rd_opt(OPT_FRAG32_V64, 0) : 0);
    if (want >= 1 && want <= 3 && kFrag32Variants[want].bn == (a.Cout > 64 ? 128 : 64)) {
      const Frag32Variant& w32 = kFrag32Variants[want];
      const int64_t strip32 = ((int64_t)a.N * (a.OH + 1) + 1) * (a.OW + 1);
      if (strip32 < ((int64_t)1 << 22) && w32.tp + 2 * (a.OW + 1) + 2 <= frag_pmax(w32.tp) && rd_opt(OPT_FRAG_LIN, -1) != 0) {
        p.v32 = want; p.tp = w32.tp; p.bn = w32.bn; p.nw = w32.nw;
      }
    }
  }
  p.ncb = (int)cdiv(a.Cout, p.bn);
  // 2-D tiles of TP/16 rows x 16 columns, or linear tiles of the virtual strip (narrow maps, where 2-D tiles are mostly padding)
  const int th = p.tp / 16;
  const int64_t t2 = (int64_t)a.N * cdiv(a.OH, th) * cdiv(a.OW, 16);
  const double eff2 = (double)a.M / ((double)t2 * p.tp);
  const int64_t strip = ((int64_t)a.N * (a.OH + 1) + 1) * (a.OW + 1);       // virtual pixels incl. the first and the last zero row
  const int64_t tl = cdiv(strip - 2 * (a.OW + 1), p.tp);
  const double effl = (double)a.M / ((double)tl * p.tp);
  const int npl = p.tp + 2 * (a.OW + 1) + 2;
  const bool lin_ok = strip < ((int64_t)1 << 22) && npl <= frag_pmax(p.tp);
  const int force_lin = rd_opt(OPT_FRAG_LIN, -1);   // test hook (rd_set_option "frag_lin"): 1 forces linear tiles where they fit, 0 forbids them
  bool lin = lin_ok && effl > eff2;
  if (force_lin == 0) lin = false;
  if (force_lin == 1) lin = lin_ok;
  if (p.v32) lin = true;      // (checked above)
  p.lin = lin ? 1 : 0;
  if (lin) { p.WT = a.OW + 1; p.tilesH = p.tilesW = 0; p.ntiles = (int)tl; p.np = npl; }
  else { p.WT = 18; p.tilesH = (int)cdiv(a.OH, th); p.tilesW = (int)cdiv(a.OW, 16); p.ntiles = (int)t2; p.np = (th + 2) * 18; }
  p.ps = ((p.np * 32 - 32 + 255) / 256) * 256 + 32;      // >= 32 np and = 32 (mod 256)
  return true;
}
bool conv3x3_frag_ok(const ConvArgs& a, int dtype) {
  FragPlan p;
  return rd_opt(OPT_CONV3X3_FRAG, 1) && frag_plan(a, dtype, p);
}
// ConvArgs::d2s on this kernel: 16-bit builds, whole 32-channel wave tiles inside one parity class, the block shapes the D2S instantiations exist for
bool conv3x3_frag_d2s_ok(const ConvArgs& a, int dtype) {
  FragPlan p;
  return dtype != 0 && a.C2 == 0 && !a.ups && !a.in_scale && a.Cout == 4 * a.D1 && (a.D1 % 32) == 0 && rd_opt(OPT_CONV3X3_FRAG, 1) && frag_plan(a, dtype, p) &&
         p.v32 == 0 && p.variant <= 2;
}
// ConvArgs::s2d on this kernel: 16-bit builds, whole 128-byte chunks of the 4 Cr (class, channel) axis, classes of 16 channels or more
bool conv3x3_frag_s2d_ok(const ConvArgs& a, int dtype) {
  FragPlan p;
  return dtype != 0 && a.C2 == 0 && !a.ups && !a.in_scale && (a.C1 & 3) == 0 && ((a.C1 >> 2) % 16) == 0 && rd_opt(OPT_CONV3X3_FRAG, 1) && frag_plan(a, dtype, p) && p.v32 == 0;
}
bool conv3x3_frag_is32(const ConvArgs& a, int dtype) { FragPlan p; return frag_plan(a, dtype, p) && p.v32 != 0; }
int conv3x3_frag_tiles(const ConvArgs& a, int dtype) { FragPlan p; frag_plan(a, dtype, p); return p.ntiles; }
int conv3x3_frag_blocks(const ConvArgs& a, int dtype) { FragPlan p; frag_plan(a, dtype, p); return p.ntiles * p.ncb; }

// the name rocprofv3 prints for the instantiation this shape runs on (dtype tag as in the kernel-trace CSV)
const char* conv3x3_frag_name(const ConvArgs& a, int dtype) {
  static thread_local char buf[96];
  FragPlan p; frag_plan(a, dtype, p);
  static const int npt[] = {8, 4, 4, 8, 4, 4, 8}, wpx[] = {1, 2, 2, 2, 4, 4, 4}, wch[] = {4, 4, 2, 2, 2, 1, 1};
  const bool multi = (a.C1 + a.C2) * (dtype == 0 ? 4 : 2) > STAGE_BYTES;
  if (p.v32) {
    const Frag32Variant& w = kFrag32Variants[p.v32];
    snprintf(buf, sizeof(buf), "conv3x3_frag32_kernel<%s, %d, %d, %d, %d, %s>", RD_T16_NAME, w.nq, w.wpx, w.wch, w.cw, multi ? "true" : "false");
    return buf;
  }
  // all nine template arguments, as rocprofv3 prints an instantiation (LIN, MULTI, AFF, D2S, S2D): bench.py matches its per-launch tallies to the
  // kernel trace by this exact string (until round 6 the D2S / S2D defaults were left out and the register-fed family never matched)
  const char* T16 = dtype == 0 ? "float" : RD_T16_NAME;
  if (a.s2d) snprintf(buf, sizeof(buf), "conv3x3_frag_kernel<%s, %d, %d, %d, %s, %s, false, false, true>", RD_T16_NAME, npt[p.variant], wpx[p.variant], wch[p.variant],
                      p.lin ? "true" : "false", multi ? "true" : "false");
  else if (a.d2s) snprintf(buf, sizeof(buf), "conv3x3_frag_kernel<%s, %d, %d, %d, %s, %s, false, true, false>", RD_T16_NAME, npt[p.variant], wpx[p.variant], wch[p.variant],
                      p.lin ? "true" : "false", multi ? "true" : "false");
  else
  snprintf(buf, sizeof(buf), "conv3x3_frag_kernel<%s, %d, %d, %d, %s, %s, %s, false, false>", T16, npt[p.variant], wpx[p.variant],
           wch[p.variant], p.lin ? "true" : "false", multi ? "true" : "false", a.in_scale ? "true" : "false");
  return buf;
}

template <typename T, int NPT, int WPX, int WCH>
static void launch_frag_v(const ConvArgs& a, const FragPlan& p, const FragGeom& g_, hipStream_t st) {
  const dim3 grid((unsigned)(p.ntiles * p.ncb)), block(64 * WPX * WCH);
  const bool aff = a.in_scale != nullptr;
  FragGeom g = g_;
  const bool multi = (a.C1 + a.C2) * (int)sizeof(T) > STAGE_BYTES;
  const int one = std::max(4 * p.ps, WPX * 32 * WCH * 2 * 4);      // the planes / the statistics scratch
  // second patch buffer of the multi-chunk layers (128-pixel tiles: two blocks of 2 x 33 KB still share a CU).  Measured (round 5,
  // tools/r05_fragdb.sh, alternating): 0.782 against 0.784-0.785 ms over the seventeen shapes of tools/bench_conv.py, the RC-Net step
  // 1077.6 against 1077.5 img/s -- the second barrier per chunk was not what the waves wait for.  Option frag_db, off by default.
  g.db = (multi && one <= 36 * 1024 && rd_opt(OPT_FRAG_DB, 0)) ? one : 0;
  g.aff = one + g.db;                                               // coefficient copy behind the buffer(s)
  const size_t lds = (size_t)g.aff + (aff ? (size_t)(a.C1 + a.C2) * 8 : 0);
#define RD_FR(LINV, MULTIV)                                                                                                         \
  { if (aff) hipLaunchKernelGGL((conv3x3_frag_kernel<T, NPT, WPX, WCH, LINV, MULTIV, true>), grid, block, lds, st, a, g);          \
    else hipLaunchKernelGGL((conv3x3_frag_kernel<T, NPT, WPX, WCH, LINV, MULTIV>), grid, block, lds, st, a, g); }
  if (a.s2d) {      // conv3x3_frag_s2d_ok: 16-bit, plain source
    if constexpr (sizeof(T) == 2) {
#define RD_FRS(LINV, MULTIV) hipLaunchKernelGGL((conv3x3_frag_kernel<T, NPT, WPX, WCH, LINV, MULTIV, false, false, true>), grid, block, lds, st, a, g);
      if (p.lin) { if (multi) RD_FRS(true, true) else RD_FRS(true, false) }
      else { if (multi) RD_FRS(false, true) else RD_FRS(false, false) }
#undef RD_FRS
    }
    return;
  }
  if (a.d2s) {      // conv3x3_frag_d2s_ok: 16-bit, the > 64-channel block shapes (variants 0-2), no consumer-side BatchNorm apply
    if constexpr (sizeof(T) == 2 && WCH >= 2 && NPT * WPX == 8) {
#define RD_FRD(LINV, MULTIV) hipLaunchKernelGGL((conv3x3_frag_kernel<T, NPT, WPX, WCH, LINV, MULTIV, false, true>), grid, block, lds, st, a, g);
      if (p.lin) { if (multi) RD_FRD(true, true) else RD_FRD(true, false) }
      else { if (multi) RD_FRD(false, true) else RD_FRD(false, false) }
#undef RD_FRD
    }
    return;
  }
  if (p.lin) { if (multi) RD_FR(true, true) else RD_FR(true, false) }
  else { if (multi) RD_FR(false, true) else RD_FR(false, false) }
#undef RD_FR
}
template <typename T, int NQ, int WPX, int WCH, int CW>
static void launch_frag32_v(const ConvArgs& a, const FragPlan& p, const FragGeom& g, hipStream_t st) {
  if constexpr (sizeof(T) == 2) {
    const dim3 grid((unsigned)(p.ntiles * p.ncb)), block(64 * WPX * WCH);
    const size_t lds = std::max<size_t>(8 * (size_t)frag32_ps(frag_pmax(32 * NQ * WPX)), (size_t)WPX * 32 * CW * WCH * 2 * 4);
    const bool multi = (a.C1 + a.C2) * (int)sizeof(T) > STAGE_BYTES;
    if (multi) hipLaunchKernelGGL((conv3x3_frag32_kernel<T, NQ, WPX, WCH, CW, true>), grid, block, lds, st, a, g);
    else hipLaunchKernelGGL((conv3x3_frag32_kernel<T, NQ, WPX, WCH, CW, false>), grid, block, lds, st, a, g);
  }
}
template <typename T>
static void launch_frag_t(const ConvArgs& a, int dtype, hipStream_t st) {
  FragPlan p; frag_plan(a, dtype, p);
  FragGeom g;
  g.lin = p.lin; g.WT = p.WT; g.tilesH = p.tilesH; g.tilesW = p.tilesW; g.ntiles = p.ntiles; g.ncb = p.ncb; g.np = p.np; g.ps = p.ps;
  const int rows_pad = conv_rows_pad(a.Cout);
  g.ctall = rows_pad / 16; g.wfrag = rows_pad * a.Kpad;
  g.rWT = 1.0f / (float)p.WT; g.rH1 = 1.0f / (float)(a.OH + 1); g.aff = 0;
  switch (p.v32) {
    case 1: launch_frag32_v<T, 2, 2, 4, 1>(a, p, g, st); return;
    case 2: launch_frag32_v<T, 4, 1, 4, 1>(a, p, g, st); return;
    case 3: launch_frag32_v<T, 2, 2, 2, 1>(a, p, g, st); return;
    default: break;
  }
  switch (p.variant) {
    case 0: launch_frag_v<T, 8, 1, 4>(a, p, g, st); break;
    case 1: launch_frag_v<T, 4, 2, 4>(a, p, g, st); break;
    case 2: launch_frag_v<T, 4, 2, 2>(a, p, g, st); break;
    case 3: launch_frag_v<T, 8, 2, 2>(a, p, g, st); break;
    case 4: launch_frag_v<T, 4, 4, 2>(a, p, g, st); break;
    case 5: launch_frag_v<T, 4, 4, 1>(a, p, g, st); break;
    default: launch_frag_v<T, 8, 4, 1>(a, p, g, st); break;
  }
}
void launch_conv3x3_frag(const ConvArgs& a, int dtype, hipStream_t st) {
  if (dtype == 0) launch_frag_t<float>(a, dtype, st);
  else launch_frag_t<bf16_t>(a, dtype, st);
}

}  // namespace rd
