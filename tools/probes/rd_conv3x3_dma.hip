// EXPERIMENTAL (off by default, RD_CONV3X3_DMA=1 enables it): bf16 3x3 / stride-1 convolution (forward and data gradient) for the wide
// layers -- input channels a multiple of 64, output channels a multiple of 32 -- on the 32x32x16 MFMA with LDS-DMA staging; an alternative
// to conv3x3_patch_kernel (rd_conv3x3.hip) for the same reference call sites (utils/net_utils.py:84-91,195-198,564-569).
//
// Status (round 2, profiles/r02_conv3x3_dma_experiment.txt): bit-exact (tests), but 0.8-0.95x the patch kernel on RC-Net's decoder layers,
// so the dispatcher keeps the patch kernel.  What was built and measured:
//   * the patch of a 64-channel chunk and the weight tiles go global -> LDS by DMA (global_load_lds_dwordx4: no staging registers, no
//     ds_write pass; the XOR-swizzled LDS image is produced by permuting the per-lane SOURCE slot; padding pixels are written as zeros by
//     the lane that owns them).  Through the builtin, hipcc puts `s_waitcnt vmcnt(0)` in front of the next ds_read (it treats the DMA as
//     an aliasing LDS write), which serialises prefetch and compute; issued through inline asm (rd_common.h) with counted waits it overlaps;
//   * weights travel in STAGES of TPS taps through a ring of RD stage buffers (counted s_waitcnt vmcnt, raw s_barrier);
//   * a wave owns 64 pixels x 32 or 64 output channels as 32x32x16 MFMA tiles; the generic (row >> 1) & 7 swizzle costs 8-12 LDS cycles per
//     32-pixel fragment read, a swizzle on the patch COORDINATES (patch_xor) brings it to the conflict-free 4.
// Why it does not win yet (PMC, same layer, both kernels at 70.8 M MFMA-busy cycles): twice the VALU and 2.5x the SALU instructions per wave
// (per-DMA M0 traffic, per-read XOR addressing that cannot use immediate offsets) and, for 64- and 32-channel tiles, fewer resident
// waves (LDS 41-81 KiB per block vs 31-39 KiB).  Kept as the starting point for a hand-scheduled K loop.
// Epilogue semantics (bias, activation, dual destination, per-block BatchNorm partial sums) are those of rd_conv_common.h.
#include "rd_conv_common.h"

namespace rd {

// pixel tile of a block: 64 * WPX pixels as TW x TH; a B fragment (32 pixels) is two tile rows of 16 or four of 8
template <int WPX, bool W8> struct DmaTile {
  static constexpr int TW = W8 ? 8 : 16;
  static constexpr int TH = 64 * WPX / TW;
  static constexpr int WT = TW + 2, HT = TH + 2, NP = HT * WT;
  static constexpr int PIT = (NP * 8 + 255) / 256;       // 16-byte patch pieces per thread and chunk
  static constexpr int PBUF = PIT * 256;                  // pieces per patch buffer (rounded up: every lane of every DMA has a slot)
};

// Patch swizzle.  A B fragment is 32 pixels = two patch rows of 16 (or four of 8) at patch-row stride TW + 2, read with ds_read_b128, whose
// lane groups are {0-3, 12-15, 20-27} / {4-11, 16-19, 28-31} (+32): the generic (row >> 1) & 7 XOR of lds_slot() costs 8 (16-wide) or 12
// (8-wide) LDS cycles per read there instead of 4.  With the XOR term taken from the patch COORDINATES -- 4 (py & 1) + ((px >> 1) & 3) for
// 16-wide tiles, 2 (py & 3) + ((px >> 1) & 1) for 8-wide ones -- the eight even-pixel lanes and the eight odd-pixel lanes of every group
// get eight different terms for every tap offset, i.e. all 16 lanes hit different 16-byte bank slots.
template <bool W8> __device__ __forceinline__ int patch_xor(int py, int px) {
  return W8 ? 2 * (py & 3) + ((px >> 1) & 1) : 4 * (py & 1) + ((px >> 1) & 3);
}

// wait until at most N of this wave's DMAs are outstanding (they complete in issue order)
template <int N> __device__ __forceinline__ void dma_wait_le() {
#ifndef RD_EMU
  if (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if (N == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
  else if (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else if (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
  else if (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
}
// workgroup barrier that does NOT drain the DMA queue (__syncthreads() waits vmcnt(0) while an LDS DMA is in flight): LDS traffic of this
// wave is complete (lgkmcnt), the DMAs the next phase reads were waited for by dma_wait_le
__device__ __forceinline__ void block_barrier_keep_dma() {
#ifdef RD_EMU
  __syncthreads();
#else
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
#endif
}

// BN output channels per block, WPX x WCO waves (WPX * WCO == 4), wave tile = 64 pixels x (BN / WCO) channels; weight stages of TPS taps
// in a ring of RD buffers
template <int BN, int WPX, int WCO, bool W8, int TPS, int RD>
__global__ __launch_bounds__(256) void conv3x3_dma_kernel(ConvArgs a, int tilesH, int tilesW, int pbufs) {
  using TL = DmaTile<WPX, W8>;
  constexpr int TW = TL::TW, TH = TL::TH, WT = TL::WT, NP = TL::NP, PIT = TL::PIT, PBUF = TL::PBUF;
  constexpr int MT = BN / WCO / 32;                       // 32-channel MFMA tiles per wave
  constexpr int WIT = BN * 8 / 256;                       // 16-byte weight pieces per thread and tap
  constexpr int SPC = 9 / TPS;                             // stages per chunk
  constexpr int WSTAGE = TPS * BN * 8;                    // 16-byte pieces per stage buffer
  static_assert(WPX * WCO == 4 && MT >= 1 && WIT >= 1 && TPS * SPC == 9 && RD >= 1 && RD <= 3, "tile configuration");
  RD_DYN_SMEM(smem);
  uint4* const sP = reinterpret_cast<uint4*>(smem);                       // [pbufs][PBUF]
  uint4* const sW = sP + pbufs * PBUF;                                     // [RD][TPS][BN * 8]

  const int t = threadIdx.x, lane = t & 63;
  const int wv = RD_WAVE_UNIFORM(t >> 6);
  int bx = blockIdx.x;
  {  // XCD-aware order (see rd_conv.hip)
    const int nb = gridDim.x, q = nb >> 3, r = nb & 7, xcd = bx & 7, idx = bx >> 3;
    bx = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int n0 = blockIdx.y * BN;
  const int Cin = a.C1 + a.C2;
  const int tw_ = bx % tilesW; const int q_ = bx / tilesW; const int th_ = q_ % tilesH; const int n = q_ / tilesH;
  const int oh0 = th_ * TH, ow0 = tw_ * TW;
  const int nchunk = Cin >> 6;

  // ---- patch pieces of this thread: linear LDS position P = (i * 4 + wave) * 64 + lane -> patch pixel P >> 3, LDS slot P & 7, which holds
  // SOURCE slot (P & 7) ^ patch_xor(pixel) (undone by the fragment loads).  Decoded once per block.
  // The per-lane part of a piece's source address -- (pixel * C + 8 * slot) * 2 bytes inside ITS source tensor -- is a constant of the
  // thread; the chunk only moves the wave-uniform base (src + 128 bytes per chunk; the concat boundary is a multiple of 64 channels or the
  // layer has one chunk per source decision made per slot, see UNI below), so a DMA costs no vector address arithmetic.
  int spix[PIT], sslot[PIT];
  unsigned soff1[PIT], soff2[PIT];
  const bool UNI = (a.C1 & 63) == 0;                     // every 64-channel chunk lies in ONE source tensor
  {
    const int Hp = a.ups ? a.H1 : a.Hin, Wp = a.ups ? a.W1 : a.Win;
#pragma unroll
    for (int i = 0; i < PIT; i++) {
      const int P = (i * 4 + wv) * 64 + lane;
      const int pp = P >> 3;
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
      sslot[i] = (P & 7) ^ patch_xor<W8>(py, px);
      soff1[i] = (unsigned)(pix * a.C1 + sslot[i] * 8) * 2u;               // byte offsets fit 32 bits (conv3x3_dma_ok)
      soff2[i] = (unsigned)(pix * a.C2 + sslot[i] * 8) * 2u;
    }
  }
  auto dma_patch = [&](int chunk, int buf) RD_INLINE_LAMBDA {
    uint4* const base = sP + buf * PBUF;
    const int c0 = chunk * 64;
    if (UNI) {
      const bool first = c0 < a.C1;
      const bf16_t* const g = first ? (const bf16_t*)a.src1 + c0 : (const bf16_t*)a.src2 + (c0 - a.C1);
#pragma unroll
      for (int i = 0; i < PIT; i++) {
        uint4* const wbase = base + (i * 4 + wv) * 64;                     // wave-uniform
        if (spix[i] >= 0) dma16_to_lds_base(g, first ? soff1[i] : soff2[i], wbase);
        else wbase[lane] = make_uint4(0, 0, 0, 0);                         // zero padding (and the unused tail of the buffer)
      }
    } else {
#pragma unroll
      for (int i = 0; i < PIT; i++) {
        const int ci = c0 + sslot[i] * 8;
        uint4* const wbase = base + (i * 4 + wv) * 64;
        if (spix[i] >= 0) {
          const bf16_t* g = ci < a.C1 ? (const bf16_t*)a.src1 + ((int64_t)spix[i] * a.C1 + ci)
                                      : (const bf16_t*)a.src2 + ((int64_t)spix[i] * a.C2 + (ci - a.C1));
          dma16_to_lds(g, wbase);
        } else {
          wbase[lane] = make_uint4(0, 0, 0, 0);
        }
      }
    }
  };
  // ---- weight pieces: packed row (output channel) = [tap][Cin] bf16, 128 contiguous bytes per (row, tap, chunk)
  const bf16_t* const wp = (const bf16_t*)a.w;
  unsigned woff[WIT];                                       // per-lane byte offset of the piece inside the packed matrix (row, swizzled slot)
#pragma unroll
  for (int i = 0; i < WIT; i++) {
    const int P = (i * 4 + wv) * 64 + lane;
    const int row = P >> 3;
    woff[i] = (unsigned)(((n0 + row) * a.Kpad + (((P & 7) ^ ((row >> 1) & 7)) * 8)) * 2);
  }
  auto dma_stage = [&](int stage) RD_INLINE_LAMBDA {        // global stage index -> (chunk, first tap), ring slot stage % RD
    const int chunk = stage / SPC, tap0 = (stage - chunk * SPC) * TPS;
    uint4* const base = sW + (stage % RD) * WSTAGE;
#pragma unroll
    for (int tp = 0; tp < TPS; tp++) {
      const bf16_t* const g = wp + ((tap0 + tp) * Cin + chunk * 64);        // wave-uniform
#pragma unroll
      for (int i = 0; i < WIT; i++) dma16_to_lds_base(g, woff[i], base + tp * (BN * 8) + (i * 4 + wv) * 64);
    }
  };

  // ---- wave tiling
  const int wpx = wv % WPX, wco = wv / WPX;
  const int l32 = lane & 31, lhi = lane >> 5;
  int ppy[2], ppx;                                          // this lane's pixel of the wave's two 32-pixel fragments (patch row / column of tap (0,0))
  ppx = W8 ? (l32 & 7) : (l32 & 15);
#pragma unroll
  for (int pt = 0; pt < 2; pt++) {
    const int q = wpx * 2 + pt;                             // 32-pixel group of the block tile
    ppy[pt] = W8 ? q * 4 + (l32 >> 3) : q * 2 + (l32 >> 4);
  }
  int wfa[MT], wfx[MT];                                     // weight-fragment address: row * 8 and the row's XOR term (slot is added per k-step)
#pragma unroll
  for (int m = 0; m < MT; m++) { const int row = (wco * MT + m) * 32 + l32; wfa[m] = row * 8; wfx[m] = (row >> 1) & 7; }

  f32x16 acc[MT][2];
#pragma unroll
  for (int m = 0; m < MT; m++)
#pragma unroll
    for (int pt = 0; pt < 2; pt++)
#pragma unroll
      for (int v = 0; v < 16; v++) acc[m][pt][v] = 0.f;

  const int nstage = nchunk * SPC;
  dma_patch(0, 0);
#pragma unroll
  for (int s = 0; s < RD - 1; s++)
    if (s < nstage) dma_stage(s);
  if (RD == 1) dma_stage(0);
  dma_wait_all();
  __syncthreads();
  for (int stage = 0; stage < nstage; stage++) {
    const int chunk = stage / SPC, tap0 = (stage - chunk * SPC) * TPS;
    const uint4* const pb = sP + (pbufs > 1 ? (chunk & 1) : 0) * PBUF;
    // requested now, needed RD - 1 stages from now: the weights of stage + RD - 1; with two patch buffers also the next chunk's patch
    // (issued FIRST so that the weight stage stays the newest entry of the DMA queue: the counted wait below leaves exactly it in flight)
    if (tap0 == 0 && pbufs > 1 && chunk + 1 < nchunk) dma_patch(chunk + 1, (chunk + 1) & 1);
    if (RD > 1 && stage + RD - 1 < nstage) dma_stage(stage + RD - 1);
    const uint4* const wb0 = sW + (stage % RD) * WSTAGE;
#pragma unroll
    for (int tp = 0; tp < TPS; tp++) {
      const int tap = tap0 + tp;
      const int kh = tap / 3, kw = tap - kh * 3;
      const uint4* const wb = wb0 + tp * (BN * 8);
      int pidx[2], pxr[2];
#pragma unroll
      for (int pt = 0; pt < 2; pt++) {
        pidx[pt] = ((ppy[pt] + kh) * WT + ppx + kw) * 8;
        pxr[pt] = patch_xor<W8>(ppy[pt] + kh, ppx + kw);
      }
#pragma unroll
      for (int ks = 0; ks < 4; ks++) {                      // 16 channels per MFMA k-step: 16-byte slot 2 ks + (lane >> 5)
        const int slot = ks * 2 + lhi;
        uint4 pf[2], wf[MT];
#pragma unroll
        for (int pt = 0; pt < 2; pt++) pf[pt] = pb[pidx[pt] + (slot ^ pxr[pt])];
#pragma unroll
        for (int m = 0; m < MT; m++) wf[m] = wb[wfa[m] + (slot ^ wfx[m])];
#pragma unroll
        for (int m = 0; m < MT; m++)
#pragma unroll
          for (int pt = 0; pt < 2; pt++) {
            s16x8 wa, pbv;
            __builtin_memcpy(&wa, &wf[m], 16);
            __builtin_memcpy(&pbv, &pf[pt], 16);
            acc[m][pt] = mfma_32x32x16_bf16(wa, pbv, acc[m][pt]);
          }
      }
    }
    if (stage + 1 >= nstage) break;
    const bool chunk_end = tap0 + TPS == 9;
    if (RD == 1) {                                          // single stage buffer: refill it once every wave is done with it
      block_barrier_keep_dma();
      if (chunk_end && pbufs == 1) dma_patch(chunk + 1, 0);
      dma_stage(stage + 1);
      dma_wait_all();
    } else if (chunk_end && pbufs == 1) {                   // single patch buffer: refill it once every wave is done with this chunk
      block_barrier_keep_dma();
      dma_patch(chunk + 1, 0);
      dma_wait_all();
    } else if (RD == 2 || stage + RD - 1 >= nstage) {
      dma_wait_all();                                       // the next stage (and patch) has landed
    } else {
      dma_wait_le<TPS * WIT>();                             // everything but the newest stage has landed
    }
    block_barrier_keep_dma();                               // ... for every wave; and every wave is done with this stage's buffer
  }

  // ---- epilogue: lane holds, for each of its two pixels, channel groups co = tile base + 8 j + 4 (lane >> 5) + (0..3), j = v >> 2
  float* const red = reinterpret_cast<float*>(smem);       // [WPX][BN][2] partial sums, reuses the patch buffer
  const int D2 = a.Cout - a.D1;
  const bool has_bias = a.bias != nullptr;
  const bool plain = !has_bias && a.act == ACT_NONE;
  float ssum[MT][4][4], ssq[MT][4][4];
#pragma unroll
  for (int m = 0; m < MT; m++)
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
      for (int r = 0; r < 4; r++) { ssum[m][j][r] = 0.f; ssq[m][j][r] = 0.f; }
#pragma unroll
  for (int pt = 0; pt < 2; pt++) {
    const int q = wpx * 2 + pt;
    const int py = W8 ? q * 4 + (l32 >> 3) : q * 2 + (l32 >> 4);
    const int px = W8 ? (l32 & 7) : (l32 & 15);
    const int oh = oh0 + py, ow = ow0 + px;
    if (oh >= a.OH || ow >= a.OW) continue;
    const int64_t mpix = ((int64_t)n * a.OH + oh) * a.OW + ow;
    bf16_t* const p1 = (bf16_t*)a.dst1 + mpix * a.D1;
    bf16_t* const p2 = (bf16_t*)a.dst2 + mpix * D2 - a.D1;   // indexed by the global channel (only dereferenced for co >= D1)
#pragma unroll
    for (int m = 0; m < MT; m++)
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int co = n0 + (wco * MT + m) * 32 + j * 8 + lhi * 4;
        if (co >= a.Cout) continue;                          // Cout is a multiple of 4 here (conv3x3_dma_ok): a group is in or out as a whole
        float x[4], xr[4];
#pragma unroll
        for (int r = 0; r < 4; r++) x[r] = acc[m][pt][j * 4 + r];
        if (!plain) {
#pragma unroll
          for (int r = 0; r < 4; r++) {
            if (has_bias) x[r] += a.bias[co + r];
            x[r] = act_fwd(x[r], a.act, a.slope);
          }
        }
        round_store4((co < a.D1 ? p1 : p2) + co, x, xr);
#pragma unroll
        for (int r = 0; r < 4; r++) { ssum[m][j][r] += xr[r]; ssq[m][j][r] += xr[r] * xr[r]; }
      }
  }
  if (!a.stats) return;
  __syncthreads();                                          // every wave is done with the LDS buffers (reused as scratch below)
#pragma unroll
  for (int m = 0; m < MT; m++)
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        float s1 = ssum[m][j][r], s2 = ssq[m][j][r];
#pragma unroll
        for (int o = 1; o < 32; o <<= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
        if (l32 == 0) {
          const int col = (wco * MT + m) * 32 + j * 8 + lhi * 4 + r;
          red[(wpx * BN + col) * 2 + 0] = s1;
          red[(wpx * BN + col) * 2 + 1] = s2;
        }
      }
  __syncthreads();
  for (int col = t; col < BN; col += 256) {
    const int co = n0 + col;
    if (co < a.Cout) {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int w = 0; w < WPX; w++) { s1 += red[(w * BN + col) * 2]; s2 += red[(w * BN + col) * 2 + 1]; }
      a.stats[((int64_t)bx * a.Cout + co) * 2 + 0] = s1;
      a.stats[((int64_t)bx * a.Cout + co) * 2 + 1] = s2;
    }
  }
}

// ---- host side -------------------------------------------------------------------------------------------------------------------
static int dma_bn(int cout) { return cout <= 32 ? 32 : (cout <= 64 ? 64 : 128); }
static int dma_wpx(int bn) { return bn == 32 ? 4 : 2; }    // 256-pixel tiles for the narrowest layers (two pixel fragments per weight fragment)

bool conv3x3_dma_ok(const ConvArgs& a, int dtype) {
  const char* e = getenv("RD_CONV3X3_DMA");             // opt-in: see the status note at the top of this file
  if (!e || atoi(e) == 0) return false;
  const int Cin = a.C1 + a.C2;
  if (dtype != 1) return false;
  // per-lane source byte offsets and packed-weight offsets are kept in 32 bits
  if ((int64_t)a.N * a.H1 * a.W1 * std::max(a.C1, a.C2) * 2 >= (int64_t)1 << 31 || (int64_t)a.Cout * a.Kpad * 2 + (1 << 20) >= (int64_t)1 << 31) return false;
  return a.KH == 3 && a.KW == 3 && a.stride == 1 && a.pad == 1 && a.dil == 1 && a.OH == a.Hin && a.OW == a.Win && (Cin % 64 == 0) &&
         (a.C1 % 8 == 0) && (a.Cout % 32 == 0) && (a.D1 % 4 == 0);
}
static bool dma_w8(const ConvArgs& a, int wpx) {            // tile shape with the smaller padded area
  const int th16 = 64 * wpx / 16, th8 = 64 * wpx / 8;
  const int64_t a16 = cdiv(a.OH, th16) * th16 * cdiv(a.OW, 16) * 16, a8 = cdiv(a.OH, th8) * th8 * cdiv(a.OW, 8) * 8;
  return a8 < a16;
}
int conv3x3_dma_tiles(const ConvArgs& a) {
  const int wpx = dma_wpx(dma_bn(a.Cout));
  const bool w8 = dma_w8(a, wpx);
  const int tw = w8 ? 8 : 16, th = 64 * wpx / tw;
  return a.N * (int)cdiv(a.OH, th) * (int)cdiv(a.OW, tw);
}
static int dma_env(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }   // experiment hooks

template <int BN, int WPX, int WCO, bool W8, int TPS, int RD>
static void launch_dma_cfg(const ConvArgs& a, hipStream_t st) {
  using TL = DmaTile<WPX, W8>;
  const int tilesH = (int)cdiv(a.OH, TL::TH), tilesW = (int)cdiv(a.OW, TL::TW);
  const int nchunk = (a.C1 + a.C2) / 64;
  // a second patch buffer (the next chunk's patch travels during the chunk) only where it does not cost a resident block (160 KiB per CU)
  const int wbytes = RD * TPS * BN * 128, pbytes = TL::PBUF * 16;
  int pbufs = 1;
  if (nchunk > 1 && (160 * 1024) / (2 * pbytes + wbytes) >= std::min(2, (160 * 1024) / (pbytes + wbytes))) pbufs = 2;
  pbufs = dma_env("RD_DMA_PBUFS", pbufs);
  if (nchunk == 1) pbufs = 1;
  const size_t lds = (size_t)pbufs * pbytes + wbytes;
  static bool attr_done = false;
  if (!attr_done) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_dma_kernel<BN, WPX, WCO, W8, TPS, RD>), hipFuncAttributeMaxDynamicSharedMemorySize,
                        160 * 1024);
    attr_done = true;
  }
  dim3 grid((unsigned)(a.N * tilesH * tilesW), (unsigned)cdiv(a.Cout, BN));
  hipLaunchKernelGGL((conv3x3_dma_kernel<BN, WPX, WCO, W8, TPS, RD>), grid, dim3(256), lds, st, a, tilesH, tilesW, pbufs);
}
template <int BN, int WPX, int WCO, int TPS, int RD>
static void launch_dma_w8(const ConvArgs& a, bool w8, hipStream_t st) {
  if (w8) launch_dma_cfg<BN, WPX, WCO, true, TPS, RD>(a, st);
  else launch_dma_cfg<BN, WPX, WCO, false, TPS, RD>(a, st);
}
void launch_conv3x3_dma(const ConvArgs& a, hipStream_t st) {
  const int bn = dma_bn(a.Cout), wpx = dma_wpx(bn);
  const bool w8 = dma_w8(a, wpx);
  const int nchunk = (a.C1 + a.C2) / 64;
  // weight staging: (taps per stage, ring depth)
  //   32 channels, one chunk: all nine taps once (36 KiB), no barrier in the K loop;  otherwise kernel rows (3 taps) double-buffered
  //   64 channels: kernel rows double-buffered (48 KiB) or single taps in a ring of three (24 KiB, one more resident block)
  //   128 channels: single taps in a ring of three (48 KiB)
  const int mode = dma_env("RD_DMA_MODE", 0);
  if (bn == 32) {
    if (nchunk == 1 && mode != 2) launch_dma_w8<32, 4, 1, 9, 1>(a, w8, st);
    else launch_dma_w8<32, 4, 1, 3, 2>(a, w8, st);
  } else if (bn == 64) {
    if (mode == 1) launch_dma_w8<64, 2, 2, 1, 3>(a, w8, st);
    else launch_dma_w8<64, 2, 2, 3, 2>(a, w8, st);
  } else {
    if (mode == 1) launch_dma_w8<128, 2, 2, 3, 2>(a, w8, st);
    else launch_dma_w8<128, 2, 2, 1, 3>(a, w8, st);
  }
}

}  // namespace rd
