// Removed from riders_amd/csrc/rd_wgrad3x3.hip in round 4 (opt-in since round 3, measured slower: 0.893 vs 0.773 ms over the eleven RC-Net
// weight-gradient shapes).  Kept here as the record of the experiment; not compiled into the library.  Depends on the helpers of that file.

// ---- linear tiling (narrow maps) --------------------------------------------------------------------------------------------------------
// The RoI maps of RC-Net are 15x6 ... 120x50 pixels: 8 x 16 (8 x 8) tiles cover them 70-78 %, i.e. a quarter of the MFMA work and of the
// staging traffic above goes to padding.  As in rd_conv3x3_frag.hip the whole tensor is treated as ONE strip of virtual pixels (rows of
// OW + 1 with a shared zero column, a shared zero row between images): a tile is 128 consecutive virtual pixels of dY, its patch the
// 128 + 2 (OW + 1) + 2 consecutive virtual pixels of x around them, tap (kh, kw) the offset kh (OW + 1) + kw.  Virtual zero pixels carry
// dY = 0 and contribute nothing, tiles cross image borders, and the layout / transpose-read roles are the ones of the tiled kernel with a
// run-time row stride (the per-wave column offsets are registers anyway).  80-97 % of the staged pixels are real.
// MEASURED SLOWER than the tiled kernel on RC-Net's layers (0.905 vs 0.831 ms over the eleven shapes of tools/bench_conv.py) although it
// issues 20-30 % fewer MFMAs: the kernel is bound by the issue rate of its 8-byte transpose reads and by the per-tile staging, not by
// MFMA work, and the strip decode adds vector instructions per staged slot.  Kept opt-in (RD_WGRAD_LIN=1; =2 forces it) with its tests.
struct LinGeom { int WT, H1, ntiles; float rWT, rH1; };

template <int CTI, int RT>
__global__ __launch_bounds__(256, (CTI * RT <= 2) ? 4 : 2) void conv3x3_wgrad_lin_kernel(WgradArgs a, LinGeom g, int nci) {
  typedef bf16_t T;
  constexpr int CIN = CTI * 16, COP = RT * 16;
  constexpr int TP = 128, NPX = 240, NPY = TP;        // dY pixels per tile; patch pixels (TP + 2 WT + 2 <= 240: WT <= 55, two blocks per CU)
  constexpr int XS = CIN / 8, YS = COP / 8;
  constexpr int NXS = NPX * XS, NYS = NPY * YS;
  constexpr int XIT = (NXS + 255) / 256, YIT = (NYS + 255) / 256;
  constexpr int KSTEPS = NPY / 32;
  constexpr int NCT = 9 * CTI, NCW = (NCT + 3) / 4;
  constexpr int XPSB = XS <= 2 ? NPX * 32 : ((NPX * 32 + 255) / 256) * 256 + (XS == 8 ? 32 : 64);
  constexpr int YPSB = YS <= 2 ? NPY * 32 : ((NPY * 32 + 255) / 256) * 256 + (YS == 8 ? 32 : 64);
  constexpr int XPS = XPSB / 2, YPS = YPSB / 2;
  constexpr int STEP = 32 * 16, R2 = 16 * 16;         // elements between k-steps / to the second transpose read (16 pixels further), both operands
  __shared__ __attribute__((aligned(16))) unsigned char sX[2][(XS / 2 > 0 ? XS / 2 : 1) * XPSB];
  __shared__ __attribute__((aligned(16))) unsigned char sY[2][(YS / 2 > 0 ? YS / 2 : 1) * YPSB];

  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  const int ci0 = ((int)blockIdx.y % nci) * CIN, co0 = ((int)blockIdx.y / nci) * COP;
  const int CinT = a.C1 + a.C2;
  const int WT = g.WT, H1 = g.H1, np = TP + 2 * WT + 2;

  const int ntiles = g.ntiles;
  const int T8 = (ntiles + 7) >> 3, G8 = gridDim.x >> 3;
  const int xcd = blockIdx.x & 7;
  const int tend = min(ntiles, (xcd + 1) * T8);
  int tile = xcd * T8 + (blockIdx.x >> 3);

  const int off = (fg * 4 + (fr >> 2)) * 16 + (fr & 3) * 4;      // this lane's pixel of a 16-pixel read, channels 4 (fr & 3) .. +3
  int coloff[NCW]; bool jv[NCW]; int jk[NCW];
#pragma unroll
  for (int j = 0; j < NCW; j++) {
    int idx = wv + 4 * j;
    jv[j] = idx < NCT;
    if (!jv[j]) idx = 0;
    const int tap = idx / CTI, ct = idx - tap * CTI;
    coloff[j] = ((tap / 3) * WT + (tap % 3)) * 16 + ct * XPS;
    jk[j] = tap * CinT + ci0 + ct * 16;
  }
  f32x4 acc[RT][NCW];
#pragma unroll
  for (int i = 0; i < RT; i++)
#pragma unroll
    for (int j = 0; j < NCW; j++) acc[i][j] = f32x4{0, 0, 0, 0};

  const bool yvec = (a.Cout & 7) == 0;
  // slot constants: patch pixel and channel slot of this thread's x slots, tile pixel and channel slot of its dY slots
  int xpp[XIT], xco[XIT];      // patch pixel | source-2 flag << 16 | valid << 17;  channel offset inside the source
#pragma unroll
  for (int i = 0; i < XIT; i++) {
    const int idx = t + 256 * i, pp = idx / XS, sl = idx - pp * XS;
    const int ci = ci0 + sl * 8;
    const bool s2 = ci >= a.C1;
    xpp[i] = pp | (s2 ? 1 << 16 : 0) | ((idx < NXS && pp < np) ? 1 << 17 : 0);
    xco[i] = s2 ? ci - a.C1 : ci;
  }
  const int Hp = a.ups ? a.H1 : a.Hin, Wp = a.ups ? a.W1 : a.Win;
  // virtual pixel -> tensor pixel (image, row, column); false for zero rows / columns and outside the strip
  auto strip_pixel = [&](int u, int& n, int& ih, int& iw) RD_INLINE_LAMBDA {
    int c, hh;
    const int vrow = fdiv_small(max(u, 0), WT, g.rWT, c);
    n = fdiv_small(vrow, H1, g.rH1, hh);
    ih = hh - 1; iw = c - 1;
    return u >= 0 && c >= 1 && hh >= 1 && n < a.N;
  };
  auto fetch = [&](int tl, uint4 (&rx)[XIT], uint4 (&ry)[YIT]) RD_INLINE_LAMBDA {
    const int u0 = WT + tl * TP;
#pragma unroll
    for (int i = 0; i < XIT; i++) {
      int n, ih, iw;
      const bool ok = strip_pixel(u0 - WT - 1 + (xpp[i] & 0xffff), n, ih, iw) && ((xpp[i] >> 17) & 1);
      int hs = ok ? ih : 0, ws = ok ? iw : 0;
      if (a.ups) {  // F.interpolate(mode='nearest') source index, ATen float formula (as conv_src_ptr)
        hs = min((int)floorf((float)hs * a.scale_h), a.H1 - 1);
        ws = min((int)floorf((float)ws * a.scale_w), a.W1 - 1);
      }
      const int pix = ok ? (n * Hp + hs) * Wp + ws : 0;
      const bool s2 = (xpp[i] >> 16) & 1;
      const T* sb = s2 ? (const T*)a.src2 : (const T*)a.src1;
      // unconditional load from a clamped address, zero selected afterwards
      const uint4 v = *reinterpret_cast<const uint4*>(sb + (int64_t)pix * (s2 ? a.C2 : a.C1) + xco[i]);
      rx[i] = ok ? v : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < YIT; i++) {
      const int idx = t + 256 * i, pp = idx / YS, sl = idx - pp * YS;
      int n, oh, ow;
      const bool ok = strip_pixel(u0 + pp, n, oh, ow) && idx < NYS && co0 + sl * 8 < a.Cout;
      const T* p = (const T*)a.dy + (int64_t)(ok ? (n * a.OH + oh) * a.OW + ow : 0) * a.Cout + (ok ? co0 + sl * 8 : 0);
      uint4 v;
      if (yvec) v = *reinterpret_cast<const uint4*>(p);
      else {  // Cout not a multiple of 8: element-wise, zero padded
        unsigned short e[8];
#pragma unroll
        for (int q = 0; q < 8; q++) e[q] = (ok && co0 + sl * 8 + q < a.Cout) ? p[q].v : (unsigned short)0;
        v.x = e[0] | ((unsigned)e[1] << 16); v.y = e[2] | ((unsigned)e[3] << 16);
        v.z = e[4] | ((unsigned)e[5] << 16); v.w = e[6] | ((unsigned)e[7] << 16);
      }
      ry[i] = ok ? v : make_uint4(0, 0, 0, 0);
    }
  };
  auto stash = [&](int buf, const uint4 (&rx)[XIT], const uint4 (&ry)[YIT]) RD_INLINE_LAMBDA {
#pragma unroll
    for (int i = 0; i < XIT; i++) {
      const int idx = t + 256 * i, pp = idx / XS, sl = idx - pp * XS;
      if (idx < NXS && pp < np) *reinterpret_cast<uint4*>(&sX[buf][(sl >> 1) * XPSB + pp * 32 + (sl & 1) * 16]) = rx[i];
    }
#pragma unroll
    for (int i = 0; i < YIT; i++) {
      const int idx = t + 256 * i, pp = idx / YS, sl = idx - pp * YS;
      if (idx < NYS) *reinterpret_cast<uint4*>(&sY[buf][(sl >> 1) * YPSB + pp * 32 + (sl & 1) * 16]) = ry[i];
    }
  };
  auto compute = [&](int buf, auto njc) RD_INLINE_LAMBDA {
    constexpr int NJ = decltype(njc)::value;
    const unsigned short* bx = reinterpret_cast<const unsigned short*>(&sX[buf][0]) + off;
    const unsigned short* by = reinterpret_cast<const unsigned short*>(&sY[buf][0]) + off;
#pragma unroll
    for (int s = 0; s < KSTEPS; s++) {
      s16x8 ya[RT];
#pragma unroll
      for (int i = 0; i < RT; i++) {
        uint2 lo = lds_read_tr16_b64(by + s * STEP + i * YPS), hi = lds_read_tr16_b64(by + s * STEP + i * YPS + R2);
        uint4 v = make_uint4(lo.x, lo.y, hi.x, hi.y);
        __builtin_memcpy(&ya[i], &v, 16);
      }
      s16x8 xb[NJ];      // all reads of the k-step, then its MFMAs (see the tiled kernel)
#pragma unroll
      for (int j = 0; j < NJ; j++) {
        uint2 lo = lds_read_tr16_b64(bx + s * STEP + coloff[j]), hi = lds_read_tr16_b64(bx + s * STEP + coloff[j] + R2);
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
  auto tile_body = [&](int buf) RD_INLINE_LAMBDA {
    if (jv[NCW - 1]) compute(buf, std::integral_constant<int, NCW>{});
    else compute(buf, std::integral_constant<int, NCW - 1>{});
  };

  uint4 xa[XIT], ya_[YIT];
  int buf = 0;
  if (tile < tend) fetch(tile, xa, ya_);
  while (tile < tend) {
    stash(buf, xa, ya_);
    __syncthreads();
    const int next = tile + G8;
    if (next < tend) fetch(next, xa, ya_);
    tile_body(buf);
    tile = next;
    buf ^= 1;
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

