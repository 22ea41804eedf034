// warp.hip -- perspective rectification of the card to 428 x 270.
//
// Replaces, for a whole batch, llcv_unwarp's CPU branch (cv/warp.cpp:153-166):
// cvWarpPerspective(src, dst, M, CV_INTER_LINEAR + CV_WARP_FILL_OUTLIERS, 0) with the
// OpenCV 2.4 semantics restated in SURVEY.md Appendix A10: per destination pixel in
// 64-wide blocks  X0 = M0*x + M1*y + M2 (fp64), then (X0 + M0*x1) * (32 / W), cvRound
// to 1/32 px, 5-bit fractions, (sum p*w + 2^14) >> 15 bilinear blend, zero outside the
// source.  Every fp64 operation is a single IEEE operation in that order
// (-ffp-contract=off), so the card is byte-exact.
//
// CDNA4 mapping: one workgroup per 64 x 32 destination tile (two of OpenCV's 64 x 16
// blocks; the x block origin -- the only one that enters the fp64 association -- is the
// same).  The kernel is fp64-VALU bound, so the work per pixel is trimmed to what
// exactness needs:
//   * 32/W: the IEEE-exact v_rcp_f64 + fma sequence the compiler itself emits for an f64
//     division, without its v_div_scale / v_div_fixup wrapper (quarter-rate instructions
//     that are the identity for |W| in [2^-500, 2^500]; anything else takes the full `/`).
//   * cvRound: one fp64 add of 1.5*2^52 (round-to-nearest-even by the adder) instead of
//     v_rndne_f64 + v_cvt_i32_f64; |f| >= 2^31 takes the saturating conversion.
//   * the tile's source window comes from its four corner pixels (a projective map with W
//     of constant sign is monotone along lines, +-1 px for rounding) and is staged into LDS
//     with aligned 32-bit row loads (zeros outside the image = BORDER_CONSTANT 0); the
//     bilinear taps are an aligned dword pair + v_alignbyte + v_dot4_u32_u8 per row.
// Tiles whose window exceeds the LDS buffer or whose W changes sign (extreme caller-supplied
// matrices) take the direct global path.  Blocks are renumbered so that all tiles of a
// frame run on one XCD (block b is dispatched to XCD b % 8) and share its L2.
#include "dmz_hip_internal.h"

namespace {

// developer ablation (tools/ablate.sh): 1 = cheap coordinates, 2 = no staging, 3 = no blend
#ifndef DMZ_WARP_ABLATE
#define DMZ_WARP_ABLATE 0
#endif

constexpr int TW = 64, TH = 32;
constexpr int kTilesX = (DMZ_CARD_WIDTH + TW - 1) / TW;   // 7
constexpr int kTilesY = (DMZ_CARD_HEIGHT + TH - 1) / TH;  // 9
constexpr int kTiles = kTilesX * kTilesY;                 // 63
constexpr int LW = 192;  // LDS window row stride: 48 dwords = 16 mod 32 banks, so the two rows a 32-lane
                         // group reads land on disjoint banks; windows up to 112 px wide are staged
constexpr int LWMAX = 112;
constexpr int LH = 64;   // rows

__device__ __forceinline__ int sat16(int v) { return v < -32768 ? -32768 : (v > 32767 ? 32767 : v); }
__device__ __forceinline__ int imin(int a, int b) { return a < b ? a : b; }
__device__ __forceinline__ int imax(int a, int b) { return a > b ? a : b; }

// W ? 32./W : 0 with IEEE-754 correctly rounded division (see file header)
__device__ __forceinline__ double div32(double W) {
  const int e = (__double2hiint(W) >> 20) & 0x7ff;
  if (__builtin_expect(e > 523 && e < 1523, 1)) {
    double y = __builtin_amdgcn_rcp(W);
    double t = __builtin_fma(-W, y, 1.0);
    y = __builtin_fma(y, t, y);
    t = __builtin_fma(-W, y, 1.0);
    y = __builtin_fma(y, t, y);
    const double q = 32.0 * y;
    const double r = __builtin_fma(-W, q, 32.0);
    return __builtin_fma(r, y, q);
  }
  return W ? 32. / W : 0;
}

// saturate_cast<int>(double) = cvRound after OpenCV's clamp to [INT_MIN, INT_MAX]
__device__ __forceinline__ int rne_i32(double f) {
  if (__builtin_expect((__double2hiint(f) & 0x7fffffff) < 0x41E00000, 1))  // |f| < 2^31
    return __double2loint(f + 6755399441055744.0);                          // 1.5 * 2^52
  return __double2int_rn(f);  // saturating (v_cvt_i32_f64 clamps like OpenCV's min/max)
}

struct SrcXY {
  int X, Y;  // fixed point, 5 fractional bits
};

// FAST: no range checks -- valid for every pixel of a tile whose four corner pixels have W of
// one sign with moderate exponents and |X|, |Y| < 2^30 (W is affine and X, Y are monotone
// along lines, so every pixel of the tile lies between the corner values).
template <bool FAST>
__device__ __forceinline__ SrcXY map_pixel(double X0, double Y0, double W0, double M0, double M3,
                                           double M6, int x1) {
  SrcXY r;
  if (DMZ_WARP_ABLATE == 1) {
    r.X = (int)(float)X0 * 32 + x1 * 32;
    r.Y = (int)(float)Y0 * 32;
    return r;
  }
  if (FAST) {
    const double Wd = W0 + M6 * x1;
    double y = __builtin_amdgcn_rcp(Wd);
    double t = __builtin_fma(-Wd, y, 1.0);
    y = __builtin_fma(y, t, y);
    t = __builtin_fma(-Wd, y, 1.0);
    y = __builtin_fma(y, t, y);
    const double q = 32.0 * y;
    const double rr = __builtin_fma(-Wd, q, 32.0);
    const double W = __builtin_fma(rr, y, q);
    r.X = __double2loint((X0 + M0 * x1) * W + 6755399441055744.0);
    r.Y = __double2loint((Y0 + M3 * x1) * W + 6755399441055744.0);
    return r;
  }
  const double W = div32(W0 + M6 * x1);
  r.X = rne_i32((X0 + M0 * x1) * W);
  r.Y = rne_i32((Y0 + M3 * x1) * W);
  return r;
}

// direct global bilinear tap fetch with BORDER_CONSTANT 0 (fallback path)
__device__ __forceinline__ void taps_global(const uint8_t *__restrict__ src, int row_stride, int sw,
                                            int sh, int sx, int sy, int &v0, int &v1, int &v2, int &v3) {
  if ((unsigned)sx < (unsigned)(sw - 1) && (unsigned)sy < (unsigned)(sh - 1)) {
    const uint8_t *p = src + (size_t)sy * row_stride + sx;
    v0 = p[0]; v1 = p[1]; v2 = p[row_stride]; v3 = p[row_stride + 1];
  } else if (sx >= sw || sx + 1 < 0 || sy >= sh || sy + 1 < 0) {
    v0 = v1 = v2 = v3 = 0;
  } else {
    const bool x0ok = sx >= 0 && sx < sw, x1ok = sx + 1 >= 0 && sx + 1 < sw;
    const bool y0ok = sy >= 0 && sy < sh, y1ok = sy + 1 >= 0 && sy + 1 < sh;
    v0 = (x0ok && y0ok) ? src[(size_t)sy * row_stride + sx] : 0;
    v1 = (x1ok && y0ok) ? src[(size_t)sy * row_stride + sx + 1] : 0;
    v2 = (x0ok && y1ok) ? src[(size_t)(sy + 1) * row_stride + sx] : 0;
    v3 = (x1ok && y1ok) ? src[(size_t)(sy + 1) * row_stride + sx + 1] : 0;
  }
}

__global__ __launch_bounds__(256) void k_warp(const uint8_t *__restrict__ planes, size_t frame_stride,
                                               int row_stride, int sw, int sh, int n, int n_pad,
                                               const DmzWarpMat *__restrict__ mats,
                                               uint8_t *__restrict__ cards, size_t card_stride) {
  __shared__ __attribute__((aligned(16))) unsigned char win[LW * LH + 8];
  __shared__ int s_corner[4][4];  // per corner: sx, sy, sign of W, unused

  // XCD-aware renumbering: logical id = xcd * (blocks/8) + k
  const unsigned int nblk = (unsigned int)n_pad * kTiles;
  const unsigned int b = blockIdx.x;
  const unsigned int logical = (b & 7u) * (nblk >> 3) + (b >> 3);
  const int frame = (int)(logical / kTiles);
  const int tile = (int)(logical - (unsigned int)frame * kTiles);
  if (frame >= n) return;
  const int ty = tile / kTilesX, tx = tile - ty * kTilesX;
  const int tid = threadIdx.x;
  const int x = tx * TW;                  // OpenCV block origin in x
  const int xq = (tid & 15) * 4;          // first of 4 pixels, relative to x
  const int yr = ty * TH + (tid >> 4);    // rows yr and yr + 16
  uint8_t *dbase = cards + (size_t)frame * card_stride;
  const DmzWarpMat &wm = mats[frame];
  const bool col_ok = x + xq < DMZ_CARD_WIDTH;  // 428 % 4 == 0: a 4-pixel group is all in or all out
  if (!wm.valid) {
    if (col_ok) {
      if (yr < DMZ_CARD_HEIGHT) *(uint32_t *)(dbase + (size_t)yr * DMZ_CARD_WIDTH + x + xq) = 0u;
      if (yr + 16 < DMZ_CARD_HEIGHT) *(uint32_t *)(dbase + (size_t)(yr + 16) * DMZ_CARD_WIDTH + x + xq) = 0u;
    }
    return;
  }
  const double M0 = wm.m[0], M1 = wm.m[1], M2 = wm.m[2], M3 = wm.m[3], M4 = wm.m[4], M5 = wm.m[5],
               M6 = wm.m[6], M7 = wm.m[7], M8 = wm.m[8];

  // ---- source window from the four corner pixels of the tile ----
  if (tid < 4) {
    const int cx1 = (tid & 1) ? imin(TW, DMZ_CARD_WIDTH - x) - 1 : 0;
    const int cy = ty * TH + ((tid & 2) ? imin(TH, DMZ_CARD_HEIGHT - ty * TH) - 1 : 0);
    const double W0 = M6 * x + M7 * cy + M8;
    const double Wc = W0 + M6 * cx1;
    const SrcXY p = map_pixel<false>(M0 * x + M1 * cy + M2, M3 * x + M4 * cy + M5, W0, M0, M3, M6, cx1);
    s_corner[tid][0] = sat16(p.X >> 5);
    s_corner[tid][1] = sat16(p.Y >> 5);
    const int we = (__double2hiint(Wc) >> 20) & 0x7ff;
    const bool tame = we > 900 && we < 1150 && p.X > -(1 << 30) && p.X < (1 << 30) && p.Y > -(1 << 30) &&
                      p.Y < (1 << 30);
    s_corner[tid][2] = !tame ? 0 : (Wc > 0. ? 1 : -1);
  }
  __syncthreads();
  const int bx0 = imin(imin(s_corner[0][0], s_corner[1][0]), imin(s_corner[2][0], s_corner[3][0])) - 1;
  const int bx1 = imax(imax(s_corner[0][0], s_corner[1][0]), imax(s_corner[2][0], s_corner[3][0])) + 1;
  const int by0 = imin(imin(s_corner[0][1], s_corner[1][1]), imin(s_corner[2][1], s_corner[3][1])) - 1;
  const int by1 = imax(imax(s_corner[0][1], s_corner[1][1]), imax(s_corner[2][1], s_corner[3][1])) + 1;
  const int sgn = s_corner[0][2];
  const bool same_sign = sgn != 0 && s_corner[1][2] == sgn && s_corner[2][2] == sgn && s_corner[3][2] == sgn;
  const int wx0 = bx0 & ~3;          // window origin, 4-aligned in x
  const int wy0 = by0;
  const int wcols = bx1 + 2 - wx0;   // + the right bilinear tap
  const int wrows = by1 + 2 - wy0;
  const uint8_t *src = planes + (size_t)frame * frame_stride;
  const bool fastxy = same_sign && wcols <= LWMAX && wrows <= LH;
  const bool staged = fastxy && DMZ_WARP_ABLATE != 4;

  // ---- stage the window: thread (tid & 31) owns one dword column, 8 rows per pass.  For a
  // window that lies inside the image (every card that is inside the frame) the loads are
  // plain predicated dword loads, all issued here and landing in registers while the fp64
  // coordinate math below runs; they are written to LDS after it.  Windows that cross the
  // image border are staged by the generic byte-checked loop.
  uint32_t stg[LH / 8];
  const int sq = tid & 31, sj = tid >> 5;
  const int wdw = (wcols + 3) >> 2;  // <= 28
  const bool interior = ((((uintptr_t)src) | (uintptr_t)row_stride) & 3) == 0 && wx0 >= 0 &&
                        wx0 + 4 * wdw <= sw && wy0 >= 0 && wy0 + wrows <= sh;
  const bool do_stage = staged && DMZ_WARP_ABLATE != 2;
  if (do_stage && interior) {
    const uint8_t *g = src + (size_t)(wy0 + sj) * row_stride + wx0 + 4 * sq;
#pragma unroll
    for (int it = 0; it < LH / 8; it++) {
      stg[it] = 0u;
      if (sq < wdw && sj + 8 * it < wrows) stg[it] = *(const uint32_t *)(g + (size_t)(8 * it) * row_stride);
    }
  } else if (do_stage) {
    for (int i = tid; i < wdw * wrows; i += 256) {
      const int j = i / wdw, q = i - j * wdw;
      const int gy = wy0 + j, gx = wx0 + 4 * q;
      uint32_t v = 0u;
      if (gy >= 0 && gy < sh) {
        const uint8_t *g = src + (size_t)gy * row_stride + gx;
        for (int k = 0; k < 4; k++)
          if (gx + k >= 0 && gx + k < sw) v |= (uint32_t)g[k] << (8 * k);
      }
      *(uint32_t *)(win + j * LW + 4 * q) = v;
    }
  }

  // ---- fixed-point source coordinates of this thread's 2 x 4 pixels (overlaps the loads) ----
  SrcXY P[2][4];
#pragma unroll
  for (int h = 0; h < 2; h++) {
    const int y = yr + 16 * h;
    const double X0 = M0 * x + M1 * y + M2;
    const double Y0 = M3 * x + M4 * y + M5;
    const double W0 = M6 * x + M7 * y + M8;
    if (fastxy) {
#pragma unroll
      for (int k = 0; k < 4; k++) P[h][k] = map_pixel<true>(X0, Y0, W0, M0, M3, M6, xq + k);
    } else {
#pragma unroll
      for (int k = 0; k < 4; k++) P[h][k] = map_pixel<false>(X0, Y0, W0, M0, M3, M6, xq + k);
    }
  }
  if (do_stage && interior && sq < wdw) {
#pragma unroll
    for (int it = 0; it < LH / 8; it++)
      if (sj + 8 * it < wrows) *(uint32_t *)(win + (sj + 8 * it) * LW + 4 * sq) = stg[it];
  }
  __syncthreads();

  // ---- bilinear blend, 4 px -> one 32-bit store ----
  // (sum p*w*32 + 2^14) >> 15 == (sum p*wx*wy + 512) >> 10 with 5-bit fractions; the two
  // horizontal taps of a row are one v_dot4_u32_u8 on an aligned-dword pair (v_alignbyte).
  const int obase = wy0 * LW + wx0;
#pragma unroll
  for (int h = 0; h < 2; h++) {
    const int y = yr + 16 * h;
    if (!col_ok || y >= DMZ_CARD_HEIGHT) continue;
    uint32_t packed = 0;
    if (staged) {
      // every pixel of a tile whose W keeps its sign lies inside the corner window (see the
      // file header); the clamp only keeps the LDS address in range
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const int Xv = P[h][k].X, Yv = P[h][k].Y;
        if (DMZ_WARP_ABLATE == 3) { packed |= (uint32_t)((Xv + Yv) & 255) << (8 * k); continue; }
        int o = (Yv >> 5) * LW + (Xv >> 5) - obase;
        o = imin(imax(o, 0), LW * (LH - 1) - 2);
        const unsigned char *p = win + (o & ~3);
        const uint32_t a0 = *(const uint32_t *)p, a1 = *(const uint32_t *)(p + 4);
        const uint32_t b0 = *(const uint32_t *)(p + LW), b1 = *(const uint32_t *)(p + LW + 4);
        const uint32_t top = __builtin_amdgcn_alignbyte(a1, a0, (uint32_t)o);  // uses o & 3
        const uint32_t bot = __builtin_amdgcn_alignbyte(b1, b0, (uint32_t)o);
        const uint32_t ax = (uint32_t)Xv & 31u, ay = (uint32_t)Yv & 31u;
        const uint32_t wx = (32u - ax) | (ax << 8);
        const uint32_t t_top = __builtin_amdgcn_udot4(top, wx, 0u, false);
        const uint32_t t_bot = __builtin_amdgcn_udot4(bot, wx, 0u, false);
        const uint32_t v = (t_top * (32u - ay) + t_bot * ay + 512u) >> 10;  // <= 255
        packed |= v << (8 * k);
      }
    } else {
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const int Xv = P[h][k].X, Yv = P[h][k].Y;
        const int sx = sat16(Xv >> 5), sy = sat16(Yv >> 5);
        const int ax = Xv & 31, ay = Yv & 31;
        int v0, v1, v2, v3;
        taps_global(src, row_stride, sw, sh, sx, sy, v0, v1, v2, v3);
        const int w00 = (32 - ax) * (32 - ay), w01 = ax * (32 - ay), w10 = (32 - ax) * ay, w11 = ax * ay;
        int v = (v0 * w00 + v1 * w01 + v2 * w10 + v3 * w11 + 512) >> 10;
        v = v > 255 ? 255 : v;
        packed |= (uint32_t)v << (8 * k);
      }
    }
    *(uint32_t *)(dbase + (size_t)y * DMZ_CARD_WIDTH + x + xq) = packed;
  }
}

}  // namespace

void dmz_launch_warp(hipStream_t s, const uint8_t *planes, size_t frame_stride, int row_stride,
                     int width, int height, int n, const DmzWarpMat *mats, uint8_t *cards,
                     size_t card_stride) {
  const int n_pad = (n + 7) & ~7;
  hipLaunchKernelGGL(k_warp, dim3((unsigned)n_pad * kTiles), dim3(256), 0, s, planes, frame_stride,
                     row_stride, width, height, n, n_pad, mats, cards, card_stride);
}
