// warp.hip -- perspective rectification of the card to 428 x 270.
//
// Replaces, for a whole batch, llcv_unwarp's CPU branch (cv/warp.cpp:153-166):
// cvWarpPerspective(src, dst, M, CV_INTER_LINEAR + CV_WARP_FILL_OUTLIERS, 0) with the
// OpenCV 2.4 semantics restated in SURVEY.md Appendix A10: per destination pixel in
// 64-wide blocks  X0 = M0*x + M1*y + M2 (fp64), then (X0 + M0*x1) * (32 / W), cvRound
// to 1/32 px, 5-bit fractions, (sum p*w + 2^14) >> 15 bilinear blend, zero outside the
// source.  fp64 operations are single IEEE operations in that order (-ffp-contract=off),
// so the card is byte-exact.
//
// CDNA4 mapping: one workgroup per 64 x 32 destination tile (two of OpenCV's 64 x 16
// blocks; the x block origin -- the only one that enters the fp64 association -- is the
// same).  Each thread first computes the fixed-point source coordinates of its 8 pixels;
// a wave/LDS reduction gives the tile's exact source bounding box, which is staged into
// LDS with aligned 32-bit row loads (zero outside the image = BORDER_CONSTANT 0), and
// the four bilinear taps per pixel become LDS byte reads instead of 4 scattered global
// byte loads (the v1 kernel was bound by vector-memory address processing).  Tiles
// whose bounding box exceeds the LDS window (extreme, caller-supplied matrices) take
// the direct global path.  Blocks are renumbered so that all tiles of a frame run on one
// XCD (block b is dispatched to XCD b % 8) and share its L2.
#include "dmz_hip_internal.h"

namespace {

constexpr int TW = 64, TH = 32;
constexpr int kTilesX = (DMZ_CARD_WIDTH + TW - 1) / TW;   // 7
constexpr int kTilesY = (DMZ_CARD_HEIGHT + TH - 1) / TH;  // 9
constexpr int kTiles = kTilesX * kTilesY;                 // 63
constexpr int LW = 112;  // LDS window: bytes per row (28 dwords)
constexpr int LH = 64;   // rows

__device__ __forceinline__ int sat16(int v) { return v < -32768 ? -32768 : (v > 32767 ? 32767 : v); }
__device__ __forceinline__ int imin(int a, int b) { return a < b ? a : b; }
__device__ __forceinline__ int imax(int a, int b) { return a > b ? a : b; }

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
  __shared__ int s_box[4];  // min sx, max sx, min sy, max sy

  // XCD-aware renumbering: logical id = xcd * (blocks/8) + k
  const unsigned int nblk = (unsigned int)n_pad * kTiles;
  const unsigned int b = blockIdx.x;
  const unsigned int logical = (b & 7u) * (nblk >> 3) + (b >> 3);
  const int frame = (int)(logical / kTiles);
  const int tile = (int)(logical - (unsigned int)frame * kTiles);
  if (frame >= n) return;
  const int ty = tile / kTilesX, tx = tile - ty * kTilesX;
  const int tid = threadIdx.x, lane = tid & 63;
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

  // ---- fixed-point source coordinates of this thread's 2 x 4 pixels ----
  int X[2][4], Y[2][4];
  int bx0 = 1 << 30, bx1 = -(1 << 30), by0 = 1 << 30, by1 = -(1 << 30);
#pragma unroll
  for (int h = 0; h < 2; h++) {
    const int y = yr + 16 * h;
    const double X0 = M0 * x + M1 * y + M2;
    const double Y0 = M3 * x + M4 * y + M5;
    const double W0 = M6 * x + M7 * y + M8;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int x1 = xq + k;
      double W = W0 + M6 * x1;
      W = W ? 32. / W : 0;
      double fX = (X0 + M0 * x1) * W;
      double fY = (Y0 + M3 * x1) * W;
      // OpenCV clamps to [INT_MIN, INT_MAX] before cvRound; v_cvt_i32_f64 saturates the same way
      X[h][k] = __double2int_rn(fX);
      Y[h][k] = __double2int_rn(fY);
      if (col_ok && y < DMZ_CARD_HEIGHT) {
        const int sx = sat16(X[h][k] >> 5), sy = sat16(Y[h][k] >> 5);
        bx0 = imin(bx0, sx); bx1 = imax(bx1, sx);
        by0 = imin(by0, sy); by1 = imax(by1, sy);
      }
    }
  }
  // ---- tile bounding box ----
  if (tid == 0) { s_box[0] = 1 << 30; s_box[1] = -(1 << 30); s_box[2] = 1 << 30; s_box[3] = -(1 << 30); }
  for (int o = 32; o > 0; o >>= 1) {
    bx0 = imin(bx0, __shfl_xor(bx0, o, 64)); bx1 = imax(bx1, __shfl_xor(bx1, o, 64));
    by0 = imin(by0, __shfl_xor(by0, o, 64)); by1 = imax(by1, __shfl_xor(by1, o, 64));
  }
  __syncthreads();
  if (lane == 0) {
    atomicMin(&s_box[0], bx0); atomicMax(&s_box[1], bx1);
    atomicMin(&s_box[2], by0); atomicMax(&s_box[3], by1);
  }
  __syncthreads();
  const int wx0 = s_box[0] & ~3;          // window origin, 4-aligned in x
  const int wy0 = s_box[2];
  const int wcols = s_box[1] + 2 - wx0;   // + the right bilinear tap
  const int wrows = s_box[3] + 2 - wy0;
  const uint8_t *src = planes + (size_t)frame * frame_stride;
  const bool staged = wcols <= LW && wrows <= LH;

  if (staged) {
    // ---- stage the window: aligned dwords where the whole word is inside the image ----
    const int wdw = (wcols + 3) >> 2;
    const bool aligned = ((((uintptr_t)src) | (uintptr_t)row_stride) & 3) == 0;
    for (int i = tid; i < wdw * wrows; i += 256) {
      const int j = i / wdw, q = i - j * wdw;
      const int gy = wy0 + j, gx = wx0 + 4 * q;
      uint32_t v = 0u;
      if (gy >= 0 && gy < sh) {
        const uint8_t *g = src + (size_t)gy * row_stride + gx;
        if (aligned && gx >= 0 && gx + 3 < sw) {
          v = *(const uint32_t *)g;
        } else {
#pragma unroll
          for (int k = 0; k < 4; k++)
            if (gx + k >= 0 && gx + k < sw) v |= (uint32_t)g[k] << (8 * k);
        }
      }
      *(uint32_t *)(win + j * LW + 4 * q) = v;
    }
    __syncthreads();
  }

  // ---- bilinear blend, 4 px -> one 32-bit store ----
  // (sum p*w*32 + 2^14) >> 15 == (sum p*wx*wy + 512) >> 10 with 5-bit fractions; the two
  // horizontal taps of a row are one v_dot4_u32_u8 on an aligned-dword pair (v_alignbyte).
  const uint32_t *win32 = (const uint32_t *)win;
#pragma unroll
  for (int h = 0; h < 2; h++) {
    const int y = yr + 16 * h;
    if (!col_ok || y >= DMZ_CARD_HEIGHT) continue;
    uint32_t packed = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int Xv = X[h][k], Yv = Y[h][k];
      const int sx = sat16(Xv >> 5), sy = sat16(Yv >> 5);
      const int ax = Xv & 31, ay = Yv & 31;
      int v;
      if (staged) {
        const int o = (sy - wy0) * LW + (sx - wx0);
        const int di = o >> 2, sh = o & 3;
        const uint32_t top = __builtin_amdgcn_alignbyte(win32[di + 1], win32[di], sh);
        const uint32_t bot = __builtin_amdgcn_alignbyte(win32[di + LW / 4 + 1], win32[di + LW / 4], sh);
        const uint32_t wx = (uint32_t)(32 - ax) | ((uint32_t)ax << 8);
        const int t_top = (int)__builtin_amdgcn_udot4(top, wx, 0u, false);
        const int t_bot = (int)__builtin_amdgcn_udot4(bot, wx, 0u, false);
        v = (t_top * (32 - ay) + t_bot * ay + 512) >> 10;
      } else {
        int v0, v1, v2, v3;
        taps_global(src, row_stride, sw, sh, sx, sy, v0, v1, v2, v3);
        const int w00 = (32 - ax) * (32 - ay), w01 = ax * (32 - ay), w10 = (32 - ax) * ay, w11 = ax * ay;
        v = (v0 * w00 + v1 * w01 + v2 * w10 + v3 * w11 + 512) >> 10;
      }
      v = v > 255 ? 255 : v;
      packed |= (uint32_t)v << (8 * k);
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
