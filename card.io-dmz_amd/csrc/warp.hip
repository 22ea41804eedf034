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
// CDNA4 mapping: one workgroup per 64 x 90 destination strip (x origin = OpenCV's 64-px block
// origin, the only one that enters the fp64 association; 428 x 270 = 7 x 3 strips).  A lane owns
// one destination column and walks down the strip's rows, so everything that depends on the
// column only (M0*x1, M3*x1, M6*x1) is computed once per lane, and everything that depends on
// the row only (M0*x + M1*y + M2, ...) is computed once per workgroup and broadcast from LDS.
// What is left per pixel is what exactness needs -- the kernel is VALU-issue bound (fp64 and
// integer instructions issue at the same rate on gfx950):
//   * 32/W = 1/(W/32) (the 2^-5 rides, exactly, in the row and column terms): the IEEE-exact
//     v_rcp_f64 + fma sequence the compiler itself emits for 1.0/x, without its v_div_scale /
//     v_div_fixup wrapper (quarter-rate instructions that are the identity for the exponents
//     k_warp_windows admits to the fast path; anything else takes the full `/`).
//   * cvRound: one fp64 add of 1.5*2^52 (round-to-nearest-even by the adder) instead of
//     v_rndne_f64 + v_cvt_i32_f64, with the (even) window origin folded into the constant so
//     the low dword is already window-relative; |f| >= 2^31 takes the saturating conversion.
//   * the strip's source window comes from its four corner pixels (a projective map with W
//     of constant sign is monotone along lines, +-1 px for rounding) and is staged into LDS
//     with aligned 32-bit row loads (zeros outside the image = BORDER_CONSTANT 0); the four
//     taps are single-byte LDS reads (lanes are adjacent columns: conflict-free).
// Strips whose window exceeds the LDS buffer or whose W changes sign (extreme caller-supplied
// matrices) take the direct global path.  Blocks are renumbered so that all strips of a
// frame run on one XCD (block b is dispatched to XCD b % 8) and share its L2.
#include "dmz_hip_internal.h"

namespace {

// developer ablation (tools/ablate.sh): 1 = cheap coordinates, 3 = no blend, 6 = no stores,
// 8 = one row record, 9 = with an LDS address clamp, 10 = one row pair per wave (block overhead)
#ifndef DMZ_WARP_ABLATE
#define DMZ_WARP_ABLATE 0
#endif

constexpr int TW = 64, TH = 90;
constexpr int kTilesX = (DMZ_CARD_WIDTH + TW - 1) / TW;   // 7
constexpr int kTilesY = (DMZ_CARD_HEIGHT + TH - 1) / TH;  // 3
constexpr int kTiles = kTilesX * kTilesY;                 // 21
constexpr int LW = 128;     // LDS window row stride (bytes)
constexpr int LWMAX = 120;  // widest staged window (px): 30 dword columns
#ifndef DMZ_WARP_LH
#define DMZ_WARP_LH 136
#endif
constexpr int LH = DMZ_WARP_LH;  // rows: a 90-row strip at up to 1.47 source px per card px; with the row
                                 // records the workgroup uses 20,288 B of LDS: eight per CU
constexpr int kStagePasses = LH / 8;

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

__device__ __forceinline__ SrcXY map_pixel(double X0, double Y0, double W0, double M0, double M3,
                                           double M6, int x1) {
  SrcXY r;
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

struct RowXYW {
  double X0, Y0, W0, W0s;  // W0s = W0 / 32
};

static_assert(kTiles == DMZ_WARP_STRIPS, "strip count");

// One thread per (frame, strip): the strip's source window from its four corner pixels.
__global__ __launch_bounds__(256) void k_warp_windows(int n, int sw, int sh, int aligned,
                                                       DmzWarpMat *__restrict__ mats) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n * kTiles) return;
  const int frame = i / kTiles, tile = i - frame * kTiles;
  const DmzWarpMat &wm = mats[frame];
  DmzWarpWin w = {0, 0, 0, 0};
  if (wm.valid) {
    const int ty = tile / kTilesX, tx = tile - ty * kTilesX;
    const int x = tx * TW, y0 = ty * TH;
    const double M0 = wm.m[0], M1 = wm.m[1], M2 = wm.m[2], M3 = wm.m[3], M4 = wm.m[4], M5 = wm.m[5],
                 M6 = wm.m[6], M7 = wm.m[7], M8 = wm.m[8];
    int bx0 = 1 << 20, bx1 = -(1 << 20), by0 = 1 << 20, by1 = -(1 << 20), npos = 0, nneg = 0;
    for (int c = 0; c < 4; c++) {
      const int cx1 = (c & 1) ? imin(TW, DMZ_CARD_WIDTH - x) - 1 : 0;
      const int cy = y0 + ((c & 2) ? TH - 1 : 0);
      const double W0 = M6 * x + M7 * cy + M8;
      const double Wc = W0 + M6 * cx1;
      const SrcXY p = map_pixel(M0 * x + M1 * cy + M2, M3 * x + M4 * cy + M5, W0, M0, M3, M6, cx1);
      const int sx = sat16(p.X >> 5), sy = sat16(p.Y >> 5);
      bx0 = imin(bx0, sx), bx1 = imax(bx1, sx), by0 = imin(by0, sy), by1 = imax(by1, sy);
      const int we = (__double2hiint(Wc) >> 20) & 0x7ff;
      const bool tame = we > 900 && we < 1150 && p.X > -(1 << 30) && p.X < (1 << 30) && p.Y > -(1 << 30) &&
                        p.Y < (1 << 30);
      if (tame) (Wc > 0. ? npos : nneg)++;
    }
    bx0 -= 1, bx1 += 1, by0 -= 1, by1 += 1;  // rounding of the interior pixels
    w.wx0 = bx0 & ~3;                         // window origin, 4-aligned in x
    w.wy0 = by0;
    const int wcols = bx1 + 2 - w.wx0;        // + the right bilinear tap
    w.wrows = by1 + 2 - w.wy0;
    const int wdw = (wcols + 3) >> 2;
    if ((npos == 4 || nneg == 4) && wcols <= LWMAX && w.wrows <= LH) {
      const bool interior = aligned && w.wx0 >= 0 && w.wx0 + 4 * wdw <= sw && w.wy0 >= 0 && w.wy0 + w.wrows <= sh;
      w.wdw = interior ? wdw : -wdw;
    }
  }
  mats[frame].win[tile] = w;
}

__global__ __launch_bounds__(256) void k_warp(const uint8_t *__restrict__ planes, size_t frame_stride,
                                               int row_stride, int sw, int sh, int n, int n_pad,
                                               const DmzWarpMat *__restrict__ mats,
                                               uint8_t *__restrict__ cards, size_t card_stride) {
  __shared__ __attribute__((aligned(16))) unsigned char win[LW * LH];
  __shared__ __attribute__((aligned(16))) RowXYW s_row[TH];

  // XCD-aware renumbering: logical id = xcd * (blocks/8) + k
  const unsigned int nblk = (unsigned int)n_pad * kTiles;
  const unsigned int b = blockIdx.x;
  const unsigned int logical = (b & 7u) * (nblk >> 3) + (b >> 3);
  const int frame = (int)(logical / kTiles);
  const int tile = (int)(logical - (unsigned int)frame * kTiles);
  if (frame >= n) return;
  const int ty = tile / kTilesX, tx = tile - ty * kTilesX;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int x = tx * TW;   // OpenCV block origin in x
  const int y0 = ty * TH;  // 270 == 3 * TH: every strip is full height
  uint8_t *dbase = cards + (size_t)frame * card_stride;
  const DmzWarpMat &wm = mats[frame];
  if (!wm.valid) {
    const bool col_ok = x + lane < DMZ_CARD_WIDTH;
    if (col_ok)
      for (int j = wave; j < TH; j += 4) dbase[(y0 + j) * DMZ_CARD_WIDTH + x + lane] = 0;
    return;
  }
  const double M0 = wm.m[0], M1 = wm.m[1], M2 = wm.m[2], M3 = wm.m[3], M4 = wm.m[4], M5 = wm.m[5],
               M6 = wm.m[6], M7 = wm.m[7], M8 = wm.m[8];
  const DmzWarpWin ww = wm.win[tile];  // uniform: scalar loads
  const int wx0 = ww.wx0, wy0 = ww.wy0, wrows = ww.wrows;
  const uint8_t *src = planes + (size_t)frame * frame_stride;

  // ---- per-row terms, one thread per row ----
  if (tid < TH) {
    const int y = y0 + tid;
    RowXYW r;
    r.X0 = M0 * x + M1 * y + M2;
    r.Y0 = M3 * x + M4 * y + M5;
    r.W0 = M6 * x + M7 * y + M8;
    r.W0s = r.W0 * 0.03125;  // exact: the fast path divides by W / 32
    s_row[tid] = r;
  }

  if (ww.wdw == 0) {
    // ---- generic path: range-checked coordinates, taps straight from global memory ----
    __syncthreads();
    for (int i = tid; i < TW * TH; i += 256) {
      const int x1 = i & (TW - 1), j = i >> 6;
      if (x + x1 >= DMZ_CARD_WIDTH) continue;
      const RowXYW r = s_row[j];
      const SrcXY p = map_pixel(r.X0, r.Y0, r.W0, M0, M3, M6, x1);
      const int sx = sat16(p.X >> 5), sy = sat16(p.Y >> 5);
      const int ax = p.X & 31, ay = p.Y & 31;
      int v0, v1, v2, v3;
      taps_global(src, row_stride, sw, sh, sx, sy, v0, v1, v2, v3);
      const int w00 = (32 - ax) * (32 - ay), w01 = ax * (32 - ay), w10 = (32 - ax) * ay, w11 = ax * ay;
      int v = (v0 * w00 + v1 * w01 + v2 * w10 + v3 * w11 + 512) >> 10;
      v = v > 255 ? 255 : v;
      dbase[(y0 + j) * DMZ_CARD_WIDTH + x + x1] = (uint8_t)v;
    }
    return;
  }

  // ---- stage the window: thread (tid & 31) owns one dword column, 8 rows per pass.  For a
  // window that lies inside the image (every card that is inside the frame) these are buffer
  // loads -- frame descriptor + per-thread offset + scalar row offset, no address arithmetic,
  // rows past the frame read as zero -- all in flight together: twelve passes cover the 96 rows
  // of a strip at scale 1, taller windows take six more.  Rows between the window and the pass
  // boundary are staged too (never read).  Windows that cross the image border are staged by
  // the byte-checked loop.
  const int sq = tid & 31, sj = tid >> 5;
  if (ww.wdw > 0) {
    if (sq < ww.wdw) {
      const size_t fbytes = (size_t)row_stride * sh;
      const __amdgpu_buffer_rsrc_t frame_rs = __builtin_amdgcn_make_buffer_rsrc(
          (void *)src, 0, fbytes > 0xfffffffcu ? 0xfffffffcu : (unsigned)fbytes, 0x00020000);
      const int voff = (wy0 + sj) * row_stride + wx0 + 4 * sq;
      unsigned char *lp = win + sj * LW + 4 * sq;
      constexpr int kA = 12;
      uint32_t sa[kA], sb[kStagePasses - kA];
#pragma unroll
      for (int it = 0; it < kA; it++) sa[it] = __builtin_amdgcn_raw_buffer_load_b32(frame_rs, voff, 8 * it * row_stride, 0);
      if (wrows > 8 * kA) {
#pragma unroll
        for (int it = kA; it < kStagePasses; it++)
          sb[it - kA] = __builtin_amdgcn_raw_buffer_load_b32(frame_rs, voff, 8 * it * row_stride, 0);
      }
#pragma unroll
      for (int it = 0; it < kA; it++) *(uint32_t *)(lp + 8 * it * LW) = sa[it];
      if (wrows > 8 * kA) {
#pragma unroll
        for (int it = kA; it < kStagePasses; it++) *(uint32_t *)(lp + 8 * it * LW) = sb[it - kA];
      }
    }
  } else {
    const int wdw = -ww.wdw;
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

  // column terms of this lane, and the rounding constants with the window origin folded in:
  // 1.5 * 2^52 - 32 * origin is an even integer, so the adder still rounds the product to the
  // nearest-even integer and the low dword is the window-relative fixed-point coordinate.
  // The W terms carry a factor 2^-5 (exact), so that 32 / W is the reciprocal of their sum.
  // Lanes beyond column 427 repeat column 427 (same value to the same byte), which keeps the
  // loop free of predication.
  const int x1 = imin(lane, DMZ_CARD_WIDTH - 1 - x);
  const double A = M0 * x1, B = M3 * x1, C = (M6 * x1) * 0.03125;
  const double magicX = 6755399441055744.0 - (double)(32 * wx0);
  const double magicY = 6755399441055744.0 - (double)(32 * wy0);
  __syncthreads();

  // every pixel of a strip whose W keeps its sign lies inside the corner window (file header);
  // the clamp only keeps the LDS address in range
  auto pixel = [&](int j) -> uint32_t {
    RowXYW r = s_row[DMZ_WARP_ABLATE == 8 ? 0 : j];
    if (DMZ_WARP_ABLATE == 8) r.X0 += j, r.Y0 += j;
    int Xv, Yv;
    if (DMZ_WARP_ABLATE == 1) {
      Xv = ((int)(float)r.X0 + x1 - wx0) * 32;
      Yv = ((int)(float)r.Y0 - wy0) * 32;
    } else {
      // 1 / Wd, correctly rounded: v_rcp_f64 and three Newton steps, the last one in the
      // residual form (the sequence the compiler emits for 1.0 / x between its v_div_scale /
      // v_div_fixup wrapper, which is the identity for these exponents)
      const double Wd = r.W0s + C;
      double yv = __builtin_amdgcn_rcp(Wd);
      double t = __builtin_fma(-Wd, yv, 1.0);
      yv = __builtin_fma(yv, t, yv);
      t = __builtin_fma(-Wd, yv, 1.0);
      yv = __builtin_fma(yv, t, yv);
      t = __builtin_fma(-Wd, yv, 1.0);
      const double W = __builtin_fma(t, yv, yv);
      Xv = __double2loint((r.X0 + A) * W + magicX);
      Yv = __double2loint((r.Y0 + B) * W + magicY);
    }
    if (DMZ_WARP_ABLATE == 3) return (uint32_t)(Xv + Yv) & 255u;
    int o = ((Yv << 2) & ~(LW - 1)) | (Xv >> 5);  // (Yv >> 5) * LW + (Xv >> 5) inside the window
    if (DMZ_WARP_ABLATE == 9) o = imin(imax(o, 0), LW * (LH - 1) - 2);
    // four single-byte LDS reads: odd-address ds_read_u16 is several times slower than the
    // aligned form on gfx950, and a byte read costs no more than a wider one here
    // (the right-hand taps are volatile reads only to keep the compiler from re-merging them)
    typedef const volatile __attribute__((address_space(3))) unsigned char *lds_vu8;
    const unsigned char *p = win + o;
    const lds_vu8 pr = (lds_vu8)win + o;
    const int p00 = p[0], p01 = pr[1], p10 = p[LW], p11 = pr[LW + 1];
    const int ax = Xv & 31, ay = Yv & 31, bx = 32 - ax;
    // (sum p*w*32 + 2^14) >> 15 == (sum p*wx*wy + 512) >> 10 with 5-bit fractions
    const int t_top = __mul24(p01, ax) + __mul24(p00, bx);
    const int t_bot = __mul24(p11, ax) + __mul24(p10, bx);
    return (uint32_t)((t_top << 5) + (__mul24(t_bot - t_top, ay) + 512)) >> 10;  // <= 255
  };

  // ---- the strip: wave w takes rows w, w + 4, ..., two at a time (independent chains) ----
  const int wave_s = __builtin_amdgcn_readfirstlane(wave);
  // buffer stores: card descriptor + scalar row offset + lane column, no per-pixel address math
  const __amdgpu_buffer_rsrc_t card =
      __builtin_amdgcn_make_buffer_rsrc(dbase, 0, DMZ_CARD_WIDTH * DMZ_CARD_HEIGHT, 0x00020000);
  const int tile_off = y0 * DMZ_CARD_WIDTH + x;
  constexpr int kPairs = TH / 8;  // 11 pairs = rows w .. w + 84; rows 88, 89 are the tail
#pragma unroll
  for (int m = 0; m < (DMZ_WARP_ABLATE == 10 ? 1 : kPairs); m++) {
    const int j0 = wave_s + 8 * m, j1 = j0 + 4;
    const uint32_t v0 = pixel(j0), v1 = pixel(j1);
    if (DMZ_WARP_ABLATE == 6 && v0 + v1 != 0x12345u) continue;
    __builtin_amdgcn_raw_buffer_store_b8((uint8_t)v0, card, x1, tile_off + j0 * DMZ_CARD_WIDTH, 0);
    __builtin_amdgcn_raw_buffer_store_b8((uint8_t)v1, card, x1, tile_off + j1 * DMZ_CARD_WIDTH, 0);
  }
  static_assert(TH == 8 * kPairs + 2, "tail rows");
  if (wave_s < 2) {
    const int j = wave_s + 8 * kPairs;
    __builtin_amdgcn_raw_buffer_store_b8((uint8_t)pixel(j), card, x1, tile_off + j * DMZ_CARD_WIDTH, 0);
  }
}

}  // namespace

void dmz_launch_warp(hipStream_t s, const uint8_t *planes, size_t frame_stride, int row_stride,
                     int width, int height, int n, DmzWarpMat *mats, uint8_t *cards,
                     size_t card_stride) {
  const int n_pad = (n + 7) & ~7;
  const int aligned = ((((uintptr_t)planes) | (uintptr_t)frame_stride | (uintptr_t)row_stride) & 3) == 0;
  hipLaunchKernelGGL(k_warp_windows, dim3((unsigned)((n * kTiles + 255) / 256)), dim3(256), 0, s, n, width,
                     height, aligned, mats);
  hipLaunchKernelGGL(k_warp, dim3((unsigned)n_pad * kTiles), dim3(256), 0, s, planes, frame_stride,
                     row_stride, width, height, n, n_pad, mats, cards, card_stride);
}
