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
// column only (M0*x1, M3*x1, M6*x1) is computed once per lane; what depends on the row only
// (M0*x + M1*y + M2, ...) the filtered path advances from the strip's first row (k_warp_windows)
// and the exact sequence evaluates where it runs.
// What is left per pixel is what exactness needs -- the kernel is VALU-issue bound (fp64 and most
// integer instructions issue at about the same rate on gfx950).  These are the EXACT sequences; since round 2 nearly every
// pixel takes the filtered-exact coordinates further down (cheap coordinates wherever they provably round like the exact
// ones: since round 5 as an affine function of an extrapolated reciprocal, 6 fp64 operations per pixel):
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
//     taps are single-byte LDS reads.  Lanes are adjacent destination columns, so their taps fall into a handful of
//     neighbouring dwords -- but a row of taps straddles 128-byte window rows wherever the quad is tilted: the PMC passes
//     show SQ_LDS_BANK_CONFLICT = 0.31 of the kernel's active LDS cycles (profiles/r2_v8, unchanged in r3), i.e. ~3 % of its
//     busy cycles; the kernel is VALU-issue bound and the LDS pipe has slack, so the layout stays.
// Strips whose window exceeds the LDS buffer or whose W changes sign (extreme caller-supplied
// matrices) take the direct global path.  Blocks are renumbered so that all strips of a
// frame run on one XCD (block b is dispatched to XCD b % 8) and share its L2.
#include <type_traits>

#include "dmz_hip_internal.h"

namespace {

// developer ablation (tools/ablate.sh): 1 = cheap coordinates, 3 = no blend, 6 = no stores,
// 9 = with an LDS address clamp, 10 = one row pair per wave (block overhead)
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
constexpr int LH = DMZ_WARP_LH;  // rows: a 90-row strip at up to 1.47 source px per card px: 17,408 B of LDS, nine
                                 // workgroups per CU (round 5: no table of row terms beside the window)
// waves per strip: each walks TH / kWaves rows of the strip's 64 columns.  Per-wave set-up (column terms, the start
// of the reciprocal chains) is ~13 % of a 23-row walk.  Three waves of 30 rows cover the 90 rows exactly (no row twice,
// no odd-row tail): 7 % fewer instructions at 24 instead of 32 waves per CU, measured 1 - 5 % faster than four waves
// of 23; two waves of 45 rows (16 waves per CU) are 4 % slower.
#ifndef DMZ_WARP_WAVES
#define DMZ_WARP_WAVES 3
#endif
constexpr int kWaves = DMZ_WARP_WAVES, kThreads = 64 * kWaves;
constexpr int kStageRows = kThreads / 32;  // window rows staged per pass (a thread = one dword column)
constexpr int kStagePasses = (LH + kStageRows - 1) / kStageRows;  // (a partial last pass is guarded)
constexpr int kFastFlag = 1 << 16;  // DmzWarpWin.wrows: the strip admits the extrapolated reciprocal (k_warp)
constexpr int kAffFlag = 1 << 17;   // ... and the coordinates as an affine function of that reciprocal (k_warp)
constexpr int kLinFlag = 1 << 18;   // ... or (hardly any perspective down a column) the reciprocal itself is linear in the row
// developer A/B (tools/ab.sh): 0 = the numerators advance by recurrence (rounds 2 - 4), 1 = affine in the reciprocal
#ifndef DMZ_WARP_AFFINE
#define DMZ_WARP_AFFINE 1
#endif
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

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

#ifdef DMZ_DEV_SELFCHECK
__device__ unsigned long long g_dev_selfcheck_warp[2];  // windows evaluated twice, pairs that differed
#endif
// One thread per (frame, strip): the strip's source window from its four corner pixels.
__global__ __launch_bounds__(256) void k_warp_windows(int n, int sw, int sh, int aligned,
                                                       DmzWarpMat *__restrict__ mats) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n * kTiles) return;
  const int frame = i / kTiles, tile = i - frame * kTiles;
  const DmzWarpMat &wm = mats[frame];
#ifdef DMZ_DEV_SELFCHECK  /* developer probe (tools/dev/homography_fault.sh selfcheck): the strip's window computed twice and compared */
  DmzWarpWin w_first = {0, 0, 0, 0, 0., 0., 0., 0., 0., 0., 0., 0., 0.};
  for (int pass = 0; pass < 2; pass++) {
#endif
  DmzWarpWin w = {0, 0, 0, 0, 0., 0., 0., 0., 0., 0., 0., 0., 0.};
  if (wm.valid) {
    const int ty = tile / kTilesX, tx = tile - ty * kTilesX;
    const int x = tx * TW, y0 = ty * TH;
    double M0 = wm.m[0], M1 = wm.m[1], M2 = wm.m[2], M3 = wm.m[3], M4 = wm.m[4], M5 = wm.m[5],
           M6 = wm.m[6], M7 = wm.m[7], M8 = wm.m[8];
#ifdef DMZ_DEV_SELFCHECK
    asm volatile("" : "+v"(M0), "+v"(M1), "+v"(M2), "+v"(M3), "+v"(M4), "+v"(M5), "+v"(M6), "+v"(M7), "+v"(M8));
#endif
    const double sW = M7 * 0.03125, ax = M1 / sW, ay = M4 / sW;
    if (tile == 0) {  // (frame-uniform)
      mats[frame].sw = sW;
      mats[frame].dw2 = 2.0 * sW;
    }
    bool affine = false;
    int bx0 = 1 << 20, bx1 = -(1 << 20), by0 = 1 << 20, by1 = -(1 << 20), npos = 0, nneg = 0;
    double wmin = 1e300;  // min |W| over the strip: W is linear, so it is at a corner
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
      wmin = fmin(wmin, fabs(Wc));
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
      // relative change of W per card row rho <= 2^-11: the reciprocal may be extrapolated along rows (k_warp's fast forms)
      if (fabs(M7) * 2048.0 <= wmin) {
#if DMZ_WARP_AFFINE
        // Down a column X = X_a + M1 dj and Wd = Wd_a + sW dj (sW = M7 / 32), so X / Wd = alpha + (X_a - alpha Wd_a) / Wd
        // with alpha = M1 / sW: the numerators need no recurrence.  The price: an error eps of the reciprocal enters as
        // |alpha| eps instead of |X / Wd| eps, so |alpha| is bounded (k_warp's error budget).  Where there is hardly any
        // perspective down a column (rho <= 2^-24; M7 == 0 included) the reciprocal itself is linear in the row to
        // (29 rho)^2 <= 2^-38 instead.  The two cover every fast strip: |alpha| = 32 |M / W| / rho with 32 |M / W| <= 64
        // (the window bounds the slope) is <= 2^30 wherever rho > 2^-24; a strip that fails both takes the exact loop.
        // The budget is checked here, per strip, not assumed: an error eps of the reciprocal enters the cheap
        // coordinate as (|fX| + |alpha|) eps with |fX| the ABSOLUTE source coordinate in 1/32 px (the window origin is folded
        // into the rounding constant, not into the product) -- the strip's corners bound it -- and the filter needs the sum
        // below 2^-17: eps = 225 r^4 for the extrapolated reciprocal, (29 r)^2 for the linear one, r = |M7| / min |W| >= rho.
        // A wide plane (source x of 8 000 px and more) near rho = 2^-11 fails the affine test and takes the exact loop.
        const double r = fabs(M7) / wmin, r2 = r * r;
        const double fmaxabs = 32.0 * (double)(imax(imax(-bx0, bx1), imax(-by0, by1)) + 2);
        const double amax = fmax(fabs(ax), fabs(ay));
        if (fabs(M7) * 16777216.0 <= wmin && fmaxabs * 841.0 * r2 <= 0x1p-19)
          w.wrows |= kFastFlag | kLinFlag;
        else if (amax <= 1073741824.0 && (amax + fmaxabs) * 225.0 * r2 * r2 <= 0x1p-19)  // |alpha| <= 2^30 (NaN fails)
          w.wrows |= kFastFlag | kAffFlag, affine = true;
#else
        w.wrows |= kFastFlag;
#endif
      }
    }
    // The rounding constants with the window origin folded in (k_warp's comments): 1.5 * 2^52 - 32 * origin for the exact
    // sequence, 1.5 * 2^36 + 0.5 - 32 * origin for the filtered one -- there plus alpha where the strip takes the affine form,
    // and that alpha as the adder rounded it ((magic' + alpha) - magic', exact).
    w.mx = 6755399441055744.0 - (double)(32 * w.wx0);
    w.my = 6755399441055744.0 - (double)(32 * w.wy0);
    const double mxf = 103079215104.0 + 0.5 - (double)(32 * w.wx0), myf = 103079215104.0 + 0.5 - (double)(32 * w.wy0);
    w.kx = affine ? mxf + ax : mxf;
    w.ky = affine ? myf + ay : myf;
    w.aqx = w.kx - mxf;
    w.aqy = w.ky - myf;
    w.rx0 = M0 * x + M1 * y0 + M2;
    w.ry0 = M3 * x + M4 * y0 + M5;
    w.rw0s = (M6 * x + M7 * y0 + M8) * 0.03125;
  }
#ifdef DMZ_DEV_SELFCHECK
  if (pass == 0) {
    w_first = w;
    continue;
  }
  {
    const unsigned *a = (const unsigned *)&w, *b = (const unsigned *)&w_first;
    bool same = true;
    for (int i = 0; i < (int)(sizeof(DmzWarpWin) / 4); i++) same = same && a[i] == b[i];
    atomicAdd(&g_dev_selfcheck_warp[0], 1ull);
    if (!same) atomicAdd(&g_dev_selfcheck_warp[1], 1ull);
  }
#endif
  mats[frame].win[tile] = w;
#ifdef DMZ_DEV_SELFCHECK
  }
#endif
}

__global__ __launch_bounds__(kThreads) void k_warp(const uint8_t *__restrict__ planes, size_t frame_stride,
                                               int row_stride, int sw, int sh, int n, int n_pad,
                                               const DmzWarpMat *__restrict__ mats,
                                               uint8_t *__restrict__ cards, size_t card_stride) {
  __shared__ __attribute__((aligned(16))) unsigned char win[LW * LH];

  // XCD-aware renumbering: logical id = xcd * (blocks/8) + k
  const unsigned int nblk = (unsigned int)n_pad * kTiles;
  const unsigned int b = blockIdx.x;
  const unsigned int logical = (b & 7u) * (nblk >> 3) + (b >> 3);
  const int frame = (int)(logical / kTiles);
  const int tile = (int)(logical - (unsigned int)frame * kTiles);
  if (frame >= n) return;
  const int ty = tile / kTilesX, tx = tile - ty * kTilesX;
#ifdef DMZ_DEV_WARP_SKIP_LAST  /* developer probe (timing only): the seventh strip column (44 of 64 lanes) is not produced */
  if (tx == kTilesX - 1) return;
#endif
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int x = tx * TW;   // OpenCV block origin in x
  const int y0 = ty * TH;  // 270 == 3 * TH: every strip is full height
  uint8_t *dbase = cards + (size_t)frame * card_stride;
  const DmzWarpMat &wm = mats[frame];
  if (!wm.valid) {
    const bool col_ok = x + lane < DMZ_CARD_WIDTH;
    if (col_ok)
      for (int j = wave; j < TH; j += kWaves) dbase[(y0 + j) * DMZ_CARD_WIDTH + x + lane] = 0;
    return;
  }
  const double M0 = wm.m[0], M1 = wm.m[1], M2 = wm.m[2], M3 = wm.m[3], M4 = wm.m[4], M5 = wm.m[5],
               M6 = wm.m[6], M7 = wm.m[7], M8 = wm.m[8];
  const DmzWarpWin ww = wm.win[tile];  // uniform: scalar loads
  const int wx0 = ww.wx0, wy0 = ww.wy0, wrows = ww.wrows & (kFastFlag - 1);
  const uint8_t *src = planes + (size_t)frame * frame_stride;

  // The row terms M0 x + M1 y + M2 (...) of strip row j, in the reference's association: the exact sequence evaluates them
  // where it runs (the generic path, the strips without the fast flag, the filter's rare fallback).  Rounds 1 - 4 kept a
  // table of them in LDS (2 880 B of the workgroup's 20 288: eight workgroups per CU; round 5: 17 408, nine).
  auto row_terms = [&](int j) {
    const int y = y0 + j;
    RowXYW r;
    r.X0 = M0 * x + M1 * y + M2;
    r.Y0 = M3 * x + M4 * y + M5;
    r.W0 = M6 * x + M7 * y + M8;
    r.W0s = r.W0 * 0.03125;  // exact: the fast path divides by W / 32
    return r;
  };

  if (ww.wdw == 0) {
    // ---- generic path: range-checked coordinates, taps straight from global memory ----
    __syncthreads();
    for (int i = tid; i < TW * TH; i += kThreads) {
      const int x1 = i & (TW - 1), j = i >> 6;
      if (x + x1 >= DMZ_CARD_WIDTH) continue;
      const RowXYW r = row_terms(j);
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
      constexpr int kA = 96 / kStageRows;  // the 96 rows of a strip at scale 1
      uint32_t sa[kA], sb[kStagePasses - kA];
#pragma unroll
      for (int it = 0; it < kA; it++)
        sa[it] = __builtin_amdgcn_raw_buffer_load_b32(frame_rs, voff, kStageRows * it * row_stride, 0);
      if (wrows > kStageRows * kA) {
#pragma unroll
        for (int it = kA; it < kStagePasses; it++)
          sb[it - kA] = __builtin_amdgcn_raw_buffer_load_b32(frame_rs, voff, kStageRows * it * row_stride, 0);
      }
#pragma unroll
      for (int it = 0; it < kA; it++) *(uint32_t *)(lp + kStageRows * it * LW) = sa[it];
      if (wrows > kStageRows * kA) {
#pragma unroll
        for (int it = kA; it < kStagePasses; it++)
          if (LH % kStageRows == 0 || it < kStagePasses - 1 || sj + kStageRows * it < LH)
            *(uint32_t *)(lp + kStageRows * it * LW) = sb[it - kA];
      }
    }
  } else {
    const int wdw = -ww.wdw;
    for (int i = tid; i < wdw * wrows; i += kThreads) {
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
  const double magicX = ww.mx, magicY = ww.my;  // (uniform, from k_warp_windows: scalar loads)
  __syncthreads();

  // Coordinates travel as "l-format" dwords: window-relative fixed point with 16 + 5 fractional
  // bits (pixel index from bit 21, the 5-bit bilinear fraction in bits 16..20, bits 0..15 below
  // the rounding position).  The exact path shifts its rounded integer up by 16.
  //
  // every pixel of a strip whose W keeps its sign lies inside the corner window (file header)
  auto exact_xy = [&](const RowXYW &r, uint32_t &Xl, uint32_t &Yl) {
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
    Xl = (uint32_t)__double2loint((r.X0 + A) * W + magicX) << 16;
    Yl = (uint32_t)__double2loint((r.Y0 + B) * W + magicY) << 16;
  };

  // bilinear blend of two pixels.  The four taps of a pixel are single-byte LDS reads (odd-address
  // ds_read_u16 is several times slower than the aligned form on gfx950), packed as the pairs
  // {p00, p10} and {p01, p11} in two registers: the horizontal
  // lerp of the top and the bottom row is then ONE 32-bit multiply-add over both halves
  // (t = 32 p0 + (p1 - p0) ax <= 8160: no carry between the halves), and the vertical lerp with
  // rounding one v_dot2_u32_u16 against {(32 - ay) 64, ay 64} + 2^15, which leaves
  // (sum p w + 2^14) >> 15 == (sum p wx wy + 512) >> 10 in bits 16..23: the byte a d16_hi store writes.
  uint32_t k2048;  // kept in a register: v_mad_u32_u24 can take one literal only
  asm("v_mov_b32 %0, 0x800" : "=v"(k2048));
  auto blend2 = [&](uint32_t Xa, uint32_t Ya, uint32_t Xb, uint32_t Yb, uint32_t &va, uint32_t &vb) {
    const uint32_t oa = ((Ya >> 14) & ~(uint32_t)(LW - 1)) | (Xa >> 21);  // (Y >> 21) * LW + (X >> 21)
    const uint32_t ob = ((Yb >> 14) & ~(uint32_t)(LW - 1)) | (Xb >> 21);
    const uint32_t la = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) unsigned char *)win + oa;
    const uint32_t lb = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) unsigned char *)win + ob;
    // (inline asm: the compiler would merge the two horizontally adjacent byte reads into one 16-bit
    // read at an odd address; the d16 / d16_hi load forms that would deliver the pairs packed zero the
    // other register half on this chip (SRAM-ECC), so the pairs cost one v_lshl_or_b32 each)
    uint32_t a00, a01, a10, a11, b00, b01, b10, b11;
    asm volatile(
        "ds_read_u8 %0, %8\n\t"
        "ds_read_u8 %1, %8 offset:1\n\t"
        "ds_read_u8 %2, %8 offset:%10\n\t"
        "ds_read_u8 %3, %8 offset:%11\n\t"
        "ds_read_u8 %4, %9\n\t"
        "ds_read_u8 %5, %9 offset:1\n\t"
        "ds_read_u8 %6, %9 offset:%10\n\t"
        "ds_read_u8 %7, %9 offset:%11"
        : "=&v"(a00), "=&v"(a01), "=&v"(a10), "=&v"(a11), "=&v"(b00), "=&v"(b01), "=&v"(b10), "=&v"(b11)
        : "v"(la), "v"(lb), "n"(LW), "n"(LW + 1));
    const uint32_t axa = (Xa >> 16) & 31u, aya = (Ya >> 16) & 31u;
    const uint32_t axb = (Xb >> 16) & 31u, ayb = (Yb >> 16) & 31u;
    const uint32_t wya = aya * (65535u * 64u) + k2048;  // {(32 - ay) * 64, ay * 64}
    const uint32_t wyb = ayb * (65535u * 64u) + k2048;
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(a00), "+v"(a01), "+v"(a10), "+v"(a11), "+v"(b00), "+v"(b01), "+v"(b10), "+v"(b11));
    const uint32_t p0a = (a10 << 16) | a00, p1a = (a11 << 16) | a01;  // {p00, p10}, {p01, p11}
    const uint32_t p0b = (b10 << 16) | b00, p1b = (b11 << 16) | b01;
    const uint32_t ta = (p0a << 5) + (p1a - p0a) * axa;  // {t_top, t_bot}
    const uint32_t tb = (p0b << 5) + (p1b - p0b) * axb;
    va = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, ta), __builtin_bit_cast(u16x2, wya), 32768u, false);
    vb = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, tb), __builtin_bit_cast(u16x2, wyb), 32768u, false);
  };

  // stores: card descriptor + scalar row offset + lane column, no per-pixel address math; the d16_hi
  // form takes bits 16..23 of the blend as they are (the buffer-store builtin has no such form)
  const int wave_s = __builtin_amdgcn_readfirstlane(wave);
  i32x4 card;
  {
    const uint64_t cb = (uint64_t)(uintptr_t)dbase;
    card.x = __builtin_amdgcn_readfirstlane((int)(uint32_t)cb);
    card.y = __builtin_amdgcn_readfirstlane((int)(uint32_t)(cb >> 32) & 0xffff);
    card.z = DMZ_CARD_WIDTH * DMZ_CARD_HEIGHT;
    card.w = 0x00020000;
  }
  const int tile_off = y0 * DMZ_CARD_WIDTH + x;
  auto store_row = [&](int j, uint32_t v) {
    const int soff = tile_off + j * DMZ_CARD_WIDTH;
    asm volatile("buffer_store_byte_d16_hi %0, %1, %2, %3 offen" : : "v"(v), "v"(x1), "s"(card), "s"(soff) : "memory");
  };

  // ---- the strip: wave w takes the 23 rows from a = min(23 w, 67) (rows 67, 68 are produced twice,
  // with the same bytes), two rows per iteration as independent chains ----
  // 45 (two waves: no row twice) / 30 (three: even, no row twice) / 23 (four: rows 67, 68 twice)
  constexpr int kRows = kWaves == 3 ? 30 : ((TH + kWaves - 1) / kWaves) | 1;
  static_assert(kWaves * kRows >= TH && TH <= kThreads, "rows per wave");
  const int a = imin(wave_s * kRows, TH - kRows);

  if (!(ww.wrows & kFastFlag)) {
    // exact coordinates for every pixel (strips whose perspective term is too strong for the
    // extrapolated reciprocal below)
#pragma unroll 2
    for (int m = 0; m < kRows / 2; m++) {
      const int j0 = a + 2 * m, j1 = j0 + 1;
      uint32_t Xa, Ya, Xb, Yb, va, vb;
      exact_xy(row_terms(j0), Xa, Ya);
      exact_xy(row_terms(j1), Xb, Yb);
      blend2(Xa, Ya, Xb, Yb, va, vb);
      store_row(j0, va);
      store_row(j1, vb);
    }
    if constexpr (kRows & 1) {
      const int j = a + kRows - 1;
      uint32_t Xa, Ya, va, vb;
      exact_xy(row_terms(j), Xa, Ya);
      blend2(Xa, Ya, Xa, Ya, va, vb);
      store_row(j, va);
    }
    return;
  }

  // ---- filtered-exact coordinates.  The card byte depends on cvRound(fX) only, so a cheap fX is
  // enough wherever it is provably on the same side of every rounding boundary as the exact one.
  //   * 1 / Wd: Wd is linear in the row, so along a lane's rows (step 2 per chain) the linear
  //     extrapolation 2 y[k-1] - y[k-2] of the previous reciprocals has relative error ~ (2 rho)^2
  //     (rho = |M7| / |W|, the relative change of W per row) and ONE Newton step against the true Wd
  //     squares that; k_warp_windows admits a strip only if rho <= 2^-11, which with the start-up
  //     errors (see below) bounds the relative error of y by 2^-36: 3 fp64 operations instead of
  //     v_rcp_f64 (quarter rate) + 6.
  //   * fX = fma(Xn, y, magic'): magic' = 1.5 * 2^36 + 0.5 - 32 * origin puts the rounding position
  //     of the adder 16 bits BELOW the integer, so the low dword holds round((fX + 0.5) * 2^16):
  //     |fX| < 2^16 (window-relative) => absolute error of the cheap fX < 2^16 * 2^-36 * 2^16 = 2^-4
  //     units of the last kept bit, plus 1/2 unit of rounding.  floor() of that dword >> 16 equals
  //     cvRound(exact fX) unless its low 16 bits are exactly 0 (a multiple of 2^16 within 3/4 unit:
  //     the only place where floor, the half-way case of round-to-even included, can differ).
  //   * lanes whose dword has zero low bits (2^-16 per coordinate) take the exact sequence; the test
  //     is three v_min_u16 + one compare per pixel pair, the branch is wave-uniform.
  // (the uniform constants come from k_warp_windows through scalar loads: the fma needs no copy of its addend, and no wave
  // spends vector instructions on them)
  const double sW = wm.sw;  // Wd(row + 1) - Wd(row)
  auto newton = [](double Wd, double g) {
    const double t = __builtin_fma(-Wd, g, 1.0);
    return __builtin_fma(g, t, g);
  };
  // reciprocals at the virtual rows a-1 (full Newton sequence) and a-2, a-3, a-4 (one step from y[a-1]: relative error
  // (k rho)^2); the first extrapolations then start with <= 15 rho^2 and the Newton step leaves <= 225 rho^4 <= 2^-36.
  // The one step from y1 against Wd(a-1) - k sW is y1 (2 - (Wd(a-1) - k sW) y1) = y1 + k (sW y1) y1 wherever y1 Wd(a-1) = 1
  // (it is, to 2^-52): ONE fma per row on q = (sW y1) y1 instead of the row's Wd and two (round 5; same (k rho)^2).
  // (the filtered path needs its operands to ~2^-40 only: the chains start from the strip's row terms of k_warp_windows)
  const double aD = (double)a;
  const double Wa = __builtin_fma(sW, aD, ww.rw0s) + C;
  auto start_chains = [&](double &yA1, double &yA2, double &yB1, double &yB2) {
    const double Wd = Wa - sW;
    double y1 = __builtin_amdgcn_rcp(Wd);
    y1 = newton(Wd, y1);
    y1 = newton(Wd, y1);
    const double q = (sW * y1) * y1;
    yA1 = y1 + q, yA2 = __builtin_fma(3.0, q, y1);  // chain A: rows a, a+2, ... (previous: a-2, a-4)
    yB1 = y1, yB2 = __builtin_fma(2.0, q, y1);       // chain B: rows a+1, a+3, ... (previous: a-1, a-3)
  };
  // Wd of a chain advances by recurrence (two rows per step; in the round 2 - 4 form the numerators too): the cheap path needs
  // its operands to ~2^-40 only, so the rounding of a dozen additions is irrelevant; the exact row terms are the exact
  // path's business (row_terms)
  struct Chain {
    double Xn, Yn, Wd, y1, y2;
  };
  const double dW2 = wm.dw2;
  const double dX2 = 2.0 * M1, dY2 = 2.0 * M4;  // (the recurrence form only: DMZ_WARP_AFFINE == 0)
  // The loop in two forms (AFF: the strip carries kAffFlag).
  //   recurrences: fX = fma(Xn, y, magic'), Xn += 2 M1 -- 8 fp64 operations per pixel;
  //   affine:      Xn = X_a + M1 dj and Wd = Wd_a + sW dj are both linear in the row, so Xn / Wd = alpha + beta / Wd with
  //     alpha = M1 / sW (uniform over the frame, k_warp_windows) and beta = X_a - alpha Wd_a (per lane):
  //     fX = fma(beta, y, magic' + alpha) -- no numerator recurrences, 6 fp64 operations per pixel.  magic' + alpha is rounded
  //     to the adder's 2^-16 grid; beta is built from THAT alpha (aq = (magic' + alpha) - magic', exact), which leaves
  //     |alpha - aq| dj sW y <= 2^-17 * 30 * 2^-11 of error instead of 2^-17.
  //     Error budget of the cheap fX (units of 1/32 px; the filter needs < 2^-17, half a unit of the last kept bit):
  //     an error eps of y now enters as |beta y| eps <= (|fX| + |alpha|) eps.  |alpha| = 32 |M1 / W| / rho and
  //     32 |M1 / W| <= |dfX/dj| + |fX| rho <= 64 (the window bounds the slope), eps <= 225 rho^4 from the extrapolation:
  //     14400 rho^3 <= 2^-19.2 at rho = 2^-11 (where |alpha| <= 2^17); at the other end, |alpha| <= 2^30 (rho ~ 2^-24), the
  //     drift of the Wd recurrence (15 additions: 2^-49) and the rounding of the Newton step (2^-51) give 2^-19 + 2^-21;
  //     the |fX| eps term as before: <= 2^-20.  Either way the sum is < 2^-18.
  //   linear:      (kLinFlag: rho <= 2^-24, the nearly affine maps whose |alpha| is beyond the bound above)  1 / Wd down the
  //     wave's rows is y_a (1 - u + u^2 - ...), u = rho' dj <= 29 * 2^-24: the linear part alone is within (29 rho)^2 <= 2^-38
  //     -- y advances by a constant, no extrapolation, no Newton step: 5 fp64 operations per pixel.
  auto run_fast = [&](auto mode_tag) {
    constexpr int MODE = decltype(mode_tag)::value;
    constexpr bool AFF = MODE == 1, LIN = MODE == 2;
    const double Kx = ww.kx, Ky = ww.ky;  // magic' (+ alpha)
    double bx = 0., by = 0.;
    Chain cA, cB;
    {
      cA.Xn = __builtin_fma(M1, aD, ww.rx0) + A, cA.Yn = __builtin_fma(M4, aD, ww.ry0) + B, cA.Wd = Wa;
      cB.Xn = cA.Xn + M1, cB.Yn = cA.Yn + M4, cB.Wd = Wa + sW;
    }
    if constexpr (!LIN) start_chains(cA.y1, cA.y2, cB.y1, cB.y2);
    if constexpr (AFF) {
      bx = __builtin_fma(-ww.aqx, cA.Wd, cA.Xn);
      by = __builtin_fma(-ww.aqy, cA.Wd, cA.Yn);
    }
    if constexpr (LIN) {
      // y at row a (v_rcp_f64 and two Newton steps: exact to the last bits), its slope dy = -sW y_a^2; chain A starts at
      // row a, chain B one row further (y2 holds the step of two rows)
      double ya = __builtin_amdgcn_rcp(Wa);
      ya = newton(Wa, ya);
      ya = newton(Wa, ya);
      const double dy = -(sW * ya) * ya;
      cA.y1 = ya, cB.y1 = ya + dy;
      cA.y2 = cB.y2 = 2.0 * dy;
    }
    auto fast_xy = [&](Chain &c, uint32_t &Xl, uint32_t &Yl) {
      double yn;
      if constexpr (LIN) {
        yn = c.y1;
        c.y1 += c.y2;
      } else {
        yn = newton(c.Wd, __builtin_fma(2.0, c.y1, -c.y2));
        c.y2 = c.y1;
        c.y1 = yn;
      }
      // (v_fma_f64 spelled out: the compiler picks the two-address v_fmac_f64 here and pays a 64-bit
      // register copy of the addend per pixel)
      double fx, fy;
      if constexpr (AFF) {
        asm("v_fma_f64 %0, %1, %2, %3" : "=v"(fx) : "v"(bx), "v"(yn), "s"(Kx));
        asm("v_fma_f64 %0, %1, %2, %3" : "=v"(fy) : "v"(by), "v"(yn), "s"(Ky));
      } else {
        asm("v_fma_f64 %0, %1, %2, %3" : "=v"(fx) : "v"(c.Xn), "v"(yn), "s"(Kx));
        asm("v_fma_f64 %0, %1, %2, %3" : "=v"(fy) : "v"(c.Yn), "v"(yn), "s"(Ky));
        c.Xn += dX2;
        c.Yn += dY2;
      }
      Xl = (uint32_t)__double2loint(fx);
      Yl = (uint32_t)__double2loint(fy);
      if constexpr (!LIN) c.Wd += dW2;
    };
#pragma unroll
    for (int m = 0; m < kRows / 2; m++) {
      const int j0 = a + 2 * m, j1 = j0 + 1;
      uint32_t Xa, Ya, Xb, Yb, va, vb;
      fast_xy(cA, Xa, Ya);
      fast_xy(cB, Xb, Yb);
      uint32_t lo16;
      unsigned long long amb;
      // (three two-operand v_min_u16: the 16-bit VOP2 forms issue at full rate on gfx950, the three-operand v_min3_u16 at a
      // quarter of it -- profiles/r4_valu_table_gfx950.txt: 2.4 against 8.5 cycles)
      asm("v_min_u16 %0, %2, %3\n\t"
          "v_min_u16 %0, %0, %4\n\t"
          "v_min_u16 %0, %0, %5\n\t"
          "v_cmp_eq_u16 %1, 0, %0"
          : "=&v"(lo16), "=s"(amb)
          : "v"(Xa), "v"(Ya), "v"(Xb), "v"(Yb));
      if (__builtin_expect(amb != 0, 0)) {
        exact_xy(row_terms(j0), Xa, Ya);
        exact_xy(row_terms(j1), Xb, Yb);
      }
#ifdef DMZ_WARP_VERIFY  // developer check (tools/dev/warp_verify.sh): every cheap coordinate against the exact sequence
      {
        uint32_t Xe, Ye, Xf, Yf;
        exact_xy(row_terms(j0), Xe, Ye);
        exact_xy(row_terms(j1), Xf, Yf);
        if ((Xe ^ Xa) >> 16 || (Ye ^ Ya) >> 16 || (Xf ^ Xb) >> 16 || (Yf ^ Yb) >> 16)
          printf("WARP MISMATCH frame %d tile %d lane %d rows %d: %08x %08x %08x %08x exact %08x %08x %08x %08x mode %d\n", frame,
                 tile, lane, j0, Xa, Ya, Xb, Yb, Xe, Ye, Xf, Yf, MODE);
        if (logical == 0 && tid == 0 && m == 0) printf("warp verify active, mode %d\n", MODE);
      }
#endif
      blend2(Xa, Ya, Xb, Yb, va, vb);
      store_row(j0, va);
      store_row(j1, vb);
    }
    if constexpr (kRows & 1) {
      const int j = a + kRows - 1;
      uint32_t Xa, Ya, va, vb;
      fast_xy(cA, Xa, Ya);
      if (__builtin_amdgcn_ballot_w64((Xa & 0xffffu) == 0u || (Ya & 0xffffu) == 0u)) exact_xy(row_terms(j), Xa, Ya);
      blend2(Xa, Ya, Xa, Ya, va, vb);
      store_row(j, va);
    }
  };
#if DMZ_WARP_AFFINE
  if (ww.wrows & kAffFlag)
    run_fast(std::integral_constant<int, 1>{});
  else
    run_fast(std::integral_constant<int, 2>{});
#else
  run_fast(std::integral_constant<int, 0>{});
#endif
}

}  // namespace

#ifdef DMZ_DEV_SELFCHECK
extern "C" void dmz_dbg_selfcheck_warp(unsigned long long *out) {
  (void)hipDeviceSynchronize();
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dev_selfcheck_warp), sizeof(unsigned long long) * 2);
}
#endif

void dmz_launch_warp(hipStream_t s, const uint8_t *planes, size_t frame_stride, int row_stride,
                     int width, int height, int n, DmzWarpMat *mats, uint8_t *cards,
                     size_t card_stride) {
  const int n_pad = (n + 7) & ~7;
  const int aligned = ((((uintptr_t)planes) | (uintptr_t)frame_stride | (uintptr_t)row_stride) & 3) == 0;
  hipLaunchKernelGGL(k_warp_windows, dim3((unsigned)((n * kTiles + 255) / 256)), dim3(256), 0, s, n, width,
                     height, aligned, mats);
  DMZ_REPEAT(warp)
  hipLaunchKernelGGL(k_warp, dim3((unsigned)n_pad * kTiles), dim3(kThreads), 0, s, planes, frame_stride,
                     row_stride, width, height, n, n_pad, mats, cards, card_stride);
}
