// detect.hip -- card-edge detection for one (frame, detection box) per workgroup.
//
// Replaces, for a whole batch, the reference's best_line_for_sample
// (dmz.cpp:224-271): Sobel-7 dx/dy (cv/sobel.cpp:476-478 = cvSobel ksize 7 on an
// isolated ROI), adaptive Canny (cv/canny.cpp:568-580, 58-336) and the
// gradient-gated 10-angle Hough (cv/hough.cpp:52-195).  All integer results are
// bit-exact with the reference semantics:
//   * Sobel: separable integer correlation, replicate border at the ROI edge,
//     int32 accumulate, saturate to int16.
//   * Canny: the reference's stack flood fill is order independent in its
//     result: edge set = 8-connected components of {NMS survivors with m > low}
//     that contain a survivor with m > high.  Here: seeds are marked, then the
//     mark is propagated over the candidate list (each hit is chased along its
//     chain) until an iteration changes nothing.
//   * Hough: votes are LDS integer atomics (order independent); the arg-max
//     reproduces the reference's r-outer / n-inner / strict-> scan by breaking
//     ties towards the smallest (r, n).
//
// CDNA4 mapping ("walk" kernel).  The box is thin (28 or 38 px) and long (389 or
// 241 px).  Everything is done in WALK SPACE: lanes are laid across the long
// axis (one lane per column for the top/bottom boxes, one per row for the
// left/right boxes, whose tile is transposed while it is loaded; 62 outputs + 2
// halo lanes per wave) and every lane walks the short axis with a 7-deep
// register window.  The across-axis 7-tap pair is three aligned LDS dwords, two
// v_alignbyte and four v_dot4_u32_u8; the separable Sobel, the |dx|+|dy|
// magnitude and the 3x3 non-maximum suppression never leave registers:
// neighbours along the walk are the lane's own previous/next step, neighbours
// across are the adjacent lanes (DPP wave shifts, fetched only in wave-steps that have
// a pixel above the low threshold).  LDS holds only the
// u8 source tile, the u8 edge map and the u16 vote counters, so one workgroup's barriers
// are covered by the others.  The adaptive thresholds need the box-wide mean of
// |dx|+|dy| before NMS.  For the standard geometry (28- and 38-step boxes of a
// 640 x 480 frame) the walk runs ONCE: the lane keeps the (dx, dy) of its 28 / 38
// steps packed as sign-flipped s16 pairs in registers (one v_sad_u16 against
// 0x80008000 gives |dx| + |dy| back), and the NMS pass reads them from there; the
// edge map then lies over the tile and the vote counters exist for the rho bins the box's
// pixels can reach only (27.7 KB / 12 KB per workgroup: 28 / 32 waves per CU).  Other box sizes take the two-walk form (sum pass, NMS pass), which
// recomputes the gradients.  (Rounds 2 - 4 kept 14 / 28 of the gradients in registers
// and parked the rest in LDS: with round 5's NMS the kernels need 52 / 63 registers
// with all of them resident.)
#include <mutex>

#include "dmz_hip_internal.h"
#include "dmz_wave.h"

namespace {

// developer ablation (tools/ablate.sh): -DDMZ_DETECT_STOP=k returns after phase k
#ifndef DMZ_DETECT_STOP
#define DMZ_DETECT_STOP 99
#endif
// developer probe: -DDMZ_DT_TIMING leaves the cycle counter at the phase boundaries in the hit record (max_val, r, n, found
// of the probed workgroup are overwritten: timing runs only)
#ifdef DMZ_DT_TIMING
__device__ long long g_dt_t[2][8];
#define DT_T(i) if (tid == 0 && blockIdx.x == gridDim.x / 2) g_dt_t[VERT ? 1 : 0][i] = clock64();
#else
#define DT_T(i)
#endif
#define DMZ_STOP_AFTER(k, expr)                                   \
  DT_T(k)                                                         \
  if (DMZ_DETECT_STOP == (k)) {                                   \
    if (tid == 0) {                                               \
      DmzBoxHit hh__ = {0, 0, 0, (int)(expr)};                    \
      hits[frame * 4 + box_id] = hh__;                            \
    }                                                             \
    return;                                                       \
  }

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ __forceinline__ int iabs(int v) { return v < 0 ? -v : v; }

// map byte bits
constexpr int MAP_CAND = 1;   // survived NMS with m > low
constexpr int MAP_EDGE = 2;   // final edge pixel (seeded with m > high, then propagated)
constexpr int MAP_GATE = 4;   // gradient direction accepted by the Hough gate

struct WalkCtx {
  const unsigned char *row0;  // LDS: aligned dword that holds tile byte (step 0, lane - 3)
  int sh;                     // byte offset of (lane - 3) inside that dword
  int sp;                     // tile row stride
  int L, S;                   // lanes across, steps along
  int l;                      // this lane's across coordinate (may be -1 or >= L on halo lanes)
};

// across-axis 7-tap pair (derivative {-1,-4,-5,0,5,4,1}, smooth {1,6,15,20,15,6,1}) of the
// tile row `a` (already clamped to [0, S-1]) centred on this lane.  (Round 3, measured and rejected: the tile with bit 7
// flipped and two CHAINED v_dot4_i32_i8 per output -- negative taps applied directly, one instruction fewer per step --
// is bit-exact and 1.4 % slower on the whole kernel: the chained accumulator serialises what were independent dot products.)
__device__ __forceinline__ void across_taps(const WalkCtx &c, int a, int &ad, int &as) {
  const uint32_t *p = (const uint32_t *)(c.row0 + a * c.sp);
  const uint32_t w0 = p[0], w1 = p[1], w2 = p[2];
  const uint32_t lo = __builtin_amdgcn_alignbyte(w1, w0, c.sh);  // taps 0..3
  const uint32_t hi = __builtin_amdgcn_alignbyte(w2, w1, c.sh);  // taps 4..6 (+1 unused)
  as = (int)__builtin_amdgcn_udot4(lo, 0x140F0601u, __builtin_amdgcn_udot4(hi, 0x0001060Fu, 0u, false), false);
  ad = (int)__builtin_amdgcn_udot4(hi, 0x00010405u, 0u, false) -
       (int)__builtin_amdgcn_udot4(lo, 0x00050401u, 0u, false);
}

// Along the walk the two 7-tap filters are binomial CASCADES instead of weighted sums over a 7-deep window: the smooth
// {1,6,15,20,15,6,1} = [1 1]^6 is six running pair sums, the derivative {-1,-4,-5,0,5,4,1} = [1 1]^4 * [-1 0 1] four pair
// sums and one difference two steps apart -- eleven plain integer additions per step (the full-rate class of
// tools/ubench/valu_table.hip; the weighted sums took five v_mul_lo_u32 and four v_add3_u32 of the half-rate class), and
// twelve registers of state instead of fourteen.  Exact: integer sums, |values| <= 255 * 10 * 64.
struct Window {
  int pd[6];     // across-derivative: the previous input and the previous value of cascade levels 1..5
  int ps[4];     // across-smooth: the previous input and the previous value of levels 1..3
  int b1, b2;    // level 4 of the across-smooth cascade one and two pushes ago
};

// push the across-axis pair (ad, as) of one along-position; after the push of position t the filters centred on t - 3 are out
__device__ __forceinline__ void window_push_taps(Window &wn, int ad, int as, int &g_ds, int &g_sd) {
  int v = ad;
#pragma unroll
  for (int k = 0; k < 6; k++) {
    const int nv = v + wn.pd[k];
    wn.pd[k] = v;
    v = nv;
  }
  g_ds = v;
  int u = as;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int nu = u + wn.ps[k];
    wn.ps[k] = u;
    u = nu;
  }
  g_sd = u - wn.b2;
  wn.b2 = wn.b1;
  wn.b1 = u;
}
// push along-position a (clamped by the caller's arithmetic)
__device__ __forceinline__ void window_push(const WalkCtx &c, Window &wn, int a, int &g_ds, int &g_sd) {
  int ad, as;
  across_taps(c, a, ad, as);
  window_push_taps(wn, ad, as, g_ds, g_sd);
}

__device__ __forceinline__ void window_init(const WalkCtx &c, Window &wn) {
  // along positions -3..+2 (position +3 is pushed by the first window_step): whatever the zeroed state contributes has left
  // the cascades by then (level k at push t holds inputs t-k..t)
#pragma unroll
  for (int k = 0; k < 6; k++) wn.pd[k] = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) wn.ps[k] = 0;
  wn.b1 = wn.b2 = 0;
  // (positions -3 .. 0 are all row 0 -- replicate border: its pair is evaluated once)
  int g_ds, g_sd, ad0, as0;
  across_taps(c, 0, ad0, as0);
#pragma unroll
  for (int i = 0; i < 4; i++) window_push_taps(wn, ad0, as0, g_ds, g_sd);
  window_push(c, wn, clampi(1, 0, c.S - 1), g_ds, g_sd);
  window_push(c, wn, clampi(2, 0, c.S - 1), g_ds, g_sd);
}

// step s: push position s+3, return the saturated (dx, dy)
// SAT = false leaves the saturation to the caller (pack_gradient does it while packing)
template <bool VERT, int K, bool SAT = true>
__device__ __forceinline__ void window_step(const WalkCtx &c, Window &wn, int s, int &dx, int &dy) {
  int g_ds, g_sd;  // along axis: the derivative smoothed, the smooth differentiated
  window_push(c, wn, clampi(s + 3, 0, c.S - 1), g_ds, g_sd);
  // top/bottom boxes: across = x  => dx = g_ds, dy = g_sd ; left/right boxes: across = y
  dx = SAT ? clampi(VERT ? g_sd : g_ds, -32768, 32767) : (VERT ? g_sd : g_ds);
  dy = SAT ? clampi(VERT ? g_ds : g_sd, -32768, 32767) : (VERT ? g_ds : g_sd);
}

// (dx, dy) of a step, saturated to 16 bits and packed with the sign bits flipped in ONE v_cvt_pk_i16_i32 + xor:
// each half is the gradient plus 32768 as an unsigned 16-bit number
__device__ __forceinline__ uint32_t pack_gradient(int gx, int gy) {
  typedef short s16x2 __attribute__((ext_vector_type(2)));
  const s16x2 p = __builtin_amdgcn_cvt_pk_i16(gx, gy);  // {sat16(gx), sat16(gy)}
  return __builtin_bit_cast(uint32_t, p) ^ 0x80008000u;
}
// sum of the saturated |dx| + |dy| (cvAbs: |-32768| -> 32767) of a packed gradient, added to acc: the half
// 0x0000 (= -32768) is lifted to 0x0001 (= -32767), then one v_sad_u16 against the bias
__device__ __forceinline__ int add_abs_sat(uint32_t g, int acc) {
  typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
  const u16x2 one = {1, 1};
  const u16x2 lifted = __builtin_elementwise_max(__builtin_bit_cast(u16x2, g), one);
  return (int)__builtin_amdgcn_sad_u16(__builtin_bit_cast(uint32_t, lifted), 0x80008000u, (uint32_t)acc);
}

// lane - 1 / lane + 1 of the wave by DPP (wave_shr:1 / wave_shl:1; the end lanes, which own no pixel, read 0):
// one VALU instruction instead of a ds_bpermute round trip
__device__ __forceinline__ int lane_below(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x138, 0xf, 0xf, true); }
__device__ __forceinline__ int lane_above(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x130, 0xf, 0xf, true); }

// Candidates per wave beyond which the hysteresis leaves the candidate lists for the bitmap flood: a card edge in front of a
// quiet background leaves a handful of weak candidates per wave, a frame without a card (texture, noise) hundreds -- and the
// list walk then costs (passes x candidates): 350 - 600 k cycles per box on noise frames against 4 k on card frames.
constexpr int kDenseList = 96;

// NMS (canny.cpp:213-285) + Hough slope gate (hough.cpp:133-150) for pixel (lane, step s),
// given its gradient and its magnitude at steps s-1 / s / s+1 (mp, mc, mn; the neighbouring lanes' are fetched here, by
// DPP, and only in wave-steps that get past the threshold test: every lane of the wave must call -- a DPP source lane has
// to be active).
// The reference's branches are kept: a wave whose 62 pixels are all below the low threshold (most of
// the background) skips everything -- the map was zeroed beforehand (thresholds_from) and is written for
// pixels above the threshold only --, and the slope gate runs only where a pixel survived
// (straight-line selects measured 17 % slower on the whole kernel).
// owner: the lane produces output for its coordinate.  LC: the box's lane count when it is a compile-time constant
// (the boxes of a 640 x 480 frame: the map offset of a step is then an immediate of the LDS store), else 0.
template <bool VERT, int LC>
__device__ __forceinline__ void nms_pixel(const DmzBoxParams &bp, const WalkCtx &c, int s, bool owner, int low,
                                          int high, int dxc, int dyc, int mp, int mc, int mn, unsigned char *map,
                                          unsigned short *seg, int seg_cap, int &ncand, int *s_int) {
  const int TG22 = 13573;  // (int)(0.4142135623730950488016887242097*(1<<15) + 0.5)
  const int m = mc;
  const bool above = owner && m > low;
  if (__builtin_amdgcn_ballot_w64(above) == 0ull) return;  // (wave-uniform)
  const int mp_lo = lane_below(mp), mp_hi = lane_above(mp);
  const int mc_lo = lane_below(mc), mc_hi = lane_above(mc);
  const int mn_lo = lane_below(mn), mn_hi = lane_above(mn);
  // neighbours in image coordinates
  int mN, mS, mW, mE, mNW, mNE, mSW, mSE;
  if (!VERT) {  // row = step, col = lane
    mN = mp; mS = mn; mW = mc_lo; mE = mc_hi;
    mNW = mp_lo; mNE = mp_hi; mSW = mn_lo; mSE = mn_hi;
  } else {      // row = lane, col = step
    mN = mc_lo; mS = mc_hi; mW = mp; mE = mn;
    mNW = mp_lo; mSW = mp_hi; mNE = mn_lo; mSE = mn_hi;
  }
  const int q = s * (LC ? LC : c.L) + c.l;  // walk-space index
  int flags = 0;
  if (above) {
    // canny.cpp:224-236 in 32 bits: |dx|, |dy| <= 32768, so x*TG22 < 2^29, tg67x < 2^32, y<<15 <= 2^30
    const unsigned ax = (unsigned)iabs(dxc), ay = (unsigned)iabs(dyc);
    const unsigned tg22x = ax * (unsigned)TG22;
    const unsigned tg67x = tg22x + ((ax + ax) << 15);
    const unsigned yy = ay << 15;
    bool is_max;
    if (yy < tg22x) {
      is_max = m > mW && m >= mE;
    } else if (yy > tg67x) {
      is_max = m > mN && m >= mS;
    } else {
      const bool neg = (dxc ^ dyc) < 0;  // s = -1
      is_max = neg ? (m > mNE && m > mSW) : (m > mNW && m > mSE);
    }
    if (is_max) {
      flags = MAP_CAND | (m > high ? MAP_EDGE : 0);
      // Hough slope gate (hough.cpp:133-150), evaluated for NMS survivors only (only edge pixels ever
      // vote): (float)dy / (float)dx against the two tangents, without the division.  The float quotient
      // is >= a exactly when the real quotient is >= the midpoint below a (no quotient of two 16-bit
      // integers can sit ON a midpoint of neighbouring floats of this magnitude), likewise <= b; with the
      // sign of dx folded into dy both become one exact fp64 multiply-and-compare (25 + 16 bits).
      bool use;
      if (dxc != 0) {
        const double ya = (double)(dxc < 0 ? -dyc : dyc), xa = (double)(int)ax;
        const bool ge_a = ya >= bp.slope_ta * xa, le_b = ya <= bp.slope_tb * xa;
        use = VERT ? (ge_a && le_b) : (ge_a || le_b);
      } else {
        use = !VERT;
      }
      if (use) flags |= MAP_GATE;
      map[q] = (unsigned char)flags;
    }
  }
  // candidates that are not seeds go on THIS WAVE's list (seeds need no propagation): the count is wave-uniform,
  // so an append is a ballot and a prefix count -- no LDS atomic, no wait
  const bool cand = (flags & (MAP_CAND | MAP_EDGE)) == MAP_CAND;
  const unsigned long long bal = __builtin_amdgcn_ballot_w64(cand);
  if (bal) {
    if (cand) {
      const int slot = ncand + __popcll(bal & ((1ull << (threadIdx.x & 63)) - 1ull));
      // a list beyond kDenseList entries (or beyond its room) sends the box to the bitmap flood below
      if (slot < seg_cap && slot < kDenseList) seg[slot] = (unsigned short)q;
      else s_int[3] = 1;
    }
    ncand += __popcll(bal);
  }
}

// SC > 0: the box has exactly SC steps (compile time) and the walk runs once, a lane keeping its SC packed
// gradients in registers for the NMS pass; SC == 0: any size, two walks.
// LC: the box's lane count when the launch guarantees it (the single-walk kernels), else 0.
template <bool VERT, int NT, int SC, int LC>
__device__ void detect_body(const uint8_t *__restrict__ planes, size_t frame_stride, int row_stride,
                            const DmzBoxParams &bp, int frame, int box_id,
                            DmzBoxHit *__restrict__ hits, unsigned char *lds) {
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // (a scalar: what depends on it stays on the scalar unit)
  DT_T(0)
  const int w = bp.w, h = bp.h;
  const int L = bp.lanes, S = bp.steps, N = L * S;
  const uint32_t inv_L = bp.inv_w;
  const int off = bp.tile_off, sp = bp.tile_stride;

  // LDS regions: source tile | edge map | vote counters | scratch.  In the single-walk form (compact layout) the edge
  // map lies over the tile -- dead once every wave has finished its walk; L * S <= tile bytes -- and what follows the
  // tile is one region with two tenants: the candidate lists until the hysteresis is done, then the vote counters.
  constexpr bool kCompact = SC > 0;
  unsigned char *tile = lds;
  unsigned char *map = kCompact ? lds : lds + bp.lds_map;
  unsigned int *acc32 = (unsigned int *)(lds + bp.lds_acc);
  // candidate list, before voting: over the vote counters
  unsigned short *list = kCompact ? (unsigned short *)(lds + bp.lds_map) : (unsigned short *)(lds + bp.lds_acc);
  const int list_cap = kCompact ? (bp.lds_red - bp.lds_map) / 2 : bp.list_cap;
  // one list segment per wave
  const int seg_cap = list_cap / (NT / 64);
  unsigned short *seg = list + wave * seg_cap;
  int ncand = 0;  // wave-uniform
  long long *s_red = (long long *)(lds + bp.lds_red);                    // 16 x 8 B
  unsigned int *s_best = (unsigned int *)(lds + bp.lds_red + 128);  // 16 x 4 B
  int *s_int = (int *)(lds + bp.lds_red + 256);                          // low, high, ncand, overflow

  // ---- A. ROI -> LDS tile in walk space (32-bit global words), replicate 3 lanes ----
  // Buffer loads: frame descriptor + per-lane byte offset + scalar row offset, so the only
  // per-word work is the LDS store.  Top/bottom boxes: a wave per image row, lanes across the
  // row's words.  Left/right boxes: a lane per image row (= walk lane), the row's words held in
  // registers and stored a byte at a time (tile step = image column): consecutive lanes write
  // consecutive bytes.
  {
    const int wpr = ((bp.x + w - 1) >> 2) - (bp.x >> 2) + 1;  // global words per image row
#ifdef DMZ_DETECT_SAMEFRAME  /* developer ablation: every workgroup reads one of 64 frames (L2 hits, the same arithmetic) */
    const uint8_t *plane = planes + (size_t)(frame & 63) * frame_stride;
#else
    const uint8_t *plane = planes + (size_t)frame * frame_stride;
#endif
    // the descriptor covers the rows of the box; cvSetImageROI clipped the box to the image
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(plane + (size_t)bp.y * row_stride), 0, (unsigned)(h * row_stride), 0x00020000);
    const bool aligned = ((((uintptr_t)plane) | (uintptr_t)row_stride) & 3) == 0;
#ifdef DMZ_DETECT_NOLOAD  /* developer ablation: the tile keeps whatever LDS held (timing only) */
    if (frame < 0) {
#else
    if (!aligned) {
#endif
      // plane base / row stride not 4-byte aligned: words assembled from bytes, one word per thread
      for (int i = tid; i < wpr * h; i += NT) {
        const int r = i / wpr, j = i - r * wpr;
        const uint8_t *g = plane + (size_t)(bp.y + r) * row_stride + ((bp.x >> 2) + j) * 4;
        const uint32_t v = (uint32_t)g[0] | ((uint32_t)g[1] << 8) | ((uint32_t)g[2] << 16) | ((uint32_t)g[3] << 24);
        const int c0 = j * 4 - (bp.x & 3);
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const int cc = c0 + k;  // image column - box.x
          if (cc >= 0 && cc < w) tile[VERT ? cc * sp + off + r : r * sp + off + cc] = (unsigned char)(v >> (8 * k));
        }
      }
#ifndef DMZ_DETECT_NOLOAD
    } else if (!VERT) {
      // A wave owns the rows wave, wave + NT / 64, ...; a row is ceil(wpr / 64) word columns per lane.  Eight (row, word
      // column) items are requested before the first is stored: the row-by-row loop (load, wait, store) was one memory round
      // trip per item, ten in a row for the standard box -- 13 - 15 k of a workgroup's 57 k cycles (-DDMZ_DT_TIMING).
#ifndef DMZ_DT_INFLIGHT  /* (4 / 6 / 8 / 10 / 12 measured: see DESIGN) */
#define DMZ_DT_INFLIGHT 8
#endif
      constexpr int kInFlight = DMZ_DT_INFLIGHT;
      const int jt = (wpr + 63) >> 6;
      int r = wave, jj = 0;  // the next item (wave-uniform, in scalar registers: the row offset is a scalar operand of the load)
      while (r < h) {
        uint32_t v[kInFlight];
        int vr[kInFlight], vj[kInFlight];
#pragma unroll
        for (int u = 0; u < kInFlight; u++) {
          vr[u] = r, vj[u] = lane + 64 * jj;
          v[u] = 0u;
          if (r < h) {
            if (vj[u] < wpr) v[u] = __builtin_amdgcn_raw_buffer_load_b32(rs, ((bp.x >> 2) + vj[u]) * 4, r * row_stride, 0);
            if (++jj == jt) jj = 0, r += NT / 64;
          } else {
            vr[u] = -1;
          }
        }
#pragma unroll
        for (int u = 0; u < kInFlight; u++)
          if (vr[u] >= 0 && vj[u] < wpr)
            *(uint32_t *)(tile + vr[u] * sp + 4 + 4 * vj[u]) = v[u];  // off = 4 + (x & 3) keeps the word alignment
      }
    } else {
      constexpr int kChunk = 12;  // words per pass: one pass for boxes up to 44 px wide
      for (int r = tid; r < h; r += NT) {
        const int voff = r * row_stride + (bp.x >> 2) * 4;
        unsigned char *tcol = tile + off + r - (bp.x & 3) * sp;
        for (int j0 = 0; j0 < wpr; j0 += kChunk) {
          // a lane's row piece as three 16-byte loads: lanes are rows (640 B apart), so every load instruction touches 64
          // separate cache lines whatever its width -- a quarter of the instructions, a quarter of the address-unit time
          // (words past the box are never stored; past the end of the descriptor they read as 0)
          uint32_t v[kChunk];
          typedef uint32_t u32x4v __attribute__((ext_vector_type(4)));
#pragma unroll
          for (int j = 0; j < kChunk; j += 4) {
            const u32x4v q = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, 4 * (j0 + j), 0);
            v[j] = q.x, v[j + 1] = q.y, v[j + 2] = q.z, v[j + 3] = q.w;
          }
#pragma unroll
          for (int j = 0; j < kChunk; j++) {
#pragma unroll
            for (int k = 0; k < 4; k++) {
              const int cc = 4 * (j0 + j) + k - (bp.x & 3);  // tile step = image column - box.x
              if (j0 + j < wpr && cc >= 0 && cc < w)
                tcol[(4 * (j0 + j) + k) * sp] = (unsigned char)(v[j] >> (8 * k));
            }
          }
        }
      }
#endif
    }
    if (tid < 4) s_int[tid] = 0;
    __syncthreads();
    for (int i = tid; i < S * 8; i += NT) {
      const int a = i >> 3, k = i & 7;
      unsigned char *row = tile + a * sp;
      if (k < 3) row[off - 1 - k] = row[off];
      else if (k < 6) row[off + L + (k - 3)] = row[off + L - 1];
    }
    __syncthreads();
  }
  DMZ_STOP_AFTER(1, tile[off] + tile[(S - 1) * sp + off + L - 1])

  WalkCtx c;
  c.sp = sp; c.L = L; c.S = S;
  c.l = 62 * wave - 1 + lane;
  {
    const int b = off + clampi(c.l, 0, L - 1) - 3;  // tile byte column of tap 0
    c.row0 = tile + (b & ~3);
    c.sh = b & 3;
  }
  const bool inbox = c.l >= 0 && c.l < L;
  const bool owner = inbox && lane >= 1 && lane <= 62;  // produces output for its coordinate

  // thresholds from the box-wide sum (canny.cpp:573-578: mean in double; low = cvFloor(mean),
  // high = cvFloor(3.0f * mean))
  auto thresholds_from = [&](long long local_sum) {
    for (int o = 32; o > 0; o >>= 1) local_sum += __shfl_down(local_sum, o, 64);
    if (lane == 0) s_red[wave] = local_sum;
    __syncthreads();
    // the edge map starts out empty (the NMS pass writes its survivors only).  Here: every wave has finished its walk, so
    // the tile the map lies over in the compact layout is dead; lds_map and the map's own room are multiples of 16 bytes.
    {
      typedef uint32_t u32x4z __attribute__((ext_vector_type(4)));
      for (int i = tid * 16; i < N; i += NT * 16) *(u32x4z *)(map + i) = (u32x4z){0u, 0u, 0u, 0u};
    }
    if (tid == 0) {
      long long tot = 0;
      for (int i = 0; i < NT / 64; i++) tot += s_red[i];
      const double mean = (double)tot / (double)N;
      const double lowt = mean, hight = 3.0f * mean;
      s_int[0] = (int)floor(lowt);
      s_int[1] = (int)floor(hight);
    }
    __syncthreads();
  };
  int low, high;
  if constexpr (SC > 0) {
    // ---- B+C, single walk.  g[s] = (dy << 16 | dx & 0xffff) ^ 0x80008000: each half is the gradient
    // plus 32768 as an unsigned 16-bit number, so |dx| + |dy| = v_sad_u16(g, 0x80008000, 0) exactly
    // (|-32768| = 32768 included), and dx, dy come back with one xor, one v_bfe_i32, one shift.
    uint32_t g[SC];
    {
      Window wn;
      window_init(c, wn);
      int acc = 0;
      int ad_last = 0, as_last = 0;  // the across pair of the last row: pushed four times (replicate border)
#pragma unroll
      for (int s0 = 0; s0 < SC; s0++) {
        int g_ds, g_sd;  // along axis: the derivative smoothed, the smooth differentiated
        if (s0 + 3 < SC - 1) {
          window_push(c, wn, s0 + 3, g_ds, g_sd);
        } else {
          if (s0 + 3 == SC - 1) across_taps(c, SC - 1, ad_last, as_last);
          window_push_taps(wn, ad_last, as_last, g_ds, g_sd);
        }
        // top/bottom boxes: across = x  => dx = g_ds, dy = g_sd ; left/right boxes: across = y
        g[s0] = pack_gradient(VERT ? g_sd : g_ds, VERT ? g_ds : g_sd);
        acc = add_abs_sat(g[s0], acc);
      }
      thresholds_from(owner ? (long long)acc : 0ll);  // steps * 65534 fits an int for any box that fits LDS
      // outside the ROI the magnitude is 0: the halo lanes beyond the box forget their gradients (once, not per step)
      if (!inbox) {
#pragma unroll
        for (int s0 = 0; s0 < SC; s0++) g[s0] = 0x80008000u;
      }
    }
    low = s_int[0], high = s_int[1];
    DMZ_STOP_AFTER(2, low + high)
    {
      // the lane's magnitudes at steps s-1 (p), s (c), s+1 (n)
      int mp = 0, mc = 0;
      uint32_t gc = 0x80008000u;
      auto nms_step = [&](int sn, uint32_t gn) {
        const int mn = (int)__builtin_amdgcn_sad_u16(gn, 0x80008000u, 0u);  // (0 from the zero gradient past the last step)
        if (sn >= 1) {
          const uint32_t h = gc ^ 0x80008000u;
          const int dxc = (int)(short)(h & 0xffffu), dyc = (int)h >> 16;
          nms_pixel<VERT, LC>(bp, c, sn - 1, owner, low, high, dxc, dyc, mp, mc, mn, map, seg, seg_cap, ncand, s_int);
        }
        mp = mc;
        mc = mn;
        gc = gn;
      };
#pragma unroll
      for (int sn = 0; sn < SC; sn++) nms_step(sn, g[sn]);
      nms_step(SC, 0x80008000u);
      __syncthreads();
    }
  } else {
  // ---- B. pass 1: sum of saturated |dx| + |dy| (cvAbs: 32768 -> 32767) -> thresholds ----
  {
    long long local_sum = 0;
    if (owner) {
      Window wn;
      window_init(c, wn);
      int acc = 0;
#define DMZ_P1_STEP(K)                                              \
  if (s0 + K < S) {                                                  \
    int dx, dy;                                                      \
    window_step<VERT, K>(c, wn, s0 + K, dx, dy);                     \
    const int ax = iabs(dx), ay = iabs(dy);                          \
    acc += (ax > 32767 ? 32767 : ax) + (ay > 32767 ? 32767 : ay);    \
  }
      for (int s0 = 0; s0 < S; s0 += 7) {
        DMZ_P1_STEP(0) DMZ_P1_STEP(1) DMZ_P1_STEP(2) DMZ_P1_STEP(3)
        DMZ_P1_STEP(4) DMZ_P1_STEP(5) DMZ_P1_STEP(6)
      }
#undef DMZ_P1_STEP
      local_sum = acc;  // steps * 65534 fits an int for any box that fits LDS
    }
    thresholds_from(local_sum);
  }
  low = s_int[0], high = s_int[1];
  DMZ_STOP_AFTER(2, low + high)

  // ---- C. pass 2: gradients again, NMS + slope gate -> edge map (walk space) ----
  {
    Window wn;
    window_init(c, wn);
    // the lane's magnitudes at steps s-1 (p), s (c), s+1 (n)
    int mp = 0, mc = 0;
    int dxc = 0, dyc = 0;
#define DMZ_P2_STEP(K)                                                                   \
  if (s0 + K <= S) {                                                                      \
    const int sn = s0 + K;                                                                \
    int dxn = 0, dyn = 0, mn = 0;                                                         \
    if (sn < S) {                                                                         \
      window_step<VERT, K>(c, wn, sn, dxn, dyn);                                          \
      mn = inbox ? iabs(dxn) + iabs(dyn) : 0; /* outside the ROI the magnitude is 0 */   \
    }                                                                                     \
    if (sn >= 1)                                                                          \
      nms_pixel<VERT, 0>(bp, c, sn - 1, owner, low, high, dxc, dyc, mp, mc, mn, map, seg, \
                         seg_cap, ncand, s_int);                                          \
    mp = mc;                                                                              \
    mc = mn;                                                                              \
    dxc = dxn; dyc = dyn;                                                                 \
  }
    for (int s0 = 0; s0 <= S; s0 += 7) {
      DMZ_P2_STEP(0) DMZ_P2_STEP(1) DMZ_P2_STEP(2) DMZ_P2_STEP(3)
      DMZ_P2_STEP(4) DMZ_P2_STEP(5) DMZ_P2_STEP(6)
    }
#undef DMZ_P2_STEP
    __syncthreads();
  }
  }
  DMZ_STOP_AFTER(3, map[0] + map[N - 1] + ncand)

  // (the appends ran on the owner lanes only: lane 1 of every wave is one)
  ncand = __builtin_amdgcn_readlane(ncand, 1);

  // ---- D. hysteresis: propagate MAP_EDGE over 8-connected candidates until stable ----
  // (8-adjacency is the same in walk space: rows = steps, columns = lanes)
  {
    // every wave works through its own list; a wave whose list grew long sends the box to the bitmap flood (or, when the
    // bitmaps do not fit a non-standard box, everybody over the whole map)
    const bool overflow = s_int[3] != 0;
    constexpr int NW = NT / 64;
    const int nwords = S * NW;
    if (overflow && S <= 64 && NW * 64 <= list_cap * 2) {
      // ---- dense candidates: flood on bitmaps held in REGISTERS.  Lane a of wave w holds the two 64-bit words of step a:
      // `cand` = NMS survivors above the low threshold (seeds included), `edge` starts as the seeds; bits 1..62 own a pixel.
      // One local step: an edge bit spreads to the candidate bits among its eight neighbours -- the rows above and below are the
      // neighbouring LANES (two DPP moves each), the columns across a wave boundary arrive as halo bits -- then along the row
      // through runs of candidates by carry propagation.  A wave repeats the step until nothing changes WITHOUT a barrier;
      // then the waves trade their border columns (one byte per step through LDS) and repeat, until no wave changed: the same
      // fixed point as the stack flood fill.  (Round 3's form kept the words in LDS, one thread per word and one barrier
      // per row of propagation: 12 - 17 passes of 2 - 4 k cycles on a noise frame.) ----
      unsigned char *bord = (unsigned char *)list;  // [wave][step]: bit 0 = edge at lane 1, bit 1 = edge at lane 62
      unsigned long long C = 0ull, E = 0ull;
      for (int a = 0; a < S; a++) {
        const int fl = inbox ? (int)map[a * L + c.l] : 0;
        const unsigned long long bc = __builtin_amdgcn_ballot_w64((fl & MAP_CAND) != 0) & 0x7ffffffffffffffeull,
                                 be = __builtin_amdgcn_ballot_w64((fl & MAP_EDGE) != 0) & 0x7ffffffffffffffeull;
        if (lane == a) C = bc, E = be;
      }
      auto lane_dn = [](unsigned long long v) {  // the word of lane - 1 (0 at lane 0)
        const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)v, 0x138, 0xf, 0xf, true);
        const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(v >> 32), 0x138, 0xf, 0xf, true);
        return ((unsigned long long)hi << 32) | lo;
      };
      auto lane_up = [](unsigned long long v) {  // the word of lane + 1 (0 at lane 63)
        const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)v, 0x130, 0xf, 0xf, true);
        const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(v >> 32), 0x130, 0xf, 0xf, true);
        return ((unsigned long long)hi << 32) | lo;
      };
      for (;;) {
        if (lane < S) bord[wave * 64 + lane] = (unsigned char)(((E >> 1) & 1ull) | (((E >> 62) & 1ull) << 1));
        __syncthreads();
        unsigned long long halo = 0ull;  // lane 62 of the wave below = our lane 0, lane 1 of the wave above = our lane 63
        if (lane < S) {
          if (wave > 0) halo |= (unsigned long long)((bord[(wave - 1) * 64 + lane] >> 1) & 1);
          if (wave < NW - 1) halo |= (unsigned long long)(bord[(wave + 1) * 64 + lane] & 1) << 63;
        }
        int any = 0;
        for (;;) {
          const unsigned long long x = E | halo;
          const unsigned long long t = x | lane_dn(x) | lane_up(x);
          unsigned long long g = ((t | (t << 1) | (t >> 1)) & C) | E;
          {
            const unsigned long long up = C & (C ^ (C + g));
            const unsigned long long rc = __builtin_bitreverse64(C), rg = __builtin_bitreverse64(g);
            g |= up | __builtin_bitreverse64(rc & (rc ^ (rc + rg)));
          }
          const bool ch = g != E;
          E = g;
          if (__builtin_amdgcn_ballot_w64(ch) == 0ull) break;
          any = 1;
        }
        if (!__syncthreads_or(any)) break;
      }
      for (int a = 0; a < S; a++) {
        const unsigned long long be = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(E >> 32), a) << 32) |
                                      (unsigned)__builtin_amdgcn_readlane((int)(unsigned)E, a);
        if (owner && ((be >> lane) & 1ull)) map[a * L + c.l] = (unsigned char)(map[a * L + c.l] | MAP_EDGE);
      }
      __syncthreads();
    } else if (overflow && nwords * 16 <= list_cap * 2) {
      // (boxes of more than 64 steps: the same flood with the words in LDS, one thread per word, one barrier per pass)
      // ---- dense candidates: flood on bitmaps.  Word (s, w) = the 64 lanes of wave w at step s (bits 1..62 own a pixel);
      // `cand` = NMS survivors above the low threshold (seeds included), `edge` starts as the seeds.  One pass: an edge bit
      // spreads to the candidate bits among its eight neighbours (three rows, the word's neighbours lend their border bits),
      // then along the row through runs of candidates (Kogge-Stone fill), until a pass changes nothing: the same fixed point as
      // the stack flood fill, at a cost that does not depend on how many candidates there are. ----
      unsigned long long *bcand = (unsigned long long *)list, *bedge = bcand + nwords;
      for (int a = 0; a < S; a++) {  // this wave's column of words
        const int q = a * L + c.l;
        const int fl = inbox ? (int)map[q] : 0;
        const unsigned long long bc = __builtin_amdgcn_ballot_w64((fl & MAP_CAND) != 0), be = __builtin_amdgcn_ballot_w64((fl & MAP_EDGE) != 0);
        if (lane == 0) {
          bcand[a * NW + wave] = bc & 0x7ffffffffffffffeull;
          bedge[a * NW + wave] = be & 0x7ffffffffffffffeull;
        }
      }
      __syncthreads();
#ifdef DMZ_DT_TIMING
      int dt_passes = 0;
      const long long dt_h0 = clock64();
#endif
      for (;;) {
#ifdef DMZ_DT_TIMING
        dt_passes++;
#endif
        int changed = 0;
        for (int t = tid; t < nwords; t += NT) {
          const int a = t / NW, w = t - a * NW;
          const unsigned long long cm = bcand[t], mine = bedge[t];
          unsigned long long d = 0ull;
          for (int r = (a > 0 ? a - 1 : a); r <= (a < S - 1 ? a + 1 : a); r++) {
            unsigned long long x = bedge[r * NW + w];
            if (w > 0) x |= (bedge[r * NW + w - 1] >> 62) & 1ull;         // lane 62 of the wave below = our lane 0
            if (w < NW - 1) x |= ((bedge[r * NW + w + 1] >> 1) & 1ull) << 63;  // lane 1 of the wave above = our lane 63
            d |= x | (x << 1) | (x >> 1);
          }
          // along the row through runs of candidates, by carry propagation instead of a Kogge-Stone fill: adding the seeds to the
          // run mask clears every run from its lowest seed upwards (and sets the bit above the run), so cm ^ (cm + seeds) marks
          // exactly those stretches; the downward direction is the same on the bit-reversed words (~25 instead of ~120 operations)
          unsigned long long g = (d & cm) | mine;
          {
            const unsigned long long up = cm & (cm ^ (cm + g));
            const unsigned long long rc = __builtin_bitreverse64(cm), rg = __builtin_bitreverse64(g);
            const unsigned long long dn = __builtin_bitreverse64(rc & (rc ^ (rc + rg)));
            g |= up | dn;
          }
          if (g != mine) {
            bedge[t] = g;
            changed = 1;
          }
        }
        if (!__syncthreads_or(changed)) break;
      }
#ifdef DMZ_DT_TIMING
      if (tid == 0 && blockIdx.x == gridDim.x / 2) printf("bitmap flood %s: %d passes, %lld cycles (build %lld)\n", VERT ? "vert" : "hz", dt_passes, clock64() - dt_h0, dt_h0 - g_dt_t[VERT ? 1 : 0][3]);
#endif
      for (int a = 0; a < S; a++) {
        const unsigned long long be = bedge[a * NW + wave];
        if (owner && ((be >> lane) & 1ull)) map[a * L + c.l] = (unsigned char)(map[a * L + c.l] | MAP_EDGE);
      }
      __syncthreads();
    } else {
    const int nitems = overflow ? N : ncand;
    const int i0 = overflow ? tid : lane, istep = overflow ? NT : 64;
    volatile unsigned char *vmap = map;
    for (;;) {
      int changed = 0;
      for (int i = i0; i < nitems; i += istep) {
        int q = overflow ? i : (int)seg[i];
        if ((vmap[q] & (MAP_CAND | MAP_EDGE)) != MAP_CAND) continue;
        int a = __umulhi((uint32_t)q, inv_L);
        int l = q - a * L;
        if (l >= L) { l -= L; a++; }
        // is one of the 8 neighbours an edge?
        bool hit = false;
        for (int da = -1; da <= 1 && !hit; da++) {
          const int aa = a + da;
          if (aa < 0 || aa >= S) continue;
          for (int dl = -1; dl <= 1; dl++) {
            const int ll = l + dl;
            if (ll < 0 || ll >= L || (da == 0 && dl == 0)) continue;
            if (vmap[aa * L + ll] & MAP_EDGE) { hit = true; break; }
          }
        }
        if (!hit) continue;
        changed = 1;
        // mark, then chase the chain of still-unmarked candidates from here
        for (;;) {
          vmap[q] = (unsigned char)(vmap[q] | MAP_EDGE);
          int next = -1, na = 0, nl = 0;
          for (int da = -1; da <= 1 && next < 0; da++) {
            const int aa = a + da;
            if (aa < 0 || aa >= S) continue;
            for (int dl = -1; dl <= 1; dl++) {
              const int ll = l + dl;
              if (ll < 0 || ll >= L || (da == 0 && dl == 0)) continue;
              if ((vmap[aa * L + ll] & (MAP_CAND | MAP_EDGE)) == MAP_CAND) {
                next = aa * L + ll; na = aa; nl = ll;
                break;
              }
            }
          }
          if (next < 0) break;
          q = next; a = na; l = nl;
        }
      }
      if (!__syncthreads_or(changed)) break;
    }
    }
  }
  DMZ_STOP_AFTER(4, map[0] + map[N - 1])

  // ---- E. Hough accumulator (hough.cpp:127-161), u16 counters packed in 32-bit words ----
  // Counters exist for the rho bins a pixel of this box can reach only (bp.rho_lo, bp.rho_cnt: fill_box_params).
  // They are kept in kDetectVoteCopies copies, a lane voting into copy lane & (copies - 1): the voters of one instruction are
  // neighbours on an edge, for the angles near the edge's own they name the same few counters, and equal addresses serialise an
  // LDS atomic at 2 cycles per lane.  Measured on one box (profiles/r5_detect_vote_copies_ab.log): one copy 4.20 ms, two 4.12,
  // four 4.14, eight 4.25 for the stage -- the conflicts are ~2 % of the kernel (the passes' 60 instructions are the rest).
  // The arg-max adds the copies (packed halves: totals < 2^16).
  const int numrho = bp.rho_cnt;
  const int copy_words = bp.acc_copy_bytes >> 2;
  {
    typedef uint32_t u32x4a __attribute__((ext_vector_type(4)));
    for (int i = tid; i < kDetectVoteCopies * copy_words / 4; i += NT) ((u32x4a *)acc32)[i] = (u32x4a){0u, 0u, 0u, 0u};
  }
  __syncthreads();
  {
    unsigned int *const acc_mine = acc32 + (lane & (kDetectVoteCopies - 1)) * copy_words;
    const int half = (bp.numrho - 1) / 2 - bp.rho_lo;
    const uint32_t *map32 = (const uint32_t *)map;  // lds_map is 16-byte aligned
    for (int q4 = tid; q4 < (N + 3) >> 2; q4 += NT) {
      const uint32_t w4 = map32[q4];
      // bytes with MAP_EDGE and MAP_GATE both set
      uint32_t hits4 = (w4 >> 1) & (w4 >> 2) & 0x01010101u;
      while (hits4) {
        const int b = (__builtin_ctz(hits4)) >> 3;
        hits4 &= hits4 - 1;
        const int q = 4 * q4 + b;
        if (q >= N) break;
        int a = __umulhi((uint32_t)q, inv_L);
        int l = q - a * L;
        if (l >= L) { l -= L; a++; }
        const int r = VERT ? l : a, col = VERT ? a : l;  // image coordinates inside the ROI
#pragma unroll
        for (int n = 0; n < kNumAngle; n++) {
          // counter of (n, rr): half n & 1 of word (n >> 1) * numrho + rr -- the half is a
          // compile-time constant per angle; counts < 65536, so no carry between the halves
          // (24-bit multiplies: coordinates < 2^16, table entries within +-1024 -- the plain int form compiled to
          // v_mul_lo_u32 + v_mad_u64_u32)
          const int t = __mul24(col, bp.tab_cos[n]) + __mul24(r, bp.tab_sin[n]);
          const int cell = (n >> 1) * numrho + half + (t >> 10);
          if (n == kNumAngle / 2) {
            // The middle angle is the box's own direction (its table entry across the box is 0 or -1): the voters of one
            // instruction -- neighbours on a card edge -- then all name ONE counter, and equal addresses serialise an LDS
            // atomic at 2 cycles per lane (tools/ubench/lds_atomic_rate.hip: 127 cycles for 64 lanes against 4 for 64
            // different counters).  One lane adds the count for all of them.
            const int c0 = __builtin_amdgcn_readfirstlane(cell);
            const unsigned long long act = __builtin_amdgcn_ballot_w64(true);
            if (__builtin_amdgcn_ballot_w64(cell != c0) == 0ull) {
              if ((int)(threadIdx.x & 63) == __builtin_ctzll(act))
                atomicAdd(&acc_mine[c0], (unsigned)__popcll(act) << ((n & 1) * 16));
              continue;
            }
          }
          atomicAdd(&acc_mine[cell], (n & 1) ? 0x10000u : 1u);
        }
      }
    }
  }
  __syncthreads();
  DMZ_STOP_AFTER(5, acc32[0] + acc32[numrho] + acc32[copy_words])
  static_assert(kNumAngle % 2 == 0, "vote counters are packed in angle pairs");

  // ---- F. arg-max with the reference's scan order (hough.cpp:163-176) ----
  // One 32-bit key per cell: votes (< 2^16) above the complement of the scan position (numrho * 10 < 2^16): the
  // maximum key is the first strict maximum of the reference's scan.  The bins without counters hold zero votes.
  unsigned int best = 0;
  {
    for (int p = 0; p < kNumAngle / 2; p++) {
      const unsigned int obase = 0xffffu - (unsigned int)(2 * p + bp.rho_lo * kNumAngle);
      for (int rr = tid; rr < numrho; rr += NT) {
        unsigned int w2 = acc32[p * numrho + rr];
#pragma unroll
        for (int cp = 1; cp < kDetectVoteCopies; cp++) w2 += acc32[cp * copy_words + p * numrho + rr];
        const unsigned int o0 = obase - (unsigned int)__mul24(rr, kNumAngle);  // 0xffff - scan position of the low half
        const unsigned int k0 = (w2 << 16) | o0, k1 = (w2 & 0xffff0000u) | (o0 - 1u);
        best = k0 > best ? k0 : best;
        best = k1 > best ? k1 : best;
      }
    }
  }
  best = dmzwave::max_u32(best);
  if (lane == 0) s_best[wave] = best;
  __syncthreads();
  if (tid == 0) {
    for (int i = 1; i < NT / 64; i++) best = s_best[i] > best ? s_best[i] : best;
    const int max_val = (int)(best >> 16);
    const unsigned int order = 0xffffu - (best & 0xffffu);
    DmzBoxHit hit;
    hit.max_val = max_val;
    hit.found = max_val > bp.threshold;
    if (max_val > 0) {
      hit.r = (int)(order / kNumAngle);
      hit.n = (int)(order % kNumAngle);
    } else {
      hit.r = 0;
      hit.n = 0;
    }
    hits[frame * 4 + box_id] = hit;
#ifdef DMZ_DT_TIMING
    if (blockIdx.x == gridDim.x / 2) {
      const long long t6 = clock64();
      printf("detect %s: load %lld walk %lld nms %lld hyst %lld votes %lld argmax %lld\n", VERT ? "vert" : "hz",
             g_dt_t[VERT][1] - g_dt_t[VERT][0], g_dt_t[VERT][2] - g_dt_t[VERT][1], g_dt_t[VERT][3] - g_dt_t[VERT][2],
             g_dt_t[VERT][4] - g_dt_t[VERT][3], g_dt_t[VERT][5] - g_dt_t[VERT][4], t6 - g_dt_t[VERT][5]);
    }
#endif
  }
}

// VERT = false: boxes 0 and 2 (top, bottom: horizontal lines, lanes = columns);
// VERT = true:  boxes 1 and 3 (left, right: vertical lines, lanes = rows).
// Developer switches: 0 = the two-walk kernels also for the standard boxes.
#ifndef DMZ_DETECT_SINGLE_H
#define DMZ_DETECT_SINGLE_H 1
#endif
#ifndef DMZ_DETECT_SINGLE_V
#define DMZ_DETECT_SINGLE_V 1
#endif
// waves per SIMD the register allocation aims at
#ifndef DMZ_DETECT_WPS_H
#define DMZ_DETECT_WPS_H 7
#endif
#ifndef DMZ_DETECT_WPS_V
#define DMZ_DETECT_WPS_V 8  /* (63 registers; the compact layout with counters for the reachable rho bins only is 12 KB: eight workgroups per CU) */
#endif
template <bool VERT, int NT, int SC, int LC>
__global__ __launch_bounds__(NT, SC == 0 ? 1 : (VERT ? DMZ_DETECT_WPS_V : DMZ_DETECT_WPS_H)) void k_detect_walk(const uint8_t *__restrict__ planes,
                                                     size_t frame_stride, int row_stride,
                                                     DmzDetectParams params,
                                                     DmzBoxHit *__restrict__ hits,
                                                     const int *__restrict__ skip_mask) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int frame = blockIdx.x >> 1;  // 1-D grid: gridDim.y is limited to 65535
  const int box_id = (blockIdx.x & 1) * 2 + (VERT ? 1 : 0);
  if (skip_mask && skip_mask[frame * 4 + box_id]) return;
  detect_body<VERT, NT, SC, LC>(planes, frame_stride, row_stride, params.box[box_id], frame, box_id, hits, lds);
}

#ifdef DMZ_DUP
#define DMZ_DUP_DETECT DMZ_TAG_CAT(DMZ_DUP)
#else
#define DMZ_DUP_DETECT 0
#endif
template <bool VERT, int NT, int SC, int LC>
int launch_pair(hipStream_t s, const uint8_t *planes, size_t frame_stride, int row_stride, int n,
                const DmzDetectParams &p, DmzBoxHit *hits, const int *skip_mask, int lds_bytes) {
  // (contexts may be driven from one host thread each: the once-per-geometry configuration below is serialised)
  static std::mutex mu;
  static int configured_lds = 0;  // per instantiation
  std::unique_lock<std::mutex> lk(mu);
  if (lds_bytes > configured_lds) {
    hipError_t e = hipFuncSetAttribute((const void *)k_detect_walk<VERT, NT, SC, LC>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    if (e != hipSuccess) return (int)e;
    configured_lds = lds_bytes;
  }
  lk.unlock();
  for (int dup__ = 0; dup__ < 1 + (int)(DMZ_DUP_DETECT == (VERT ? 2 : 1)); dup__++)
  hipLaunchKernelGGL((k_detect_walk<VERT, NT, SC, LC>), dim3(2u * (unsigned)n), dim3(NT), lds_bytes, s, planes,
                     frame_stride, row_stride, p, hits, skip_mask);
  return 0;
}

template <bool VERT>
int launch_pair_nt(hipStream_t s, const uint8_t *planes, size_t frame_stride, int row_stride, int n,
                   const DmzDetectParams &p, DmzBoxHit *hits, const int *skip_mask) {
  const DmzBoxParams &a = p.box[VERT ? 1 : 0], &b = p.box[VERT ? 3 : 2];
  const int nt = a.nthreads > b.nthreads ? a.nthreads : b.nthreads;
  const int lds = a.lds_total > b.lds_total ? a.lds_total : b.lds_total;
  // the boxes of a 640 x 480 frame (28 steps x 389 lanes, 38 steps x 241 lanes): single-walk kernels
#ifdef DMZ_DEV_HZ_LANES  /* developer probe (with capi.cpp's): what a top / bottom box of fewer waves costs */
  constexpr int kSteps = VERT ? 38 : 28, kLanes = VERT ? 241 : DMZ_DEV_HZ_LANES, kNt = VERT ? 256 : 64 * ((DMZ_DEV_HZ_LANES + 61) / 62);
#else
  constexpr int kSteps = VERT ? 38 : 28, kLanes = VERT ? 241 : 389, kNt = VERT ? 256 : 448;
#endif
  auto compact = [](const DmzBoxParams &q) {  // a list of >= 1024 entries fits behind the tile, the map fits the tile
    return q.lds_red - q.lds_map >= 2048 && q.lanes * q.steps <= q.lds_map;
  };
  if (a.steps == kSteps && b.steps == kSteps && a.lanes == kLanes && b.lanes == kLanes && nt <= kNt &&
      (VERT ? DMZ_DETECT_SINGLE_V : DMZ_DETECT_SINGLE_H) && compact(a) && compact(b)) {
    // Compact layout: the edge map lies over the tile, so what follows the tile is ONE region with two tenants -- the
    // candidate lists until the hysteresis is done, then the vote counters: tile | region | scratch
    // (20.6 KB instead of 30 KB for the left/right boxes of a 640 x 480 frame).
    // What the device holds, asked once: LDS per CU, and how many workgroups of this kernel the register file and the wave
    // slots admit (the occupancy query with no dynamic LDS).  No literals: a different part or compiler changes the answer.
    static std::mutex mu;  // the cached device answers and the per-geometry check: one thread at a time
    std::unique_lock<std::mutex> lk(mu);
    static int lds_cu = 0, wgs_regs = 0;
    const void *kfn = (const void *)k_detect_walk<VERT, kNt, kSteps, kLanes>;
    if (lds_cu == 0) {
      int dev = 0, v = 0;
      if (hipGetDevice(&dev) != hipSuccess ||
          hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerMultiprocessor, dev) != hipSuccess || v <= 0)
        v = 65536;  // the architectural minimum: the lists then keep their minimal size
      int nb = 0;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kfn, kNt, 0) != hipSuccess || nb < 1) nb = 1;
      (void)hipGetLastError();  // (a failed query must not surface as the launch's error)
      lds_cu = v, wgs_regs = nb;
    }
    DmzDetectParams q = p;
    int total = 0;
    auto layout = [&](bool grow_lists) {
      q = p;
      total = 0;
      int wgs_min = 1 << 20;
      for (int e = VERT ? 1 : 0; e < 4; e += 2) {
        DmzBoxParams &bx = q.box[e];
        const int acc = bx.lds_red - bx.lds_acc;
        int region = ((2048 > acc ? 2048 : acc) + 15) & ~15;
        // the candidate lists take what the workgroups-per-CU count leaves over (busy frames then stay on the list path
        // instead of the whole-map fallback): the count is the smaller of what LDS and the registers / wave slots allow
        const int tile = bx.lds_map;
        int wgs = lds_cu / (tile + region + 512);
        if (wgs > wgs_regs) wgs = wgs_regs;
        if (grow_lists && wgs >= 1) {
          // (1.5 KB short of an equal share: a kernel allocated to within 16 bytes of it lost a workgroup, tools/ubench/lds_granularity.hip)
          const int budget = ((lds_cu / wgs - 1536) & ~15) - tile - 512;
          if (budget > region && budget <= 65536) region = budget;
        }
        wgs_min = wgs < wgs_min ? wgs : wgs_min;
        bx.lds_acc = bx.lds_map;
        bx.lds_red = bx.lds_map + region;
        bx.lds_total = bx.lds_red + 512;
        total = bx.lds_total > total ? bx.lds_total : total;
      }
      return wgs_min;
    };
    // the grown layout must not cost a workgroup: checked against the occupancy query once per geometry, else the minimal one
    static int checked_total = -1, checked_ok = 0;
    const int want = layout(true);
    if (total != checked_total) {
      int nb = 0;
      // (the query needs the kernel's dynamic-LDS limit raised first, as the launch does)
      checked_ok = hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, total) == hipSuccess &&
                   hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kfn, kNt, (size_t)total) == hipSuccess && nb >= want;
      (void)hipGetLastError();
      checked_total = total;
    }
    if (!checked_ok) (void)layout(false);
    lk.unlock();
    return launch_pair<VERT, kNt, kSteps, kLanes>(s, planes, frame_stride, row_stride, n, q, hits, skip_mask, total);
  }
  if (nt <= 256) return launch_pair<VERT, 256, 0, 0>(s, planes, frame_stride, row_stride, n, p, hits, skip_mask, lds);
  if (nt <= 448) return launch_pair<VERT, 448, 0, 0>(s, planes, frame_stride, row_stride, n, p, hits, skip_mask, lds);
  return launch_pair<VERT, 1024, 0, 0>(s, planes, frame_stride, row_stride, n, p, hits, skip_mask, lds);
}

}  // namespace

int dmz_launch_detect(hipStream_t s, const uint8_t *planes, size_t frame_stride, int row_stride,
                      int n, const DmzDetectParams &p, DmzBoxHit *hits, const int *skip_mask) {
  int e = launch_pair_nt<false>(s, planes, frame_stride, row_stride, n, p, hits, skip_mask);
  if (e) return e;
  return launch_pair_nt<true>(s, planes, frame_stride, row_stride, n, p, hits, skip_mask);
}

int dmz_configure_detect(void) { return 0; }  // LDS limits are raised per launch geometry
