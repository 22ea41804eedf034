// expiry.hip -- the expiry path of the scan, batched over frames:
//   best_expiry_seg          scan/expiry_seg.cpp:707-902  (k_expiry_stripes)
//   find_character_groups_for_stripe  :437-704, gather_into_groups :131-167,
//   strip_group_white_space :101-129, regrid_group :169-229, optimize_character_rects :231-339,
//   is_slash :52-58 with applym_730c4cbd                     (k_expiry_seg)
//   prepare_image_for_cat    scan/expiry_categorize.cpp:35-70, categorize_expiry_digits :138-160
//   with applyc_bf4dd6c8 (models/expiry/modelc_bf4dd6c8.cpp:12500-13505)   (k_expiry_cat)
//
// Exactness: everything up to and including the character rectangles is integer or
// order-preserving IEEE float/double arithmetic (no contraction: the file is compiled with
// -ffp-contract=off and uses explicit fmaf only inside the CNN), so stripes, groups and rects
// are bit-exact.  The slash MLP's hidden layer runs on v_mfma_f32_16x16x32_bf16 with EXACT operand splits
// (integer samples, weights in three bf16 parts), the CNN's two convolutions on v_mfma_f32_16x16x32_f16 with both operands
// split in two f16 parts and three products (the default DMZ_HIP_EXPIRY_CONV_F16X3: fp32 to ~2^-22 per product; BF16X3 /
// BF16 keep the bf16 forms of round 2, the F32 variant packed FMAs and v_mfma_f32_16x16x4_f32), tanh is
// exp2/rcp based, the dense layers run on v_mfma_f32_16x16x4_f32 (a k-ordered fmaf chain):
// the slash decision P > 0.7 can differ from the oracle only within float noise of the threshold
// (none in 4 x 65 536 frames) and the digit scores agree to 5e-6 with F16X3 and F32 (measured 4.4e-6 / 3e-6 on 65 536 frames;
// 5e-5 with BF16X3; the reference's own KAT tolerance is 1e-5 -- tests/test_gpu_expiry.py).
//
// std::sort in the reference is unstable: candidates with equal sums are visited in the order libstdc++'s introsort
// leaves them (dmz_stdsort.h; oracle/orc_expiry.c is pinned on the reference's own instantiation).  Both picks run
// in parallel rounds with the column / row as tie-break and WATCH for a tie that could change the outcome (two live
// candidates with equal sums that exclude each other); only then the library's order is computed and the pick repeated.
//
// Mapping: the list logic is short, serial and data-dependent, so one 64-lane wave owns one
// (frame, stripe): lanes are columns, candidate rects, groups, grid hypotheses or character
// rects as the step requires, with ballot/popcount compaction between steps.  The CNN stage is
// one 256-thread workgroup per frame; the convolutions of a group run two digits at a time (LDS), the dense layers
// see its four digits.
#include <float.h>
#include <math.h>

#include <type_traits>
#include <utility>

#include "dmz_hip_internal.h"
#include "dmz_wave.h"
#include "dmz_stdsort.h"

// developer ablation (tools/ablate.sh): extra dynamic LDS per workgroup = fewer workgroups per CU
#ifndef DMZ_LDS_PAD
#define DMZ_LDS_PAD 0
#endif

namespace {

constexpr int CW = DMZ_CARD_WIDTH, CH = DMZ_CARD_HEIGHT;
constexpr int kNumberHeight = 27;  // dmz_constants.h kNumberHeight
constexpr int SCW = 9, SCH = 15;   // kSmallCharacterWidth / Height, expiry_types.h:16-17
constexpr int TW = 11, TH = 16;    // kTrimmedCharacterImageWidth / Height, expiry_types.h:18-19

__device__ __forceinline__ int imax(int a, int b) { return a > b ? a : b; }
__device__ __forceinline__ int imin(int a, int b) { return a < b ? a : b; }
__device__ __forceinline__ int iabs(int a) { return a < 0 ? -a : a; }

// Wave-wide reductions on DPP (dmz_wave.h)
#define DMZ_DPP_SHR(v, n) DMZ_DPP_SHR0(v, n)
__device__ __forceinline__ unsigned wave_max_u32(unsigned x) { return dmzwave::max_u32(x); }
__device__ __forceinline__ int wave_sum_i32(int v) { return dmzwave::sum_i32(v); }
__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v) {
  return dmzwave::min_u64((unsigned)(v >> 32), (unsigned)v);
}
__device__ __forceinline__ unsigned long long lanemask_lt(int lane) { return (1ull << lane) - 1ull; }

// ---------------------------------------------------------------------------------------------
// k_expiry_stripes: line sums of the |Scharr dx| image and the <= 3 probable stripes.
// The 3/10/3 vertical pass commutes with the row sum, so line_sum[r] = 3 (I[r-1] + I[r+1]) +
// 10 I[r] with I[r] = sum over columns 27..284 of |p[c+1] - p[c-1]| (one v_sad_u8 per dword) and
// the row index clamped to the ROI [y0, 269] (sobel.cpp:765-766).
// ---------------------------------------------------------------------------------------------
// f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>): an unrolled loop whose index is a constant expression
template <class F, int... Is>
__device__ __forceinline__ void static_for(F &&f, std::integer_sequence<int, Is...>) {
  (f(std::integral_constant<int, Is>{}), ...);
}
// v with lane LANE replaced by the (uniform) value x: one v_writelane_b32
template <int LANE>
__device__ __forceinline__ int with_lane(int x, int v) {
  asm("v_writelane_b32 %0, %1, %2" : "+v"(v) : "s"(x), "n"(LANE));
  return v;
}

// LDS the stripe search needs (a kernel's own, or -- in the fused kernel -- the head of the segmentation's block)
struct StripeLds {
  int I[128];
  int line[128 + 16];
  unsigned sv[128];
  unsigned sstack[20];
  unsigned short sq[128];
};
// the body for frame f by one wave: clears the frame's expiry record and stage counters, finds the stripes, writes them to
// out[f] and returns them (uniform) in base_row[] / sum[]; the return value is their number
__device__ __forceinline__ int expiry_stripes_body(const int f, const int lane, const uint8_t *__restrict__ cards, size_t card_stride,
                                                   const dmz_hip_frame_result *__restrict__ results,
                                                   dmz_hip_expiry_result *__restrict__ out, DmzExpiryStage *__restrict__ stage,
                                                   StripeLds &SL, int (&base_row)[3], long long (&sum_out)[3]) {
  int *const I = SL.I, *const line = SL.line;
  base_row[0] = base_row[1] = base_row[2] = 0;
  sum_out[0] = sum_out[1] = sum_out[2] = 0;
  {
    uint32_t *o32 = (uint32_t *)(out + f);
    for (int i = lane; i < (int)(sizeof(dmz_hip_expiry_result) / 4); i += 64) o32[i] = 0u;
    if (lane < 3) stage[(size_t)f * 3 + lane].n = 0;
  }
  const int flags = results[f].flags, yoff = results[f].vseg_y_offset;
  // frame.cpp:72 -- and the vseg gates of frame.cpp:38-47 that precede it
  if (!(flags & DMZ_HIP_FLAG_VSEG_OK) || !(yoff < CH - 2 * SCH) || yoff < 0) return 0;
  const int y0 = yoff + kNumberHeight;
  const int nrows = CH - y0;
  if (nrows > 128 || nrows < 18) return 0;  // y_offset >= 121 for a card that is not upside down
  const uint8_t *card = cards + (size_t)f * card_stride;

  // 32 rows per trip, all of a trip's loads in flight together.  A lane loads dword 6 + lane of every row (p[24 + 4 lane ..]);
  // its right-hand neighbour dword comes from lane + 1 by DPP, for lane 63 and for the two columns past the last whole dword
  // (p[284], p[285]) from ONE more load per trip: lane 2 u + k holds dword 70 + k of row u, and what depends on those is
  // uniform (scalar unit).  Round 5: 33 loads and ~500 VALU instructions per 32 rows; the form before it loaded both dwords
  // of a lane per row and evaluated the two extra columns on lane 0 under a branch (64 loads, ~640 instructions): stage
  // 3.42 -> 3.33 ms, step -0.15 ms (profiles/r5_stripes_trip32_ab.log).  Two rows per wave reduction (a lane's sum is < 2^11,
  // a 16-lane row's < 2^15).  The kernel re-reads the rows below the number in 258-byte pieces, ~92 per card: a four-wave form
  // with every row load of a frame in flight at once (round 3) has the same run time -- it is the memory system, not the wave's
  // latency chain; and run in the tail of k_vseg (round 5: the workgroup that has just found the number row, four waves, one
  // round trip) it costs k_vseg 0.44 ms where this kernel costs the three-queue step 0.50
  // (tools/dev/rejected/vseg_tail_stripes_r5.patch.txt, profiles/r5_vseg_tail_stripes_ab.log).
  constexpr int kTrip = 32;
  for (int r0 = 0; r0 < nrows; r0 += kTrip) {
    uint32_t a[kTrip];
#pragma unroll
    for (int u = 0; u < kTrip; u++)
      a[u] = ((const uint32_t *)(card + (size_t)(y0 + imin(r0 + u, nrows - 1)) * CW))[6 + lane];
    const uint32_t e = ((const uint32_t *)(card + (size_t)(y0 + imin(r0 + (lane >> 1), nrows - 1)) * CW))[70 + (lane & 1)];
    int sums = 0;  // lane u: I[r0 + u]
    static_for([&](auto pair) {
      constexpr int u = 2 * decltype(pair)::value;
      unsigned s[2], ex[2];
#pragma unroll
      for (int h = 0; h < 2; h++) {
        const uint32_t w70 = (uint32_t)__builtin_amdgcn_readlane((int)e, 2 * (u + h));      // p[280..283]
        const uint32_t w71 = (uint32_t)__builtin_amdgcn_readlane((int)e, 2 * (u + h) + 1);  // p[284..287]
        const uint32_t nb = (uint32_t)with_lane<63>(
            (int)w70, __builtin_amdgcn_update_dpp(0, (int)a[u + h], 0x130, 0xf, 0xf, true));  // dword 7 + lane (wave_shl:1)
        const uint32_t left = __builtin_amdgcn_alignbyte(nb, a[u + h], 2);                    // p[26 + 4 lane ..]
        s[h] = __builtin_amdgcn_sad_u8(nb, left, 0u);                                          // p[28 + 4 lane ..] vs left
        // |p[284] - p[282]| + |p[285] - p[283]|
        ex[h] = (unsigned)iabs((int)(w71 & 255u) - (int)((w70 >> 16) & 255u)) +
                (unsigned)iabs((int)((w71 >> 8) & 255u) - (int)(w70 >> 24));
      }
      int v = (int)(s[0] | (s[1] << 16));
      v += DMZ_DPP_SHR(v, 1);
      v += DMZ_DPP_SHR(v, 2);
      v += DMZ_DPP_SHR(v, 4);
      v += DMZ_DPP_SHR(v, 8);
      const unsigned q0 = (unsigned)__builtin_amdgcn_readlane(v, 15), q1 = (unsigned)__builtin_amdgcn_readlane(v, 31);
      const unsigned q2 = (unsigned)__builtin_amdgcn_readlane(v, 47), q3 = (unsigned)__builtin_amdgcn_readlane(v, 63);
      const unsigned lo = (q0 & 0xffffu) + (q1 & 0xffffu) + (q2 & 0xffffu) + (q3 & 0xffffu) + ex[0];
      const unsigned hi = (q0 >> 16) + (q1 >> 16) + (q2 >> 16) + (q3 >> 16) + ex[1];
      sums = with_lane<u>((int)lo, sums);
      sums = with_lane<u + 1>((int)hi, sums);
    }, std::make_integer_sequence<int, kTrip / 2>{});
    if (lane < kTrip && r0 + lane < nrows) I[r0 + lane] = sums;
  }
  __syncthreads();
  for (int r = lane; r < nrows; r += 64)
    line[r] = 3 * (I[imax(r - 1, 0)] + I[imin(r + 1, nrows - 1)]) + 10 * I[r];
  __syncthreads();

  // expiry_seg.cpp:790-836: stripes of 15 rows, base_row in [y0 + 1, 254)
  const int ncand = (CH - (SCH + 1)) - (y0 + 1);
  unsigned key[2];
#pragma unroll
  for (int j = 0; j < 2; j++) {
    const int idx = lane + 64 * j;
    key[j] = 0u;
    if (idx < ncand) {
      const int b = 1 + idx;  // row index relative to y0
      int sum = 0, mx = 0;
      for (int k = 0; k < SCH; k++) {
        const int v = line[b + k];
        sum += v;
        mx = imax(mx, v);
      }
      const int thr = mx / 2;
      bool good = !(line[b] + line[b + 1] < thr) && !(line[b + SCH - 2] + line[b + SCH - 1] < thr);
      for (int k = 0; k < SCH - 3; k++)
        if (line[b + k + 1] < thr && line[b + k + 2] < thr) good = false;
      if (good) key[j] = ((unsigned)sum << 7) | (unsigned)(127 - idx);
    }
  }
  // descending by sum, up to three that do not overlap (838-866): three rounds of "the best live candidate".  Equal sums:
  // the reference visits them in std::sort's order.  A round whose best sum is held by two live candidates is the only place
  // where that order can matter (0.3 % of the corpus' cards): then the good stripes are sorted as the library sorts them
  // (one lane; <= 111 elements) and the rounds repeat with the sorted position as tie-break.
  unsigned *const sv = SL.sv, *const sstack = SL.sstack;
  unsigned short *const sq = SL.sq;
  const unsigned k0 = key[0], k1 = key[1];
  int np = 0;
  auto rounds = [&](bool watch) -> bool {
    bool tie = false;
    np = 0;
    for (int round = 0; round < 3; round++) {
      const unsigned m = wave_max_u32(key[0] > key[1] ? key[0] : key[1]);
      if (m == 0u) break;
      // the holder of the best key (keys are distinct); its low bits are the row only in the watched run
      const unsigned long long h0 = __builtin_amdgcn_ballot_w64(key[0] == m), h1 = __builtin_amdgcn_ballot_w64(key[1] == m);
      const int idx = h0 ? __builtin_ctzll(h0) : 64 + __builtin_ctzll(h1);
      if (watch)  // another live candidate with the best sum?
        tie |= __builtin_amdgcn_ballot_w64((key[0] != m && (key[0] >> 7) == (m >> 7)) ||
                                           (key[1] != m && (key[1] >> 7) == (m >> 7))) != 0ull;
      if (lane == 0) {
        out[f].stripe_base_row[np] = y0 + 1 + idx;
        out[f].stripe_sum[np] = (int64_t)(m >> 7);
      }
      base_row[np] = y0 + 1 + idx;
      sum_out[np] = (long long)(m >> 7);
      np++;
#pragma unroll
      for (int j = 0; j < 2; j++)
        if (iabs(lane + 64 * j - idx) < SCH) key[j] = 0u;
    }
    return tie;
  };
  if (rounds(true)) {
    // stripe_sums in base-row order (830-835), std::sort (842)
    const unsigned long long b0 = __builtin_amdgcn_ballot_w64(k0 != 0u), b1 = __builtin_amdgcn_ballot_w64(k1 != 0u);
    const int n0 = __popcll(b0), ns = n0 + __popcll(b1);
    const int c0 = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(b0 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)b0, 0u));
    const int c1 = n0 + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(b1 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)b1, 0u));
    if (k0) sv[c0] = (k0 & ~127u) | (unsigned)lane;
    if (k1) sv[c1] = (k1 & ~127u) | (unsigned)(lane + 64);
    __syncthreads();
    if (lane == 0) dmzsort::serial_sort<7>(sv, ns, sstack);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 2; j++)
      if (lane + 64 * j < ns) sq[sv[lane + 64 * j] & 127u] = (unsigned short)(lane + 64 * j);
    __syncthreads();
    key[0] = k0 ? (k0 & ~127u) | (127u - sq[lane]) : 0u;
    key[1] = k1 ? (k1 & ~127u) | (127u - sq[lane + 64]) : 0u;
    rounds(false);
    if (lane == 0)
      for (int i = np; i < 3; i++) out[f].stripe_base_row[i] = 0, out[f].stripe_sum[i] = 0;
  }
  if (lane == 0) out[f].n_stripes = np;
  return np;
}

__global__ __launch_bounds__(64) void k_expiry_stripes(const uint8_t *__restrict__ cards, size_t card_stride,
                                                       int n, const dmz_hip_frame_result *__restrict__ results,
                                                       dmz_hip_expiry_result *__restrict__ out,
                                                       DmzExpiryStage *__restrict__ stage) {
  const int f = blockIdx.x, lane = threadIdx.x;
  if (f >= n) return;
  __shared__ StripeLds SL;
  int br[3];
  long long su[3];
  (void)expiry_stripes_body(f, lane, cards, card_stride, results, out, stage, SL, br, su);
}

// ---------------------------------------------------------------------------------------------
// k_expiry_seg: one wave per (frame, stripe)
//
// LDS (13.6 KB -> twelve workgroups per CU): the horizontal pass of the Scharr operator,
// inter[t][c] = |p[c+1] - p[c-1]| as bytes for the 23 image rows base-4 .. base+18; a Scharr sample
// is 3 (inter[k] + inter[k+2]) + 10 inter[k+1], three byte reads -- half the footprint of parking
// the int16 samples, which is what buys the occupancy for this latency-bound list logic.
// ---------------------------------------------------------------------------------------------
// developer ablation (tools/ablate.sh): -DDMZ_XSEG_STOP=k returns after phase k
#ifndef DMZ_XSEG_WAVES
#define DMZ_XSEG_WAVES 3
#endif
#ifndef DMZ_XSEG_STOP
#define DMZ_XSEG_STOP 99
#endif
#ifndef DMZ_XSEG_TIES  /* developer ablation (timing only, wrong ties): 0 = column tie-break, no watch; 1 = watch, never re-order */
#define DMZ_XSEG_TIES 2
#endif
#ifndef DMZ_XSEG_NT_FAST  /* developer A/B: 0 = the normalise-and-threshold of a Scharr sample on v_rndne / v_cvt / v_cmp / v_cndmask (rounds 2 - 5) */
#define DMZ_XSEG_NT_FAST 1
#endif
#ifndef DMZ_XSEG_FUSED  /* developer A/B: 1 = k_expiry_seg_fused, a wave per frame does the stripe search and then its stripes (round 6:
                           stage -0.07 ms, timed step unchanged: not the default; profiles/r6_expiry_fused_stripes_ab.log) */
#define DMZ_XSEG_FUSED 0
#endif
#ifndef DMZ_XSEG_FOLD  /* developer switch: 0 = the slash MLP always on Scharr samples */
#define DMZ_XSEG_FOLD 1
#endif
// developer probe (-DDMZ_XSEG_TL; tools/dev/xseg_tl.py): cycles per phase summed over all waves
#ifdef DMZ_XSEG_TL
__device__ unsigned long long g_xs_tl[16];
#define XS_TL(i) { const long long tl_now = (long long)__builtin_readcyclecounter(); tl_acc[i] += tl_now - tl_last; tl_last = tl_now; }
#else
#define XS_TL(i)
#endif
#define XSEG_STOP(k, expr)                     \
  if (DMZ_XSEG_STOP == (k)) {                  \
    if (lane == 0) sg->n = (int)(expr) & 0;    \
    return;                                    \
  }
// (developer timing of the pick, -DDMZ_XSEG_DBG: the counters g_xs_dbg[] live in dmz_stdsort.h; tools/dev/xseg_dbg.py)
constexpr int IROWS = 23;    // inter rows: image rows base-4 .. base+18 (clamped to the ROI)
constexpr int ISTRIDE = 428; // bytes per inter row (107 dwords)
constexpr int XT_PITCH = 20; // bytes per row of a thresholded character tile (19 used)
constexpr int kMaxRects = 80;
struct SegLds {
  // (padded to a multiple of 16 bytes: the compiler merges the column sums' stores into ds_write_b128, and a b64 / b128
  // LDS access off its natural alignment is replayed at 64 cycles per instruction -- SQ_LDS_UNALIGNED_STALL)
  __attribute__((aligned(16))) unsigned char inter[IROWS * ISTRIDE + 12];  // 9,856 B
  __attribute__((aligned(16))) int colB[428];                         // column sums over rows base-1 .. base+15 (regrid_group)
  // three tenants, one after the other (a single wave: its LDS operations execute in program order)
  union {
    int colA[428];                       // column sums over rows base .. base+16, until the rect sums are in registers
    struct {                             // the candidate order of the pick when equal sums matter (dmz_stdsort.h)
      unsigned v[420];
      unsigned stack[36];
    } s;
    struct {                             // the picked rects sorted by left, until the local groups are formed
      int itemS[64];
      short itemL[64];
      short gstart[66];
    } a;
    struct {                             // per group
      // thresholded tiles of optimize_character_rects, [slot][row 21][XT_PITCH]: rows start dword-aligned and the row
      // sums read aligned dwords (the compiler had turned the byte loop into ds_read_u16 at odd addresses: those and the
      // misaligned b128 above were SQ_LDS_UNALIGNED_STALL = 1.85 x the kernel's active LDS cycles in round 2)
      __attribute__((aligned(4))) unsigned char tile[3 * 21 * XT_PITCH + 4];
      // 16-bit: positions and widths < 432, sums of at most 21 bytes (the workgroup's LDS decides how
      // many stripes a CU holds, and the kernel is latency-bound)
      short rL[64];                      // their left edges
      // optimised character rects of ALL the stripe's groups, one after the other: at most nine groups of at most
      // (width + 36) / 11 + 1 regridded rects each, widths adding up to <= 428: 77
      short cLeft[kMaxRects], cTop[kMaxRects];
      unsigned char cand[kMaxRects];     // slash candidates: index of a window's middle character in cLeft / cTop
      // per slot of optimize_character_rects: 24 entries apart (21 used), so that the wide reads the compiler
      // merges a slot's 18 consecutive entries into start 16-byte aligned (a 42-byte slot pitch made them
      // unaligned ds_read_b128: SQ_LDS_UNALIGNED_STALL was twice the kernel's LDS busy cycles).  Column maxima,
      // then column sums, then row sums.
      __attribute__((aligned(16))) short cm[80];
    } b;
  } u;
  // surviving local groups (a group is at least four 9-px rects and groups are a rect apart: at most nine fit 428 columns)
  short gL[16], gW[16];
  int stripe_row[4], stripe_sum[4];  // the fused kernel's stripes (k_expiry_seg_fused): kept here, not in registers, across a stripe's body
};
static_assert(sizeof(SegLds) <= 13648, "twelve stripes per CU");

// strip_group_white_space (expiry_seg.cpp:101-129) on the index range [s, e) of a sum array
__device__ __forceinline__ void strip_white_space(const int *__restrict__ sums, int &s, int &e) {
  while (e - s > 5) {
    const int idx = s + (e - s - 4) / 2;
    const long long q = ((long long)sums[idx] + sums[idx + 1] + sums[idx + 2] + sums[idx + 3]) / 4;
    const long long thr = (long long)((double)q * 0.8);
    if ((long long)sums[s] < thr) s++;
    else if ((long long)sums[e - 1] < thr) e--;
    else break;
  }
}

// the same on sums held one per lane (lane k = sums[k]); s and e are wave-uniform
__device__ __forceinline__ void strip_white_space_lanes(int mine, int &s, int &e) {
  auto at = [&](int i) { return (long long)__builtin_amdgcn_readlane(mine, __builtin_amdgcn_readfirstlane(i)); };
  while (e - s > 5) {
    const int idx = s + (e - s - 4) / 2;
    const long long q = (at(idx) + at(idx + 1) + at(idx + 2) + at(idx + 3)) / 4;
    const long long thr = (long long)((double)q * 0.8);
    if (at(s) < thr) s++;
    else if (at(e - 1) < thr) e--;
    else break;
  }
}

// the value cvNormalize(255, CV_C) + cvThreshold(100, TOZERO) leave for a Scharr sample
__device__ __forceinline__ int norm_thresh(int v, float scale) {
#if DMZ_XSEG_NT_FAST
  // Round 6: the same value on full-rate instructions.  x + 1.5 * 2^23 rounds x to the nearest integer, ties
  // to even, for |x| < 2^22 -- exactly v_rndne_f32 + v_cvt_i32_f32 -- and leaves it in the sum's low mantissa bits; the
  // threshold is a sign mask instead of a compare + select: multiply, add, three integer subtractions / shifts, an AND (19.6
  // issue cycles per sample by profiles/r4_valu_table_gfx950.txt) where convert, multiply, round, convert, compare, select
  // took 23.4.  Bit-exact by construction (no magic multiplier to prove).
  const float y = (float)v * scale + 12582912.0f;
  const int iv = __float_as_int(y) - 0x4B400000;
  return iv & ((100 - iv) >> 31);
#else
  const int iv = __float2int_rn((float)v * scale);
  return iv > 100 ? iv : 0;
#endif
}

__device__ __forceinline__ float row16_sum(float x) {  // total of a 16-lane DPP row, in its lane 15
  x += __builtin_bit_cast(float, DMZ_DPP_SHR(__builtin_bit_cast(int, x), 1));
  x += __builtin_bit_cast(float, DMZ_DPP_SHR(__builtin_bit_cast(int, x), 2));
  x += __builtin_bit_cast(float, DMZ_DPP_SHR(__builtin_bit_cast(int, x), 4));
  x += __builtin_bit_cast(float, DMZ_DPP_SHR(__builtin_bit_cast(int, x), 8));
  return x;
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// tanh(x) = 1 - 2 / (exp(2x) + 1) on v_exp_f32 / v_rcp_f32: |error| ~ 2e-7 absolute
__device__ __forceinline__ float fast_tanh(float x) {
  const float e = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);  // 2 * log2(e)
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
}

// one (frame, stripe) by one wave
__device__ __forceinline__ void expiry_seg_stripe(SegLds &L, const float *__restrict__ wts, const float *__restrict__ xw,
                                                  const uint8_t *__restrict__ card, const int n, const int y0, const int base,
                                                  const long long stripe_sum, DmzExpiryStage *__restrict__ sg, const int lane) {
  int n_emitted = 0;
#ifdef DMZ_XSEG_DBG
  const long long dbg_start = __builtin_readcyclecounter();
#endif
#ifdef DMZ_XSEG_TL
  long long tl_last = (long long)__builtin_readcyclecounter();
  long long tl_acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
  // window rows k = 0..20 <-> image rows base-3+k; the Scharr image is zero outside [y0, 269]
  unsigned vmask = 0u;
  for (int k = 0; k < 21; k++) {
    const int R = base - 3 + k;
    if (R >= y0 && R <= CH - 1) vmask |= 1u << k;
  }

  // ---- horizontal pass of rows base-4 .. base+18 (row index clamped to the ROI, sobel.cpp:765-766) AND the column sums
  // (456-486) in one go (round 6).  Lane l < 54 owns the dword PAIR (2 l, 2 l + 1) of every row: one 8-byte load per row, all
  // 23 in flight together; of the four neighbour dwords a pair needs, two are its own and two come from the adjacent lanes
  // (DPP wave shifts; the column clamp at 0 / 427, sobel.cpp:729-734, rides in the shifts' `old` operand and in one v_perm).
  // |p[c+1] - p[c-1]| on 16-bit fields: the left / right neighbours of a dword's even and odd columns are each ONE v_perm_b32
  // (or a mask) of (previous | this | next) dword, the absolute difference is two saturating v_pk_sub_u16 and an OR
  // (12 VALU per dword where the alignbyte / mask / negate / max form took 26).  The even / odd fields are exactly what the
  // column sums accumulate -- colA[c] = sum over window rows k = 3 .. 19 of the Scharr sample v_k (0 outside the ROI), colB
  // the same over k = 2 .. 18; v_k = 3 inter[k] + 10 inter[k + 1] + 3 inter[k + 2], so both are WEIGHTED SUMS OF THE INTER
  // ROWS with wave-uniform weights (<= 16) that fold the ROI mask in: one v_mad_u32_u24 per field pair, row and sum, ten rows
  // per accumulator (10 x 16 x 255 < 2^16), no LDS round trip.  Integer arithmetic throughout: the sums are exact. ----
  {
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    auto absdiff = [](uint32_t r, uint32_t l) {
      const u16x2 rv = __builtin_bit_cast(u16x2, r), lv = __builtin_bit_cast(u16x2, l);
      return __builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(rv, lv)) |
             __builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(lv, rv));
    };
    unsigned ca[24], cb[24];  // uniform (scalar registers)
#pragma unroll
    for (int r = 0; r < IROWS; r++) {
      unsigned wa = 0u, wb = 0u;
#pragma unroll
      for (int t = 0; t < 3; t++) {  // inter row r is tap t of sample k = r - t (weights 3, 10, 3)
        const int k = r - t;
        const unsigned wt = t == 1 ? 10u : 3u, m = (vmask >> (k < 0 ? 0 : k)) & 1u;
        if (k >= 3 && k <= 19) wa += wt * m;
        if (k >= 2 && k <= 18) wb += wt * m;
      }
      ca[r] = wa, cb[r] = wb;
    }
    // (buffer loads: the dword past a card's last row -- lane 53's second one, never used -- reads as zero)
    const __amdgpu_buffer_rsrc_t crs = __builtin_amdgcn_make_buffer_rsrc((void *)card, 0, CW * CH, 0x00020000);
    const int pl = lane < 54 ? lane : 53;
    u32x2 cc[IROWS];
#pragma unroll
    for (int t = 0; t < IROWS; t++) {
      const int rowc = imin(imax(base - 4 + t, y0), CH - 1);
      cc[t] = __builtin_amdgcn_raw_buffer_load_b64(crs, 8 * pl, rowc * CW, 0);
    }
    // lane 53's second dword does not exist: the column clamp wants its first dword's last byte there (byte 0)
    const uint32_t ysel = lane == 53 ? 0x0c0c0c07u : 0x03020100u;
    // Lane 53's second dword does not exist: it is stored as a second copy of its first, at the first one's address (one
    // predicate for both stores of a row; lanes past 53 store nothing).
    const bool two = lane < 53, live = lane < 54;
    unsigned char *const px = L.inter + 8 * pl, *const py = px + (two ? 4 : 0);
    uint32_t aE[2][2] = {{0u, 0u}, {0u, 0u}}, aO[2][2] = {{0u, 0u}, {0u, 0u}}, bE[2][2] = {{0u, 0u}, {0u, 0u}}, bO[2][2] = {{0u, 0u}, {0u, 0u}};
#pragma unroll
    for (int t = 0; t < IROWS; t++) {
      const uint32_t x = cc[t].x, y = __builtin_amdgcn_perm(x, cc[t].y, ysel);
      // dword 2 l - 1 (lane 0: the clamp, p[-1] = p[0], as byte 3) and dword 2 l + 2 (lanes past 52: unused)
      const uint32_t px_ = (uint32_t)__builtin_amdgcn_update_dpp((int)(x << 24), (int)y, 0x138, 0xf, 0xf, false);
      const uint32_t ny = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x130, 0xf, 0xf, true);
      // even columns (0, 2) / odd columns (1, 3) of a dword as two 16-bit fields: left = p[c - 1], right = p[c + 1]
      const uint32_t xe = absdiff(__builtin_amdgcn_perm(x, x, 0x0c030c01u), __builtin_amdgcn_perm(x, px_, 0x0c050c03u));
      const uint32_t xo = absdiff(__builtin_amdgcn_perm(y, x, 0x0c040c02u), x & 0x00FF00FFu);
      const uint32_t ye = absdiff(__builtin_amdgcn_perm(y, y, 0x0c030c01u), __builtin_amdgcn_perm(y, x, 0x0c050c03u));
      const uint32_t yo = absdiff(__builtin_amdgcn_perm(ny, y, 0x0c040c02u), y & 0x00FF00FFu);
      const uint32_t ox = xe | (xo << 8), oy = ye | (yo << 8);
      if (live) {
        *(uint32_t *)(px + t * ISTRIDE) = ox;
        *(uint32_t *)(py + t * ISTRIDE) = two ? oy : ox;
      }
      if (t >= 2 && t <= 21) {
        const int g = t >= 12;
        aE[0][g] = __umul24(xe, ca[t]) + aE[0][g], aO[0][g] = __umul24(xo, ca[t]) + aO[0][g];
        bE[0][g] = __umul24(xe, cb[t]) + bE[0][g], bO[0][g] = __umul24(xo, cb[t]) + bO[0][g];
        aE[1][g] = __umul24(ye, ca[t]) + aE[1][g], aO[1][g] = __umul24(yo, ca[t]) + aO[1][g];
        bE[1][g] = __umul24(ye, cb[t]) + bE[1][g], bO[1][g] = __umul24(yo, cb[t]) + bO[1][g];
      }
    }
    {
      int *const qa = L.u.colA + 8 * pl, *const qb = L.colB + 8 * pl;
      int sa[2][4], sb[2][4];
#pragma unroll
      for (int h = 0; h < 2; h++) {
        sa[h][0] = (int)((aE[h][0] & 0xffffu) + (aE[h][1] & 0xffffu)), sa[h][1] = (int)((aO[h][0] & 0xffffu) + (aO[h][1] & 0xffffu));
        sa[h][2] = (int)((aE[h][0] >> 16) + (aE[h][1] >> 16)), sa[h][3] = (int)((aO[h][0] >> 16) + (aO[h][1] >> 16));
        sb[h][0] = (int)((bE[h][0] & 0xffffu) + (bE[h][1] & 0xffffu)), sb[h][1] = (int)((bO[h][0] & 0xffffu) + (bO[h][1] & 0xffffu));
        sb[h][2] = (int)((bE[h][0] >> 16) + (bE[h][1] >> 16)), sb[h][3] = (int)((bO[h][0] >> 16) + (bO[h][1] >> 16));
      }
      const int yo4 = two ? 4 : 0;
      if (live) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
          qa[k] = sa[0][k], qb[k] = sb[0][k];
          qa[yo4 + k] = two ? sa[1][k] : sa[0][k], qb[yo4 + k] = two ? sb[1][k] : sb[0][k];
        }
      }
    }
  }
  __syncthreads();
  XS_TL(0)
  XSEG_STOP(1, L.inter[lane])
  XS_TL(1)
  XSEG_STOP(2, L.u.colA[lane])
  // thresholds (expiry_seg.cpp:447-449, 488-494).  While the running total stays below 2^24 every
  // float addition of these integers is exact, so the float total equals the integer total whenever
  // that is < 2^24 (the common case); only beyond that the additions round and the reference's
  // column order has to be replayed.
  const float thr1 = (float)(((stripe_sum * SCW) / CW) / 5);
  float total;
  int cnt;
  // lane l < 60 owns the seven rect positions c = 7 l + j: the sliding 9-wide rect sums of columns c .. c + 8
  // (expiry_seg.cpp:456-486) come from fifteen column sums, and the neighbourhood of a position (+-8 columns) is
  // lanes l - 2 .. l + 2 -- what the parallel pick below works on
  int rs7[7];
  {
    int isum = 0, icnt = 0;
    int cv[15];
#pragma unroll
    for (int i = 0; i < 15; i++) cv[i] = lane < 60 ? L.u.colA[7 * lane + i] : 0;
    int run = 0;
#pragma unroll
    for (int i = 0; i < SCW; i++) run += cv[i];
#pragma unroll
    for (int j = 0; j < 7; j++) {
      rs7[j] = run;
      if (j < 6) run += cv[j + SCW] - cv[j];
      if (lane < 60 && (float)rs7[j] > thr1) isum += rs7[j], icnt++;
    }
    // rect sums are < 2^20 and there are <= 420 of them: the integer total fits 32 bits
    isum = wave_sum_i32(isum);
    cnt = wave_sum_i32(icnt);
    if ((unsigned)isum < (1u << 24)) {
      total = (float)isum;
    } else {
      total = 0.0f;
      for (int c = 0; c < CW - SCW + 1; c++) {
        int rs = 0;
        for (int k = 0; k < SCW; k++) rs += L.u.colA[c + k];
        const float sv = (float)rs;
        if (sv > thr1) total += sv;
      }
    }
  }
  if (cnt == 0) return;
  const float avg = total / (float)cnt;
  const float thr2 = (float)(0.8 * (double)avg);

  XS_TL(2)
  XSEG_STOP(3, thr2)
  // ---- greedy non-overlapping pick in descending sum order (expiry_seg.cpp:496-529), in parallel rounds: a
  // candidate that beats every live candidate within 8 columns is what the sequential scan would pick next in its
  // neighbourhood (keys are distinct: sum, then the smaller column), so all such local maxima are picked at once,
  // everything within 8 columns of a pick dies, and the rounds repeat until nothing is alive -- the same set as
  // the sequential greedy (the priority-ordered maximal independent set), in ~4 rounds instead of ~30 picks. ----
  // Equal sums: the reference visits them in std::sort's order (dmz_stdsort.h).  The rounds first run with the smaller column
  // as tie-break and watch for the one event where that order can change the outcome: a pick with a LIVE candidate of the same
  // sum within its eight columns (with this tie-break such a candidate lies to the right of the pick, and it is the best of its
  // right-hand neighbours) -- 15 % of the corpus' stripes.  Then the list is ordered as the library orders it and the rounds
  // repeat with the position after the library's partition phase as tie-break: the sequential greedy in the sorted order.
  unsigned key[7];
  unsigned picked = 0u;  // bit j: the rect at column 7 lane + j was picked
  unsigned tiemask = 0u;  // bit j: the pick at slot j had a live candidate of its own sum to its right (an OPEN tie)
  auto pick_rounds = [&](bool watch, unsigned resolved) -> bool {
    // lane - 1 / lane + 1 of the wave (DPP wave_shr:1 / wave_shl:1), 0 at the ends
    auto below = [](unsigned v) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xf, 0xf, true); };
    auto above = [](unsigned v) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xf, 0xf, true); };
    auto umax = [](unsigned a, unsigned b) { return a > b ? a : b; };
    tiemask = 0u;
    picked = 0u;
    for (;;) {
      unsigned alive = key[0];
#pragma unroll
      for (int j = 1; j < 7; j++) alive |= key[j];
      if (__builtin_amdgcn_ballot_w64(alive != 0u) == 0ull) break;
      unsigned suf[7], pre[7];  // maxima of slots k .. 6 / 0 .. k
      suf[6] = key[6];
#pragma unroll
      for (int k = 5; k >= 0; k--) suf[k] = umax(key[k], suf[k + 1]);
      pre[0] = key[0];
#pragma unroll
      for (int k = 1; k < 7; k++) pre[k] = umax(key[k], pre[k - 1]);
      unsigned now = 0u;
#pragma unroll
      for (int j = 0; j < 7; j++) {
        // columns 7 l + j - 8 .. 7 l + j + 8: slots j - 1 .. 6 of lane l - 1, all of lane l, slots 0 .. j + 1 of lane
        // l + 1, and for the end slots one column of lane l -+ 2
        unsigned right = above(pre[j < 6 ? j + 1 : 6]);  // the columns to the right outside the own lane
        if (j == 6) right = umax(right, above(above(key[0])));
        unsigned w = umax(suf[0], below(suf[j > 0 ? j - 1 : 0]));
        w = umax(w, right);
        if (j == 0) w = umax(w, below(below(key[6])));
        const bool pick = key[j] != 0u && key[j] == w;
        now |= pick ? 1u << j : 0u;
        if (watch) {
          if (j < 6) right = umax(right, suf[j + 1]);
          // (a dead neighbour's key is 0: its sum field differs from a live one's; windows whose order among equal sums is
          // already the library's -- `resolved` -- are settled)
          tiemask |= (pick && ((right ^ key[j]) >> 9) == 0u && !((resolved >> j) & 1u)) ? 1u << j : 0u;
        }
      }
      picked |= now;
      // bit i of `near`: column 7 l - 8 + i holds a pick of this round (i = 0 .. 22)
      const unsigned p1m = below(now), p1p = above(now);
      const unsigned near = ((below(p1m) >> 6) & 1u) | (p1m << 1) | (now << 8) | (p1p << 15) | ((above(p1p) & 1u) << 22);
#pragma unroll
      for (int j = 0; j < 7; j++) key[j] = ((near >> j) & 0x1FFFFu) ? 0u : key[j];
    }
    return __builtin_amdgcn_ballot_w64(tiemask != 0u) != 0ull;
  };
#pragma unroll
  for (int j = 0; j < 7; j++) {
    key[j] = 0u;
    if (lane < 60 && (float)rs7[j] > thr1 && (float)rs7[j] > thr2)
      key[j] = ((unsigned)rs7[j] << 9) | (unsigned)(511 - (7 * lane + j));
  }
#ifdef DMZ_XSEG_DBG
  const long long dbg_t0 = __builtin_readcyclecounter();
  const bool dbg_tie = pick_rounds(true, 0u);
  const long long dbg_t1 = __builtin_readcyclecounter();
  if (lane == 0) {
    atomicAdd(&g_xs_dbg[0], 1ull);
    atomicAdd(&g_xs_dbg[1], dbg_tie ? 1ull : 0ull);
    atomicAdd(&g_xs_dbg[2], (unsigned long long)(dbg_t1 - dbg_t0));
  }
  if (dbg_tie) {
#else
  if (pick_rounds(DMZ_XSEG_TIES >= 1, 0u) && DMZ_XSEG_TIES >= 2 && (DMZ_XSEG_TIES != 3 || n == -12345)) {  // (3: the code is there, never run)
#endif
    // An open tie: the order of equal sums matters.  It is settled LEVEL BY LEVEL from the top: the greedy walks the sums in
    // descending order, so everything above the highest open tie is already what the reference picks; the windows of that
    // sum are MARKED, the library's order is computed for the marked windows (dmz_stdsort.h follows only the ranges that hold
    // two of them), and the rounds repeat with it as their tie-break -- still watching: an open tie further down (one run in five)
    // adds its level to the marks.  After three levels whatever could ever tie is marked at once: candidates with an
    // equal-sum candidate within eight columns (only such a pair can be a pick and its live neighbour).
    unsigned cs[7];  // candidate sums (0: not a candidate; a candidate's sum is >= 1)
#pragma unroll
    for (int j = 0; j < 7; j++) cs[j] = (lane < 60 && (float)rs7[j] > thr1 && (float)rs7[j] > thr2) ? (unsigned)rs7[j] : 0u;
    auto near_tie_marks = [&]() -> unsigned {
      auto below = [](unsigned v) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xf, 0xf, true); };
      auto above = [](unsigned v) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xf, 0xf, true); };
      unsigned m = 0u, nx[7];
#pragma unroll
      for (int j = 0; j < 7; j++) nx[j] = above(cs[j]);  // the next lane's
      const unsigned nx2 = above(nx[0]);                  // column 7 (lane + 2)
      unsigned fwd1 = 0u, fwd2 = 0u;                      // partners found in the next lane's slots / in slot 0 of the lane after it
#pragma unroll
      for (int j = 0; j < 7; j++) {
        if (cs[j] == 0u) continue;
#pragma unroll
        for (int k = j + 1; k < 7; k++)  // columns c + 1 .. within the own lane
          if (cs[k] == cs[j]) m |= (1u << j) | (1u << k);
#pragma unroll
        for (int k = 0; k <= (j < 6 ? j + 1 : 6); k++)  // columns 7 (lane + 1) + k <= c + 8
          if (nx[k] == cs[j]) m |= 1u << j, fwd1 |= 1u << k;
        if (j == 6 && nx2 == cs[j]) m |= 1u << j, fwd2 = 1u;
      }
      return m | below(fwd1) | below(below(fwd2));
    };
    // rect_list: the windows above the first threshold in column order (expiry_seg.cpp:461-470); std::sort (:496).
    // The partition phase needs a table of 210 dwords beside the list: the first 210 column sums of colB (kept for
    // regrid_group) wait in registers meanwhile.
    unsigned *const sv = L.u.s.v;
    unsigned *const tb = (unsigned *)L.colB;
    int keep[4];
#pragma unroll
    for (int i = 0; i < 4; i++) keep[i] = lane + 64 * i < dmzsort::TPAIRS ? L.colB[lane + 64 * i] : 0;
    unsigned mark = 0u;
    auto fill_list = [&](bool marks) {
      unsigned pos = 0u;
#pragma unroll
      for (int j = 0; j < 7; j++) {
        const unsigned long long bal = __builtin_amdgcn_ballot_w64(lane < 60 && (float)rs7[j] > thr1);
        pos = __builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, pos));
      }
#pragma unroll
      for (int j = 0; j < 7; j++)
        if (lane < 60 && (float)rs7[j] > thr1)
          sv[pos++] = ((unsigned)rs7[j] << 9) | (unsigned)(7 * lane + j) | ((marks && ((mark >> j) & 1u)) ? dmzsort::MARK : 0u);
      __syncthreads();
    };
    for (int level = 0;; level++) {
      bool last = level >= 3;
      if (!last) {
        unsigned lv = 0u;  // the highest sum with an open tie
#pragma unroll
        for (int j = 0; j < 7; j++) lv = ((tiemask >> j) & 1u) && cs[j] > lv ? cs[j] : lv;
        lv = wave_max_u32(lv);
#pragma unroll
        for (int j = 0; j < 7; j++) mark |= (cs[j] == lv) ? 1u << j : 0u;
      } else {
        mark |= near_tie_marks();
      }
      fill_list(true);
      if (!dmzsort::wave_mark_partitions<9, 0xFFFFFu>(sv, cnt, lane, tb, L.u.s.stack)) {
        // the depth limit of the introsort loop (adversarial lists only): the library's whole sort on one lane, every
        // window then carries its final position
        __syncthreads();
        fill_list(false);
        if (lane == 0) dmzsort::serial_sort<9>(sv, cnt, L.u.s.stack);
        mark = 0x7Fu;
        last = true;
      }
      __syncthreads();
      unsigned short *const sq = (unsigned short *)tb;  // position of a column's window in that order (marked windows)
      for (int p = lane; p < cnt; p += 64) {
        const unsigned el = sv[p];
        sq[el & 511u] = (unsigned short)p;
      }
      __syncthreads();
#pragma unroll
      for (int j = 0; j < 7; j++)
        key[j] = cs[j] ? (cs[j] << 9) | (511u - (((mark >> j) & 1u) ? (unsigned)sq[7 * lane + j] : (unsigned)(7 * lane + j))) : 0u;
      __syncthreads();
      if (!pick_rounds(!last, mark) || last) {
#ifdef DMZ_XSEG_DBG
        if (lane == 0) atomicAdd(&g_xs_dbg[4], (unsigned long long)(level + 1)), atomicAdd(&g_xs_dbg[7], last ? 1ull : 0ull);
#endif
        break;
      }
    }
#pragma unroll
    for (int i = 0; i < 4; i++)
      if (lane + 64 * i < dmzsort::TPAIRS) L.colB[lane + 64 * i] = keep[i];
#ifdef DMZ_XSEG_DBG
    if (lane == 0) atomicAdd(&g_xs_dbg[3], (unsigned long long)(__builtin_readcyclecounter() - dbg_t1));
#endif
  }
  XS_TL(3)
  XSEG_STOP(4, picked)
  // sorted by left = column order = lane-major, slot-minor: a lane's first item follows the picks of the lanes below
  int n_items = 0;
  {
    int pos = 0;
#pragma unroll
    for (int j = 0; j < 7; j++) {
      const unsigned long long bal = __ballot((picked >> j) & 1u);
      pos += __popcll(bal & lanemask_lt(lane));
      n_items += __popcll(bal);
    }
#pragma unroll
    for (int j = 0; j < 7; j++)
      if ((picked >> j) & 1u) {
        L.u.a.itemL[pos] = (short)(7 * lane + j);
        L.u.a.itemS[pos] = rs7[j];
        pos++;
      }
  }
  __syncthreads();
  if (n_items == 0) return;

  // ---- gather_into_groups (131-167): chain while the gap is < 9; then strip white space ----
  int n_groups;
  {
    const int myL = lane < n_items ? L.u.a.itemL[lane] : 0;
    const int prevL = __shfl_up(myL, 1, 64);
    const bool boundary = lane < n_items && (lane == 0 || myL - (prevL + SCW) >= SCW);
    const unsigned long long bal = __ballot(boundary);
    n_groups = __popcll(bal);
    if (boundary) L.u.a.gstart[__popcll(bal & lanemask_lt(lane))] = lane;
    if (lane == 0) L.u.a.gstart[n_groups] = n_items;
  }
  __syncthreads();
  int G;
  {
    int s = 0, e = 0;
    bool keep = false;
    if (lane < n_groups) {
      s = L.u.a.gstart[lane];
      e = L.u.a.gstart[lane + 1];
      strip_white_space(L.u.a.itemS, s, e);
      keep = e - s >= 4;  // kMinimumExpiryStripCharacters - 1 (expiry_seg.cpp:566-571)
    }
    const unsigned long long bal = __ballot(keep);
    G = __popcll(bal);
    if (keep) {
      const int pos = __popcll(bal & lanemask_lt(lane));
      L.gL[pos] = L.u.a.itemL[s];
      L.gW[pos] = L.u.a.itemL[e - 1] + SCW - L.u.a.itemL[s];
    }
  }
  __syncthreads();  // u.a is dead from here on; u.b takes its place

  XS_TL(4)
  XSEG_STOP(5, G)
  const int g_top = base - 1;  // expanded stripe top; group height 17
  int rbase = 0, ncand = 0;  // rects / slash candidates of the groups so far
  for (int g = 0; g < G; g++) {
    XS_TL(7)
    // ---- regrid_group (169-229) ----
    const int left = L.gL[g], width = L.gW[g];
    const int bl = imax(left - 2 * SCW, 0), br = imin(left + width + 2 * SCW, CW);
    const int bw = br - bl;
    const int min_lines = (int)floorf((float)bw / 11.0f);
    int gs = 0;
    for (int c = bl + lane; c < br; c += 64) gs += L.colB[c];
    const float group_sum = (float)wave_sum_i32(gs);
    // The 65 hypotheses (spacing 11 .. 15, starting column < spacing): lane q scores hypothesis q, lane 0 also the 65th.  The
    // reference adds the grid lines' column sums in float, but they are integers (<= 17 x 4 080) and at most 39 of them:
    // every partial sum stays below 2^24, so the float sum IS the integer sum in any order -- four lines per trip with their
    // LDS reads in flight together instead of a read, a wait and a conversion per line (the loop was ~5 k cycles of LDS
    // round trips per group, -DDMZ_XSEG_TL).
    unsigned long long best = ~0ull;
    {
      int sp, so;
      if (lane < 11) sp = 11, so = lane;
      else if (lane < 23) sp = 12, so = lane - 11;
      else if (lane < 36) sp = 13, so = lane - 23;
      else if (lane < 50) sp = 14, so = lane - 36;
      else sp = 15, so = lane - 50;
      int acc = 0, acc64 = 0;
      const int trips = ((bw + 10) / 11 + 3) >> 2;  // the most lines any hypothesis has: spacing 11 from column 0
      int off = so, off64 = 14;
      for (int t = 0; t < trips; t++) {
        int v[4], w[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
          v[u] = L.colB[bl + imin(off + u * sp, bw - 1)];
          w[u] = L.colB[bl + imin(off64 + u * 15, bw - 1)];
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
          acc += off + u * sp < bw ? v[u] : 0;
          acc64 += off64 + u * 15 < bw ? w[u] : 0;
        }
        off += 4 * sp;
        off64 += 60;
      }
      auto consider = [&](int q, int lines_from, int spacing, int sum) {
        const int nl = lines_from < bw ? (bw - 1 - lines_from) / spacing + 1 : 0;
        float gls = (float)sum;
        const float average = gls / (float)nl;
        gls = average * (float)min_lines;
        const float ratio = gls / (group_sum - gls);
        if (ratio < FLT_MAX) {  // false for NaN / inf: such a candidate never replaces the best
          unsigned u = __float_as_uint(ratio);
          u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
          const unsigned long long k = ((unsigned long long)u << 32) | (unsigned)q;
          best = k < best ? k : best;
        }
      };
      consider(lane, so, sp, acc);
      if (lane == 0) consider(64, 14, 15, acc64);
    }
    best = wave_min_u64(best);
    int sp = 11, so = 0;
    if (best != ~0ull) {
      const int q = (int)(best & 0xFFFFFFFFull);
      if (q < 11) sp = 11, so = q;
      else if (q < 23) sp = 12, so = q - 11;
      else if (q < 36) sp = 13, so = q - 23;
      else if (q < 50) sp = 14, so = q - 36;
      else sp = 15, so = q - 50;
    }
    int nR;
    int my_sum = 0;  // sum of the regridded rect `lane`
    {
      const int off = so + lane * sp;
      const bool ok = off + 1 < bw;
      if (ok) {
        const int cend = imin(off + sp, bw);
        // (at most fourteen columns; all reads in flight together instead of one LDS round trip per column)
        int cv[14];
#pragma unroll
        for (int u = 0; u < 14; u++) cv[u] = L.colB[bl + imin(off + 1 + u, bw - 1)];
#pragma unroll
        for (int u = 0; u < 14; u++) my_sum += off + 1 + u < cend ? cv[u] : 0;
        L.u.b.rL[lane] = bl + off + 1;
      }
      nR = __popcll(__ballot(ok));
    }
    __syncthreads();
    const int cw = sp - 1;
    int rs = 0, re = nR;
    strip_white_space_lanes(my_sum, rs, re);
    XS_TL(5)
    if (DMZ_XSEG_STOP == 6) continue;

    // ---- optimize_character_rects (231-339), round 6: LAZILY.  A rect is dropped for its position alone (:259-266) and
    // every rect is trimmed on its own, so the group's rect COUNT is known here, and the trimmed positions are needed of the
    // slash candidates only (the middle characters of the windows of five, :643) -- and, for a window whose middle character
    // turns out to be a slash, of its other four.  Here the rects are listed untrimmed (cLeft = left of the expanded image,
    // cTop = minus its width: pending); the trimming runs once over the stripe's candidates, and after the slash search over
    // what the hits still need.  The synthetic corpus: 17.3 rects per stripe, 8 of them candidates. ----
    const int ciw = cw + 4, cih = 17 + 4;
    {
      const int rect_left = (lane >= rs && lane < re) ? L.u.b.rL[lane] - 2 : 0;
      const bool keep = lane >= rs && lane < re && !(rect_left < 0 || rect_left + ciw > CW || (g_top - 2) + cih > CH);
      const unsigned long long kbal = __ballot(keep);
      const int n2 = __popcll(kbal);
      const int rank = __popcll(kbal & lanemask_lt(lane));
      if (keep && rbase + rank < kMaxRects) {  // (always: see SegLds)
        L.u.b.cLeft[rbase + rank] = (short)rect_left;
        L.u.b.cTop[rbase + rank] = (short)-ciw;
      }
      // kMinimumExpiryStripCharacters (expiry_seg.cpp:617-623); the windows of five of this group: middle characters 2 .. n2 - 3
      if (n2 >= 5 && DMZ_XSEG_STOP != 7 && rbase + n2 <= kMaxRects) {
        if (lane >= 2 && lane < n2 - 2) L.u.b.cand[ncand + lane - 2] = (unsigned char)(rbase + lane);
        ncand += n2 - 4;
      }
      rbase += n2;
    }
    __syncthreads();
    XS_TL(6)
  }
  // three rects per pass, 21 lanes each.  Lane c of a slot owns column c of the 21-row window: one column of Scharr samples
  // in registers serves the max, the normalise+threshold and the column sum; the row sums come from a transposed read of
  // the thresholded tile (lane r = row r).  `idx` (per lane, uniform within a slot): the slot's rect in cLeft / cTop, < 0 none.
  auto optimize_batch = [&](const int idx) {
    const int sl = lane / 21, c = lane - sl * 21;
    unsigned char *tile = L.u.b.tile;  // [3][21][19]
    constexpr int cih = 17 + 4;
    const bool have = sl < 3 && idx >= 0;
    const int rect_left = have ? L.u.b.cLeft[idx] : 0;
    const int ciw = have ? -L.u.b.cTop[idx] : 0;
    const bool col = have && c < ciw;
    int v[21];
    int mx = 0;
    {
      // (idle lanes -- columns past the window, the slot-less lane 63 -- read column 0 and are zeroed once below: a
      // predicated load per row was a v_mov, an exec save and an exec restore each)
      int iv[IROWS];
      const int rl = col ? rect_left + c : 0;
#pragma unroll
      for (int t = 0; t < IROWS; t++) iv[t] = (int)L.inter[t * ISTRIDE + rl];
      if (vmask == 0x1FFFFFu) {  // (wave-uniform) the usual case: all 21 window rows inside the ROI, no per-row select
#pragma unroll
        for (int r = 0; r < 21; r++) {
          v[r] = 3 * (iv[r] + iv[r + 2]) + 10 * iv[r + 1];
          mx = imax(mx, v[r]);
        }
      } else {
#pragma unroll
        for (int r = 0; r < 21; r++) {
          v[r] = ((vmask >> r) & 1u) ? 3 * (iv[r] + iv[r + 2]) + 10 * iv[r + 1] : 0;
          mx = imax(mx, v[r]);
        }
      }
    }
    const int sidx = sl * 24 + c;  // this lane's entry of the per-slot arrays
    if (!col) {
      mx = 0;
#pragma unroll
      for (int r = 0; r < 21; r++) v[r] = 0;
    }
    L.u.b.cm[sidx] = mx;
    __syncthreads();
    if (sl < 3) {
#pragma unroll
      for (int j = 0; j < 18; j++) mx = imax(mx, L.u.b.cm[sl * 24 + j]);
    }
    // cvNormalize's scale is (float)(255.0 / (double)max); for every integer max in [1, 32767] that
    // equals the correctly rounded float quotient (checked exhaustively, tests/test_oracle_units.py)
    const float scale = mx > 0 ? 255.0f / (float)mx : 0.0f;
    int cs = 0;
    if (sl < 3) {
#pragma unroll
      for (int r = 0; r < 21; r++) {
        const int t = norm_thresh(v[r], scale);
        cs += t;
        if (c < 19) tile[(sl * 21 + r) * XT_PITCH + c] = t;
      }
    }
    L.u.b.cm[sidx] = cs;
    __syncthreads();
    // column trimming (every lane of the slot replays it: uniform within the slot)
    int lc = 0, rc = ciw - 1;
    if (sl < 3)
      for (int wv = ciw; wv > TW; wv--) {
        if (L.u.b.cm[sl * 24 + lc] <= L.u.b.cm[sl * 24 + rc]) lc++;
        else rc--;
      }
    int rsm = 0;
    if (sl < 3) {  // lane c is row c here: the eleven bytes lc .. lc + 10 (= rc) of its row, from four aligned dwords
      typedef const volatile __attribute__((address_space(3))) uint32_t *lds_vu32;  // (volatile: no merging into b64 / b128)
      const lds_vu32 rowp = (lds_vu32)(tile + (sl * 21 + c) * XT_PITCH + (lc & ~3));
      const uint32_t w0 = rowp[0], w1 = rowp[1], w2 = rowp[2], w3 = rowp[3];
      const uint32_t sh = (uint32_t)(lc & 3);
      rsm = (int)__builtin_amdgcn_sad_u8(__builtin_amdgcn_alignbyte(w1, w0, sh), 0u, 0u);
      rsm = (int)__builtin_amdgcn_sad_u8(__builtin_amdgcn_alignbyte(w2, w1, sh), 0u, (uint32_t)rsm);
      rsm = (int)__builtin_amdgcn_sad_u8(__builtin_amdgcn_alignbyte(w3, w2, sh) & 0x00FFFFFFu, 0u, (uint32_t)rsm);
    }
    __syncthreads();
    L.u.b.cm[sidx] = rsm;  // row sums
    __syncthreads();
    if (have && c == 0) {  // the slot's first lane walks the rows and files the trimmed rect
      int tr = 0, brw = cih - 1;
      for (int hv = cih; hv > TH; hv--) {
        if (L.u.b.cm[sl * 24 + tr] <= L.u.b.cm[sl * 24 + brw]) tr++;
        else brw--;
      }
      L.u.b.cLeft[idx] = (short)(rect_left + lc);
      L.u.b.cTop[idx] = (short)((g_top - 2) + tr);
    }
    __syncthreads();
  };
  for (int b0 = 0; b0 < ncand; b0 += 3) {
    const int k = b0 + lane / 21;
    optimize_batch(lane < 63 && k < ncand ? (int)L.u.b.cand[k] : -1);
  }
  XS_TL(7)
  {
    // ---- slash search (643-674): character first+2 of every window of five, sixteen candidates
    // per pass -- over the candidates of ALL the stripe's groups (round 3: a pass per group fetched the 90 KB of weight
    // fragments from L2 and built / evaluated sixteen rows for the two or three candidates a group has; a stripe's groups
    // together rarely exceed sixteen).  applym_730c4cbd (176 -> 80 tanh -> 2 softmax): the hidden layer is a
    // [16 x 176] x [176 x 80] product.  On v_mfma_f32_16x16x32_bf16 with EXACT operand splits: a Scharr
    // sample is an integer <= 4080 = 256 a + b, and 256 a and b are both bf16 numbers; the weights,
    // pre-divided by 255 (the reference's x = s * (1/255) differs from that by one float rounding
    // per input), are split into three bf16 parts (24 bits).  Six matrix instructions per (k-step of 32,
    // tile) reproduce the fp32 product to ~2^-24 per term at 1/2.7 of the fp32 matrix-core time.
    // A[m = lane & 15][k' = 32 ks + 8 (lane >> 4) + e] is candidate m's sample (row 8 (kk & 1) + e, column 2 ks + (kk >> 1));
    // D[row = candidate][col = hidden unit] comes back as 4 candidates x 5 tiles per lane.  The output
    // layer is a DPP row reduction over the 16 lanes that share a candidate.
    // Each k-step: the 15 weight fragments are requested (L2), the A fragments are built meanwhile, all
    // loads are waited for, and only then the 30 matrix instructions issue (see the conv2 loop below for
    // why no load stays in flight across them).
    for (int k0 = 0; k0 < ncand; k0 += 16) {
      const int nc = imin(16, ncand - k0);
      const int m = lane & 15, kk = lane >> 4;
      const bool live = m < nc;
      const int ci = live ? (int)L.u.b.cand[k0 + m] : 0;
      const int pl = live ? L.u.b.cLeft[ci] : 0, pt = live ? L.u.b.cTop[ci] - (base - 3) : 0;
      // (buffer loads: descriptor + 32-bit lane offset + scalar fragment offset, no 64-bit address arithmetic)
      const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
          (void *)(xw + dmzx::SLASH_B3), 0, 3 * dmzx::SLASH_KSTEPS * 5 * 64 * 16, 0x00020000);
      const int r0 = pt + 8 * (kk & 1);  // first window row of this lane's samples (r0 + 9 <= 22)
      const unsigned char *ip = L.inter + r0 * ISTRIDE + pl + (kk >> 1);
      int rowmask[8];                    // all-ones where the sample's row lies inside the ROI (and the lane is live)
#pragma unroll
      for (int e = 0; e < 8; e++) rowmask[e] = live ? __builtin_amdgcn_sbfe((int)vmask, r0 + e, 1) : 0;
      f32x4 acc[5];
#pragma unroll
      for (int t = 0; t < 5; t++) acc[t] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
      // ---- the common case: every candidate's sixteen sample rows lie inside the ROI.  A Scharr sample is then the fixed
      // combination 3 / 10 / 3 of three `inter` bytes, so the vertical pass folds into the weights (dmzx::SLASH_F3: W' =
      // W (x) [3 10 3] / 255 over the 18 x 11 `inter` bytes under the window, computed in double on the host) and the A
      // operand is the `inter` bytes THEMSELVES -- a byte zero-extended to 16 bits is the bf16 number d x 2^-133 (see
      // k_vseg), the weights carry 2^100, the 2^33 left over comes back before the tanh.  Per k-step: eight byte reads and
      // four packs instead of ten reads, 24 operations of sample arithmetic, 32 of splitting and converting and 8 packs; one
      // exact A part instead of two (15 matrix instructions instead of 30); seven k-steps instead of six (K = 16 x 12 as
      // before + one step for `inter` rows 16, 17).  The products are exact, the sums differ from the sample form by fp32
      // rounding (~1e-7 of the pre-activation): the decision P > 0.7 moves only within float noise of the threshold.
      // Windows that touch the ROI edge (a stripe at the very top or bottom) keep the sample form below. ----
      const bool inside = !live || ((vmask >> pt) & 0xFFFFu) == 0xFFFFu;
      const bool folded = DMZ_XSEG_FOLD && __builtin_amdgcn_ballot_w64(!inside) == 0ull;
      if (folded) {
        const __amdgpu_buffer_rsrc_t frs = __builtin_amdgcn_make_buffer_rsrc(
            (void *)(xw + dmzx::SLASH_F3), 0, 3 * dmzx::SLASH_FSTEPS * 5 * 64 * 16, 0x00020000);
        const unsigned char *ipa = ip;                                                       // rows r0 .. r0 + 7, column 2 ks + (kk >> 1)
        const unsigned char *ipb = L.inter + (pt + 16 + (kk >> 1)) * ISTRIDE + pl + 8 * (kk & 1);  // rows 16 / 17, columns 8 (kk & 1) ..
#pragma unroll 1
        for (int ks = 0; ks < (DMZ_XSEG_STOP == 8 ? 1 : dmzx::SLASH_FSTEPS); ks++) {
          typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
          u32x4 wb[3][5];
#pragma unroll
          for (int part = 0; part < 3; part++)
#pragma unroll
            for (int t = 0; t < 5; t++)
              wb[part][t] = __builtin_amdgcn_raw_buffer_load_b128(frs, lane * 16, ((part * dmzx::SLASH_FSTEPS + ks) * 5 + t) * 1024, 0);
          uint32_t by[8];
          if (ks < dmzx::SLASH_KSTEPS) {  // uniform
#pragma unroll
            for (int e = 0; e < 8; e++) by[e] = ipa[e * ISTRIDE];
            ipa += 2;
          } else {
#pragma unroll
            for (int e = 0; e < 8; e++) by[e] = ipb[e];
          }
          const u32x4 a = {by[0] | (by[1] << 16), by[2] | (by[3] << 16), by[4] | (by[5] << 16), by[6] | (by[7] << 16)};
          const bf16x8 av = __builtin_bit_cast(bf16x8, a);
          asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int t = 0; t < 5; t++) {  // small terms first
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, __builtin_bit_cast(bf16x8, wb[2][t]), acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, __builtin_bit_cast(bf16x8, wb[1][t]), acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, __builtin_bit_cast(bf16x8, wb[0][t]), acc[t], 0, 0, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int t = 0; t < 5; t++) acc[t] = acc[t] * 0x1p33f;  // (A carried 2^-133, B 2^100: exact)
      } else
#pragma unroll 1
      for (int ks = 0; ks < (DMZ_XSEG_STOP == 8 ? 1 : dmzx::SLASH_KSTEPS); ks++) {
        bf16x8 wb[3][5];
#pragma unroll
        for (int part = 0; part < 3; part++)
#pragma unroll
          for (int t = 0; t < 5; t++) wb[part][t] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(
                wrs, lane * 16, ((part * dmzx::SLASH_KSTEPS + ks) * 5 + t) * 1024, 0));
        // eight samples of this lane: k' = 32 ks + 8 kk + e <-> window column 2 ks + (kk >> 1), rows 8 (kk & 1) + e
        // (the K order is ours to choose -- capi.cpp lays the weights out to match): eight vertically adjacent
        // samples share ten bytes of one `inter` column.  Column 11 (ks = 5, kk >= 2) meets zero weights.
        int iv[10];
#pragma unroll
        for (int i = 0; i < 10; i++) iv[i] = ip[i * ISTRIDE];
        ip += 2;
        uint32_t hi[4], lo[4];
#pragma unroll
        for (int e2 = 0; e2 < 4; e2++) {
          float fh[2], fl[2];
#pragma unroll
          for (int h = 0; h < 2; h++) {
            const int e = 2 * e2 + h;
            const int sv = (3 * (iv[e] + iv[e + 2]) + 10 * iv[e + 1]) & rowmask[e];
            fh[h] = (float)(sv & ~255);                              // 256 a: a bf16 number
            fl[h] = (float)(sv & 255);                               // b (v_cvt_f32_ubyte0)
          }
          // the upper halves of the two floats = their (exact) bf16 forms
          hi[e2] = __builtin_amdgcn_perm(__float_as_uint(fh[1]), __float_as_uint(fh[0]), 0x07060302u);
          lo[e2] = __builtin_amdgcn_perm(__float_as_uint(fl[1]), __float_as_uint(fl[0]), 0x07060302u);
        }
        typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
        const bf16x8 ah = __builtin_bit_cast(bf16x8, (u32x4){hi[0], hi[1], hi[2], hi[3]});
        const bf16x8 al = __builtin_bit_cast(bf16x8, (u32x4){lo[0], lo[1], lo[2], lo[3]});
#ifndef DMZ_SLASH_NOFENCE  /* developer probe */
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
        for (int t = 0; t < 5; t++) {  // small terms first
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, wb[2][t], acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, wb[2][t], acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, wb[1][t], acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, wb[1][t], acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, wb[0][t], acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, wb[0][t], acc[t], 0, 0, 0);
        }
#ifndef DMZ_SLASH_NOFENCE
        __builtin_amdgcn_sched_barrier(0);
#endif
      }
      const float *sw = wts + dmzw::SLASH;
      float o0[4] = {0.0f, 0.0f, 0.0f, 0.0f}, o1[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
      for (int t = 0; t < 5; t++) {
        const int hn = 16 * t + m;
        const float b1 = sw[dmzw::S_B1 + hn], w20 = sw[dmzw::S_W2 + hn], w21 = sw[dmzw::S_W2 + 80 + hn];
#pragma unroll
        for (int v = 0; v < 4; v++) {
          const float h = DMZ_XSEG_STOP == 9 ? acc[t][v] + b1 : fast_tanh(acc[t][v] + b1);
          o0[v] = fmaf(w20, h, o0[v]);
          o1[v] = fmaf(w21, h, o1[v]);
        }
      }
      unsigned hits = 0u;  // bit v: candidate 4 kk + v is a slash (valid in lane 15 of each row)
#pragma unroll
      for (int v = 0; v < 4; v++) {
        const float e0 = expf(row16_sum(o0[v]) + sw[dmzw::S_B2 + 0]);
        const float e1 = expf(row16_sum(o1[v]) + sw[dmzw::S_B2 + 1]);
        if (e0 / (e0 + e1) > 0.7f) hits |= 1u << v;
      }
      const unsigned hitmask = (unsigned)__builtin_amdgcn_readlane((int)hits, 15) |
                               ((unsigned)__builtin_amdgcn_readlane((int)hits, 31) << 4) |
                               ((unsigned)__builtin_amdgcn_readlane((int)hits, 47) << 8) |
                               ((unsigned)__builtin_amdgcn_readlane((int)hits, 63) << 12);
      if (hitmask) {
        // the other four characters of the windows with a slash: whatever is still pending is trimmed now (rect indices are
        // < 80: lane i looks at rects i and i + 64; the list goes where the regridded lefts were)
        unsigned long long need0 = 0ull, need1 = 0ull;
        for (int q = 0; q < nc; q++)
          if ((hitmask >> q) & 1u) {
            const int first = (int)L.u.b.cand[k0 + q] - 2;  // (uniform)
            const unsigned long long five = 0x1Full;
            need0 |= first < 64 ? five << first : 0ull;
            need1 |= first >= 64 ? five << (first - 64) : (first > 59 ? five >> (64 - first) : 0ull);
          }
        const bool p0 = ((need0 >> lane) & 1ull) && L.u.b.cTop[lane] < 0;
        const bool p1 = lane < kMaxRects - 64 && ((need1 >> lane) & 1ull) && L.u.b.cTop[64 + lane] < 0;
        const unsigned long long bp0 = __ballot(p0), bp1 = __ballot(p1);
        const int np0 = __popcll(bp0), npend = np0 + __popcll(bp1);
        if (p0) L.u.b.rL[__popcll(bp0 & lanemask_lt(lane))] = (short)lane;
        if (p1) L.u.b.rL[np0 + __popcll(bp1 & lanemask_lt(lane))] = (short)(64 + lane);
        __syncthreads();
        for (int b0 = 0; b0 < npend; b0 += 3) {
          const int k = b0 + lane / 21;
          optimize_batch(lane < 63 && k < npend ? (int)L.u.b.rL[k] : -1);
        }
      }
      for (int q = 0; q < nc; q++) {
        if (!((hitmask >> q) & 1u)) continue;
        if (lane == 0 && n_emitted < DMZ_HIP_EXPIRY_MAX_GROUPS) {
          const int first = (int)L.u.b.cand[k0 + q] - 2;
          int top = L.u.b.cTop[first], gleft = L.u.b.cLeft[first], gwidth = SCW, gheight = SCH;
          for (int i = 0; i < 5; i++) {
            const int ct = L.u.b.cTop[first + i], cl = L.u.b.cLeft[first + i];
            const int former_bottom = top + gheight;
            top = imin(ct, top);
            gwidth = (cl + SCW) - gleft;
            gheight = imax(ct + SCH, former_bottom) - top;
          }
          short *h = sg->hdr[n_emitted];
          h[0] = (short)top, h[1] = (short)gleft, h[2] = (short)gwidth, h[3] = (short)gheight;
          for (int i = 0; i < 5; i++) {
            h[4 + i] = (short)L.u.b.cTop[first + i];
            h[9 + i] = (short)L.u.b.cLeft[first + i];
          }
          h[14] = (short)base;
          h[15] = 0;
        }
        n_emitted++;
      }
    }
  }
  XS_TL(8)
#ifdef DMZ_XSEG_TL
  if (lane == 0 && (blockIdx.x & 1023) == 7) {  // (a sample of the waves: every wave adding to nine counters slows the kernel down)
    for (int i = 0; i < 9; i++) atomicAdd(&g_xs_tl[i], (unsigned long long)tl_acc[i]);
    atomicAdd(&g_xs_tl[15], 1ull);
  }
#endif
  if (lane == 0) sg->n = n_emitted;
#ifdef DMZ_XSEG_DBG
  if (lane == 0) atomicAdd(&g_xs_dbg[5], (unsigned long long)(__builtin_readcyclecounter() - dbg_start)), atomicAdd(&g_xs_dbg[6], 1ull);
#endif
}

__global__ __launch_bounds__(64, DMZ_XSEG_WAVES) void k_expiry_seg(const float *__restrict__ wts, const float *__restrict__ xw,
                                                      const uint8_t *__restrict__ cards, size_t card_stride, int n,
                                                      const dmz_hip_frame_result *__restrict__ results,
                                                      const dmz_hip_expiry_result *__restrict__ er,
                                                      DmzExpiryStage *__restrict__ stage) {
  const int f = blockIdx.x / 3, st = blockIdx.x - f * 3, lane = threadIdx.x;
  if (f >= n) return;
  if (st >= er[f].n_stripes) return;
  __shared__ SegLds L;
  expiry_seg_stripe(L, wts, xw, cards + (size_t)f * card_stride, n, results[f].vseg_y_offset + kNumberHeight,
                    er[f].stripe_base_row[st], er[f].stripe_sum[st], stage + (size_t)f * 3 + st, lane);
}

// Round 6, measured, not the default: stripes + segmentation in ONE kernel, a wave per frame -- the row sums
// below the number, the stripe search, then the frame's (up to three) stripes one after the other.  The stripe search's loads
// (~92 scattered 258-byte row pieces per card, a pass of its own over HBM) then wait beside eleven other waves' list logic, and
// the 23 rows a stripe stages come back from the cache the search has just pulled them through.  expiry_seg stage 2.66 ->
// 2.59 ms, the three-queue step 18.88 -> 18.91 ms (profiles/r6_expiry_fused_stripes_ab.log): the search's 13 k cycles of waiting
// cost a 12-wave-per-CU kernel about what they cost a kernel of their own at 32 waves per CU.  (Its 168 registers also lie in
// the 161 .. 199 band in which k_homography lost quarter-waves beside two 160-register waves -- geometry.hip, DESIGN_LOG.md round 6.)
static_assert(sizeof(StripeLds) <= IROWS * ISTRIDE, "the stripe search's arrays lie over the horizontal-pass bytes");
__global__ __launch_bounds__(64, DMZ_XSEG_WAVES) void k_expiry_seg_fused(const float *__restrict__ wts, const float *__restrict__ xw,
                                                            const uint8_t *__restrict__ cards, size_t card_stride, int n,
                                                            const dmz_hip_frame_result *__restrict__ results,
                                                            dmz_hip_expiry_result *__restrict__ out,
                                                            DmzExpiryStage *__restrict__ stage) {
  const int f = blockIdx.x, lane = threadIdx.x;
  if (f >= n) return;
  __shared__ SegLds L;
  int ns;
  {
    int br[3];
    long long su[3];
    ns = expiry_stripes_body(f, lane, cards, card_stride, results, out, stage, *(StripeLds *)L.inter, br, su);
    if (lane < 3) {
      L.stripe_row[lane] = lane == 0 ? br[0] : (lane == 1 ? br[1] : br[2]);
      L.stripe_sum[lane] = (int)(lane == 0 ? su[0] : (lane == 1 ? su[1] : su[2]));  // (a sum is the upper 25 bits of a 32-bit key)
    }
  }
#pragma unroll 1
  for (int st = 0; st < ns; st++) {
    __syncthreads();  // (one wave: orders the LDS traffic of two stripes)
    const int base = __builtin_amdgcn_readfirstlane(L.stripe_row[st]);
    const long long sum = (long long)__builtin_amdgcn_readfirstlane(L.stripe_sum[st]);
    // (the lane index is made opaque per stripe: with it loop-invariant the compiler hoists every lane-derived constant of the
    // body -- column keys, lane masks, addresses -- out of the stripe loop and keeps them alive across it: 60 bytes of scratch)
    int lane_st = (int)threadIdx.x;
    asm volatile("" : "+v"(lane_st));
    expiry_seg_stripe(L, wts, xw, cards + (size_t)blockIdx.x * card_stride, n,
                      __builtin_amdgcn_readfirstlane(results[blockIdx.x].vseg_y_offset) + kNumberHeight, base, sum,
                      stage + (size_t)blockIdx.x * 3 + st, lane_st);
  }
}

// ---------------------------------------------------------------------------------------------
// Expiry digit CNN (applyc_bf4dd6c8) for up to four 16x11 inputs resident in LDS.
// ---------------------------------------------------------------------------------------------
constexpr int XC_THREADS = 256;
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int XIN_W = 20, XIN_H = 24;  // zero-padded input: 4 rows/cols of padding before, 4/5 after
constexpr int C2_KSTEPS = dmzx::C2_KSTEPS, C2_MP = dmzx::C2_MAPS_PAD;  // conv2, bf16 variants: see dmz_hip_internal.h
// The CNN runs its two convolutions for XND digits at a time (layer-1 output of two digits: 31 KB instead of 63 KB, which
// is what lets three workgroups share a CU); the dense layers see all four digits again.
#ifndef DMZ_XND
#define DMZ_XND 2
#endif
constexpr int XND = DMZ_XND;                          // digits per convolution pass
constexpr int XROWS = 18 * XND;                 // conv2 output rows of a pass (6 x 3 per digit)
constexpr int XMT = (XROWS + 15) / 16;          // 16-row tiles
constexpr int L1_BF16_ELEMS = XND * 70 * C2_MP;  // [digit of the pass][pooled position 10 x 7][map, padded to 56]
struct CatLds {
  // layer-1 output of one pass.  F32 variant: float [digit][map 50][70] (28,000 B).  bf16 variants: the bf16 rounding of the
  // activations and the bf16 rounding of the remainder, each [digit][position 70][map 56] (2 x 15,680 B), so that
  // the eight k of a matrix-core fragment (eight maps of one tap) are one aligned 16-byte read.  Prep-time
  // scratch and the conv2 partial sums overlay it.
  __attribute__((aligned(16))) unsigned char l1raw[2 * L1_BF16_ELEMS * 2];  // 31,360 B
  // zero-padded, mean-free inputs [digit][24][20]: floats (F32 variant) or three bf16 planes hi / mid / lo whose sum is
  // the float (the A operand of the matrix-core conv1)
  union {
    __attribute__((aligned(8))) float xin[4 * XIN_H * XIN_W];  // 7,680 B
    unsigned short xin3[3][4 * XIN_H * XIN_W];                 // 11,520 B
  };
  float es[4 * 16];
  float mean[4];
  // the raw inputs, until the mean is subtracted; then (F32 variant only) the conv1 weights, tap-major, rewritten for
  // every group; after layer 1 the outputs of layer 2 (4 x 120) and of the hidden layer (4 x 176) -- 48 KB in all: three
  // workgroups per CU
  union {
    __attribute__((aligned(16))) float xf[4 * 176];  // (16 bytes: the mean's reads are ds_read_b128)
    __attribute__((aligned(8))) float c1w[25 * 50];
  };
  short hdr[DMZ_HIP_EXPIRY_MAX_GROUPS][16];
  int n_groups;
  float cwt[256], swt[8];  // the bilateral filter's tables (DmzExpiryTables)
};
static_assert(4 * 120 + 4 * 176 <= 25 * 50, "l2 + l3 overlay the conv1 weights");
static_assert(sizeof(CatLds) <= 163840 / (XND == 1 ? 4 : 3), "three (four) workgroups per CU");
static_assert(4 * XROWS * 40 * 4 <= 2 * L1_BF16_ELEMS * 2, "the conv2 partial sums fit over the layer-1 output");
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float tree_sum10(const float *v) {  // Eigen scalar redux order (Redux.h:77-90)
  const float a = (v[0] + v[1]) + (v[2] + (v[3] + v[4]));
  const float b = (v[5] + v[6]) + (v[7] + (v[8] + v[9]));
  return a + b;
}

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
// the 16-bit matrix-core product of the variant: f16 operands (F16X3) or bf16 operands, fp32 accumulation
template <int MODE>
__device__ __forceinline__ f32x4 mma16(u32x4 a, u32x4 b, f32x4 c) {
  if constexpr (MODE == DMZ_HIP_EXPIRY_CONV_F16X3)
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// v = hi + lo to 2^-22 |v| (or 2^-25: f16 subnormals, which the matrix core keeps -- tools/ubench/mfma_f16_denorm.hip)
__device__ __forceinline__ void split_f16(float v, unsigned short &hi, unsigned short &lo) {
  const _Float16 h = (_Float16)v;
  const _Float16 l = (_Float16)(v - (float)h);
  hi = __builtin_bit_cast(unsigned short, h);
  lo = __builtin_bit_cast(unsigned short, l);
}

#ifndef DMZ_XCAT_STOP
#define DMZ_XCAT_STOP 99
#endif
// developer probe: -DDMZ_XC_TIMING prints the cycle counter at the phase boundaries of one categorised workgroup
#ifdef DMZ_XC_TIMING
__device__ long long g_xc_t[16];
__device__ int g_xc_block = -1;
#define XC_T(i) if (threadIdx.x == 0 && (int)blockIdx.x == g_xc_block) g_xc_t[i] = clock64();
#else
#define XC_T(i)
#endif
// xf[nd][176] raw inputs -> scores (global, nd x 10 floats at `out`, row stride 10).  MODE = DMZ_HIP_EXPIRY_CONV_*
template <int MODE>
__device__ __forceinline__ void expiry_cnn_block(const float *__restrict__ wts, const float *__restrict__ xw, CatLds &S, int nd,
                                 float *__restrict__ out, int tid_in) {
  int tid = tid_in;
  const float *xm = wts + dmzw::EXPIRY;
  float *const l1f = (float *)S.l1raw;                                     // F32 variant
  unsigned short *const l1h = (unsigned short *)S.l1raw;                   // bf16 variants: high parts ...
  unsigned short *const l1l = l1h + L1_BF16_ELEMS;                         // ... and remainders
  // layer-2 output: over the conv1 weights' buffer, which only the F32 variant still needs while the second pass runs
  // (there: behind the float input planes); hidden layer: over that buffer, after both passes
  float *const l2 = MODE == DMZ_HIP_EXPIRY_CONV_F32 ? S.xin + 4 * XIN_H * XIN_W : S.c1w;
  float *const l3 = S.c1w + 4 * 120;
  static_assert(sizeof(S.xin3) >= sizeof(S.xin) + 4 * 120 * sizeof(float), "F32 variant: l2 behind the float planes");
  XC_T(2)
  if (DMZ_XCAT_STOP == 1) return;
  // modelc_bf4dd6c8.cpp:13459: subtract the mean (sequential 176-term sum)
  if (tid < nd) {
    const f32x4 *x4 = (const f32x4 *)(S.xf + tid * 176);  // (the same sequential order, four terms per LDS read)
    float m = 0.0f;
#pragma unroll 4
    for (int i = 0; i < 44; i++) {
      const f32x4 v = x4[i];
      m = i == 0 ? v[0] : m + v[0];
      m = m + v[1];
      m = m + v[2];
      m = m + v[3];
    }
    S.mean[tid] = m / 176.0f;
  }
  __syncthreads();
  for (int i = tid; i < nd * 176; i += XC_THREADS) {
    const int d = i / 176, p = i - d * 176, r = p / 11, c = p - r * 11;
    const float v = S.xf[i] - S.mean[d];
    const int at = d * XIN_H * XIN_W + (r + 4) * XIN_W + (c + 4);
    if (MODE == DMZ_HIP_EXPIRY_CONV_F32) {
      S.xin[at] = v;
    } else if (MODE == DMZ_HIP_EXPIRY_CONV_F16X3) {
      split_f16(v, S.xin3[0][at], S.xin3[1][at]);
    } else {
      // v = hi + mid + lo exactly (three bf16 numbers: 24 bits of mantissa)
      const __bf16 hi = (__bf16)v;
      const float r1 = v - (float)hi;
      const __bf16 mid = (__bf16)r1;
      const __bf16 lo = (__bf16)(r1 - (float)mid);
      S.xin3[0][at] = __builtin_bit_cast(unsigned short, hi);
      S.xin3[1][at] = __builtin_bit_cast(unsigned short, mid);
      S.xin3[2][at] = __builtin_bit_cast(unsigned short, lo);
    }
  }
  __syncthreads();
  if constexpr (MODE == DMZ_HIP_EXPIRY_CONV_F32) {
    for (int i = tid; i < 1250; i += XC_THREADS) {  // (the raw inputs that shared this buffer are consumed)
      const int k = i / 25, t = i - k * 25;
      S.c1w[t * 50 + k] = xm[dmzw::X_C1W + i];
    }
    __syncthreads();
  }
  XC_T(3)
#pragma unroll 1
  for (int d0 = 0; d0 < nd; d0 += XND) {  // ---- the two convolutions, XND digits per pass ----
  const int ndp = imin(XND, nd - d0);
  asm volatile("" : "+v"(tid));  // (per pass: keeps the passes' operand fragments and index tables out of the callers' loops)
  if constexpr (MODE != DMZ_HIP_EXPIRY_CONV_F32) {
    // layer 1 on the matrix cores: the "full" 5x5 correlation (20 x 14), pool 2x2 -> 10 x 7, + bias, ReLU as
    // out[p][n] = sum_k patch[p][k] W[k][n], p = pre-pool position, k = tap (25 -> 32), n = map (50 -> 64), on
    // v_mfma_f32_16x16x32_bf16 with both operands split in three bf16 parts and the six products that carry 2^-24 kept
    // (hi*hi, hi*mid, mid*hi, hi*lo, lo*hi, mid*mid): the fp32 convolution to rounding, at half the time of the
    // packed-FMA form (below, F32 variant).  A tile's sixteen rows are four pool windows x their four positions, so the
    // four accumulator elements of a lane ARE a pool window: max, bias, ReLU, hi/lo split for conv2, one store pair.
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), m16 = lane & 15, kk = lane >> 4;
    // F16X3: both operands in two f16 parts (22 bits) and the three products that carry 2^-22 (lo*hi, hi*lo, hi*hi)
    constexpr int NP = MODE == DMZ_HIP_EXPIRY_CONV_F16X3 ? 2 : 3;
    u32x4 wb[NP][4];
    {
      const u32x4 *bsrc = (const u32x4 *)(xw + (MODE == DMZ_HIP_EXPIRY_CONV_F16X3 ? dmzx::CONV1_F2 : dmzx::CONV1_B3)) + lane;
#pragma unroll
      for (int part = 0; part < NP; part++)
#pragma unroll
        for (int nt = 0; nt < 4; nt++) wb[part][nt] = bsrc[(part * 4 + nt) * 64];
    }
    float bias[4];
#pragma unroll
    for (int nt = 0; nt < 4; nt++) bias[nt] = 16 * nt + m16 < 50 ? xm[dmzw::X_C1B + 16 * nt + m16] : 0.0f;
    int toff[8];  // element offset of tap k = 8 kk + e inside the padded input (taps >= 25 meet zero weights)
#pragma unroll
    for (int e = 0; e < 8; e++) {
      const int k = 8 * kk + e, ti = (k * 205) >> 10;  // k / 5 for k < 32
      toff[e] = k < 25 ? ti * XIN_W + (k - 5 * ti) : 0;
    }
    typedef const volatile __attribute__((address_space(3))) unsigned short *lds_vu16;  // (keeps the reads 16-bit)
    const int nwin = ndp * 70;  // pool windows of this pass: [digit of the pass][10 x 7]
    for (int t = wave; 4 * t < nwin; t += XC_THREADS / 64) {
      int W = 4 * t + (m16 >> 2);
      W = W < nwin ? W : 0;
      const int d = (W * 937) >> 16, pos = W - 70 * d;     // W / 70 for W < 280
      const int pr = (pos * 37) >> 8, pc = pos - 7 * pr;   // pos / 7 for pos < 70
      const int base = (d0 + d) * XIN_H * XIN_W + (2 * pr + ((m16 >> 1) & 1)) * XIN_W + 2 * pc + (m16 & 1);
      uint32_t a[NP][4];
#pragma unroll
      for (int part = 0; part < NP; part++) {
        const lds_vu16 pl = (lds_vu16)S.xin3[part] + base;
#pragma unroll
        for (int e2 = 0; e2 < 4; e2++) a[part][e2] = (uint32_t)pl[toff[2 * e2]] | ((uint32_t)pl[toff[2 * e2 + 1]] << 16);
      }
      const u32x4 ah = {a[0][0], a[0][1], a[0][2], a[0][3]};
      const u32x4 am = {a[1][0], a[1][1], a[1][2], a[1][3]};                    // (F16X3: the low part)
      const u32x4 al = {a[NP - 1][0], a[NP - 1][1], a[NP - 1][2], a[NP - 1][3]};
      // (no fence here, unlike the conv2 loop: the B fragments are resident and the A fragments come from LDS only; run to
      // run identical on 4 x 16 384 frames, tools/dev/det_variant.sh, and 4.6 % faster than the fenced form)
#ifdef DMZ_C1_FENCE  /* developer probe */
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
#endif
      f32x4 acc[4];
      // small terms first; the four map tiles are independent chains, interleaved
#pragma unroll
      for (int nt = 0; nt < 4; nt++) acc[nt] = mma16<MODE>(al, wb[0][nt], (f32x4){0.f, 0.f, 0.f, 0.f});
#pragma unroll
      for (int nt = 0; nt < 4; nt++) acc[nt] = mma16<MODE>(ah, wb[NP - 1][nt], acc[nt]);
      if constexpr (NP == 3) {
#pragma unroll
        for (int nt = 0; nt < 4; nt++) acc[nt] = mma16<MODE>(am, wb[1][nt], acc[nt]);
#pragma unroll
        for (int nt = 0; nt < 4; nt++) acc[nt] = mma16<MODE>(am, wb[0][nt], acc[nt]);
#pragma unroll
        for (int nt = 0; nt < 4; nt++) acc[nt] = mma16<MODE>(ah, wb[1][nt], acc[nt]);
      }
#pragma unroll
      for (int nt = 0; nt < 4; nt++) acc[nt] = mma16<MODE>(ah, wb[0][nt], acc[nt]);
#ifdef DMZ_C1_FENCE
      __builtin_amdgcn_sched_barrier(0);
#endif
      // D: column (map) = 16 nt + (lane & 15), rows 4 kk .. 4 kk + 3 = the four positions of pool window 4 t + kk
      const int Wd = 4 * t + kk;
      if (Wd < nwin) {
#pragma unroll
        for (int nt = 0; nt < 4; nt++) {
          const int n = 16 * nt + m16;
          if (n < C2_MP) {  // maps 50 .. 55 multiply zero weights in conv2 but must be finite: written as 0
            const float mx = fmaxf(fmaxf(acc[nt][0], acc[nt][1]), fmaxf(acc[nt][2], acc[nt][3]));
            const float v = n < 50 ? fmaxf(mx + bias[nt], 0.0f) : 0.0f;
            if constexpr (MODE == DMZ_HIP_EXPIRY_CONV_F16X3) {
              split_f16(v, l1h[Wd * C2_MP + n], l1l[Wd * C2_MP + n]);
            } else {
              const __bf16 hi = (__bf16)v;
              const __bf16 lo = (__bf16)(v - (float)hi);
              l1h[Wd * C2_MP + n] = __builtin_bit_cast(unsigned short, hi);
              l1l[Wd * C2_MP + n] = __builtin_bit_cast(unsigned short, lo);
            }
          }
        }
      }
    }
  } else {
  // layer 1, F32 variant: "full" 5x5 correlation (20 x 14 of it), pool 2x2 -> 10 x 7, + bias, ReLU.
  // Work item = (map pair, digit, pooled row): v_pk_fma_f32 carries two maps per instruction;
  // the 6 x 18 input strip of the pooled row sits in registers for its seven outputs.
  for (int idx = tid; idx < 25 * ndp * 10; idx += XC_THREADS) {
    const int pr = idx / (25 * ndp), rem = idx - pr * (25 * ndp);
    const int d = rem / 25, kp = rem - d * 25;
    f32x2 w[25];
#pragma unroll
    for (int i = 0; i < 25; i++) {
      const float2 t = *(const float2 *)(S.c1w + i * 50 + 2 * kp);
      w[i] = (f32x2){t.x, t.y};
    }
    const f32x2 bias = {xm[dmzw::X_C1B + 2 * kp], xm[dmzw::X_C1B + 2 * kp + 1]};
    const float *xi = S.xin + (d0 + d) * XIN_H * XIN_W + (2 * pr) * XIN_W;
    float *o0 = l1f + (d * 50 + 2 * kp) * 70 + pr * 7, *o1 = o0 + 70;
#pragma unroll 1
    for (int pc = 0; pc < 7; pc++) {
      float patch[6][6];
#pragma unroll
      for (int a = 0; a < 6; a++)
#pragma unroll
        for (int b = 0; b < 3; b++) {
          const float2 t = *(const float2 *)(xi + a * XIN_W + 2 * pc + 2 * b);
          patch[a][2 * b] = t.x;
          patch[a][2 * b + 1] = t.y;
        }
      f32x2 m = {0.0f, 0.0f};
#pragma unroll
      for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++) {
          f32x2 acc = {0.0f, 0.0f};
#pragma unroll
          for (int i = 0; i < 5; i++)
#pragma unroll
            for (int j = 0; j < 5; j++) {
              const float x = patch[a + i][b + j];
              acc = __builtin_elementwise_fma(w[i * 5 + j], (f32x2){x, x}, acc);
            }
          m = (a == 0 && b == 0) ? acc : __builtin_elementwise_max(m, acc);
        }
      f32x2 v = m + bias;
      v.x = v.x > 0.0f ? v.x : 0.0f;
      v.y = v.y > 0.0f ? v.y : 0.0f;
      o0[pc] = v.x;
      o1[pc] = v.y;
    }
  }
  }  // F32 variant
  __syncthreads();
  if (d0 == 0) { XC_T(4) } else { XC_T(6) }
  if (DMZ_XCAT_STOP == 2) return;
  // layer 2: valid 5x5 correlation summed over the 50 maps -> 6 x 3, pool 2x3 -> 3, + bias, ReLU,
  // as the GEMM  out[p][n] = sum_k patch[p][k] W[k][n]  (p = digit x 18 positions = 36 rows per pass,
  // n = 40 maps, k = map x 5 x 5 = 1250) on v_mfma_f32_16x16x4_f32: 3 x 3 tiles of 16 x 16, the
  // k range split over the four waves (A gathered from l1, B from the
  // tap-major zero-padded weight copy, prefetched one k-step ahead); the four partial sums meet in
  // LDS (over l1, dead by then) where the 2 x 3 max-pool, bias and ReLU finish the layer.
  {
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), m16 = lane & 15, kk = lane >> 4;
    f32x4 acc[XMT][3];
#pragma unroll
    for (int mt = 0; mt < XMT; mt++)
#pragma unroll
      for (int nt = 0; nt < 3; nt++) acc[mt][nt] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
    if constexpr (MODE == DMZ_HIP_EXPIRY_CONV_F32) {
    int baseA[XMT];
#pragma unroll
    for (int mt = 0; mt < XMT; mt++) {
      const int pp = 16 * mt + m16, pc = pp < XROWS ? pp : 0;
      const int d = pc / 18, pos = pc - 18 * d, r = pos / 3, c = pos - 3 * r;
      baseA[mt] = d * 3500 + r * 7 + c;
    }
    const float *c2p = xw + dmzx::CONV2_P + m16;
    const int ks0 = wave * 79, ks1 = imin(ks0 + 79, 313);
    // k -> offset of tap (map, i, j) inside l1[d]; the two padded k rows multiply zero weights
    auto tap_off = [](int k) {
      const int mp = (k * 1311) >> 15, t = k - 25 * mp, i = (t * 13) >> 6;  // k / 25, k % 25, t / 5
      return k < 1250 ? mp * 70 + i * 7 + (t - 5 * i) : 0;
    };
    // B (global, L2 latency) is fetched one block of four k-steps ahead, A (LDS) one k-step ahead
    float bnx[4][3], an[XMT];
#pragma unroll
    for (int u = 0; u < 4; u++)
#pragma unroll
      for (int nt = 0; nt < 3; nt++) bnx[u][nt] = c2p[(4 * imin(ks0 + u, ks1 - 1) + kk) * 48 + 16 * nt];
    {
      const int off = tap_off(4 * ks0 + kk);
#pragma unroll
      for (int mt = 0; mt < XMT; mt++) an[mt] = l1f[baseA[mt] + off];
    }
    for (int kb = ks0; kb < ks1; kb += 4) {
      float bc[4][3];
#pragma unroll
      for (int u = 0; u < 4; u++)
#pragma unroll
        for (int nt = 0; nt < 3; nt++) {
          bc[u][nt] = bnx[u][nt];
          bnx[u][nt] = c2p[(4 * imin(kb + 4 + u, ks1 - 1) + kk) * 48 + 16 * nt];
        }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int ks = kb + u;
        if (ks < ks1) {  // uniform per wave
          float av[XMT];
#pragma unroll
          for (int mt = 0; mt < XMT; mt++) av[mt] = an[mt];
          const int off = tap_off(4 * imin(ks + 1, ks1 - 1) + kk);
#pragma unroll
          for (int mt = 0; mt < XMT; mt++) an[mt] = l1f[baseA[mt] + off];
#pragma unroll
          for (int mt = 0; mt < XMT; mt++)
#pragma unroll
            for (int nt = 0; nt < 3; nt++)
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mt], bc[u][nt], acc[mt][nt], 0, 0, 0);
        }
      }
    }
    } else {
      // bf16 variants on v_mfma_f32_16x16x32_bf16.  K is ordered tap-major, eight maps per run, four runs per
      // k-step: lane (row m16, run kk) of k-step ks holds run R = 4 ks + kk = tap R / 7, maps 8 (R % 7) .. + 7 --
      // one aligned 16-byte LDS read per operand part; B comes fragment-ordered from global memory (L2).
      // BF16X3: a.b ~ al.bh + ah.bl + ah.bh (small terms first), fp32 accumulation.
      int baseA[XMT];  // byte offset of (digit, row, column) of the 6 x 3 output grid, tap (0, 0), map 0
#pragma unroll
      for (int mt = 0; mt < XMT; mt++) {
        const int pp = 16 * mt + m16, pc = pp < XROWS ? pp : 0;
        const int d = pc / 18, pos = pc - 18 * d, r = pos / 3, c = pos - 3 * r;
        baseA[mt] = (d * 70 + r * 7 + c) * C2_MP * 2;
      }
      const unsigned char *ah_b = (const unsigned char *)l1h;
      constexpr bool kSplit = MODE == DMZ_HIP_EXPIRY_CONV_BF16X3 || MODE == DMZ_HIP_EXPIRY_CONV_F16X3;  // three products
      const u32x4 *bhp = (const u32x4 *)(xw + (MODE == DMZ_HIP_EXPIRY_CONV_F16X3 ? dmzx::CONV2_FH : dmzx::CONV2_BH)) + lane;
      const u32x4 *blp = (const u32x4 *)(xw + (MODE == DMZ_HIP_EXPIRY_CONV_F16X3 ? dmzx::CONV2_FL : dmzx::CONV2_BL)) + lane;
      constexpr int kPerWave = C2_KSTEPS / 4;
      static_assert(kPerWave * 4 == C2_KSTEPS, "k-steps split evenly over the four waves");
      const int ks0 = wave * kPerWave;
      auto load_b = [&](int ks, u32x4 (&h)[3], u32x4 (&l)[3]) {
#pragma unroll
        for (int nt = 0; nt < 3; nt++) {
          h[nt] = bhp[(ks * 3 + nt) * 64];
          if (kSplit) l[nt] = blp[(ks * 3 + nt) * 64];
        }
      };
      // One k-step: all operand fragments loaded and WAITED FOR, then the 45 (15) matrix instructions, with
      // scheduling barriers so that the compiler neither starts them under outstanding loads nor hoists the
      // next k-step's loads into them.  Overlapped schedules (the compiler's own, or register sets rotating
      // under software prefetch) were faster by a few per cent on an otherwise idle CU and WRONG in the
      // younger of two workgroups sharing a CU: accumulators came out different from run to run while
      // operand fragments, LDS contents and the matrix instruction itself check out one by one
      // (tools/dev/expiry_model_dup.py, tools/ubench/mfma_*_coresident.hip).  The strict form is
      // deterministic in every configuration tried; the other workgroup of the CU hides its bubbles.
#ifndef DMZ_C2_SCHED  /* developer probe: 0 strict, 1 no fences */
#define DMZ_C2_SCHED 0
#endif
#ifndef DMZ_C2_KB  /* k-steps whose fragments are requested together, ahead of one wait */
#define DMZ_C2_KB 1
#endif
      // KB k-steps per round: their B fragments (L2) and A fragments (LDS) are requested together, waited for once, then
      // the 27 (9) matrix instructions of each k-step issue (the load -> wait -> matrix order of the strict form).  The
      // timeline (-DDMZ_XC_TIMING) shows a k-step as ~2 k cycles of L2 latency + 460 cycles of matrix instructions, but
      // batching does not pay: KB = 2 (133 registers) 2.12 vs 2.09 ms per 65 536 frames, KB = 3 2.35 -- the other two waves of
      // the SIMD already fill the matrix pipe while one waits.
      constexpr int KB = DMZ_C2_KB;
#pragma unroll 1
      for (int ks = ks0; ks < ks0 + kPerWave; ks += KB) {
        u32x4 ah[KB][XMT], al[KB][XMT], bh[KB][3], bl[KB][3];
#pragma unroll
        for (int u = 0; u < KB; u++) {
          const int kq = imin(ks + u, ks0 + kPerWave - 1);  // (a partial last round re-reads its last k-step: not used)
          load_b(kq, bh[u], bl[u]);
          const int R = 4 * kq + kk;
          const int q7 = (R * 9363) >> 16;  // R / 7 for R < 176
          const int t = imin(q7, 24);       // run 175 is padding (zero weights)
          const int i5 = (t * 13) >> 6;     // t / 5
          const int offA = ((i5 * 7 + (t - 5 * i5)) * C2_MP + 8 * (R - 7 * q7)) * 2;
#pragma unroll
          for (int mt = 0; mt < XMT; mt++) {
            ah[u][mt] = *(const u32x4 *)(ah_b + baseA[mt] + offA);
            if (kSplit) al[u][mt] = *(const u32x4 *)(ah_b + L1_BF16_ELEMS * 2 + baseA[mt] + offA);
          }
        }
        if (DMZ_C2_SCHED != 1) {
          asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int u = 0; u < KB; u++) {
          if (ks + u < ks0 + kPerWave) {  // uniform
#pragma unroll
            for (int mt = 0; mt < XMT; mt++)
#pragma unroll
              for (int nt = 0; nt < 3; nt++) {
                if (kSplit) {  // small terms first
                  acc[mt][nt] = mma16<MODE>(al[u][mt], bh[u][nt], acc[mt][nt]);
                  acc[mt][nt] = mma16<MODE>(ah[u][mt], bl[u][nt], acc[mt][nt]);
                }
                acc[mt][nt] = mma16<MODE>(ah[u][mt], bh[u][nt], acc[mt][nt]);
              }
          }
        }
        if (DMZ_C2_SCHED != 1) __builtin_amdgcn_sched_barrier(0);
      }
    }
    __syncthreads();  // every wave is done with l1
    float *part = l1f;  // [4 waves][XROWS][40]
#pragma unroll
    for (int mt = 0; mt < XMT; mt++)
#pragma unroll
      for (int nt = 0; nt < 3; nt++)
#pragma unroll
        for (int v = 0; v < 4; v++) {
          const int pp = 16 * mt + 4 * kk + v, nn = 16 * nt + m16;
          if (pp < XROWS && nn < 40) part[(wave * XROWS + pp) * 40 + nn] = acc[mt][nt][v];
        }
    __syncthreads();
    for (int idx = tid; idx < ndp * 120; idx += XC_THREADS) {
      const int d = idx / 120, rem = idx - d * 120, nn = rem / 3, pr = rem - nn * 3;
      float m = 0.0f;
#pragma unroll
      for (int q = 0; q < 6; q++) {
        const int pp = d * 18 + 6 * pr + q;
        const float t = (part[(0 * XROWS + pp) * 40 + nn] + part[(1 * XROWS + pp) * 40 + nn]) +
                        (part[(2 * XROWS + pp) * 40 + nn] + part[(3 * XROWS + pp) * 40 + nn]);
        m = q == 0 ? t : fmaxf(m, t);
      }
      const float v = m + xm[dmzw::X_C2B + nn];
      l2[d0 * 120 + idx] = v > 0.0f ? v : 0.0f;
    }
  }
  __syncthreads();  // (the partial sums over l1 are consumed before the next pass writes l1)
  if (d0 == 0) { XC_T(5) } else { XC_T(7) }
  }  // pass
  if (DMZ_XCAT_STOP == 3) return;
  // FC 120 -> 176, ReLU and FC 176 -> 10, softmax -- on the matrix core, not for its throughput
  // (four rows of sixteen are real) but because every lane's weight loads are then independent:
  // the VALU version walked each output's 120 (176) weights as one dependent fmaf chain fed from L2.
  // The accumulation order is unchanged: v_mfma_f32_16x16x4_f32 adds its four k in order.
  // Weights in fragment order (dmzx::FC1_F / FC2_F: four k-steps of a lane per 16-byte load), ALL of a wave's tiles and
  // the output layer's weights requested up front: one L2 round trip for the two layers instead of one per tile and two
  // for the output layer (the timeline showed 21 - 69 k cycles here under load, of ~170 k per group).
  f32x4 b2v[11];
  {
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), m16 = lane & 15, kk = lane >> 4;
    const f32x4 *fc1f = (const f32x4 *)(xw + dmzx::FC1_F) + lane;
    if (wave == 0) {
      const f32x4 *fc2f = (const f32x4 *)(xw + dmzx::FC2_F) + lane;
#pragma unroll
      for (int g = 0; g < 11; g++) b2v[g] = fc2f[g * 64];
    }
    // (two tiles' fragments in flight; the third tile's take the first tile's registers once that tile is done)
    f32x4 bv[2][8];
    auto load_tile = [&](int r, f32x4 (&dst)[8]) {
      const int nt = imin(wave + 4 * r, 10);
#pragma unroll
      for (int g = 0; g < 8; g++) dst[g] = fc1f[(nt * 8 + g) * 64];
    };
    load_tile(0, bv[0]);
    load_tile(1, bv[1]);
    float a1[30];
#pragma unroll
    for (int ks = 0; ks < 30; ks++) a1[ks] = m16 < nd ? l2[m16 * 120 + 4 * ks + kk] : 0.0f;
#pragma unroll
    for (int r = 0; r < 3; r++) {
      const int nt = wave + 4 * r;
      if (nt < 11) {  // uniform per wave
        f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int ks = 0; ks < 30; ks++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[ks], bv[r & 1][ks >> 2][ks & 3], acc, 0, 0, 0);
        if (r == 0) load_tile(2, bv[0]);
        // D: row (digit) = 4 * kk + v, column (unit) = m16: the digits sit in the lanes with kk == 0
        if (kk == 0) {
          const int j = 16 * nt + m16;
          const float hb = xm[dmzw::X_HB + j];
#pragma unroll
          for (int v = 0; v < 4; v++)
            if (v < nd) {
              const float t = acc[v] + hb;
              l3[v * 176 + j] = t > 0.0f ? t : 0.0f;
            }
        }
      }
    }
  }
  __syncthreads();
  if (tid < 64) {
    const int m16 = tid & 15, kk = tid >> 4;
    f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
    float a2[44];
#pragma unroll
    for (int ks = 0; ks < 44; ks++) a2[ks] = m16 < nd ? l3[m16 * 176 + 4 * ks + kk] : 0.0f;
#pragma unroll
    for (int ks = 0; ks < 44; ks++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[ks], b2v[ks >> 2][ks & 3], acc, 0, 0, 0);
    if (kk == 0 && m16 < 10) {
      const float lb = xm[dmzw::X_LB + m16];
#pragma unroll
      for (int v = 0; v < 4; v++)
        if (v < nd) S.es[v * 16 + m16] = expf(acc[v] + lb);
    }
  }
  __syncthreads();
  if (tid < nd * 10) {
    const int d = tid / 10, k = tid - d * 10;
    out[d * 10 + k] = S.es[d * 16 + k] / tree_sum10(S.es + d * 16);
  }
  __syncthreads();
  XC_T(8)
}

template <int MODE>
__global__ __launch_bounds__(XC_THREADS, XND == 1 ? 4 : 3) void k_expiry_cat(const float *__restrict__ wts, const float *__restrict__ xw,
                                                           const DmzExpiryTables *__restrict__ tab,
                                                           const uint8_t *__restrict__ cards, size_t card_stride,
                                                           int n, const dmz_hip_frame_result *__restrict__ results,
                                                           const DmzExpiryStage *__restrict__ stage,
                                                           dmz_hip_expiry_result *__restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  CatLds &S = *(CatLds *)smem_raw;
  const int f = blockIdx.x, tid = threadIdx.x;
  if (f >= n) return;
  dmz_hip_expiry_result *er = out + f;
  const int n_stripes = er->n_stripes;
  const int flags = results[f].flags, yoff = results[f].vseg_y_offset;
  const bool gate = (flags & DMZ_HIP_FLAG_VSEG_OK) && yoff < CH - 2 * SCH && yoff >= 0;  // frame.cpp:72
  if (n_stripes == 0) {
    if (tid == 0 && gate && (flags & DMZ_HIP_FLAG_USABLE)) er->categorised = 1;
    return;
  }
  // ---- merge the per-stripe staging in stripe order (FrameScanResult.expiry_groups order) ----
  {
    int cnt[3], tot = 0;
    for (int s = 0; s < 3; s++) {
      cnt[s] = s < n_stripes ? stage[(size_t)f * 3 + s].n : 0;
      tot += cnt[s];
    }
    if (tid < 24) {
      const int s = tid >> 3, i = tid & 7;
      int pos = i;
      for (int q = 0; q < s; q++) pos += imin(cnt[q], DMZ_HIP_EXPIRY_MAX_GROUPS);
      if (i < imin(cnt[s], DMZ_HIP_EXPIRY_MAX_GROUPS) && pos < DMZ_HIP_EXPIRY_MAX_GROUPS) {
        const short *h = stage[(size_t)f * 3 + s].hdr[i];
        short *dst = (short *)&er->groups[pos];
        for (int q = 0; q < 16; q++) {
          dst[q] = h[q];
          S.hdr[pos][q] = h[q];
        }
      }
    }
    if (tid == 0) {
      er->n_found = tot;
      er->n_groups = imin(tot, DMZ_HIP_EXPIRY_MAX_GROUPS);
      S.n_groups = imin(tot, DMZ_HIP_EXPIRY_MAX_GROUPS);
    }
  }
  if (!(flags & DMZ_HIP_FLAG_USABLE)) return;  // scan.cpp:57-59
  if (DMZ_XCAT_STOP == 0) return;  // (developer ablation: the cost of the workgroups that have nothing to categorise)
#ifdef DMZ_XC_TIMING
  if (tid == 0 && f > n / 2 && S.n_groups > 0) atomicCAS(&g_xc_block, -1, f);
  __syncthreads();
#endif
  XC_T(0)
  if (tid == 0) er->categorised = 1;
  for (int i = tid; i < (int)(sizeof(S.xin3) / 4); i += XC_THREADS) ((uint32_t *)S.xin3)[i] = 0u;  // the zero padding
  S.cwt[tid] = tab->color_weight[tid];  // (XC_THREADS = 256 entries)
  if (tid < 8) S.swt[tid] = tab->space_weight[tid];
  __syncthreads();
  const int n_groups = S.n_groups;
  const uint8_t *card = cards + (size_t)f * card_stride;
  // prep scratch over l1 (dead until the first convolution)
  unsigned char *gp = S.l1raw;                             // 4 x 176 gradient / equalised patch
  unsigned char *sm = gp + 4 * 176;                        // 4 x 176 smoothed
  unsigned int *hist = (unsigned int *)(gp + 2 * 4 * 176); // 4 x 256
  unsigned char *roil = gp + 2 * 4 * 176 + 4 * 256 * 4;    // 4 x [16 rows][16 bytes]: the characters' pixels
  for (int g = 0; g < n_groups; g++) {
    // (the thread index is made opaque per group: hoisted out of this loop, the index arithmetic of all the phases below
    // is some eighty registers of loop invariants -- more than three waves per SIMD leave)
    int tid = (int)threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // ---- prepare_image_for_cat (expiry_categorize.cpp:35-70): wave d = character d, on its own until the CNN (a wave's
    // LDS operations execute in order: no workgroup barrier between the steps).  The 16 x 11 pixels arrive as one aligned
    // dword per lane (the taps are clamped to the character, so nothing else is read) instead of fifteen byte loads. ----
    const int d = wave, ci = d < 2 ? d : d + 1;
    const int left = S.hdr[g][9 + ci], top = S.hdr[g][4 + ci];
    unsigned char *rl = roil + d * 256;
    {
      const int row = lane >> 2, dw = (left >> 2) + (lane & 3);
      if (dw <= (left + TW - 1) >> 2) ((uint32_t *)rl)[lane] = *(const uint32_t *)(card + (size_t)(top + row) * CW + 4 * dw);
    }
    const unsigned char *roi = rl + (left & 3);
    for (int i = lane; i < 256; i += 64) hist[d * 256 + i] = 0u;
    __builtin_amdgcn_wave_barrier();
    for (int p = lane; p < 176; p += 64) {
      const int r = p / TW, c = p - r * TW;
      const int ru = r > 0 ? r - 1 : r, rd = r < TH - 1 ? r + 1 : r;
      const int cl = c > 0 ? c - 1 : c, cr = c < TW - 1 ? c + 1 : c;
      const int nn = roi[ru * 16 + c], ww = roi[r * 16 + cl], cc = roi[r * 16 + c], ee = roi[r * 16 + cr],
                ss = roi[rd * 16 + c];
      const int gv = imax(nn, imax(ww, imax(cc, imax(ee, ss)))) - imin(nn, imin(ww, imin(cc, imin(ee, ss))));
      gp[d * 176 + p] = (unsigned char)gv;
      atomicAdd(&hist[d * 256 + gv], 1u);
    }
    __builtin_amdgcn_wave_barrier();
    {  // llcv_equalize_hist LUT (stats.cpp:135-151): 4 bins per lane
      unsigned int *h = hist + d * 256;
      const int h0 = (int)h[lane * 4 + 0], h1 = (int)h[lane * 4 + 1], h2 = (int)h[lane * 4 + 2], h3 = (int)h[lane * 4 + 3];
      const int tot = h0 + h1 + h2 + h3;
      const int incl = dmzwave::inclusive_scan_i32(tot);
      const int excl = incl - tot;
      const float scale = 255.f / (TW * TH);
      const int c0 = excl + h0, c1 = c0 + h1, c2 = c1 + h2, c3 = c2 + h3;
      int l0 = __float2int_rn((float)c0 * scale), l1 = __float2int_rn((float)c1 * scale),
          l2 = __float2int_rn((float)c2 * scale), l3 = __float2int_rn((float)c3 * scale);
      l0 = imin(255, imax(0, l0)), l1 = imin(255, imax(0, l1)), l2 = imin(255, imax(0, l2)), l3 = imin(255, imax(0, l3));
      if (lane == 0) l0 = 0;
      __builtin_amdgcn_wave_barrier();
      h[lane * 4 + 0] = (unsigned)l0, h[lane * 4 + 1] = (unsigned)l1, h[lane * 4 + 2] = (unsigned)l2, h[lane * 4 + 3] = (unsigned)l3;
    }
    __builtin_amdgcn_wave_barrier();
    for (int p = lane; p < 176; p += 64) gp[d * 176 + p] = (unsigned char)hist[d * 256 + gp[d * 176 + p]];
    __builtin_amdgcn_wave_barrier();
    // cv::bilateralFilter(d = 3, BORDER_REPLICATE), generic accumulation order
    for (int p = lane; p < 176; p += 64) {
      const int r = p / TW, c = p - r * TW;
      const unsigned char *e = gp + d * 176;
      const int val0 = e[p];
      const int rr[5] = {imax(r - 1, 0), r, r, r, imin(r + 1, TH - 1)};
      const int cq[5] = {c, imax(c - 1, 0), c, imin(c + 1, TW - 1), c};
      float sum = 0.0f, wsum = 0.0f;
#pragma unroll
      for (int k = 0; k < 5; k++) {
        const int val = e[rr[k] * TW + cq[k]];
        const float w = S.swt[k] * S.cwt[iabs(val - val0)];
        sum += (float)val * w;
        wsum += w;
      }
      sm[d * 176 + p] = (unsigned char)__float2int_rn(sum / wsum);
    }
    __builtin_amdgcn_wave_barrier();
    for (int p = lane; p < 176; p += 64) S.xf[d * 176 + p] = (float)sm[d * 176 + p] * (1.0f / 255.0f);
    __syncthreads();
    XC_T(1)
    expiry_cnn_block<MODE>(wts, xw, S, 4, &er->groups[g].scores[0][0], tid);
#ifdef DMZ_XC_TIMING
    if (threadIdx.x == 0 && f == g_xc_block && g == 0)
      printf("expiry_cat: prep %lld mean/split %lld conv1a %lld conv2a %lld conv1b %lld conv2b %lld fc %lld\n", g_xc_t[1] - g_xc_t[0],
             g_xc_t[3] - g_xc_t[2], g_xc_t[4] - g_xc_t[3], g_xc_t[5] - g_xc_t[4], g_xc_t[6] - g_xc_t[5], g_xc_t[7] - g_xc_t[6], g_xc_t[8] - g_xc_t[7]);
#endif
  }
}

// applym_730c4cbd on n inputs (known-answer tests): one wave per input
__global__ __launch_bounds__(64) void k_slash_model(const float *__restrict__ wts, const float *__restrict__ xw,
                                                    const float *__restrict__ x, int n, float *__restrict__ out) {
  const int i = blockIdx.x, lane = threadIdx.x;
  if (i >= n) return;
  __shared__ float xs[176], hid[80];
  for (int k = lane; k < 176; k += 64) xs[k] = x[(size_t)i * 176 + k];
  __syncthreads();
  const float *w1t = xw + dmzx::SLASH_W1T;
  const float *sw = wts + dmzw::SLASH;
  const int j1 = imin(64 + lane, 79);
  float s0 = 0.0f, s1 = 0.0f;
  for (int k = 0; k < 176; k++) {
    s0 += w1t[k * 80 + lane] * xs[k];
    s1 += w1t[k * 80 + j1] * xs[k];
  }
  hid[lane] = tanhf(s0 + sw[dmzw::S_B1 + lane]);
  if (lane < 16) hid[64 + lane] = tanhf(s1 + sw[dmzw::S_B1 + 64 + lane]);
  __syncthreads();
  float e = 0.0f;
  if (lane < 2) {
    float s = 0.0f;
    for (int j = 0; j < 80; j++) s += sw[dmzw::S_W2 + lane * 80 + j] * hid[j];
    e = expf(s + sw[dmzw::S_B2 + lane]);
  }
  const float e0 = __shfl(e, 0, 64), e1 = __shfl(e, 1, 64);
  if (lane < 2) out[(size_t)i * 2 + lane] = e / (e0 + e1);
}

// applyc_bf4dd6c8 on n inputs, four per workgroup
template <int MODE>
__global__ __launch_bounds__(XC_THREADS, 3) void k_expiry_model(const float *__restrict__ wts, const float *__restrict__ xw,
                                                             const float *__restrict__ x, int n, float *__restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  CatLds &S = *(CatLds *)smem_raw;
  const int tid = threadIdx.x, first = blockIdx.x * 4;
  const int nd = imin(4, n - first);
  if (nd <= 0) return;
  for (int i = tid; i < (int)(sizeof(S.xin3) / 4); i += XC_THREADS) ((uint32_t *)S.xin3)[i] = 0u;  // the zero padding
  for (int i = tid; i < nd * 176; i += XC_THREADS) S.xf[i] = x[(size_t)first * 176 + i];
  __syncthreads();
  expiry_cnn_block<MODE>(wts, xw, S, nd, out + (size_t)first * 10, tid);
}

// ---------------------------------------------------------------------------------------------
// k_sort_order: the candidate order of dmz_stdsort.h on caller-supplied key lists (diagnostic entry
// dmz_hip_expiry_sort_positions; tests/test_gpu_expiry.py checks it against the reference's own std::sort).
// One wave per list.  pos[i] = position of element i: kind 0 after the partition phase as far as the wave form follows
// it (the form k_expiry_seg uses: marked elements -- marks[i] != 0, all if marks is null -- are ordered among equal keys by
// "key descending, then pos"), kinds 1 / 2 after the whole sort on one lane (9- / 7-bit index field).
// flags[list] = 1 when the wave form hit the depth limit and the serial form took over.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_sort_order(const int *__restrict__ keys, const int *__restrict__ marks,
                                                   const int *__restrict__ lens, int stride, int kind, int *__restrict__ pos,
                                                   int *__restrict__ flags) {
  __shared__ unsigned sv[448];
  __shared__ unsigned tb[dmzsort::TPAIRS];
  __shared__ unsigned stack[36];
  const int list = blockIdx.x, lane = threadIdx.x;
  const int n = lens[list];
  const int *k = keys + (size_t)list * stride;
  const int *mk = marks ? marks + (size_t)list * stride : nullptr;
  int *po = pos + (size_t)list * stride;
  int fell_back = 0;
  const int sh = kind == 2 ? 7 : 9;
  for (int i = lane; i < n; i += 64)
    sv[i] = ((unsigned)k[i] << sh) | (unsigned)i | ((kind == 0 && (!mk || mk[i])) ? dmzsort::MARK : 0u);
  __syncthreads();
  bool sorted = false;
  if (kind == 0) {
    sorted = dmzsort::wave_mark_partitions<9, 0xFFFFFu>(sv, n, lane, tb, stack);
    __syncthreads();
    if (!sorted) {
      fell_back = 1;
      for (int i = lane; i < n; i += 64) sv[i] = ((unsigned)k[i] << 9) | (unsigned)i;
      __syncthreads();
    }
  }
  if (!sorted) {
    if (lane == 0) {
      if (sh == 9) dmzsort::serial_sort<9>(sv, n, stack);
      else dmzsort::serial_sort<7>(sv, n, stack);
    }
    __syncthreads();
  }
  for (int i = lane; i < n; i += 64) po[sv[i] & ((1u << sh) - 1u)] = i;
  if (lane == 0) flags[list] = fell_back;
}

}  // namespace

int dmz_configure_expiry(void) {
#ifndef DMZ_XCAT_PAD  /* developer ablation: extra dynamic LDS = fewer workgroups per CU */
#define DMZ_XCAT_PAD 0
#endif
  const void *kernels[8] = {(const void *)k_expiry_cat<0>,   (const void *)k_expiry_cat<1>,   (const void *)k_expiry_cat<2>,
                            (const void *)k_expiry_cat<3>,   (const void *)k_expiry_model<0>, (const void *)k_expiry_model<1>,
                            (const void *)k_expiry_model<2>, (const void *)k_expiry_model<3>};
  for (int i = 0; i < 8; i++) {
    hipError_t e = hipFuncSetAttribute(kernels[i], hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)sizeof(CatLds) + DMZ_XCAT_PAD);
    if (e != hipSuccess) return (int)e;
  }
  return 0;
}

void dmz_launch_expiry(hipStream_t s, const float *weights, const float *xw, const DmzExpiryTables *tables,
                       const uint8_t *cards, size_t card_stride, int n, const dmz_hip_frame_result *results,
                       DmzExpiryStage *stage, dmz_hip_expiry_result *out, hipEvent_t mid, int conv_mode, int phases) {
  if (phases & 1) {
#if DMZ_XSEG_FUSED
    DMZ_REPEAT(xseg)
    hipLaunchKernelGGL(k_expiry_seg_fused, dim3((unsigned)n), dim3(64), DMZ_LDS_PAD, s, weights, xw, cards, card_stride, n, results,
                       out, stage);
#else
    DMZ_REPEAT(stripes)
    hipLaunchKernelGGL(k_expiry_stripes, dim3((unsigned)n), dim3(64), 0, s, cards, card_stride, n, results, out, stage);
    DMZ_REPEAT(xseg)
    hipLaunchKernelGGL(k_expiry_seg, dim3((unsigned)n * 3), dim3(64), DMZ_LDS_PAD, s, weights, xw, cards, card_stride, n, results,
                       out, stage);
#endif
  }
  if (mid) (void)hipEventRecord(mid, s);
  if (!(phases & 2)) return;
  const dim3 grid((unsigned)n), block(XC_THREADS);
  const size_t lds = sizeof(CatLds) + DMZ_XCAT_PAD;
  if (conv_mode == DMZ_HIP_EXPIRY_CONV_F32)
    hipLaunchKernelGGL(k_expiry_cat<DMZ_HIP_EXPIRY_CONV_F32>, grid, block, lds, s, weights, xw, tables, cards, card_stride, n,
                       results, stage, out);
  else if (conv_mode == DMZ_HIP_EXPIRY_CONV_BF16)
    hipLaunchKernelGGL(k_expiry_cat<DMZ_HIP_EXPIRY_CONV_BF16>, grid, block, lds, s, weights, xw, tables, cards, card_stride, n,
                       results, stage, out);
  else if (conv_mode == DMZ_HIP_EXPIRY_CONV_F16X3)
    DMZ_REPEAT(xcat)
    hipLaunchKernelGGL(k_expiry_cat<DMZ_HIP_EXPIRY_CONV_F16X3>, grid, block, lds, s, weights, xw, tables, cards, card_stride, n,
                       results, stage, out);
  else
    hipLaunchKernelGGL(k_expiry_cat<DMZ_HIP_EXPIRY_CONV_BF16X3>, grid, block, lds, s, weights, xw, tables, cards, card_stride,
                       n, results, stage, out);
}

#ifdef DMZ_XSEG_TL
extern "C" void dmz_dbg_xseg_tl(unsigned long long *out, int reset) {
  hipDeviceSynchronize();
  hipMemcpyFromSymbol(out, HIP_SYMBOL(g_xs_tl), sizeof(unsigned long long) * 16);
  if (reset) {
    unsigned long long z[16] = {0};
    hipMemcpyToSymbol(HIP_SYMBOL(g_xs_tl), z, sizeof(z));
  }
}
#endif
#ifdef DMZ_XSEG_DBG
extern "C" void dmz_dbg_xseg(unsigned long long *out, int reset) {
  hipDeviceSynchronize();
  hipMemcpyFromSymbol(out, HIP_SYMBOL(g_xs_dbg), sizeof(unsigned long long) * 16);
  if (reset) {
    unsigned long long z[16] = {0};
    hipMemcpyToSymbol(HIP_SYMBOL(g_xs_dbg), z, sizeof(z));
  }
}
#endif

void dmz_launch_sort_order(hipStream_t s, const int *keys, const int *marks, const int *lens, int n_lists, int stride, int kind,
                           int *pos, int *flags) {
  hipLaunchKernelGGL(k_sort_order, dim3((unsigned)n_lists), dim3(64), 0, s, keys, marks, lens, stride, kind, pos, flags);
}

void dmz_launch_slash_model(hipStream_t s, const float *weights, const float *xw, const float *x, int n, float *out) {
  hipLaunchKernelGGL(k_slash_model, dim3((unsigned)n), dim3(64), 0, s, weights, xw, x, n, out);
}

void dmz_launch_expiry_model(hipStream_t s, const float *weights, const float *xw, const float *x, int n, float *out,
                             int conv_mode) {
  const dim3 grid((unsigned)((n + 3) / 4)), block(XC_THREADS);
  if (conv_mode == DMZ_HIP_EXPIRY_CONV_F32)
    hipLaunchKernelGGL(k_expiry_model<DMZ_HIP_EXPIRY_CONV_F32>, grid, block, sizeof(CatLds), s, weights, xw, x, n, out);
  else if (conv_mode == DMZ_HIP_EXPIRY_CONV_BF16)
    hipLaunchKernelGGL(k_expiry_model<DMZ_HIP_EXPIRY_CONV_BF16>, grid, block, sizeof(CatLds), s, weights, xw, x, n, out);
  else if (conv_mode == DMZ_HIP_EXPIRY_CONV_F16X3)
    hipLaunchKernelGGL(k_expiry_model<DMZ_HIP_EXPIRY_CONV_F16X3>, grid, block, sizeof(CatLds), s, weights, xw, x, n, out);
  else
    hipLaunchKernelGGL(k_expiry_model<DMZ_HIP_EXPIRY_CONV_BF16X3>, grid, block, sizeof(CatLds) + DMZ_XCAT_PAD, s, weights, xw, x, n, out);
}
