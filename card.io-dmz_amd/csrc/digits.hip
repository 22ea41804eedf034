// digits.hip -- categorisation of the 15/16 card-number digits of each card.
//
// Replaces number_scores (scan/n_categorize.cpp:75-107), scores_for_number_image
// (n_categorize.cpp:45-71) and the usable gate of scan/frame.cpp:63-64: per digit ROI
// (offset, y_offset, 19, 27): 5-tap cross morphological gradient clamped at the ROI edge
// (cv/morph.cpp:190-220), llcv_equalize_hist (cv/stats.cpp:116-159), x 1/255, three CNNs
// (modelc_{5c241121,01266c1b,b00bf70c}.cpp:1893-1937: 8 3x3 correlations computing 24x15
// outputs, 3x3 max-pool, +bias, tanh, FC 320->32 tanh, FC 32->10 softmax) and the
// (sum - max) / 2 vote.
//
// Integer work (gradient, histogram, LUT, labels) is bit-exact.  The CNN arithmetic differs from the
// reference's float evaluation order (matrix-core accumulation, a v_exp_f32-based tanh, the device
// expf): contract |delta| <= 1e-4 on the scores (the reference's own KAT tolerance of 1e-5 is met by the
// device models, tests/test_gpu_stages.py).
//
// CDNA4 mapping (round 3): one workgroup (4 waves) per card, 31 KB of LDS (five cards per CU).
//   * gradient / histogram / LUT: a wave per digit, 57 lanes x 9 pixels; the gradient bytes stay in
//     registers, the histogram is wave-private (512 B), the equalised pixel is written ONCE, as the bf16
//     number the convolution wants (a byte is a bf16 number exactly): xb[digit][27][20].
//   * the 3x3 convolution of ALL THREE models as one matrix-core product per tile:
//     v_mfma_f32_16x16x32_bf16 with M = 16 conv positions (2 digits x 8 pooled rows, one conv column and
//     one (ox, j) offset inside the pool window per tile), N = 2 x 16 >= 24 maps (3 models x 8), and
//     K = 32 >= 27 = 9 taps x 3 bf16 parts of the weight / 255 (hi + mid + lo = the fp32 weight to 2^-24):
//     lane group kg = lane >> 4 < 3 holds eight taps against part kg, group 3 the ninth tap against all
//     three parts.  The eight taps of a lane are three ALIGNED 32-bit LDS reads (horizontal pairs; every
//     row of a tile has the same conv column, so the pairs' alignment is a compile-time property of the
//     tile and selects one of two K layouts) and two 16-bit reads -- no conversion, one v_lshl_or to pack.
//     The max-pool is an elementwise max over the nine tiles of a pool window (v_max3_f32), bias and
//     tanh run on the accumulators: 8 instead of ~400 VALU instructions per 16 positions x 24 maps.
//   * FC 320->32 accumulated per pooled column: the pooled activations of one column (16 digits x 3 models x 8 maps x 8 rows
//     = 12 KB, XOR-swizzled 16-byte runs) are all that ever exists in LDS.  Rounds 3 - 5: v_mfma_f32_16x16x4_f32, wave q
//     multiplying K-quarter q into six accumulators.  Round 6: v_mfma_f32_16x16x32_f16 with both operands in two f16 parts
//     (three products, 2^-22; the epilogue writes the tanh values as hi / lo f16 planes), wave w = k-step w >> 1 of n-tile
//     w & 1 -- nine 16-cycle matrix instructions per wave and column instead of twenty-four 32-cycle ones that, being fp32,
//     share the VALU's multipliers and overlap with nothing (profiles/r6_mfma_valu_overlap_probe.log).
//   * hidden tanh, FC 32->10, softmax, vote, arg-max and the usable gate as before.
//   * Round 6, LDS banks (32 banks, a half-wave per cycle): the digit stride is 272 dwords, not 270 -- at 270 the sixteen
//     (digit, pooled row) addresses of EVERY conv read met two-way on one bank (XD below); n-tile 1's two half-empty row-tiles
//     share one tanh / split / store pass.
#include <float.h>

#include "dmz_hip_internal.h"
#include "dmz_wave.h"

// developer ablation (tools/ablate.sh): extra dynamic LDS per workgroup = fewer workgroups per CU
#ifndef DMZ_DG_NT1_MERGE
#define DMZ_DG_NT1_MERGE 1
#endif
#ifndef DMZ_DG_SPLIT_MIX
#define DMZ_DG_SPLIT_MIX 1
#endif
#ifndef DMZ_LDS_PAD
#define DMZ_LDS_PAD 0
#endif

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) unsigned char lds_u8;

__device__ __forceinline__ int imax(int a, int b) { return a > b ? a : b; }
__device__ __forceinline__ int imin(int a, int b) { return a < b ? a : b; }

// developer ablation (tools/ablate.sh): return after phase k
#ifndef DMZ_DIGITS_STOP
#define DMZ_DIGITS_STOP 99
#endif
#define DG_STOP(k, expr) if (DMZ_DIGITS_STOP == (k)) { if (tid == 0) res->number_score = (float)(expr); return; }

// developer probe: -DDMZ_DG_TIMING makes thread 0 leave the cycle counter at the phase boundaries in scores[15][..]
#ifdef DMZ_DG_TIMING
__device__ long long g_dg_t[16];
#define DG_T(i) if (threadIdx.x == 0 && blockIdx.x == gridDim.x / 2) g_dg_t[i] = clock64();
// (-DDMZ_DG_TIMING2 as well: every WAVE of a few sampled workgroups leaves its own timeline of pooled column 0 ..)
#ifdef DMZ_DG_TIMING2
__device__ long long g_dg_w[4][4][48];  // [sampled workgroup][wave][point]
#define DG_W(i)                                                                                                      \
  if ((threadIdx.x & 63) == 0 && (blockIdx.x & 4095) == 2049 && (blockIdx.x >> 12) < 4)                               \
    g_dg_w[blockIdx.x >> 12][threadIdx.x >> 6][i] = (long long)__builtin_readcyclecounter();
#endif
#else
#define DG_T(i)
#endif
#ifndef DG_W
#define DG_W(i)
#endif
constexpr int DG_THREADS = 256;
constexpr int XS = 20;                 // xb row stride in elements (19 used; even: horizontal pairs stay dword-aligned)
// Elements per digit: 27 rows + 4 elements of padding = 272 dwords.  The LDS serves a half-wave (32 lanes) per cycle over 32 banks.
// The sixteen (digit, pooled row) addresses of a conv read lie 30 dwords (three rows) apart over the eight pooled rows -- the
// even banks 0, 30, .., 18 -- and one digit = D dwords apart: at D = 270 (= 14 mod 32) the second digit's rows sit on 14, 12,
// .., 0 and bank 0 is hit twice: EVERY conv read took two passes (SQ_LDS_BANK_CONFLICT 41 % of SQ_LDS_IDX_ACTIVE, the LDS 78 %
// busy: it, not the VALU, paced the kernel).  D = 16 mod 32 puts the second digit on the other eight even banks; lane group 3's
// reads (21 / 19 / 9 / -1 dwords beside the others') and the 16-bit singles (one dword beside) fall on the odd banks.
#ifndef DMZ_DG_XPAD
#define DMZ_DG_XPAD 4
#endif
constexpr int XD = 27 * XS + DMZ_DG_XPAD;  // elements per digit
constexpr int XPLANE = 16 * XD * 2;    // bytes of one bf16 plane of the 16 digits: 17,408
constexpr int CHUNK_BYTES = 3 * 16 * 64 * 4;  // pooled activations of one pooled column: [model][digit][map 8 x row 8]
constexpr int PART_BYTES = 4 * 3 * 16 * 32 * 4;  // FC1 partial sums of the four waves
constexpr int HID_PITCH = dmzv::DT_PITCH;     // floats per row of the hidden activations / of the logistic weights
constexpr int DG_RAW = 31616;                 // xb + chunk (29,696 B), partial sums + hidden activations (31,488 B)
constexpr int DG_TAILW = dmzv::DT_FLOATS * 4; // 7,488 B

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains the vector-memory counter, i.e. every
// barrier would wait for weight fragments requested ahead of their use and for result stores already on their way;
// the kernels below exchange data between waves through LDS alone.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// tanh(x) = 1 - 2 / (exp(2x) + 1) on v_exp_f32 / v_rcp_f32: |error| ~ 2e-7 absolute
__device__ __forceinline__ float fast_tanh(float x) {
  const float e = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);  // 2 * log2(e)
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
}

// Lane state of the convolution: byte addresses (LDS) of this lane's reads for row-tile 0, before the
// tile's compile-time offset.  PAR = parity of the tile's conv column c = 3 pc + ox.
//   even c, kg < 3: pairs (R + 0/1/2, c .. c+1), singles (R + 0/1, c+2);   kg = 3: the pair (R+2, c+2 .. c+3) three times
//   odd c,  kg < 3: pairs (R + 0/1/2, c+1 .. c+2), singles (R + 0/1, c);   kg = 3: the pair (R+2, c-1 .. c) three times
struct ConvLane {
  const lds_u8 *p[2][3];  // [parity][pair read]
  const lds_u8 *s[2];     // [parity] first single (the second is one row below)
};

__device__ __forceinline__ void conv_lane_init(ConvLane &cl, const lds_u8 *xb, int wave, int lane) {
  const int i = lane & 15, kg = lane >> 4;
  const int dd = i >> 3, pr = i & 7;
  const lds_u8 *base = xb + ((4 * wave + dd) * XD + 3 * pr * XS) * 2;
  const bool k3 = kg == 3;
#pragma unroll
  for (int r = 0; r < 3; r++) {
    cl.p[0][r] = base + (k3 ? (2 * XS + 2) * 2 : r * XS * 2);
    cl.p[1][r] = base + (k3 ? (2 * XS - 1) * 2 : (r * XS + 1) * 2);
  }
  cl.s[0] = base + (k3 ? 0 : 4);
  cl.s[1] = base;
}

// One tile: D[16 positions][2 x 16 maps] for the conv positions (row 3 pr + J, column C) of row-tile T.
// NPL input planes (1: bytes as bf16; 3: hi / mid / lo planes of a float input, smallest first).
// The pair reads are `volatile`: left alone, the load/store optimiser merges the same lane pointer's reads of two different
// tiles into one ds_read2_b32, whose register pair then has to be copied into the two A-operand tuples (214 v_mov_b32 and 94
// address adds per card for the offsets ds_read2 cannot encode; 44 and 25 with plain ds_read_b32, which carries a 16-bit offset).
#ifndef DG_VOL
#define DG_VOL volatile
#endif
template <int NPL, int T, int C, int J>
__device__ __forceinline__ void conv_tile(const ConvLane &cl, const bf16x8 (&bw)[2][2], f32x4 &o0, f32x4 &o1) {
  constexpr int PAR = C & 1;
  // (the parity's own +1 / +2 / -1 column shifts are in the lane addresses)
  constexpr int IMM = (T * 2 * XD + J * XS + C) * 2;
  o0 = (f32x4){0.f, 0.f, 0.f, 0.f};
  o1 = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int pl = NPL - 1; pl >= 0; pl--) {
    const int off = IMM + pl * XPLANE;
    u32x4 a;
    a.x = *(DG_VOL const __attribute__((address_space(3))) uint32_t *)(cl.p[PAR][0] + off);
    a.y = *(DG_VOL const __attribute__((address_space(3))) uint32_t *)(cl.p[PAR][1] + off);
    a.z = *(DG_VOL const __attribute__((address_space(3))) uint32_t *)(cl.p[PAR][2] + off);
    const uint32_t t0 = *(const __attribute__((address_space(3))) unsigned short *)(cl.s[PAR] + off);
    const uint32_t t1 = *(const __attribute__((address_space(3))) unsigned short *)(cl.s[PAR] + off + XS * 2);
    a.w = (t1 << 16) | t0;
    const bf16x8 av = __builtin_bit_cast(bf16x8, a);
    o0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bw[PAR][0], o0, 0, 0, 0);
    o1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bw[PAR][1], o1, 0, 0, 0);
  }
}

__device__ __forceinline__ f32x4 max3v(f32x4 a, f32x4 b, f32x4 c) {
  f32x4 r;
#pragma unroll
  for (int v = 0; v < 4; v++) r[v] = __builtin_fmaxf(__builtin_fmaxf(a[v], b[v]), c[v]);
  return r;
}

// the nine tiles of the pool windows of pooled column PC, row-tile T: elementwise max
template <int NPL, int T, int PC>
__device__ __forceinline__ void conv_pool_column(const ConvLane &cl, const bf16x8 (&bw)[2][2], f32x4 &m0, f32x4 &m1) {
  f32x4 a0, a1, b0, b1;
  conv_tile<NPL, T, 3 * PC + 0, 0>(cl, bw, m0, m1);
  conv_tile<NPL, T, 3 * PC + 1, 0>(cl, bw, a0, a1);
  conv_tile<NPL, T, 3 * PC + 2, 0>(cl, bw, b0, b1);
  m0 = max3v(m0, a0, b0), m1 = max3v(m1, a1, b1);
  conv_tile<NPL, T, 3 * PC + 0, 1>(cl, bw, a0, a1);
  conv_tile<NPL, T, 3 * PC + 1, 1>(cl, bw, b0, b1);
  m0 = max3v(m0, a0, b0), m1 = max3v(m1, a1, b1);
  conv_tile<NPL, T, 3 * PC + 2, 1>(cl, bw, a0, a1);
  conv_tile<NPL, T, 3 * PC + 0, 2>(cl, bw, b0, b1);
  m0 = max3v(m0, a0, b0), m1 = max3v(m1, a1, b1);
  conv_tile<NPL, T, 3 * PC + 1, 2>(cl, bw, a0, a1);
  conv_tile<NPL, T, 3 * PC + 2, 2>(cl, bw, b0, b1);
  m0 = max3v(m0, a0, b0), m1 = max3v(m1, a1, b1);
}

// The three CNNs for the 16 digit patches in xb (NPL bf16 planes): prob[model][digit][10] at the start of `raw`.
// `raw` is the workgroup's whole LDS block: xb planes first, then the chunk region; the FC1 partial sums,
// the hidden activations and finally the probabilities overlay it from offset 0 once the convolutions are done.
// All 256 threads call this; it starts and ends with a barrier of its own.
// conv weights of a lane: B fragments [parity][n-tile], biases of its two maps (pre-multiplied: see the epilogue)
constexpr float kTanhK = 2.8853900817779268f;  // 2 log2(e)
struct ConvWeights {
  bf16x8 bw[2][2];
  float bias0, bias1;
};
__device__ __forceinline__ void conv_weights_load(ConvWeights &cw, const float *__restrict__ hidw, int lane) {
  const u32x4 *bf = (const u32x4 *)(hidw + dmzv::WFRAG + dmzv::DCONV_B);
#pragma unroll
  for (int par = 0; par < 2; par++)
#pragma unroll
    for (int nt = 0; nt < 2; nt++) cw.bw[par][nt] = __builtin_bit_cast(bf16x8, bf[(par * 2 + nt) * 64 + lane]);
  // tanh(m + b) = 1 - 2 / (exp2(k m + k b) + 1), k = 2 log2(e): the bias enters the exponent's fma
  cw.bias0 = kTanhK * hidw[dmzv::WFRAG + dmzv::DCONV_BIAS + (lane & 15)];
  // (n-tile 1 has eight maps: lanes 8 - 15 of a row hold the bias of map lane & 7 as well -- they receive the other row-tile's
  // values of that map when the two tiles' halves are merged, digits_cnn)
  cw.bias1 = kTanhK * hidw[dmzv::WFRAG + dmzv::DCONV_BIAS + 16 + (lane & 7)];
}

template <int NPL>
__device__ __forceinline__ void digits_cnn(const float *__restrict__ wts, const float *__restrict__ hidw, const ConvWeights &cw,
                                           unsigned char *raw, const float *tw /* LDS: dmzv::DTAIL block */, int nd, int tid) {
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const lds_u8 *xb = (const lds_u8 *)raw;
  float *chunk = (float *)(raw + NPL * XPLANE);
  const bf16x8 (&bw)[2][2] = cw.bw;
  const float bias0 = cw.bias0, bias1 = cw.bias1;
  ConvLane cl;
  conv_lane_init(cl, xb, wave, lane);
  // epilogue: this lane's four pooled rows 4 (g & 1) .. + 3 of digit 4 wave + 2 T + (g >> 1), map lane & 15 of n-tile
  // nt -> one 16-byte granule of chunk[model][digit][map * 8 + row], granule index XORed with the digit
  const int n16 = lane & 15, g = lane >> 4;
  int cst[2][2];  // (the f16 form uses [t][0] only: n-tile 1 sits a constant two models further on)
  [[maybe_unused]] const bool st1 = n16 < 8;  // n-tile 1 holds model 2 in its first eight columns, nothing beyond
#if DMZ_DG_FC1_F16
  // Round 6: FC 320 -> 32 on v_mfma_f32_16x16x32_f16 with both operands in two f16 parts and the three products that carry
  // 2^-22 (lo * hi, hi * lo, hi * hi; fp32 accumulation) -- nine 16-cycle matrix instructions per wave and pooled column where
  // the f32 form (v_mfma_f32_16x16x4_f32) issued twenty-four 32-cycle ones: the probe that simply dropped three quarters of
  // those ran the digits stage 2.29 -> 2.06 ms (profiles/r6_digits_fc1_f16_ab.log).  The pooled activations of a column are
  // TWO f16 planes [part][model][digit][k = map * 8 + row] (128 B per row; the 16-byte run of a map XORed with digit & 7: a
  // lane's A fragment -- eight consecutive k -- is one aligned ds_read_b128 per part); the epilogue splits a tanh value into its
  // f16 truncation and the f16 truncation of the remainder.  Wave w takes k-step w >> 1 of n-tile w & 1 for the three models.
  constexpr int kPlane = 3 * 16 * 128;  // bytes of one f16 plane of a column's activations
#pragma unroll
  for (int t = 0; t < 2; t++) {
    const int m = n16 >> 3, map = n16 & 7, digit = 4 * wave + 2 * t + (g >> 1);
    cst[t][0] = (m * 16 + digit) * 128 + ((map ^ (digit & 7)) << 4) + ((g & 1) << 3);  // byte offset inside a plane
  }
  f32x4 fc[3];
#pragma unroll
  for (int m = 0; m < 3; m++) fc[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int fks = wave >> 1, fnt = wave & 1;
  const int fa = n16 * 128 + (((4 * fks + g) ^ (n16 & 7)) << 4);  // row digit = lane & 15, run 4 ks + (lane >> 4)
  const u32x4 *fcw = (const u32x4 *)hidw + (size_t)((fks * 2 + fnt) * 2) * 64 + lane;  // [m][pc][ks][nt][part][lane]
#else
#pragma unroll
  for (int t = 0; t < 2; t++)
#pragma unroll
    for (int nt = 0; nt < 2; nt++) {
      const int n = 16 * nt + n16, m = n >> 3, map = n & 7, digit = 4 * wave + 2 * t + (g >> 1);
      cst[t][nt] = (m * 16 + digit) * 64 + (((map * 2 + (g & 1)) ^ digit) << 2);
    }
  // FC1: wave q = K-quarter q of every chunk; A granule (4 q + kq) ^ digit of row digit = lane & 15
  f32x4 fc[3][2];
#pragma unroll
  for (int m = 0; m < 3; m++)
#pragma unroll
    for (int nt = 0; nt < 2; nt++) fc[m][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int fa = n16 * 64 + (((4 * wave + g) ^ n16) << 2);
  const f32x4 *fcw = (const f32x4 *)hidw + (size_t)wave * 2 * 64 + lane;  // [m][pc][q][nt][lane]
#endif

#if DMZ_DG_FC1_F16 && DMZ_DG_NT1_MERGE
  // n-tile 1 carries eight maps in sixteen columns: the upper half of every row of lanes would run its tanh, split and stores on
  // nothing.  Row-tile 0's n-tile-1 maxima wait for row-tile 1's, which move eight lanes up (one DPP move per register) into
  // the idle half; one tanh / split / store pass then serves both.  cstm: where a lane's merged values go.
  f32x4 m1keep = {0.f, 0.f, 0.f, 0.f};
  // (one DPP move with every lane active, lanes 0 - 7 of a row keeping their own: a DPP read of a lane that EXEC has switched off
  // returns nothing)
  const int cstm = __builtin_amdgcn_update_dpp(cst[0][0], cst[1][0], 0x118 /* row_shr:8 */, 0xf, 0xc /* lanes 8 - 15 */, false) + 2 * 16 * 128;
#endif
  auto epilogue = [&](int t, const f32x4 &m0, const f32x4 &m1) {
    f32x4 v0, v1;
#if DMZ_DG_FC1_F16
    // (n-tile 0 is computed, split and stored before n-tile 1 is touched: the kernel sits at its 128 registers)
#pragma unroll
    for (int v = 0; v < 4; v++)
      v0[v] = fmaf(__builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(fmaf(m0[v], kTanhK, bias0)) + 1.0f), -2.0f, 1.0f);
#else
#pragma unroll
    for (int v = 0; v < 4; v++) {
      v0[v] = fmaf(__builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(fmaf(m0[v], kTanhK, bias0)) + 1.0f), -2.0f, 1.0f);
      v1[v] = fmaf(__builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(fmaf(m1[v], kTanhK, bias1)) + 1.0f), -2.0f, 1.0f);
    }
#endif
#if DMZ_DG_FC1_F16
    // v = hi + lo to 2^-22 |v| (truncations: v - hi is exact in fp32; f16 subnormals are kept by the matrix core)
    auto split4 = [](const f32x4 &v, uint32_t (&h)[2], uint32_t (&l)[2]) {
#pragma unroll
      for (int p = 0; p < 2; p++) {
        typedef __fp16 f16x2 __attribute__((ext_vector_type(2)));
        const f16x2 hh = __builtin_amdgcn_cvt_pkrtz(v[2 * p], v[2 * p + 1]);
#if DMZ_DG_SPLIT_MIX
        // v - (float)half in one instruction (exactly the subtraction: the product with -1 is exact)
        float r0, r1;
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(__builtin_bit_cast(uint32_t, hh)), "v"(v[2 * p]));
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(__builtin_bit_cast(uint32_t, hh)), "v"(v[2 * p + 1]));
        const f16x2 ll = __builtin_amdgcn_cvt_pkrtz(r0, r1);
#else  /* (v_cvt_f32_f16 + v_sub_f32 per value: digits stage +0.01 ms) */
        const f16x2 ll = __builtin_amdgcn_cvt_pkrtz(v[2 * p] - (float)hh[0], v[2 * p + 1] - (float)hh[1]);
#endif
        h[p] = __builtin_bit_cast(uint32_t, hh);
        l[p] = __builtin_bit_cast(uint32_t, ll);
      }
    };
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    unsigned char *const cb = (unsigned char *)chunk + cst[t][0];
    uint32_t h[2], l[2];
    split4(v0, h, l);
    *(u32x2 *)cb = (u32x2){h[0], h[1]};
    *(u32x2 *)(cb + kPlane) = (u32x2){l[0], l[1]};
    __builtin_amdgcn_sched_barrier(0);
#if DMZ_DG_NT1_MERGE
    if (t == 0) {
      m1keep = m1;
    } else {
      f32x4 mm;
#pragma unroll
      for (int v = 0; v < 4; v++)  // lanes 8 - 15 of a row: row-tile 1's value of lane - 8; lanes 0 - 7 keep row-tile 0's
      {
        const float keep = m1keep[v], come = m1[v];
        mm[v] = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(keep), __float_as_int(come), 0x118 /* row_shr:8 */, 0xf,
                                                           0xc /* banks 2, 3 */, false));
      }
#pragma unroll
      for (int v = 0; v < 4; v++)
        v1[v] = fmaf(__builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(fmaf(mm[v], kTanhK, bias1)) + 1.0f), -2.0f, 1.0f);
      split4(v1, h, l);
      unsigned char *const cm = (unsigned char *)chunk + cstm;
      *(u32x2 *)cm = (u32x2){h[0], h[1]};
      *(u32x2 *)(cm + kPlane) = (u32x2){l[0], l[1]};
    }
#else
    if (st1) {  // (these lanes hold model 0 in n-tile 0 and model 2 in n-tile 1: two models further on)
#pragma unroll
      for (int v = 0; v < 4; v++)
        v1[v] = fmaf(__builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(fmaf(m1[v], kTanhK, bias1)) + 1.0f), -2.0f, 1.0f);
      split4(v1, h, l);
      *(u32x2 *)(cb + 2 * 16 * 128) = (u32x2){h[0], h[1]};
      *(u32x2 *)(cb + 2 * 16 * 128 + kPlane) = (u32x2){l[0], l[1]};
    }
#endif
#else
    *(f32x4 *)(chunk + cst[t][0]) = v0;
    if (st1) *(f32x4 *)(chunk + cst[t][1]) = v1;
#endif
  };
  // B fragments of column pc are requested before the column's convolutions (L2 latency hidden behind them); the chunk
  // is read into registers between two barriers that sit close together, so that nobody waits for anybody's
  // matrix instructions: they drain while the wave convolves the next column.
#if DMZ_DG_FC1_F16
  typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
  u32x4 fb[3][2];  // [model][part]
  auto fc1_fetch = [&](int pc) {
#pragma unroll
    for (int m = 0; m < 3; m++)
#pragma unroll
      for (int part = 0; part < 2; part++) fb[m][part] = fcw[((size_t)(m * 5 + pc) * 8 + part) * 64];
  };
  auto fc1_chunk = [&]() {
    u32x4 ah[3], al[3];
    const unsigned char *const cb = (const unsigned char *)chunk;
    lds_barrier();  // the chunk is complete
    DG_W(40)
#pragma unroll
    for (int m = 0; m < 3; m++) {
      ah[m] = *(const u32x4 *)(cb + m * 16 * 128 + fa);
      al[m] = *(const u32x4 *)(cb + kPlane + m * 16 * 128 + fa);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    lds_barrier();  // ... and in everybody's registers: the next column may overwrite it
    DG_W(41)
#pragma unroll
    for (int m = 0; m < 3; m++)  // small terms first
      fc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, al[m]), __builtin_bit_cast(f16x8, fb[m][0]), fc[m], 0, 0, 0);
#pragma unroll
    for (int m = 0; m < 3; m++)
      fc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, ah[m]), __builtin_bit_cast(f16x8, fb[m][1]), fc[m], 0, 0, 0);
#pragma unroll
    for (int m = 0; m < 3; m++)
      fc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, ah[m]), __builtin_bit_cast(f16x8, fb[m][0]), fc[m], 0, 0, 0);
  };
#else
  f32x4 fb[3][2];
  auto fc1_fetch = [&](int pc) {
#pragma unroll
    for (int m = 0; m < 3; m++)
#pragma unroll
#ifdef DMZ_DG_NOFC1LOAD  /* developer probe (timing only, wrong results): one cache line instead of the column's fragments */
      for (int nt = 0; nt < 2; nt++) fb[m][nt] = fcw[0];
#else
      for (int nt = 0; nt < 2; nt++) fb[m][nt] = fcw[((size_t)(m * 5 + pc) * 4 * 2 + nt) * 64];
#endif
  };
  auto fc1_chunk = [&]() {
    f32x4 a[3];
    lds_barrier();  // the chunk is complete
    DG_W(40)  // (the last column's value stays: points 40 / 41 = the two barriers of column 4)
#pragma unroll
    for (int m = 0; m < 3; m++) a[m] = *(const f32x4 *)(chunk + m * 16 * 64 + fa);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    lds_barrier();  // ... and in everybody's registers: the next column may overwrite it
    DG_W(41)
#ifdef DMZ_DG_FC1_PROBE  /* developer probe (timing only, wrong scores): DMZ_DG_FC1_PROBE of the chunk's 24 matrix instructions (6: what an
                            f16 x 3 form on v_mfma_f32_16x16x32_f16 would issue per wave and column, at half the cycles each) */
#pragma unroll
    for (int i = 0; i < DMZ_DG_FC1_PROBE; i++)
      fc[i % 3][(i / 3) & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i % 3][i & 3], fb[i % 3][(i / 3) & 1][i & 3], fc[i % 3][(i / 3) & 1], 0, 0, 0);
#else
#pragma unroll
    for (int e = 0; e < 4; e++)
#pragma unroll
      for (int m = 0; m < 3; m++)
#pragma unroll
        for (int nt = 0; nt < 2; nt++)
          fc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m][e], fb[m][nt][e], fc[m][nt], 0, 0, 0);
#endif
  };
#endif
  lds_barrier();  // xb is complete, the chunk region (number strip / histograms until now) is free
  DG_T(3)
  {
    f32x4 m0, m1;
#ifndef DMZ_DG_FETCH  /* developer probe: where the FC1 weight fragments of a column are requested (0 / 1 / 2) */
#define DMZ_DG_FETCH 2
#endif
#define DG_COLUMN(PC)                                   \
  DG_W(8 * PC + 0)                                      \
  if (DMZ_DG_FETCH == 0) fc1_fetch(PC);                 \
  conv_pool_column<NPL, 0, PC>(cl, bw, m0, m1);         \
  if (PC == 0) { DG_T(14) }                             \
  DG_W(8 * PC + 1)                                      \
  epilogue(0, m0, m1);                                  \
  if (PC == 0) { DG_T(15) }                             \
  DG_W(8 * PC + 2)                                      \
  if (DMZ_DG_FETCH == 1) fc1_fetch(PC);                 \
  conv_pool_column<NPL, 1, PC>(cl, bw, m0, m1);         \
  DG_W(8 * PC + 3)                                      \
  epilogue(1, m0, m1);                                  \
  DG_W(8 * PC + 4)                                      \
  if (DMZ_DIGITS_STOP == 3 && PC == 0) return;          \
  if (DMZ_DG_FETCH == 2) fc1_fetch(PC);                 \
  fc1_chunk();                                          \
  DG_W(8 * PC + 7)                                      \
  DG_T(4 + PC)
#ifdef DMZ_DG_REVERSE  /* developer probe */
    DG_COLUMN(4) DG_COLUMN(3) DG_COLUMN(2) DG_COLUMN(1) DG_COLUMN(0)
#else
    DG_COLUMN(0) DG_COLUMN(1) DG_COLUMN(2) DG_COLUMN(3) DG_COLUMN(4)
#endif
#undef DG_COLUMN
  }
  // ---- the four partial sums -> hidden tanh -> logistic layer (matrix core) -> exp ----
  // part[q][m][nt][v][lane]: every store and every load below touches 64 consecutive dwords
  float *part = (float *)raw;
  float *hid = (float *)(raw + PART_BYTES);  // [m][digit][HID_PITCH]
#if DMZ_DG_FC1_F16
  // (wave = (k-step, n-tile): the two k-step halves of an output meet below)
#pragma unroll
  for (int m = 0; m < 3; m++)
#pragma unroll
    for (int v = 0; v < 4; v++) part[((((fks * 3 + m) * 2 + fnt) * 4 + v) << 6) + lane] = fc[m][v];
#else
#pragma unroll
  for (int m = 0; m < 3; m++)
#pragma unroll
    for (int nt = 0; nt < 2; nt++)
#pragma unroll
      for (int v = 0; v < 4; v++) part[((((wave * 3 + m) * 2 + nt) * 4 + v) << 6) + lane] = fc[m][nt][v];
#endif
  lds_barrier();
  DG_T(9)
  if (DMZ_DIGITS_STOP == 4) return;
#pragma unroll
  for (int r = 0; r < 6; r++) {  // i = tid + 256 r = ((m * 2 + nt) * 4 + v) * 64 + lane': m = r >> 1, nt = r & 1
    const int i = tid + DG_THREADS * r;
    const int m = r >> 1, nt = r & 1, v = (tid >> 6) & 3, digit = 4 * ((tid >> 4) & 3) + v, j = 16 * nt + (tid & 15);
#if DMZ_DG_FC1_F16
    const float sum = part[i] + part[1536 + i];
#else
    const float sum = (part[i] + part[1536 + i]) + (part[2 * 1536 + i] + part[3 * 1536 + i]);
#endif
    hid[(m * 16 + digit) * HID_PITCH + j] = fast_tanh(sum + tw[dmzv::DT_HB + m * 32 + j]);
  }
  lds_barrier();
  DG_T(10)
  if (DMZ_DIGITS_STOP == 5) return;
  // logistic layer of model m on wave m: [16 digits x 32] x [32 x 16 (10 classes)] as eight v_mfma_f32_16x16x4_f32; lane
  // (row / column lane & 15, kq = lane >> 4) holds k = 8 kq .. + 7 of its row of A and its column of B (two aligned
  // 16-byte reads each; rows are 36 floats apart: conflict-free)
  float *prob = (float *)raw;  // exp(logit) [m][digit][16]: over the partial sums, which are dead now
  if (wave < 3) {
    const float *ap = hid + (wave * 16 + n16) * HID_PITCH + 8 * g;
    const float *bp = tw + dmzv::DT_LW + (wave * 16 + n16) * HID_PITCH + 8 * g;
    const f32x4 a0 = *(const f32x4 *)ap, a1 = *(const f32x4 *)(ap + 4), b0 = *(const f32x4 *)bp, b1 = *(const f32x4 *)(bp + 4);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 4; e++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[e], b0[e], acc, 0, 0, 0);
#pragma unroll
    for (int e = 0; e < 4; e++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[e], b1[e], acc, 0, 0, 0);
    // D: column (class) = lane & 15, row (digit) = 4 (lane >> 4) + v
    const float lb = tw[dmzv::DT_LB + wave * 16 + n16];
#pragma unroll
    for (int v = 0; v < 4; v++) prob[(wave * 16 + 4 * g + v) * 16 + n16] = expf(acc[v] + lb);
  }
  lds_barrier();
  DG_T(11)
}

// softmax of model m's row d at class c from the exp(logit) table of digits_cnn (Eigen's 10-element redux tree:
// ((0+1)+(2+(3+4))) + ((5+6)+(7+(8+9))))
__device__ __forceinline__ float digit_softmax(const float *__restrict__ prob, int m, int d, int c) {
  const float *pp = prob + (m * 16 + d) * 16;
  const float sum = ((pp[0] + pp[1]) + (pp[2] + (pp[3] + pp[4]))) + ((pp[5] + pp[6]) + (pp[7] + (pp[8] + pp[9])));
  return pp[c] / sum;
}

constexpr int DG_LDS = XPLANE + CHUNK_BYTES;  // 29,696 B
static_assert(PART_BYTES + 3 * 16 * HID_PITCH * 4 <= DG_RAW && DG_LDS <= DG_RAW, "partial sums + hidden activations overlay xb and the chunk");

// k_digit_patches: the equalised digit patches of a card, as bf16 numbers, [16 digits][27][20] + 4 (17,408 B per card in a
// scratch buffer; column 19 and the sixteenth digit of a 15-digit number are never written: whatever finite values they
// hold meet zero weights / are never read back).  Its own kernel because this phase is short chains of LDS round trips
// that need nothing but occupancy (64 registers, 19 KB: 32 waves per CU), which the CNN kernel (128 registers) cannot give.
// (Measured and rejected: the nine pixels of a lane straight from the card with DPP / ds_bpermute neighbours instead of the
// LDS strip -- 36 byte loads per lane keep the texture path busier than 180 LDS byte reads keep the LDS: 1.9 vs 0.9 ms
// per 65 536 cards.)
constexpr int STRIP_BYTES = 11568;     // 27 x 428 = 11,556, padded to 16
constexpr int HIST_BYTES = 16 * 512;   // (with the strip: the 16 KB of the four-fold histograms)
constexpr int DP_LDS = STRIP_BYTES + HIST_BYTES;  // 19,760 B: eight workgroups per CU
#ifndef DMZ_DP_WGS
#define DMZ_DP_WGS 8
#endif
#ifndef DMZ_DP_HROT  /* rotation of histogram copy q, in dwords per q (0: the round 3 - 5 layout) */
#define DMZ_DP_HROT 8
#endif
#ifndef DMZ_DP_STOP  /* developer ablation: k_digit_patches returns after phase k */
#define DMZ_DP_STOP 99
#endif
__global__ __launch_bounds__(DG_THREADS, DMZ_DP_WGS) void k_digit_patches(const uint8_t *__restrict__ cards, size_t card_stride, int n,
                                                               const dmz_hip_frame_result *__restrict__ results,
                                                               unsigned short *__restrict__ patches) {
  __shared__ __attribute__((aligned(16))) unsigned char raw[DP_LDS];
  unsigned char *strip_l = raw;  // 27 x 428 bytes
  const int f = blockIdx.x;
  if (f >= n) return;
  const dmz_hip_frame_result *res = results + f;
  if (!(res->flags & DMZ_HIP_FLAG_VSEG_OK)) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nd = res->n_offsets;
  const int y_off = res->vseg_y_offset;
  const int my_off = res->offsets[lane & 15];  // (one load: lane d holds offsets[d])
  const uint32_t *strip = (const uint32_t *)(cards + (size_t)f * card_stride + (size_t)y_off * DMZ_CARD_WIDTH);
  unsigned short *xb = patches + (size_t)f * (XPLANE / 2);

  // ---- number strip -> LDS (2889 aligned dwords) ----
  // (all twelve dwords of a thread requested before the first is stored: the rolled loop was load, wait, store twelve times
  // over -- twelve memory round trips at the head of every workgroup)
  {
    constexpr int kIt = (27 * 107 + DG_THREADS - 1) / DG_THREADS;  // 12
    uint32_t sv[kIt];
#pragma unroll
    for (int k = 0; k < kIt; k++) sv[k] = strip[imin(tid + k * DG_THREADS, 27 * 107 - 1)];
#pragma unroll
    for (int k = 0; k < kIt; k++)
      if (tid + k * DG_THREADS < 27 * 107) ((uint32_t *)strip_l)[tid + k * DG_THREADS] = sv[k];
  }
  __syncthreads();
  if (DMZ_DP_STOP == 1) return;
  // ---- per digit, a wave each (wave w owns digits 4 w .. 4 w + 3): cross gradient clamped at the 19x27 ROI edge,
  // histogram, equalisation LUT (stats.cpp:135-151), equalised pixels as bf16.
  // Lane l < 57 owns column l % 19 of rows l / 19, l / 19 + 3, ...: pixel index p = 57 k + l in step k, so nine steps
  // cover the 513 pixels with no per-pixel division and the five taps at constant offsets from one address.
  // The wave's four digits go through every step together. ----
  {
    const int rr = lane / 19, c = lane - 19 * rr;
    const int cl = c > 0 ? -1 : 0, cr = c < 18 ? 1 : 0;
    int gv[4][9];
#pragma unroll
    for (int dj = 0; dj < 4; dj++) {
      const int d = 4 * wave + dj;
      const int off = __builtin_amdgcn_readlane(my_off, d);
      if (d < nd && lane < 57) {
        const unsigned char *px = strip_l + off + rr * DMZ_CARD_WIDTH + c;
#pragma unroll
        for (int k = 0; k < 9; k++) {
          const int r = 3 * k + rr;
          const unsigned char *q = px + 3 * k * DMZ_CARD_WIDTH;
          const int nn = q[r > 0 ? -DMZ_CARD_WIDTH : 0], ww = q[cl], cc = q[0], ee = q[cr],
                    ss = q[r < 26 ? DMZ_CARD_WIDTH : 0];
          gv[dj][k] = imax(nn, imax(ww, imax(cc, imax(ee, ss)))) - imin(nn, imin(ww, imin(cc, imin(ee, ss))));
        }
      }
    }
    if (DMZ_DP_STOP == 2) {
      int acc = 0;
      for (int dj = 0; dj < 4; dj++)
        for (int k = 0; k < 9; k++) acc += gv[dj][k];
      if (acc == 12345678) xb[0] = 1;
      return;
    }
    // The gradients are in registers: the strip is dead once every wave is here, and its LDS becomes the histograms.
    // The pixels of a digit crowd into a few bins, and lanes that meet in one counter serialise an LDS atomic at two
    // cycles each (tools/ubench/lds_atomic_rate.hip; a probe with conflict-free addresses ran the kernel 28 % faster) -- so a
    // digit gets FOUR histograms, one per lane & 3, of 8-bit counters (a quarter of the lanes own <= 135 pixels: no carry
    // between the four counters of a dword): 4 x 256 B per digit, 4 KB per wave over the old strip; the LUT pass adds them.
    __syncthreads();
    unsigned int *hq = (unsigned int *)raw + wave * 1024;  // [digit 4][copy 4][64 dwords]
    {
      const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
      for (int i = 0; i < 4; i++) *(u32x4 *)(hq + 256 * i + 4 * lane) = z;
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int dj = 0; dj < 4; dj++)
      if (4 * wave + dj < nd && lane < 57) {
        unsigned int *hc = hq + (dj * 4 + (lane & 3)) * 64;
#pragma unroll
        // value v -> byte v >> 6 of dword v & 63: neighbouring VALUES (neighbouring lanes see similar gradients) fall into
        // different dwords / banks; only equal values, or values 64 apart, still meet in one counter word
        // (the four copies are 64 dwords = two trips round the 32 banks apart, so copy q keeps its counters rotated by 8 q dwords:
        // the lanes of a half-wave that see similar values then spread over four bank octets instead of queueing on one)
        for (int k = 0; k < 9; k++) atomicAdd(&hc[(gv[dj][k] + DMZ_DP_HROT * (lane & 3)) & 63], 1u << ((gv[dj][k] >> 6) * 8));
      }
    __builtin_amdgcn_wave_barrier();
    if (DMZ_DP_STOP == 3) return;
    uint32_t lutb[4][4];
#pragma unroll
    for (int dj = 0; dj < 4; dj++) {
      // lane l holds the counts of the values l, l + 64, l + 128, l + 192 (the four bytes of dword l of every copy, added as
      // 16-bit fields); the cumulative counts are two wave scans over packed pairs plus the totals of the quarters below
      unsigned int ev = 0u, od = 0u;
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const unsigned int cq = hq[(dj * 4 + q) * 64 + ((lane + DMZ_DP_HROT * q) & 63)];
        ev += cq & 0x00FF00FFu;         // {count(l), count(l + 128)}
        od += (cq >> 8) & 0x00FF00FFu;  // {count(l + 64), count(l + 192)}
      }
      const int p01 = (int)((ev & 0xffffu) | (od << 16)), p23 = (int)((ev >> 16) | (od & 0xffff0000u));
      const int i01 = dmzwave::inclusive_scan_i32(p01), i23 = dmzwave::inclusive_scan_i32(p23);  // (fields <= 513: no carry)
      const int t01 = __builtin_amdgcn_readlane(i01, 63), t23 = __builtin_amdgcn_readlane(i23, 63);
      const int T0 = t01 & 0xffff, T1 = (int)((unsigned)t01 >> 16), T2 = t23 & 0xffff;
      const float scale = 255.f / (19 * 27);
      const int c0 = i01 & 0xffff, c1 = T0 + (int)((unsigned)i01 >> 16), c2 = T0 + T1 + (i23 & 0xffff),
                c3 = T0 + T1 + T2 + (int)((unsigned)i23 >> 16);
      int l0 = __float2int_rn((float)c0 * scale), l1 = __float2int_rn((float)c1 * scale),
          l2 = __float2int_rn((float)c2 * scale), l3 = __float2int_rn((float)c3 * scale);
      l0 = imin(255, imax(0, l0)); l1 = imin(255, imax(0, l1));
      l2 = imin(255, imax(0, l2)); l3 = imin(255, imax(0, l3));
      if (lane == 0) l0 = 0;  // lut[0] = 0 (stats.cpp:151)
      // the LUT holds the equalised value as a bf16 number (the upper half of its float)
      lutb[dj][0] = __float_as_uint((float)l0) >> 16, lutb[dj][1] = __float_as_uint((float)l1) >> 16;
      lutb[dj][2] = __float_as_uint((float)l2) >> 16, lutb[dj][3] = __float_as_uint((float)l3) >> 16;
    }
    __builtin_amdgcn_wave_barrier();
    // the LUTs (256 bf16 each) over the first half of the wave's region
    constexpr int HW = 128;
    unsigned int *hd = hq;
#pragma unroll
    for (int dj = 0; dj < 4; dj++) {
      unsigned short *l16 = (unsigned short *)(hd + dj * HW);
#pragma unroll
      for (int q = 0; q < 4; q++) l16[lane + 64 * q] = (unsigned short)lutb[dj][q];
    }
    __builtin_amdgcn_wave_barrier();
    if (DMZ_DP_STOP == 4) return;
#pragma unroll
    for (int dj = 0; dj < 4; dj++)
      if (4 * wave + dj < nd && lane < 57) {
        const unsigned short *h16 = (const unsigned short *)(hd + dj * HW);
        unsigned short *xd = xb + (4 * wave + dj) * XD + rr * XS + c;
#pragma unroll
        for (int k = 0; k < 9; k++) xd[3 * k * XS] = h16[gv[dj][k]];
      }
  }
}

#ifndef DMZ_DIGITS_WGS  /* workgroups per CU the register allocation and the LDS layout aim at */
#define DMZ_DIGITS_WGS 4
#endif
__global__ __launch_bounds__(DG_THREADS, DMZ_DIGITS_WGS) void k_digits(const float *__restrict__ wts,
                                                        const float *__restrict__ hidw /* dmzv layout */,
                                                        const uint32_t *__restrict__ patches, int n,
                                                        dmz_hip_frame_result *__restrict__ results) {
  __shared__ __attribute__((aligned(16))) unsigned char raw[DG_RAW + DG_TAILW];
  const int f = blockIdx.x;
  if (f >= n) return;
  dmz_hip_frame_result *res = results + f;
  DG_T(0)
#ifndef DMZ_DG_PRIO  /* developer switch: issue priority of a workgroup while it requests its inputs */
#define DMZ_DG_PRIO 3
#endif
  __builtin_amdgcn_s_setprio(DMZ_DG_PRIO);
  // the card's patches (4320 dwords, fixed address: requested before the record says whether they are needed)
  constexpr int kStage = (XPLANE / 4 + DG_THREADS - 1) / DG_THREADS;  // 17
  const int tid = threadIdx.x;
  uint32_t st[kStage];
  {
#ifdef DMZ_DG_SAMEPATCH  /* developer probe (timing only, wrong results): every card reads one of 64 cache-resident patch sets */
    const uint32_t *src = patches + (size_t)(f & 63) * (XPLANE / 4);
#else
    const uint32_t *src = patches + (size_t)f * (XPLANE / 4);
#endif
#pragma unroll
    for (int k = 0; k < kStage; k++) st[k] = tid + k * DG_THREADS < XPLANE / 4 ? src[tid + k * DG_THREADS] : 0u;
  }
#ifdef DMZ_DG_LOADWAIT  /* developer probe: cycles until the card's patches have arrived (a sample of the workgroups) */
  {
    const long long t0 = (long long)__builtin_readcyclecounter();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long t1 = (long long)__builtin_readcyclecounter();
    if (tid == 0 && (blockIdx.x & 255) == 3) printf("k_digits block %d: patches arrived after %lld cycles\n", blockIdx.x, t1 - t0);
  }
#endif
  ConvWeights cw;
  conv_weights_load(cw, hidw, tid & 63);
  // the tail's weights (hidden biases, logistic layer): LDS, so that the short serial steps at the end wait for no L2
  const f32x4 *twsrc = (const f32x4 *)(hidw + dmzv::WFRAG + dmzv::DTAIL);
  const f32x4 tw0 = twsrc[tid], tw1 = tid + DG_THREADS < DG_TAILW / 16 ? twsrc[tid + DG_THREADS] : (f32x4){0.f, 0.f, 0.f, 0.f};
  __builtin_amdgcn_s_setprio(0);
  if (!(res->flags & DMZ_HIP_FLAG_VSEG_OK)) return;
  const int nd = res->n_offsets;
#pragma unroll
  for (int k = 0; k < kStage; k++)
    if (tid + k * DG_THREADS < XPLANE / 4) ((uint32_t *)raw)[tid + k * DG_THREADS] = st[k];
  ((f32x4 *)(raw + DG_RAW))[tid] = tw0;
  if (tid + DG_THREADS < DG_TAILW / 16) ((f32x4 *)(raw + DG_RAW))[tid + DG_THREADS] = tw1;
  DG_T(1)
  DG_STOP(2, raw[0] + raw[100])
  DG_T(2)

  digits_cnn<1>(wts, hidw, cw, raw, (const float *)(raw + DG_RAW), nd, tid);
  if (DMZ_DIGITS_STOP >= 3 && DMZ_DIGITS_STOP <= 7) return;

  DG_T(12)
  // ---- vote (n_categorize.cpp:69-70), arg-max, usable gate (frame.cpp:63-64) ----
  const float *prob = (const float *)raw;
  float *fin = (float *)raw + 768;  // 160 floats
  float myv = 0.0f;
  if (tid < 160) {
    const int d = tid / 10, c = tid - 10 * d;
    if (d < nd) {
      const float r0 = digit_softmax(prob, 0, d, c), r1 = digit_softmax(prob, 1, d, c), r2 = digit_softmax(prob, 2, d, c);
      float mx = r0 > r1 ? r0 : r1;
      mx = mx > r2 ? mx : r2;
      myv = (((r0 + r1) + r2) - mx) / 2.0f;
    }
    fin[tid] = myv;
  }
  lds_barrier();
  if (tid < 160) (&res->scores[0][0])[tid] = myv;  // (after the barrier: nobody waits for the stores)
  if (tid >= 64 && tid < 80) {  // (a wave of its own: beside the sequential sum below)
    const int dgt = tid - 64;
    int best = 0;
    for (int c = 1; c < 10; c++)
      if (fin[dgt * 10 + c] > fin[dgt * 10 + best]) best = c;
    res->digits[dgt] = (uint8_t)best;
  }
  if (tid == 0) {
    float sum = fin[0];
    for (int i = 1; i < 160; i++) sum = sum + fin[i];  // sequential, Redux.h:168-184
    const float number_score = (float)nd - sum;
    res->number_score = number_score;
    if (number_score < 3.0f) res->flags = res->flags | DMZ_HIP_FLAG_USABLE;
#ifdef DMZ_DG_TIMING
    if (f == n / 2) {
      g_dg_t[13] = clock64();
      for (int i = 0; i < 13; i++) res->scores[15][i % 10] = 0;  // (keeps the record valid-looking)
      for (int i = 1; i < 16; i++) (&res->scores[0][0])[i] = (float)(g_dg_t[i] - g_dg_t[0]);
    }
#endif
  }
}

// Stand-alone digit model entry point (KAT): up to 16 float patches per workgroup through the same
// convolution / pooling / matrix-core / head code.  A float input x is scaled by 255 (the conv weights
// carry the 1 / 255 of n_categorize.cpp:99) and split into three bf16 planes hi + mid + lo = 255 x
// exactly; the tile runs once per plane into the same accumulators.
constexpr int DGM_RAW = 3 * XPLANE + CHUNK_BYTES;  // (the tail's weights follow)
constexpr int DGM_LDS = DGM_RAW + DG_TAILW;
__global__ __launch_bounds__(DG_THREADS) void k_digit_model(const float *__restrict__ wts,
                                                             const float *__restrict__ hidw, int model,
                                                             const float *__restrict__ xin, int n,
                                                             float *__restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char rawm[];
  const int tid = threadIdx.x;
  const int base = blockIdx.x * 16;
  const int rows = imin(16, n - base);
  if (rows <= 0) return;
  for (int i = tid; i < 3 * XPLANE / 4; i += DG_THREADS) ((uint32_t *)rawm)[i] = 0u;
  __syncthreads();
  for (int i = tid; i < rows * 513; i += DG_THREADS) {
    const int d = i / 513, p = i - d * 513, r = p / 19, c = p - r * 19;
    const float xs = xin[(size_t)(base + d) * 513 + p] * 255.0f;
    const uint32_t hi = __float_as_uint(xs) & 0xffff0000u;  // (truncation: the remainders stay exact)
    const float r1 = xs - __uint_as_float(hi);
    const uint32_t mid = __float_as_uint(r1) & 0xffff0000u;
    const float r2 = r1 - __uint_as_float(mid);
    const uint32_t lo = __float_as_uint(r2) & 0xffff0000u;
    unsigned short *xe = (unsigned short *)rawm + d * XD + r * XS + c;
    xe[0] = (unsigned short)(hi >> 16);
    xe[XPLANE / 2] = (unsigned short)(mid >> 16);
    xe[XPLANE] = (unsigned short)(lo >> 16);
  }
  ConvWeights cw;
  conv_weights_load(cw, hidw, tid & 63);
  for (int i = tid; i < DG_TAILW / 4; i += DG_THREADS) ((float *)(rawm + DGM_RAW))[i] = hidw[dmzv::WFRAG + dmzv::DTAIL + i];
  digits_cnn<3>(wts, hidw, cw, rawm, (const float *)(rawm + DGM_RAW), rows, tid);
  if (tid < rows * 10) out[(size_t)base * 10 + tid] = digit_softmax((const float *)rawm, model, tid / 10, tid % 10);
}

}  // namespace

void dmz_launch_digits(hipStream_t s, const float *weights, const float *hidw, const uint8_t *cards,
                       size_t card_stride, int n, dmz_hip_frame_result *results, void *patches) {
  DMZ_REPEAT(patches)
  hipLaunchKernelGGL(k_digit_patches, dim3(n), dim3(DG_THREADS), 0, s, cards, card_stride, n, results,
                     (unsigned short *)patches);
  DMZ_REPEAT(digits)
  hipLaunchKernelGGL(k_digits, dim3(n), dim3(DG_THREADS), DMZ_LDS_PAD, s, weights, hidw, (const uint32_t *)patches, n,
                     results);
}
size_t dmz_digit_patch_bytes(void) { return (size_t)XPLANE; }
#ifdef DMZ_DG_TIMING2
extern "C" void dmz_dbg_digits_waves(long long *out /* 4 x 4 x 48 */) {
  (void)hipDeviceSynchronize();
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dg_w), sizeof(long long) * 4 * 4 * 48);
}
#endif

void dmz_launch_digit_model(hipStream_t s, const float *weights, const float *hidw, int model,
                            const float *x, int n, float *out) {
  hipLaunchKernelGGL(k_digit_model, dim3((n + 15) / 16), dim3(DG_THREADS), DGM_LDS, s, weights, hidw, model, x,
                     n, out);
}

int dmz_configure_scan(void) {
  hipError_t e = hipFuncSetAttribute((const void *)k_digit_model, hipFuncAttributeMaxDynamicSharedMemorySize, DGM_LDS);
  if (e != hipSuccess) return (int)e;
  return dmz_configure_vseg();
}
