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
// Integer work (gradient, histogram, LUT, labels) is bit-exact.  The CNN arithmetic uses
// fused multiply-adds, a v_exp_f32-based tanh and the device expf: contract |delta| <= 1e-4
// on the scores (the reference's own KAT tolerance of 1e-5 is met by the device models).
//
// CDNA4 mapping: one workgroup (4 waves) per card, 37 KB LDS (4 cards per CU).
//   * equalised digit patches stay u8 in LDS (16 x 513 B); the float input of the CNN is
//     rebuilt on load (one cvt + one multiply, the reference's own x * (1/255)).
//   * conv + pool: one thread per (digit, pooled position); the 5x5 patch lives in
//     registers and is reused by all 8 kernels; two kernels per instruction with
//     v_pk_fma_f32, conv weights are wave-uniform scalar loads.
//   * FC 320->32 = [16 digits x 320] x [320 x 32] on v_mfma_f32_16x16x4_f32: wave w owns
//     output tile (w & 1) and K half (w >> 1), its B operands (10 float4 per lane) come
//     straight from the row-major weight matrix; A operands are ds_read_b128 from the pooled
//     activations (row stride 324 floats: conflict-free).
#include <float.h>

#include "dmz_hip_internal.h"
#include "dmz_wave.h"

// developer ablation (tools/ablate.sh): extra dynamic LDS per workgroup = fewer workgroups per CU
#ifndef DMZ_LDS_PAD
#define DMZ_LDS_PAD 0
#endif

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int imax(int a, int b) { return a > b ? a : b; }
__device__ __forceinline__ int imin(int a, int b) { return a < b ? a : b; }

// developer ablation (tools/ablate.sh): return after phase k
#ifndef DMZ_DIGITS_STOP
#define DMZ_DIGITS_STOP 99
#endif
#define DG_STOP(k, expr) if (DMZ_DIGITS_STOP == (k)) { if (tid == 0) res->number_score = (float)(expr); return; }

constexpr int DG_THREADS = 256;
constexpr int DG_ESTRIDE = 528;  // bytes per equalised digit patch (513 used)
constexpr int DG_PSTRIDE = 324;  // floats per pooled row (320 used)

// tanh(x) = 1 - 2 / (exp(2x) + 1) on v_exp_f32 / v_rcp_f32: |error| ~ 2e-7 absolute
__device__ __forceinline__ float fast_tanh(float x) {
  const float e = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);  // 2 * log2(e)
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
}

// conv 3x3 valid -> 3x3 max pool -> + bias -> tanh for one pooled position; the 8 kernels
// share the 5x5 input patch, two kernels per v_pk_fma_f32.
// cw: the 8 x 9 conv weights (k_digits: pre-multiplied by 1/255, the input patch holds the raw byte values)
__device__ __forceinline__ void digit_conv_pool(const float (&in)[5][5], const float *__restrict__ mw,
                                                const float *__restrict__ cw, int pos, float *__restrict__ pooled /* row */) {
#ifndef DMZ_DIGITS_CONV_UNROLL
#define DMZ_DIGITS_CONV_UNROLL 2
#endif
#pragma unroll DMZ_DIGITS_CONV_UNROLL
  for (int k = 0; k < 8; k += 2) {
    f32x2 acc[9];
#pragma unroll
    for (int t = 0; t < 9; t++) {
      const f32x2 w2 = {cw[k * 9 + t], cw[(k + 1) * 9 + t]};
      const int ti = t / 3, tj = t - 3 * (t / 3);
#pragma unroll
      for (int oy = 0; oy < 3; oy++)
#pragma unroll
        for (int ox = 0; ox < 3; ox++) {
          const f32x2 x2 = {in[oy + ti][ox + tj], in[oy + ti][ox + tj]};
          // (the first tap starts the sum: no zero-initialised accumulators)
          acc[oy * 3 + ox] = t == 0 ? w2 * x2 : __builtin_elementwise_fma(w2, x2, acc[oy * 3 + ox]);
        }
    }
    f32x2 m = acc[0];
#pragma unroll
    for (int o = 1; o < 9; o++) m = __builtin_elementwise_max(m, acc[o]);
    pooled[k * 40 + pos] = fast_tanh(m.x + mw[dmzw::D_CONV_B + k]);
    pooled[(k + 1) * 40 + pos] = fast_tanh(m.y + mw[dmzw::D_CONV_B + k + 1]);
  }
}

// FC 320 -> 32 for 16 rows of `pooled` on the matrix core; part[khalf][row][32].
// The B operands (this wave's 16 hidden units x its K half) are fetched early by the caller
// so that the L2 latency hides behind the convolution.
struct Fc1B {
  f32x4 b[10];
};

__device__ __forceinline__ void digit_fc1_load(const float *__restrict__ hw /* [32][320], 16-B aligned */,
                                               int wave, int lane, Fc1B &w) {
  const int nt = wave & 1, kh = wave >> 1;
  const int ii = lane & 15, kk = lane >> 4;
  const float *bp = hw + (nt * 16 + ii) * 320 + kh * 160 + 4 * kk;
#pragma unroll
  for (int u = 0; u < 10; u++) w.b[u] = *(const f32x4 *)(bp + 16 * u);
}

__device__ __forceinline__ void digit_fc1(const Fc1B &w, const float *__restrict__ pooled,
                                          float *__restrict__ part, int wave, int lane) {
  const int nt = wave & 1, kh = wave >> 1;
  const int ii = lane & 15, kk = lane >> 4;
  const float *ap = pooled + ii * DG_PSTRIDE + kh * 160 + 4 * kk;
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int u = 0; u < 10; u += 2) {
    const f32x4 b0 = w.b[u], b1 = w.b[u + 1];
    const f32x4 a0 = *(const f32x4 *)(ap + 16 * u), a1 = *(const f32x4 *)(ap + 16 * u + 16);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, b0.x, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, b1.x, acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, b0.y, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, b1.y, acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, b0.z, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.z, b1.z, acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, b0.w, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.w, b1.w, acc1, 0, 0, 0);
  }
  // D: column (hidden unit) = lane & 15, row (digit) = 4 * (lane >> 4) + v
  __syncthreads();  // `part` lies over the pooled activations the other waves may still be reading
#pragma unroll
  for (int v = 0; v < 4; v++) part[(kh * 16 + 4 * kk + v) * 32 + nt * 16 + ii] = acc0[v] + acc1[v];
}

// hidden tanh, logistic layer, exp, softmax for `nd` rows (all 256 threads call this)
__device__ __forceinline__ void digit_head(const float *__restrict__ mw, const float *__restrict__ part,
                                           float *__restrict__ hid, float *__restrict__ prob /* [16][10] */,
                                           int nd, int tid) {
  for (int i = tid; i < 16 * 32; i += DG_THREADS) {
    const int j = i & 31;
    hid[i] = fast_tanh((part[i] + part[512 + i]) + mw[dmzw::D_HID_B + j]);
  }
  __syncthreads();
  if (tid < nd * 10) {
    const int d = tid / 10, c = tid - d * 10;
    float a = 0.0f;
#pragma unroll
    for (int j = 0; j < 32; j++) a = fmaf(mw[dmzw::D_LOG_W + c * 32 + j], hid[d * 32 + j], a);
    prob[d * 10 + c] = expf(a + mw[dmzw::D_LOG_B + c]);
  }
  __syncthreads();
  if (tid < nd) {
    float *pp = prob + tid * 10;
    // Eigen 10-element redux tree: ((0+1)+(2+(3+4))) + ((5+6)+(7+(8+9)))
    const float sum = ((pp[0] + pp[1]) + (pp[2] + (pp[3] + pp[4]))) +
                      ((pp[5] + pp[6]) + (pp[7] + (pp[8] + pp[9])));
    for (int c = 0; c < 10; c++) pp[c] = pp[c] / sum;
  }
  __syncthreads();
}

#ifndef DMZ_DIGITS_WGS  /* workgroups per CU the register allocation and the LDS layout aim at */
#define DMZ_DIGITS_WGS 5
#endif
__global__ __launch_bounds__(DG_THREADS, DMZ_DIGITS_WGS) void k_digits(const float *__restrict__ wts,
                                                        const float *__restrict__ hidw /* 3 x [32][320] */,
                                                        const uint8_t *__restrict__ cards,
                                                        size_t card_stride, int n,
                                                        dmz_hip_frame_result *__restrict__ results) {
  __shared__ __attribute__((aligned(16))) float pooled[16 * DG_PSTRIDE];  // 20,736 B; hist overlays it
  __shared__ __attribute__((aligned(16))) unsigned char eq[16 * DG_ESTRIDE];  // 8,448 B
  __shared__ float prob[3 * 16 * 10];
  // the FC1 partial sums and the hidden activations reuse the first 6 KB of `pooled` (rows 0 .. 4, rewritten by
  // every model's convolution): 31 KB per workgroup, five workgroups per CU
  float *part = pooled;               // 2 x 16 x 32
  float *hid = pooled + 2 * 16 * 32;  // 16 x 32
  // pooled is dead until the first conv: it first holds the 16 histograms (u16 counters,
  // 8 KB) and the 27 x 428 number strip (11.6 KB)
  unsigned int *hist32 = (unsigned int *)pooled;                       // 16 x 128 words
  unsigned short *hist16 = (unsigned short *)pooled;                   // 16 x 256 counters
  unsigned char *strip_l = (unsigned char *)pooled + 16 * 256 * 2;     // 27 x 428 bytes

  const int f = blockIdx.x;
  if (f >= n) return;
  dmz_hip_frame_result *res = results + f;
  if (!(res->flags & DMZ_HIP_FLAG_VSEG_OK)) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nd = res->n_offsets;
  const int y_off = res->vseg_y_offset;
  const uint32_t *strip = (const uint32_t *)(cards + (size_t)f * card_stride + (size_t)y_off * DMZ_CARD_WIDTH);

  // ---- number strip -> LDS (2889 aligned dwords), histograms cleared ----
  for (int i = tid; i < 27 * 107; i += DG_THREADS) ((uint32_t *)strip_l)[i] = strip[i];
  for (int i = tid; i < 16 * 128; i += DG_THREADS) hist32[i] = 0u;
  __syncthreads();
  // ---- per digit: cross gradient clamped at the 19x27 ROI edge, histogram ----
  // A wave per digit; lane l < 57 owns column l % 19 of rows l / 19, l / 19 + 3, ...: pixel index
  // p = 57 k + l in step k, so nine steps cover the 513 pixels with no per-pixel division and the
  // five taps at constant offsets from one address.
  if (lane < 57) {
    const int rr = lane / 19, c = lane - 19 * rr;
    const int cl = c > 0 ? -1 : 0, cr = c < 18 ? 1 : 0;
    for (int d = wave; d < nd; d += DG_THREADS / 64) {
      const unsigned char *px = strip_l + res->offsets[d] + rr * DMZ_CARD_WIDTH + c;
      unsigned char *eqd = eq + d * DG_ESTRIDE + lane;
      unsigned int *hd = hist32 + d * 128;
#pragma unroll
      for (int k = 0; k < 9; k++) {
        const int r = 3 * k + rr;
        const unsigned char *q = px + 3 * k * DMZ_CARD_WIDTH;
        const int nn = q[r > 0 ? -DMZ_CARD_WIDTH : 0], ww = q[cl], cc = q[0], ee = q[cr],
                  ss = q[r < 26 ? DMZ_CARD_WIDTH : 0];
        const int gv = imax(nn, imax(ww, imax(cc, imax(ee, ss)))) - imin(nn, imin(ww, imin(cc, imin(ee, ss))));
        eqd[57 * k] = (unsigned char)gv;
        atomicAdd(&hd[gv >> 1], 1u << ((gv & 1) * 16));  // counts <= 513: no carry
      }
    }
  }
  __syncthreads();
  DG_STOP(1, eq[0] + hist32[3])
  // ---- equalisation LUT (stats.cpp:135-151): one wave per digit, 4 bins per lane ----
  for (int d = wave; d < nd; d += DG_THREADS / 64) {
    unsigned short *h = hist16 + d * 256;
    const int h0 = h[lane * 4 + 0], h1 = h[lane * 4 + 1], h2 = h[lane * 4 + 2], h3 = h[lane * 4 + 3];
    const int tot = h0 + h1 + h2 + h3;
    const int incl = dmzwave::inclusive_scan_i32(tot);
    const int excl = incl - tot;
    const float scale = 255.f / (19 * 27);
    const int c0 = excl + h0, c1 = c0 + h1, c2 = c1 + h2, c3 = c2 + h3;
    int l0 = __float2int_rn((float)c0 * scale), l1 = __float2int_rn((float)c1 * scale),
        l2 = __float2int_rn((float)c2 * scale), l3 = __float2int_rn((float)c3 * scale);
    l0 = imin(255, imax(0, l0)); l1 = imin(255, imax(0, l1));
    l2 = imin(255, imax(0, l2)); l3 = imin(255, imax(0, l3));
    if (lane == 0) l0 = 0;  // lut[0] = 0 (stats.cpp:151)
    h[lane * 4 + 0] = (unsigned short)l0; h[lane * 4 + 1] = (unsigned short)l1;
    h[lane * 4 + 2] = (unsigned short)l2; h[lane * 4 + 3] = (unsigned short)l3;
  }
  __syncthreads();
  if (lane < 57)
    for (int d = wave; d < nd; d += DG_THREADS / 64) {
      unsigned char *eqd = eq + d * DG_ESTRIDE + lane;
#pragma unroll
      for (int k = 0; k < 9; k++) eqd[57 * k] = (unsigned char)hist16[d * 256 + eqd[57 * k]];
    }
  __syncthreads();
  DG_STOP(2, eq[0] + eq[512])
  // rows of unused digits (nd = 15) must be finite for the matrix core
  for (int i = tid; i < 16 * DG_PSTRIDE; i += DG_THREADS) pooled[i] = 0.0f;
  __syncthreads();

  // ---- three CNNs ----
  for (int m = 0; m < 3; m++) {
    const float *mw = wts + dmzw::DIGIT0 + m * dmzw::DIGIT_STRIDE;
    const float *cws = hidw + dmzv::WFRAG + dmzv::CONVS + m * 72;
    Fc1B fcb;
    if (DMZ_DIGITS_WGS < 5) digit_fc1_load(hidw + m * 32 * 320, wave, lane, fcb);  // early: 40 registers across the conv
    for (int i = tid; i < nd * 40; i += DG_THREADS) {
      const int d = i / 40, pos = i - d * 40;
      const int pr = pos / 5, pc = pos - pr * 5;
      // single-byte LDS reads (volatile only so that the compiler does not merge them into 16-bit
      // reads: at odd addresses those stall the LDS pipe -- SQ_LDS_UNALIGNED_STALL)
      typedef const volatile __attribute__((address_space(3))) unsigned char *lds_vu8;
      const lds_vu8 xp = (lds_vu8)eq + d * DG_ESTRIDE + (pr * 3) * 19 + pc * 3;
      float in[5][5];
#pragma unroll
      for (int a = 0; a < 5; a++)
#pragma unroll
        for (int b = 0; b < 5; b++) in[a][b] = (float)xp[a * 19 + b];  // the x 1/255 of n_categorize.cpp:99 is in cws
      digit_conv_pool(in, mw, cws, pos, pooled + d * DG_PSTRIDE);
    }
    __syncthreads();
    DG_STOP(3, pooled[0] + pooled[300])
    if (DMZ_DIGITS_WGS >= 5) digit_fc1_load(hidw + m * 32 * 320, wave, lane, fcb);
    digit_fc1(fcb, pooled, part, wave, lane);
    __syncthreads();
    DG_STOP(4, part[0] + part[600])
    digit_head(mw, part, hid, prob + m * 160, nd, tid);
  }
  // ---- vote (n_categorize.cpp:69-70), arg-max, usable gate (frame.cpp:63-64) ----
  float *fin = pooled;  // reuse: 160 floats
  if (tid < 160) {
    const int d = tid / 10, c = tid - d * 10;
    float v = 0.0f;
    if (d < nd) {
      const float r0 = prob[0 * 160 + tid], r1 = prob[1 * 160 + tid], r2 = prob[2 * 160 + tid];
      float mx = r0 > r1 ? r0 : r1;
      mx = mx > r2 ? mx : r2;
      v = (((r0 + r1) + r2) - mx) / 2.0f;
    }
    (void)c;
    fin[tid] = v;
    (&res->scores[0][0])[tid] = v;
  }
  __syncthreads();
  if (tid < 16) {
    int best = 0;
    for (int c = 1; c < 10; c++)
      if (fin[tid * 10 + c] > fin[tid * 10 + best]) best = c;
    res->digits[tid] = (uint8_t)best;
  }
  if (tid == 0) {
    float sum = fin[0];
    for (int i = 1; i < 160; i++) sum = sum + fin[i];  // sequential, Redux.h:168-184
    const float number_score = (float)nd - sum;
    res->number_score = number_score;
    if (number_score < 3.0f) res->flags = res->flags | DMZ_HIP_FLAG_USABLE;
  }
}

// Stand-alone digit model entry point (KAT): up to 16 float patches per workgroup through
// the same conv / matrix-core / head code.
__global__ __launch_bounds__(DG_THREADS) void k_digit_model(const float *__restrict__ wts,
                                                             const float *__restrict__ hidw, int model,
                                                             const float *__restrict__ xin, int n,
                                                             float *__restrict__ out) {
  __shared__ __attribute__((aligned(16))) float pooled[16 * DG_PSTRIDE];
  __shared__ float x[16 * 516];
  __shared__ float part[2 * 16 * 32];
  __shared__ float hid[16 * 32];
  __shared__ float prob[16 * 10];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int base = blockIdx.x * 16;
  const int rows = imin(16, n - base);
  if (rows <= 0) return;
  const float *mw = wts + dmzw::DIGIT0 + model * dmzw::DIGIT_STRIDE;
  for (int i = tid; i < rows * 513; i += DG_THREADS) {
    const int d = i / 513, p = i - d * 513;
    x[d * 516 + p] = xin[(size_t)(base + d) * 513 + p];
  }
  for (int i = tid; i < 16 * DG_PSTRIDE; i += DG_THREADS) pooled[i] = 0.0f;
  __syncthreads();
  for (int i = tid; i < rows * 40; i += DG_THREADS) {
    const int d = i / 40, pos = i - d * 40;
    const int pr = pos / 5, pc = pos - pr * 5;
    const float *xp = x + d * 516 + (pr * 3) * 19 + pc * 3;
    float in[5][5];
#pragma unroll
    for (int a = 0; a < 5; a++)
#pragma unroll
      for (int b = 0; b < 5; b++) in[a][b] = xp[a * 19 + b];
    digit_conv_pool(in, mw, mw + dmzw::D_CONV_W, pos, pooled + d * DG_PSTRIDE);
  }
  __syncthreads();
  Fc1B fcb;
  digit_fc1_load(hidw + model * 32 * 320, wave, lane, fcb);
  digit_fc1(fcb, pooled, part, wave, lane);
  __syncthreads();
  digit_head(mw, part, hid, prob, rows, tid);
  if (tid < rows * 10) out[(size_t)base * 10 + tid] = prob[tid];
}

}  // namespace

void dmz_launch_digits(hipStream_t s, const float *weights, const float *hidw, const uint8_t *cards,
                       size_t card_stride, int n, dmz_hip_frame_result *results) {
  hipLaunchKernelGGL(k_digits, dim3(n), dim3(DG_THREADS), DMZ_LDS_PAD, s, weights, hidw, cards, card_stride, n,
                     results);
}

void dmz_launch_digit_model(hipStream_t s, const float *weights, const float *hidw, int model,
                            const float *x, int n, float *out) {
  hipLaunchKernelGGL(k_digit_model, dim3((n + 15) / 16), dim3(DG_THREADS), 0, s, weights, hidw, model, x,
                     n, out);
}

int dmz_configure_scan(void) { return dmz_configure_vseg(); }
