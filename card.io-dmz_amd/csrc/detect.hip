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
//     that contain a survivor with m > high.  Components are found with a
//     lock-free union-find in LDS (path halving + CAS hooking).
//   * Hough: votes are LDS integer atomics (order independent); the arg-max
//     reproduces the reference's r-outer / n-inner / strict-> scan by breaking
//     ties towards the smallest (r, n).
//
// CDNA4 mapping: the whole box lives in LDS (<= 11264 px): padded source tile,
// one buffer of packed (h-derivative, h-smooth) int16 pairs, one of packed
// (dx, dy) pairs; the union-find labels and later the Hough accumulator reuse
// the dead buffers (101.5 KB per workgroup, 16 waves).  Rows are fetched from
// HBM as aligned 32-bit words so that a wave reads contiguous row segments.
#include "dmz_hip_internal.h"

namespace {

constexpr int NT = kDetectThreads;

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

__device__ __forceinline__ int pack16(int lo, int hi) { return (lo & 0xffff) | (hi << 16); }
__device__ __forceinline__ int lo16(int v) { return (int)(short)(v & 0xffff); }
__device__ __forceinline__ int hi16(int v) { return v >> 16; }

// union-find on LDS labels (ECL-CC style): labels only ever decrease.
__device__ __forceinline__ int uf_rep(volatile int *lab, int v) {
  int cur = lab[v];
  if (cur != v) {
    int prev = v, next;
    while (cur > (next = lab[cur])) {
      lab[prev] = next;  // path halving
      prev = cur;
      cur = next;
    }
  }
  return cur;
}

__device__ __forceinline__ void uf_unite(int *lab, int a, int b) {
  int ra = uf_rep(lab, a), rb = uf_rep(lab, b);
  while (ra != rb) {
    if (ra < rb) { int t = ra; ra = rb; rb = t; }  // ra > rb: hook the larger root under the smaller
    int old = atomicCAS(&lab[ra], ra, rb);
    if (old == ra) break;
    ra = uf_rep(lab, old);
    rb = uf_rep(lab, rb);
  }
}

// map byte bits
constexpr int MAP_CAND = 1;    // survived NMS with m > low
constexpr int MAP_STRONG = 2;  // ... and m > high
constexpr int MAP_GATE = 4;    // gradient direction accepted by the Hough gate
constexpr int MAP_ROOT_STRONG = 8;

__global__ __launch_bounds__(NT) void k_detect_box(const uint8_t *__restrict__ planes,
                                                    size_t frame_stride, int row_stride,
                                                    DmzDetectParams params,
                                                    DmzBoxHit *__restrict__ hits,
                                                    const int *__restrict__ skip_mask) {
  // All LDS is carved from the dynamic region (keeps its base 16-byte aligned).
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  long long *s_red = (long long *)(lds + kDetectLdsBytes);                      // NT/64
  unsigned long long *s_best = (unsigned long long *)(lds + kDetectLdsBytes + 128);  // NT/64
  int *s_thr = (int *)(lds + kDetectLdsBytes + 256);                            // low, high

  const int box_id = blockIdx.x & 3;  // 1-D grid: gridDim.y is limited to 65535
  const int frame = blockIdx.x >> 2;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;

  if (skip_mask && skip_mask[frame * 4 + box_id]) return;

  const DmzBoxParams &bp = params.box[box_id];
  const int w = bp.w, h = bp.h, N = w * h;
  const uint32_t inv_w = bp.inv_w;

  unsigned char *srcp = lds;                           // padded source rows, later the map
  int *bufA = (int *)(lds + kDetectSrcBytes);          // (hderiv, hsmooth) -> labels
  int *bufB = bufA + kDetectMaxPixels;                 // (dx, dy) -> hough accumulator

  // ---- A. ROI -> LDS, aligned 32-bit words, then replicate 3 px left/right ----
  const int off = 4 + (bp.x & 3);                      // LDS column of ROI pixel 0
  const int sp = (off + w + 3 + 3) & ~3;               // LDS row stride (bytes)
  const int wpr = ((bp.x + w - 1) >> 2) - (bp.x >> 2) + 1;  // global words per row
  const uint8_t *plane = planes + (size_t)frame * frame_stride;
  for (int i = tid; i < wpr * h; i += NT) {
    int r = i / wpr, j = i - r * wpr;
    const uint8_t *g = plane + (size_t)(bp.y + r) * row_stride + ((bp.x >> 2) + j) * 4;
    uint32_t v;
    if ((((uintptr_t)g) & 3) == 0) {
      v = *(const uint32_t *)g;
    } else {  // plane base / row stride not 4-byte aligned: assemble from bytes
      v = (uint32_t)g[0] | ((uint32_t)g[1] << 8) | ((uint32_t)g[2] << 16) | ((uint32_t)g[3] << 24);
    }
    *(uint32_t *)(srcp + r * sp + 4 + j * 4) = v;
  }
  __syncthreads();
  for (int i = tid; i < h * 6; i += NT) {
    int r = i / 6, k = i - r * 6;
    unsigned char *row = srcp + r * sp;
    if (k < 3) row[off - 1 - k] = row[off];
    else row[off + w + (k - 3)] = row[off + w - 1];
  }
  __syncthreads();

  // ---- B. horizontal 7-tap pass: deriv {-1,-4,-5,0,5,4,1}, smooth {1,6,15,20,15,6,1} ----
  for (int p = tid; p < N; p += NT) {
    int r = __umulhi((uint32_t)p, inv_w);
    int c = p - r * w;
    if (c >= w) { c -= w; r++; }  // guard the (never hit for p*w < 2^32) rounding case
    const unsigned char *s = srcp + r * sp + off + c - 3;
    int a0 = s[0], a1 = s[1], a2 = s[2], a3 = s[3], a4 = s[4], a5 = s[5], a6 = s[6];
    int hd = (a6 - a0) + 4 * (a5 - a1) + 5 * (a4 - a2);
    int hs = (a0 + a6) + 6 * (a1 + a5) + 15 * (a2 + a4) + 20 * a3;
    bufA[p] = pack16(hd, hs);
  }
  __syncthreads();

  // ---- C. vertical pass -> dx, dy (saturated int16) + sum of saturated |.| ----
  long long local_sum = 0;
  for (int p = tid; p < N; p += NT) {
    int r = __umulhi((uint32_t)p, inv_w);
    int c = p - r * w;
    if (c >= w) { c -= w; r++; }
    int v0 = bufA[clampi(r - 3, 0, h - 1) * w + c];
    int v1 = bufA[clampi(r - 2, 0, h - 1) * w + c];
    int v2 = bufA[clampi(r - 1, 0, h - 1) * w + c];
    int v3 = bufA[p];
    int v4 = bufA[clampi(r + 1, 0, h - 1) * w + c];
    int v5 = bufA[clampi(r + 2, 0, h - 1) * w + c];
    int v6 = bufA[clampi(r + 3, 0, h - 1) * w + c];
    int dx = (lo16(v0) + lo16(v6)) + 6 * (lo16(v1) + lo16(v5)) + 15 * (lo16(v2) + lo16(v4)) + 20 * lo16(v3);
    int dy = (hi16(v6) - hi16(v0)) + 4 * (hi16(v5) - hi16(v1)) + 5 * (hi16(v4) - hi16(v2));
    dx = clampi(dx, -32768, 32767);
    dy = clampi(dy, -32768, 32767);
    bufB[p] = pack16(dx, dy);
    int ax = dx < 0 ? -dx : dx, ay = dy < 0 ? -dy : dy;  // cvAbs saturates 32768 -> 32767
    local_sum += (ax > 32767 ? 32767 : ax) + (ay > 32767 ? 32767 : ay);
  }
  for (int o = 32; o > 0; o >>= 1) local_sum += __shfl_down(local_sum, o, 64);
  if (lane == 0) s_red[wave] = local_sum;
  __syncthreads();
  if (tid == 0) {
    long long tot = 0;
    for (int i = 0; i < NT / 64; i++) tot += s_red[i];
    // canny.cpp:573-578: mean in double; low = cvFloor(mean), high = cvFloor(3.0f * mean)
    double mean = (double)tot / (double)N;
    double lowt = mean, hight = 3.0f * mean;
    s_thr[0] = (int)floor(lowt);
    s_thr[1] = (int)floor(hight);
  }
  __syncthreads();
  const int low = s_thr[0], high = s_thr[1];

  // ---- D. non-maximum suppression (canny.cpp:213-285) + Hough slope gate ----
  unsigned char *map = srcp;  // source tile is dead
  int *lab = bufA;            // (hderiv, hsmooth) is dead
  const int TG22 = 13573;     // (int)(0.4142135623730950488016887242097*(1<<15) + 0.5)
  for (int p = tid; p < N; p += NT) {
    int r = __umulhi((uint32_t)p, inv_w);
    int c = p - r * w;
    if (c >= w) { c -= w; r++; }
    const int v = bufB[p];
    const int dxv = lo16(v), dyv = hi16(v);
    const int ax = dxv < 0 ? -dxv : dxv, ay = dyv < 0 ? -dyv : dyv;
    const int m = ax + ay;
    int flags = 0;
    if (m > low) {
      const long long tg22x = (long long)ax * TG22;
      const long long tg67x = tg22x + ((long long)(ax + ax) << 15);
      const long long yy = (long long)ay << 15;
      int q1, q2, ge2;  // neighbour pixel indices (-1 = outside, magnitude 0)
      if (yy < tg22x) {          // sector 0: compare left / right
        q1 = c > 0 ? p - 1 : -1;
        q2 = c < w - 1 ? p + 1 : -1;
        ge2 = 1;
      } else if (yy > tg67x) {   // sector 2: compare up / down
        q1 = r > 0 ? p - w : -1;
        q2 = r < h - 1 ? p + w : -1;
        ge2 = 1;
      } else {                   // diagonal sectors
        const int s = ((dxv ^ dyv) < 0) ? -1 : 1;
        const int c1 = c - s, c2 = c + s;
        q1 = (r > 0 && c1 >= 0 && c1 < w) ? p - w - s : -1;
        q2 = (r < h - 1 && c2 >= 0 && c2 < w) ? p + w + s : -1;
        ge2 = 0;
      }
      int m1 = 0, m2 = 0;
      if (q1 >= 0) { int u = bufB[q1]; int a = lo16(u), b = hi16(u); m1 = (a < 0 ? -a : a) + (b < 0 ? -b : b); }
      if (q2 >= 0) { int u = bufB[q2]; int a = lo16(u), b = hi16(u); m2 = (a < 0 ? -a : a) + (b < 0 ? -b : b); }
      const bool is_max = (m > m1) && (ge2 ? (m >= m2) : (m > m2));
      if (is_max) flags = MAP_CAND | (m > high ? MAP_STRONG : 0);
    }
    // hough.cpp:133-150
    bool use;
    if (dxv != 0) {
      const float slope = (float)dyv / (float)dxv;
      use = bp.vertical ? (slope >= bp.slope_a && slope <= bp.slope_b)
                        : (slope >= bp.slope_a || slope <= bp.slope_b);
    } else {
      use = !bp.vertical;
    }
    if (use) flags |= MAP_GATE;
    map[p] = (unsigned char)flags;
    lab[p] = p;
  }
  __syncthreads();

  // ---- E. hysteresis: 8-connected components of candidates (union-find) ----
  for (int p = tid; p < N; p += NT) {
    if (!(map[p] & MAP_CAND)) continue;
    int r = __umulhi((uint32_t)p, inv_w);
    int c = p - r * w;
    if (c >= w) { c -= w; r++; }
    if (c > 0 && (map[p - 1] & MAP_CAND)) uf_unite(lab, p, p - 1);
    if (r > 0) {
      if (map[p - w] & MAP_CAND) uf_unite(lab, p, p - w);
      if (c > 0 && (map[p - w - 1] & MAP_CAND)) uf_unite(lab, p, p - w - 1);
      if (c < w - 1 && (map[p - w + 1] & MAP_CAND)) uf_unite(lab, p, p - w + 1);
    }
  }
  __syncthreads();
  for (int p = tid; p < N; p += NT) {
    if ((map[p] & (MAP_CAND | MAP_STRONG)) == (MAP_CAND | MAP_STRONG)) {
      int root = uf_rep(lab, p);
      // byte-wise OR through the containing 32-bit word
      atomicOr((unsigned int *)(map + (root & ~3)), (unsigned int)MAP_ROOT_STRONG << ((root & 3) * 8));
    }
  }
  // ---- F. Hough accumulator (hough.cpp:127-161) ----
  int *accum = bufB;  // (dx, dy) no longer needed: the gate bit is in the map
  const int numrho = bp.numrho;
  __syncthreads();
  for (int i = tid; i < kNumAngle * numrho; i += NT) accum[i] = 0;
  __syncthreads();
  const int half = (numrho - 1) / 2;
  for (int p = tid; p < N; p += NT) {
    const int f = map[p];
    if ((f & (MAP_CAND | MAP_GATE)) != (MAP_CAND | MAP_GATE)) continue;
    const int root = uf_rep(lab, p);
    if (!(((volatile unsigned char *)map)[root] & MAP_ROOT_STRONG)) continue;
    int r = __umulhi((uint32_t)p, inv_w);
    int c = p - r * w;
    if (c >= w) { c -= w; r++; }
#pragma unroll
    for (int n = 0; n < kNumAngle; n++) {
      int rr = ((c * bp.tab_cos[n] + r * bp.tab_sin[n]) >> 10) + half;
      atomicAdd(&accum[n * numrho + rr], 1);
    }
  }
  __syncthreads();

  // ---- G. arg-max with the reference's scan order (hough.cpp:163-176) ----
  unsigned long long best = 0;
  for (int i = tid; i < kNumAngle * numrho; i += NT) {
    const int n = i / numrho, rr = i - n * numrho;
    const unsigned int val = (unsigned int)accum[i];
    const unsigned int order = (unsigned int)(rr * kNumAngle + n);  // scan position
    const unsigned long long key = ((unsigned long long)val << 32) | (0xffffffffu - order);
    best = key > best ? key : best;
  }
  for (int o = 32; o > 0; o >>= 1) {
    unsigned long long other = __shfl_down(best, o, 64);
    best = other > best ? other : best;
  }
  if (lane == 0) s_best[wave] = best;
  __syncthreads();
  if (tid == 0) {
    for (int i = 1; i < NT / 64; i++) best = s_best[i] > best ? s_best[i] : best;
    const int max_val = (int)(best >> 32);
    const unsigned int order = 0xffffffffu - (unsigned int)(best & 0xffffffffu);
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
  }
}

}  // namespace

void dmz_launch_detect(hipStream_t s, const uint8_t *planes, size_t frame_stride, int row_stride,
                       int n, const DmzDetectParams &p, DmzBoxHit *hits, const int *skip_mask) {
  dim3 grid(4u * (unsigned)n);
  hipLaunchKernelGGL(k_detect_box, grid, dim3(NT), kDetectLdsBytes + 512, s, planes, frame_stride,
                     row_stride, p, hits, skip_mask);
}

int dmz_configure_detect(void) {
  return (int)hipFuncSetAttribute((const void *)k_detect_box,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, kDetectLdsBytes + 512);
}
