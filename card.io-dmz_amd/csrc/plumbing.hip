// plumbing.hip -- camera-side plumbing around the scan path, batched and
// HBM-bound: every byte is read once and written once with 4- to 16-byte accesses per lane.
//   dmz_deinterleave_uint8_c2  dmz.cpp:49-56 -> llcv_split_u8 cv/convert.cpp:105-107 (cvSplit)
//   dmz_deinterleave_RGBA_to_R dmz.cpp:62-105
//   dmz_YCbCr_to_RGB           dmz.cpp:58-60 -> llcv_YCbCr2RGB_u8_c cv/convert.cpp:448-490
// Integer arithmetic only: bit-exact against oracle/orc_plumbing.c.
#include "dmz_hip_internal.h"

namespace {

// 8 interleaved bytes (4 pixel pairs) -> 4 bytes to each plane
__global__ __launch_bounds__(256) void k_split_c2(const uint8_t *__restrict__ src, size_t n_pairs,
                                                  uint8_t *__restrict__ c1, uint8_t *__restrict__ c2) {
  const size_t q = (size_t)blockIdx.x * 256 + threadIdx.x;  // group of 4 pairs
  const size_t first = q * 4;
  if (first >= n_pairs) return;
  if (first + 4 <= n_pairs) {
    const uint2 v = *(const uint2 *)(src + first * 2);
    // bytes: a0 b0 a1 b1 | a2 b2 a3 b3
    const uint32_t a = __builtin_amdgcn_perm(v.y, v.x, 0x06040200u), b = __builtin_amdgcn_perm(v.y, v.x, 0x07050301u);
    *(uint32_t *)(c1 + first) = a;
    *(uint32_t *)(c2 + first) = b;
  } else {
    for (size_t i = first; i < n_pairs; i++) {
      c1[i] = src[2 * i];
      c2[i] = src[2 * i + 1];
    }
  }
}

// 16 RGBA bytes -> 4 R bytes
__global__ __launch_bounds__(256) void k_rgba_to_r(const uint8_t *__restrict__ src, size_t n_px, uint8_t *__restrict__ dst) {
  const size_t q = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t first = q * 4;
  if (first >= n_px) return;  // n_px is a multiple of 4
  const uint4 v = *(const uint4 *)(src + first * 4);
  *(uint32_t *)(dst + first) = (v.x & 255u) | ((v.y & 255u) << 8) | ((v.z & 255u) << 16) | ((v.w & 255u) << 24);
}

__device__ __forceinline__ uint32_t sat8(int v) { return (uint32_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

// 4 pixels per lane: three dword loads, 12 (RGB) or 16 (RGBA) bytes stored
template <int CH>
__global__ __launch_bounds__(256) void k_ycbcr_to_rgb(const uint8_t *__restrict__ y, const uint8_t *__restrict__ cb,
                                                      const uint8_t *__restrict__ cr, size_t n_px,
                                                      uint8_t *__restrict__ rgb) {
  const size_t q = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t first = q * 4;
  if (first >= n_px) return;
  const int cnt = n_px - first >= 4 ? 4 : (int)(n_px - first);
  uint32_t wy = 0, wb = 0, wr = 0;
  if (cnt == 4) {
    wy = *(const uint32_t *)(y + first);
    wb = *(const uint32_t *)(cb + first);
    wr = *(const uint32_t *)(cr + first);
  } else {
    for (int k = 0; k < cnt; k++) {
      wy |= (uint32_t)y[first + k] << (8 * k);
      wb |= (uint32_t)cb[first + k] << (8 * k);
      wr |= (uint32_t)cr[first + k] << (8 * k);
    }
  }
  uint32_t R[4], G[4], B[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int py = (int)((wy >> (8 * k)) & 255u);
    const int sCb = (int)((wb >> (8 * k)) & 255u) - 128, sCr = (int)((wr >> (8 * k)) & 255u) - 128;  // int8 range
    B[k] = sat8(py + ((sCb * 29049 + (1 << 13)) >> 14));
    G[k] = sat8(py + ((sCb * -5636 + sCr * -11698 + (1 << 13)) >> 14));
    R[k] = sat8(py + ((sCr * 22987 + (1 << 13)) >> 14));
  }
  uint8_t *o = rgb + first * CH;
  if (cnt == 4) {
    if (CH == 3) {
      uint32_t *o32 = (uint32_t *)o;  // first * 3 is a multiple of 12
      o32[0] = R[0] | (G[0] << 8) | (B[0] << 16) | (R[1] << 24);
      o32[1] = G[1] | (B[1] << 8) | (R[2] << 16) | (G[2] << 24);
      o32[2] = B[2] | (R[3] << 8) | (G[3] << 16) | (B[3] << 24);
    } else {
      uint4 v;
      v.x = R[0] | (G[0] << 8) | (B[0] << 16) | 0xFF000000u;
      v.y = R[1] | (G[1] << 8) | (B[1] << 16) | 0xFF000000u;
      v.z = R[2] | (G[2] << 8) | (B[2] << 16) | 0xFF000000u;
      v.w = R[3] | (G[3] << 8) | (B[3] << 16) | 0xFF000000u;
      *(uint4 *)o = v;
    }
  } else {
    for (int k = 0; k < cnt; k++) {
      o[k * CH + 0] = (uint8_t)R[k];
      o[k * CH + 1] = (uint8_t)G[k];
      o[k * CH + 2] = (uint8_t)B[k];
      if (CH == 4) o[k * CH + 3] = 0xff;
    }
  }
}

}  // namespace

void dmz_launch_split_c2(hipStream_t s, const uint8_t *src, size_t n_pairs, uint8_t *c1, uint8_t *c2) {
  const size_t groups = (n_pairs + 3) / 4;
  hipLaunchKernelGGL(k_split_c2, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, s, src, n_pairs, c1, c2);
}

void dmz_launch_rgba_to_r(hipStream_t s, const uint8_t *src, size_t n_px, uint8_t *dst) {
  const size_t groups = n_px / 4;
  hipLaunchKernelGGL(k_rgba_to_r, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, s, src, n_px, dst);
}

void dmz_launch_ycbcr_to_rgb(hipStream_t s, const uint8_t *y, const uint8_t *cb, const uint8_t *cr, size_t n_px,
                             int channels, uint8_t *rgb) {
  const size_t groups = (n_px + 3) / 4;
  if (channels == 4)
    hipLaunchKernelGGL(k_ycbcr_to_rgb<4>, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, s, y, cb, cr, n_px, rgb);
  else
    hipLaunchKernelGGL(k_ycbcr_to_rgb<3>, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, s, y, cb, cr, n_px, rgb);
}

// ---------------------------------------------------------------------------------------------
// Quality scores: dmz_focus_score / dmz_brightness_score (dmz.cpp:114-199) on
// the scoring ROI of every frame of a batch.  focus = stddev of |sobel3_dx_dy| (sobel.cpp:556-607,
// stats.cpp:97-102 -> cv::meanStdDev), brightness = cvAvg.  The sums are exact integers; the
// final mean / variance / sqrt steps are the reference's double operations in the same order.
// ---------------------------------------------------------------------------------------------
namespace {

__global__ __launch_bounds__(256) void k_scores(const uint8_t *__restrict__ y, size_t frame_stride, int row_stride, int n,
                                                int rx, int ry, int rw, int rh, float *__restrict__ focus,
                                                float *__restrict__ brightness) {
  const int f = blockIdx.x, tid = threadIdx.x;
  if (f >= n) return;
  const uint8_t *roi = y + (size_t)f * frame_stride + (size_t)ry * row_stride + rx;
  unsigned s_abs = 0u, s_px = 0u;
  unsigned long long s_sq = 0ull;
  // d(r, c) = e(r-1, c) - e(r+1, c) with e(r, c) = p[r][c-1] - p[r][c+1] (indices clamped): wave w walks a
  // quarter of the rows, lanes run along the row; the e values of eight rows are fetched together (the
  // loads are the latency that matters here), each row is read once for its e and once for the mean
  const int lane = tid & 63, wave = tid >> 6;
  const int rows_per = (rh + 3) >> 2, rbeg = wave * rows_per, rend = rbeg + rows_per < rh ? rbeg + rows_per : rh;
  for (int c = lane; c < rw; c += 64) {
    const int cl = c == 0 ? 0 : c - 1, cr = c == rw - 1 ? rw - 1 : c + 1;
    // e of the row above the block (clamped) and of its first row
    int e_prev, e_cur;
    {
      const uint8_t *ra = roi + (size_t)(rbeg == 0 ? 0 : rbeg - 1) * row_stride, *rb = roi + (size_t)rbeg * row_stride;
      e_prev = (int)ra[cl] - (int)ra[cr];
      e_cur = (int)rb[cl] - (int)rb[cr];
    }
    for (int r0 = rbeg; r0 < rend; r0 += 8) {
      int e_next[8], px[8];
#pragma unroll
      for (int u = 0; u < 8; u++) {  // rows r0+u+1 (clamped) for e, r0+u for the mean
        const int rn = r0 + u + 1 < rh ? r0 + u + 1 : rh - 1, rc = r0 + u < rh ? r0 + u : rh - 1;
        const uint8_t *pn = roi + (size_t)rn * row_stride;
        e_next[u] = (int)pn[cl] - (int)pn[cr];
        px[u] = roi[(size_t)rc * row_stride + c];
      }
#pragma unroll
      for (int u = 0; u < 8; u++) {
        if (r0 + u < rend) {
          int d = e_prev - e_next[u];
          d = d < 0 ? -d : d;
          s_abs += (unsigned)d;
          s_sq += (unsigned long long)(d * d);
          s_px += (unsigned)px[u];
        }
        e_prev = e_cur;
        e_cur = e_next[u];
      }
    }
  }
  __shared__ unsigned sh_abs[256], sh_px[256];
  __shared__ unsigned long long sh_sq[256];
  sh_abs[tid] = s_abs;
  sh_px[tid] = s_px;
  sh_sq[tid] = s_sq;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) {
      sh_abs[tid] += sh_abs[tid + o];
      sh_px[tid] += sh_px[tid + o];
      sh_sq[tid] += sh_sq[tid + o];
    }
    __syncthreads();
  }
  if (tid == 0) {
    const double scale = 1. / ((double)rw * (double)rh);
    if (focus) {
      const double mean = (double)sh_abs[0] * scale;
      double var = (double)sh_sq[0] * scale - mean * mean;
      if (!(var > 0.)) var = 0.;
      focus[f] = (float)sqrt(var);
    }
    if (brightness) brightness[f] = (float)((double)sh_px[0] * scale);
  }
}

}  // namespace

void dmz_launch_scores(hipStream_t s, const uint8_t *y, size_t frame_stride, int row_stride, int n, int rx, int ry,
                       int rw, int rh, float *focus, float *brightness) {
  hipLaunchKernelGGL(k_scores, dim3((unsigned)n), dim3(256), 0, s, y, frame_stride, row_stride, n, rx, ry, rw, rh, focus,
                     brightness);
}

// ---------------------------------------------------------------------------------------------
// dmz_blur_card (dmz.cpp:499-515): cv::medianBlur(25) in place on the boxes of the leading digits
// of the result image -- a one-shot step per finished session, batched over cards.  One workgroup per
// card walks its boxes in digit order (a box sees what the previous boxes left); per box the ROI is
// copied to LDS (BORDER_REPLICATE at the ROI edge) and every output is the exact median of its
// 25 x 25 window, found by bisection on the value (8 counting passes).
// ---------------------------------------------------------------------------------------------
namespace {

constexpr int BLUR_MAX_W = 64, BLUR_MAX_H = 58, BLUR_K = 25;

__global__ __launch_bounds__(256) void k_blur_cards(uint8_t *__restrict__ rgb, size_t card_stride, int channels, int n,
                                                    const dmz_hip_session_result *__restrict__ sessions,
                                                    int unblur_digits) {
  const int card = blockIdx.x, tid = threadIdx.x;
  if (card >= n || unblur_digits < 0) return;
  __shared__ unsigned char roi[BLUR_MAX_W * BLUR_MAX_H * 4];
  uint8_t *img = rgb + (size_t)card * card_stride;
  const dmz_hip_session_result *ss = sessions + card;
  const int n_offsets = ss->n_offsets, blur_count = n_offsets - unblur_digits;
  const int stride = DMZ_CARD_WIDTH * channels;
  for (int i = 0; i < n_offsets && i < blur_count && i < 16; i++) {
    int x = (int)ss->offsets[i] - 1, y = ss->vseg_y_offset - 1;
    int w = (int)(ss->number_width + 2), h = 27 + 2;
    if (i < 4) h *= 2;
    int x1 = x + w, y1 = y + h;
    x = x < 0 ? 0 : x;
    y = y < 0 ? 0 : y;
    x1 = x1 > DMZ_CARD_WIDTH ? DMZ_CARD_WIDTH : x1;
    y1 = y1 > DMZ_CARD_HEIGHT ? DMZ_CARD_HEIGHT : y1;
    w = x1 - x;
    h = y1 - y;
    if (w <= 0 || h <= 0) continue;
    if (w > BLUR_MAX_W) w = BLUR_MAX_W;  // number_width beyond any real segmentation (validated by the host)
    const int rowb = w * channels;
    for (int k = tid; k < rowb * h; k += 256) {
      const int yy = k / rowb, b = k - yy * rowb;
      roi[k] = img[(size_t)(y + yy) * stride + (size_t)x * channels + b];
    }
    __syncthreads();
    const int r = BLUR_K / 2, t = BLUR_K * BLUR_K / 2;
    for (int k = tid; k < rowb * h; k += 256) {
      const int yy = k / rowb, b = k - yy * rowb, xx = b / channels, c = b - xx * channels;
      int lo = 0, hi = 255;  // smallest v with count(<= v) > t
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        int cnt = 0;
        for (int dy = -r; dy <= r; dy++) {
          int sy = yy + dy;
          sy = sy < 0 ? 0 : (sy > h - 1 ? h - 1 : sy);
          const unsigned char *row = roi + sy * rowb + c;
          for (int dx = -r; dx <= r; dx++) {
            int sx = xx + dx;
            sx = sx < 0 ? 0 : (sx > w - 1 ? w - 1 : sx);
            cnt += row[sx * channels] <= mid ? 1 : 0;
          }
        }
        if (cnt > t) hi = mid;
        else lo = mid + 1;
      }
      img[(size_t)(y + yy) * stride + (size_t)x * channels + b] = (unsigned char)lo;
    }
    __threadfence();
    __syncthreads();
  }
}

// llcv_scharr3_dx_abs (cv/sobel.cpp:706-804, the x86 scalar branch; dmz_scharr3_dx_abs of the Cython flavour, dmz.h:105):
// |right - left| with the column index clamped at the image edge, then 3 / 10 / 3 down the column with the row index clamped.
// One thread per output pixel; a diagnostic / compatibility entry (the scan path computes the same operator inside
// k_expiry_seg, on the rows it needs).
__global__ __launch_bounds__(256) void k_scharr3_dx_abs(const uint8_t *__restrict__ src, int src_stride, int w, int h,
                                                        int16_t *__restrict__ dst, int dst_stride) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= w * h) return;
  const int r = i / w, c = i - r * w;
  const int cl = c == 0 ? 0 : c - 1, cr = c == w - 1 ? w - 1 : c + 1;
  const int rt = r == 0 ? 0 : r - 1, rb = r == h - 1 ? h - 1 : r + 1;
  auto inter = [&](int rr) {
    const uint8_t *row = src + (size_t)rr * src_stride;
    const int d = (int)row[cr] - (int)row[cl];
    return d < 0 ? -d : d;
  };
  dst[(size_t)r * dst_stride + c] = (int16_t)(3 * (inter(rt) + inter(rb)) + 10 * inter(r));
}

}  // namespace

void dmz_launch_scharr3_dx_abs(hipStream_t s, const uint8_t *src, int src_stride, int w, int h, int16_t *dst, int dst_stride) {
  hipLaunchKernelGGL(k_scharr3_dx_abs, dim3((unsigned)((w * h + 255) / 256)), dim3(256), 0, s, src, src_stride, w, h, dst, dst_stride);
}

void dmz_launch_blur_cards(hipStream_t s, uint8_t *rgb, size_t card_stride, int channels, int n,
                           const dmz_hip_session_result *sessions, int unblur_digits) {
  hipLaunchKernelGGL(k_blur_cards, dim3((unsigned)n), dim3(256), 0, s, rgb, card_stride, channels, n, sessions,
                     unblur_digits);
}
