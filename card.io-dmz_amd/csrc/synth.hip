// synth.hip -- on-device generator of the synthetic benchmark / test inputs
// (640x480 luma frames, 428x270 card crops), so that tens of GB of frames can be
// made resident in HBM without crossing PCIe.  Scene and determinism contract:
// see oracle/orc_synth.c, whose CPU generator must produce identical bytes
// (tests/test_synth_parity.py).  Per-frame parameters are derived on the host
// (splitmix64, IEEE double homography); the kernels are IEEE double (+,*,/) for
// the frame->card mapping and pure integer after that.  -ffp-contract=off.
#include <math.h>
#include <string.h>

#include <vector>

#include "dmz_hip_internal.h"

namespace {

struct SynthParams {
  double h[9];
  int bg, card, ink, rim, noise_bg, noise_card, x0, y0, pitch;
  uint32_t key_frame, key_card;
  uint8_t digits[16];
  int ex0, ey0, epitch;
  uint8_t exp_digits[4];
  int kind;  // 0: 16 digits 4-4-4-4; 1: 15 digits 4-6-5, prefix 34 / 37
};

uint64_t splitmix64(uint64_t *s) {
  uint64_t z = (*s += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

__host__ __device__ inline uint32_t hash32(uint32_t a) {
  a ^= a >> 16; a *= 0x7feb352du;
  a ^= a >> 15; a *= 0x846ca68bu;
  a ^= a >> 16;
  return a;
}

void quad_homography(const double q[8], double hinv[9]) {
  const double x0 = q[0], y0 = q[1], x1 = q[2], y1 = q[3], x2 = q[4], y2 = q[5], x3 = q[6], y3 = q[7];
  const double dx1 = x1 - x3, dx2 = x2 - x3, dx3 = x0 - x1 + x3 - x2;
  const double dy1 = y1 - y3, dy2 = y2 - y3, dy3 = y0 - y1 + y3 - y2;
  const double den = dx1 * dy2 - dx2 * dy1;
  const double g = (dx3 * dy2 - dx2 * dy3) / den;
  const double hh = (dx1 * dy3 - dx3 * dy1) / den;
  const double su = 1.0 / 427.0, sv = 1.0 / 269.0;
  double m[9];
  m[0] = (x1 - x0 + g * x1) * su; m[1] = (x2 - x0 + hh * x2) * sv; m[2] = x0;
  m[3] = (y1 - y0 + g * y1) * su; m[4] = (y2 - y0 + hh * y2) * sv; m[5] = y0;
  m[6] = g * su;                  m[7] = hh * sv;                  m[8] = 1.0;
  hinv[0] = m[4] * m[8] - m[5] * m[7];
  hinv[1] = m[2] * m[7] - m[1] * m[8];
  hinv[2] = m[1] * m[5] - m[2] * m[4];
  hinv[3] = m[5] * m[6] - m[3] * m[8];
  hinv[4] = m[0] * m[8] - m[2] * m[6];
  hinv[5] = m[2] * m[3] - m[0] * m[5];
  hinv[6] = m[3] * m[7] - m[4] * m[6];
  hinv[7] = m[1] * m[6] - m[0] * m[7];
  hinv[8] = m[0] * m[4] - m[1] * m[3];
}

void make_params(uint64_t seed, uint64_t frame, SynthParams *p) {
  uint64_t s0 = seed ^ 0xCA4D10ull, s1 = frame;
  uint64_t s = splitmix64(&s0) ^ (splitmix64(&s1) * 0xD1342543DE82EF95ull);
  double q[8];
  static const int base[8] = {106, 105, 533, 105, 106, 374, 533, 374};
  for (int i = 0; i < 8; i++) {
    int j = (int)(splitmix64(&s) % 193) - 96;
    q[i] = (double)base[i] + (double)j * 0.0625;
  }
  quad_homography(q, p->h);
  p->bg = 56 + (int)(splitmix64(&s) % 17);
  p->card = 168 + (int)(splitmix64(&s) % 17);
  p->ink = -96;
  p->rim = 40;
  p->noise_bg = 6;
  p->noise_card = 3;
  p->x0 = 40 * 16 + (int)(splitmix64(&s) % 65) - 32;
  p->y0 = 151 * 16 + (int)(splitmix64(&s) % 129) - 64;
  p->pitch = 293;
  int sum = 0;
  p->digits[0] = 4;
  for (int i = 1; i < 15; i++) p->digits[i] = (uint8_t)(splitmix64(&s) % 10);
  for (int i = 0; i < 15; i++) {
    int d = p->digits[i];
    if ((i & 1) == 0) { d *= 2; d = d % 10 + d / 10; }
    sum += d;
  }
  p->digits[15] = (uint8_t)((10 - sum % 10) % 10);
  p->key_frame = hash32((uint32_t)(seed * 0x9E3779B1u) ^ hash32((uint32_t)frame) ^ (uint32_t)(frame >> 32));
  p->key_card = p->key_frame ^ 0x5bd1e995u;
  // expiry line "MM/YY" (orc_synth.c make_params: same draws in the same order)
  p->ex0 = 190 * 16 + (int)(splitmix64(&s) % 257) - 128;
  p->ey0 = 206 * 16 + (int)(splitmix64(&s) % 129) - 64;
  p->epitch = 13 * 16 + (int)(splitmix64(&s) % 17) - 8;
  const int month = 1 + (int)(splitmix64(&s) % 12), year = 27 + (int)(splitmix64(&s) % 4);
  p->exp_digits[0] = (uint8_t)(month / 10);
  p->exp_digits[1] = (uint8_t)(month % 10);
  p->exp_digits[2] = (uint8_t)(year / 10);
  p->exp_digits[3] = (uint8_t)(year % 10);
  // the card kind: the last draw (orc_synth.c)
  p->kind = (splitmix64(&s) % 10) == 0;
  if (p->kind) {
    p->digits[0] = 3;
    p->digits[1] = (p->digits[1] & 1) ? 7 : 4;
    sum = 0;
    for (int i = 0; i < 14; i++) {
      int d = p->digits[i];
      if (i & 1) { d *= 2; d = d % 10 + d / 10; }
      sum += d;
    }
    p->digits[14] = (uint8_t)((10 - sum % 10) % 10);
    p->digits[15] = 0;
  }
}

__constant__ short c_seg[7][4] = {
    {2 * 16, 0 * 16, 15 * 16, 3 * 16},   {14 * 16, 1 * 16, 17 * 16, 13 * 16},
    {14 * 16, 12 * 16, 17 * 16, 24 * 16}, {2 * 16, 22 * 16, 15 * 16, 25 * 16},
    {0 * 16, 12 * 16, 3 * 16, 24 * 16},  {0 * 16, 1 * 16, 3 * 16, 13 * 16},
    {2 * 16, 11 * 16, 15 * 16, 14 * 16},
};
__constant__ short c_seg_small[7][4] = {
    {1 * 16, 0 * 16, 8 * 16, 2 * 16},   {7 * 16, 1 * 16, 9 * 16, 8 * 16},  {7 * 16, 7 * 16, 9 * 16, 14 * 16},
    {1 * 16, 13 * 16, 8 * 16, 15 * 16}, {0 * 16, 7 * 16, 2 * 16, 14 * 16}, {0 * 16, 1 * 16, 2 * 16, 8 * 16},
    {1 * 16, 104, 8 * 16, 136},
};
__constant__ unsigned char c_digit_segs[10] = {0x3F, 0x06, 0x5B, 0x4F, 0x66, 0x6D, 0x7D, 0x07, 0x7F, 0x6F};

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

__device__ __forceinline__ int noise_at(uint32_t key, int x, int y) {
  const uint32_t h = hash32(key ^ hash32((uint32_t)(y * 1024 + x) + 0x9e3779b9u));
  return (int)(h & 255) + (int)((h >> 8) & 255) + (int)((h >> 16) & 255) + (int)(h >> 24) - 510;
}

template <bool SMALL>
__device__ int stroke_cov_tab(int segs, int px, int py) {
  int best = 0;
  for (int s = 0; s < 7; s++) {
    if (!(segs & (1 << s))) continue;
    const short *sg = SMALL ? c_seg_small[s] : c_seg[s];
    const int ax = px - sg[0], bx = sg[2] - px;
    const int ay = py - sg[1], by = sg[3] - py;
    const int cx = clampi((ax < bx ? ax : bx) + 8, 0, 16);
    const int cy = clampi((ay < by ? ay : by) + 8, 0, 16);
    const int c = cx * cy;
    if (c > best) best = c;
  }
  return best;
}
__device__ int stroke_cov(int segs, int px, int py) { return stroke_cov_tab<false>(segs, px, py); }

__device__ int slash_cov(int px, int py) {
  const int xl = 24 + ((240 - py) * 96) / 240;
  const int dx = px - xl < 0 ? xl - px : px - xl;
  const int cx = clampi(16 - dx + 8, 0, 16);
  const int cy = clampi((py < 240 - py ? py : 240 - py) + 8, 0, 16);
  return cx * cy;
}

__device__ int card_delta(const SynthParams &p, int U, int V) {
  int d = 0;
  const int t = (U >> 4) & 127;
  d += ((t < 64 ? t : 127 - t) - 32) >> 3;
  const int ry = V - p.y0;
  if (ry >= -32 && ry < 25 * 16 + 32) {
    const int rx = U - p.x0;
    if (rx >= -32) {
      const int slot = (rx + 32) / p.pitch;
      if (p.kind ? (slot < 17 && slot != 4 && slot != 11) : (slot < 19 && (slot % 5) != 4)) {
        const int di = p.kind ? slot - (slot > 4) - (slot > 11) : slot - slot / 5;
        const int lx = rx - slot * p.pitch;
        const int segs = c_digit_segs[p.digits[di]];
        const int c0 = stroke_cov(segs, lx, ry);
        const int c1 = stroke_cov(segs, lx + 16, ry + 16);
        const int c2 = stroke_cov(segs, lx - 16, ry - 16);
        d += (p.ink * c0) >> 8;
        d += (p.rim * (c1 - c0 > 0 ? c1 - c0 : 0)) >> 8;
        d -= (p.rim * (c2 - c0 > 0 ? c2 - c0 : 0)) >> 8;
      }
    }
  }
  const int ey = V - p.ey0;
  if (ey >= -32 && ey < 15 * 16 + 32) {
    const int ex = U - p.ex0;
    if (ex >= -32) {
      const int slot = (ex + 32) / p.epitch;
      if (slot < 5) {
        const int lx = ex - slot * p.epitch;
        int c0, c1, c2;
        if (slot == 2) {
          c0 = slash_cov(lx, ey);
          c1 = slash_cov(lx + 16, ey + 16);
          c2 = slash_cov(lx - 16, ey - 16);
        } else {
          const int segs = c_digit_segs[p.exp_digits[slot > 2 ? slot - 1 : slot]];
          c0 = stroke_cov_tab<true>(segs, lx, ey);
          c1 = stroke_cov_tab<true>(segs, lx + 16, ey + 16);
          c2 = stroke_cov_tab<true>(segs, lx - 16, ey - 16);
        }
        d += (p.ink * c0) >> 8;
        d += (p.rim * (c1 - c0 > 0 ? c1 - c0 : 0)) >> 8;
        d -= (p.rim * (c2 - c0 > 0 ? c2 - c0 : 0)) >> 8;
      }
    }
  }
  return d;
}

// one thread = 4 horizontally adjacent pixels, 32-bit store
__global__ __launch_bounds__(256) void k_synth_frames(const SynthParams *__restrict__ params, int n,
                                                       uint8_t *__restrict__ y) {
  const int f = blockIdx.x / 300;  // 300 blocks per frame (1-D grid)
  const int q = (blockIdx.x - f * 300) * 256 + threadIdx.x;  // 4-pixel group index, 160 per row
  if (f >= n || q >= 160 * 480) return;
  __shared__ SynthParams sp;
  if (threadIdx.x == 0) sp = params[f];
  __syncthreads();
  const int yy = q / 160, xx0 = (q - yy * 160) * 4;
  uint32_t packed = 0;
  for (int k = 0; k < 4; k++) {
    const int xx = xx0 + k;
    const double fx = (double)xx, fy = (double)yy;
    const double w = (sp.h[6] * fx + sp.h[7] * fy) + sp.h[8];
    const double u = ((sp.h[0] * fx + sp.h[1] * fy) + sp.h[2]) / w;
    const double v = ((sp.h[3] * fx + sp.h[4] * fy) + sp.h[5]) / w;
    const int nz = noise_at(sp.key_frame, xx, yy);
    int val;
    const int U = (int)floor(u * 16.0), V = (int)floor(v * 16.0);
    int cu = clampi((U < 427 * 16 - U ? U : 427 * 16 - U) + 8, 0, 16);
    const int cv = clampi((V < 269 * 16 - V ? V : 269 * 16 - V) + 8, 0, 16);
    if (u < -4.0 || u > 431.0 || v < -4.0 || v > 273.0) cu = 0;
    const int cov = cu * cv;
    const int bgv = sp.bg + ((nz * sp.noise_bg) >> 8);
    if (cov == 0) {
      val = bgv;
    } else {
      const int cardv = sp.card + card_delta(sp, U, V) + ((nz * sp.noise_card) >> 8);
      val = (bgv * (256 - cov) + cardv * cov + 128) >> 8;
    }
    packed |= (uint32_t)clampi(val, 0, 255) << (8 * k);
  }
  *(uint32_t *)(y + (size_t)f * (640 * 480) + (size_t)yy * 640 + xx0) = packed;
}

__global__ __launch_bounds__(256) void k_synth_cards(const SynthParams *__restrict__ params, int n,
                                                      uint8_t *__restrict__ cards) {
  const int f = blockIdx.x / 113;  // 113 blocks per card (1-D grid)
  const int q = (blockIdx.x - f * 113) * 256 + threadIdx.x;  // 4-pixel group, 107 per row
  if (f >= n || q >= 107 * 270) return;
  __shared__ SynthParams sp;
  if (threadIdx.x == 0) sp = params[f];
  __syncthreads();
  const int v = q / 107, u0 = (q - v * 107) * 4;
  uint32_t packed = 0;
  for (int k = 0; k < 4; k++) {
    const int u = u0 + k;
    const int nz = noise_at(sp.key_card, u, v);
    const int val = sp.card + card_delta(sp, u * 16, v * 16) + ((nz * sp.noise_card) >> 8);
    packed |= (uint32_t)clampi(val, 0, 255) << (8 * k);
  }
  *(uint32_t *)(cards + (size_t)f * (428 * 270) + (size_t)v * 428 + u0) = packed;
}

// test utility (include/dmz_hip_test.h): every CU's LDS is overwritten with `word` (a kernel that reads LDS it did not write
// sees it afterwards).  Dynamic LDS: the launcher sizes a workgroup's share from the device attributes so that the workgroups
// resident on a CU together own ALL of its LDS (gfx950: one workgroup with the whole 160 KiB).
__global__ void k_fill_lds(uint32_t word, int words, uint32_t *__restrict__ sink) {
  extern __shared__ uint32_t fill_lds[];
  for (int i = threadIdx.x; i < words; i += blockDim.x) fill_lds[i] = word;
  __syncthreads();
  // (the stores must not be eliminated; with more than one workgroup per CU the spin keeps them resident together)
  uint32_t acc = 0;
  for (int r = 0; r < 64; r++) acc += fill_lds[(threadIdx.x * 17u + r * 1031u) % (unsigned)words];
  if (acc == 0x12345u && sink) sink[0] = acc;
}

}  // namespace

// Host side: derive the per-frame parameters, upload, launch.  `scratch` must hold
// n * sizeof(SynthParams) bytes of device memory.
size_t dmz_synth_params_bytes(int n) { return (size_t)n * sizeof(SynthParams); }

int dmz_synth_upload_params(hipStream_t s, uint64_t seed, uint64_t first, int n, void *scratch) {
  std::vector<SynthParams> host((size_t)n);
  for (int i = 0; i < n; i++) make_params(seed, first + (uint64_t)i, &host[(size_t)i]);
  hipError_t e = hipMemcpyAsync(scratch, host.data(), host.size() * sizeof(SynthParams),
                                hipMemcpyHostToDevice, s);
  if (e != hipSuccess) return (int)e;
  return (int)hipStreamSynchronize(s);  // `host` goes out of scope
}

void dmz_launch_synth_frames(hipStream_t s, const void *params, int n, uint8_t *y) {
  hipLaunchKernelGGL(k_synth_frames, dim3(300u * (unsigned)n), dim3(256), 0, s,
                     (const SynthParams *)params, n, y);
}

void dmz_launch_synth_cards(hipStream_t s, const void *params, int n, uint8_t *cards) {
  hipLaunchKernelGGL(k_synth_cards, dim3(113u * (unsigned)n), dim3(256), 0, s,
                     (const SynthParams *)params, n, cards);
}

int dmz_launch_fill_lds(hipStream_t s, int device, uint32_t word) {
  int per_cu = 0, per_wg = 0, cus = 0;
  if (hipDeviceGetAttribute(&per_cu, hipDeviceAttributeMaxSharedMemoryPerMultiprocessor, device) != hipSuccess ||
      hipDeviceGetAttribute(&per_wg, hipDeviceAttributeMaxSharedMemoryPerBlock, device) != hipSuccess ||
      hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || per_cu <= 0 || per_wg <= 0)
    return 1;
  const int nper = (per_cu + per_wg - 1) / per_wg;        // workgroups that share a CU's LDS (1 on gfx950)
  const int bytes = (per_cu / nper) & ~3;
  // (1024 threads = 16 waves: two such workgroups could meet on a CU by wave slots, never by LDS when nper == 1)
  const int threads = nper == 1 ? 1024 : 256;
  if (hipFuncSetAttribute((const void *)k_fill_lds, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return 1;
  hipLaunchKernelGGL(k_fill_lds, dim3((unsigned)(cus * nper * 8)), dim3(threads), bytes, s, word, bytes / 4, (uint32_t *)nullptr);
  return 0;
}
