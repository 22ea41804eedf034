// geometry.hip -- per-frame geometry (edges -> corners -> homography); the warp itself
// (perspective rectification of the card to 428 x 270) is in warp.hip.
//
// Replaces, for a whole batch:
//   find_line_in_detection_rects' origin shift (dmz.cpp:364-366, geometry.cpp:34-43),
//   parametricIntersect (geometry.cpp:14-32, Eigen 2x2 inverse Inverse.h:70-89),
//   dmz_transform_card's corner ordering (dmz.cpp:443-471),
//   llcv_calc_persp_transform (warp.cpp:34-125; Eigen 3.2.4 HouseholderQR<8x8 float>
//   in scalar evaluation order: HouseholderQR.h:219-250,306-334, Householder.h:65-130),
//   cvWarpPerspective(INTER_LINEAR | FILL_OUTLIERS, 0) (warp.cpp:165; OpenCV 2.4
//   semantics).
//
// Bit-exactness: this file is compiled with -ffp-contract=off; every float /
// double expression is a sequence of single IEEE operations in the reference's
// order (fp32 sqrt and division are correctly rounded by default under hipcc).
// No transcendental is evaluated on the device: cosf/sinf of the ten possible
// line angles and the origin-shift terms arrive as host-computed tables.
#include <float.h>

#include "dmz_hip_internal.h"

namespace {

#define QA(r, c) a[(c) * 8 + (r)]

// x = A.householderQr().solve(b), column-major 8x8 float, scalar Eigen order.
// Every loop is unrolled (all bounds are compile-time once k is): the matrix then lives in registers.  The rolled form
// indexed a[] dynamically and kept it in scratch memory (272 B per lane) -- slow, and the one kernel of the library whose
// result depended on private-segment memory: with other queues' kernels in flight beside it, one 64-byte scratch line (one
// matrix element of 16 consecutive frames) came back wrong about once in ten 65 536-frame runs (round 5, tools/dev/determinism.py).
// SSE = the summation order of a stock x86-64 build of the reference (Eigen's SSE2 packet paths; DMZ_HIP_OPT_EIGEN_SSE2): the
// three reductions -- tail.squaredNorm(), every coefficient of essential^T * bottom, and the inner products of Q^T b -- are
// linear vectorised sums whose packets start at element 0: the first four terms as (x0 + x2) + (x1 + x3), the rest one by
// one; fewer than four terms sequentially (oracle/orc_cv.c gives the Eigen source lines; pinned on the reference's build).
template <bool SSE, int N>
__device__ __forceinline__ float eigen_redux(const float (&x)[7]) {
  float r;
  if (SSE && N >= 4) {
    r = (x[0] + x[2]) + (x[1] + x[3]);
#pragma unroll
    for (int i = 4; i < N; i++) r = r + x[i];
  } else {
    r = x[0];
#pragma unroll
    for (int i = 1; i < N; i++) r = r + x[i];
  }
  return r;
}
// (N is a compile-time constant once k is: dispatch on the unrolled loop's counter)
template <bool SSE>
__device__ __forceinline__ float eigen_redux_n(const float (&x)[7], int n) {
  switch (n) {
    case 1: return eigen_redux<SSE, 1>(x);
    case 2: return eigen_redux<SSE, 2>(x);
    case 3: return eigen_redux<SSE, 3>(x);
    case 4: return eigen_redux<SSE, 4>(x);
    case 5: return eigen_redux<SSE, 5>(x);
    case 6: return eigen_redux<SSE, 6>(x);
    default: return eigen_redux<SSE, 7>(x);
  }
}

// (developer probe, -DDMZ_DEV_HTRACE: `tr` receives a hash of the matrix state after each of the eight Householder steps, of
// Q^T b and of the solution -- which step two evaluations first disagree at; nullptr in every product build)
__device__ __forceinline__ unsigned htrace_hash(const float *v, int n) {
  unsigned h = 0x9E3779B9u;
#pragma unroll
  for (int i = 0; i < n; i++) h = ((h << 5) | (h >> 27)) ^ __float_as_uint(v[i]);
  return h;
}
template <bool SSE>
__device__ __forceinline__ void householder_qr_solve8(float *a, float *b, unsigned *tr = nullptr) {
  float hcoef[8];
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const int rem = 8 - k;
    float tail_sq = 0.0f;
    if (rem > 1) {
      float sq[7];
#pragma unroll
      for (int i = 1; i < rem; i++) sq[i - 1] = QA(k + i, k) * QA(k + i, k);
      tail_sq = eigen_redux_n<SSE>(sq, rem - 1);
    }
    const float c0 = QA(k, k);
    float tau, beta;
    if (rem == 1 || tail_sq == 0.0f) {
      tau = 0.0f;
      beta = c0;
#pragma unroll
      for (int i = 1; i < rem; i++) QA(k + i, k) = 0.0f;
    } else {
      beta = sqrtf(c0 * c0 + tail_sq);
      if (c0 >= 0.0f) beta = -beta;
      const float denom = c0 - beta;
#pragma unroll
      for (int i = 1; i < rem; i++) QA(k + i, k) = QA(k + i, k) / denom;
      tau = (beta - c0) / beta;
    }
    hcoef[k] = tau;
    QA(k, k) = beta;
    const int rcols = 8 - k - 1;
    if (rcols > 0 && rem > 1) {
#pragma unroll
      for (int c = 0; c < rcols; c++) {
        const int col = k + 1 + c;
        float pr[7];
#pragma unroll
        for (int i = 1; i < rem; i++) pr[i - 1] = QA(k + i, k) * QA(k + i, col);
        float tmp = eigen_redux_n<SSE>(pr, rem - 1);
        tmp = tmp + QA(k, col);
        QA(k, col) = QA(k, col) - tau * tmp;
#pragma unroll
        for (int i = 1; i < rem; i++) QA(k + i, col) = QA(k + i, col) - (tau * QA(k + i, k)) * tmp;
      }
    }
    if (tr) tr[k] = __float_as_uint(beta) ^ ((__float_as_uint(tau) << 13) | (__float_as_uint(tau) >> 19));  // (pivot and coefficient of the step)
  }
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const int rem = 8 - k;
    const float tau = hcoef[k];
    if (rem == 1) {
      b[k] = b[k] * (1.0f - tau);
    } else {
      float pr[7];
#pragma unroll
      for (int i = 1; i < rem; i++) pr[i - 1] = QA(k + i, k) * b[k + i];
      float tmp = eigen_redux_n<SSE>(pr, rem - 1);
      tmp = tmp + b[k];
      b[k] = b[k] - tau * tmp;
#pragma unroll
      for (int i = 1; i < rem; i++) b[k + i] = b[k + i] - (tau * QA(k + i, k)) * tmp;
    }
  }
  if (tr) tr[8] = htrace_hash(b, 8);
#pragma unroll
  for (int i = 7; i >= 0; i--) {
    b[i] = b[i] / QA(i, i);
#pragma unroll
    for (int r = 0; r < i; r++) b[r] = b[r] - b[i] * QA(r, i);
  }
  if (tr) tr[9] = htrace_hash(b, 8);
}

template <bool SSE>
__device__ __forceinline__ void calc_persp_transform(const float *sp, const float *dp, float *m, unsigned *tr = nullptr) {
  float a[64], b[8];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const float sx = sp[2 * i], sy = sp[2 * i + 1], dx = dp[2 * i], dy = dp[2 * i + 1];
    QA(i, 0) = sx; QA(i, 1) = sy; QA(i, 2) = 1; QA(i, 3) = 0; QA(i, 4) = 0; QA(i, 5) = 0;
    QA(i, 6) = -sx * dx; QA(i, 7) = -sy * dx;
    QA(i + 4, 0) = 0; QA(i + 4, 1) = 0; QA(i + 4, 2) = 0;
    QA(i + 4, 3) = sx; QA(i + 4, 4) = sy; QA(i + 4, 5) = 1;
    QA(i + 4, 6) = -sx * dy; QA(i + 4, 7) = -sy * dy;
    b[i] = dx;
    b[i + 4] = dy;
  }
  householder_qr_solve8<SSE>(a, b, tr);
  m[0] = b[0]; m[1] = b[1]; m[2] = b[2];
  m[3] = b[3]; m[4] = b[4]; m[5] = b[5];
  m[6] = b[6]; m[7] = b[7]; m[8] = 1.0f;
}

// cv::invert of the 3x3 (float -> double) matrix, as cvWarpPerspective does when
// CV_WARP_INVERSE_MAP is absent.
__device__ void invert3x3(const float *mf, DmzWarpMat *out) {
  double s[9];
  for (int i = 0; i < 9; i++) s[i] = (double)mf[i];
  const double det = s[0] * (s[4] * s[8] - s[5] * s[7]) - s[1] * (s[3] * s[8] - s[5] * s[6]) +
                     s[2] * (s[3] * s[7] - s[4] * s[6]);
  if (det != 0.) {
    const double d = 1. / det;
    out->m[0] = (s[4] * s[8] - s[5] * s[7]) * d;
    out->m[1] = (s[2] * s[7] - s[1] * s[8]) * d;
    out->m[2] = (s[1] * s[5] - s[2] * s[4]) * d;
    out->m[3] = (s[5] * s[6] - s[3] * s[8]) * d;
    out->m[4] = (s[0] * s[8] - s[2] * s[6]) * d;
    out->m[5] = (s[2] * s[3] - s[0] * s[5]) * d;
    out->m[6] = (s[3] * s[7] - s[4] * s[6]) * d;
    out->m[7] = (s[1] * s[6] - s[0] * s[7]) * d;
    out->m[8] = (s[0] * s[4] - s[1] * s[3]) * d;
  } else {
    for (int i = 0; i < 9; i++) out->m[i] = 0.;
  }
}

// geometry.cpp:14-32
__device__ bool parametric_intersect(float rho1, float c1, float s1, float rho2, float c2, float s2,
                                     float *x, float *y) {
  const float det = c1 * s2 - c2 * s1;
  if ((double)det < 1e-10) return false;
  const float invdet = 1.0f / det;
  const float i00 = s2 * invdet, i10 = -c2 * invdet, i01 = -s1 * invdet, i11 = c1 * invdet;
  *x = i00 * rho1 + i01 * rho2;
  *y = i10 * rho1 + i11 * rho2;
  return true;
}

#ifdef DMZ_DEV_SELFCHECK
__device__ unsigned long long g_dev_selfcheck_geom[2];  // frames with four edges evaluated twice, frames whose corners differed
#endif
__global__ __launch_bounds__(64) void k_geometry(int n, const DmzDetectParams *__restrict__ params,
                           const DmzBoxHit *__restrict__ hits, int nplanes,
                           dmz_hip_frame_result *__restrict__ results) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= n) return;
  dmz_hip_frame_result *res = results + f;
  float rho[4], ct[4], st[4];
  int found[4];
  for (int e = 0; e < 4; e++) {
    found[e] = 0;
    rho[e] = FLT_MAX;  // ParametricLineNone(), geometry.h:24-29
    ct[e] = 0.0f;
    st[e] = 0.0f;
    float theta = FLT_MAX;
    for (int pl = 0; pl < nplanes && !found[e]; pl++) {
      const DmzBoxHit hit = hits[((size_t)pl * n + f) * 4 + e];
      if (!hit.found) continue;
      const DmzBoxParams &bp = params[pl].box[e];
      // hough.cpp:190-191, then geometry.cpp:41, then dmz.cpp:365
      const float rho_local = ((float)hit.r - (float)(bp.numrho - 1) * 0.5f) * 1.0f;
      float shifted = (float)((double)rho_local + bp.delta_rho[hit.n]);
      shifted = shifted * bp.rho_multiplier;
      rho[e] = shifted;
      theta = bp.theta_n[hit.n];
      ct[e] = bp.cos_t[hit.n];
      st[e] = bp.sin_t[hit.n];
      found[e] = 1;
    }
    res->found[e] = found[e];
    res->rho[e] = rho[e];
    res->theta[e] = theta;
  }
  bool all = found[0] && found[1] && found[2] && found[3];
  float cx[4] = {0, 0, 0, 0}, cy[4] = {0, 0, 0, 0};
  if (all) {
    // edges: 0 top, 1 left, 2 bottom, 3 right; corners: tl, bl, tr, br (dmz.cpp:420-423)
    const bool a = parametric_intersect(rho[0], ct[0], st[0], rho[1], ct[1], st[1], &cx[0], &cy[0]);
    const bool b = parametric_intersect(rho[2], ct[2], st[2], rho[1], ct[1], st[1], &cx[1], &cy[1]);
    const bool c = parametric_intersect(rho[0], ct[0], st[0], rho[3], ct[3], st[3], &cx[2], &cy[2]);
    const bool d = parametric_intersect(rho[2], ct[2], st[2], rho[3], ct[3], st[3], &cx[3], &cy[3]);
    all = a && b && c && d;
  }
#ifdef DMZ_DEV_SELFCHECK  /* developer probe (tools/dev/homography_fault.sh selfcheck): the intersections computed twice and compared */
  if (found[0] && found[1] && found[2] && found[3]) {
    float r2[4], c2[4], s2[4], x2[4] = {0, 0, 0, 0}, y2[4] = {0, 0, 0, 0};
    for (int e = 0; e < 4; e++) {
      r2[e] = rho[e], c2[e] = ct[e], s2[e] = st[e];
      asm volatile("" : "+v"(r2[e]), "+v"(c2[e]), "+v"(s2[e]));
    }
    (void)parametric_intersect(r2[0], c2[0], s2[0], r2[1], c2[1], s2[1], &x2[0], &y2[0]);
    (void)parametric_intersect(r2[2], c2[2], s2[2], r2[1], c2[1], s2[1], &x2[1], &y2[1]);
    (void)parametric_intersect(r2[0], c2[0], s2[0], r2[3], c2[3], s2[3], &x2[2], &y2[2]);
    (void)parametric_intersect(r2[2], c2[2], s2[2], r2[3], c2[3], s2[3], &x2[3], &y2[3]);
    bool same = true;
    for (int i = 0; i < 4; i++)
      same = same && __float_as_uint(x2[i]) == __float_as_uint(cx[i]) && __float_as_uint(y2[i]) == __float_as_uint(cy[i]);
    atomicAdd(&g_dev_selfcheck_geom[0], 1ull);
    if (!same) atomicAdd(&g_dev_selfcheck_geom[1], 1ull);
  }
#endif
  for (int i = 0; i < 4; i++) {
    res->corners[2 * i] = cx[i];
    res->corners[2 * i + 1] = cy[i];
  }
  res->found_all = all ? 1 : 0;
  res->flags = 0;
}

#ifndef DMZ_HOMOGRAPHY_PAD256  /* developer switch (tools/dev/homography_fault.sh nocheck / wait run with 0: the 162-register kernel of round 5) */
#define DMZ_HOMOGRAPHY_PAD256 1
#endif
// calc_persp_transform, evaluated again until two consecutive results agree bit for bit (at most six times); false = they
// never did.  Round 5 measured why: with another queue's kernels (the expiry CNN of a previous frame chunk) in flight beside
// k_homography, about once per 65 536 frames one quarter-wave (always lanes 48..63) came out of this register-to-register
// computation with a wrong matrix from the right corners; a repeated evaluation in the same wave gave the right one
// (DESIGN_LOG.md "transient fault in k_homography", profiles/r5_homography_quarter_wave.log; cause not established).  The
// library's own pipeline never runs these kernels beside another one of the same context, a second context on the same GPU
// may: 40 us per 65 536 frames buy the check.
#ifdef DMZ_DEV_HCANARY  /* tools/dev/homography_fault.sh canary: registers the kernel never uses hold a pattern from its first to its last instruction */
__device__ unsigned g_hcanary[256][4];  // [event][frame of lane 0, register, changed lanes lo, hi]
__device__ unsigned g_hcanary_n;
#define HCANARY_SET(R) asm volatile("v_mov_b32 v" #R ", %0" ::"v"(0x5A5A0000u + R) : "v" #R);
#define HCANARY_CHK(R)                                                                                       \
  {                                                                                                          \
    unsigned long long cm_;                                                                                  \
    asm volatile("v_cmp_ne_u32 %0, v" #R ", %1" : "=s"(cm_) : "v"(0x5A5A0000u + R) : "v" #R);                \
    if (cm_ && threadIdx.x == 0) {                                                                           \
      const unsigned e_ = atomicAdd(&g_hcanary_n, 1u);                                                       \
      if (e_ < 256u) g_hcanary[e_][0] = blockIdx.x * 64u, g_hcanary[e_][1] = R, g_hcanary[e_][2] = (unsigned)cm_, g_hcanary[e_][3] = (unsigned)(cm_ >> 32); \
    }                                                                                                        \
  }
#define HCANARY_ALL(F) F(164) F(165) F(166) F(167) F(168) F(169) F(170) F(171) F(172) F(173) F(174) F(175) F(176) F(177) F(178) F(179) \
  F(180) F(181) F(182) F(183) F(184) F(185) F(186) F(187) F(188) F(189) F(190) F(191)
#endif
#ifdef DMZ_DEV_HTRACE  /* tools/dev/homography_fault.sh trace: two traced evaluations; a pair that disagrees is dumped */
__device__ unsigned g_htrace[64][40];  // [event][frame, 10 + 10 step hashes, 9 + 9 matrix bits, lane]
__device__ unsigned g_htrace_n;
#endif
template <bool SSE>
__device__ __forceinline__ bool persp_checked(const float *sp, const float *dp, float *m, int frame = -1) {
#ifdef DMZ_DEV_HTRACE
  {
    unsigned t1[10], t2[10];
    float m1[9], sp2[8];
    calc_persp_transform<SSE>(sp, dp, m1, t1);
    // (the second evaluation starts when the first has finished: its inputs are tied to the first one's last hash -- interleaved,
    // the two took 437 registers and the kernel no longer shared a SIMD with what makes it fault)
    for (int i = 0; i < 8; i++) {
      sp2[i] = sp[i];
      asm volatile("" : "+v"(sp2[i]), "+v"(t1[9]), "+v"(m1[i]));
    }
    __builtin_amdgcn_sched_barrier(0);
    calc_persp_transform<SSE>(sp2, dp, m, t2);
    bool same = true;
    for (int i = 0; i < 9; i++) same = same && __float_as_uint(m1[i]) == __float_as_uint(m[i]);
    if (!same) {
      const unsigned e = atomicAdd(&g_htrace_n, 1u);
      if (e < 64u) {
        g_htrace[e][0] = (unsigned)frame;
        for (int i = 0; i < 10; i++) g_htrace[e][1 + i] = t1[i], g_htrace[e][11 + i] = t2[i];
        for (int i = 0; i < 9; i++) g_htrace[e][21 + i] = __float_as_uint(m1[i]), g_htrace[e][30 + i] = __float_as_uint(m[i]);
        g_htrace[e][39] = threadIdx.x;
      }
    }
    return true;  // (the second evaluation is used, as the shipped kernel's first repeat would)
  }
#endif
  calc_persp_transform<SSE>(sp, dp, m);
#ifdef DMZ_HOMOGRAPHY_NOCHECK  /* developer switch (tools/dev/two_context_stress.py): the single evaluation of rounds 1 - 4 */
  return true;
#else
  for (int tries = 0; tries < 6; tries++) {
    float m2[9], sp2[8];
    for (int i = 0; i < 8; i++) {
      sp2[i] = sp[i];
      asm volatile("" : "+v"(sp2[i]));  // (not the same value to the compiler: the second evaluation is not folded into the first)
    }
    calc_persp_transform<SSE>(sp2, dp, m2);
    bool same = true;
    for (int i = 0; i < 9; i++) same = same && (__float_as_uint(m2[i]) == __float_as_uint(m[i]));
    for (int i = 0; i < 9; i++) m[i] = m2[i];
    if (same) return true;
  }
  return false;
#endif
}

// dmz_transform_card's part before the warp: corners -> source points -> float
// homography -> inverse double matrix (dmz.cpp:446-471, warp.cpp:153-165).
template <bool SSE>
__global__ __launch_bounds__(64) void k_homography(int n, int orientation, int options,
                             dmz_hip_frame_result *__restrict__ results,
                             DmzWarpMat *__restrict__ mats) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
#ifdef DMZ_DEV_HCANARY
  HCANARY_ALL(HCANARY_SET)
#endif
  if (f >= n) return;
  dmz_hip_frame_result *res = results + f;
  const bool all = res->found_all != 0;
  DmzWarpMat wm;
  wm.valid = all ? 1 : 0;
  wm.pad_ = 0;
  for (int i = 0; i < 9; i++) wm.m[i] = 0.;
  if (all) {
    int o0, o1, o2, o3;
    switch (orientation) {
      case 1: o0 = 1; o1 = 0; o2 = 3; o3 = 2; break;  // portrait: bl, tl, br, tr
      case 2: o0 = 2; o1 = 3; o2 = 0; o3 = 1; break;  // upside down: tr, br, tl, bl
      case 4: o0 = 3; o1 = 1; o2 = 2; o3 = 0; break;  // landscape left: br, bl, tr, tl
      default: o0 = 0; o1 = 2; o2 = 1; o3 = 3; break; // landscape right: tl, tr, bl, br
    }
    const int ord[4] = {o0, o1, o2, o3};
    float sp[8], dp[8], m[9];
    for (int i = 0; i < 4; i++) {
      float px = res->corners[2 * ord[i]], py = res->corners[2 * ord[i] + 1];
      if (options & DMZ_HIP_OPT_TRUNCATE_CORNERS) {
        px = (float)(int)px;
        py = (float)(int)py;
      }
      if (options & DMZ_HIP_OPT_UPSAMPLE) {  // dmz.cpp:473-481: a half-size chroma plane is being rectified
        px /= 2.0f;
        py /= 2.0f;
      }
      sp[2 * i] = px;
      sp[2 * i + 1] = py;
    }
#ifdef DMZ_DEV_HWAIT  /* developer probe (tools/dev/homography_fault.sh): every load of the kernel has landed, and then some, before the computation starts */
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_sleep 16" ::: "memory");
    for (int i = 0; i < 8; i++) asm volatile("" : "+v"(sp[i]));
#endif
#ifdef DMZ_DEV_HPADN  /* developer probe: claim registers up to v<DMZ_DEV_HPADN> (tools/dev/homography_fault.sh padn <N>) */
#define DMZ_STR2(x) #x
#define DMZ_STR(x) DMZ_STR2(x)
    asm volatile("v_mov_b32 v" DMZ_STR(DMZ_DEV_HPADN) ", 0" ::: "v" DMZ_STR(DMZ_DEV_HPADN));
#endif
#if DMZ_HOMOGRAPHY_PAD256
    // Round 6: the kernel claims a 256-register allocation.  The transient fault of round 5 does NOT depend on this kernel's
    // instruction stream: the same stream with every load landed and slept on (-DDMZ_DEV_HWAIT) faults as before (8 events in
    // 24 chunked passes of 65 536 frames, as the plain single evaluation), while the same stream with this one extra register
    // write -- 162 -> 256 registers, so that at most one other large wave shares its SIMD's register file instead of two -- did
    // not fault once in 24 passes; neither did the traced build (437 registers).  k_geometry and k_warp_windows (the same
    // division sequences, 40 / 62 registers) evaluated twice under the same load: 0 of 23 million pairs differ.  What decides is
    // what shares the register file, not what the wave executes (profiles/r6_homography_fault_probes.log).  The self-check
    // below stays: it costs 40 us per 65 536 frames and is what makes a recurrence visible (DMZ_HIP_FLAG_FAULT).
    asm volatile("v_mov_b32 v255, 0" ::: "v255");
#endif
    const float rw = (float)(DMZ_CARD_WIDTH - 1), rh = (float)(DMZ_CARD_HEIGHT - 1);
    dp[0] = 0.0f; dp[1] = 0.0f; dp[2] = 0.0f + rw; dp[3] = 0.0f;
    dp[4] = 0.0f; dp[5] = 0.0f + rh; dp[6] = 0.0f + rw; dp[7] = 0.0f + rh;
    if (persp_checked<SSE>(sp, dp, m, f)) {
      invert3x3(m, &wm);
      res->flags = (res->flags & ~(DMZ_HIP_FLAG_WARPED | DMZ_HIP_FLAG_FAULT)) | DMZ_HIP_FLAG_WARPED;
    } else {
      // six evaluations, no two consecutive ones alike: not a result.  The frame is not rectified (its card stays zero, the
      // scan stages skip it) and the record says why.
      wm.valid = 0;
      res->flags = (res->flags & ~DMZ_HIP_FLAG_WARPED) | DMZ_HIP_FLAG_FAULT;
    }
  } else {
    res->flags = res->flags & ~DMZ_HIP_FLAG_WARPED;
  }
  dmz_store_mat_head(mats + f, wm);
#ifdef DMZ_DEV_HCANARY
  HCANARY_ALL(HCANARY_CHK)
#endif
}

template <bool SSE>
__global__ __launch_bounds__(64) void k_persp(int n, const float *__restrict__ src_pts, const float *__restrict__ dst_pts,
                        float *__restrict__ m9) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= n) return;
  float sp[8], dp[8], m[9];
  for (int i = 0; i < 8; i++) {
    sp[i] = src_pts[f * 8 + i];
    dp[i] = dst_pts[f * 8 + i];
  }
#if DMZ_HOMOGRAPHY_PAD256
  asm volatile("v_mov_b32 v255, 0" ::: "v255");  // (as k_homography)
#endif
  // (no flag travels with a bare matrix: a self-check that never settles returns NaNs, not a plausible wrong homography)
  const bool ok = persp_checked<SSE>(sp, dp, m);
  for (int i = 0; i < 9; i++) m9[f * 9 + i] = ok ? m[i] : __uint_as_float(0x7FC00000u);
}

__global__ __launch_bounds__(64) void k_mats_from_float(int n, const float *__restrict__ m9, DmzWarpMat *__restrict__ mats) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= n) return;
  float m[9];
  for (int i = 0; i < 9; i++) m[i] = m9[f * 9 + i];
  DmzWarpMat wm;
  wm.valid = 1;
  wm.pad_ = 0;
  invert3x3(m, &wm);
  dmz_store_mat_head(mats + f, wm);
}

}  // namespace

#ifdef DMZ_DEV_HCANARY
extern "C" int dmz_dbg_hcanary(unsigned *out /* 256 x 4 */) {
  unsigned n = 0;
  (void)hipDeviceSynchronize();
  (void)hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_hcanary_n), sizeof(n));
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_hcanary), sizeof(unsigned) * 256 * 4);
  return (int)n;
}
#endif
#ifdef DMZ_DEV_HTRACE
extern "C" int dmz_dbg_htrace(unsigned *out /* 64 x 40 */) {
  unsigned n = 0;
  (void)hipDeviceSynchronize();
  (void)hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_htrace_n), sizeof(n));
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_htrace), sizeof(unsigned) * 64 * 40);
  return (int)n;
}
#endif
#ifdef DMZ_DEV_SELFCHECK
extern "C" void dmz_dbg_selfcheck_geom(unsigned long long *out) {
  (void)hipDeviceSynchronize();
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dev_selfcheck_geom), sizeof(unsigned long long) * 2);
}
#endif

void dmz_launch_geometry(hipStream_t s, int n, const DmzDetectParams *params, const DmzBoxHit *hits,
                         int nplanes, dmz_hip_frame_result *results) {
  hipLaunchKernelGGL(k_geometry, dim3((n + 63) / 64), dim3(64), 0, s, n, params, hits, nplanes,
                     results);
}

void dmz_launch_homography(hipStream_t s, int n, int orientation, int options,
                           dmz_hip_frame_result *results, DmzWarpMat *mats) {
  if (options & DMZ_HIP_OPT_EIGEN_SSE2)
    hipLaunchKernelGGL(k_homography<true>, dim3((n + 63) / 64), dim3(64), 0, s, n, orientation, options, results, mats);
  else
    hipLaunchKernelGGL(k_homography<false>, dim3((n + 63) / 64), dim3(64), 0, s, n, orientation, options, results, mats);
}

void dmz_launch_persp(hipStream_t s, int n, const float *src_pts, const float *dst_pts, float *m9, int options) {
  if (options & DMZ_HIP_OPT_EIGEN_SSE2)
    hipLaunchKernelGGL(k_persp<true>, dim3((n + 63) / 64), dim3(64), 0, s, n, src_pts, dst_pts, m9);
  else
    hipLaunchKernelGGL(k_persp<false>, dim3((n + 63) / 64), dim3(64), 0, s, n, src_pts, dst_pts, m9);
}

void dmz_launch_mats_from_float(hipStream_t s, int n, const float *m9, DmzWarpMat *mats) {
  hipLaunchKernelGGL(k_mats_from_float, dim3((n + 63) / 64), dim3(64), 0, s, n, m9, mats);
}

