// session.hip -- the per-session half of the scan, batched: sessions x frames of per-frame records
// (dmz_hip_frame_result / dmz_hip_expiry_result, already resident in HBM) are reduced on the
// device to one dmz_hip_session_result per session, so that a server that scans many video
// sessions never moves the 2.6 KB-per-frame records across PCIe.
//
// Replays, frame by frame, what an SDK loop does with the reference:
//   scanner_add_frame_with_expiry  scan/scan.cpp:41-86   (what is still needed, the decayed score
//                                  sums, expiry_extract's aggregation half)
//   scanner_result                 scan/scan.cpp:88-194  (count lead, stability, issuer prefix,
//                                  Luhn, the wait for an expiry)
//   expiry_aggregate_grouped_rects / get_stable_expiry_month_and_year
//                                  scan/expiry_categorize.cpp:162-330
//   dmz_passes_luhn_checksum / dmz_card_info_for_prefix_and_length   dmz_olm.cpp:40-130
// with the wall clock replaced by a frame clock (frame f is handled at t = 1 + f * interval ms)
// and the calendar date passed in.  All of it is exact integer / ordered float arithmetic (no
// contraction), so the result is bit-identical to the CPU oracle (oracle/orc_session.c), which is
// itself pinned against the reference's own compiled scanner_result.
//
// One wave per session: the state (two 16 x 10 score sums, <= 32 aggregated expiry groups) lives in
// LDS; the 160-element decayed sums and the per-digit stability test are lane-parallel, the list
// logic of the expiry aggregation is a few dozen steps on lane 0.
#include "dmz_hip_internal.h"

namespace {

constexpr int MAX_AGG = 32;
constexpr float kDecayFactor = 0.8f, kMinStability = 0.7f;
constexpr float kExpiryDecayFactor = 0.7f, kExpiryMinStability = 0.7f;
constexpr int V_ALLOW = 16 / 2, H_ALLOW = 11 / 2;
constexpr long EXTRA_TIME_FOR_EXPIRY = 1000;  // scan.cpp:14, compared with milliseconds

struct AggGroup {
  int top, left, n_chars, recently_seen, total_seen;
  float scores[5][10];
};

struct SessLds {
  float agg15[160], agg16[160];
  AggGroup groups[MAX_AGG];
  AggGroup nw[DMZ_HIP_EXPIRY_MAX_GROUPS];
  int n_groups, em, ey;
  int pred[16], unstable[16];
};

__device__ __forceinline__ int iabs(int a) { return a < 0 ? -a : a; }
__device__ __forceinline__ float row_sum10(const float *p) {  // Eigen scalar redux of 10
  return ((p[0] + p[1]) + (p[2] + (p[3] + p[4]))) + ((p[5] + p[6]) + (p[7] + (p[8] + p[9])));
}
__device__ __forceinline__ int row_argmax10(const float *p) {  // first maximum
  int b = 0;
  for (int k = 1; k < 10; k++)
    if (p[k] > p[b]) b = k;
  return b;
}

// dmz_olm.cpp:51-130: issuer prefix ranges and the single-match rule (complete numbers only)
__device__ int card_type_of(const int *digits, int n) {
  const int types[20] = {5, 6, 6, 2, 3, 6, 6, 2, 4, 7, 5, 7, 6, 7, 6, 7, 6, 6, 7, 6};
  const int lens[20] = {16, 14, 14, 15, 16, 14, 14, 15, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16};
  const int plens[20] = {4, 3, 3, 2, 4, 2, 2, 2, 1, 2, 2, 2, 4, 2, 2, 2, 3, 2, 2, 2};
  const int lo[20] = {2221, 300, 309, 34, 3528, 36, 38, 37, 4, 50, 51, 56, 6011, 61, 62, 63, 644, 65, 66, 88};
  const int hi[20] = {2720, 305, 309, 34, 3589, 36, 39, 37, 4, 50, 55, 59, 6011, 61, 62, 63, 649, 65, 69, 88};
  int matches = 0, type = 0;
  for (int t = 0; t < 20; t++) {
    if (n != lens[t]) continue;
    int prefix = 0;
    for (int j = 0; j < plens[t]; j++) prefix = prefix * 10 + digits[j];
    if (prefix >= lo[t] && prefix <= hi[t]) {
      matches++;
      type = types[t];
    }
  }
  return matches == 1 ? type : (matches > 1 ? 1 : 0);
}

__device__ bool passes_luhn(const int *digits, int n) {  // dmz_olm.cpp:40-49
  int sum = 0;
  bool doubled = false;
  for (int i = n - 1; i >= 0; i--) {
    const int addend = digits[i] * (doubled ? 2 : 1);
    sum += addend % 10 + addend / 10;
    doubled = !doubled;
  }
  return sum % 10 == 0;
}

// expiry_categorize.cpp:230-330 for the MM/YY pattern
__device__ void stable_month_year(const AggGroup &g, int now_year, int now_month, int allow_past, int *em, int *ey) {
  int ch[5] = {-1, -1, -1, -1, -1};  // -1 = ' '
  for (int i = 0; i < g.n_chars && i < 5; i++) {
    if (i == 2) continue;
    const float *p = g.scores[i];
    const int best = row_argmax10(p);
    const float stability = p[best] / row_sum10(p);
    ch[i] = stability < kExpiryMinStability ? -1 : best;
  }
  int month = -1, year = -1;
  if (ch[0] >= 0 && ch[1] >= 0 && ch[3] >= 0 && ch[4] >= 0) {
    month = ch[0] * 10 + ch[1];
    year = ch[3] * 10 + ch[4];
  }
  if (month > 12 && year > 0 && year <= 12) {
    const int t = month;
    month = year;
    year = t;
  }
  int full_year = year + 2000;
  if (month > 0 && month <= 12 && (full_year > *ey || (full_year == *ey && month > *em))) {
    if (full_year < now_year + 5 && (full_year > now_year || (full_year == now_year && month >= now_month))) {
      *em = month;
      *ey = full_year;
    } else if (allow_past) {
      if (year > 60) full_year = year + 1900;
      if (full_year < now_year + 5) {
        *em = month;
        *ey = full_year;
      }
    }
  }
}

__device__ void erase_group(AggGroup *list, int *n, int idx) {
  for (int i = idx; i + 1 < *n; i++) list[i] = list[i + 1];
  (*n)--;
}

// expiry_categorize.cpp:162-228 on fixed arrays
__device__ void aggregate_groups(AggGroup *agg, int *n_agg, AggGroup *nw, int n_new) {
  for (int i1 = 0; i1 < n_new; i1++) {
    float coalesced = 1;
    for (int i2 = n_new - 1; i2 > i1; i2--) {
      if (iabs(nw[i2].top - nw[i1].top) > V_ALLOW || iabs(nw[i2].left - nw[i1].left) > H_ALLOW ||
          nw[i2].n_chars != nw[i1].n_chars)
        continue;
      for (int r = 0; r < 5; r++)
        for (int c = 0; c < 10; c++)
          nw[i1].scores[r][c] = ((nw[i1].scores[r][c] * coalesced) + nw[i2].scores[r][c]) / (coalesced + 1);
      coalesced++;
      erase_group(nw, &n_new, i2);
    }
  }
  for (int o = 0; o < *n_agg; o++) {
    const int old_top = agg[o].top, old_left = agg[o].left;
    for (int ni = n_new - 1; ni >= 0; ni--) {
      if (iabs(nw[ni].top - old_top) > V_ALLOW || iabs(nw[ni].left - old_left) > H_ALLOW ||
          nw[ni].n_chars != agg[o].n_chars)
        continue;
      agg[o].recently_seen++;
      agg[o].total_seen++;
      for (int r = 0; r < 5; r++)
        for (int c = 0; c < 10; c++)
          agg[o].scores[r][c] = (agg[o].scores[r][c] * kExpiryDecayFactor) + (nw[ni].scores[r][c] * (1 - kExpiryDecayFactor));
      agg[o].top = nw[ni].top;
      agg[o].left = nw[ni].left;
      erase_group(nw, &n_new, ni);
    }
  }
  for (int o = *n_agg - 1; o >= 0; o--) {
    agg[o].recently_seen--;
    if (agg[o].recently_seen <= 0) erase_group(agg, n_agg, o);
  }
  for (int i = 0; i < n_new && *n_agg < MAX_AGG; i++) {
    agg[*n_agg] = nw[i];
    agg[*n_agg].recently_seen = 3;
    agg[*n_agg].total_seen = 1;
    (*n_agg)++;
  }
}

__global__ __launch_bounds__(64) void k_scan_sessions(const dmz_hip_frame_result *__restrict__ frames,
                                                      const dmz_hip_expiry_result *__restrict__ expiry,
                                                      int n_sessions, int frames_per_session, int scan_expiry,
                                                      int frame_interval_ms, int now_year, int now_month,
                                                      int allow_past, dmz_hip_session_result *__restrict__ out) {
  const int s = blockIdx.x, lane = threadIdx.x;
  if (s >= n_sessions) return;
  __shared__ SessLds L;
  for (int i = lane; i < 160; i += 64) L.agg15[i] = 0.0f, L.agg16[i] = 0.0f;
  if (lane == 0) L.n_groups = 0, L.em = 0, L.ey = 0;
  __syncthreads();
  // ScannerState scalars: every lane keeps an identical copy
  int count15 = 0, count16 = 0, st_scan_expiry = 0, em = 0, ey = 0, usable_frames = 0;
  long t_number = 0;
  int recent = -1;  // frame index of mostRecentUsableHSeg/VSeg
  // successfulCardNumberResult
  int succ_n = 0, succ_type = 0, succ_recent = -1;
  int succ_pred[16];
#pragma unroll
  for (int i = 0; i < 16; i++) succ_pred[i] = 0;
  int number_frame = -1, complete_frame = -1, complete = 0, res_em = 0, res_ey = 0;

  const dmz_hip_frame_result *fbase = frames + (size_t)s * frames_per_session;
  const dmz_hip_expiry_result *xbase = expiry ? expiry + (size_t)s * frames_per_session : nullptr;
  for (int f = 0; f < frames_per_session; f++) {
    const long now = 1 + (long)f * frame_interval_ms;
    const dmz_hip_frame_result *fr = fbase + f;
    const int flags = fr->flags, noff = fr->n_offsets;
    // ---- scanner_add_frame_with_expiry (scan.cpp:41-86) ----
    const bool need_number = t_number == 0;
    const bool need_expiry = scan_expiry && (em == 0 || ey == 0);
    bool usable = false;
    if (!(flags & DMZ_HIP_FLAG_UPSIDE_DOWN))
      usable = need_number ? (flags & DMZ_HIP_FLAG_USABLE) != 0 : (flags & DMZ_HIP_FLAG_VSEG_OK) != 0;
    if (usable) {
      usable_frames++;
      if (need_expiry) {
        st_scan_expiry = 1;
        const dmz_hip_expiry_result *x = xbase ? xbase + f : nullptr;
        if (x && x->categorised && x->n_groups > 0) {  // expiry_extract (expiry_categorize.cpp:332-376)
          const int ng = x->n_groups < DMZ_HIP_EXPIRY_MAX_GROUPS ? x->n_groups : DMZ_HIP_EXPIRY_MAX_GROUPS;
          for (int i = lane; i < ng * 50; i += 64) {
            const int g = i / 50, e = i - g * 50, r = e / 10, c = e - r * 10;
            float v = 0.0f;
            if (r != 2) v = x->groups[g].scores[r < 2 ? r : r - 1][c];
            L.nw[g].scores[r][c] = v;
          }
          if (lane < ng) {
            L.nw[lane].top = x->groups[lane].top;
            L.nw[lane].left = x->groups[lane].left;
            L.nw[lane].n_chars = 5;
            L.nw[lane].recently_seen = 0;
            L.nw[lane].total_seen = 0;
          }
          __syncthreads();
          if (lane == 0) {
            int n_groups = L.n_groups, m = em, y = ey;
            aggregate_groups(L.groups, &n_groups, L.nw, ng);
            for (int g = 0; g < n_groups; g++) {
              if (L.groups[g].total_seen < 3) continue;
              stable_month_year(L.groups[g], now_year, now_month, allow_past, &m, &y);
            }
            L.n_groups = n_groups;
            L.em = m;
            L.ey = y;
          }
          __syncthreads();
          em = L.em;
          ey = L.ey;
        }
      }
      if (need_number) {
        recent = f;
        if (noff == 15 || noff == 16) {
          float *agg = noff == 15 ? L.agg15 : L.agg16;
          const float *sc = &fr->scores[0][0];
          for (int i = lane; i < 160; i += 64) {
            float a = agg[i];
            a = a * kDecayFactor;
            a = a + sc[i] * (1 - kDecayFactor);
            agg[i] = a;
          }
          if (noff == 15) count15++;
          else count16++;
          __syncthreads();
        }
      }
    }
    // ---- scanner_result (scan.cpp:88-194) ----
    bool bail = false;
    int res_n = 0, res_recent = -1;
    if (t_number > 0) {
      res_n = succ_n;
      res_recent = succ_recent;
    } else {
      const int max_count = count15 > count16 ? count15 : count16, min_count = count15 > count16 ? count16 : count15;
      if (max_count - min_count < 3 || min_count * 2 > max_count) bail = true;
      if (!bail) {
        res_recent = recent;
        const float *agg;
        if (count15 > count16) res_n = 15, agg = L.agg15;
        else res_n = 16, agg = L.agg16;
        if (lane < 16) {
          const int best = row_argmax10(agg + lane * 10);
          L.pred[lane] = best;
          L.unstable[lane] = lane < res_n && agg[lane * 10 + best] / row_sum10(agg + lane * 10) < kMinStability;
        }
        __syncthreads();
        int pred[16];
        bool any_unstable = false;
#pragma unroll
        for (int i = 0; i < 16; i++) {
          pred[i] = L.pred[i];
          any_unstable = any_unstable || (L.unstable[i] != 0);
        }
        __syncthreads();
        if (any_unstable) bail = true;
        if (!bail) {
          const int type = card_type_of(pred, res_n);
          if (type != 0 && type != 1 && passes_luhn(pred, res_n)) {
            t_number = now;
            succ_n = res_n;
            succ_type = type;
            succ_recent = res_recent;
#pragma unroll
            for (int i = 0; i < 16; i++) succ_pred[i] = i < res_n ? pred[i] : 0;
            number_frame = f;
          }
        }
      }
    }
    if (!bail && t_number > 0) {
      if (st_scan_expiry) {
        if ((em > 0 && ey > 0) || now - t_number > EXTRA_TIME_FOR_EXPIRY) {
          res_em = em;
          res_ey = ey;
          complete = 1;
        }
      } else {
        res_em = 0;
        res_ey = 0;
        complete = 1;
      }
    }
    if (complete) {
      complete_frame = f;
      break;
    }
  }
  if (lane == 0) {
    dmz_hip_session_result r;
    uint32_t *z = (uint32_t *)&r;
    for (int i = 0; i < (int)(sizeof(r) / 4); i++) z[i] = 0u;
    r.complete = complete;
    r.complete_frame = complete_frame;
    r.number_frame = number_frame;
    if (t_number > 0) {
      r.n_numbers = succ_n;
      for (int i = 0; i < 16; i++) r.predictions[i] = (uint8_t)succ_pred[i];
      r.card_type = succ_type;
      if (succ_recent >= 0) {
        const dmz_hip_frame_result *fr = fbase + succ_recent;
        r.vseg_y_offset = fr->vseg_y_offset;
        r.n_offsets = fr->n_offsets;
        for (int i = 0; i < 16; i++) r.offsets[i] = fr->offsets[i];
        r.number_width = fr->number_width;
      }
    }
    r.expiry_month = complete ? res_em : em;
    r.expiry_year = complete ? res_ey : ey;
    r.count15 = count15;
    r.count16 = count16;
    r.usable_frames = usable_frames;
    r.n_expiry_groups = L.n_groups;
    out[s] = r;
  }
}

}  // namespace

void dmz_launch_sessions(hipStream_t st, const dmz_hip_frame_result *frames, const dmz_hip_expiry_result *expiry,
                         int n_sessions, int frames_per_session, int scan_expiry, int frame_interval_ms, int now_year,
                         int now_month, int allow_past, dmz_hip_session_result *out) {
  hipLaunchKernelGGL(k_scan_sessions, dim3((unsigned)n_sessions), dim3(64), 0, st, frames, expiry, n_sessions,
                     frames_per_session, scan_expiry, frame_interval_ms, now_year, now_month, allow_past, out);
}
