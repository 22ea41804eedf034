/*
 * orc_session.c -- CPU ORACLE (test infrastructure, see dmz_oracle.h) for the per-session half
 * of the scan: scan/scan.cpp:41-194 (scanner_add_frame_with_expiry's state update, scanner_result's
 * policy), dmz_olm.cpp:40-130 (Luhn, issuer prefix table) and the cross-frame half of
 * scan/expiry_categorize.cpp:162-376 (expiry_aggregate_grouped_rects, get_stable_expiry_month_and_year,
 * expiry_extract's loop), replayed over the per-frame records of one session.
 *
 * The replay is the SDK loop  add frame -> scanner_result -> stop when complete  with the wall
 * clock replaced by a frame clock: frame f is handled at t = 1 + f * frame_interval_ms, and the
 * calendar date the expiry policy compares with is an argument.  Pinned against the reference's
 * own scanner_result / expiry functions (tests/test_oracle_vs_ref.py) at frame_interval_ms = 0.
 *
 * Batched-mode convention: the per-frame records were produced with the number path always on
 * (collect_card_number = true).  After the number is accepted the reference stops collecting
 * it and `usable` is the vseg gate alone (frame.cpp:43,49); the replay uses ORC_FLAG_VSEG_OK
 * there, and an expiry record that was not categorised contributes no groups.
 */
#include "dmz_oracle.h"

#include <stdlib.h>
#include <string.h>

#define kDecayFactor 0.8f
#define kMinStability 0.7f
#define EXTRA_TIME_FOR_EXPIRY 1000 /* scan.cpp:14, compared with milliseconds */
#define kExpiryDecayFactor 0.7f
#define kExpiryMinStability 0.7f
#define V_ALLOW (16 / 2)
#define H_ALLOW (11 / 2)
#define MAX_AGG 32

typedef struct {
  int top, left, n_chars;
  float scores[5][10];
  int recently_seen, total_seen;
} agg_group;

/* dmz_olm.cpp:51-130; returns the card type, 0 unrecognized, 1 ambiguous */
int orc_card_type(const uint8_t *digits, int n, int allow_incomplete, int *number_length) {
  static const struct { int type, len, plen; long lo, hi; } table[] = {
      {5, 16, 4, 2221, 2720}, {6, 14, 3, 300, 305}, {6, 14, 3, 309, 309}, {2, 15, 2, 34, 34},
      {3, 16, 4, 3528, 3589}, {6, 14, 2, 36, 36},   {6, 14, 2, 38, 39},   {2, 15, 2, 37, 37},
      {4, 16, 1, 4, 4},       {7, 16, 2, 50, 50},   {5, 16, 2, 51, 55},   {7, 16, 2, 56, 59},
      {6, 16, 4, 6011, 6011}, {7, 16, 2, 61, 61},   {6, 16, 2, 62, 62},   {7, 16, 2, 63, 63},
      {6, 16, 3, 644, 649},   {6, 16, 2, 65, 65},   {7, 16, 2, 66, 69},   {6, 16, 2, 88, 88},
  };
  int matches = 0, type = 0, len = -1;
  if (n == 0) {
    if (number_length) *number_length = -1;
    return 0;
  }
  for (size_t t = 0; t < sizeof(table) / sizeof(table[0]); t++) {
    if (allow_incomplete ? n > table[t].len : n != table[t].len) continue;
    int plen = table[t].plen;
    long factor = 1, prefix = 0;
    for (; plen > n; plen--) factor *= 10;
    for (int j = 0; j < plen; j++) prefix = prefix * 10 + digits[j];
    if (prefix >= table[t].lo / factor && prefix <= table[t].hi / factor) {
      matches++;
      type = table[t].type;
      len = table[t].len;
    }
  }
  if (matches != 1) {
    type = matches > 1 ? 1 : 0;
    len = -1;
  }
  if (number_length) *number_length = len;
  return type;
}

static float row_sum10(const float *p) { /* Eigen scalar redux of 10 */
  return ((p[0] + p[1]) + (p[2] + (p[3] + p[4]))) + ((p[5] + p[6]) + (p[7] + (p[8] + p[9])));
}
static int row_argmax10(const float *p) { /* first maximum */
  int b = 0;
  for (int k = 1; k < 10; k++)
    if (p[k] > p[b]) b = k;
  return b;
}

/* expiry_categorize.cpp:230-330 for the MM/YY pattern */
static void stable_month_year(const agg_group *g, int now_year, int now_month, int allow_past, int *em, int *ey) {
  char s[8] = {0};
  for (int i = 0; i < g->n_chars; i++) {
    if (i == 2) continue;
    const float *p = g->scores[i];
    const int best = row_argmax10(p);
    const float stability = p[best] / row_sum10(p);
    s[i] = stability < kExpiryMinStability ? ' ' : (char)('0' + best);
  }
  int month = -1, year = -1;
  if (s[0] != ' ' && s[1] != ' ' && s[3] != ' ' && s[4] != ' ') {
    month = (uint8_t)(s[0] - '0') * 10 + (uint8_t)(s[1] - '0');
    year = (uint8_t)(s[3] - '0') * 10 + (uint8_t)(s[4] - '0');
  }
  if (month > 12 && year > 0 && year <= 12) {
    int t = month;
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

/* expiry_categorize.cpp:162-228 on fixed arrays */
static void aggregate_groups(agg_group *agg, int *n_agg, agg_group *nw, int n_new) {
  for (int i1 = 0; i1 < n_new; i1++) {
    float coalesced = 1;
    for (int i2 = n_new - 1; i2 > i1; i2--) {
      if (abs(nw[i2].top - nw[i1].top) > V_ALLOW || abs(nw[i2].left - nw[i1].left) > H_ALLOW ||
          nw[i2].n_chars != nw[i1].n_chars)
        continue;
      for (int r = 0; r < 5; r++)
        for (int c = 0; c < 10; c++)
          nw[i1].scores[r][c] = ((nw[i1].scores[r][c] * coalesced) + nw[i2].scores[r][c]) / (coalesced + 1);
      coalesced++;
      memmove(&nw[i2], &nw[i2 + 1], sizeof(agg_group) * (size_t)(n_new - 1 - i2));
      n_new--;
    }
  }
  for (int o = 0; o < *n_agg; o++) {
    const int old_top = agg[o].top, old_left = agg[o].left; /* captured before the loop, as the reference does */
    for (int ni = n_new - 1; ni >= 0; ni--) {
      if (abs(nw[ni].top - old_top) > V_ALLOW || abs(nw[ni].left - old_left) > H_ALLOW ||
          nw[ni].n_chars != agg[o].n_chars)
        continue;
      agg[o].recently_seen++;
      agg[o].total_seen++;
      for (int r = 0; r < 5; r++)
        for (int c = 0; c < 10; c++)
          agg[o].scores[r][c] = (agg[o].scores[r][c] * kExpiryDecayFactor) + (nw[ni].scores[r][c] * (1 - kExpiryDecayFactor));
      agg[o].top = nw[ni].top;
      agg[o].left = nw[ni].left;
      memmove(&nw[ni], &nw[ni + 1], sizeof(agg_group) * (size_t)(n_new - 1 - ni));
      n_new--;
    }
  }
  for (int o = *n_agg - 1; o >= 0; o--) {
    agg[o].recently_seen--;
    if (agg[o].recently_seen <= 0) {
      memmove(&agg[o], &agg[o + 1], sizeof(agg_group) * (size_t)(*n_agg - 1 - o));
      (*n_agg)--;
    }
  }
  for (int i = 0; i < n_new && *n_agg < MAX_AGG; i++) {
    agg[*n_agg] = nw[i];
    agg[*n_agg].recently_seen = 3;
    agg[*n_agg].total_seen = 1;
    (*n_agg)++;
  }
}

void orc_scan_session(const orc_frame_result *frames, const orc_expiry_result *expiry, int n_frames,
                      int scan_expiry, int frame_interval_ms, int now_year, int now_month, int allow_past_expiry,
                      orc_session_result *out) {
  /* ScannerState (scan.h:31-48) */
  int count15 = 0, count16 = 0;
  float agg15[16][10], agg16[16][10];
  memset(agg15, 0, sizeof(agg15));
  memset(agg16, 0, sizeof(agg16));
  long t_number = 0;
  int st_scan_expiry = 0, em = 0, ey = 0;
  static __thread agg_group groups[MAX_AGG];
  int n_groups = 0;
  const orc_frame_result *recent = NULL;
  orc_session_result success; /* successfulCardNumberResult */
  memset(&success, 0, sizeof(success));
  memset(out, 0, sizeof(*out));
  out->complete_frame = -1;
  out->number_frame = -1;

  for (int f = 0; f < n_frames; f++) {
    const long now = 1 + (long)f * frame_interval_ms;
    const orc_frame_result *fr = &frames[f];
    /* ---- scanner_add_frame_with_expiry (scan.cpp:41-86) ---- */
    const int need_number = t_number == 0;
    const int need_expiry = scan_expiry && (em == 0 || ey == 0);
    int usable = 0;
    if (!(fr->flags & ORC_FLAG_UPSIDE_DOWN))
      usable = need_number ? (fr->flags & ORC_FLAG_USABLE) != 0 : (fr->flags & ORC_FLAG_VSEG_OK) != 0;
    if (usable) {
      out->usable_frames++;
      if (need_expiry) {
        st_scan_expiry = 1;
        const orc_expiry_result *x = expiry ? &expiry[f] : NULL;
        if (x && x->categorised && x->n_groups > 0) { /* expiry_extract (expiry_categorize.cpp:332-376) */
          agg_group nw[ORC_EXPIRY_MAX_GROUPS];
          for (int g = 0; g < x->n_groups; g++) {
            nw[g].top = x->groups[g].top;
            nw[g].left = x->groups[g].left;
            nw[g].n_chars = 5;
            memset(nw[g].scores, 0, sizeof(nw[g].scores));
            for (int r = 0; r < 4; r++) memcpy(nw[g].scores[r < 2 ? r : r + 1], x->groups[g].scores[r], sizeof(float) * 10);
            nw[g].recently_seen = nw[g].total_seen = 0;
          }
          aggregate_groups(groups, &n_groups, nw, x->n_groups);
          for (int g = 0; g < n_groups; g++) {
            if (groups[g].total_seen < 3) continue;
            stable_month_year(&groups[g], now_year, now_month, allow_past_expiry, &em, &ey);
          }
        }
      }
      if (need_number) {
        recent = fr;
        float(*agg)[10] = fr->n_offsets == 15 ? agg15 : (fr->n_offsets == 16 ? agg16 : NULL);
        if (agg) {
          for (int i = 0; i < 16; i++)
            for (int k = 0; k < 10; k++) {
              agg[i][k] = agg[i][k] * kDecayFactor;
              agg[i][k] = agg[i][k] + fr->scores[i][k] * (1 - kDecayFactor);
            }
          if (fr->n_offsets == 15) count15++;
          else count16++;
        }
      }
    }
    /* ---- scanner_result (scan.cpp:88-194) ---- */
    orc_session_result res;
    memset(&res, 0, sizeof(res));
    int bail = 0;
    if (t_number > 0) {
      res = success;
    } else {
      const int max_count = count15 > count16 ? count15 : count16, min_count = count15 > count16 ? count16 : count15;
      if (max_count - min_count < 3 || min_count * 2 > max_count) bail = 1;
      if (!bail) {
        res.vseg_y_offset = recent->vseg_y_offset;
        res.n_offsets = recent->n_offsets;
        memcpy(res.offsets, recent->offsets, sizeof(res.offsets));
        res.number_width = recent->number_width;
        float(*agg)[10];
        if (count15 > count16) res.n_numbers = 15, agg = agg15;
        else res.n_numbers = 16, agg = agg16;
        for (int i = 0; i < res.n_numbers && !bail; i++) {
          const int best = row_argmax10(agg[i]);
          res.predictions[i] = (uint8_t)best;
          if (agg[i][best] / row_sum10(agg[i]) < kMinStability) bail = 1;
        }
      }
      if (!bail) {
        const int type = orc_card_type(res.predictions, res.n_numbers, 0, NULL);
        if (type != 0 && type != 1 && orc_passes_luhn(res.predictions, res.n_numbers)) {
          t_number = now;
          res.card_type = type;
          success = res;
          out->number_frame = f;
        }
      }
    }
    if (!bail && t_number > 0) {
      if (st_scan_expiry) {
        if ((em > 0 && ey > 0) || now - t_number > EXTRA_TIME_FOR_EXPIRY) {
          res.expiry_month = em;
          res.expiry_year = ey;
          res.complete = 1;
        }
      } else {
        res.expiry_month = 0;
        res.expiry_year = 0;
        res.complete = 1;
      }
    }
    if (res.complete) {
      const int uf = out->usable_frames, nf = out->number_frame;
      *out = res;
      out->usable_frames = uf;
      out->number_frame = nf;
      out->complete_frame = f;
      break;
    }
  }
  if (!out->complete) { /* what the session knows so far */
    const int uf = out->usable_frames, nf = out->number_frame;
    if (t_number > 0) *out = success;
    out->usable_frames = uf;
    out->number_frame = nf;
    out->complete = 0;
    out->complete_frame = -1;
    out->expiry_month = em;
    out->expiry_year = ey;
  }
  out->count15 = count15;
  out->count16 = count16;
  out->n_expiry_groups = n_groups;
}
