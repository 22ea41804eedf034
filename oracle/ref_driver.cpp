// ref_driver.cpp -- recipe TU for the PARTIAL REFERENCE BUILD oracle/_ref/libdmzref.so.
//
// Test infrastructure only (see dmz_oracle.h).  It compiles the reference's own
// unity translation unit from where it lies (-I/root/reference; nothing is
// copied) and exports thin extern "C" wrappers around those reference functions
// that need only the vendored Eigen / OpenCV *headers*.  The OpenCV *libraries*
// are not in the image, so every reference function that calls a cv* library
// symbol (cvGetMat, cvGetSize, cvSobel, cvWarpPerspective, ...) is left
// unreachable and dropped by --gc-sections; no stand-in for any of them is
// written.  Flavour: -DCYTHON_DMZ=1 -DSCAN_EXPIRY=1 -DEIGEN_DONT_VECTORIZE
// -O2 -ffp-contract=off (the pinned oracle flavour).
//
// What this pins (tests/test_oracle_vs_ref.py): the six generated models and
// their pass*_() known-answer tests, geometry.cpp (lineByShiftingOrigin,
// parametricIntersect), llcv_calc_persp_transform (Eigen HouseholderQR),
// best_segmentation_for_vseg_scores, best_n_hseg_constrained, Luhn.
#include "dmz_all.cpp"

#define REF_API extern "C" __attribute__((visibility("default")))

REF_API int ref_pass_kats(void) {
  int ok = 0;
  ok |= passm_befe75da() ? 1 : 0;
  ok |= passc_5c241121() ? 2 : 0;
  ok |= passc_01266c1b() ? 4 : 0;
  ok |= passc_b00bf70c() ? 8 : 0;
  ok |= passm_730c4cbd() ? 16 : 0;
  ok |= passc_bf4dd6c8() ? 32 : 0;
  return ok;
}

REF_API void ref_applym_vseg(const float *x, float *out) {
  ModelMInput_befe75da in;
  for (int i = 0; i < 204; i++) in(i, 0) = x[i];
  ModelMOutput_befe75da o = applym_befe75da(in);
  for (int i = 0; i < 3; i++) out[i] = o(i, 0);
}

REF_API void ref_applyc_digit(int model, const float *x, float *out) {
  ModelCInput_5c241121 in;
  for (int r = 0; r < 27; r++)
    for (int c = 0; c < 19; c++) in(r, c) = x[r * 19 + c];
  ModelCOutput_5c241121 o;
  if (model == 0) o = applyc_5c241121(in);
  else if (model == 1) o = applyc_01266c1b(in);
  else o = applyc_b00bf70c(in);
  for (int i = 0; i < 10; i++) out[i] = o(i, 0);
}

REF_API void ref_applym_slash(const float *x, float *out) {
  ModelMInput_730c4cbd in;
  for (int i = 0; i < 176; i++) in(i, 0) = x[i];
  ModelMOutput_730c4cbd o = applym_730c4cbd(in);
  for (int i = 0; i < 2; i++) out[i] = o(i, 0);
}

REF_API void ref_applyc_expiry(const float *x, float *out) {
  ModelCInput_bf4dd6c8 in;
  for (int r = 0; r < 16; r++)
    for (int c = 0; c < 11; c++) in(r, c) = x[r * 11 + c];
  ModelCOutput_bf4dd6c8 o = applyc_bf4dd6c8(in);
  for (int i = 0; i < 10; i++) out[i] = o(i, 0);
}

REF_API void ref_line_by_shifting_origin(float rho, float theta, int xo, int yo, float *rho_out,
                                         float *theta_out) {
  ParametricLine l;
  l.rho = rho;
  l.theta = theta;
  ParametricLine n = lineByShiftingOrigin(l, xo, yo);
  *rho_out = n.rho;
  *theta_out = n.theta;
}

REF_API int ref_parametric_intersect(float rho1, float theta1, float rho2, float theta2, float *x,
                                     float *y) {
  ParametricLine a, b;
  a.rho = rho1; a.theta = theta1;
  b.rho = rho2; b.theta = theta2;
  return parametricIntersect(a, b, x, y) ? 1 : 0;
}

REF_API void ref_calc_persp_transform(const float *src_pts, const float *dst_pts, float *m9) {
  dmz_point s[4], d[4];
  for (int i = 0; i < 4; i++) {
    s[i] = dmz_create_point(src_pts[2 * i], src_pts[2 * i + 1]);
    d[i] = dmz_create_point(dst_pts[2 * i], dst_pts[2 * i + 1]);
  }
  llcv_calc_persp_transform(m9, 9, true, s, d);
}

REF_API void ref_card_dest_points(float *dst_pts) {
  dmz_point d[4];
  dmz_rect_get_points(dmz_create_rect(0, 0, kCreditCardTargetWidth - 1, kCreditCardTargetHeight - 1), d);
  for (int i = 0; i < 4; i++) { dst_pts[2 * i] = d[i].x; dst_pts[2 * i + 1] = d[i].y; }
}

REF_API void ref_best_segmentation_for_vseg_scores(float *visa, float *amex, float *score,
                                                   int *y_offset, int *pattern) {
  NVerticalSegmentation best;
  best_segmentation_for_vseg_scores(visa, amex, &best);
  *score = best.score;
  *y_offset = best.y_offset;
  *pattern = best.pattern_type;
}

REF_API void ref_best_n_hseg_constrained(float *grad_sums, int pattern_type, float wmin, float wmax,
                                         float wstep, int omin, int omax, int ostep,
                                         uint16_t *offsets, float *score, float *number_width,
                                         int *pattern_offset) {
  NVerticalSegmentation vseg;
  vseg.pattern_type = (NumberPatternType)pattern_type;
  vseg.number_pattern_length = NumberPatternLengthForPatternType[pattern_type];
  memcpy(&vseg.number_pattern, NumberPatternForPatternType[pattern_type], sizeof(vseg.number_pattern));
  vseg.number_length = NumberLengthForNumberPatternType[pattern_type];
  NHorizontalSegmentation best;
  best.n_offsets = vseg.number_length;
  best.score = *score;
  best.number_width = *number_width;
  best.pattern_offset = (uint16_t)*pattern_offset;
  memcpy(best.offsets, offsets, sizeof(best.offsets));
  SliceF32 ws; ws.min = wmin; ws.max = wmax; ws.step = wstep;
  SliceU16 os; os.min = (uint16_t)omin; os.max = (uint16_t)omax; os.step = (uint16_t)ostep;
  best = best_n_hseg_constrained(grad_sums, vseg, best, ws, os);
  memcpy(offsets, best.offsets, sizeof(best.offsets));
  *score = best.score;
  *number_width = best.number_width;
  *pattern_offset = best.pattern_offset;
}

REF_API int ref_passes_luhn(uint8_t *digits, int n) {
  return dmz_passes_luhn_checksum(digits, (uint8_t)n) ? 1 : 0;
}

REF_API int ref_card_type(uint8_t *digits, int n, int allow_incomplete, int *number_length) {
  dmz_card_info info = dmz_card_info_for_prefix_and_length(digits, (uint8_t)n, allow_incomplete != 0);
  *number_length = info.number_length;
  return info.card_type;
}

// ---- dmz_olm.cpp:20-23,134-179 and processor_support.cpp:112-118 (the CYTHON flavour) ----
REF_API void ref_scale_point(const float *p /* x y */, const float *src /* x y w h */, const float *dst, float *out) {
  dmz_point q = dmz_scale_point(dmz_create_point(p[0], p[1]), dmz_create_rect(src[0], src[1], src[2], src[3]),
                                dmz_create_rect(dst[0], dst[1], dst[2], dst[3]));
  out[0] = q.x;
  out[1] = q.y;
}
REF_API void ref_guide_frame(int orientation, float preview_width, float preview_height, float *out /* x y w h */) {
  dmz_rect r = dmz_guide_frame((FrameOrientation)orientation, preview_width, preview_height);
  out[0] = r.x; out[1] = r.y; out[2] = r.w; out[3] = r.h;
}
REF_API int ref_opposite_orientation(int orientation) { return dmz_opposite_orientation((FrameOrientation)orientation); }
REF_API int ref_processor_support(int which) {
  return which == 0 ? dmz_has_neon_runtime() : which == 1 ? dmz_use_vfp3_16() : dmz_use_gles_warp();
}

// ---- expiry segmentation list logic (scan/expiry_seg.cpp): the two functions below touch the
// Scharr image only through the CV_IMAGE_ELEM header macro, so they link without OpenCV ----
REF_API int ref_expiry_gather_into_groups(int n_items, const int *lefts, const int64_t *sums, int top,
                                          int height, int *group_n, int *group_left, int *group_width,
                                          int *rect_left, int64_t *rect_sum) {
  GroupedRectsList items, groups;
  for (int i = 0; i < n_items; i++) {
    GroupedRects g;
    g.top = top;
    g.left = lefts[i];
    g.width = kSmallCharacterWidth;
    g.height = height;
    g.grouped_yet = false;
    g.sum = (long)sums[i];
    g.character_width = kSmallCharacterWidth;
    items.push_back(g);
  }
  gather_into_groups(groups, items, kSmallCharacterWidth);
  int k = 0;
  for (size_t i = 0; i < groups.size(); i++) {
    group_n[i] = (int)groups[i].character_rects.size();
    group_left[i] = groups[i].left;
    group_width[i] = groups[i].width;
    for (size_t j = 0; j < groups[i].character_rects.size(); j++, k++) {
      rect_left[k] = groups[i].character_rects[j].left;
      rect_sum[k] = groups[i].character_rects[j].sum;
    }
  }
  return (int)groups.size();
}

REF_API void ref_expiry_regrid_group(const int16_t *sobel, int top, int height, int *left, int *width,
                                     int *character_width, int *n, int *rect_left, int64_t *rect_sum) {
  IplImage img;
  memset(&img, 0, sizeof(img));
  img.nSize = sizeof(IplImage);
  img.nChannels = 1;
  img.depth = IPL_DEPTH_16S;
  img.width = kCreditCardTargetWidth;
  img.height = kCreditCardTargetHeight;
  img.widthStep = kCreditCardTargetWidth * 2;
  img.imageData = (char *)sobel;
  GroupedRects g;
  g.top = top;
  g.height = height;
  g.left = *left;
  g.width = *width;
  g.character_width = *character_width;
  g.grouped_yet = false;
  g.sum = 0;
  regrid_group(&img, g);
  *left = g.left;
  *width = g.width;
  *character_width = g.character_width;
  *n = (int)g.character_rects.size();
  for (size_t i = 0; i < g.character_rects.size(); i++) {
    rect_left[i] = g.character_rects[i].left;
    rect_sum[i] = g.character_rects[i].sum;
  }
}

// ---- the cross-frame half of expiry_extract (scan/expiry_categorize.cpp:162-376): pure C++/Eigen ----
REF_API int ref_expiry_session_replay(int n_frames, const int *groups_per_frame, const int16_t *tops,
                                      const int16_t *lefts, const float *scores /* [g][4][10] */,
                                      int *months_out, int *years_out, int *n_aggregated_out) {
  GroupedRectsList aggregated;
  int month = 0, year = 0, g = 0;
  for (int f = 0; f < n_frames; f++) {
    GroupedRectsList new_groups;
    for (int i = 0; i < groups_per_frame[f]; i++, g++) {
      GroupedRects gr;
      gr.top = tops[g];
      gr.left = lefts[g];
      gr.width = 5 * 13;
      gr.height = kSmallCharacterHeight;
      gr.grouped_yet = false;
      gr.sum = 0;
      gr.character_width = kTrimmedCharacterImageWidth;
      gr.pattern = ExpiryPatternMMsYY;
      gr.recently_seen_count = 0;
      gr.total_seen_count = 0;
      gr.scores.setZero();
      for (int c = 0; c < 5; c++) gr.character_rects.push_back(CharacterRect(tops[g], lefts[g] + 13 * c, 0));
      for (int row = 0; row < 4; row++)
        for (int k = 0; k < 10; k++) gr.scores(row < 2 ? row : row + 1, k) = scores[(size_t)g * 40 + row * 10 + k];
      new_groups.push_back(gr);
    }
    if (!new_groups.empty()) {
      expiry_aggregate_grouped_rects(aggregated, new_groups);
      for (GroupedRectsListIterator group = aggregated.begin(); group != aggregated.end(); ++group) {
        if (group->total_seen_count < 3) continue;
        get_stable_expiry_month_and_year(*group, &month, &year);
      }
    }
    months_out[f] = month;
    years_out[f] = year;
    n_aggregated_out[f] = (int)aggregated.size();
  }
  return 0;
}

// ---- session policy: the reference's own scanner_result (scan.cpp:88-194), expiry aggregation
// and dmz_olm code, driven by replayed per-frame records.  scanner_add_frame_with_expiry itself
// needs scan_card_image (OpenCV), so its state update -- scan.cpp:44-46 (what is still needed),
// :50-59 (upside-down / unusable frames are dropped), :62-66 (expiry_extract's aggregation half,
// expiry_categorize.cpp:351-375) and :69-84 (the decayed score sum) -- is applied here to the
// reference's ScannerState with the reference's types and operators. ----
struct RefSessionOut {
  int32_t complete, complete_frame, number_frame, n_numbers;
  uint8_t predictions[16];
  int32_t card_type, expiry_month, expiry_year, count15, count16, usable_frames, n_expiry_groups;
  int32_t vseg_y_offset, n_offsets;
  uint16_t offsets[16];
  float number_width;
  int32_t reserved[6];
};

REF_API void ref_scan_session(const uint8_t *frames /* n x 1024-byte records */,
                              const uint8_t *expiry /* n x 1592-byte records or NULL */, int n_frames,
                              int scan_expiry, RefSessionOut *out) {
  // field offsets of orc_frame_result / orc_expiry_result (oracle/dmz_oracle.h)
  enum { F_FLAGS = 84, F_YOFF = 92, F_NOFF = 100, F_OFFS = 104, F_NWIDTH = 140, F_SCORES = 168 };
  enum { X_NGROUPS = 0, X_CATEG = 48, X_GROUPS = 56, G_SIZE = 192, G_TOP = 0, G_LEFT = 2, G_CTOP = 8, G_CLEFT = 18, G_SCORES = 32 };
  ScannerState state;
  scanner_initialize(&state);
  memset(out, 0, sizeof(*out));
  out->complete_frame = -1;
  out->number_frame = -1;
  for (int f = 0; f < n_frames; f++) {
    const uint8_t *fr = frames + (size_t)f * 1024;
    int32_t flags, yoff, noff;
    memcpy(&flags, fr + F_FLAGS, 4);
    memcpy(&yoff, fr + F_YOFF, 4);
    memcpy(&noff, fr + F_NOFF, 4);
    const bool need_number = state.timeOfCardNumberCompletionInMilliseconds == 0;
    const bool need_expiry = scan_expiry && (state.expiry_month == 0 || state.expiry_year == 0);
    bool usable = false;
    if (!(flags & 2)) usable = need_number ? (flags & 1) != 0 : (flags & 4) != 0;
    if (usable) {
      out->usable_frames++;
      if (need_expiry) {
        state.scan_expiry = true;
        GroupedRectsList new_groups;
        if (expiry) {
          const uint8_t *x = expiry + (size_t)f * 1592;
          int32_t ng, categ;
          memcpy(&ng, x + X_NGROUPS, 4);
          memcpy(&categ, x + X_CATEG, 4);
          for (int g = 0; categ && g < ng; g++) {
            const uint8_t *gp = x + X_GROUPS + (size_t)g * G_SIZE;
            int16_t top, left, ct[5], cl[5];
            memcpy(&top, gp + G_TOP, 2);
            memcpy(&left, gp + G_LEFT, 2);
            memcpy(ct, gp + G_CTOP, 10);
            memcpy(cl, gp + G_CLEFT, 10);
            GroupedRects gr;
            gr.top = top;
            gr.left = left;
            gr.width = 0;
            gr.height = 0;
            gr.grouped_yet = false;
            gr.sum = 0;
            gr.character_width = kTrimmedCharacterImageWidth;
            gr.pattern = ExpiryPatternMMsYY;
            gr.recently_seen_count = 0;
            gr.total_seen_count = 0;
            gr.scores.setZero();
            for (int c = 0; c < 5; c++) gr.character_rects.push_back(CharacterRect(ct[c], cl[c], 0));
            float sc[40];
            memcpy(sc, gp + G_SCORES, sizeof(sc));
            for (int r = 0; r < 4; r++)
              for (int k = 0; k < 10; k++) gr.scores(r < 2 ? r : r + 1, k) = sc[r * 10 + k];
            new_groups.push_back(gr);
          }
        }
        if (!new_groups.empty()) {
          expiry_aggregate_grouped_rects(state.expiry_groups, new_groups);
          for (GroupedRectsListIterator group = state.expiry_groups.begin(); group != state.expiry_groups.end(); ++group) {
            if (group->total_seen_count < 3) continue;
            get_stable_expiry_month_and_year(*group, &state.expiry_month, &state.expiry_year);
          }
        }
      }
      if (need_number) {
        NHorizontalSegmentation hs;
        memset(&hs, 0, sizeof(hs));
        hs.n_offsets = (uint8_t)noff;
        memcpy(hs.offsets, fr + F_OFFS, 32);
        memcpy(&hs.number_width, fr + F_NWIDTH, 4);
        NVerticalSegmentation vs;
        memset(&vs, 0, sizeof(vs));
        vs.y_offset = (uint16_t)yoff;
        state.mostRecentUsableHSeg = hs;
        state.mostRecentUsableVSeg = vs;
        NumberScores scores;
        float sc[160];
        memcpy(sc, fr + F_SCORES, sizeof(sc));
        for (int i = 0; i < 16; i++)
          for (int k = 0; k < 10; k++) scores(i, k) = sc[i * 10 + k];
        if (noff == 15) {
          state.aggregated15 *= 0.8f;
          state.aggregated15 += scores * (1 - 0.8f);
          state.count15++;
        } else if (noff == 16) {
          state.aggregated16 *= 0.8f;
          state.aggregated16 += scores * (1 - 0.8f);
          state.count16++;
        }
      }
    }
    ScannerResult res;
    scanner_result(&state, &res);
    if (state.timeOfCardNumberCompletionInMilliseconds > 0 && out->number_frame < 0) out->number_frame = f;
    if (res.complete) {
      out->complete = 1;
      out->complete_frame = f;
      out->n_numbers = res.n_numbers;
      for (int i = 0; i < res.n_numbers; i++) out->predictions[i] = (uint8_t)res.predictions(i, 0);
      out->card_type = dmz_card_info_for_prefix_and_length(out->predictions, res.n_numbers, false).card_type;
      out->expiry_month = res.expiry_month;
      out->expiry_year = res.expiry_year;
      out->vseg_y_offset = res.vseg.y_offset;
      out->n_offsets = res.hseg.n_offsets;
      memcpy(out->offsets, res.hseg.offsets, 32);
      out->number_width = res.hseg.number_width;
      break;
    }
  }
  if (!out->complete) {
    if (state.timeOfCardNumberCompletionInMilliseconds > 0) {
      const ScannerResult &s = state.successfulCardNumberResult;
      out->n_numbers = s.n_numbers;
      for (int i = 0; i < s.n_numbers; i++) out->predictions[i] = (uint8_t)s.predictions(i, 0);
      out->card_type = dmz_card_info_for_prefix_and_length(out->predictions, s.n_numbers, false).card_type;
      out->vseg_y_offset = s.vseg.y_offset;
      out->n_offsets = s.hseg.n_offsets;
      memcpy(out->offsets, s.hseg.offsets, 32);
      out->number_width = s.hseg.number_width;
    }
    out->expiry_month = state.expiry_month;
    out->expiry_year = state.expiry_year;
  }
  out->count15 = state.count15;
  out->count16 = state.count16;
  out->n_expiry_groups = (int)state.expiry_groups.size();
}

// ---- dmz_deinterleave_RGBA_to_R (dmz.cpp:62-105): plain C, no OpenCV symbol ----
REF_API void ref_deinterleave_rgba_to_r(uint8_t *source, uint8_t *dest, int size) {
  dmz_deinterleave_RGBA_to_R(source, dest, size);
}

// ---- dmz_card_rect_for_screen (dmz.cpp:138-165): header-only CvRect / CvSize ----
REF_API void ref_card_rect_for_screen(int card_w, int card_h, int std_w, int std_h, int act_w, int act_h, int *rect) {
  CvRect r = dmz_card_rect_for_screen(cvSize(card_w, card_h), cvSize(std_w, std_h), cvSize(act_w, act_h));
  rect[0] = r.x;
  rect[1] = r.y;
  rect[2] = r.width;
  rect[3] = r.height;
}

// ---- the two candidate orders of the expiry segmentation (expiry_seg.cpp:456, :842): the reference's own
// element types and comparators (expiry_seg.cpp:69-87, expiry_types.h:51-66) through this toolchain's
// std::sort, exactly as the unity TU instantiates it.  order_out[k] = original index of the k-th visited
// element.  Pins orc_expiry.c's restatement of libstdc++'s introsort permutation.
REF_API void ref_sort_rect_sums(int n, const long *sums, int *order_out) {
  CharacterRectList rect_list;
  for (int i = 0; i < n; i++) {
    CharacterRect rect;
    rect.top = 0;
    rect.left = i;
    rect.sum = sums[i];
    rect_list.push_back(rect);
  }
  std::sort(rect_list.begin(), rect_list.end(), CharacterRectCompareSumDescending());
  for (int i = 0; i < n; i++) order_out[i] = rect_list[i].left;
}

REF_API void ref_sort_stripe_sums(int n, const long *sums, int *order_out) {
  std::vector<StripeSum> stripe_sums;
  for (int i = 0; i < n; i++) {
    StripeSum stripe_sum;
    stripe_sum.base_row = i;
    stripe_sum.sum = sums[i];
    stripe_sums.push_back(stripe_sum);
  }
  std::sort(stripe_sums.begin(), stripe_sums.end(), StripeSumCompareDescending());
  for (int i = 0; i < n; i++) order_out[i] = stripe_sums[i].base_row;
}
