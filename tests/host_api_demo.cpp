// Drives the reference-style C++ entry points (card.io-dmz_amd/host/dmz.h) the way an SDK
// call site would: detect -> transform -> scanner_add_frame x N -> scanner_result.
// usage: host_api_demo <frames.raw (n x 640x480 u8)> <n>   -> one line of text per frame + a summary
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "dmz.h"

int main(int argc, char **argv) {
  if (argc < 3) return 2;
  const int n = atoi(argv[2]);
  std::vector<unsigned char> buf((size_t)n * 640 * 480);
  FILE *f = fopen(argv[1], "rb");
  if (!f || fread(buf.data(), 1, buf.size(), f) != buf.size()) return 3;
  fclose(f);
  if (!dmz_has_hip_runtime()) { printf("NOGPU\n"); return 4; }
  dmz_context *dmz = dmz_context_create();
  if (!dmz) return 5;
  ScannerState state;
  scanner_initialize(&state);
  state.dmz = dmz;
  for (int i = 0; i < n; i++) {
    IplImage y;
    memset(&y, 0, sizeof(y));
    y.nSize = sizeof(y); y.nChannels = 1; y.depth = IPL_DEPTH_8U; y.width = 640; y.height = 480;
    y.widthStep = 640; y.imageData = (char *)buf.data() + (size_t)i * 640 * 480; y.imageSize = 640 * 480;
    dmz_edges edges;
    dmz_corner_points corners;
    memset(&corners, 0, sizeof(corners));
    const bool found = dmz_detect_edges(&y, NULL, NULL, FrameOrientationLandscapeRight, &edges, &corners);
    printf("frame %d found %d edges %d%d%d%d tl %.9g %.9g br %.9g %.9g", i, found, edges.top.found,
           edges.left.found, edges.bottom.found, edges.right.found, corners.top_left.x, corners.top_left.y,
           corners.bottom_right.x, corners.bottom_right.y);
    if (found) {
      IplImage *card = NULL;
      dmz_transform_card(dmz, &y, corners, FrameOrientationLandscapeRight, false, &card);
      unsigned long sum = 0;
      for (int k = 0; k < 428 * 270; k++) sum += (unsigned char)card->imageData[k] * (unsigned long)(k % 251 + 1);
      FrameScanResult fr = FrameScanResult();
      scanner_add_frame_with_expiry(&state, card, true, &fr);
      printf(" cardsum %lu usable %d upside %d y_offset %d vscore %.6f digits ", sum, fr.usable, fr.upside_down,
             fr.vseg.y_offset, fr.vseg.score);
      for (int d = 0; d < fr.hseg.n_offsets; d++) {
        int best = 0;
        for (int k = 1; k < 10; k++) if (fr.scores.v[d][k] > fr.scores.v[d][best]) best = k;
        printf("%d", best);
      }
      printf(" expiry_groups %d", (int)fr.expiry_groups.size());
      for (size_t g = 0; g < fr.expiry_groups.size(); g++) {
        const GroupedRects &gr = fr.expiry_groups[g];
        printf(" [%d %d %d %d :", gr.top, gr.left, gr.width, gr.height);
        for (size_t c = 0; c < gr.character_rects.size(); c++) printf(" %d,%d", gr.character_rects[c].left, gr.character_rects[c].top);
        printf("]");
      }
      dmz_release_image(&card);
    }
    printf("\n");
  }
  ScannerResult res;
  scanner_result(&state, &res);
  printf("session count15 %d count16 %d complete %d\n", state.count15, state.count16, res.complete);
  printf("expiry scan_expiry %d aggregated %d seen", state.scan_expiry, (int)state.expiry_groups.size());
  for (size_t g = 0; g < state.expiry_groups.size(); g++) printf(" %d", state.expiry_groups[g].total_seen_count);
  printf(" month %d year %d\n", state.expiry_month, state.expiry_year);
  {  // the Cython flavour's entry (scan/frame.cpp:84-98) on the first frame's card: the call configs[0]'s sequence ends in
    IplImage y;
    memset(&y, 0, sizeof(y));
    y.nSize = sizeof(y); y.nChannels = 1; y.depth = IPL_DEPTH_8U; y.width = 640; y.height = 480;
    y.widthStep = 640; y.imageData = (char *)buf.data(); y.imageSize = 640 * 480;
    dmz_edges edges;
    dmz_corner_points corners;
    memset(&corners, 0, sizeof(corners));
    CythonFrameScanResult cr;
    memset(&cr, 0, sizeof(cr));
    if (dmz_detect_edges(&y, NULL, NULL, FrameOrientationLandscapeRight, &edges, &corners)) {
      IplImage *card = NULL;
      dmz_transform_card(dmz, &y, corners, FrameOrientationLandscapeRight, false, &card);
      cython_scan_card_image(card, &cr);
      dmz_release_image(&card);
    }
    printf("cython usable %d y_offset %d pattern %d n_offsets %d offsets", cr.usable, cr.vseg.y_offset, (int)cr.vseg.pattern_type,
           cr.hseg.n_offsets);
    for (int d = 0; d < cr.hseg.n_offsets; d++) printf(" %d", cr.hseg.offsets[d]);
    printf("\n");
    // the Cython flavour's expiry calls (dmz.h:105-119) on the same card, three times like three frames of a session:
    // dmz_best_expiry_seg, then dmz_expiry_extract with the groups it returned; and the Scharr image through py_mz_* headers
    IplImage *card = NULL;
    if (dmz_detect_edges(&y, NULL, NULL, FrameOrientationLandscapeRight, &edges, &corners))
      dmz_transform_card(dmz, &y, corners, FrameOrientationLandscapeRight, false, &card);
    if (card) {
      uint16_t n_session = 0;
      CythonGroupedRects *session = NULL;
      int month = 0, year = 0;
      for (int rep = 0; rep < 3; rep++) {
        uint16_t n_new = 0;
        CythonGroupedRects *fresh = NULL;
        dmz_best_expiry_seg(card, cr.vseg.y_offset, &fresh, &n_new);
        if (rep == 0) {
          printf("cyseg %d", n_new);
          for (int g = 0; g < n_new; g++) {
            printf(" [%d %d %d %d :", fresh[g].top, fresh[g].left, fresh[g].width, fresh[g].height);
            for (int c = 0; c < fresh[g].number_of_character_rects; c++) printf(" %d,%d", fresh[g].character_rects[c].left, fresh[g].character_rects[c].top);
            printf("]");
          }
          printf("\n");
        }
        dmz_expiry_extract(card, &n_session, &session, &n_new, &fresh, &month, &year);
        if (rep == 0) {
          // (as in the reference, dmz.cpp:625-655, the categorised scores come back in the SESSION's groups: the new-groups
          // array is left untouched; an empty session takes the frame's groups in their order)
          printf("cycat %d", n_session);
          for (int g = 0; g < n_session; g++)
            for (int ch = 0; ch < 5; ch++) {
              if (ch == 2) continue;
              int best = 0;
              for (int d = 1; d < 10; d++) if (session[g].scores[ch][d] > session[g].scores[ch][best]) best = d;
              printf(" %d:%.6f", best, session[g].scores[ch][best]);
            }
          printf("\n");
        }
        for (int g = 0; g < n_new; g++) free(fresh[g].character_rects);
        free(fresh);
      }
      printf("cysession groups %d month %d year %d\n", n_session, month, year);
      for (int g = 0; g < n_session; g++) free(session[g].character_rects);
      free(session);
      // dmz_scharr3_dx_abs on rows 180..269 of the card, through image headers made by the py_mz_* helpers
      std::vector<int16_t> sch((size_t)428 * 90);
      IplImage *src = py_mz_create_from_cv_image_data(card->imageData, card->imageSize, 428, 270, IPL_DEPTH_8U, 1, 0, 180, 428, 90);
      IplImage *dst = py_mz_create_from_cv_image_data((char *)sch.data(), 428 * 90 * 2, 428, 90, IPL_DEPTH_16S, 1, 0, 0, 428, 90);
      dmz_scharr3_dx_abs(src, dst);
      long long ssum = 0, wsum = 0;
      for (size_t i = 0; i < sch.size(); i++) ssum += sch[i], wsum += (long long)sch[i] * (long long)(i % 997 + 1);
      printf("cyscharr %lld %lld\n", ssum, wsum);
      py_mz_release_ipl_image(src);
      py_mz_release_ipl_image(dst);
      dmz_release_image(&card);
    }
  }
  float m[9];
  dmz_point s[4] = {{106, 105}, {533, 105}, {106, 374}, {533, 374}}, d[4];
  dmz_rect_get_points(dmz_create_rect(0, 0, 427, 269), d);
  llcv_calc_persp_transform(m, 9, true, s, d);
  printf("persp %.9g %.9g %.9g\n", m[0], m[2], m[5]);
  scanner_destroy(&state);
  dmz_context_destroy(dmz);
  return 0;
}
