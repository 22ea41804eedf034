/*
 * orc_expiry.c -- CPU ORACLE (test infrastructure, see dmz_oracle.h) for the expiry path:
 * scan/expiry_seg.cpp (segmentation of the MM/YY group) and the per-frame half of
 * scan/expiry_categorize.cpp (digit scores of each group).
 *
 * Pinning: the two models (slash MLP 730c4cbd, digit CNN bf4dd6c8) are pinned by the
 * reference's embedded known-answer vectors (tests/test_oracle_kats.py).  Everything else in
 * this file is "parity unpinned": the functions need OpenCV library symbols to run
 * (cvGetSize, cvSum, cvNormalize, cvThreshold, cvMorphologyEx, cvSmooth) and the reference
 * holds no fixture for them; they are restated from the in-tree source and the published
 * OpenCV 2.4.9 semantics.
 *
 * Candidate order: the reference orders the window sums and the stripe sums with std::sort and a
 * "sum >" comparator (expiry_seg.cpp:75-87, 456, 842).  std::sort is not stable, so the order of
 * equal sums is whatever the standard library's algorithm leaves -- and it decides which of two
 * overlapping windows / stripes with equal sums the greedy picks below take.  The reference's
 * standard library here is libstdc++ (GCC 11.4); its std::sort is a deterministic function of the
 * comparison results, restated in orc_sort_order_desc() below: introsort loop (ranges > 16:
 * median of first+1 / middle / last-1 swapped to first, unguarded Hoare partition around it, the
 * right part recursed, depth limit 2*floor(log2 n), heap sort of a range once it is reached),
 * then one insertion sort over everything.  Pinned against the reference's own types and
 * comparators compiled with this toolchain (oracle/_ref: ref_sort_rect_sums /
 * ref_sort_stripe_sums; tests/test_oracle_vs_ref.py: random lists with ties, adversarial lists
 * that reach the heap sort, and every list of a corpus sample).
 */
#include "dmz_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define W ORC_CARD_W
#define H ORC_CARD_H
#define kSmallCharacterWidth 9
#define kSmallCharacterHeight 15
#define kTrimmedW 11
#define kTrimmedH 16
#define kMinimumExpiryStripCharacters 5
#define MAX_RECTS 96

typedef struct {
  int top, left;
  long sum;
} char_rect;

typedef struct {
  int top, left, width, height;
  int grouped_yet;
  long sum;
  int character_width;
  int n;
  char_rect rects[MAX_RECTS];
} grouped_rects;

static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }

/* ---- libstdc++'s std::sort(first, last, comp) with comp(a, b) = key[a] > key[b], as a permutation:
 * v[] holds element indices and is rearranged exactly as the library rearranges the elements
 * (bits/stl_algo.h of GCC 11: __introsort_loop, __move_median_to_first, __unguarded_partition,
 * __final_insertion_sort; bits/stl_heap.h: __make_heap / __adjust_heap / __push_heap / __pop_heap for the
 * depth-limit fallback).  The algorithm is a published one; the code below is a restatement, the
 * pin is oracle/_ref (the reference's own instantiation). ---- */
typedef struct {
  const long *key;
  int *v;
} sort_ctx;

#define SORT_GT(c, a, b) ((c)->key[(a)] > (c)->key[(b)]) /* the comparator on ELEMENTS a, b */

static void sort_swap(sort_ctx *c, int i, int j) {
  int t = c->v[i];
  c->v[i] = c->v[j];
  c->v[j] = t;
}

/* heap helpers on v[first .. first+len) */
static void sort_push_heap(sort_ctx *c, int first, int hole, int top, int value) {
  int parent = (hole - 1) / 2;
  while (hole > top && SORT_GT(c, c->v[first + parent], value)) {
    c->v[first + hole] = c->v[first + parent];
    hole = parent;
    parent = (hole - 1) / 2;
  }
  c->v[first + hole] = value;
}

static void sort_adjust_heap(sort_ctx *c, int first, int hole, int len, int value) {
  const int top = hole;
  int child = hole;
  while (child < (len - 1) / 2) {
    child = 2 * (child + 1);
    if (SORT_GT(c, c->v[first + child], c->v[first + child - 1])) child--;
    c->v[first + hole] = c->v[first + child];
    hole = child;
  }
  if ((len & 1) == 0 && child == (len - 2) / 2) {
    child = 2 * (child + 1);
    c->v[first + hole] = c->v[first + child - 1];
    hole = child - 1;
  }
  sort_push_heap(c, first, hole, top, value);
}

static __thread int heap_sorts; /* test hook: how often the depth limit was reached */
int orc_sort_heap_sorts(void) { return heap_sorts; }

static void sort_heap_range(sort_ctx *c, int first, int last) { /* __partial_sort(first, last, last) */
  int len = last - first;
  heap_sorts++;
  if (len >= 2)
    for (int parent = (len - 2) / 2;; parent--) {
      sort_adjust_heap(c, first, parent, len, c->v[first + parent]);
      if (parent == 0) break;
    }
  while (last - first > 1) {
    --last;
    int value = c->v[last];
    c->v[last] = c->v[first];
    sort_adjust_heap(c, first, 0, last - first, value);
  }
}

static void sort_introsort_loop(sort_ctx *c, int first, int last, int depth_limit) {
  while (last - first > 16) {
    if (depth_limit == 0) {
      sort_heap_range(c, first, last);
      return;
    }
    --depth_limit;
    /* __move_median_to_first(first, first + 1, mid, last - 1) */
    int a = first + 1, b = first + (last - first) / 2, cc = last - 1;
    int ea = c->v[a], eb = c->v[b], ec = c->v[cc];
    if (SORT_GT(c, ea, eb)) {
      if (SORT_GT(c, eb, ec)) sort_swap(c, first, b);
      else if (SORT_GT(c, ea, ec)) sort_swap(c, first, cc);
      else sort_swap(c, first, a);
    } else if (SORT_GT(c, ea, ec)) sort_swap(c, first, a);
    else if (SORT_GT(c, eb, ec)) sort_swap(c, first, cc);
    else sort_swap(c, first, b);
    /* __unguarded_partition(first + 1, last, pivot = first) */
    int lo = first + 1, hi = last;
    for (;;) {
      while (SORT_GT(c, c->v[lo], c->v[first])) ++lo;
      --hi;
      while (SORT_GT(c, c->v[first], c->v[hi])) --hi;
      if (!(lo < hi)) break;
      sort_swap(c, lo, hi);
      ++lo;
    }
    sort_introsort_loop(c, lo, last, depth_limit);
    last = lo;
  }
}

static void sort_linear_insert(sort_ctx *c, int last) { /* __unguarded_linear_insert */
  int value = c->v[last];
  int next = last - 1;
  while (SORT_GT(c, value, c->v[next])) {
    c->v[last] = c->v[next];
    last = next;
    --next;
  }
  c->v[last] = value;
}

static void sort_insertion(sort_ctx *c, int first, int last) { /* __insertion_sort */
  if (first == last) return;
  for (int i = first + 1; i != last; ++i) {
    if (SORT_GT(c, c->v[i], c->v[first])) {
      int value = c->v[i];
      memmove(&c->v[first + 1], &c->v[first], sizeof(int) * (size_t)(i - first));
      c->v[first] = value;
    } else
      sort_linear_insert(c, i);
  }
}

/* test hook: when armed, every list handed to the sort below is appended to cap_buf as
 * [n, key_0 .. key_{n-1}] (tests/test_oracle_vs_ref.py runs the corpus' own lists through the
 * reference's std::sort) */
static __thread int64_t *cap_buf;
static __thread int cap_len, cap_used;
void orc_expiry_capture_sort_lists(int64_t *buf, int len) { cap_buf = buf, cap_len = len, cap_used = 0; }
int orc_expiry_captured_len(void) { return cap_used; }

/* order[k] = index (into key[]) of the k-th element of the list after
 * std::sort(list.begin(), list.end(), <sum descending>) */
void orc_sort_order_desc(int n, const long *key, int *order) {
  sort_ctx c = {key, order};
  if (cap_buf && cap_used + n + 1 <= cap_len) {
    cap_buf[cap_used++] = n;
    for (int i = 0; i < n; i++) cap_buf[cap_used++] = key[i];
  }
  for (int i = 0; i < n; i++) order[i] = i;
  if (n == 0) return;
  int lg = 0;
  for (unsigned m = (unsigned)n; m > 1; m >>= 1) lg++;
  sort_introsort_loop(&c, 0, n, 2 * lg);
  if (n > 16) {
    sort_insertion(&c, 0, 16);
    for (int i = 16; i != n; ++i) sort_linear_insert(&c, i);
  } else
    sort_insertion(&c, 0, n);
}

/* sobel.cpp:706-804, scalar branch: |right-left| with the column index clamped, then
 * 3/10/3 down the column with the row index clamped -- both clamps at the ROI edge. */
void orc_scharr3_dx_abs(const uint8_t *src, int stride, int w, int h, int16_t *dst, int dstride) {
  int16_t *inter = (int16_t *)malloc(sizeof(int16_t) * (size_t)w * h);
  for (int r = 0; r < h; r++) {
    const uint8_t *row = src + (size_t)r * stride;
    for (int c = 0; c < w; c++) {
      int cl = c == 0 ? 0 : c - 1, cr = c == w - 1 ? w - 1 : c + 1;
      inter[(size_t)r * w + c] = (int16_t)abs(row[cr] - row[cl]);
    }
  }
  for (int c = 0; c < w; c++)
    for (int r = 0; r < h; r++) {
      int rt = r == 0 ? 0 : r - 1, rb = r == h - 1 ? h - 1 : r + 1;
      dst[(size_t)r * dstride + c] = (int16_t)(3 * (inter[(size_t)rt * w + c] + inter[(size_t)rb * w + c]) +
                                               10 * inter[(size_t)r * w + c]);
    }
  free(inter);
}

/* expiry_seg.cpp:101-129 (recursion unrolled into a loop) */
static void strip_group_white_space(grouped_rects *g) {
  while (g->n > 5) {
    int white_space_found = 0;
    int index = (g->n - 4) / 2;
    long threshold_sum = (long)(((g->rects[index + 0].sum + g->rects[index + 1].sum +
                                  g->rects[index + 2].sum + g->rects[index + 3].sum) / 4) * 0.8);
    if (g->rects[0].sum < threshold_sum) {
      memmove(&g->rects[0], &g->rects[1], sizeof(char_rect) * (size_t)(g->n - 1));
      g->n--;
      g->left = g->rects[0].left;
      white_space_found = 1;
    } else if (g->rects[g->n - 1].sum < threshold_sum) {
      g->n--;
      white_space_found = 1;
    }
    if (!white_space_found) break;
    g->width = g->rects[g->n - 1].left + g->character_width - g->left;
  }
}

/* expiry_seg.cpp:169-229 */
static void regrid_group(const int16_t *sobel, grouped_rects *g) {
  int best_grid_spacing = 0, best_starting_col_offset = 0;
  float best_ratio = FLT_MAX; /* MAXFLOAT */
  int bounds_left = imax(g->left - 2 * kSmallCharacterWidth, 0);
  int bounds_right = imin(g->left + g->width + 2 * kSmallCharacterWidth, W);
  int bounds_width = bounds_right - bounds_left;
  int min_lines = (int)(floorf((float)bounds_width / (float)11));
  long group_sum = 0;
  long col_sums[W];
  for (int col = bounds_left; col < bounds_right; col++) {
    long col_sum = 0;
    for (int row = g->top; row < g->top + g->height; row++) col_sum += sobel[(size_t)row * W + col];
    col_sums[col - bounds_left] = col_sum;
    group_sum += col_sum;
  }
  for (int spacing = 11; spacing <= 15; spacing++)
    for (int start = 0; start < spacing; start++) {
      float grid_line_sum = 0.0f;
      int n_lines = 0;
      for (int off = start; off < bounds_width; off += spacing) {
        n_lines += 1;
        grid_line_sum += (float)col_sums[off];
      }
      float average = grid_line_sum / (float)n_lines;
      grid_line_sum = average * (float)min_lines;
      float ratio = grid_line_sum / ((float)group_sum - grid_line_sum);
      if (ratio < best_ratio) {
        best_ratio = ratio;
        best_grid_spacing = spacing;
        best_starting_col_offset = start;
      }
    }
  /* every ratio NaN/inf-degenerate (blank group): the reference would loop forever with
   * spacing 0; a blank group cannot reach this point (its rects passed a > threshold test),
   * keep the search's first candidate to stay finite */
  if (best_grid_spacing == 0) best_grid_spacing = 11;
  int n = 0;
  for (int off = best_starting_col_offset; off + 1 < bounds_width; off += best_grid_spacing) {
    long sum = 0;
    for (int col = off + 1; col < imin(off + best_grid_spacing, bounds_width); col++) sum += col_sums[col];
    if (n < MAX_RECTS) {
      g->rects[n].top = g->top;
      g->rects[n].left = bounds_left + off + 1;
      g->rects[n].sum = sum;
      n++;
    }
  }
  g->n = n;
  g->character_width = best_grid_spacing - 1;
  g->left = g->rects[0].left;
  g->width = g->rects[n - 1].left + g->character_width - g->left;
  strip_group_white_space(g);
}

/* expiry_seg.cpp:231-339 */
static void optimize_character_rects(const int16_t *sobel, grouped_rects *g) {
  int ciw = g->character_width + 4, cih = g->height + 4;
  for (int ri = g->n - 1; ri >= 0; ri--) {
    int rect_left = g->rects[ri].left - 2, rect_top = g->top - 2;
    if (rect_left < 0 || rect_left + ciw > W || rect_top + cih > H) {
      memmove(&g->rects[ri], &g->rects[ri + 1], sizeof(char_rect) * (size_t)(g->n - 1 - ri));
      g->n--;
      continue;
    }
    int16_t img[64][64];
    /* cvNormalize(.., 255, 0, CV_C): scale = 255/max|x| in double, applied by
     * cvtScale_<short,short,float> (float multiply, round-to-nearest-even, saturate) */
    int maxabs = 0;
    for (int r = 0; r < cih; r++)
      for (int c = 0; c < ciw; c++) {
        int v = sobel[(size_t)(rect_top + r) * W + rect_left + c];
        img[r][c] = (int16_t)v;
        if (abs(v) > maxabs) maxabs = abs(v);
      }
    double scale_d = (double)maxabs > DBL_EPSILON ? 255.0 / (double)maxabs : 0.0;
    float scale = (float)scale_d;
    for (int r = 0; r < cih; r++)
      for (int c = 0; c < ciw; c++) {
        float f = (float)img[r][c] * scale + 0.0f;
        long iv = lrint((double)f);
        if (iv > 32767) iv = 32767;
        if (iv < -32768) iv = -32768;
        /* cvThreshold(.., 100, 255, CV_THRESH_TOZERO) */
        img[r][c] = (int16_t)(iv > 100 ? iv : 0);
      }
    int cw = ciw, ch = cih;
    int col_sums[64], row_sums[64];
    int left_col = 0, right_col = cw - 1, top_row = 0, bottom_row = ch - 1;
    for (int c = left_col; c <= right_col; c++) {
      col_sums[c] = 0;
      for (int r = top_row; r <= bottom_row; r++) col_sums[c] += img[r][c];
    }
    while (cw > kTrimmedW) {
      if (col_sums[left_col] <= col_sums[right_col]) left_col++;
      else right_col--;
      cw--;
    }
    for (int r = top_row; r <= bottom_row; r++) {
      row_sums[r] = 0;
      for (int c = left_col; c <= right_col; c++) row_sums[r] += img[r][c];
    }
    while (ch > kTrimmedH) {
      if (row_sums[top_row] <= row_sums[bottom_row]) top_row++;
      else bottom_row--;
      ch--;
    }
    g->rects[ri].left = rect_left + left_col;
    g->rects[ri].top = rect_top + top_row;
  }
  if (g->n > 0) {
    int highest_top = H, lowest_top = 0;
    for (int i = 0; i < g->n; i++) {
      highest_top = imin(highest_top, g->rects[i].top);
      lowest_top = imax(lowest_top, g->rects[i].top);
    }
    g->character_width = kTrimmedW;
    g->left = g->rects[0].left;
    g->width = g->rects[g->n - 1].left + kTrimmedW - g->left;
    g->top = highest_top;
    g->height = lowest_top + kTrimmedH - g->top;
  }
}

/* expiry_seg.cpp:31-58: 11x16 ROI of the int16 Scharr image x (1/255) -> slash MLP, P0 > 0.7 */
static int is_slash(const int16_t *sobel, const char_rect *rect) {
  float x[176], p[2];
  const float s = 1.0f / 255.0f;
  for (int r = 0; r < kTrimmedH; r++)
    for (int c = 0; c < kTrimmedW; c++)
      x[r * kTrimmedW + c] = (float)sobel[(size_t)(rect->top + r) * W + rect->left + c] * s;
  orc_applym_slash(x, p);
  return p[0] > 0.7f;
}

static void emit_group(orc_expiry_result *out, const grouped_rects *g, int first, int stripe_base_row) {
  int top = g->rects[first].top, left = g->rects[first].left;
  int width = kSmallCharacterWidth, height = kSmallCharacterHeight;
  for (int i = 0; i < 5; i++) {
    const char_rect *cr = &g->rects[first + i];
    int former_bottom = top + height;
    top = imin(cr->top, top);
    width = (cr->left + kSmallCharacterWidth) - left;
    height = imax(cr->top + kSmallCharacterHeight, former_bottom) - top;
  }
  if (out->n_found < ORC_EXPIRY_MAX_GROUPS) {
    orc_expiry_group *o = &out->groups[out->n_found];
    memset(o, 0, sizeof(*o));
    o->top = (int16_t)top;
    o->left = (int16_t)left;
    o->width = (int16_t)width;
    o->height = (int16_t)height;
    o->stripe_base_row = (int16_t)stripe_base_row;
    for (int i = 0; i < 5; i++) {
      o->char_top[i] = (int16_t)g->rects[first + i].top;
      o->char_left[i] = (int16_t)g->rects[first + i].left;
    }
    out->n_groups = out->n_found + 1;
  }
  out->n_found++;
}

/* expiry_seg.cpp:131-167: sort by left (lefts are distinct here, so any sort gives this
 * order), chain items while the gap to the running group is < kSmallCharacterWidth, then strip
 * leading/trailing white space from every group */
static int gather_into_groups(grouped_rects *items, int n_items, grouped_rects *groups) {
  for (int i = 1; i < n_items; i++) {
    grouped_rects t = items[i];
    int j = i - 1;
    while (j >= 0 && items[j].left > t.left) {
      items[j + 1] = items[j];
      j--;
    }
    items[j + 1] = t;
  }
  int n_groups = 0;
  for (int bi = 0; bi < n_items; bi++) {
    grouped_rects *base = &items[bi];
    if (base->grouped_yet) continue;
    grouped_rects *g = &groups[n_groups++];
    *g = *base;
    g->sum = base->sum;
    g->n = 0;
    g->rects[g->n].top = base->top, g->rects[g->n].left = base->left, g->rects[g->n].sum = base->sum, g->n++;
    base->grouped_yet = 1;
    for (int i = bi + 1; i < n_items; i++) {
      grouped_rects *it = &items[i];
      if (it->left - (g->left + g->width) >= kSmallCharacterWidth) break;
      if (!it->grouped_yet) {
        it->grouped_yet = 1;
        int former_bottom = g->top + g->height;
        g->top = imin(g->top, it->top);
        g->width = it->left + it->width - base->left;
        g->height = imax(former_bottom, it->top + it->height) - g->top;
        g->sum += it->sum;
        g->rects[g->n].top = it->top, g->rects[g->n].left = it->left, g->rects[g->n].sum = it->sum, g->n++;
      }
    }
  }
  for (int i = 0; i < n_groups; i++) strip_group_white_space(&groups[i]);
  return n_groups;
}

/* expiry_seg.cpp:437-704 */
static void find_character_groups_for_stripe(const int16_t *sobel, int stripe_base_row, long stripe_sum,
                                             orc_expiry_result *out) {
  const int expanded_top = stripe_base_row - 1;
  const int expanded_h = imin(kSmallCharacterHeight + 2, H - expanded_top);
  long rect_average = (stripe_sum * kSmallCharacterWidth) / W;
  float summation_threshold = (float)(rect_average / 5);

  /* [1] sliding 9-wide sums of rows stripe_base_row .. stripe_base_row + expanded_h - 1 */
  static __thread char_rect rect_list[W]; /* (thread-local: the parity tests call the oracle from several threads) */
  int n_rects = 0;
  float rect_sum_total = 0;
  long rect_sum = 0;
  for (int col = 0; col < kSmallCharacterWidth; col++)
    for (int row = 0; row < expanded_h; row++) rect_sum += sobel[(size_t)(stripe_base_row + row) * W + col];
  for (int col = 0; col < W - kSmallCharacterWidth + 1; col++) {
    if ((float)rect_sum > summation_threshold) {
      rect_list[n_rects].top = expanded_top;
      rect_list[n_rects].left = col;
      rect_list[n_rects].sum = rect_sum;
      n_rects++;
      rect_sum_total += (float)rect_sum;
    }
    if (col < W - kSmallCharacterWidth)
      for (int row = 0; row < expanded_h; row++) {
        rect_sum -= sobel[(size_t)(stripe_base_row + row) * W + col];
        rect_sum += sobel[(size_t)(stripe_base_row + row) * W + col + kSmallCharacterWidth];
      }
  }
  if (n_rects == 0) return;
  float rect_sum_average = rect_sum_total / (float)n_rects;
  float rect_sum_threshold = (float)(0.8 * rect_sum_average);

  /* [2] std::sort descending by sum -- the library's own permutation, ties included (see above);
   * [3] greedy non-overlapping pick in that order */
  static __thread grouped_rects items[W / kSmallCharacterWidth + 2];
  static __thread long sort_keys[W];
  static __thread int sort_order[W];
  int n_items = 0;
  uint8_t mask[W + 16];
  memset(mask, 0, sizeof(mask));
  for (int i = 0; i < n_rects; i++) sort_keys[i] = rect_list[i].sum;
  orc_sort_order_desc(n_rects, sort_keys, sort_order);
  for (int k = 0; k < n_rects; k++) {
    const char_rect *r = &rect_list[sort_order[k]];
    if ((float)r->sum <= rect_sum_threshold) break;
    if (!mask[r->left] && !mask[r->left + kSmallCharacterWidth - 1]) {
      grouped_rects *it = &items[n_items++];
      it->top = r->top;
      it->left = r->left;
      it->width = kSmallCharacterWidth;
      it->height = expanded_h;
      it->grouped_yet = 0;
      it->sum = r->sum;
      it->character_width = kSmallCharacterWidth;
      it->n = 0;
      for (int k2 = 0; k2 < kSmallCharacterWidth; k2++) mask[r->left + k2] = 1;
    }
  }

  /* [4] local groups */
  static __thread grouped_rects groups[W / kSmallCharacterWidth + 2];
  int n_groups = gather_into_groups(items, n_items, groups);

  /* expiry_seg.cpp:566-573: keep groups of >= 4, regrid, optimise (dropping emptied groups),
   * keep groups of >= 5 (617-623) */
  int m = 0;
  for (int i = 0; i < n_groups; i++)
    if (groups[i].n >= kMinimumExpiryStripCharacters - 1) {
      if (m != i) groups[m] = groups[i];
      m++;
    }
  n_groups = m;
  for (int i = 0; i < n_groups; i++) regrid_group(sobel, &groups[i]);
  for (int i = n_groups - 1; i >= 0; i--) optimize_character_rects(sobel, &groups[i]);
  for (int i = 0; i < n_groups; i++) {
    grouped_rects *g = &groups[i];
    if (g->n < kMinimumExpiryStripCharacters) continue;
    /* expiry_seg.cpp:643-674 */
    for (int first = 0; first + 4 < g->n; first++)
      if (is_slash(sobel, &g->rects[first + 2])) emit_group(out, g, first, stripe_base_row);
  }
}

/* expiry_seg.cpp:707-902 */
void orc_best_expiry_seg(const uint8_t *card, int stride, int starting_y_offset, orc_expiry_result *out) {
  memset(out, 0, sizeof(*out));
  static __thread int16_t sobel[H * W];
  memset(sobel, 0, sizeof(sobel));
  const int y0 = starting_y_offset + ORC_NUM_H;
  if (y0 >= H) return;
  orc_scharr3_dx_abs(card + (size_t)y0 * stride, stride, W, H - y0, sobel + (size_t)y0 * W, W);

  const int first_stripe_base_row = y0 + 1;
  const int last_stripe_base_row = H - (kSmallCharacterHeight + 1);
  long line_sum[H];
  memset(line_sum, 0, sizeof(line_sum));
  const int left_edge = kSmallCharacterWidth * 3, right_edge = (W * 2) / 3;
  for (int row = first_stripe_base_row - 1; row < H; row++) {
    long s = 0;
    for (int c = left_edge; c < right_edge; c++) s += sobel[(size_t)row * W + c];
    line_sum[row] = s;
  }

  struct { int base_row; long sum; } stripes[H];
  int n_stripes = 0;
  for (int base_row = first_stripe_base_row; base_row < last_stripe_base_row; base_row++) {
    long sum = 0, threshold = 0;
    for (int row = base_row; row < base_row + kSmallCharacterHeight; row++) sum += line_sum[row];
    for (int row = base_row; row < base_row + kSmallCharacterHeight; row++)
      if (line_sum[row] > threshold) threshold = line_sum[row];
    threshold = threshold / 2;
    if (line_sum[base_row] + line_sum[base_row + 1] < threshold) continue;
    if (line_sum[base_row + kSmallCharacterHeight - 2] + line_sum[base_row + kSmallCharacterHeight - 1] < threshold)
      continue;
    int good = 1;
    for (int row = base_row; row < base_row + kSmallCharacterHeight - 3; row++)
      if (line_sum[row + 1] < threshold && line_sum[row + 2] < threshold) {
        good = 0;
        break;
      }
    if (good) {
      stripes[n_stripes].base_row = base_row;
      stripes[n_stripes].sum = sum;
      n_stripes++;
    }
  }
  /* std::sort descending by sum (the library's permutation, ties included); keep up to 3 that do
   * not overlap (expiry_seg.cpp:842-866) */
  long stripe_keys[H];
  int stripe_order[H];
  for (int i = 0; i < n_stripes; i++) stripe_keys[i] = stripes[i].sum;
  orc_sort_order_desc(n_stripes, stripe_keys, stripe_order);
  int probable_rows[3];
  long probable_sums[3];
  int n_probable = 0;
  for (int k = 0; k < n_stripes; k++) {
    const int best = stripe_order[k];
    int overlap = 0;
    for (int p = 0; p < n_probable; p++)
      if (probable_rows[p] - kSmallCharacterHeight < stripes[best].base_row &&
          stripes[best].base_row < probable_rows[p] + kSmallCharacterHeight) {
        overlap = 1;
        break;
      }
    if (!overlap) {
      probable_rows[n_probable] = stripes[best].base_row;
      probable_sums[n_probable] = stripes[best].sum;
      n_probable++;
      if (n_probable >= 3) break;
    }
  }
  out->n_stripes = n_probable;
  for (int p = 0; p < n_probable; p++) {
    out->stripe_base_row[p] = probable_rows[p];
    out->stripe_sum[p] = probable_sums[p];
  }
  for (int p = 0; p < n_probable; p++)
    find_character_groups_for_stripe(sobel, probable_rows[p], probable_sums[p], out);
}

/* expiry_categorize.cpp:35-70.  cvMorphologyEx(GRADIENT, 3x3 cross) on the isolated 11x16
 * ROI; llcv_equalize_hist; cvSmooth(CV_BILATERAL, 3, 3, 0.95, 2/3) = cv::bilateralFilter(d=3,
 * sigmaColor=0.95, sigmaSpace=2/3, BORDER_REPLICATE) -- radius 1, the five taps with r <= 1 in
 * the order (-1,0) (0,-1) (0,0) (0,1) (1,0), generic (non-SSE) accumulation; x (1/255). */
void orc_prepare_image_for_cat(const uint8_t *card, int stride, int left, int top, float x[176]) {
  uint8_t grad[kTrimmedH * kTrimmedW], sm[kTrimmedH * kTrimmedW];
  orc_morph_grad3_2d_cross(card + (size_t)top * stride + left, stride, kTrimmedW, kTrimmedH, grad, kTrimmedW);
  orc_equalize_hist(grad, kTrimmedW, kTrimmedW, kTrimmedH);
  const int aperture = 3;
  const double space_sigma = (aperture / 2.0 - 1) * 0.3 + 0.8;
  const double color_sigma = (aperture - 1) / 3.0;
  const double sigma_color = space_sigma, sigma_space = color_sigma; /* cvSmooth param3, param4 */
  const double gauss_color_coeff = -0.5 / (sigma_color * sigma_color);
  const double gauss_space_coeff = -0.5 / (sigma_space * sigma_space);
  float color_weight[256];
  for (int i = 0; i < 256; i++) color_weight[i] = (float)exp(i * i * gauss_color_coeff);
  static const int di[5] = {-1, 0, 0, 0, 1}, dj[5] = {0, -1, 0, 1, 0};
  float space_weight[5];
  for (int k = 0; k < 5; k++) {
    double r = sqrt((double)di[k] * di[k] + (double)dj[k] * dj[k]);
    space_weight[k] = (float)exp(r * r * gauss_space_coeff);
  }
  for (int r = 0; r < kTrimmedH; r++)
    for (int c = 0; c < kTrimmedW; c++) {
      float sum = 0, wsum = 0;
      int val0 = grad[r * kTrimmedW + c];
      for (int k = 0; k < 5; k++) {
        int rr = imin(imax(r + di[k], 0), kTrimmedH - 1), cc = imin(imax(c + dj[k], 0), kTrimmedW - 1);
        int val = grad[rr * kTrimmedW + cc];
        float w = space_weight[k] * color_weight[abs(val - val0)];
        sum += (float)val * w;
        wsum += w;
      }
      long v = lrint((double)(sum / wsum));
      sm[r * kTrimmedW + c] = (uint8_t)v;
    }
  const float s = 1.0f / 255.0f;
  for (int i = 0; i < 176; i++) x[i] = (float)sm[i] * s;
}

/* expiry_categorize.cpp:138-160: characters 0,1,3,4 of each group through the digit CNN */
void orc_categorize_expiry_groups(const uint8_t *card, int stride, orc_expiry_result *out) {
  for (int g = 0; g < out->n_groups; g++) {
    orc_expiry_group *grp = &out->groups[g];
    for (int ci = 0, row = 0; ci < 5; ci++) {
      if (ci == 2) continue;
      float x[176];
      orc_prepare_image_for_cat(card, stride, grp->char_left[ci], grp->char_top[ci], x);
      orc_applyc_expiry(x, grp->scores[row], NULL, NULL, NULL);
      row++;
    }
  }
  out->categorised = 1;
}

/* frame.cpp:71-73 (segmentation whenever the vseg gate passed and the number row leaves room
 * below it) and scan.cpp:57-64 (categorisation only for usable frames) */
void orc_scan_card_expiry(const uint8_t *card, int stride, const orc_frame_result *res,
                          orc_expiry_result *out) {
  memset(out, 0, sizeof(*out));
  if (!(res->flags & ORC_FLAG_VSEG_OK)) return;
  if (!(res->vseg_y_offset < H - 2 * kSmallCharacterHeight)) return;
  orc_best_expiry_seg(card, stride, res->vseg_y_offset, out);
  if (res->flags & ORC_FLAG_USABLE) orc_categorize_expiry_groups(card, stride, out);
}

/* ---- flat-array test hooks (tests/test_oracle_vs_ref.py pins these two against the
 * reference's own gather_into_groups / regrid_group, which need no OpenCV library symbol) ---- */
int orc_expiry_gather_into_groups(int n_items, const int *lefts, const int64_t *sums, int top, int height,
                                  int *group_n, int *group_left, int *group_width, int *rect_left,
                                  int64_t *rect_sum) {
  static __thread grouped_rects items[64], groups[64];
  if (n_items > 64) n_items = 64;
  for (int i = 0; i < n_items; i++) {
    items[i].top = top, items[i].left = lefts[i], items[i].width = kSmallCharacterWidth;
    items[i].height = height, items[i].grouped_yet = 0, items[i].sum = (long)sums[i];
    items[i].character_width = kSmallCharacterWidth, items[i].n = 0;
  }
  int n_groups = gather_into_groups(items, n_items, groups);
  int k = 0;
  for (int i = 0; i < n_groups; i++) {
    group_n[i] = groups[i].n, group_left[i] = groups[i].left, group_width[i] = groups[i].width;
    for (int j = 0; j < groups[i].n; j++, k++) rect_left[k] = groups[i].rects[j].left, rect_sum[k] = groups[i].rects[j].sum;
  }
  return n_groups;
}

void orc_expiry_regrid_group(const int16_t *sobel, int top, int height, int *left, int *width, int *character_width,
                             int *n, int *rect_left, int64_t *rect_sum) {
  static __thread grouped_rects g;
  g.top = top, g.height = height, g.left = *left, g.width = *width, g.character_width = *character_width, g.n = 0;
  regrid_group(sobel, &g);
  *left = g.left, *width = g.width, *character_width = g.character_width, *n = g.n;
  for (int i = 0; i < g.n; i++) rect_left[i] = g.rects[i].left, rect_sum[i] = g.rects[i].sum;
}
