// dmz_stdsort.h -- the candidate order of the expiry segmentation (device code only).
//
// The reference orders its window sums and its stripe sums with std::sort and a "sum >" comparator
// (scan/expiry_seg.cpp:75-87, 456, 842).  std::sort is not stable: which of two EQUAL sums comes first is whatever the
// standard library's algorithm leaves, and the greedy picks that follow (expiry_seg.cpp:496-529, 848-866) take the first of two
// overlapping candidates.  The reference's library is libstdc++; its std::sort is a deterministic function of the comparison
// results: an introsort loop (ranges of more than 16: median of first+1 / middle / last-1 swapped to first, an unguarded
// Hoare partition around it, depth limit 2 floor(log2 n), heap sort of a range that reaches it) followed by ONE insertion
// sort over everything.  Insertion sort is stable, so the final order is: sum descending, equal sums in the order of their
// positions after the partition phase.  That is what the two forms below deliver -- oracle/orc_expiry.c restates the same
// algorithm on the CPU and is pinned on the reference's own instantiation (oracle/_ref, tests/test_oracle_vs_ref.py).
//
// Elements are packed as (sum << SH) | index; the comparator looks at the sum only.
//   serial_sort      one lane, the whole std::sort (heap sort included): v[] ends up in the library's final order.
//   wave_mark_partitions  a whole wave, the partition phase only, and only the ranges that still hold two MARKED elements
//                    (the elements whose order the caller needs).  Returns false when the depth limit is reached
//                    (adversarial inputs): the caller then takes serial_sort.
#pragma once

#include "dmz_wave.h"

#ifdef DMZ_XSEG_DBG  /* developer counters (tools/dev/xseg_dbg.py) */
__device__ unsigned long long g_xs_dbg[16];
#endif

namespace dmzsort {

template <int SH>
__device__ __forceinline__ bool gt(unsigned a, unsigned b) {
  return (a >> SH) > (b >> SH);
}

// ---- heap helpers of the depth-limit fallback (bits/stl_heap.h: __push_heap, __adjust_heap) on v[first .. first + len) ----
template <int SH>
__device__ inline void adjust_heap(unsigned *v, int first, int hole, int len, unsigned value) {
  const int top = hole;
  int child = hole;
  while (child < (len - 1) / 2) {
    child = 2 * (child + 1);
    if (gt<SH>(v[first + child], v[first + child - 1])) child--;
    v[first + hole] = v[first + child];
    hole = child;
  }
  if ((len & 1) == 0 && child == (len - 2) / 2) {
    child = 2 * (child + 1);
    v[first + hole] = v[first + child - 1];
    hole = child - 1;
  }
  int parent = (hole - 1) / 2;
  while (hole > top && gt<SH>(v[first + parent], value)) {
    v[first + hole] = v[first + parent];
    hole = parent;
    parent = (hole - 1) / 2;
  }
  v[first + hole] = value;
}

template <int SH>
__device__ inline void linear_insert(unsigned *v, int last) {  // __unguarded_linear_insert
  const unsigned value = v[last];
  int next = last - 1;
  while (gt<SH>(value, v[next])) {
    v[last] = v[next];
    last = next;
    --next;
  }
  v[last] = value;
}

// std::sort(v, v + n, "sum >") executed by the calling lane alone; stack: >= 2 floor(log2 n) + 2 dwords
template <int SH>
__device__ inline void serial_sort(unsigned *v, int n, unsigned *stack) {
  if (n <= 0) return;
  int sp = 0;
  int first = 0, last = n, depth = 2 * (31 - __builtin_clz((unsigned)n));
  for (;;) {
    while (last - first > 16) {
      if (depth == 0) {  // __partial_sort(first, last, last): make_heap, then sort_heap
        const int len = last - first;
        for (int parent = (len - 2) / 2;; parent--) {
          adjust_heap<SH>(v, first, parent, len, v[first + parent]);
          if (parent == 0) break;
        }
        for (int end = last; end - first > 1;) {
          --end;
          const unsigned value = v[end];
          v[end] = v[first];
          adjust_heap<SH>(v, first, 0, end - first, value);
        }
        break;
      }
      --depth;
      {  // __move_median_to_first(first, first + 1, mid, last - 1)
        const int a = first + 1, b = first + (last - first) / 2, c = last - 1;
        const unsigned ea = v[a], eb = v[b], ec = v[c];
        int m;
        if (gt<SH>(ea, eb)) m = gt<SH>(eb, ec) ? b : (gt<SH>(ea, ec) ? c : a);
        else m = gt<SH>(ea, ec) ? a : (gt<SH>(eb, ec) ? c : b);
        const unsigned t = v[first];
        v[first] = v[m];
        v[m] = t;
      }
      const unsigned pivot = v[first];
      int lo = first + 1, hi = last;  // __unguarded_partition(first + 1, last, first)
      for (;;) {
        while (gt<SH>(v[lo], pivot)) ++lo;
        --hi;
        while (gt<SH>(pivot, v[hi])) --hi;
        if (!(lo < hi)) break;
        const unsigned t = v[lo];
        v[lo] = v[hi];
        v[hi] = t;
        ++lo;
      }
      stack[sp++] = (unsigned)lo | ((unsigned)last << 10) | ((unsigned)depth << 20);  // the right part, later
      last = lo;
    }
    if (sp == 0) break;
    const unsigned s = stack[--sp];
    first = (int)(s & 1023u), last = (int)((s >> 10) & 1023u), depth = (int)(s >> 20);
  }
  // __final_insertion_sort
  const int head = n > 16 ? 16 : n;
  for (int i = 1; i < head; ++i) {
    if (gt<SH>(v[i], v[0])) {
      const unsigned value = v[i];
      for (int k = i; k > 0; k--) v[k] = v[k - 1];
      v[0] = value;
    } else
      linear_insert<SH>(v, i);
  }
  for (int i = head; i < n; ++i) linear_insert<SH>(v, i);
}

// The partition phase for a wave, restricted to what the caller needs: the relative order of MARKED elements (bit 29 of an
// element; the caller marks the elements whose order among equal sums can matter).  v[0 .. n) is the list in LDS; ranges are
// taken one at a time (explicit stack), a range is partitioned only while it is longer than 16 AND holds at least two marked
// elements -- the arrangement inside any other range cannot change the order of two marked elements with equal sums (ranges are
// disjoint, and an element never leaves its range).  On return every marked element sits where the library's partition
// phase would leave it relative to every other marked element of equal sum.  Returns false when a range reaches the depth
// limit (adversarial inputs): the caller then takes serial_sort.
//
// One range [F, L), position p = F + 64 j + lane in slot j.  A Hoare partition in closed form: with a_1 < a_2 < ... the
// positions after F whose element is NOT "> pivot" and b_1 > b_2 > ... those whose element is NOT "< pivot", the loop swaps
// (a_k, b_k) for k = 1 .. K while a_k < b_k, and returns a_{K+1} if that lies before b_K (b_0 = L), else b_K; untouched
// positions keep their elements, the pivot stays at F.  Ranks are ballots + popcounts, the k-th stops meet in T[k]
// (16-bit halves; K <= 209), the exchange reads the old arrangement before it writes (LDS operations of one wave execute
// in order).  One wave per workgroup; no barriers are needed, only the compiler must keep the order of the LDS accesses.
constexpr unsigned MARK = 1u << 29;
constexpr int TPAIRS = 210;
#define DMZ_LDS_ORDER() __asm__ volatile("" ::: "memory")

// one partition of [F, L) with NS >= ceil((L - F) / 64) slots, straight-line: every LDS load of a phase is issued before the
// first one is used (addresses of idle slots are clamped to something readable), so a partition costs four LDS round trips.
// Returns the cut; mL / mR = marked elements left / right of it.
template <int NS, int SH, unsigned KMASK>
__device__ __forceinline__ int partition_range(unsigned *v, const int F, const int L, const int lane, unsigned *T, int &mL,
                                               int &mR) {
  unsigned short *const T16 = (unsigned short *)T;
  // __move_median_to_first(F, F + 1, mid, L - 1): wave-uniform
  unsigned pk;
  {
    const int a = F + 1, b = F + ((L - F) >> 1), c = L - 1;
    const unsigned ea = v[a], eb = v[b], ec = v[c], ef = v[F];
    const unsigned ka = (ea >> SH) & KMASK, kb = (eb >> SH) & KMASK, kc = (ec >> SH) & KMASK;
    const int m = ka > kb ? (kb > kc ? b : (ka > kc ? c : a)) : (ka > kc ? a : (kb > kc ? c : b));
    const unsigned em = m == a ? ea : (m == b ? eb : ec);
    pk = (unsigned)__builtin_amdgcn_readfirstlane((int)((em >> SH) & KMASK));
    DMZ_LDS_ORDER();
    if (lane == 0) {
      v[m] = ef;
      v[F] = em;
    }
    DMZ_LDS_ORDER();
  }
  unsigned e[NS], rA[NS], rB[NS];
  bool inA[NS], inB[NS], valid[NS];
#pragma unroll
  for (int j = 0; j < NS; j++) {
    const int p = F + 64 * j + lane;
    valid[j] = p < L;
    e[j] = v[valid[j] ? p : F];
  }
  int nA = 0, nB = 0;
#pragma unroll
  for (int j = 0; j < NS; j++) {
    const int p = F + 64 * j + lane;
    const unsigned k = (e[j] >> SH) & KMASK;
    inA[j] = valid[j] && p != F && k <= pk;
    inB[j] = valid[j] && p != F && k >= pk;
    const unsigned long long ba = __builtin_amdgcn_ballot_w64(inA[j]), bb = __builtin_amdgcn_ballot_w64(inB[j]);
    rA[j] = __builtin_amdgcn_mbcnt_hi((unsigned)(ba >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)ba, (unsigned)nA));
    rB[j] = __builtin_amdgcn_mbcnt_hi((unsigned)(bb >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bb, (unsigned)nB));
    nA += __popcll(ba);
    nB += __popcll(bb);
  }
  DMZ_LDS_ORDER();
  const unsigned npair = (unsigned)(nA < nB ? (nA < TPAIRS ? nA : TPAIRS) : (nB < TPAIRS ? nB : TPAIRS));
#pragma unroll
  for (int j = 0; j < NS; j++) {
    const int p = F + 64 * j + lane;
    rB[j] = (unsigned)(nB - 1) - rB[j];  // rank among the right stops, from the right
    if (inA[j] && rA[j] < npair) T16[2 * rA[j]] = (unsigned short)p;
    if (inB[j] && rB[j] < npair) T16[2 * rB[j] + 1] = (unsigned short)p;
  }
  DMZ_LDS_ORDER();
  unsigned tA[NS], tB[NS];
#pragma unroll
  for (int j = 0; j < NS; j++) {
    tA[j] = T[inA[j] && rA[j] < npair ? rA[j] : 0u];
    tB[j] = T[inB[j] && rB[j] < npair ? rB[j] : 0u];
  }
  int src[NS];
  unsigned stop = 0xffffu;  // the first left stop that stays / the last right stop that moves: the cut is the smaller
#pragma unroll
  for (int j = 0; j < NS; j++) {
    const int p = F + 64 * j + lane;
    const int bk = (int)(tA[j] >> 16), ak = (int)(tB[j] & 0xffffu);
    const bool swA = inA[j] && rA[j] < npair && p < bk;
    const bool swB = inB[j] && rB[j] < npair && ak < p;
    src[j] = swA ? bk : (swB ? ak : (valid[j] ? p : F));
    if ((inA[j] && !swA) || swB) stop = stop < (unsigned)p ? stop : (unsigned)p;
  }
  const int first_stop = (int)dmzwave::min_u32(stop);
  const int cut = first_stop < L ? first_stop : L;
  DMZ_LDS_ORDER();
  unsigned en[NS];
#pragma unroll
  for (int j = 0; j < NS; j++) en[j] = v[src[j]];  // the old arrangement
  DMZ_LDS_ORDER();
  mL = mR = 0;
#pragma unroll
  for (int j = 0; j < NS; j++) {
    const int p = F + 64 * j + lane;
    if (valid[j] && src[j] != p) v[p] = en[j];
    const bool mk = valid[j] && (en[j] & MARK) != 0u;
    mL += __popcll(__builtin_amdgcn_ballot_w64(mk && p < cut));
    mR += __popcll(__builtin_amdgcn_ballot_w64(mk && p >= cut));
  }
  DMZ_LDS_ORDER();
  return cut;
}

template <int SH, unsigned KMASK>
__device__ __forceinline__ bool wave_mark_partitions(unsigned *v, const int n, const int lane, unsigned *T, unsigned *stack) {
  if (n <= 16) return true;
  int sp = 0;
  int F = 0, L = n, depth = 2 * (31 - __builtin_clz((unsigned)n));
  for (;;) {
    if (depth == 0) return false;
    --depth;
    const int len = L - F;
#ifdef DMZ_XSEG_DBG
    if (lane == 0) atomicAdd(&g_xs_dbg[8], 1ull), atomicAdd(&g_xs_dbg[9], (unsigned long long)((len + 63) >> 6));
#endif
    int mL, mR, cut;
    switch ((len + 63) >> 6) {  // straight-line code per range size: 64 positions per slot
      case 1: cut = partition_range<1, SH, KMASK>(v, F, L, lane, T, mL, mR); break;
      case 2: cut = partition_range<2, SH, KMASK>(v, F, L, lane, T, mL, mR); break;
      case 3: cut = partition_range<3, SH, KMASK>(v, F, L, lane, T, mL, mR); break;
      case 4: cut = partition_range<4, SH, KMASK>(v, F, L, lane, T, mL, mR); break;
      case 5: cut = partition_range<5, SH, KMASK>(v, F, L, lane, T, mL, mR); break;
      case 6: cut = partition_range<6, SH, KMASK>(v, F, L, lane, T, mL, mR); break;
      default: cut = partition_range<7, SH, KMASK>(v, F, L, lane, T, mL, mR); break;
    }
    const bool goL = cut - F > 16 && mL >= 2, goR = L - cut > 16 && mR >= 2;
    if (goL && goR) {
      if (lane == 0) stack[sp] = (unsigned)cut | ((unsigned)L << 10) | ((unsigned)depth << 20);
      sp++;
      L = cut;
    } else if (goL) {
      L = cut;
    } else if (goR) {
      F = cut;
    } else {
      if (sp == 0) return true;
      DMZ_LDS_ORDER();
      const unsigned s = (unsigned)__builtin_amdgcn_readfirstlane((int)stack[--sp]);
      F = (int)(s & 1023u), L = (int)((s >> 10) & 1023u), depth = (int)(s >> 20);
    }
  }
}
#undef DMZ_LDS_ORDER

}  // namespace dmzsort
