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
//   wave_partitions  a whole wave, the partition phase only, level by level over ALL ranges of a level at once (every
//                    range of one level has the same depth limit); NS consecutive positions per lane.  Returns false when
//                    the depth limit is reached (adversarial inputs): the caller then takes serial_sort.
#pragma once

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

// The partition phase for a wave: position p = NS lane + j holds e[j] (p < n); on return e[] is the arrangement the
// introsort loop leaves.  LDS: v[n] dwords (element exchange; between the exchanges it holds, per range, the k-th stop of
// the left scan and of the right scan as two 16-bit halves), and three small tables indexed by (range start >> 4) -- ranges
// in work are longer than 16, so that index is unique among them: tabP (the pivot, later the cut), tabB / tabE (how many
// left / right stops precede the range's start / lie before its end).  One wave per workgroup: __syncthreads() orders
// the LDS phases.
//
// A Hoare partition in closed form: with a_1 < a_2 < ... the positions (after `first`) whose element is NOT "> pivot" and
// b_1 > b_2 > ... those whose element is NOT "< pivot", the loop swaps (a_k, b_k) for k = 1 .. K while a_k < b_k, and returns
// a_{K+1} if that lies before b_K (b_0 = last), else b_K; untouched positions keep their elements, the pivot stays at first.
template <int NS, int SH>
__device__ __forceinline__ bool wave_partitions(unsigned (&e)[NS], const int n, const int lane, unsigned *v, unsigned *tabP,
                                                unsigned *tabB, unsigned *tabE) {
  unsigned FL[NS];  // range of the slot's position: first | last << 16
#pragma unroll
  for (int j = 0; j < NS; j++) FL[j] = (unsigned)n << 16;
  if (n <= 16) return true;
  int depth = 2 * (31 - __builtin_clz((unsigned)n));
  unsigned short *const pair = (unsigned short *)v;
  for (;;) {
    bool act[NS];
    bool any = false;
#pragma unroll
    for (int j = 0; j < NS; j++) {
      const int p = NS * lane + j;
      act[j] = p < n && (int)(FL[j] >> 16) - (int)(FL[j] & 0xffffu) > 16;
      any |= act[j];
    }
    if (__builtin_amdgcn_ballot_w64(any) == 0ull) return true;
    if (depth == 0) return false;
    --depth;
#pragma unroll
    for (int j = 0; j < NS; j++)
      if (NS * lane + j < n) v[NS * lane + j] = e[j];
    __syncthreads();
    // the slot at a range's first position moves the median of three there
#pragma unroll
    for (int j = 0; j < NS; j++) {
      const int p = NS * lane + j, F = (int)(FL[j] & 0xffffu), L = (int)(FL[j] >> 16);
      if (act[j] && p == F) {
        const int a = F + 1, b = F + (L - F) / 2, c = L - 1;
        const unsigned ea = v[a], eb = v[b], ec = v[c];
        int m;
        if (gt<SH>(ea, eb)) m = gt<SH>(eb, ec) ? b : (gt<SH>(ea, ec) ? c : a);
        else m = gt<SH>(ea, ec) ? a : (gt<SH>(eb, ec) ? c : b);
        const unsigned em = m == a ? ea : (m == b ? eb : ec);
        v[m] = e[j];
        v[F] = em;
        tabP[F >> 4] = em;
      }
    }
    __syncthreads();
    bool inA[NS], inB[NS];
    unsigned cA = 0u, cB = 0u;  // stops in the lanes below
#pragma unroll
    for (int j = 0; j < NS; j++) {
      const int p = NS * lane + j, F = (int)(FL[j] & 0xffffu);
      if (p < n) e[j] = v[p];
      const unsigned pk = act[j] ? tabP[F >> 4] >> SH : 0u;
      const unsigned k = e[j] >> SH;
      inA[j] = act[j] && p != F && k <= pk;
      inB[j] = act[j] && p != F && k >= pk;
      const unsigned long long ba = __builtin_amdgcn_ballot_w64(inA[j]), bb = __builtin_amdgcn_ballot_w64(inB[j]);
      cA = __builtin_amdgcn_mbcnt_hi((unsigned)(ba >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)ba, cA));
      cB = __builtin_amdgcn_mbcnt_hi((unsigned)(bb >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bb, cB));
    }
    unsigned pax[NS], pbx[NS];  // stops before the slot's position, over the whole list
#pragma unroll
    for (int j = 0; j < NS; j++) {
      pax[j] = cA, pbx[j] = cB;
      cA += inA[j] ? 1u : 0u;
      cB += inB[j] ? 1u : 0u;
      const int p = NS * lane + j, F = (int)(FL[j] & 0xffffu), L = (int)(FL[j] >> 16);
      if (act[j] && p == F) tabB[F >> 4] = pax[j] | (pbx[j] << 16);
      if (act[j] && p == L - 1) tabE[F >> 4] = cA | (cB << 16);
    }
    __syncthreads();
    unsigned rA[NS], rB[NS], nAB[NS];
#pragma unroll
    for (int j = 0; j < NS; j++) {
      const int F = (int)(FL[j] & 0xffffu);
      rA[j] = rB[j] = nAB[j] = 0u;
      if (act[j]) {
        const unsigned bs = tabB[F >> 4], en = tabE[F >> 4];
        const unsigned nA = (en & 0xffffu) - (bs & 0xffffu), nB = (en >> 16) - (bs >> 16);
        nAB[j] = nA | (nB << 16);
        rA[j] = pax[j] - (bs & 0xffffu);             // rank among the left stops, from the left
        rB[j] = nB - 1u - (pbx[j] - (bs >> 16));     // rank among the right stops, from the right
      }
    }
    __syncthreads();  // (the tables were read; v is rewritten as the pair table)
#pragma unroll
    for (int j = 0; j < NS; j++) {
      const int p = NS * lane + j, F = (int)(FL[j] & 0xffffu);
      if (inA[j]) pair[2 * (F + (int)rA[j])] = (unsigned short)p;
      if (inB[j]) pair[2 * (F + (int)rB[j]) + 1] = (unsigned short)p;
    }
    __syncthreads();
    int src[NS];
#pragma unroll
    for (int j = 0; j < NS; j++) {
      const int p = NS * lane + j, F = (int)(FL[j] & 0xffffu), L = (int)(FL[j] >> 16);
      const unsigned nA = nAB[j] & 0xffffu, nB = nAB[j] >> 16;
      src[j] = p;
      if (inA[j]) {
        const unsigned cur = v[F + (int)rA[j]];
        const bool sw = rA[j] < nB && p < (int)(cur >> 16);
        if (sw) {
          src[j] = (int)(cur >> 16);
        } else {
          const unsigned prev = rA[j] > 0u ? v[F + (int)rA[j] - 1] : 0u;
          const bool sw_prev = rA[j] > 0u && rA[j] - 1u < nB && (prev & 0xffffu) < (prev >> 16);
          if (rA[j] == 0u || sw_prev) {  // the first left stop that stays: the cut is here, or at the last right stop swapped
            const int bK = rA[j] > 0u ? (int)(prev >> 16) : L;
            tabP[F >> 4] = (unsigned)(p < bK ? p : bK);
          }
        }
      }
      if (inB[j]) {
        const unsigned cur = v[F + (int)rB[j]];
        if (rB[j] < nA && (int)(cur & 0xffffu) < p) {
          src[j] = (int)(cur & 0xffffu);
          if (rB[j] + 1u == nA) tabP[F >> 4] = (unsigned)p;  // every left stop was swapped: the scan ends on the last right one
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NS; j++)
      if (NS * lane + j < n) v[NS * lane + j] = e[j];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NS; j++) {
      const int p = NS * lane + j, F = (int)(FL[j] & 0xffffu), L = (int)(FL[j] >> 16);
      if (src[j] != p) e[j] = v[src[j]];
      if (act[j]) {
        const int cut = (int)tabP[F >> 4];
        FL[j] = p < cut ? ((unsigned)F | ((unsigned)cut << 16)) : ((unsigned)cut | ((unsigned)L << 16));
      }
    }
    __syncthreads();
  }
}

}  // namespace dmzsort
