// orbx_introsort.h — exact replica of libstdc++'s std::sort (introsort) on an array of 64-bit elements.
//
// Why: ORBextractor::DistributeOctTree sorts (count, UL.x) pairs with std::sort and a comparator that ties
// on equal (count, UL.x) (src/ORBextractor.cc:542-555,686).  std::sort is unstable, so the order of tied
// nodes — which decides which nodes are split before the feature quota is reached and the order of the
// output keypoints — is whatever libstdc++'s introsort does.  To be bit-identical the device quadtree runs
// the same algorithm: median-of-3 quicksort down to 16-element runs with a 2*floor(log2 n) depth limit
// (heapsort beyond it), then one final insertion sort (bits/stl_algo.h: __sort, __introsort_loop,
// __unguarded_partition_pivot, __final_insertion_sort; bits/stl_heap.h).
//
// Elements are uint64; `Less` compares the KEY part only (the payload bits ride along), which is exactly
// how equivalent elements behave under std::sort.  tests/test_host_logic.py checks this replica against
// std::sort on tie-heavy inputs through orbx_debug_introsort.
#pragma once
#include <stdint.h>

#ifndef ORBX_HD
#ifdef __HIPCC__
#define ORBX_HD __host__ __device__
#else
#define ORBX_HD
#endif
#endif

namespace orbx {

struct KeyLess {  // key = bits 16..63 (count << 28 | ulx << 16), payload = bits 0..15
  ORBX_HD bool operator()(uint64_t a, uint64_t b) const { return (a >> 16) < (b >> 16); }
};

template <class T, class Less>
ORBX_HD inline void is_swap(T* a, int i, int j) {
  T t = a[i];
  a[i] = a[j];
  a[j] = t;
}

template <class T, class Less>
ORBX_HD inline void is_push_heap(T* a, int first, int hole, int top, T value, Less less) {
  int parent = (hole - 1) / 2;
  while (hole > top && less(a[first + parent], value)) {
    a[first + hole] = a[first + parent];
    hole = parent;
    parent = (hole - 1) / 2;
  }
  a[first + hole] = value;
}

template <class T, class Less>
ORBX_HD inline void is_adjust_heap(T* a, int first, int hole, int len, T value, Less less) {
  const int top = hole;
  int child = hole;
  while (child < (len - 1) / 2) {
    child = 2 * (child + 1);
    if (less(a[first + child], a[first + child - 1])) child--;
    a[first + hole] = a[first + child];
    hole = child;
  }
  if ((len & 1) == 0 && child == (len - 2) / 2) {
    child = 2 * (child + 1);
    a[first + hole] = a[first + child - 1];
    hole = child - 1;
  }
  is_push_heap<T, Less>(a, first, hole, top, value, less);
}

template <class T, class Less>
ORBX_HD inline void is_heapsort(T* a, int first, int last, Less less) {  // __partial_sort(first,last,last)
  const int len = last - first;
  if (len >= 2) {  // __make_heap
    int parent = (len - 2) / 2;
    while (true) {
      T v = a[first + parent];
      is_adjust_heap<T, Less>(a, first, parent, len, v, less);
      if (parent == 0) break;
      parent--;
    }
  }
  int l = last;
  while (l - first > 1) {  // __sort_heap
    --l;
    T v = a[l];
    a[l] = a[first];
    is_adjust_heap<T, Less>(a, first, 0, l - first, v, less);
  }
}

template <class T, class Less>
ORBX_HD inline void is_unguarded_linear_insert(T* a, int last, Less less) {
  T val = a[last];
  int next = last - 1;
  while (less(val, a[next])) {
    a[last] = a[next];
    last = next;
    --next;
  }
  a[last] = val;
}

template <class T, class Less>
ORBX_HD inline void is_insertion_sort(T* a, int first, int last, Less less) {
  if (first == last) return;
  for (int i = first + 1; i != last; ++i) {
    if (less(a[i], a[first])) {
      T val = a[i];
      for (int j = i; j > first; --j) a[j] = a[j - 1];
      a[first] = val;
    } else {
      is_unguarded_linear_insert<T, Less>(a, i, less);
    }
  }
}

template <class T, class Less>
ORBX_HD inline void introsort(T* a, int n, Less less) {
  if (n <= 0) return;
  int lg = 0;
  for (int t = n; t > 1; t >>= 1) lg++;
  // explicit stack replaces the recursion on the right partition; sub-ranges are disjoint, so the
  // processing order does not change the result, only each range's depth budget matters.
  int stf[64], stl[64], std_[64];
  int sp = 0;
  stf[0] = 0;
  stl[0] = n;
  std_[0] = 2 * lg;
  sp = 1;
  while (sp > 0) {
    --sp;
    int first = stf[sp], last = stl[sp], depth = std_[sp];
    while (last - first > 16) {
      if (depth == 0) {
        is_heapsort<T, Less>(a, first, last, less);
        break;
      }
      --depth;
      // __unguarded_partition_pivot
      const int mid = first + (last - first) / 2;
      {
        const int ia = first + 1, ib = mid, ic = last - 1;
        if (less(a[ia], a[ib])) {
          if (less(a[ib], a[ic])) is_swap<T, Less>(a, first, ib);
          else if (less(a[ia], a[ic])) is_swap<T, Less>(a, first, ic);
          else is_swap<T, Less>(a, first, ia);
        } else if (less(a[ia], a[ic])) is_swap<T, Less>(a, first, ia);
        else if (less(a[ib], a[ic])) is_swap<T, Less>(a, first, ic);
        else is_swap<T, Less>(a, first, ib);
      }
      int lo = first + 1, hi = last;
      const T pivot = a[first];  // the pivot slot is never touched by the partition loop
      while (true) {
        while (less(a[lo], pivot)) ++lo;
        --hi;
        while (less(pivot, a[hi])) --hi;
        if (!(lo < hi)) break;
        is_swap<T, Less>(a, lo, hi);
        ++lo;
      }
      const int cut = lo;
      stf[sp] = cut;
      stl[sp] = last;
      std_[sp] = depth;
      ++sp;
      last = cut;
    }
  }
  // __final_insertion_sort
  if (n > 16) {
    is_insertion_sort<T, Less>(a, 0, 16, less);
    for (int i = 16; i != n; ++i) is_unguarded_linear_insert<T, Less>(a, i, less);
  } else {
    is_insertion_sort<T, Less>(a, 0, n, less);
  }
}

}  // namespace orbx
