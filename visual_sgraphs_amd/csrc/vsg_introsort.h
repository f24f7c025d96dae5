// vsg_introsort.h -- move-for-move re-implementation of libstdc++'s std::sort (bits/stl_algo.h,
// bits/stl_heap.h; unchanged between GCC 5 and 13) on an array of 64-bit items compared by a 32-bit key.
//
// Why: DistributeOctTree sorts (size, node*) pairs with a comparator that has many ties
// (ORBextractor.cc:539-560, :707) and then consumes the result back to front, so the output ORDER of
// equal keys -- which is decided by introsort's exact swap sequence -- decides the keypoint order the
// reference emits.  A device thread replays the same sequence.
//
// Item = (key << 32) | payload; ordering uses ONLY the key (strict '<'), like compareNodes.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define VSG_SORT_HD __host__ __device__ inline
#else
#define VSG_SORT_HD inline
#endif

namespace vsg {
namespace introsort {

typedef uint64_t item_t;
VSG_SORT_HD bool less(item_t a, item_t b) { return (uint32_t)(a >> 32) < (uint32_t)(b >> 32); }
VSG_SORT_HD void iter_swap(item_t *a, item_t *b) {
  item_t t = *a;
  *a = *b;
  *b = t;
}

// ---- heap pieces (stl_heap.h)
VSG_SORT_HD void push_heap_(item_t *first, int holeIndex, int topIndex, item_t value) {
  int parent = (holeIndex - 1) / 2;
  while (holeIndex > topIndex && less(first[parent], value)) {
    first[holeIndex] = first[parent];
    holeIndex = parent;
    parent = (holeIndex - 1) / 2;
  }
  first[holeIndex] = value;
}

VSG_SORT_HD void adjust_heap_(item_t *first, int holeIndex, int len, item_t value) {
  const int topIndex = holeIndex;
  int secondChild = holeIndex;
  while (secondChild < (len - 1) / 2) {
    secondChild = 2 * (secondChild + 1);
    if (less(first[secondChild], first[secondChild - 1])) secondChild--;
    first[holeIndex] = first[secondChild];
    holeIndex = secondChild;
  }
  if ((len & 1) == 0 && secondChild == (len - 2) / 2) {
    secondChild = 2 * (secondChild + 1);
    first[holeIndex] = first[secondChild - 1];
    holeIndex = secondChild - 1;
  }
  push_heap_(first, holeIndex, topIndex, value);
}

// std::__partial_sort(first, last, last) == __heap_select (make_heap only) + __sort_heap
VSG_SORT_HD void heap_sort_(item_t *first, int len) {
  if (len >= 2) {
    int parent = (len - 2) / 2;
    while (true) {
      item_t value = first[parent];
      adjust_heap_(first, parent, len, value);
      if (parent == 0) break;
      parent--;
    }
  }
  int last = len;
  while (last > 1) {
    --last;
    item_t value = first[last];
    first[last] = first[0];
    adjust_heap_(first, 0, last, value);
  }
}

// ---- quicksort pieces (stl_algo.h)
VSG_SORT_HD void move_median_to_first_(item_t *result, item_t *a, item_t *b, item_t *c) {
  if (less(*a, *b)) {
    if (less(*b, *c))
      iter_swap(result, b);
    else if (less(*a, *c))
      iter_swap(result, c);
    else
      iter_swap(result, a);
  } else if (less(*a, *c))
    iter_swap(result, a);
  else if (less(*b, *c))
    iter_swap(result, c);
  else
    iter_swap(result, b);
}

VSG_SORT_HD item_t *unguarded_partition_(item_t *first, item_t *last, item_t *pivot) {
  while (true) {
    while (less(*first, *pivot)) ++first;
    --last;
    while (less(*pivot, *last)) --last;
    if (!(first < last)) return first;
    iter_swap(first, last);
    ++first;
  }
}

VSG_SORT_HD void unguarded_linear_insert_(item_t *last) {
  item_t val = *last;
  item_t *next = last;
  --next;
  while (less(val, *next)) {
    *last = *next;
    last = next;
    --next;
  }
  *last = val;
}

VSG_SORT_HD void insertion_sort_(item_t *first, item_t *last) {
  if (first == last) return;
  for (item_t *i = first + 1; i != last; ++i) {
    if (less(*i, *first)) {
      item_t val = *i;
      for (item_t *p = i; p != first; --p) *p = *(p - 1);  // std::move_backward(first, i, i + 1)
      *first = val;
    } else
      unguarded_linear_insert_(i);
  }
}

VSG_SORT_HD int lg_(int n) {  // std::__lg: floor(log2(n)), n > 0
  int r = 0;
  while (n > 1) {
    n >>= 1;
    r++;
  }
  return r;
}

// The __introsort_loop half of std::sort: quicksort partitioning down to runs of <= 16 (heapsort past the depth
// limit).  What remains for std::sort is __final_insertion_sort, which moves an element left only past strictly
// greater ones -- i.e. it is exactly a STABLE sort of the array this function leaves behind.  A caller with many
// threads runs this on one of them and then ranks the items in parallel (rank = #smaller + #equal-before).
VSG_SORT_HD void partition_phase(item_t *first, int n, int depth_limit_override = -1) {
  const int kThreshold = 16;
  // __introsort_loop, recursion on the right part turned into an explicit stack
  // (depth <= 2*lg(n) + 1 <= 64 for any int n)
  struct Frame {
    int lo, hi, depth;
  };
  Frame stack[64];
  int sp = 0;
  int lo = 0, hi = n, depth = depth_limit_override >= 0 ? depth_limit_override : lg_(n) * 2;
  while (true) {
    while (hi - lo > kThreshold) {
      if (depth == 0) {
        heap_sort_(first + lo, hi - lo);
        break;
      }
      --depth;
      item_t *f = first + lo, *l = first + hi;
      item_t *mid = f + (l - f) / 2;
      move_median_to_first_(f, f + 1, mid, l - 1);
      int cut = (int)(unguarded_partition_(f + 1, l, f) - first);
      // the library recurses into [cut, hi) first, then loops on [lo, cut)
      stack[sp].lo = lo;
      stack[sp].hi = cut;
      stack[sp].depth = depth;
      sp++;
      lo = cut;
    }
    if (sp == 0) break;
    --sp;
    lo = stack[sp].lo;
    hi = stack[sp].hi;
    depth = stack[sp].depth;
  }
}

// std::sort(first, first + n, comp).  `depth_limit_override` < 0 uses the library's 2*lg(n).
VSG_SORT_HD void sort(item_t *first, int n, int depth_limit_override = -1) {
  if (n <= 0) return;
  const int kThreshold = 16;
  partition_phase(first, n, depth_limit_override);
  // __final_insertion_sort
  if (n > kThreshold) {
    insertion_sort_(first, first + kThreshold);
    for (item_t *i = first + kThreshold; i != first + n; ++i) unguarded_linear_insert_(i);
  } else
    insertion_sort_(first, first + n);
}

}  // namespace introsort
}  // namespace vsg
