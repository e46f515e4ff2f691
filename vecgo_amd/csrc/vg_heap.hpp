// vg_heap.hpp — searcher.PriorityQueue (internal/searcher/queue.go) restated operation by operation on a
// per-wave array, executed uniformly by the wave (same sift loops, same strict comparisons), so that
// equal-distance items fall exactly where the reference's 4-ary heap lets them fall.  Shared by the graph
// searches (k_graph.hip) and the HNSW builder (k_hnsw_build.hip).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace vg {

struct HItem {
    uint32_t node;
    float dist;
};

// ---- searcher.PriorityQueue (queue.go:161-183, 221-290), uniform over the wave ---------------
// An item is one 8-byte access (the heaps may live in HBM scratch, where every access is an L2 round trip)
__device__ __forceinline__ HItem heap_load(const HItem *p)
{
    const uint64_t v = *reinterpret_cast<const uint64_t *>(p);
    HItem it;
    it.node = static_cast<uint32_t>(v);
    it.dist = __uint_as_float(static_cast<uint32_t>(v >> 32));
    return it;
}
__device__ __forceinline__ void heap_store(HItem *p, HItem it)
{
    *reinterpret_cast<uint64_t *>(p) = static_cast<uint64_t>(it.node) | (static_cast<uint64_t>(__float_as_uint(it.dist)) << 32);
}

// Where a heap lives: a plain array (LDS, or HBM scratch), or SPLIT — items [0, nl) in LDS and the rest in HBM
// scratch under the same index.  The top levels of a 4-ary heap are where the sifts spend their steps (nl = 512
// covers levels 0-4 and half of level 5), the leaves are where the bulk of a large heap is.
struct SplitHeap {
    HItem *lo;  // LDS, items [0, nl)
    HItem *hi;  // HBM scratch, indexed by the item's heap index (its first nl entries are unused)
    int nl;
};
__device__ __forceinline__ HItem heap_get(const HItem *h, int i) { return heap_load(h + i); }
__device__ __forceinline__ void heap_put(HItem *h, int i, HItem it) { heap_store(h + i, it); }
// (a wave-uniform branch, not a pointer select: each side keeps its address space — ds_read / global_load
// instead of flat_load)
__device__ __forceinline__ HItem heap_get(const SplitHeap &h, int i)
{
    if (i < h.nl) return heap_load(h.lo + i);
    return heap_load(h.hi + i);
}
__device__ __forceinline__ void heap_put(const SplitHeap &h, int i, HItem it)
{
    if (i < h.nl)
        heap_store(h.lo + i, it);
    else
        heap_store(h.hi + i, it);
}

// the up-to-4 children fc .. fc+3 of a node (indices past `last` read `last` again), requested together
__device__ __forceinline__ void heap_get4(const HItem *h, int fc, int last, HItem (&c)[4])
{
#pragma unroll
    for (int j = 0; j < 4; j++) c[j] = heap_load(h + (fc + j < last ? fc + j : last));
}
__device__ __forceinline__ void heap_get4(const SplitHeap &h, int fc, int last, HItem (&c)[4])
{
    if (fc + 3 < h.nl) {
        heap_get4(h.lo, fc, last, c);
    } else if (fc >= h.nl) {
        heap_get4(h.hi, fc, last, c);
    } else {
#pragma unroll
        for (int j = 0; j < 4; j++) c[j] = heap_get(h, fc + j < last ? fc + j : last);
    }
}

// `it` is the item that belongs at slot i (not yet stored there)
template <bool MAX, typename H>
__device__ __forceinline__ void heap_sift_up(H h, int i, const HItem it)
{
    while (i > 0) {
        const int p = (i - 1) >> 2;
        const HItem pi = heap_get(h, p);
        if (MAX ? (it.dist <= pi.dist) : (it.dist >= pi.dist)) break;
        heap_put(h, i, pi);
        i = p;
    }
    heap_put(h, i, it);
}

// The up-to-4 children of a node are requested together (one round trip per level, not one per child plus one for
// the move) and compared in the reference's order: first child, then each next one with a strict comparison.
// `it` is the item that belongs at slot i (not yet stored there)
template <bool MAX, typename H>
__device__ __forceinline__ void heap_sift_down(H h, int n, int i, const HItem it)
{
    for (;;) {
        const int fc = 4 * i + 1;
        if (fc >= n) break;
        const int last = n - 1;
        HItem c[4];
        heap_get4(h, fc, last, c);
        int best = fc;
        HItem bi = c[0];
#pragma unroll
        for (int j = 1; j < 4; j++) {
            if (fc + j < n && (MAX ? (c[j].dist > bi.dist) : (c[j].dist < bi.dist))) {
                best = fc + j;
                bi = c[j];
            }
        }
        if (MAX ? (it.dist >= bi.dist) : (it.dist <= bi.dist)) break;
        heap_put(h, i, bi);
        i = best;
    }
    heap_put(h, i, it);
}

template <bool MAX, typename H>
__device__ __forceinline__ void heap_push(H h, int &len, HItem it)
{
    len++;
    heap_sift_up<MAX>(h, len - 1, it);
}

template <bool MAX, typename H>
__device__ __forceinline__ HItem heap_pop(H h, int &len)
{
    const HItem top = heap_get(h, 0);
    len--;
    if (len > 0) heap_sift_down<MAX>(h, len, 0, heap_get(h, len));
    return top;
}

// PushItemBounded (queue.go:67-92) on the max-heap of results
template <typename H>
__device__ __forceinline__ void res_push_bounded(H h, int &len, HItem it, int capacity)
{
    if (len < capacity) {
        heap_push<true>(h, len, it);
        return;
    }
    if (it.dist < heap_get(h, 0).dist) heap_sift_down<true>(h, len, 0, it);
}

// TryPushBounded (queue.go:190-215) on the MIN-heap of exploration candidates: at capacity the
// new item replaces the top (the closest!) when it is farther — restated as written
template <typename H>
__device__ __forceinline__ void cand_try_push_bounded(H h, int &len, HItem it, int max_size)
{
    if (len < max_size) {
        heap_push<false>(h, len, it);
        return;
    }
    if (it.dist <= heap_get(h, 0).dist) return;
    heap_sift_down<false>(h, len, 0, it);
}

// next up-to-4 set bits of `mask` (ascending): the lane's 16-lane group gets the (lane>>4)-th
__device__ __forceinline__ int take4(uint64_t &mask, int lane)
{
    int mine = -1;
#pragma unroll
    for (int g = 0; g < 4; g++) {
        if (mask) {
            const int j = __builtin_ctzll(mask);
            mask &= mask - 1;
            if ((lane >> 4) == g) mine = j;
        }
    }
    return mine;
}

enum { kMetricL2 = 0, kMetricCos = 1, kMetricDot = 2 };

}  // namespace vg
