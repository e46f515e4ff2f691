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

// `it` is the item that belongs at slot i (not yet stored there)
template <bool MAX>
__device__ __forceinline__ void heap_sift_up(HItem *h, int i, const HItem it)
{
    while (i > 0) {
        const int p = (i - 1) >> 2;
        const HItem pi = heap_load(h + p);
        if (MAX ? (it.dist <= pi.dist) : (it.dist >= pi.dist)) break;
        heap_store(h + i, pi);
        i = p;
    }
    heap_store(h + i, it);
}

// The up-to-4 children of a node are requested together (one round trip per level, not one per child plus one for
// the move) and compared in the reference's order: first child, then each next one with a strict comparison.
// `it` is the item that belongs at slot i (not yet stored there)
template <bool MAX>
__device__ __forceinline__ void heap_sift_down(HItem *h, int n, int i, const HItem it)
{
    for (;;) {
        const int fc = 4 * i + 1;
        if (fc >= n) break;
        const int last = n - 1;
        const HItem c0 = heap_load(h + fc);
        const HItem c1 = heap_load(h + (fc + 1 < last ? fc + 1 : last));
        const HItem c2 = heap_load(h + (fc + 2 < last ? fc + 2 : last));
        const HItem c3 = heap_load(h + (fc + 3 < last ? fc + 3 : last));
        int best = fc;
        HItem bi = c0;
        if (fc + 1 < n && (MAX ? (c1.dist > bi.dist) : (c1.dist < bi.dist))) {
            best = fc + 1;
            bi = c1;
        }
        if (fc + 2 < n && (MAX ? (c2.dist > bi.dist) : (c2.dist < bi.dist))) {
            best = fc + 2;
            bi = c2;
        }
        if (fc + 3 < n && (MAX ? (c3.dist > bi.dist) : (c3.dist < bi.dist))) {
            best = fc + 3;
            bi = c3;
        }
        if (MAX ? (it.dist >= bi.dist) : (it.dist <= bi.dist)) break;
        heap_store(h + i, bi);
        i = best;
    }
    heap_store(h + i, it);
}

template <bool MAX>
__device__ __forceinline__ void heap_push(HItem *h, int &len, HItem it)
{
    len++;
    heap_sift_up<MAX>(h, len - 1, it);
}

template <bool MAX>
__device__ __forceinline__ HItem heap_pop(HItem *h, int &len)
{
    const HItem top = heap_load(h);
    len--;
    if (len > 0) heap_sift_down<MAX>(h, len, 0, heap_load(h + len));
    return top;
}

// PushItemBounded (queue.go:67-92) on the max-heap of results
__device__ __forceinline__ void res_push_bounded(HItem *h, int &len, HItem it, int capacity)
{
    if (len < capacity) {
        heap_push<true>(h, len, it);
        return;
    }
    if (it.dist < h[0].dist) heap_sift_down<true>(h, len, 0, it);
}

// TryPushBounded (queue.go:190-215) on the MIN-heap of exploration candidates: at capacity the
// new item replaces the top (the closest!) when it is farther — restated as written
__device__ __forceinline__ void cand_try_push_bounded(HItem *h, int &len, HItem it, int max_size)
{
    if (len < max_size) {
        heap_push<false>(h, len, it);
        return;
    }
    if (it.dist <= h[0].dist) return;
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
