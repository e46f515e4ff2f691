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
template <bool MAX>
__device__ __forceinline__ void heap_sift_up(HItem *h, int i)
{
    const HItem it = h[i];
    while (i > 0) {
        const int p = (i - 1) >> 2;
        const float pd = h[p].dist;
        if (MAX ? (it.dist <= pd) : (it.dist >= pd)) break;
        h[i] = h[p];
        i = p;
    }
    h[i] = it;
}

template <bool MAX>
__device__ __forceinline__ void heap_sift_down(HItem *h, int n, int i)
{
    const HItem it = h[i];
    for (;;) {
        const int fc = 4 * i + 1;
        if (fc >= n) break;
        int best = fc;
        float bd = h[fc].dist;
        const int lc = fc + 4 < n ? fc + 4 : n;
        for (int c = fc + 1; c < lc; c++) {
            const float cd = h[c].dist;
            if (MAX ? (cd > bd) : (cd < bd)) {
                best = c;
                bd = cd;
            }
        }
        if (MAX ? (it.dist >= bd) : (it.dist <= bd)) break;
        h[i] = h[best];
        i = best;
    }
    h[i] = it;
}

template <bool MAX>
__device__ __forceinline__ void heap_push(HItem *h, int &len, HItem it)
{
    h[len] = it;
    len++;
    heap_sift_up<MAX>(h, len - 1);
}

template <bool MAX>
__device__ __forceinline__ HItem heap_pop(HItem *h, int &len)
{
    const HItem top = h[0];
    h[0] = h[len - 1];
    len--;
    if (len > 0) heap_sift_down<MAX>(h, len, 0);
    return top;
}

// PushItemBounded (queue.go:67-92) on the max-heap of results
__device__ __forceinline__ void res_push_bounded(HItem *h, int &len, HItem it, int capacity)
{
    if (len < capacity) {
        heap_push<true>(h, len, it);
        return;
    }
    if (it.dist < h[0].dist) {
        h[0] = it;
        heap_sift_down<true>(h, len, 0);
    }
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
    h[0] = it;
    heap_sift_down<false>(h, len, 0);
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
