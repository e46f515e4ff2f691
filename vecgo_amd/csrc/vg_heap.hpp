// vg_heap.hpp — searcher.PriorityQueue (internal/searcher/queue.go) restated operation by operation on a
// per-wave array, executed uniformly by the wave (same sift loops, same strict comparisons), so that
// equal-distance items fall exactly where the reference's 4-ary heap lets them fall.  Shared by the graph
// searches (k_graph.hip) and the HNSW builder (k_hnsw_build.hip).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <type_traits>

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
// (the struct carries generic pointers; telling the compiler which memory each side is keeps ds_read / global_load
// where a merged branch would otherwise become a pointer select and a flat_load — slower for the LDS side, and
// counted against both wait counters)
__device__ __forceinline__ HItem *split_lo(const SplitHeap &h)
{
    HItem *p = h.lo;
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_assume(__builtin_amdgcn_is_shared(p));
#endif
    return p;
}
__device__ __forceinline__ HItem *split_hi(const SplitHeap &h)
{
    HItem *p = h.hi;
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_assume(!__builtin_amdgcn_is_shared(p));
    __builtin_assume(!__builtin_amdgcn_is_private(p));
#endif
    return p;
}
__device__ __forceinline__ HItem heap_get(const HItem *h, int i) { return heap_load(h + i); }
__device__ __forceinline__ void heap_put(HItem *h, int i, HItem it) { heap_store(h + i, it); }
// (a wave-uniform branch, not a pointer select: each side keeps its address space — ds_read / global_load
// instead of flat_load)
__device__ __forceinline__ HItem heap_get(const SplitHeap &h, int i)
{
    if (i < h.nl) return heap_load(split_lo(h) + i);
    return heap_load(split_hi(h) + i);
}
__device__ __forceinline__ void heap_put(const SplitHeap &h, int i, HItem it)
{
    if (i < h.nl)
        heap_store(split_lo(h) + i, it);
    else
        heap_store(split_hi(h) + i, it);
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
        heap_get4(split_lo(h), fc, last, c);
    } else if (fc >= h.nl) {
        heap_get4(split_hi(h), fc, last, c);
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
__device__ __forceinline__ void heap_sift_down_f32(H h, int n, int i, const HItem it, float *root = nullptr)
{
    if (root) *root = it.dist;  // what slot i holds afterwards: `it`, or the first child that moved up
    bool moved = false;
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
        if (root && !moved) *root = bi.dist;
        moved = true;
        i = best;
    }
    heap_put(h, i, it);
}

__device__ __forceinline__ int heap_lane() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
__device__ __forceinline__ uint32_t heap_readlane(uint32_t v, int l)
{
    return static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(v), l));
}

// UK ("unsigned keys"): every distance in the heap is >= +0 (squared L2, half of it for cosine, PQ / RaBitQ sums of
// squares; never the negated dot product), so the order of the floats is the order of their bit patterns as
// unsigned integers and every comparison below is the reference's comparison.  (NOT for NaN: a NaN distance — NaN
// or Inf in a row or a query — has a bit pattern above +Inf's and sorts as the largest key here, while in the
// reference every comparison with it is false; the graph searches choose UK from the metric alone, watch every scored
// neighbour list for a NaN, and answer a query that meets one with the float instantiation — k_graph.hip `redo`.)  That moves a sift's decisions from
// the vector ALU — where a wave-uniform step costs a full 64-lane issue slot — to the scalar ALU, and lets one
// round trip serve three heap levels:
//  * which child is best does not depend on the item being sifted, and a sift only moves items UP its path, so the 4
//    children, 16 grandchildren and 64 great-grandchildren of slot i (three contiguous runs) can be fetched before
//    the first decision: lanes 0-3 / 0-15 / 0-63 issue ONE read each, three levels per LDS (or HBM) round trip
//    instead of one — at ef >= 512 the walk was waiting on exactly these round trips (9 waves per CU, ~10 dependent
//    LDS reads per push);
//  * the candidates' distances come to scalar registers by v_readlane, the first-strict-best scan and the
//    comparison with the sifted item are s_cmp / s_cselect, one more v_readlane fetches the winner's node:
//    ~9 vector instructions per level where the r02 loop spent ~20.
// Slots are read as 8-byte items by per-lane index; lanes past the heap's end do not load.
__device__ __forceinline__ uint64_t heap_load_u64(const HItem *h, int i) { return *reinterpret_cast<const uint64_t *>(h + i); }
__device__ __forceinline__ uint64_t heap_load_u64(const SplitHeap &h, int i)
{
    if (i < h.nl) return *reinterpret_cast<const uint64_t *>(split_lo(h) + i);
    return *reinterpret_cast<const uint64_t *>(split_hi(h) + i);
}
// lanes 0 .. cnt-1 read the run of slots first .. first+cnt-1 (those below n).  A run lies on one side of the split
// nearly always: a wave-uniform branch per side keeps ds_read / global_load (a per-lane choice becomes a pointer select
// and a flat_load); only a run that straddles the boundary chooses per lane
__device__ __forceinline__ uint64_t heap_load_run_u64(const HItem *h, int first, int cnt, int n, int lane)
{
    uint64_t v = 0;
    if (lane < cnt && first + lane < n) v = heap_load_u64(h, first + lane);
    return v;
}
__device__ __forceinline__ uint64_t heap_load_run_u64(const SplitHeap &h, int first, int cnt, int n, int lane)
{
    uint64_t v = 0;
    const bool mine = lane < cnt && first + lane < n;
    if (first + cnt <= h.nl) {
        if (mine) v = *reinterpret_cast<const uint64_t *>(split_lo(h) + first + lane);
    } else if (first >= h.nl) {
        if (mine) v = *reinterpret_cast<const uint64_t *>(split_hi(h) + first + lane);
    } else if (mine) {
        v = heap_load_u64(h, first + lane);
    }
    return v;
}

// the reference's choice among the `cnt` (1..4) children whose distances lanes l0 .. l0+cnt-1 hold in `vd`
template <bool MAX>
__device__ __forceinline__ void heap_pick4_uk(uint32_t vd, int l0, int cnt, int &best, uint32_t &bk)
{
    best = 0;
    bk = heap_readlane(vd, l0);
#pragma unroll
    for (int j = 1; j < 4; j++) {
        if (j < cnt) {
            const uint32_t dj = heap_readlane(vd, l0 + j);
            if (MAX ? (dj > bk) : (dj < bk)) {
                best = j;
                bk = dj;
            }
        }
    }
}

// one level per round trip: heaps in LDS (a round trip is ~100 cycles; fetching levels the sift may not reach cost
// more than it saved: 2.53 vs 2.64 ms per 8192 PQ walks at ef 128, 12.7 vs 13.5 at ef 512)
template <bool MAX>
__device__ __forceinline__ void heap_sift_down_uk(HItem *h, int n, int i, const HItem it, float *root = nullptr)
{
    const int lane = heap_lane();
    const uint32_t itk = heap_readlane(__float_as_uint(it.dist), 0);
    if (root) *root = it.dist;
    bool moved = false;
    for (;;) {
        const int fc = 4 * i + 1;
        if (fc >= n) break;
        uint64_t c = 0;
        if (lane < 4 && fc + lane < n) c = heap_load_u64(h, fc + lane);
        int b1;
        uint32_t bk;
        heap_pick4_uk<MAX>(static_cast<uint32_t>(c >> 32), 0, n - fc < 4 ? n - fc : 4, b1, bk);
        if (MAX ? (itk >= bk) : (itk <= bk)) break;
        heap_put(h, i, HItem{heap_readlane(static_cast<uint32_t>(c), b1), __uint_as_float(bk)});
        if (root && !moved) *root = __uint_as_float(bk);
        moved = true;
        i = fc + b1;
    }
    heap_put(h, i, it);
}

// three levels per round trip: heaps whose lower levels live in HBM scratch (SplitHeap)
template <bool MAX>
__device__ __forceinline__ void heap_sift_down_uk(const SplitHeap &h, int n, int i, const HItem it, float *root = nullptr)
{
    const int lane = heap_lane();
    const uint32_t itk = heap_readlane(__float_as_uint(it.dist), 0);
    if (root) *root = it.dist;
    bool moved = false;
    for (;;) {
        const int fc = 4 * i + 1;
        if (fc >= n) break;
        const int fg = 4 * fc + 1, fgg = 16 * fc + 5;  // first grandchild, first great-grandchild
        uint64_t g = 0, gg = 0;
        const uint64_t c = heap_load_run_u64(h, fc, 4, n, lane);
        if (fg < n) {
            g = heap_load_run_u64(h, fg, 16, n, lane);
            if (fgg < n) gg = heap_load_run_u64(h, fgg, 64, n, lane);
        }
        int b1, b2, b3;
        uint32_t bk;
        heap_pick4_uk<MAX>(static_cast<uint32_t>(c >> 32), 0, n - fc < 4 ? n - fc : 4, b1, bk);
        if (MAX ? (itk >= bk) : (itk <= bk)) break;
        heap_put(h, i, HItem{heap_readlane(static_cast<uint32_t>(c), b1), __uint_as_float(bk)});
        if (root && !moved) *root = __uint_as_float(bk);
        moved = true;
        i = fc + b1;
        const int fc2 = 4 * i + 1;
        if (fc2 >= n) break;
        heap_pick4_uk<MAX>(static_cast<uint32_t>(g >> 32), 4 * b1, n - fc2 < 4 ? n - fc2 : 4, b2, bk);
        if (MAX ? (itk >= bk) : (itk <= bk)) break;
        heap_put(h, i, HItem{heap_readlane(static_cast<uint32_t>(g), 4 * b1 + b2), __uint_as_float(bk)});
        i = fc2 + b2;
        const int fc3 = 4 * i + 1;
        if (fc3 >= n) break;
        heap_pick4_uk<MAX>(static_cast<uint32_t>(gg >> 32), 16 * b1 + 4 * b2, n - fc3 < 4 ? n - fc3 : 4, b3, bk);
        if (MAX ? (itk >= bk) : (itk <= bk)) break;
        heap_put(h, i, HItem{heap_readlane(static_cast<uint32_t>(gg), 16 * b1 + 4 * b2 + b3), __uint_as_float(bk)});
        i = fc3 + b3;
    }
    heap_put(h, i, it);
}

template <bool MAX, bool UK = false, typename H>
__device__ __forceinline__ void heap_sift_down(H h, int n, int i, const HItem it, float *root = nullptr)
{
    if constexpr (UK)
        heap_sift_down_uk<MAX>(h, n, i, it, root);
    else
        heap_sift_down_f32<MAX>(h, n, i, it, root);
}

template <bool MAX, typename H>
__device__ __forceinline__ void heap_push(H h, int &len, HItem it)
{
    len++;
    heap_sift_up<MAX>(h, len - 1, it);
}

template <bool MAX, bool UK = false, typename H>
__device__ __forceinline__ HItem heap_pop(H h, int &len)
{
    const HItem top = heap_get(h, 0);
    len--;
    if (len > 0) heap_sift_down<MAX, UK>(h, len, 0, heap_get(h, len));
    return top;
}

// ---- a run of PushItem calls on the MIN-heap of candidates, in the order of the set bits of `mask` ----------------
// (lane j holds item j = {id_lane, d_lane}).  Pushed one by one, every item costs a dependent LDS round trip for its
// parent before anything else can happen; a run of n pushes lands in the consecutive slots len .. len+n-1, whose
// parents are at most n/4 + 2 consecutive slots that NO item of the run occupies (true once len >= 25), so they are
// fetched ONCE into registers (lane t holds parent t) and the run is replayed against the registers: item and parent
// distances meet by v_readlane, an item that stays below its parent (the common case) costs no memory access at all,
// an item that climbs swaps with the register copy and continues through the untracked ancestors in memory exactly
// as heap_sift_up does.  One store writes the run's slots, one writes the parents back.  Every comparison and its
// order are the reference's (queue.go:221-245), so the heap ends byte for byte as n single pushes leave it.
__device__ __forceinline__ void heap_store_lane(HItem *h, int i, uint32_t node, float dist) { heap_store(h + i, HItem{node, dist}); }
__device__ __forceinline__ void heap_store_lane(const SplitHeap &h, int i, uint32_t node, float dist) { heap_put(h, i, HItem{node, dist}); }
__device__ __forceinline__ HItem heap_load_lane(const HItem *h, int i) { return heap_load(h + i); }
__device__ __forceinline__ HItem heap_load_lane(const SplitHeap &h, int i) { return heap_get(h, i); }

template <typename H>
__device__ __forceinline__ void heap_push_run_min(H h, int &len, uint64_t mask, uint32_t id_lane, float d_lane)
{
    const int n = __popcll(mask);
    if (n == 0) return;
    const int len0 = len;
    if (len0 < 32 || n < 3) {  // tiny heaps (a run slot could be another's parent / a tracked slot an ancestor: len < 25) and short runs
        while (mask) {
            const int j = __builtin_ctzll(mask);
            mask &= mask - 1;
            heap_push<false>(h, len, HItem{heap_readlane(id_lane, j), __uint_as_float(heap_readlane(__float_as_uint(d_lane), j))});
        }
        return;
    }
    const int lane = heap_lane();
    const int p0 = (len0 - 1) >> 2, np = ((len0 + n - 2) >> 2) - p0 + 1;  // tracked parents: slots p0 .. p0+np-1
    HItem P{0u, 0.0f};
    if (lane < np) P = heap_load_lane(h, p0 + lane);
    uint32_t pend_n = id_lane;  // what the run's slot of this lane's item will hold
    float pend_d = d_lane;
    int r = 0;
    for (uint64_t todo = mask; todo; todo &= todo - 1, r++) {
        const int j = __builtin_ctzll(todo);
        const int t = ((len0 + r - 1) >> 2) - p0;
        const float itd = __uint_as_float(heap_readlane(__float_as_uint(d_lane), j));
        const float pd = __uint_as_float(heap_readlane(__float_as_uint(P.dist), t));
        if (itd >= pd) continue;  // heap_sift_up's first test: the item stays in its slot
        const uint32_t itn = heap_readlane(id_lane, j);
        const uint32_t pn = heap_readlane(P.node, t);
        if (lane == j) {  // the parent steps down into the run's slot
            pend_n = pn;
            pend_d = pd;
        }
        // the item now stands at the tracked slot p0 + t; its ancestors are in memory
        const int i = p0 + t;
        HItem put{itn, itd};
        if (i > 0) {
            const int pp = (i - 1) >> 2;
            const HItem gi = heap_get(h, pp);
            if (!(itd >= gi.dist)) {  // climbs on: the grandparent steps down into the tracked slot
                put = gi;
                heap_sift_up<false>(h, pp, HItem{itn, itd});
            }
        }
        if (lane == t) P = put;
    }
    if ((mask >> lane) & 1) heap_store_lane(h, len0 + __popcll(mask & ((1ull << lane) - 1)), pend_n, pend_d);
    if (lane < np) heap_store_lane(h, p0 + lane, P.node, P.dist);
    len = len0 + n;
}

// PushItemBounded (queue.go:67-92) on the max-heap of results
template <bool UK = false, typename H>
__device__ __forceinline__ void res_push_bounded(H h, int &len, HItem it, int capacity)
{
    if (len < capacity) {
        heap_push<true>(h, len, it);
        return;
    }
    if (it.dist < heap_get(h, 0).dist) heap_sift_down<true, UK>(h, len, 0, it);
}

// PushItemBounded on a FULL results heap whose top distance the caller tracks (`top`, updated): one LDS round trip
// less per call than reading it back
template <bool UK = false, typename H>
__device__ __forceinline__ void res_replace_top(H h, int len, HItem it, float &top)
{
    if (it.dist < top) heap_sift_down<true, UK>(h, len, 0, it, &top);
}

// TryPushBounded (queue.go:190-215) on the MIN-heap of exploration candidates: at capacity the
// new item replaces the top (the closest!) when it is farther — restated as written
template <bool UK = false, typename H>
__device__ __forceinline__ void cand_try_push_bounded(H h, int &len, HItem it, int max_size)
{
    if (len < max_size) {
        heap_push<false>(h, len, it);
        return;
    }
    if (it.dist <= heap_get(h, 0).dist) return;
    heap_sift_down<false, UK>(h, len, 0, it);
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
