// vg_cand_replay.hpp — searcher.CandidateHeap (internal/searcher/candidate_queue.go) replayed operation by operation, for the
// queries whose scores may hold a NaN.
//
// The scans keep their best k rows by a 64-bit key (score bits, row id): a total order, which is what the reference's heap
// implements as long as no score is a NaN.  A NaN score is neither better nor worse than anything (candidate_queue.go:12-38:
// `a.Score != b.Score` is true, both `<` and `>` false), so the reference's outcome is then decided by the heap's layout: a NaN
// that enters while the heap is filling stays; at the root it is never replaced (flat/segment.go:714-721 asks
// InternalCandidateBetter(cand, top)); as a first child it stops a sift (down() :151-183).  That outcome is DEFINED — the loop is
// sequential — and this file reproduces it: one workgroup per query walks the rows in the reference's order, scores them with the
// caller's exact row arithmetic, and wave 0 runs the reference's Push / ReplaceTop on an LDS array with float comparisons.  The
// result is what the engine takes out of the heap: Pop() until empty (engine/search.go:859-862), reported best first.
//
// Which queries: Scorer::risk() — a conservative test on the INPUTS: could a score be a NaN or an Inf?  (A non-finite query value
// or index datum; magnitudes whose partial sums could overflow.  +Inf scores are ties the reference breaks by row id INSIDE the
// heap's history; the scans' pre-tests `score < bound` drop them while the bound is still +Inf.)  A query without risk returns at once (the
// launch costs a few microseconds per search call); a query with risk overwrites what the fast path wrote for it.  Rare by
// construction — such inputs are garbage — so the walk is written for exactness, not speed: ~n * dim / 100 GB/s per query.
#pragma once

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>

#include "vg_device.hpp"
#include "vg_exact.hpp"
#include "vg_internal.hpp"

namespace vg {

struct CItem {
    float score;
    uint32_t row;
};

// InternalCandidateBetter / Worse (candidate_queue.go:12-38) within one segment
__device__ __forceinline__ bool cand_better(const CItem a, const CItem b, const bool desc)
{
    if (a.score != b.score) return desc ? a.score > b.score : a.score < b.score;
    return a.row < b.row;
}
__device__ __forceinline__ bool cand_worse(const CItem a, const CItem b, const bool desc)
{
    if (a.score != b.score) return desc ? a.score < b.score : a.score > b.score;
    return a.row > b.row;
}
__device__ __forceinline__ CItem cand_load(const CItem *h, int i)
{
    const uint64_t v = reinterpret_cast<const uint64_t *>(h)[i];
    return CItem{__uint_as_float(static_cast<uint32_t>(v >> 32)), static_cast<uint32_t>(v)};
}
__device__ __forceinline__ void cand_store(CItem *h, int i, const CItem it)
{
    reinterpret_cast<uint64_t *>(h)[i] = static_cast<uint64_t>(it.row) | (static_cast<uint64_t>(__float_as_uint(it.score)) << 32);
}
// up() :136-147 — `it` belongs at slot j
__device__ __forceinline__ void cand_up(CItem *h, int j, const CItem it, const bool desc)
{
    while (j > 0) {
        const int i = (j - 1) >> 2;
        const CItem p = cand_load(h, i);
        if (!cand_worse(it, p, desc)) break;
        cand_store(h, j, p);
        j = i;
    }
    cand_store(h, j, it);
}
// down() :151-183 — `it` belongs at slot i of a heap of n items
__device__ __forceinline__ void cand_down(CItem *h, int i, const int n, const CItem it, const bool desc)
{
    for (;;) {
        const int fc = 4 * i + 1;
        if (fc >= n) break;
        int best = fc;
        CItem bi = cand_load(h, fc);
        const int lc = fc + 4 < n ? fc + 4 : n;
        for (int c = fc + 1; c < lc; c++) {
            const CItem ci = cand_load(h, c);
            if (cand_worse(ci, bi, desc)) {
                best = c;
                bi = ci;
            }
        }
        if (!cand_worse(bi, it, desc)) break;
        cand_store(h, i, bi);
        i = best;
    }
    cand_store(h, i, it);
}
// Pop() :75-82: Swap(0, n), down(0, n), the old root leaves
__device__ __forceinline__ CItem cand_pop(CItem *h, int &len, const bool desc)
{
    const int n = len - 1;
    const CItem root = cand_load(h, 0), last = cand_load(h, n);
    cand_store(h, n, root);
    cand_down(h, 0, n, last, desc);
    len = n;
    return root;
}

// the heap a search is written with: CandidateHeap here; searcher.PriorityQueue for hnsw.BruteSearch / searchBitmap (k_brute.hip)
struct CandHeapPolicy {
    // the reference's test in front of the heap operation, against the heap as it stands (flat/segment.go:714-721)
    __device__ static bool accepts(const CItem x, int len, int k, const CItem root, bool desc) { return len < k || cand_better(x, root, desc); }
    __device__ static void offer(CItem *h, int &len, int k, const CItem x, bool desc)
    {
        if (len < k) {  // h.Push(cand)
            cand_up(h, len, x, desc);
            len++;
        } else {  // h.ReplaceTop(cand)
            cand_down(h, 0, len, x, desc);
        }
    }
    __device__ static CItem pop(CItem *h, int &len, bool desc) { return cand_pop(h, len, desc); }
};

// 16 waves score a step's rows (a row's score is a chain load -> arithmetic -> next row per 16-lane group or lane: the walk's pace is
// how many of those chains run side by side — 256 threads: 114 ms for one query over 1M x 768, 1024: 90; tools/nan_replay_time.py)
constexpr int kReplayThreads = 1024;
constexpr int kReplayChunk = 1024;  // rows scored per step (the per-row scorers map thread t to row row0 + t)

// Scorer (by value; device pointers inside):
//   (qi: the query's index in the call, q: its vector)
//   bool risk(int64_t qi, const float *q, int tid)  block-uniform: may a score of this query be a NaN?  (all threads call it)
//   void prepare(int64_t qi, const float *q, int tid)  once per risky query, before the first chunk (LDS staging, ...)
//   void score_chunk(int64_t qi, const float *q, int64_t row0, int64_t n, int tid, float *out)
//                                                 out[i] = score of row row0 + i for i < min(kReplayChunk, n - row0), by all threads
// mask: a row filter per query (bit i of byte i / 8: the row takes part), or null.
template <class Scorer, class Heap>
__global__ __launch_bounds__(kReplayThreads) void cand_replay_kernel(Scorer sc, const float *__restrict__ queries, int dim, int64_t n, int k,
                                                                     bool desc, const uint8_t *__restrict__ mask, int64_t mask_stride,
                                                                     uint32_t *__restrict__ ids, float *__restrict__ scores,
                                                                     int *__restrict__ replayed, int64_t q_first,
                                                                     const uint32_t *__restrict__ probes, int np,
                                                                     const uint32_t *__restrict__ part_off)
{
    extern __shared__ uint64_t replay_lds[];
    CItem *heap = reinterpret_cast<CItem *>(replay_lds);  // k items
    float *chunk = reinterpret_cast<float *>(replay_lds + k + 4);
    const int tid = threadIdx.x, lane = tid & 63;
    const int64_t q = blockIdx.x;
    const float *qv = queries + q * dim;
    const int64_t qi = q_first + q;
    if (!sc.risk(qi, qv, tid)) return;
    if (replayed && tid == 0) atomicAdd(replayed, 1);
    const uint8_t *mq = mask ? mask + q * mask_stride : nullptr;
    sc.prepare(qi, qv, tid);
    int len = 0;  // wave 0's copy is the live one
    // the rows in the reference's order: the whole segment, or — a partitioned segment — the row ranges of the query's probed
    // partitions in FindClosestCentroids' order, one after the other into the same heap (flat/segment.go:727-744)
    for (int jr = 0; jr < (probes ? np : 1); jr++) {
    int64_t r0 = 0, r1 = n;
    if (probes) {
        const uint32_t p = probes[q * np + jr];
        r0 = part_off[p];
        r1 = part_off[p + 1];
    }
    for (int64_t row0 = r0; row0 < r1; row0 += kReplayChunk) {
        __syncthreads();  // the previous chunk has been replayed
        sc.score_chunk(qi, qv, row0, r1, tid, chunk);
        __syncthreads();
        if (tid >= 64) continue;
        const int cnt = static_cast<int>(r1 - row0 < kReplayChunk ? r1 - row0 : kReplayChunk);
        for (int j0 = 0; j0 < cnt; j0 += 64) {
            const int j = j0 + lane;
            const bool live = j < cnt && mask_bit(mq, row0 + j);  // filter.Matches before the row is scored (flat/segment.go:631-635)
            const CItem mine{live ? chunk[j] : 0.0f, static_cast<uint32_t>(row0 + j)};
            int from = 0;  // lanes below it have been decided
            for (;;) {
                // the reference's test against the heap AS IT STANDS (flat/segment.go:714-721): a lane that fails it now is
                // retested after every accepted row before it, so nothing is assumed about how the root moves
                const CItem root = len > 0 ? cand_load(heap, 0) : CItem{0.0f, 0u};
                const uint64_t m = __ballot(live && lane >= from && Heap::accepts(mine, len, k, root, desc));
                if (m == 0) break;
                const int b = __builtin_ctzll(m);
                const CItem x{__uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(mine.score), b)),
                              static_cast<uint32_t>(__builtin_amdgcn_readlane(mine.row, b))};
                Heap::offer(heap, len, k, x, desc);
                from = b + 1;
            }
        }
    }
    }
    if (tid >= 64) return;
    const int nres = len;
    for (int i = nres - 1; i >= 0; i--) {
        const CItem it = Heap::pop(heap, len, desc);
        if (lane == 0) {
            ids[q * k + i] = it.row;
            scores[q * k + i] = it.score;
        }
    }
    for (int i = nres + lane; i < k; i += 64) {
        ids[q * k + i] = VG_INVALID_ID;
        scores[q * k + i] = desc ? -INFINITY : INFINITY;
    }
}

// block-uniform "any thread saw it" (all threads call; LDS word provided by the caller)
__device__ __forceinline__ bool block_any(bool mine, int *flag, int tid)
{
    if (tid == 0) *flag = 0;
    __syncthreads();
    if (mine) *flag = 1;  // (same value from every writer)
    __syncthreads();
    const bool r = *flag != 0;
    __syncthreads();
    return r;
}
__device__ __forceinline__ bool is_finite_f32(float x) { return (__float_as_uint(x) & 0x7F800000u) != 0x7F800000u; }
// the largest |term| of a dot product / squared distance between a value of magnitude a and one of magnitude b
__device__ __forceinline__ float score_bound(float a, float b, bool dot) { return dot ? a * b : (a + b) * (a + b); }

// probes / np / part_off: a partitioned segment's probe lists (device, nq * np partition ids) and partition bounds, or null
template <class Heap = CandHeapPolicy, class Scorer>
inline int32_t launch_cand_replay(const Scorer &sc, const float *queries, int dim, int64_t n, int64_t nq, int k, bool desc, const uint8_t *mask,
                                  int64_t mask_stride, uint32_t *ids, float *scores, hipStream_t st, int *replayed = nullptr,
                                  const uint32_t *probes = nullptr, int np = 0, const uint32_t *part_off = nullptr)
{
    if (nq == 0 || k == 0 || hook(kHookNoCandReplay)) return VG_OK;
    const size_t lds = sizeof(uint64_t) * (static_cast<size_t>(k) + 4) + sizeof(float) * kReplayChunk;  // (+ 4: vg_heap.hpp reads a node's four children together)
    for (int64_t q0 = 0; q0 < nq; q0 += 1 << 30) {
        const int64_t cnt = std::min<int64_t>(nq - q0, 1 << 30);
        VG_LAUNCH((cand_replay_kernel<Scorer, Heap>), dim3(static_cast<unsigned>(cnt)), dim3(kReplayThreads), lds, st, sc, queries + q0 * dim, dim, n, k,
                  desc, mask ? mask + q0 * mask_stride : nullptr, mask_stride, ids + q0 * k, scores + q0 * k, replayed, q0, probes ? probes + q0 * np : nullptr, np, part_off);
    }
    return VG_OK;
}

// The scorer for fp32 rows: distance.SquaredL2 / distance.Dot of flat/segment.go:691-701 (squaredL2Avx512 / dotProductAvx512
// order, 16 lanes per pair: 16 rows per step of the workgroup); conv: 0 the value, 1 its negative, 2 half of it — the HNSW index's
// distances (-dot for Dot, 0.5 * squared L2 for Cosine: hnsw.go:2218-2238)
struct FlatF32Scorer {
    const float *base;
    const float *maxabs;  // [1]: max |x| over the rows, +Inf when one of them is not finite (vg_index_set_vectors)
    int dim;
    bool dot;
    int conv;
    __device__ bool risk(int64_t, const float *q, int tid) const
    {
        __shared__ int flag;
        const float ma = maxabs[0];
        bool bad = !is_finite_f32(ma);
        for (int j = tid; j < dim; j += kReplayThreads) {
            const float v = q[j];
            // finite values: no partial sum overflows below dim * max|q| * max|x| (dot products: a NaN needs +Inf and -Inf) resp.
            // dim * (max|q| + max|x|)^2 (squared distances: +Inf scores — ties the scans' keys do not keep like the heap does)
            bad = bad || !is_finite_f32(v) || !(score_bound(fabsf(v), ma, dot) * static_cast<float>(dim) < 1e38f);
        }
        return block_any(bad, &flag, tid);
    }
    __device__ void prepare(int64_t, const float *, int) const {}
    __device__ void score_chunk(int64_t, const float *q, int64_t row0, int64_t n, int tid, float *out) const
    {
        const Sub16 sub = Sub16::make(tid);
        for (int r = tid >> 4; r < kReplayChunk; r += kReplayThreads / 16) {
            const int64_t row = row0 + r;
            if (row >= n) break;  // (a whole 16-lane group)
            const float *x = base + row * dim;
            float v = dot ? exact_pair16<true, kPair>(x, q, dim, sub) : exact_pair16<false, kPair>(x, q, dim, sub);
            v = conv == 1 ? -v : conv == 2 ? 0.5f * v : v;
            if ((tid & 15) == 0) out[r] = v;
        }
    }
};

}  // namespace vg
