// vg_pq_direct.hpp — scoring graph nodes from their PQ codes, pq.ComputeAsymmetricDistance (pq.go:234-260):
//     distance = 0; for s in 0..m-1: distance += squaredL2Int8Dequantized(q_s, codebook[s][code[s]], scale_s, offset_s)
// the way the reference scores the nodes of a DiskANN walk (diskann/segment.go:536-557).
//
// r02 read every term from the query's BuildDistanceTable image (m * 256 floats = 96 KiB at m = 96) in global memory.
// A popped node's fresh neighbours look up random centroids of every sub-quantizer, so one pop pulls nearly the WHOLE
// table through the CU's 32 KiB L1 (8 lines of 128 B per table row, 96 rows), and thousands of resident queries'
// tables (96 KiB each) live in neither L2 (4 MiB per XCD) nor the memory-side cache: 20 x the code bytes in fabric
// traffic (profiles/r02_traffic.json).  Here the terms of D of every SPL consecutive sub-quantizers are COMPUTED from
// the quantizer's int8 codebook — shared by every query, so one copy of those rows sits in the workgroup's LDS
// (D/SPL * m * 2 KiB) — and only the other SPL - D come from the query's table.  Both give the same bits: a table entry
// IS the term (BuildDistanceTable = the same five separately rounded fp32 ops per dimension, pq.go:468-491,
// kernels.go:354-374), and the terms are added in sub-quantizer order either way.
//
// Work split (sub-dimension 8, m = 8 * SPL): 8 lanes per node, 8 nodes per round.  Lane l of a group owns the SPL
// consecutive sub-quantizers [SPL*l, SPL*l + SPL): the first D direct (their 8 query floats, scale and offset live in
// registers for the whole walk), the rest from the table.  The sequential sum then runs through the group once: lane l
// continues lane l-1's partial sum (one DPP shift per hand-over).
#pragma once

#include "vg_device.hpp"
#include "vg_heap.hpp"

namespace vg {

// LDS written and read by ONE wave needs no s_barrier (a wave's LDS instructions execute in order); the fence keeps
// the compiler from moving the accesses across it.  Used where several waves of a workgroup walk different queries.
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// squaredL2Int8DequantizedGeneric (kernels.go:354-362) over one 8-dim sub-vector: the centroid's 8 int8 in `e`
__device__ __forceinline__ float pq_term8(uint2 e, const float (&q)[8], float scale, float offset)
{
    float sum = 0.0f;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const uint32_t w = j < 4 ? e.x : e.y;
        const int b = static_cast<int>(static_cast<int8_t>(w >> (8 * (j & 3))));
        float v = static_cast<float>(b) * scale;
        v = v + offset;
        const float d = q[j] - v;
        const float dd = d * d;
        sum = sum + dd;
    }
    return sum;
}

// next up-to-8 set bits of `mask` (ascending): the lane's 8-lane group gets the (lane>>3)-th
__device__ __forceinline__ int take8(uint64_t &mask, int lane)
{
    int mine = -1;
#pragma unroll
    for (int g = 0; g < 8; g++) {
        if (mask) {
            const int j = __builtin_ctzll(mask);
            mask &= mask - 1;
            if ((lane >> 3) == g) mine = j;
        }
    }
    return mine;
}

constexpr int kDppRowShr1 = 0x111;  // lane l reads lane l-1 of its row of 16

// image row of the LDS codebook copy -> sub-quantizer
template <int SPL, int D>
__device__ __forceinline__ int pq_direct_sub(int image_row)
{
    return SPL * (image_row / D) + (image_row % D);
}

// all threads of the workgroup: copy the D-of-every-SPL sub-quantizers' centroids (256 x 8 int8 each) into LDS
template <int SPL, int D>
__device__ __forceinline__ void pq_direct_stage(uint2 *cb_lds, const int8_t *__restrict__ codebooks, int tid, int nthreads)
{
    const uint2 *src = reinterpret_cast<const uint2 *>(codebooks);
    for (int e = tid; e < 8 * D * 256; e += nthreads) cb_lds[e] = src[pq_direct_sub<SPL, D>(e >> 8) * 256 + (e & 255)];
}

template <int SPL, int D>
struct PqDirectScorer {
    static_assert(SPL % 4 == 0 && SPL <= 16 && D >= 0 && D <= SPL, "codes are read as SPL/4 aligned dwords per lane");
    static constexpr int T = SPL - D;
    static constexpr bool kBounded = false;
    const uint8_t *rows;  // n * m code bytes, m = 8 * SPL
    const float *lut;     // this query's BuildDistanceTable image [m][256]; only rows s with s % SPL >= D are read
    const uint2 *cb;      // LDS, pq_direct_stage's image
    float q[D > 0 ? D : 1][8];
    float scale[D > 0 ? D : 1], offset[D > 0 ? D : 1];
    int l8;

    __device__ __forceinline__ static void sync() { wave_sync(); }

    __device__ __forceinline__ void load_query(const float *__restrict__ qv, const float *__restrict__ scales,
                                               const float *__restrict__ offsets, int lane)
    {
        l8 = lane & 7;
#pragma unroll
        for (int i = 0; i < D; i++) {
            const int s = SPL * l8 + i;
            const float4 a = *reinterpret_cast<const float4 *>(qv + s * 8);
            const float4 b = *reinterpret_cast<const float4 *>(qv + s * 8 + 4);
            q[i][0] = a.x, q[i][1] = a.y, q[i][2] = a.z, q[i][3] = a.w;
            q[i][4] = b.x, q[i][5] = b.y, q[i][6] = b.z, q[i][7] = b.w;
            scale[i] = scales[s];
            offset[i] = offsets[s];
        }
    }

    // one node per 8-lane group; the distance is returned in the group's last lane (l8 == 7)
    __device__ __forceinline__ float group_score(uint32_t id) const
    {
        const uint32_t *cw = reinterpret_cast<const uint32_t *>(rows + static_cast<int64_t>(id) * (8 * SPL) + SPL * l8);
        uint32_t w[SPL / 4];
#pragma unroll
        for (int u = 0; u < SPL / 4; u++) w[u] = cw[u];
        float tt[T > 0 ? T : 1];
#pragma unroll
        for (int i = 0; i < T; i++) {  // table rows first: the longer round trip
            const int s = D + i;
            tt[i] = lut[(SPL * l8 + s) * 256 + ((w[s >> 2] >> (8 * (s & 3))) & 0xFFu)];
        }
        uint2 e[D > 0 ? D : 1];
#pragma unroll
        for (int i = 0; i < D; i++) e[i] = cb[(l8 * D + i) * 256 + ((w[i >> 2] >> (8 * (i & 3))) & 0xFFu)];
        float td[D > 0 ? D : 1];
#pragma unroll
        for (int i = 0; i < D; i++) td[i] = pq_term8(e[i], q[i], scale[i], offset[i]);
        // distance += term, s = 0 .. m-1: lane l continues where lane l-1 stopped
        float out = 0.0f, in = 0.0f;
#pragma unroll
        for (int step = 0; step < 8; step++) {
            if (l8 == step) {
                float a = step == 0 ? 0.0f : in;
#pragma unroll
                for (int i = 0; i < D; i++) a = a + td[i];
#pragma unroll
                for (int i = 0; i < T; i++) a = a + tt[i];
                out = a;
            }
            if (step < 7)
                in = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(out), kDppRowShr1, 0xF, 0xF, false));
        }
        return out;
    }

    __device__ __forceinline__ float one(uint32_t id) const
    {
        const float o = group_score(id);  // every group scores the same node
        return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(o), 7));
    }

    __device__ __forceinline__ void many(uint64_t mask, uint32_t id_lane, int lane, float *nb_pair, float *nb_bnd) const
    {
        while (mask) {
            const int mine = take8(mask, lane);
            const uint32_t id = __shfl(id_lane, mine < 0 ? 0 : mine);
            if (mine >= 0) {
                const float o = group_score(id);
                if (l8 == 7) {
                    nb_pair[mine] = o;
                    if (nb_bnd) nb_bnd[mine] = o;
                }
            }
        }
    }
};

}  // namespace vg
