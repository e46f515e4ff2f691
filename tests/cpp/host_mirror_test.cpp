// host_mirror_test.cpp — exercises include/vecgo_hip.hpp (the C++ mirror of the reference's Go
// interfaces) the way the reference's own tests exercise the Go ones:
//   internal/quantization/pq_test.go:10-140, rabitq_test.go:10-97, internal/kmeans/kmeans_test.go:12-93,
//   distance/distance_test.go, internal/segment/flat/pq_test.go:17-93.
// The flat search is checked bit-for-bit against the CPU oracle (test infrastructure).
// Exit code 0 = all checks passed; 77 = no gfx950 device (the mirror refused to run: there is no
// CPU fallback).
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>

#include "vecgo_hip.hpp"
#include "vg_oracle.h"

using namespace vecgo;
using quantization::ProductQuantizer;
using quantization::RaBitQuantizer;

static int g_fail = 0;
#define EXPECT(cond)                                                         \
    do {                                                                     \
        if (!(cond)) {                                                       \
            std::fprintf(stderr, "FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond); \
            g_fail++;                                                        \
        }                                                                    \
    } while (0)

template <typename F>
static int32_t status_of(F f)
{
    try {
        f();
    } catch (const Error &e) {
        return e.status();
    }
    return VG_OK;
}

static std::vector<std::vector<float>> unit_vectors(std::mt19937 &rng, int n, int dim)
{
    std::normal_distribution<float> nd;
    std::vector<std::vector<float>> out(n, std::vector<float>(dim));
    for (auto &v : out) {
        double s = 0;
        for (auto &x : v) {
            x = nd(rng);
            s += double(x) * x;
        }
        const float inv = float(1.0 / std::sqrt(s));
        for (auto &x : v) x *= inv;
    }
    return out;
}

// what a host allocates per query for the graph searches' counters (the struct grew from 4 to 5 fields)
static_assert(sizeof(vg_search_stats) == 5 * sizeof(int64_t), "vg_search_stats is five int64: bump VG_ABI_VERSION with it");

int main()
{
    if (vg_abi_version() != VG_ABI_VERSION) {
        std::printf("libvecgo_hip.so has ABI version %d, the header %d\n", vg_abi_version(), VG_ABI_VERSION);
        return 1;
    }
    std::shared_ptr<Context> ctx;
    try {
        ctx = std::make_shared<Context>(0);
    } catch (const Error &e) {
        std::printf("no usable device: status %d (%s)\n", e.status(), e.what());
        return e.status() == VG_ERR_NO_DEVICE ? 77 : 1;
    }
    std::mt19937 rng(0);

    // ---- ProductQuantizer (pq_test.go) ----------------------------------------------------
    {
        const int dim = 128, m = 8, k = 256;
        ProductQuantizer pq(ctx, dim, m, k);
        EXPECT(!pq.IsTrained());
        auto probe = unit_vectors(rng, 1, dim)[0];
        EXPECT(status_of([&] { pq.Encode(probe); }) == VG_ERR_NOT_TRAINED);  // pq.go:148-150
        auto train = unit_vectors(rng, 1000, dim);
        pq.Train(train);
        EXPECT(pq.IsTrained());
        auto test = unit_vectors(rng, 2, dim);
        auto codes = pq.Encode(test[0]);
        EXPECT(int(codes.size()) == m);
        auto rec = pq.Decode(codes);
        EXPECT(int(rec.size()) == dim);
        double mse = 0;
        for (int i = 0; i < dim; i++) mse += double(test[0][i] - rec[i]) * (test[0][i] - rec[i]);
        EXPECT(mse / dim < 0.5);                                             // pq_test.go:58-71
        EXPECT(std::fabs(pq.CompressionRatio() - double(dim) * 4 / m) < 1e-9);
        const float adc = pq.ComputeAsymmetricDistance(test[1], codes);      // pq_test.go:84-128
        float full = 0;
        for (int i = 0; i < dim; i++) full += (test[1][i] - rec[i]) * (test[1][i] - rec[i]);
        EXPECT(std::fabs(adc - full) <= 1e-3f);
        auto table = pq.BuildDistanceTable(test[1]);
        EXPECT(int(table.size()) == m * k);
        EXPECT(std::fabs(pq.AdcDistance(table, codes) - adc) <= 1e-4f);
        EXPECT(status_of([&] { pq.Encode(std::vector<float>(100)); }) == VG_ERR_DIM_MISMATCH);
        EXPECT(status_of([&] { pq.Decode(std::vector<uint8_t>(3)); }) == VG_ERR_CODE_LENGTH);
        EXPECT(status_of([&] { ProductQuantizer bad(ctx, 100, 7, 256); }) == VG_ERR_INVALID_ARG);  // pq_test.go:130-136
        EXPECT(status_of([&] { ProductQuantizer bad(ctx, 128, 8, 257); }) == VG_ERR_INVALID_ARG);
        // SetCodebooks round trip (pq.go:452-464)
        std::vector<int8_t> cb;
        std::vector<float> sc, of;
        pq.Codebooks(cb, sc, of);
        ProductQuantizer pq2(ctx, dim, m, k);
        pq2.SetCodebooks(cb, sc, of);
        EXPECT(pq2.Encode(test[0]) == codes);
    }

    // ---- RaBitQuantizer (rabitq_test.go) --------------------------------------------------
    {
        const int dim = 128;
        RaBitQuantizer rq(ctx, dim);
        std::uniform_real_distribution<float> ud(-1.f, 1.f);
        std::vector<float> v(dim), q(dim);
        double ss = 0;
        for (auto &x : v) {
            x = ud(rng);
            ss += double(x) * x;
        }
        for (auto &x : q) x = ud(rng);
        auto code = rq.Encode(v);
        EXPECT(int64_t(code.size()) == ((dim + 63) / 64 * 8) + 4);
        float stored;
        std::memcpy(&stored, code.data() + code.size() - 4, 4);
        EXPECT(std::fabs(stored - float(std::sqrt(ss))) < 1e-5f);
        EXPECT(rq.Distance(q, code) >= 0.0f);
        EXPECT(status_of([&] { rq.Encode(std::vector<float>(dim - 1)); }) == VG_ERR_DIM_MISMATCH);
        EXPECT(status_of([&] { rq.Distance(q, std::vector<uint8_t>(5)); }) == VG_ERR_CODE_LENGTH);
    }

    // ---- ScalarQuantizer (quantizer_test.go) ---------------------------------------------------
    {
        quantization::ScalarQuantizer sq(ctx, 3);
        EXPECT(status_of([&] { sq.Encode({1.f, 2.f, 3.f}); }) == VG_ERR_NOT_TRAINED);
        sq.Train({{-1.0f, 0.0f, 1.0f}, {-0.5f, 0.5f, 2.0f}, {-2.0f, 1.0f, 3.0f}});  // quantizer_test.go:8-37
        auto mins = sq.Mins(), maxs = sq.Maxs();
        EXPECT(mins[0] == -2.0f && maxs[0] == -0.5f && mins[2] == 1.0f && maxs[2] == 3.0f);
        quantization::ScalarQuantizer sq4(ctx, 4);                                  // quantizer_test.go:230-271
        sq4.SetBounds({0, 0, 0, 0}, {10, 10, 10, 10});
        auto c1 = sq4.Encode({1, 2, 3, 4}), c2 = sq4.Encode({2, 3, 4, 5});
        std::vector<uint8_t> codes(c1);
        codes.insert(codes.end(), c2.begin(), c2.end());
        std::vector<float> out(2);
        sq4.L2DistanceBatch({1, 2, 3, 4}, codes, 2, out);
        EXPECT(out[0] <= 0.1f && std::fabs(out[1] - 4.0f) <= 0.2f);
        auto dec = sq4.Decode(sq4.Encode({-5.f, 5.f, 20.f, 10.f}));                 // clamping, :148-179
        EXPECT(dec[0] >= -0.01f && dec[2] <= 10.01f && std::fabs(dec[1] - 5.f) <= 10.f / 255.f);
        EXPECT(status_of([&] { sq4.Encode(std::vector<float>(3)); }) == VG_ERR_DIM_MISMATCH);
        EXPECT(sq4.BytesPerDimension() == 1 && sq4.CompressionRatio() == 4.0);
    }

    // ---- Int4Quantizer (int4_test.go) ------------------------------------------------------------
    {
        quantization::Int4Quantizer odd(ctx, 3);
        odd.Train({{0.1f, 0.5f, 0.9f}});                                   // int4_test.go:50-66
        auto enc = odd.Encode({0.1f, 0.5f, 0.9f});
        EXPECT(enc.size() == 2 && (enc[1] & 0x0F) == 0);
        auto dec = odd.Decode(enc);
        EXPECT(dec.size() == 3 && std::fabs(dec[0] - 0.1f) <= 0.1f && std::fabs(dec[1] - 0.5f) <= 0.1f &&
               std::fabs(dec[2] - 0.9f) <= 0.1f);
        quantization::Int4Quantizer iq(ctx, 4);
        iq.SetParams({0, 0, 0, 0}, {1, 1, 1, 1});
        auto c = iq.Encode({0.0f, 1.0f, 0.5f, 2.0f});                      // 0, 15, round(7.5) = 8, clamp -> 15
        EXPECT(c.size() == 2 && c[0] == 0x0F && c[1] == 0x8F);
        const float d1 = iq.L2Distance({0.0f, 1.0f, 8.0f / 15.0f, 1.0f}, c);
        std::vector<float> out(1);
        iq.L2DistanceBatch({0.0f, 1.0f, 8.0f / 15.0f, 1.0f}, c, 1, out);
        EXPECT(d1 < 1e-9f && out[0] < 1e-9f);
        EXPECT(status_of([&] { iq.Encode(std::vector<float>(3)); }) == VG_ERR_DIM_MISMATCH);
        EXPECT(iq.BytesPerDimension() == 0);
    }

    // ---- kmeans (kmeans_test.go) --------------------------------------------------------------
    {
        std::normal_distribution<float> nd;
        std::vector<float> x;
        for (int i = 0; i < 100; i++)
            for (int d = 0; d < 4; d++) x.push_back(nd(rng) * 0.1f + (i < 50 ? 0.f : 10.f));
        auto c = kmeans::TrainKMeans(*ctx, x, 4, 2, distance::Metric::L2, 10, 3);
        EXPECT(c.size() == 8);
        const float m0 = (c[0] + c[1] + c[2] + c[3]) / 4, m1 = (c[4] + c[5] + c[6] + c[7]) / 4;
        EXPECT(std::fabs(std::min(m0, m1)) < 0.5f && std::fabs(std::max(m0, m1) - 10.f) < 0.5f);
        EXPECT(kmeans::TrainKMeans(*ctx, std::vector<float>(4, 1.f), 4, 2, distance::Metric::L2, 10).empty());
        EXPECT(status_of([&] { kmeans::TrainKMeans(*ctx, x, 4, 2, distance::Metric::Hamming, 10); }) == VG_ERR_UNSUPPORTED);
        std::vector<float> cents = {0, 0, 1, 1, 5, 5, 10, 10};
        auto near = kmeans::FindClosestCentroids(*ctx, {0.9f, 0.9f}, cents, 2, 2, distance::Metric::L2);
        EXPECT(near.size() == 2 && near[0] == 1 && near[1] == 0);
        EXPECT(kmeans::AssignPartition(*ctx, {9.f, 9.f}, cents, 2, distance::Metric::L2) == 3);
    }

    // ---- distance.Provider -----------------------------------------------------------------------
    {
        std::vector<float> a = {1, 2, 3}, b = {4, 5, 6};
        float out = 0;
        distance::Provider(distance::Metric::L2)(*ctx, a.data(), b.data(), 3, 1, &out);
        EXPECT(out == 27.0f);  // floats_test.go:61
        distance::Provider(distance::Metric::Cosine)(*ctx, a.data(), b.data(), 3, 1, &out);
        EXPECT(out == 32.0f);  // floats_test.go:17
        EXPECT(status_of([&] { distance::Provider(distance::Metric::Hamming); }) == VG_ERR_UNSUPPORTED);
        // distance_test.go:119-152 TestMetric: String() of every metric and of an unknown one, Provider's errors
        EXPECT(distance::String(distance::Metric::L2) == "L2" && distance::String(distance::Metric::Cosine) == "Cosine");
        EXPECT(distance::String(distance::Metric::Dot) == "Dot" && distance::String(distance::Metric::Hamming) == "Hamming");
        EXPECT(distance::String(static_cast<distance::Metric>(99)) == "Unknown(99)");
        EXPECT(status_of([&] { distance::Provider(static_cast<distance::Metric>(99)); }) == VG_ERR_UNSUPPORTED);
    }

    // ---- flat segment: exact search and PQ search, vs the oracle ------------------------------
    {
        const int n = 4000, dim = 128, k = 10, nq = 5;
        std::normal_distribution<float> nd;
        std::vector<float> base(size_t(n) * dim), q(size_t(nq) * dim);
        for (auto &x : base) x = nd(rng);
        for (auto &x : q) x = nd(rng);
        Segment seg(ctx, n, dim, distance::Metric::L2);
        seg.SetVectors(base.data());
        auto r = seg.SearchFlat(q.data(), nq, k);
        for (int i = 0; i < nq; i++) {
            uint32_t eid[k];
            float esc[k];
            vgo_flat_search_f32(base.data(), n, dim, VGO_METRIC_L2, q.data() + size_t(i) * dim, k, eid, esc);
            EXPECT(std::memcmp(eid, r.ids.data() + size_t(i) * k, sizeof eid) == 0);
            EXPECT(std::memcmp(esc, r.scores.data() + size_t(i) * k, sizeof esc) == 0);
        }
        // the opt-in bfloat16 nomination filter: same ids, same fp32 scores
        {
            const int nq2 = 96;
            std::vector<float> q2(size_t(nq2) * dim);
            for (auto &x : q2) x = nd(rng);
            auto plain = seg.SearchFlat(q2.data(), nq2, k);
            seg.EnableBF16Filter(true);
            auto filtered = seg.SearchFlat(q2.data(), nq2, k);
            seg.EnableBF16Filter(false);
            EXPECT(plain.ids == filtered.ids);
            EXPECT(std::memcmp(plain.scores.data(), filtered.scores.data(), plain.scores.size() * sizeof(float)) == 0);
        }
        // flat/pq_test.go:17-93: the zero query finds the zero vector with a small score
        auto pq = std::make_shared<ProductQuantizer>(ctx, dim, 16, 256);
        std::vector<float> with_zero(base);
        std::fill(with_zero.begin() + 7 * dim, with_zero.begin() + 8 * dim, 0.0f);
        pq->TrainFlat(with_zero.data(), 1000, 20, 1);
        std::vector<uint8_t> codes(size_t(n) * 16);
        pq->EncodeBatch(with_zero.data(), n, codes.data());
        Segment pseg(ctx, n, dim, distance::Metric::L2);
        pseg.SetVectors(with_zero.data());
        pseg.SetPQCodes(pq, codes.data());
        std::vector<float> zero(dim, 0.0f);
        auto pr = pseg.SearchPQ(zero.data(), 1, 10);
        bool found = false;
        for (int i = 0; i < 10; i++) found |= pr.ids[i] == 7u;
        EXPECT(found);
        auto rr = pseg.Rerank(zero.data(), 1, pr.ids.data(), 10, 1);
        EXPECT(rr.ids[0] == 7u && rr.scores[0] == 0.0f);

        // IVF partitions (flat/segment.go:727-749): four row ranges with the first rows of each as
        // centroids; NProbes 2 scans the two closest partitions only — vs the oracle's restatement
        const int parts = 4;
        std::vector<uint32_t> off = {0, 1000, 1900, 3100, 4000};
        std::vector<float> cent(size_t(parts) * dim);
        for (int p = 0; p < parts; p++) std::copy_n(base.data() + size_t(off[p]) * dim, dim, cent.data() + size_t(p) * dim);
        seg.SetPartitions(cent.data(), off.data(), parts);
        auto probed = seg.SearchProbed(q.data(), nq, k, 2, VG_SCAN_F32);
        vgo_flat_segment os{};
        os.n = n; os.dim = dim; os.metric = VGO_METRIC_L2; os.base = base.data();
        os.num_partitions = parts; os.centroids = cent.data(); os.part_offsets = off.data();
        for (int i = 0; i < nq; i++) {
            uint32_t eid[k];
            float esc[k];
            EXPECT(vgo_flat_segment_search(&os, q.data() + size_t(i) * dim, k, 2, eid, esc) == k);
            EXPECT(std::memcmp(eid, probed.ids.data() + size_t(i) * k, sizeof eid) == 0);
            EXPECT(std::memcmp(esc, probed.scores.data() + size_t(i) * k, sizeof esc) == 0);
        }
    }

    // ---- BinaryQuantizer (binary_test.go) ---------------------------------------------------
    {
        vecgo::quantization::BinaryQuantizer bq(ctx, 128);
        std::vector<float> v(128);
        for (int i = 0; i < 128; i++) v[i] = i % 2 == 0 ? 1.0f : -1.0f;
        auto w = bq.EncodeUint64(v);                                         // binary_test.go:9-39
        EXPECT(w.size() == 2 && w[0] == 0x5555555555555555ull && w[1] == 0x5555555555555555ull);
        vecgo::quantization::BinaryQuantizer b4(ctx, 4);
        b4.Train({{1, 2, 3, 4}, {5, 6, 7, 8}});                              // binary_test.go:41-63
        EXPECT(b4.IsTrained() && b4.Threshold() == 4.5f);
        EXPECT(status_of([&] { b4.Train({}); }) == VG_ERR_INVALID_ARG);
        vecgo::quantization::BinaryQuantizer b8(ctx, 8);
        b8.WithThreshold(0.5f);                                              // binary_test.go:64-100
        auto code = b8.Encode({0.1f, 0.9f, 0.5f, 0.4f, 0.6f, 0.3f, 0.8f, 0.2f});
        EXPECT(code[0] == 0b01010110);
        auto dec = b8.Decode(code);
        EXPECT(dec[0] == 0.0f && dec[1] == 1.0f && dec[2] == 1.0f && dec[3] == 0.0f);
        std::vector<float> a(128, 1.0f), b(128, 1.0f);
        for (int i = 0; i < 10; i++) b[i * 7] = -1.0f;
        EXPECT(bq.ComputeHammingDistance(a, bq.EncodeUint64(b)) == 10);      // binary_test.go:102-123 shape
        EXPECT(bq.BytesTotal() == 16 && bq.CompressionRatio() == 32.0f);
        EXPECT(status_of([&] { bq.Encode(std::vector<float>(3)); }) == VG_ERR_DIM_MISMATCH);
    }

    // ---- NormalizeL2InPlace (distance_test.go) ----------------------------------------------
    {
        std::vector<float> v{3.0f, 4.0f};
        EXPECT(vecgo::distance::NormalizeL2InPlace(*ctx, v));
        EXPECT(std::fabs(v[0] - 0.6f) < 1e-6f && std::fabs(v[1] - 0.8f) < 1e-6f);
        std::vector<float> z{0.0f, 0.0f, 0.0f}, e;
        EXPECT(!vecgo::distance::NormalizeL2InPlace(*ctx, z) && z[0] == 0.0f);
        EXPECT(!vecgo::distance::NormalizeL2InPlace(*ctx, e));
    }

    // ---- OptimizedProductQuantizer (opq_test.go) --------------------------------------------
    {
        const int dim = 32, m = 8;
        vecgo::quantization::OptimizedProductQuantizer opq(ctx, dim, m, 16, 3);
        EXPECT(!opq.IsTrained());
        EXPECT(status_of([&] { opq.Encode(std::vector<float>(dim)); }) == VG_ERR_NOT_READY);   // "not trained" opq_test.go:202-221
        EXPECT(status_of([&] { vecgo::quantization::OptimizedProductQuantizer bad(ctx, 30, 8, 16, 1); }) == VG_ERR_INVALID_ARG);
        auto train = unit_vectors(rng, 400, dim);
        opq.Train(train);                                                    // opq_test.go:26-54
        EXPECT(opq.IsTrained());
        auto codes = opq.Encode(train[0]);
        EXPECT(int(codes.size()) == m && opq.BytesPerVector() == m);
        auto rec = opq.Decode(codes);
        EXPECT(int(rec.size()) == dim);
        EXPECT(std::fabs(opq.CompressionRatio() - 16.0) < 0.1);
        int bs = 0, nb = 0;                                                  // opq_test.go:56-99: R R^T = I
        auto rot = opq.Rotations(bs, nb);
        EXPECT(bs * nb == dim);
        for (int b = 0; b < nb; b++)
            for (int i = 0; i < bs; i++)
                for (int j = 0; j < bs; j++) {
                    float s = 0;
                    for (int k2 = 0; k2 < bs; k2++) s += rot[(size_t(b) * bs + i) * bs + k2] * rot[(size_t(b) * bs + j) * bs + k2];
                    EXPECT(std::fabs(s - (i == j ? 1.0f : 0.0f)) < 0.1f);
                }
        const float self = opq.ComputeAsymmetricDistance(train[0], codes);   // opq_test.go:101-131
        const float other = opq.ComputeAsymmetricDistance(train[1], codes);
        EXPECT(other > 0.0f && self < other);
    }

    // ---- hnsw build + search (hnsw_test.go:43-60: 1000 x 16, M = 8, EF = 200 -> precision >= 0.99) ----
    {
        const int n = 1000, dim = 16, k = 10, nq = 50;
        std::uniform_real_distribution<float> ud(0.f, 1.f);
        std::vector<float> base(size_t(n) * dim), q(size_t(nq) * dim);
        for (auto &x : base) x = ud(rng);
        for (auto &x : q) x = ud(rng);
        Segment seg(ctx, n, dim, distance::Metric::L2);
        seg.SetVectors(base.data());
        seg.BuildHNSW(8, 200, 16, 32);
        auto got = seg.SearchHNSW(q.data(), nq, k, 200);
        auto want = seg.SearchFlat(q.data(), nq, k);
        int hit = 0;
        for (int i = 0; i < nq; i++)
            for (int a = 0; a < k; a++)
                for (int b = 0; b < k; b++) hit += got.ids[size_t(i) * k + a] == want.ids[size_t(i) * k + b];
        EXPECT(hit >= int(0.99 * nq * k));
        // hnsw.BruteSearch (hnsw.go:2021-2101) and searchBitmap (:2240-2263) vs the oracle's replay of the same heaps,
        // on a tie-heavy copy of the rows (two decimals of a 16-dim uniform: equal distances do occur)
        std::vector<float> coarse(base);
        for (auto &x : coarse) x = std::floor(x * 3.0f);
        Segment tie(ctx, n, dim, distance::Metric::L2);
        tie.SetVectors(coarse.data());
        std::vector<float> cq(q);
        for (auto &x : cq) x = std::floor(x * 3.0f);
        std::vector<uint8_t> mask((n + 7) / 8);
        for (auto &b : mask) b = uint8_t(rng());
        std::vector<uint32_t> none(size_t(n), 0xFFFFFFFFu);
        vgo_hnsw_graph og{};
        og.n = n, og.dim = dim, og.metric = VGO_METRIC_L2, og.base = coarse.data(), og.m0 = 1, og.l0 = none.data();
        for (int mode = 0; mode < 2; mode++)
            for (int masked = 0; masked < 2; masked++) {
                auto br = tie.SearchHNSWBrute(cq.data(), nq, k, mode, masked ? mask.data() : nullptr);
                for (int i = 0; i < nq; i++) {
                    uint32_t eid[k];
                    float esc[k];
                    const int r = vgo_hnsw_brute_search(&og, cq.data() + size_t(i) * dim, k, mode, masked ? mask.data() : nullptr, eid, esc);
                    EXPECT(r == k);
                    EXPECT(std::memcmp(eid, br.ids.data() + size_t(i) * k, sizeof eid) == 0);
                    EXPECT(std::memcmp(esc, br.scores.data() + size_t(i) * k, sizeof esc) == 0);
                }
            }
    }

    if (g_fail) {
        std::fprintf(stderr, "%d check(s) failed\n", g_fail);
        return 1;
    }
    std::printf("host mirror: all checks passed\n");
    return 0;
}
