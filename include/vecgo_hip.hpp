// vecgo_hip.hpp — C++ host-side mirror of the reference's Go interfaces over the C ABI.
//
// The reference is compiled Go; no Go toolchain exists in the build image, so the host side
// above the C ABI is C++ (header-only, no dependency beyond vecgo_hip.h).  Names, argument
// meaning and error behaviour follow the reference:
//   vecgo::distance::Metric / Provider           distance/distance.go:66-116
//   vecgo::quantization::Quantizer               internal/quantization/quantizer.go:12-24
//   vecgo::quantization::ProductQuantizer        internal/quantization/pq.go:20-500
//   vecgo::quantization::RaBitQuantizer          internal/quantization/rabitq.go:26-190
//   vecgo::kmeans::TrainKMeans / AssignPartition internal/kmeans/kmeans.go:16-280
//   vecgo::Segment                               flat.Segment.Search / Rerank, hnsw.KNNSearch,
//                                                diskann searchInternal (one query batch per call)
// Go's (value, error) becomes a thrown vecgo::Error carrying the Go error string.
#pragma once

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "vecgo_hip.h"

namespace vecgo {

class Error : public std::runtime_error {
public:
    Error(int32_t status, const std::string &msg) : std::runtime_error(msg), status_(status) {}
    int32_t status() const { return status_; }

private:
    int32_t status_;
};

inline void check(int32_t status)
{
    if (status == VG_OK) return;
    const char *m = vg_last_error();
    throw Error(status, (m && *m) ? m : vg_status_string(status));
}

// One per (process, GPU).
class Context {
public:
    explicit Context(int device = 0) { check(vg_ctx_create(device, &h_)); }
    ~Context() { vg_ctx_destroy(h_); }
    Context(const Context &) = delete;
    Context &operator=(const Context &) = delete;
    vg_ctx *handle() const { return h_; }

private:
    vg_ctx *h_ = nullptr;
};

namespace distance {

// distance.Metric (distance/distance.go:66-73)
enum class Metric : int32_t { L2 = VG_METRIC_L2, Cosine = VG_METRIC_COSINE, Dot = VG_METRIC_DOT, Hamming = VG_METRIC_HAMMING };

// Metric.String (distance/distance.go:75-88), "Unknown(%d)" for anything else.  (Returns std::string since r04, when
// the "Unknown(%d)" form was added — a source break for `const char *s = String(m)`; CString keeps the C-string form
// for the four named metrics and says "Unknown" for the rest.)
inline const char *CString(Metric m)
{
    switch (m) {
    case Metric::L2: return "L2";
    case Metric::Cosine: return "Cosine";
    case Metric::Dot: return "Dot";
    case Metric::Hamming: return "Hamming";
    }
    return "Unknown";
}
inline std::string String(Metric m)
{
    switch (m) {
    case Metric::L2: return "L2";
    case Metric::Cosine: return "Cosine";
    case Metric::Dot: return "Dot";
    case Metric::Hamming: return "Hamming";
    }
    return "Unknown(" + std::to_string(static_cast<int32_t>(m)) + ")";
}

// distance.Func is a single-pair function in the reference; on the GPU the unit is one query
// against n contiguous targets (simd.SquaredL2Batch / DotBatch, internal/simd/kernels.go:61-68).
using BatchFunc = void (*)(const Context &, const float *query, const float *targets, int64_t dim, int64_t n, float *out);

inline void SquaredL2Batch(const Context &c, const float *q, const float *t, int64_t dim, int64_t n, float *out)
{
    check(vg_squared_l2_batch(c.handle(), q, t, dim, n, out, nullptr));
}
inline void DotBatch(const Context &c, const float *q, const float *t, int64_t dim, int64_t n, float *out)
{
    check(vg_dot_batch(c.handle(), q, t, dim, n, out, nullptr));
}

// distance.NormalizeL2InPlace (distance/distance.go:40-53): false for an empty or zero-norm vector (left untouched)
inline bool NormalizeL2InPlace(const Context &c, std::vector<float> &v)
{
    if (v.empty()) return false;
    uint8_t ok = 0;
    check(vg_normalize_l2(c.handle(), v.data(), 1, static_cast<int32_t>(v.size()), &ok, nullptr));
    return ok != 0;
}

// distance.Provider (distance/distance.go:91-106)
inline BatchFunc Provider(Metric m)
{
    switch (m) {
    case Metric::L2: return SquaredL2Batch;
    case Metric::Cosine:
    case Metric::Dot: return DotBatch;
    default: throw Error(VG_ERR_UNSUPPORTED, std::string("unsupported metric for float32: ") + String(m));
    }
}

}  // namespace distance

namespace quantization {

// quantization.Quantizer (internal/quantization/quantizer.go:12-24)
class Quantizer {
public:
    virtual ~Quantizer() = default;
    virtual std::vector<uint8_t> Encode(const std::vector<float> &v) = 0;
    virtual std::vector<float> Decode(const std::vector<uint8_t> &b) = 0;
    virtual void Train(const std::vector<std::vector<float>> &vectors) = 0;
    virtual int BytesPerDimension() const = 0;
};

// quantization.ProductQuantizer (pq.go:20-29)
class ProductQuantizer : public Quantizer {
public:
    // NewProductQuantizer (pq.go:36-64)
    ProductQuantizer(std::shared_ptr<Context> ctx, int dimension, int numSubvectors, int numCentroids)
        : ctx_(std::move(ctx)), dim_(dimension), m_(numSubvectors), k_(numCentroids)
    {
        check(vg_pq_create(ctx_->handle(), dimension, numSubvectors, numCentroids, &h_));
    }
    ~ProductQuantizer() override { vg_pq_destroy(h_); }
    vg_pq *handle() const { return h_; }

    // Train (pq.go:68-143); the reference runs 20 Lloyd iterations
    void Train(const std::vector<std::vector<float>> &vectors) override
    {
        if (vectors.empty()) throw Error(VG_ERR_INVALID_ARG, "no vectors provided for training");
        if (static_cast<int>(vectors[0].size()) != dim_) throw Error(VG_ERR_DIM_MISMATCH, "vector dimension mismatch");
        std::vector<float> flat;
        flat.reserve(vectors.size() * static_cast<size_t>(dim_));
        for (const auto &v : vectors) flat.insert(flat.end(), v.begin(), v.end());
        TrainFlat(flat.data(), static_cast<int64_t>(vectors.size()), 20, seed_);
    }
    void TrainFlat(const float *vectors, int64_t n, int iters, uint64_t seed)
    {
        check(vg_pq_train(h_, vectors, n, iters, seed, nullptr));
    }
    void SetSeed(uint64_t seed) { seed_ = seed; }

    // Encode (pq.go:147-176)
    std::vector<uint8_t> Encode(const std::vector<float> &vec) override
    {
        if (!IsTrained()) throw Error(VG_ERR_NOT_TRAINED, "ProductQuantizer not trained");
        if (static_cast<int>(vec.size()) != dim_) throw Error(VG_ERR_DIM_MISMATCH, "vector dimension mismatch");
        std::vector<uint8_t> codes(static_cast<size_t>(m_));
        check(vg_pq_encode(h_, vec.data(), 1, codes.data(), nullptr));
        return codes;
    }
    void EncodeBatch(const float *vectors, int64_t n, uint8_t *codes) { check(vg_pq_encode(h_, vectors, n, codes, nullptr)); }

    // Decode (pq.go:185-229)
    std::vector<float> Decode(const std::vector<uint8_t> &codes) override
    {
        if (!IsTrained()) throw Error(VG_ERR_NOT_TRAINED, "ProductQuantizer not trained");
        if (static_cast<int>(codes.size()) != m_) throw Error(VG_ERR_CODE_LENGTH, "invalid code length");
        std::vector<float> out(static_cast<size_t>(dim_));
        check(vg_pq_decode(h_, codes.data(), 1, out.data(), nullptr));
        return out;
    }

    // ComputeAsymmetricDistance (pq.go:234-260)
    float ComputeAsymmetricDistance(const std::vector<float> &query, const std::vector<uint8_t> &codes)
    {
        if (!IsTrained()) throw Error(VG_ERR_NOT_TRAINED, "ProductQuantizer not trained");
        float d = 0.0f;
        check(vg_pq_asymmetric_distance_batch(h_, query.data(), codes.data(), 1, &d, nullptr));
        return d;
    }

    // BuildDistanceTable (pq.go:468-491) / AdcDistance (pq.go:495-500)
    std::vector<float> BuildDistanceTable(const std::vector<float> &query)
    {
        if (static_cast<int>(query.size()) != dim_)
            throw Error(VG_ERR_DIM_MISMATCH, "query dimension mismatch: expected " + std::to_string(dim_) + ", got " +
                                                 std::to_string(query.size()));
        std::vector<float> table(static_cast<size_t>(m_) * k_);
        check(vg_pq_build_distance_table(h_, query.data(), 1, table.data(), nullptr));
        return table;
    }
    float AdcDistance(const std::vector<float> &table, const std::vector<uint8_t> &codes)
    {
        if (static_cast<int>(codes.size()) != m_) throw Error(VG_ERR_CODE_LENGTH, "codes length mismatch");
        float d = 0.0f;
        check(vg_pq_adc_lookup_batch(ctx_->handle(), table.data(), codes.data(), m_, 1, &d, nullptr));
        return d;
    }

    // Codebooks / SetCodebooks (pq.go:452-464)
    void SetCodebooks(const std::vector<int8_t> &codebooks, const std::vector<float> &scales, const std::vector<float> &offsets)
    {
        check(vg_pq_set_codebooks(h_, codebooks.data(), scales.data(), offsets.data()));
    }
    void Codebooks(std::vector<int8_t> &codebooks, std::vector<float> &scales, std::vector<float> &offsets)
    {
        codebooks.resize(static_cast<size_t>(m_) * k_ * (dim_ / m_));
        scales.resize(static_cast<size_t>(m_));
        offsets.resize(static_cast<size_t>(m_));
        check(vg_pq_get_codebooks(h_, codebooks.data(), scales.data(), offsets.data()));
    }

    int BytesPerDimension() const override { return 0; }
    int BytesPerVector() const { return m_; }                                         // pq.go:263-265
    double CompressionRatio() const { return static_cast<double>(dim_) * 4.0 / m_; }  // pq.go:268-272
    int NumSubvectors() const { return m_; }
    int NumCentroids() const { return k_; }
    bool IsTrained() const { return vg_pq_is_trained(h_) != 0; }

private:
    std::shared_ptr<Context> ctx_;
    vg_pq *h_ = nullptr;
    int dim_, m_, k_;
    uint64_t seed_ = 1;
};

// quantization.RaBitQuantizer (rabitq.go:26-49)
class RaBitQuantizer : public Quantizer {
public:
    RaBitQuantizer(std::shared_ptr<Context> ctx, int dimension) : ctx_(std::move(ctx)), dim_(dimension) {}
    std::vector<uint8_t> Encode(const std::vector<float> &v) override
    {
        if (static_cast<int>(v.size()) != dim_) throw Error(VG_ERR_DIM_MISMATCH, "vector dimension mismatch");
        std::vector<uint8_t> out(static_cast<size_t>(BytesTotal()));
        check(vg_rabitq_encode(ctx_->handle(), dim_, v.data(), 1, out.data(), nullptr));
        return out;
    }
    std::vector<float> Decode(const std::vector<uint8_t> &) override
    {
        throw Error(VG_ERR_UNSUPPORTED, "RaBitQuantizer.Decode is not on the hot path");
    }
    void Train(const std::vector<std::vector<float>> &) override {}  // rabitq.go:179-181: no-op
    // Distance (rabitq.go:119-176)
    float Distance(const std::vector<float> &query, const std::vector<uint8_t> &code)
    {
        if (static_cast<int64_t>(code.size()) < BytesTotal()) throw Error(VG_ERR_CODE_LENGTH, "invalid code length");
        float d = 0.0f;
        check(vg_rabitq_distance_batch(ctx_->handle(), dim_, query.data(), code.data(), 1, &d, nullptr));
        return d;
    }
    int BytesPerDimension() const override { return 0; }                   // rabitq.go:183-185
    int64_t BytesTotal() const { return vg_rabitq_code_bytes(dim_); }      // rabitq.go:187-190

private:
    std::shared_ptr<Context> ctx_;
    int dim_;
};

// quantization.BinaryQuantizer (binary.go:23-262)
class BinaryQuantizer : public Quantizer {
public:
    BinaryQuantizer(std::shared_ptr<Context> ctx, int dimension) : ctx_(std::move(ctx)), dim_(dimension) {}
    BinaryQuantizer &WithThreshold(float t)  // binary.go:52-56
    {
        threshold_ = t;
        trained_ = true;
        return *this;
    }
    void Train(const std::vector<std::vector<float>> &vectors) override  // binary.go:59-79
    {
        if (vectors.empty()) throw Error(VG_ERR_INVALID_ARG, "no vectors provided for training");
        std::vector<float> flat;
        for (const auto &v : vectors) flat.insert(flat.end(), v.begin(), v.end());
        check(vg_binary_train(ctx_->handle(), dim_, flat.data(), static_cast<int64_t>(vectors.size()), &threshold_, nullptr));
        trained_ = true;
    }
    std::vector<uint8_t> Encode(const std::vector<float> &v) override  // binary.go:86-115 (= EncodeUint64's words)
    {
        if (static_cast<int>(v.size()) != dim_) throw Error(VG_ERR_DIM_MISMATCH, "vector dimension mismatch");
        std::vector<uint8_t> out(static_cast<size_t>(vg_binary_code_bytes(dim_)));
        check(vg_binary_encode(ctx_->handle(), dim_, threshold_, v.data(), 1, out.data(), nullptr));
        return out;
    }
    std::vector<uint64_t> EncodeUint64(const std::vector<float> &v)  // binary.go:118-154
    {
        const auto bytes = Encode(v);
        std::vector<uint64_t> words(bytes.size() / 8);
        std::memcpy(words.data(), bytes.data(), bytes.size());
        return words;
    }
    std::vector<float> Decode(const std::vector<uint8_t> &b) override  // binary.go:173-188
    {
        std::vector<float> out(static_cast<size_t>(dim_));
        check(vg_binary_decode(ctx_->handle(), dim_, threshold_, b.data(), 1, static_cast<int32_t>(b.size()), out.data(), nullptr));
        return out;
    }
    int ComputeHammingDistance(const std::vector<float> &query, const std::vector<uint64_t> &codes)  // binary.go:158-171
    {
        if (static_cast<int>(query.size()) != dim_) throw Error(VG_ERR_DIM_MISMATCH, "vector dimension mismatch");
        std::vector<uint64_t> padded(static_cast<size_t>((dim_ + 63) / 64), 0);  // HammingDistance truncates to the shorter
        std::memcpy(padded.data(), codes.data(), std::min(codes.size(), padded.size()) * 8);
        int32_t d = 0;
        check(vg_binary_hamming_batch(ctx_->handle(), dim_, threshold_, query.data(),
                                      reinterpret_cast<const uint8_t *>(padded.data()), 1, &d, nullptr));
        return d;
    }
    int BytesPerDimension() const override { return 0; }     // binary.go:193-195
    int BytesTotal() const { return (dim_ + 7) / 8; }        // binary.go:198-200
    float Threshold() const { return threshold_; }
    bool IsTrained() const { return trained_; }
    float CompressionRatio() const { return 32.0f; }

private:
    std::shared_ptr<Context> ctx_;
    int dim_;
    float threshold_ = 0.0f;
    bool trained_ = false;
};

// quantization.OptimizedProductQuantizer (opq.go:15-307)
class OptimizedProductQuantizer : public Quantizer {
public:
    OptimizedProductQuantizer(std::shared_ptr<Context> ctx, int dimension, int numSubvectors, int numCentroids, int numIterations)
        : ctx_(std::move(ctx)), dim_(dimension), m_(numSubvectors)
    {
        check(vg_opq_create(ctx_->handle(), dimension, numSubvectors, numCentroids, numIterations, &h_));
    }
    ~OptimizedProductQuantizer() override { vg_opq_destroy(h_); }
    OptimizedProductQuantizer(const OptimizedProductQuantizer &) = delete;
    OptimizedProductQuantizer &operator=(const OptimizedProductQuantizer &) = delete;
    void Train(const std::vector<std::vector<float>> &vectors) override  // opq.go:89-193
    {
        if (vectors.empty()) throw Error(VG_ERR_INVALID_ARG, "no vectors provided for training");
        if (static_cast<int>(vectors[0].size()) != dim_) throw Error(VG_ERR_DIM_MISMATCH, "vector dimension mismatch");
        std::vector<float> flat;
        for (const auto &v : vectors) flat.insert(flat.end(), v.begin(), v.end());
        check(vg_opq_train(h_, flat.data(), static_cast<int64_t>(vectors.size()), 20, 1, nullptr));
    }
    std::vector<uint8_t> Encode(const std::vector<float> &v) override  // opq.go:218-231
    {
        if (static_cast<int>(v.size()) != dim_) throw Error(VG_ERR_DIM_MISMATCH, "vector dimension mismatch");
        std::vector<uint8_t> out(static_cast<size_t>(m_));
        check(vg_opq_encode(h_, v.data(), 1, out.data(), nullptr));
        return out;
    }
    std::vector<float> Decode(const std::vector<uint8_t> &b) override  // opq.go:234-269
    {
        if (static_cast<int>(b.size()) != m_) throw Error(VG_ERR_CODE_LENGTH, "invalid code length");
        std::vector<float> out(static_cast<size_t>(dim_));
        check(vg_opq_decode(h_, b.data(), 1, out.data(), nullptr));
        return out;
    }
    float ComputeAsymmetricDistance(const std::vector<float> &query, const std::vector<uint8_t> &codes)  // opq.go:272-286
    {
        float d = 0.0f;
        check(vg_opq_asymmetric_distance_batch(h_, query.data(), codes.data(), 1, &d, nullptr));
        return d;
    }
    std::vector<float> Rotations(int &blockSize, int &numBlocks) const
    {
        int32_t b = 0, nb = 0;
        check(vg_opq_get_rotations(h_, &b, &nb, nullptr));
        std::vector<float> r(static_cast<size_t>(nb) * b * b);
        check(vg_opq_get_rotations(h_, nullptr, nullptr, r.data()));
        blockSize = b;
        numBlocks = nb;
        return r;
    }
    int BytesPerDimension() const override { return 0; }
    int BytesPerVector() const { return m_; }                               // opq.go:289-291
    double CompressionRatio() const { return double(dim_) * 4.0 / m_; }     // opq.go:294-296
    bool IsTrained() const { return vg_opq_is_trained(h_) != 0; }

private:
    std::shared_ptr<Context> ctx_;
    vg_opq *h_ = nullptr;
    int dim_, m_;
};

// quantization.ScalarQuantizer (quantizer.go:27-39)
class ScalarQuantizer : public Quantizer {
public:
    // NewScalarQuantizer (quantizer.go:119-125)
    ScalarQuantizer(std::shared_ptr<Context> ctx, int dimension) : ctx_(std::move(ctx)), dim_(dimension)
    {
        check(vg_sq8_create(ctx_->handle(), dimension, &h_));
    }
    ~ScalarQuantizer() override { vg_sq8_destroy(h_); }
    vg_sq8 *handle() const { return h_; }
    bool IsTrained() const { return vg_sq8_is_trained(h_) != 0; }

    // Train (quantizer.go:127-180)
    void Train(const std::vector<std::vector<float>> &vectors) override
    {
        if (vectors.empty()) throw Error(VG_ERR_INVALID_ARG, "no vectors provided for training");
        if (static_cast<int>(vectors[0].size()) != dim_) throw Error(VG_ERR_DIM_MISMATCH, "vector dimension mismatch");
        std::vector<float> flat;
        flat.reserve(vectors.size() * static_cast<size_t>(dim_));
        for (const auto &v : vectors) {
            if (static_cast<int>(v.size()) != dim_) throw Error(VG_ERR_DIM_MISMATCH, "inconsistent vector dimension");
            flat.insert(flat.end(), v.begin(), v.end());
        }
        check(vg_sq8_train(h_, flat.data(), static_cast<int64_t>(vectors.size()), nullptr));
    }
    // SetBounds (quantizer.go:52-78)
    void SetBounds(const std::vector<float> &mins, const std::vector<float> &maxs)
    {
        if (static_cast<int>(mins.size()) != dim_ || static_cast<int>(maxs.size()) != dim_)
            throw Error(VG_ERR_DIM_MISMATCH, "dimension mismatch");
        check(vg_sq8_set_bounds(h_, mins.data(), maxs.data()));
    }
    std::vector<float> Mins() const
    {
        std::vector<float> v(static_cast<size_t>(dim_));
        check(vg_sq8_get_params(h_, v.data(), nullptr, nullptr, nullptr));
        return v;
    }
    std::vector<float> Maxs() const
    {
        std::vector<float> v(static_cast<size_t>(dim_));
        check(vg_sq8_get_params(h_, nullptr, v.data(), nullptr, nullptr));
        return v;
    }
    // Encode / Decode (quantizer.go:183-250)
    std::vector<uint8_t> Encode(const std::vector<float> &v) override
    {
        if (!IsTrained()) throw Error(VG_ERR_NOT_TRAINED, "ScalarQuantizer not trained");
        if (static_cast<int>(v.size()) != dim_) throw Error(VG_ERR_DIM_MISMATCH, "vector dimension mismatch");
        std::vector<uint8_t> out(static_cast<size_t>(dim_));
        check(vg_sq8_encode(h_, v.data(), 1, out.data(), nullptr));
        return out;
    }
    std::vector<float> Decode(const std::vector<uint8_t> &b) override
    {
        if (!IsTrained()) throw Error(VG_ERR_NOT_TRAINED, "ScalarQuantizer not trained");
        if (static_cast<int>(b.size()) != dim_) throw Error(VG_ERR_DIM_MISMATCH, "vector dimension mismatch");
        std::vector<float> out(static_cast<size_t>(dim_));
        check(vg_sq8_decode(h_, b.data(), 1, out.data(), nullptr));
        return out;
    }
    // L2DistanceBatch (quantizer.go:93-106)
    void L2DistanceBatch(const std::vector<float> &q, const std::vector<uint8_t> &codes, int n, std::vector<float> &out)
    {
        if (static_cast<int>(q.size()) != dim_) throw Error(VG_ERR_DIM_MISMATCH, "query dimension mismatch");
        if (static_cast<int64_t>(codes.size()) < static_cast<int64_t>(n) * dim_)
            throw Error(VG_ERR_INVALID_ARG, "codes buffer too small");
        if (static_cast<int>(out.size()) < n) throw Error(VG_ERR_INVALID_ARG, "output buffer too small");
        check(vg_sq8_l2_distance_batch(h_, q.data(), codes.data(), n, out.data(), nullptr));
    }
    int BytesPerDimension() const override { return 1; }
    double CompressionRatio() const { return 4.0; }

private:
    std::shared_ptr<Context> ctx_;
    int dim_;
    vg_sq8 *h_ = nullptr;
};

// quantization.Int4Quantizer (int4.go:12-20)
class Int4Quantizer : public Quantizer {
public:
    Int4Quantizer(std::shared_ptr<Context> ctx, int dimension) : ctx_(std::move(ctx)), dim_(dimension)
    {
        check(vg_int4_create(ctx_->handle(), dimension, &h_));
    }
    ~Int4Quantizer() override { vg_int4_destroy(h_); }
    vg_int4 *handle() const { return h_; }
    bool IsTrained() const { return vg_int4_is_trained(h_) != 0; }

    // Train (int4.go:29-62)
    void Train(const std::vector<std::vector<float>> &vectors) override
    {
        if (vectors.empty()) throw Error(VG_ERR_INVALID_ARG, "no vectors provided for training");
        std::vector<float> flat;
        flat.reserve(vectors.size() * static_cast<size_t>(dim_));
        for (const auto &v : vectors) {
            if (static_cast<int>(v.size()) != dim_) throw Error(VG_ERR_DIM_MISMATCH, "dimension mismatch");
            flat.insert(flat.end(), v.begin(), v.end());
        }
        check(vg_int4_train(h_, flat.data(), static_cast<int64_t>(vectors.size()), nullptr));
    }
    // UnmarshalBinary (int4.go:190-219): min[dim] then diff[dim]
    void SetParams(const std::vector<float> &minVal, const std::vector<float> &diff)
    {
        if (static_cast<int>(minVal.size()) != dim_ || static_cast<int>(diff.size()) != dim_)
            throw Error(VG_ERR_DIM_MISMATCH, "data size mismatch");
        check(vg_int4_set_params(h_, minVal.data(), diff.data()));
    }
    // Encode / Decode (int4.go:65-130)
    std::vector<uint8_t> Encode(const std::vector<float> &v) override
    {
        if (static_cast<int>(v.size()) != dim_) throw Error(VG_ERR_DIM_MISMATCH, "dimension mismatch");
        std::vector<uint8_t> out(static_cast<size_t>((dim_ + 1) / 2));
        check(vg_int4_encode(h_, v.data(), 1, out.data(), nullptr));
        return out;
    }
    std::vector<float> Decode(const std::vector<uint8_t> &b) override
    {
        if (static_cast<int>(b.size()) != (dim_ + 1) / 2) throw Error(VG_ERR_DIM_MISMATCH, "dimension mismatch");
        std::vector<float> out(static_cast<size_t>(dim_));
        check(vg_int4_decode(h_, b.data(), 1, out.data(), nullptr));
        return out;
    }
    // L2Distance (int4.go:133-147: lookup-table kernel) / L2DistanceBatch (int4.go:150-164)
    float L2Distance(const std::vector<float> &query, const std::vector<uint8_t> &code)
    {
        if (static_cast<int>(query.size()) != dim_ || static_cast<int>(code.size()) != (dim_ + 1) / 2)
            throw Error(VG_ERR_DIM_MISMATCH, "dimension mismatch");
        float d = 0.0f;
        check(vg_int4_l2_distance_batch(h_, query.data(), code.data(), 1, 1, &d, nullptr));
        return d;
    }
    void L2DistanceBatch(const std::vector<float> &query, const std::vector<uint8_t> &codes, int n, std::vector<float> &out)
    {
        if (static_cast<int64_t>(codes.size()) < static_cast<int64_t>(n) * ((dim_ + 1) / 2))
            throw Error(VG_ERR_INVALID_ARG, "codes buffer too small");
        if (static_cast<int>(out.size()) < n) throw Error(VG_ERR_INVALID_ARG, "output buffer too small");
        check(vg_int4_l2_distance_batch(h_, query.data(), codes.data(), n, 0, out.data(), nullptr));
    }
    int BytesPerDimension() const override { return 0; }  // sub-byte (int4.go:167-169)

private:
    std::shared_ptr<Context> ctx_;
    int dim_;
    vg_int4 *h_ = nullptr;
};

}  // namespace quantization

namespace kmeans {

// TrainKMeans (kmeans.go:16-138): empty result when n < k (the reference returns (nil, nil))
inline std::vector<float> TrainKMeans(const Context &c, const std::vector<float> &vectors, int dim, int k,
                                      distance::Metric metric, int maxIter, uint64_t seed = 1)
{
    std::vector<float> cent(static_cast<size_t>(k) * dim);
    int32_t produced = 0;
    check(vg_kmeans_train(c.handle(), vectors.data(), static_cast<int64_t>(vectors.size() / dim), dim, k,
                          static_cast<int32_t>(metric), maxIter, seed, cent.data(), &produced, nullptr));
    if (!produced) cent.clear();
    return cent;
}

// AssignPartition (kmeans.go:142-196)
inline int AssignPartition(const Context &c, const std::vector<float> &vec, const std::vector<float> &centroids, int dim,
                           distance::Metric metric)
{
    int32_t out = -1;
    check(vg_kmeans_assign(c.handle(), vec.data(), 1, dim, centroids.data(), static_cast<int32_t>(centroids.size() / dim),
                           static_cast<int32_t>(metric), &out, nullptr));
    return out;
}

// FindClosestCentroids (kmeans.go:217-280)
inline std::vector<int> FindClosestCentroids(const Context &c, const std::vector<float> &query,
                                             const std::vector<float> &centroids, int dim, int n, distance::Metric metric)
{
    const int k = static_cast<int>(centroids.size() / dim);
    std::vector<int32_t> out(static_cast<size_t>(std::max(1, std::min(n, k))));
    int32_t cnt = 0;
    check(vg_find_closest_centroids(c.handle(), query.data(), centroids.data(), dim, k, n, static_cast<int32_t>(metric),
                                    out.data(), &cnt, nullptr));
    return std::vector<int>(out.begin(), out.begin() + cnt);
}

}  // namespace kmeans

struct Result {
    std::vector<uint32_t> ids;   // [nq * k], best first, VG_INVALID_ID padded
    std::vector<float> scores;   // [nq * k]
};

// One resident segment (flat / memtable-HNSW / DiskANN): the batch-level search entry points.
class Segment {
public:
    Segment(std::shared_ptr<Context> ctx, int64_t rows, int dim, distance::Metric metric)
        : ctx_(std::move(ctx)), n_(rows), dim_(dim)
    {
        check(vg_index_create(ctx_->handle(), rows, dim, static_cast<int32_t>(metric), &h_));
    }
    ~Segment() { vg_index_destroy(h_); }
    Segment(const Segment &) = delete;
    Segment &operator=(const Segment &) = delete;

    void SetVectors(const float *base) { check(vg_index_set_vectors(h_, base, nullptr)); }
    void SetPQCodes(std::shared_ptr<quantization::ProductQuantizer> pq, const uint8_t *codes)
    {
        pq_ = std::move(pq);
        check(vg_index_set_pq_codes(h_, pq_->handle(), codes, nullptr));
    }
    void SetRaBitQCodes(const uint8_t *codes) { check(vg_index_set_rabitq_codes(h_, codes, nullptr)); }
    void SetVamanaGraph(int r, const uint32_t *graph, uint32_t entry) { check(vg_index_set_vamana_graph(h_, r, graph, entry, nullptr)); }
    void SetHNSWLayer0(int m0, const uint32_t *l0, uint32_t entry)
    {
        check(vg_index_set_hnsw_graph(h_, m0, l0, 0, m0 / 2 > 0 ? m0 / 2 : 1, nullptr, nullptr, nullptr, entry, nullptr));
    }

    // flat.Segment.Search fp32 branch (flat/segment.go:691-721)
    Result SearchFlat(const float *queries, int64_t nq, int k) { return run(nq, k, [&](Result &r) { return vg_search_flat(h_, queries, nq, k, r.ids.data(), r.scores.data(), nullptr); }); }
    // opt-in: nominate with a bfloat16 MFMA GEMM over a bf16 copy of the rows; results stay bit-identical (vecgo_hip.h)
    void EnableBF16Filter(bool on = true) { check(vg_index_enable_bf16_filter(h_, on ? 1 : 0, nullptr)); }
    // flat.Segment.Search PQ branch (flat/segment.go:476-483,678-689)
    Result SearchPQ(const float *queries, int64_t nq, int k) { return run(nq, k, [&](Result &r) { return vg_search_pq_adc(h_, queries, nq, k, r.ids.data(), r.scores.data(), nullptr); }); }
    // opt-in: batches of SearchPQ nominated by a bfloat16 MFMA GEMM over the decoded rows, re-scored from the codes; results unchanged
    void EnablePQNomination(bool on = true) { check(vg_index_enable_pq_nomination(h_, on ? 1 : 0, nullptr)); }
    Result SearchRaBitQ(const float *queries, int64_t nq, int k) { return run(nq, k, [&](Result &r) { return vg_search_rabitq(h_, queries, nq, k, r.ids.data(), r.scores.data(), nullptr); }); }
    // IVF partitions (flat/segment.go:187-207) and the probed scan of flat.Segment.Search (:727-749):
    // scan = VG_SCAN_F32 / VG_SCAN_PQ / VG_SCAN_SQ8, nprobes <= 0 means 1 as in the reference
    void SetPartitions(const float *centroids, const uint32_t *part_offsets, int num_partitions)
    {
        check(vg_index_set_partitions(h_, centroids, part_offsets, num_partitions, nullptr));
    }
    // Segment.Search with a row filter (flat/segment.go:631-635): mask bit i of byte i/8 = filter.Matches(i)
    Result SearchFiltered(const float *queries, int64_t nq, int k, int nprobes, int scan, const uint8_t *mask, int64_t mask_stride) { return run(nq, k, [&](Result &r) { return vg_search_flat_filtered(h_, queries, nq, k, nprobes, scan, mask, mask_stride, r.ids.data(), r.scores.data(), nullptr); }); }
    Result SearchProbed(const float *queries, int64_t nq, int k, int nprobes, int scan) { return run(nq, k, [&](Result &r) { return vg_search_flat_probed(h_, queries, nq, k, nprobes, scan, r.ids.data(), r.scores.data(), nullptr); }); }
    // hnsw.Insert over the segment's rows (hnsw.go:713-984, ids and levels of ApplyInsert); replaces the segment's graph
    void BuildHNSW(int m = 32, int efConstruction = 300, int maxBatch = 8192, int growthDiv = 32)
    {
        check(vg_hnsw_build(h_, m, efConstruction, maxBatch, growthDiv, nullptr));
    }
    // the graph walk scored from PQ codes (candidates for Rerank, engine/search.go:914-965)
    Result SearchHNSWPQ(const float *queries, int64_t nq, int k, int ef) { return run(nq, k, [&](Result &r) { return vg_search_hnsw_pq(h_, queries, nq, k, ef, r.ids.data(), r.scores.data(), nullptr, nullptr); }); }
    // hnsw.KNNSearch (hnsw.go:1650-1755)
    Result SearchHNSW(const float *queries, int64_t nq, int k, int ef) { return run(nq, k, [&](Result &r) { return vg_search_hnsw(h_, queries, nq, k, ef, r.ids.data(), r.scores.data(), nullptr, nullptr); }); }
    // hnsw.BruteSearch + scanSegment (hnsw.go:2021-2101; VG_BRUTE_SCAN) / searchBitmap (:2240-2263; VG_BRUTE_BITMAP)
    Result SearchHNSWBrute(const float *queries, int64_t nq, int k, int mode, const uint8_t *mask = nullptr, int64_t mask_stride = 0) { return run(nq, k, [&](Result &r) { return vg_search_hnsw_brute(h_, queries, nq, k, mode, mask, mask_stride, r.ids.data(), r.scores.data(), nullptr); }); }
    // diskann searchInternal (diskann/segment.go:503-706); kind 0 fp32, 1 PQ, 2 RaBitQ
    // searchInternal with a row filter (diskann/segment.go:616-627): mask bit i of byte i/8 = filter.Matches(i)
    Result SearchVamanaFiltered(const float *queries, int64_t nq, int k, int kind, const uint8_t *mask, int64_t mask_stride) { return run(nq, k, [&](Result &r) { return vg_search_vamana_filtered(h_, queries, nq, k, kind, mask, mask_stride, r.ids.data(), r.scores.data(), nullptr, nullptr); }); }
    Result SearchVamana(const float *queries, int64_t nq, int k, int kind) { return run(nq, k, [&](Result &r) { return vg_search_vamana(h_, queries, nq, k, kind, r.ids.data(), r.scores.data(), nullptr, nullptr); }); }
    // Segment.Rerank (flat/segment.go:754-780) + top-k
    Result Rerank(const float *queries, int64_t nq, const uint32_t *cand, int nc, int k) { return run(nq, k, [&](Result &r) { return vg_rerank(h_, queries, nq, cand, nc, k, r.ids.data(), r.scores.data(), nullptr); }); }

private:
    template <typename F>
    Result run(int64_t nq, int k, F f)
    {
        Result r;
        r.ids.resize(static_cast<size_t>(nq) * k);
        r.scores.resize(static_cast<size_t>(nq) * k);
        check(f(r));
        return r;
    }
    std::shared_ptr<Context> ctx_;
    std::shared_ptr<quantization::ProductQuantizer> pq_;
    vg_index *h_ = nullptr;
    int64_t n_;
    int dim_;
};

}  // namespace vecgo
