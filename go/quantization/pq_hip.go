//go:build hip && cgo

package quantization

// HIPProductQuantizer: quantization.Quantizer (quantizer.go:12-24) plus the ProductQuantizer extras the segment
// code uses (pq.go:234-260 ComputeAsymmetricDistance, :263 BytesPerVector, :436-446 accessors, :452-464 Codebooks /
// SetCodebooks, :468-491 BuildDistanceTable, :495-500 AdcDistance), over the C ABI.  Codebooks trained by the
// reference import through SetCodebooks; codebooks trained here differ from a reference run only because the
// reference draws from the unseeded global math/rand (pq.go:294,308,314,409).

/*
#cgo CFLAGS: -I${SRCDIR}/../../third_party/vecgo_hip/include
#cgo LDFLAGS: -L${SRCDIR}/../../third_party/vecgo_hip -lvecgo_hip
#include "vecgo_hip.h"
*/
import "C"

import (
	"errors"
	"unsafe"

	"github.com/hupe1980/vecgo/internal/hipctx"
	"github.com/hupe1980/vecgo/internal/simd"
)

const pqTrainIterations = 20 // pq.go:75 kmeans(..., 20)

type HIPProductQuantizer struct {
	h                                        *C.vg_pq
	dimension, numSubvectors, numCentroids   int
	subvectorDim                             int
}

var _ Quantizer = (*HIPProductQuantizer)(nil)

func f32(p []float32) *C.float   { return (*C.float)(unsafe.Pointer(&p[0])) }
func u8(p []byte) *C.uint8_t     { return (*C.uint8_t)(unsafe.Pointer(&p[0])) }
func i8(p []int8) *C.int8_t      { return (*C.int8_t)(unsafe.Pointer(&p[0])) }
func flatten(v [][]float32, dim int) ([]float32, error) {
	flat := make([]float32, 0, len(v)*dim)
	for _, r := range v {
		if len(r) != dim {
			return nil, hipctx.ErrDimensionMismatch
		}
		flat = append(flat, r...)
	}
	return flat, nil
}

// NewHIPProductQuantizer mirrors NewProductQuantizer (pq.go:36-64), same argument errors.
func NewHIPProductQuantizer(dimension, numSubvectors, numCentroids int) (*HIPProductQuantizer, error) {
	p, err := hipctx.Ptr()
	if err != nil {
		return nil, err
	}
	var h *C.vg_pq
	if st := C.vg_pq_create((*C.vg_ctx)(p), C.int32_t(dimension), C.int32_t(numSubvectors), C.int32_t(numCentroids), &h); st != C.VG_OK {
		return nil, errors.New(C.GoString(C.vg_last_error())) // "dimension must be divisible by numSubvectors", ...
	}
	return &HIPProductQuantizer{h: h, dimension: dimension, numSubvectors: numSubvectors, numCentroids: numCentroids,
		subvectorDim: dimension / numSubvectors}, nil
}

func (pq *HIPProductQuantizer) Close() { C.vg_pq_destroy(pq.h); pq.h = nil }

// Train: pq.go:68-143 (k-means++ init, 20 Lloyd iterations per sub-quantizer, int8 codebooks).
func (pq *HIPProductQuantizer) Train(vectors [][]float32) error {
	if len(vectors) == 0 {
		return errors.New("no training vectors provided") // pq.go:69-71
	}
	flat, err := flatten(vectors, pq.dimension)
	if err != nil {
		return err
	}
	return hipctx.Err(int32(C.vg_pq_train(pq.h, f32(flat), C.int64_t(len(vectors)), pqTrainIterations, 0, nil)))
}

// TrainRange trains sub-quantizers [begin, begin+count) only (pq.go:83-138 trains them independently): the
// multi-GPU split.  The ranges are exchanged with CodebooksRange / SetCodebooks.
func (pq *HIPProductQuantizer) TrainRange(vectors [][]float32, seed uint64, begin, count int) error {
	flat, err := flatten(vectors, pq.dimension)
	if err != nil {
		return err
	}
	return hipctx.Err(int32(C.vg_pq_train_subset(pq.h, f32(flat), C.int64_t(len(vectors)), pqTrainIterations,
		C.uint64_t(seed), C.int32_t(begin), C.int32_t(count), nil)))
}

func (pq *HIPProductQuantizer) CodebooksRange(begin, count int) ([]int8, []float32, []float32, error) {
	cb := make([]int8, count*pq.numCentroids*pq.subvectorDim)
	sc, of := make([]float32, count), make([]float32, count)
	st := C.vg_pq_get_codebooks_range(pq.h, C.int32_t(begin), C.int32_t(count), i8(cb), f32(sc), f32(of))
	return cb, sc, of, hipctx.Err(int32(st))
}

// Encode: pq.go:147-176.
func (pq *HIPProductQuantizer) Encode(vec []float32) ([]byte, error) {
	if len(vec) != pq.dimension {
		return nil, hipctx.ErrDimensionMismatch
	}
	codes := make([]byte, pq.numSubvectors)
	if st := C.vg_pq_encode(pq.h, f32(vec), 1, u8(codes), nil); st != C.VG_OK {
		return nil, hipctx.Err(int32(st))
	}
	return codes, nil
}

// EncodeBatch encodes n row-major vectors in one call (what a GPU is for).
func (pq *HIPProductQuantizer) EncodeBatch(vectors []float32, n int) ([]byte, error) {
	if len(vectors) != n*pq.dimension {
		return nil, hipctx.ErrDimensionMismatch
	}
	codes := make([]byte, n*pq.numSubvectors)
	if n == 0 {
		return codes, nil
	}
	return codes, hipctx.Err(int32(C.vg_pq_encode(pq.h, f32(vectors), C.int64_t(n), u8(codes), nil)))
}

// Decode: pq.go:185-229.
func (pq *HIPProductQuantizer) Decode(codes []byte) ([]float32, error) {
	if len(codes) != pq.numSubvectors {
		return nil, hipctx.ErrInvalidCodeLength // pq.go:190
	}
	out := make([]float32, pq.dimension)
	if st := C.vg_pq_decode(pq.h, u8(codes), 1, f32(out), nil); st != C.VG_OK {
		return nil, hipctx.Err(int32(st))
	}
	return out, nil
}

// ComputeAsymmetricDistance: pq.go:234-260 (sequential sum over the sub-quantizers).
func (pq *HIPProductQuantizer) ComputeAsymmetricDistance(query []float32, codes []byte) (float32, error) {
	if len(query) != pq.dimension {
		return 0, hipctx.ErrDimensionMismatch
	}
	if len(codes) != pq.numSubvectors {
		return 0, hipctx.ErrCodeLength
	}
	var out float32
	st := C.vg_pq_asymmetric_distance_batch(pq.h, f32(query), u8(codes), 1, (*C.float)(unsafe.Pointer(&out)), nil)
	return out, hipctx.Err(int32(st))
}

// ComputeAsymmetricDistanceBatch scores n codes against one query.
func (pq *HIPProductQuantizer) ComputeAsymmetricDistanceBatch(query []float32, codes []byte, out []float32) error {
	if len(query) != pq.dimension {
		return hipctx.ErrDimensionMismatch
	}
	if len(codes) != len(out)*pq.numSubvectors {
		return hipctx.ErrCodeLength
	}
	if len(out) == 0 {
		return nil
	}
	return hipctx.Err(int32(C.vg_pq_asymmetric_distance_batch(pq.h, f32(query), u8(codes), C.int64_t(len(out)), f32(out), nil)))
}

func (pq *HIPProductQuantizer) BytesPerDimension() int { return 0 } // PQ is sized per vector (pq.go:263)
func (pq *HIPProductQuantizer) BytesPerVector() int    { return pq.numSubvectors }
func (pq *HIPProductQuantizer) CompressionRatio() float64 {
	return float64(pq.dimension*4) / float64(pq.numSubvectors) // pq.go:268-272
}
func (pq *HIPProductQuantizer) NumSubvectors() int { return pq.numSubvectors }
func (pq *HIPProductQuantizer) NumCentroids() int  { return pq.numCentroids }
func (pq *HIPProductQuantizer) IsTrained() bool    { return C.vg_pq_is_trained(pq.h) != 0 }

// Codebooks: pq.go:452-455.
func (pq *HIPProductQuantizer) Codebooks() ([]int8, []float32, []float32) {
	cb := make([]int8, pq.numSubvectors*pq.numCentroids*pq.subvectorDim)
	sc, of := make([]float32, pq.numSubvectors), make([]float32, pq.numSubvectors)
	if C.vg_pq_get_codebooks(pq.h, i8(cb), f32(sc), f32(of)) != C.VG_OK {
		return nil, nil, nil
	}
	return cb, sc, of
}

// SetCodebooks: pq.go:457-464 — the import path for reference-trained codebooks.
func (pq *HIPProductQuantizer) SetCodebooks(codebooks []int8, scales, offsets []float32) {
	C.vg_pq_set_codebooks(pq.h, i8(codebooks), f32(scales), f32(offsets))
}

// BuildDistanceTable: pq.go:468-491 (m x 256 floats, stride 256 whatever numCentroids is).
func (pq *HIPProductQuantizer) BuildDistanceTable(query []float32) ([]float32, error) {
	if len(query) != pq.dimension {
		return nil, hipctx.ErrDimensionMismatch
	}
	table := make([]float32, pq.numSubvectors*pq.numCentroids)
	if st := C.vg_pq_build_distance_table(pq.h, f32(query), 1, f32(table), nil); st != C.VG_OK {
		return nil, hipctx.Err(int32(st))
	}
	return table, nil
}

// AdcDistance: pq.go:495-500.  One table, one code: the CPU kernel (no PCIe round trip for 96 bytes).
func (pq *HIPProductQuantizer) AdcDistance(table []float32, codes []byte) (float32, error) {
	if len(codes) != pq.numSubvectors {
		return 0, hipctx.ErrCodeLength
	}
	return simd.PqAdcLookup(table, codes, pq.numSubvectors), nil
}

// Handle exposes the vg_pq to the segment package (cgo types are per package).
func (pq *HIPProductQuantizer) Handle() unsafe.Pointer { return unsafe.Pointer(pq.h) }
