//go:build hip && cgo

package quantization

// HIPOptimizedProductQuantizer: OptimizedProductQuantizer's method set (opq.go:28-298) over the C ABI: block
// rotations learned by alternating PQ training and a Procrustes solve (opq.go:89-194, svd.go), Encode = rotate then
// PQ-encode, Decode = PQ-decode then rotate back.

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
)

type HIPOptimizedProductQuantizer struct {
	h                        *C.vg_opq
	dimension, numSubvectors int
}

var _ Quantizer = (*HIPOptimizedProductQuantizer)(nil)

// NewHIPOptimizedProductQuantizer mirrors NewOptimizedProductQuantizer (opq.go:28-79), same block-size rule
// (vg_opq_block_size) and argument errors.
func NewHIPOptimizedProductQuantizer(dimension, numSubvectors, numCentroids, numIterations int) (*HIPOptimizedProductQuantizer, error) {
	p, err := hipctx.Ptr()
	if err != nil {
		return nil, err
	}
	var h *C.vg_opq
	st := C.vg_opq_create((*C.vg_ctx)(p), C.int32_t(dimension), C.int32_t(numSubvectors), C.int32_t(numCentroids),
		C.int32_t(numIterations), &h)
	if st != C.VG_OK {
		return nil, errors.New(C.GoString(C.vg_last_error()))
	}
	return &HIPOptimizedProductQuantizer{h: h, dimension: dimension, numSubvectors: numSubvectors}, nil
}

func (o *HIPOptimizedProductQuantizer) Close() { C.vg_opq_destroy(o.h); o.h = nil }

// BlockSize is the rotation block the constructor chose (opq.go:39-58).
func (o *HIPOptimizedProductQuantizer) BlockSize() int {
	return int(C.vg_opq_block_size(C.int32_t(o.dimension), C.int32_t(o.numSubvectors)))
}

// Train: opq.go:89-194.
func (o *HIPOptimizedProductQuantizer) Train(vectors [][]float32) error {
	if len(vectors) == 0 {
		return errors.New("no training vectors provided")
	}
	flat, err := flatten(vectors, o.dimension)
	if err != nil {
		return err
	}
	return hipctx.Err(int32(C.vg_opq_train(o.h, f32(flat), C.int64_t(len(vectors)), pqTrainIterations, 0, nil)))
}

// Rotate applies the learned block rotations to n row-major vectors (rotateVector, opq.go:196-215).
func (o *HIPOptimizedProductQuantizer) Rotate(vectors []float32, n int) ([]float32, error) {
	if len(vectors) != n*o.dimension {
		return nil, hipctx.ErrDimensionMismatch
	}
	out := make([]float32, len(vectors))
	if n == 0 {
		return out, nil
	}
	return out, hipctx.Err(int32(C.vg_opq_rotate(o.h, f32(vectors), C.int64_t(n), f32(out), nil)))
}

// Encode: opq.go:217-229.
func (o *HIPOptimizedProductQuantizer) Encode(vec []float32) ([]byte, error) {
	if len(vec) != o.dimension {
		return nil, hipctx.ErrDimensionMismatch
	}
	codes := make([]byte, o.numSubvectors)
	return codes, hipctx.Err(int32(C.vg_opq_encode(o.h, f32(vec), 1, u8(codes), nil)))
}

// Decode: opq.go:232-266.
func (o *HIPOptimizedProductQuantizer) Decode(codes []byte) ([]float32, error) {
	if len(codes) != o.numSubvectors {
		return nil, hipctx.ErrInvalidCodeLength
	}
	out := make([]float32, o.dimension)
	return out, hipctx.Err(int32(C.vg_opq_decode(o.h, u8(codes), 1, f32(out), nil)))
}

// ComputeAsymmetricDistance: opq.go:269-282 (rotate the query, then the PQ's asymmetric distance).
func (o *HIPOptimizedProductQuantizer) ComputeAsymmetricDistance(query []float32, codes []byte) (float32, error) {
	if len(query) != o.dimension {
		return 0, hipctx.ErrDimensionMismatch
	}
	if len(codes) != o.numSubvectors {
		return 0, hipctx.ErrCodeLength
	}
	var out float32
	st := C.vg_opq_asymmetric_distance_batch(o.h, f32(query), u8(codes), 1, (*C.float)(unsafe.Pointer(&out)), nil)
	return out, hipctx.Err(int32(st))
}

// Rotations / SetRotations move the learned matrices ([block][row][col], row-major) in and out.
func (o *HIPOptimizedProductQuantizer) Rotations() (block, nblocks int, rotations []float32, err error) {
	var b, nb C.int32_t
	if st := C.vg_opq_get_rotations(o.h, &b, &nb, nil); st != C.VG_OK {
		return 0, 0, nil, hipctx.Err(int32(st))
	}
	rotations = make([]float32, int(nb)*int(b)*int(b))
	st := C.vg_opq_get_rotations(o.h, &b, &nb, f32(rotations))
	return int(b), int(nb), rotations, hipctx.Err(int32(st))
}

func (o *HIPOptimizedProductQuantizer) SetRotations(rotations []float32) error {
	return hipctx.Err(int32(C.vg_opq_set_rotations(o.h, f32(rotations))))
}

func (o *HIPOptimizedProductQuantizer) BytesPerDimension() int { return 0 }
func (o *HIPOptimizedProductQuantizer) BytesPerVector() int    { return o.numSubvectors } // opq.go:285-287
func (o *HIPOptimizedProductQuantizer) CompressionRatio() float64 {
	return float64(o.dimension*4) / float64(o.numSubvectors) // opq.go:290-292
}
func (o *HIPOptimizedProductQuantizer) IsTrained() bool { return C.vg_opq_is_trained(o.h) != 0 }
