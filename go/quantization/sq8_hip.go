//go:build hip && cgo

package quantization

// HIPScalarQuantizer: ScalarQuantizer's method set (quantizer.go:27-339) over the C ABI.

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

type HIPScalarQuantizer struct {
	h         *C.vg_sq8
	dimension int
}

var _ Quantizer = (*HIPScalarQuantizer)(nil)

// NewHIPScalarQuantizer mirrors NewScalarQuantizer (quantizer.go:122-127).
func NewHIPScalarQuantizer(dimension int) (*HIPScalarQuantizer, error) {
	p, err := hipctx.Ptr()
	if err != nil {
		return nil, err
	}
	var h *C.vg_sq8
	if st := C.vg_sq8_create((*C.vg_ctx)(p), C.int32_t(dimension), &h); st != C.VG_OK {
		return nil, hipctx.Err(int32(st))
	}
	return &HIPScalarQuantizer{h: h, dimension: dimension}, nil
}

func (sq *HIPScalarQuantizer) Close() { C.vg_sq8_destroy(sq.h); sq.h = nil }

// Train: quantizer.go:130-181 (per-dimension min / max).
func (sq *HIPScalarQuantizer) Train(vectors [][]float32) error {
	if len(vectors) == 0 {
		return nil
	}
	flat, err := flatten(vectors, sq.dimension)
	if err != nil {
		return err
	}
	return hipctx.Err(int32(C.vg_sq8_train(sq.h, f32(flat), C.int64_t(len(vectors)), nil)))
}

// SetBounds: quantizer.go:51-75.
func (sq *HIPScalarQuantizer) SetBounds(mins, maxs []float32) error {
	if len(mins) != sq.dimension || len(maxs) != sq.dimension {
		return errors.New("dimension mismatch")
	}
	return hipctx.Err(int32(C.vg_sq8_set_bounds(sq.h, f32(mins), f32(maxs))))
}

func (sq *HIPScalarQuantizer) params() (mins, maxs, scales, inv []float32) {
	mins, maxs = make([]float32, sq.dimension), make([]float32, sq.dimension)
	scales, inv = make([]float32, sq.dimension), make([]float32, sq.dimension)
	C.vg_sq8_get_params(sq.h, f32(mins), f32(maxs), f32(scales), f32(inv))
	return
}

// Mins / Maxs: quantizer.go:41-48.
func (sq *HIPScalarQuantizer) Mins() []float32 { m, _, _, _ := sq.params(); return m }
func (sq *HIPScalarQuantizer) Maxs() []float32 { _, m, _, _ := sq.params(); return m }

// Encode: quantizer.go:184-197.
func (sq *HIPScalarQuantizer) Encode(v []float32) ([]byte, error) {
	if len(v) != sq.dimension {
		return nil, hipctx.ErrDimensionMismatch
	}
	out := make([]byte, sq.dimension)
	return out, hipctx.Err(int32(C.vg_sq8_encode(sq.h, f32(v), 1, u8(out), nil)))
}

// Decode: quantizer.go:225-238.
func (sq *HIPScalarQuantizer) Decode(b []byte) ([]float32, error) {
	if len(b) != sq.dimension {
		return nil, hipctx.ErrInvalidCodeLength
	}
	out := make([]float32, sq.dimension)
	return out, hipctx.Err(int32(C.vg_sq8_decode(sq.h, u8(b), 1, f32(out), nil)))
}

// L2DistanceBatch: quantizer.go:94-106 (simd.Sq8uL2BatchPerDimension order).
func (sq *HIPScalarQuantizer) L2DistanceBatch(q []float32, codes []byte, n int, out []float32) error {
	if len(q) != sq.dimension || len(codes) < n*sq.dimension || len(out) < n {
		return hipctx.ErrDimensionMismatch
	}
	if n == 0 {
		return nil
	}
	return hipctx.Err(int32(C.vg_sq8_l2_distance_batch(sq.h, f32(q), u8(codes), C.int64_t(n), f32(out), nil)))
}

// L2Distance: quantizer.go:78-91 — the batch kernel with n = 1.
func (sq *HIPScalarQuantizer) L2Distance(q []float32, code []byte) (float32, error) {
	var out [1]float32
	err := sq.L2DistanceBatch(q, code, 1, out[:])
	return out[0], err
}

func (sq *HIPScalarQuantizer) BytesPerDimension() int { return 1 } // quantizer.go:251-253
func (sq *HIPScalarQuantizer) IsTrained() bool        { return C.vg_sq8_is_trained(sq.h) != 0 }
func (sq *HIPScalarQuantizer) Handle() unsafe.Pointer { return unsafe.Pointer(sq.h) }
