//go:build hip && cgo

package quantization

// HIPInt4Quantizer: Int4Quantizer's method set (int4.go:12-219) over the C ABI.  L2Distance is the lookup-table
// kernel's summation order (int4L2DistancePrecomputedAvx512, int4.go:140-147), L2DistanceBatch the batch kernel's
// (int4.go:152-164): `precomputed` selects it.

/*
#cgo CFLAGS: -I${SRCDIR}/../../third_party/vecgo_hip/include
#cgo LDFLAGS: -L${SRCDIR}/../../third_party/vecgo_hip -lvecgo_hip
#include "vecgo_hip.h"
*/
import "C"

import (
	"unsafe"

	"github.com/hupe1980/vecgo/internal/hipctx"
)

type HIPInt4Quantizer struct {
	h   *C.vg_int4
	dim int
}

var _ Quantizer = (*HIPInt4Quantizer)(nil)

func NewHIPInt4Quantizer(dim int) (*HIPInt4Quantizer, error) {
	p, err := hipctx.Ptr()
	if err != nil {
		return nil, err
	}
	var h *C.vg_int4
	if st := C.vg_int4_create((*C.vg_ctx)(p), C.int32_t(dim), &h); st != C.VG_OK {
		return nil, hipctx.Err(int32(st))
	}
	return &HIPInt4Quantizer{h: h, dim: dim}, nil
}

func (q *HIPInt4Quantizer) Close() { C.vg_int4_destroy(q.h); q.h = nil }

// Train: int4.go:29-65.
func (q *HIPInt4Quantizer) Train(vectors [][]float32) error {
	if len(vectors) == 0 {
		return nil
	}
	flat, err := flatten(vectors, q.dim)
	if err != nil {
		return err
	}
	return hipctx.Err(int32(C.vg_int4_train(q.h, f32(flat), C.int64_t(len(vectors)), nil)))
}

// SetParams / Params: the state MarshalBinary / UnmarshalBinary carry (int4.go:171-219).
func (q *HIPInt4Quantizer) SetParams(minVal, diff []float32) error {
	if len(minVal) != q.dim || len(diff) != q.dim {
		return hipctx.ErrDimensionMismatch
	}
	return hipctx.Err(int32(C.vg_int4_set_params(q.h, f32(minVal), f32(diff))))
}

func (q *HIPInt4Quantizer) Params() (minVal, diff []float32) {
	minVal, diff = make([]float32, q.dim), make([]float32, q.dim)
	C.vg_int4_get_params(q.h, f32(minVal), f32(diff), nil)
	return
}

func (q *HIPInt4Quantizer) codeBytes() int { return (q.dim + 1) / 2 }

// Encode: int4.go:68-106.
func (q *HIPInt4Quantizer) Encode(v []float32) ([]byte, error) {
	if len(v) != q.dim {
		return nil, hipctx.ErrDimensionMismatch
	}
	out := make([]byte, q.codeBytes())
	return out, hipctx.Err(int32(C.vg_int4_encode(q.h, f32(v), 1, u8(out), nil)))
}

// Decode: int4.go:109-133.
func (q *HIPInt4Quantizer) Decode(b []byte) ([]float32, error) {
	if len(b) != q.codeBytes() {
		return nil, hipctx.ErrInvalidCodeLength
	}
	out := make([]float32, q.dim)
	return out, hipctx.Err(int32(C.vg_int4_decode(q.h, u8(b), 1, f32(out), nil)))
}

// L2Distance: int4.go:136-149.
func (q *HIPInt4Quantizer) L2Distance(query []float32, code []byte) (float32, error) {
	if len(query) != q.dim || len(code) != q.codeBytes() {
		return 0, hipctx.ErrDimensionMismatch
	}
	var out float32
	st := C.vg_int4_l2_distance_batch(q.h, f32(query), u8(code), 1, 1, (*C.float)(unsafe.Pointer(&out)), nil)
	return out, hipctx.Err(int32(st))
}

// L2DistanceBatch: int4.go:152-164.
func (q *HIPInt4Quantizer) L2DistanceBatch(query []float32, codes []byte, n int, out []float32) error {
	if len(query) != q.dim || len(codes) < n*q.codeBytes() || len(out) < n {
		return hipctx.ErrDimensionMismatch
	}
	if n == 0 {
		return nil
	}
	return hipctx.Err(int32(C.vg_int4_l2_distance_batch(q.h, f32(query), u8(codes), C.int64_t(n), 0, f32(out), nil)))
}

func (q *HIPInt4Quantizer) BytesPerDimension() int  { return 1 } // int4.go:166-168 reports 1 (rounded up)
func (q *HIPInt4Quantizer) IsTrained() bool         { return C.vg_int4_is_trained(q.h) != 0 }
func (q *HIPInt4Quantizer) Handle() unsafe.Pointer  { return unsafe.Pointer(q.h) }
