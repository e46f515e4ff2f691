//go:build hip && cgo

package quantization

// HIPBinaryQuantizer: BinaryQuantizer's method set (binary.go:23-262) over the C ABI.  Codes are ceil(dim/64)
// little-endian uint64 words (bit i = v[i] >= threshold).

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

type HIPBinaryQuantizer struct {
	ctx       *C.vg_ctx
	dimension int
	threshold float32
	trained   bool
}

var _ Quantizer = (*HIPBinaryQuantizer)(nil)

func NewHIPBinaryQuantizer(dimension int) (*HIPBinaryQuantizer, error) {
	p, err := hipctx.Ptr()
	if err != nil {
		return nil, err
	}
	return &HIPBinaryQuantizer{ctx: (*C.vg_ctx)(p), dimension: dimension}, nil
}

// WithThreshold: binary.go:51-55.
func (bq *HIPBinaryQuantizer) WithThreshold(threshold float32) *HIPBinaryQuantizer {
	bq.threshold, bq.trained = threshold, true
	return bq
}

// Train: binary.go:59-83 (threshold = mean of every component).
func (bq *HIPBinaryQuantizer) Train(vectors [][]float32) error {
	if len(vectors) == 0 {
		return errors.New("no training vectors provided")
	}
	flat, err := flatten(vectors, bq.dimension)
	if err != nil {
		return err
	}
	var th C.float
	if st := C.vg_binary_train(bq.ctx, C.int32_t(bq.dimension), f32(flat), C.int64_t(len(vectors)), &th, nil); st != C.VG_OK {
		return hipctx.Err(int32(st))
	}
	bq.threshold, bq.trained = float32(th), true
	return nil
}

func (bq *HIPBinaryQuantizer) words() int { return (bq.dimension + 63) / 64 }

// EncodeUint64Into: binary.go:130-154.
func (bq *HIPBinaryQuantizer) EncodeUint64Into(dst []uint64, v []float32) error {
	if len(v) != bq.dimension {
		return hipctx.ErrDimensionMismatch
	}
	if len(dst) < bq.words() {
		return errors.New("destination buffer too small")
	}
	st := C.vg_binary_encode(bq.ctx, C.int32_t(bq.dimension), C.float(bq.threshold), f32(v), 1,
		(*C.uint8_t)(unsafe.Pointer(&dst[0])), nil)
	return hipctx.Err(int32(st))
}

// EncodeUint64: binary.go:115-127.
func (bq *HIPBinaryQuantizer) EncodeUint64(v []float32) ([]uint64, error) {
	dst := make([]uint64, bq.words())
	return dst, bq.EncodeUint64Into(dst, v)
}

// Encode: binary.go:87-112 — the first ceil(dim/8) bytes of the words.
func (bq *HIPBinaryQuantizer) Encode(v []float32) ([]byte, error) {
	w, err := bq.EncodeUint64(v)
	if err != nil {
		return nil, err
	}
	full := unsafe.Slice((*byte)(unsafe.Pointer(&w[0])), len(w)*8)
	out := make([]byte, bq.BytesTotal())
	copy(out, full)
	return out, nil
}

// EncodeBatch encodes n row-major vectors into n * words() uint64 in one call.
func (bq *HIPBinaryQuantizer) EncodeBatch(vectors []float32, n int) ([]uint64, error) {
	if len(vectors) != n*bq.dimension {
		return nil, hipctx.ErrDimensionMismatch
	}
	dst := make([]uint64, n*bq.words())
	if n == 0 {
		return dst, nil
	}
	st := C.vg_binary_encode(bq.ctx, C.int32_t(bq.dimension), C.float(bq.threshold), f32(vectors), C.int64_t(n),
		(*C.uint8_t)(unsafe.Pointer(&dst[0])), nil)
	return dst, hipctx.Err(int32(st))
}

// Decode: binary.go:175-190 (threshold +- 0.5; bits beyond len(b) read as 0).
func (bq *HIPBinaryQuantizer) Decode(b []byte) ([]float32, error) {
	out := make([]float32, bq.dimension)
	if len(b) == 0 {
		for i := range out {
			out[i] = bq.threshold - 0.5
		}
		return out, nil
	}
	st := C.vg_binary_decode(bq.ctx, C.int32_t(bq.dimension), C.float(bq.threshold), u8(b), 1, C.int32_t(len(b)), f32(out), nil)
	return out, hipctx.Err(int32(st))
}

// ComputeHammingDistance: binary.go:158-171.
func (bq *HIPBinaryQuantizer) ComputeHammingDistance(query []float32, codes []uint64) (int, error) {
	if len(query) != bq.dimension {
		return 0, hipctx.ErrDimensionMismatch
	}
	var out C.int32_t
	st := C.vg_binary_hamming_batch(bq.ctx, C.int32_t(bq.dimension), C.float(bq.threshold), f32(query),
		(*C.uint8_t)(unsafe.Pointer(&codes[0])), 1, &out, nil)
	return int(out), hipctx.Err(int32(st))
}

// ComputeHammingDistanceBatch scores n codes of words() uint64 each against one float query.
func (bq *HIPBinaryQuantizer) ComputeHammingDistanceBatch(query []float32, codes []uint64, out []int32) error {
	if len(query) != bq.dimension || len(codes) < len(out)*bq.words() {
		return hipctx.ErrDimensionMismatch
	}
	if len(out) == 0 {
		return nil
	}
	st := C.vg_binary_hamming_batch(bq.ctx, C.int32_t(bq.dimension), C.float(bq.threshold), f32(query),
		(*C.uint8_t)(unsafe.Pointer(&codes[0])), C.int64_t(len(out)), (*C.int32_t)(unsafe.Pointer(&out[0])), nil)
	return hipctx.Err(int32(st))
}

func (bq *HIPBinaryQuantizer) BytesPerDimension() int { return 0 }                         // binary.go:194-196
func (bq *HIPBinaryQuantizer) BytesTotal() int        { return (bq.dimension + 7) / 8 }    // binary.go:199-201
func (bq *HIPBinaryQuantizer) WordBytes() int         { return int(C.vg_binary_code_bytes(C.int32_t(bq.dimension))) }
func (bq *HIPBinaryQuantizer) Dimension() int         { return bq.dimension }
func (bq *HIPBinaryQuantizer) Threshold() float32     { return bq.threshold }
func (bq *HIPBinaryQuantizer) IsTrained() bool        { return bq.trained }
func (bq *HIPBinaryQuantizer) CompressionRatio() float32 {
	return float32(bq.dimension*4) / float32(bq.BytesTotal()) // binary.go:260-262
}
