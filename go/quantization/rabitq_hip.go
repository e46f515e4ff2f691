//go:build hip && cgo

package quantization

// HIPRaBitQuantizer: quantization.Quantizer (quantizer.go:12-24) + Distance / BytesTotal (rabitq.go:119-176,
// :187-190) over the C ABI.  Codes are the reference's: sign bits (bit i -> byte i/8, bit i%8, padded to 8-byte
// words) followed by the little-endian float32 norm.

/*
#cgo CFLAGS: -I${SRCDIR}/../../third_party/vecgo_hip/include
#cgo LDFLAGS: -L${SRCDIR}/../../third_party/vecgo_hip -lvecgo_hip
#include "vecgo_hip.h"
*/
import "C"

import (
	"encoding/binary"
	"errors"
	"math"
	"unsafe"

	"github.com/hupe1980/vecgo/internal/hipctx"
)

type HIPRaBitQuantizer struct {
	ctx       *C.vg_ctx
	dimension int
}

var _ Quantizer = (*HIPRaBitQuantizer)(nil)

func NewHIPRaBitQuantizer(dimension int) (*HIPRaBitQuantizer, error) {
	p, err := hipctx.Ptr()
	if err != nil {
		return nil, err
	}
	return &HIPRaBitQuantizer{ctx: (*C.vg_ctx)(p), dimension: dimension}, nil
}

// BytesTotal: rabitq.go:187-190.
func (rq *HIPRaBitQuantizer) BytesTotal() int { return (rq.dimension+63)/64*8 + 4 }

// BytesPerDimension: rabitq.go:183-185 (sub-byte).
func (rq *HIPRaBitQuantizer) BytesPerDimension() int { return 0 }

// Train: rabitq.go:179-181 (nothing to learn).
func (rq *HIPRaBitQuantizer) Train(vectors [][]float32) error { return nil }

// Encode: rabitq.go:51-78.
func (rq *HIPRaBitQuantizer) Encode(v []float32) ([]byte, error) {
	if len(v) != rq.dimension {
		return nil, hipctx.ErrDimensionMismatch
	}
	code := make([]byte, rq.BytesTotal())
	st := C.vg_rabitq_encode(rq.ctx, C.int32_t(rq.dimension), (*C.float)(unsafe.Pointer(&v[0])), 1,
		(*C.uint8_t)(unsafe.Pointer(&code[0])), nil)
	return code, hipctx.Err(int32(st))
}

// EncodeBatch encodes n row-major vectors in one call.
func (rq *HIPRaBitQuantizer) EncodeBatch(vectors []float32, n int) ([]byte, error) {
	if len(vectors) != n*rq.dimension {
		return nil, hipctx.ErrDimensionMismatch
	}
	codes := make([]byte, n*rq.BytesTotal())
	if n == 0 {
		return codes, nil
	}
	st := C.vg_rabitq_encode(rq.ctx, C.int32_t(rq.dimension), (*C.float)(unsafe.Pointer(&vectors[0])), C.int64_t(n),
		(*C.uint8_t)(unsafe.Pointer(&codes[0])), nil)
	return codes, hipctx.Err(int32(st))
}

// Decode: rabitq.go:82-116 — every component is +-norm/sqrt(dim) by its sign bit.  Host arithmetic: one vector,
// nothing to gain from a PCIe round trip.
func (rq *HIPRaBitQuantizer) Decode(b []byte) ([]float32, error) {
	nb := rq.BytesTotal() - 4
	if len(b) < nb+4 {
		return nil, errors.New("invalid encoded data length")
	}
	norm := math.Float32frombits(binary.LittleEndian.Uint32(b[nb:]))
	scale := norm / float32(math.Sqrt(float64(rq.dimension)))
	out := make([]float32, rq.dimension)
	for i := range out {
		if b[i>>3]&(1<<(uint(i)&7)) != 0 {
			out[i] = scale
		} else {
			out[i] = -scale
		}
	}
	return out, nil
}

// Distance: rabitq.go:119-176.
func (rq *HIPRaBitQuantizer) Distance(query []float32, code []byte) (float32, error) {
	if len(code) < rq.BytesTotal() {
		return 0, errors.New("invalid code length") // rabitq.go:124
	}
	if len(query) != rq.dimension {
		return 0, hipctx.ErrDimensionMismatch
	}
	var out float32
	st := C.vg_rabitq_distance_batch(rq.ctx, C.int32_t(rq.dimension), (*C.float)(unsafe.Pointer(&query[0])),
		(*C.uint8_t)(unsafe.Pointer(&code[0])), 1, (*C.float)(unsafe.Pointer(&out)), nil)
	return out, hipctx.Err(int32(st))
}

// DistanceBatch scores n contiguous codes against one query (the form a GPU is for).
func (rq *HIPRaBitQuantizer) DistanceBatch(query []float32, codes []byte, out []float32) error {
	if len(query) != rq.dimension {
		return hipctx.ErrDimensionMismatch
	}
	if len(codes) < len(out)*rq.BytesTotal() {
		return errors.New("invalid code length")
	}
	if len(out) == 0 {
		return nil
	}
	st := C.vg_rabitq_distance_batch(rq.ctx, C.int32_t(rq.dimension), (*C.float)(unsafe.Pointer(&query[0])),
		(*C.uint8_t)(unsafe.Pointer(&codes[0])), C.int64_t(len(out)), (*C.float)(unsafe.Pointer(&out[0])), nil)
	return hipctx.Err(int32(st))
}
