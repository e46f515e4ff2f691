//go:build hip && cgo

package simd

// GPU "ISA" for the batch-shaped kernels.  The reference dispatches through package-level function variables
// assigned once in init() (kernels.go:11-30; kernels_amd64.go:266-289 setAVX512Kernels).  Single-pair kernels
// (kernelDot, kernelSquaredL2, kernelPqAdc, kernelHamming) stay on the CPU: a cgo call plus a PCIe round trip
// per 3 KiB vector can never win.  The batch-shaped ones move: squaredL2BatchAVX512 / dotBatchAVX512
// (kernels_amd64.go:344-354) become one C call each, and the GPU-only batch forms (ADC lookups, Hamming, bounded
// L2 over many rows) are exported for the segment code.
//
// Ownership: like the //go:noescape assembly stubs, no Go pointer is retained after a call returns.

/*
#cgo CFLAGS: -I${SRCDIR}/../../third_party/vecgo_hip/include
#cgo LDFLAGS: -L${SRCDIR}/../../third_party/vecgo_hip -lvecgo_hip
#include "vecgo_hip.h"
*/
import "C"

import (
	"os"
	"unsafe"

	"github.com/hupe1980/vecgo/internal/hipctx"
)

var hipCtx *C.vg_ctx

func init() {
	if v := os.Getenv("VECGO_SIMD"); v != "" && v != "hip" { // capability.go:82-97: explicit ISA override wins
		return
	}
	p, err := hipctx.Ptr()
	if err != nil {
		return // no gfx950 device: keep the table setAVX512Kernels / setAVX2Kernels built
	}
	hipCtx = (*C.vg_ctx)(p)
	kernelSquaredL2Batch = squaredL2BatchHIP // was squaredL2BatchAVX512 (kernels_amd64.go:344)
	kernelDotBatch = dotBatchHIP             // was dotBatchAVX512       (kernels_amd64.go:350)
}

func squaredL2BatchHIP(query []float32, targets []float32, dim int, out []float32) {
	if len(out) == 0 {
		return
	}
	C.vg_squared_l2_batch(hipCtx, (*C.float)(unsafe.Pointer(&query[0])), (*C.float)(unsafe.Pointer(&targets[0])),
		C.int64_t(dim), C.int64_t(len(out)), (*C.float)(unsafe.Pointer(&out[0])), nil)
}

func dotBatchHIP(query []float32, targets []float32, dim int, out []float32) {
	if len(out) == 0 {
		return
	}
	C.vg_dot_batch(hipCtx, (*C.float)(unsafe.Pointer(&query[0])), (*C.float)(unsafe.Pointer(&targets[0])),
		C.int64_t(dim), C.int64_t(len(out)), (*C.float)(unsafe.Pointer(&out[0])), nil)
}

// SquaredL2BoundedBatch is SquaredL2Bounded (kernels.go:173) of one query against n contiguous rows.
// bounds has 1 entry (shared) or n entries; (dist[i], exceeded[i]) is the pair simd.SquaredL2Bounded returns for row i,
// including the partial sum of the 64-float block at which squaredL2BoundedAvx512 exits when the bound is exceeded
// (bounded_l2_avx512.c:60-75).
func SquaredL2BoundedBatch(query, targets []float32, dim int, bounds []float32, dist []float32, exceeded []int32) {
	n := len(dist)
	if n == 0 {
		return
	}
	C.vg_squared_l2_bounded_batch(hipCtx, (*C.float)(unsafe.Pointer(&query[0])),
		(*C.float)(unsafe.Pointer(&targets[0])), C.int64_t(dim), C.int64_t(n),
		(*C.float)(unsafe.Pointer(&bounds[0])), C.int64_t(len(bounds)), (*C.float)(unsafe.Pointer(&dist[0])),
		(*C.int32_t)(unsafe.Pointer(&exceeded[0])), nil)
}

// PqAdcLookupBatch is PqAdcLookup (kernels.go:56) of one table against n codes of m bytes each.
func PqAdcLookupBatch(table []float32, codes []byte, m int, out []float32) {
	if len(out) == 0 || m == 0 {
		return
	}
	C.vg_pq_adc_lookup_batch(hipCtx, (*C.float)(unsafe.Pointer(&table[0])), (*C.uint8_t)(unsafe.Pointer(&codes[0])),
		C.int64_t(m), C.int64_t(len(out)), (*C.float)(unsafe.Pointer(&out[0])), nil)
}

// HammingBatch is Hamming (kernels.go:71) of one code against n codes of len(a) bytes each.
func HammingBatch(a []byte, codes []byte, out []int32) {
	if len(out) == 0 || len(a) == 0 {
		return
	}
	C.vg_hamming_batch(hipCtx, (*C.uint8_t)(unsafe.Pointer(&a[0])), (*C.uint8_t)(unsafe.Pointer(&codes[0])),
		C.int64_t(len(a)), C.int64_t(len(out)), (*C.int32_t)(unsafe.Pointer(&out[0])), nil)
}
