//go:build hip && cgo

// Package hipctx owns the process-wide vg_ctx (one per GPU) and maps vg_status codes to the error
// strings the reference returns (internal/quantization/pq.go:148-153,190,496; rabitq.go:53,124).
package hipctx

/*
#cgo CFLAGS: -I${SRCDIR}/../../third_party/vecgo_hip/include
#cgo LDFLAGS: -L${SRCDIR}/../../third_party/vecgo_hip -lvecgo_hip
#include "vecgo_hip.h"
*/
import "C"

import (
	"errors"
	"sync"
	"unsafe"
)

var (
	once sync.Once
	ctx  *C.vg_ctx
	cerr error
)

// Reference error values (same strings, so errors.Is / string comparisons in callers keep working).
var (
	ErrDimensionMismatch = errors.New("vector dimension mismatch")
	ErrNotTrained        = errors.New("ProductQuantizer not trained")
	ErrCodeLength        = errors.New("codes length mismatch")
	ErrInvalidCodeLength = errors.New("invalid code length")
)

// Ptr returns the vg_ctx of device 0 as an opaque pointer (cgo types are per package), creating it on
// first use.  An error means there is no usable gfx950 device: callers keep their CPU path.
func Ptr() (unsafe.Pointer, error) {
	once.Do(func() {
		if C.vg_abi_version() != C.VG_ABI_VERSION {
			cerr = errors.New("libvecgo_hip.so: ABI version mismatch")
			return
		}
		if st := C.vg_ctx_create(0, &ctx); st != C.VG_OK {
			cerr = errors.New(C.GoString(C.vg_last_error()))
		}
	})
	return unsafe.Pointer(ctx), cerr
}

// ABIMinor: the library's count of entry-point additions (vecgo_hip.h VG_ABI_MINOR); a caller that wants an entry point
// added after minor m checks ABIMinor() >= m before using it.
func ABIMinor() int { return int(C.vg_abi_minor()) }

// Close destroys the context (tests / orderly shutdown).
func Close() {
	if ctx != nil {
		C.vg_ctx_destroy(ctx)
		ctx = nil
	}
}

// Synchronize waits for the context's stream.
func Synchronize() error {
	return Err(int32(C.vg_ctx_synchronize(ctx, nil)))
}

// Err maps a vg_status to the reference's error values.
func Err(status int32) error {
	switch C.vg_status(status) {
	case C.VG_OK:
		return nil
	case C.VG_ERR_DIM_MISMATCH:
		return ErrDimensionMismatch
	case C.VG_ERR_NOT_TRAINED:
		return ErrNotTrained
	case C.VG_ERR_CODE_LENGTH:
		return ErrCodeLength
	case C.VG_ERR_INVALID_ARG, C.VG_ERR_UNSUPPORTED, C.VG_ERR_NOT_READY, C.VG_ERR_OUT_OF_MEMORY, C.VG_ERR_HIP,
		C.VG_ERR_NO_DEVICE, C.VG_ERR_FORMAT, C.VG_ERR_CHECKSUM:
		// the library's message carries the reference's text where it has one ("unsupported metric for
		// float32: ...", "invalid magic number", "checksum mismatch: expected %x, got %x")
		return errors.New(C.GoString(C.vg_last_error()))
	default:
		return errors.New(C.GoString(C.vg_status_string(C.int32_t(status))))
	}
}
