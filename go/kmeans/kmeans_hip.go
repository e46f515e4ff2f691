//go:build hip && cgo

// Package kmeans (HIP twin): TrainKMeans / AssignPartition / FindClosestCentroids (internal/kmeans/kmeans.go:16-280)
// over the C ABI.  Same return conventions: TrainKMeans returns (nil, nil) when there are fewer vectors than
// clusters (kmeans.go:17-20) and ctx.Err() when cancelled before the call.
package kmeans

/*
#cgo CFLAGS: -I${SRCDIR}/../../third_party/vecgo_hip/include
#cgo LDFLAGS: -L${SRCDIR}/../../third_party/vecgo_hip -lvecgo_hip
#include "vecgo_hip.h"
*/
import "C"

import (
	"context"
	"unsafe"

	"github.com/hupe1980/vecgo/distance"
	"github.com/hupe1980/vecgo/internal/hipctx"
)

func cf(p []float32) *C.float { return (*C.float)(unsafe.Pointer(&p[0])) }

// TrainKMeansHIP: kmeans.go:16-138.  The reference seeds from the unseeded global math/rand (rand.Perm :25, empty
// cluster re-seeding :131); here the stream is a counter-based generator keyed by `seed`.
func TrainKMeansHIP(ctx context.Context, vectors []float32, dim int, k int, metric distance.Metric, maxIter int, seed uint64) ([]float32, error) {
	n := len(vectors) / dim
	if n < k {
		return nil, nil
	}
	if err := ctx.Err(); err != nil {
		return nil, err
	}
	p, err := hipctx.Ptr()
	if err != nil {
		return nil, err
	}
	centroids := make([]float32, k*dim)
	var produced C.int32_t
	st := C.vg_kmeans_train((*C.vg_ctx)(p), cf(vectors), C.int64_t(n), C.int32_t(dim), C.int32_t(k), C.int32_t(metric),
		C.int32_t(maxIter), C.uint64_t(seed), cf(centroids), &produced, nil)
	if st != C.VG_OK {
		return nil, hipctx.Err(int32(st))
	}
	if produced == 0 {
		return nil, nil
	}
	return centroids, nil
}

// AssignPartitionBatch: AssignPartition (kmeans.go:142-196) of n row-major vectors at once.
func AssignPartitionBatch(vectors []float32, centroids []float32, dim int, metric distance.Metric, out []int32) error {
	n := len(vectors) / dim
	if n == 0 {
		return nil
	}
	p, err := hipctx.Ptr()
	if err != nil {
		return err
	}
	st := C.vg_kmeans_assign((*C.vg_ctx)(p), cf(vectors), C.int64_t(n), C.int32_t(dim), cf(centroids),
		C.int32_t(len(centroids)/dim), C.int32_t(metric), (*C.int32_t)(unsafe.Pointer(&out[0])), nil)
	return hipctx.Err(int32(st))
}

// AssignPartitionHIP: kmeans.go:142-196.
func AssignPartitionHIP(vec []float32, centroids []float32, dim int, metric distance.Metric) (int, error) {
	var out [1]int32
	err := AssignPartitionBatch(vec, centroids, dim, metric, out[:])
	return int(out[0]), err
}

// FindClosestCentroidsHIP: kmeans.go:217-280 (selection when n <= k/4 && n < 16, else the sort — same result order).
func FindClosestCentroidsHIP(query []float32, centroids []float32, dim int, n int, metric distance.Metric) ([]int, error) {
	k := len(centroids) / dim
	if n > k {
		n = k
	}
	if n <= 0 {
		return []int{}, nil
	}
	p, err := hipctx.Ptr()
	if err != nil {
		return nil, err
	}
	ids := make([]int32, n)
	var got C.int32_t
	st := C.vg_find_closest_centroids((*C.vg_ctx)(p), cf(query), cf(centroids), C.int32_t(dim), C.int32_t(k), C.int32_t(n),
		C.int32_t(metric), (*C.int32_t)(unsafe.Pointer(&ids[0])), &got, nil)
	if st != C.VG_OK {
		return nil, hipctx.Err(int32(st))
	}
	out := make([]int, int(got))
	for i := range out {
		out[i] = int(ids[i])
	}
	return out, nil
}
