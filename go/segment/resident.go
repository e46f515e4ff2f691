//go:build hip && cgo

// Package segment (HIP twin): a device-resident copy of one segment — what flat.Segment.Search
// (internal/segment/flat/segment.go:447-749), the DiskANN beam search (diskann/segment.go:503-706), the memtable's
// hnsw.KNNSearch (hnsw.go:1650-1755) and Segment.Rerank (flat/segment.go:754-780) run against, ONE call per query
// batch instead of one distance call per candidate.  filter == nil paths only: metadata filters and tombstones stay
// on the Go side.
//
// Ownership: Upload* copy the slices to HBM and retain nothing; results are written into caller-owned slices;
// Close frees the device memory.
package segment

/*
#cgo CFLAGS: -I${SRCDIR}/../../third_party/vecgo_hip/include
#cgo LDFLAGS: -L${SRCDIR}/../../third_party/vecgo_hip -lvecgo_hip
#include "vecgo_hip.h"
*/
import "C"

import (
	"fmt"
	"unsafe"

	"github.com/hupe1980/vecgo/distance"
	"github.com/hupe1980/vecgo/internal/hipctx"
)

// Stats mirrors vg_search_stats = the reference's FilterGateStats counters (searcher/searcher.go:114-137) plus
// the rows the greedy descent scored.  Its size is part of the ABI (VG_ABI_VERSION).
type Stats struct {
	NodesVisited, DistanceComputations, DistanceShortCircuits, Pops int64
	DescentDistanceComputations                                    int64
}

const (
	ScanF32 = int32(C.VG_SCAN_F32)
	ScanPQ  = int32(C.VG_SCAN_PQ)
	ScanSQ8 = int32(C.VG_SCAN_SQ8)
)

// Vamana node scorers (diskann/segment.go:512-589 distFn).
const (
	VamanaF32, VamanaPQ, VamanaRaBitQ, VamanaInt4 = 0, 1, 2, 3
)

type Resident struct {
	ctx  *C.vg_ctx
	h    *C.vg_index
	seg  *C.vg_segment // set when opened from a segment image
	rows int
	dim  int
}

func fp(p []float32) *C.float  { return (*C.float)(unsafe.Pointer(&p[0])) }
func up(p []uint32) *C.uint32_t { return (*C.uint32_t)(unsafe.Pointer(&p[0])) }
func bp(p []byte) *C.uint8_t    { return (*C.uint8_t)(unsafe.Pointer(&p[0])) }

// NewResident creates an empty twin of a segment of `rows` x `dim`.
func NewResident(rows, dim int, metric distance.Metric) (*Resident, error) {
	p, err := hipctx.Ptr()
	if err != nil {
		return nil, err
	}
	r := &Resident{ctx: (*C.vg_ctx)(p), rows: rows, dim: dim}
	if st := C.vg_index_create(r.ctx, C.int64_t(rows), C.int32_t(dim), C.int32_t(metric), &r.h); st != C.VG_OK {
		return nil, hipctx.Err(int32(st))
	}
	return r, nil
}

// OpenFlat / OpenDiskANN hand a whole segment file (mmap or read) to the library: header, section bounds and —
// with verify — the CRC32C are checked as flat.Open / diskann Open check them (flat/segment.go:105-300,
// diskann/segment.go:165-440); the sections are uploaded as they lie in the file.
func OpenFlat(image []byte, verify bool) (*Resident, error)    { return open(image, verify, false) }
func OpenDiskANN(image []byte, verify bool) (*Resident, error) { return open(image, verify, true) }

func open(image []byte, verify, diskann bool) (*Resident, error) {
	p, err := hipctx.Ptr()
	if err != nil {
		return nil, err
	}
	r := &Resident{ctx: (*C.vg_ctx)(p)}
	v := C.int32_t(0)
	if verify {
		v = 1
	}
	var st C.int32_t
	if diskann {
		st = C.vg_segment_open_diskann(r.ctx, unsafe.Pointer(&image[0]), C.int64_t(len(image)), v, &r.seg, nil)
	} else {
		st = C.vg_segment_open_flat(r.ctx, unsafe.Pointer(&image[0]), C.int64_t(len(image)), v, &r.seg, nil)
	}
	if st != C.VG_OK {
		return nil, hipctx.Err(int32(st)) // "invalid magic number", "checksum mismatch: ...", "file too short for ..."
	}
	var info C.vg_segment_info
	if st := C.vg_segment_get_info(r.seg, &info); st != C.VG_OK {
		C.vg_segment_close(r.seg)
		return nil, hipctx.Err(int32(st))
	}
	r.h = C.vg_segment_index(r.seg) // borrowed: lives until vg_segment_close
	r.rows, r.dim = int(info.rows), int(info.dim)
	return r, nil
}

func (r *Resident) Close() {
	if r.seg != nil {
		C.vg_segment_close(r.seg)
	} else if r.h != nil {
		C.vg_index_destroy(r.h)
	}
	r.seg, r.h = nil, nil
}

// ---- uploads (the reference's in-memory layouts, copied) ---------------------------------------------------------

// UploadVectors: one n*dim row-major array (vectorstore/columnar.go:21-24, flat/segment.go:692).
func (r *Resident) UploadVectors(base []float32) error {
	return hipctx.Err(int32(C.vg_index_set_vectors(r.h, fp(base), nil)))
}

// UploadPQCodes: n*m code bytes (flat/segment.go:678-680) scored with the quantizer behind `pq`
// (HIPProductQuantizer.Handle()).
func (r *Resident) UploadPQCodes(pq unsafe.Pointer, codes []byte) error {
	return hipctx.Err(int32(C.vg_index_set_pq_codes(r.h, (*C.vg_pq)(pq), bp(codes), nil)))
}

// UploadSQ8Codes: n*dim code bytes (flat/segment.go:550) with the quantizer behind `sq`.
func (r *Resident) UploadSQ8Codes(sq unsafe.Pointer, codes []byte) error {
	return hipctx.Err(int32(C.vg_index_set_sq8_codes(r.h, (*C.vg_sq8)(sq), bp(codes), nil)))
}

// UploadInt4Codes: n*ceil(dim/2) code bytes (diskann/segment.go:378-416).
func (r *Resident) UploadInt4Codes(iq unsafe.Pointer, codes []byte) error {
	return hipctx.Err(int32(C.vg_index_set_int4_codes(r.h, (*C.vg_int4)(iq), bp(codes), nil)))
}

// UploadRaBitQCodes: n*BytesTotal() code bytes (diskann/segment.go:1393-1408).
func (r *Resident) UploadRaBitQCodes(codes []byte) error {
	return hipctx.Err(int32(C.vg_index_set_rabitq_codes(r.h, bp(codes), nil)))
}

// UploadPartitions: IVF centroids and the first row of every partition (flat/segment.go:727-749).
func (r *Resident) UploadPartitions(centroids []float32, partitionOffsets []uint32, numPartitions int) error {
	return hipctx.Err(int32(C.vg_index_set_partitions(r.h, fp(centroids), up(partitionOffsets), C.int32_t(numPartitions), nil)))
}

// UploadVamanaGraph: n*R neighbour ids, 0xFFFFFFFF = none (diskann/segment.go:1376-1391).
func (r *Resident) UploadVamanaGraph(R int, graph []uint32, entry uint32) error {
	return hipctx.Err(int32(C.vg_index_set_vamana_graph(r.h, C.int32_t(R), up(graph), C.uint32_t(entry), nil)))
}

// UploadHNSWGraph: layer 0 as n*m0 ids; the upper layers as per-level slot tables and adjacency rows.
func (r *Resident) UploadHNSWGraph(m0 int, l0 []uint32, maxLevel, m int, upperSlot, upperAdj []uint32, levelRows []int64, entry uint32) error {
	var slot, adj *C.uint32_t
	var lr *C.int64_t
	if maxLevel > 0 {
		slot, adj, lr = up(upperSlot), up(upperAdj), (*C.int64_t)(unsafe.Pointer(&levelRows[0]))
	}
	return hipctx.Err(int32(C.vg_index_set_hnsw_graph(r.h, C.int32_t(m0), up(l0), C.int32_t(maxLevel), C.int32_t(m), slot, adj, lr,
		C.uint32_t(entry), nil)))
}

// BuildHNSW: ApplyBatchInsert over the rows (hnsw.go:639-684): ids = row numbers, levels = layerForApplyInsert.
func (r *Resident) BuildHNSW(m, efConstruction int) error {
	return hipctx.Err(int32(C.vg_hnsw_build(r.h, C.int32_t(m), C.int32_t(efConstruction), 8192, 32, nil)))
}

// ---- searches: nq row-major queries in, nq*k (RowID, Score) best-first out -----------------------------------------

func (r *Resident) out(nq, k int) ([]uint32, []float32) { return make([]uint32, nq*k), make([]float32, nq*k) }

// SearchFlat: flat.Segment.Search's fp32 branch (flat/segment.go:691-721); hnsw.BruteSearch is SearchHNSWBrute.
func (r *Resident) SearchFlat(queries []float32, nq, k int) ([]uint32, []float32, error) {
	ids, sc := r.out(nq, k)
	st := C.vg_search_flat(r.h, fp(queries), C.int64_t(nq), C.int32_t(k), up(ids), fp(sc), nil)
	return ids, sc, hipctx.Err(int32(st))
}

// SearchPQ: the PQ branch (flat/segment.go:476-483,678-689): BuildDistanceTable + PqAdcLookup per row.
func (r *Resident) SearchPQ(queries []float32, nq, k int) ([]uint32, []float32, error) {
	ids, sc := r.out(nq, k)
	st := C.vg_search_pq_adc(r.h, fp(queries), C.int64_t(nq), C.int32_t(k), up(ids), fp(sc), nil)
	return ids, sc, hipctx.Err(int32(st))
}

// SearchSQ8: the SQ8 branch (flat/segment.go:517-604 L2, :659-667 Dot).
func (r *Resident) SearchSQ8(queries []float32, nq, k int) ([]uint32, []float32, error) {
	ids, sc := r.out(nq, k)
	st := C.vg_search_sq8(r.h, fp(queries), C.int64_t(nq), C.int32_t(k), up(ids), fp(sc), nil)
	return ids, sc, hipctx.Err(int32(st))
}

// EnableSQ8Nomination: batches of SearchSQ8 are nominated by a bfloat16 MFMA GEMM over the dequantised rows (+ rows*dim*2 bytes of
// device memory) and re-scored exactly from the codes; results unchanged.
func (r *Resident) EnableSQ8Nomination(on bool) error {
	v := C.int32_t(0)
	if on {
		v = 1
	}
	return hipctx.Err(int32(C.vg_index_enable_sq8_nomination(r.h, v, nil)))
}

// EnablePQNomination: batches of SearchPQ (queries x rows >= 24M, k <= 256) are nominated by a bfloat16 MFMA GEMM over the decoded
// rows (+ rows*dim*2 bytes of device memory) and re-scored from the codes against the query's distance table; results unchanged.
func (r *Resident) EnablePQNomination(on bool) error {
	v := C.int32_t(0)
	if on {
		v = 1
	}
	return hipctx.Err(int32(C.vg_index_enable_pq_nomination(r.h, v, nil)))
}

// SearchRaBitQ: exhaustive scan of the RaBitQ codes (rq.Distance per row).
func (r *Resident) SearchRaBitQ(queries []float32, nq, k int) ([]uint32, []float32, error) {
	ids, sc := r.out(nq, k)
	st := C.vg_search_rabitq(r.h, fp(queries), C.int64_t(nq), C.int32_t(k), up(ids), fp(sc), nil)
	return ids, sc, hipctx.Err(int32(st))
}

// SearchProbed: a partitioned flat segment, only the nprobes closest partitions (flat/segment.go:727-749).
func (r *Resident) SearchProbed(queries []float32, nq, k, nprobes int, scan int32) ([]uint32, []float32, error) {
	ids, sc := r.out(nq, k)
	st := C.vg_search_flat_probed(r.h, fp(queries), C.int64_t(nq), C.int32_t(k), C.int32_t(nprobes), C.int32_t(scan), up(ids), fp(sc), nil)
	return ids, sc, hipctx.Err(int32(st))
}

// SearchSegmentFiltered: Segment.Search with a filter for a segment opened from its file (flat or DiskANN, by what it holds).
func (r *Resident) SearchSegmentFiltered(queries []float32, nq, k, nprobes int, mask []byte, maskStride int) ([]uint32, []float32, error) {
	ids, sc := r.out(nq, k)
	need := (r.rows + 7) / 8
	if maskStride != 0 {
		if maskStride < need {
			return nil, nil, fmt.Errorf("SearchSegmentFiltered: maskStride %d is shorter than a mask (%d bytes)", maskStride, need)
		}
		need += (nq - 1) * maskStride
	}
	if len(mask) < need {
		return nil, nil, fmt.Errorf("SearchSegmentFiltered: mask holds %d bytes, %d needed", len(mask), need)
	}
	st := C.vg_segment_search_filtered(r.seg, fp(queries), C.int64_t(nq), C.int32_t(k), C.int32_t(nprobes), bp(mask), C.int64_t(maskStride), up(ids), fp(sc), nil)
	return ids, sc, hipctx.Err(int32(st))
}

// SearchFiltered: Segment.Search with `filter segment.Filter` set (flat/segment.go:631-635, :559-561): rows whose mask bit
// is clear are skipped.  mask: bit i of byte i/8 = filter.Matches(i), len(mask) == ceil(rows/8) for one mask (maskStride 0)
// or (nq-1)*maskStride + ceil(rows/8) for one per query.
func (r *Resident) SearchFiltered(queries []float32, nq, k, nprobes int, scan int32, mask []byte, maskStride int) ([]uint32, []float32, error) {
	ids, sc := r.out(nq, k)
	need := (r.rows + 7) / 8
	if maskStride != 0 {
		if maskStride < need {
			return nil, nil, fmt.Errorf("SearchFiltered: maskStride %d is shorter than a mask (%d bytes)", maskStride, need)
		}
		need += (nq - 1) * maskStride
	}
	if len(mask) < need {
		return nil, nil, fmt.Errorf("SearchFiltered: mask holds %d bytes, %d needed", len(mask), need)
	}
	st := C.vg_search_flat_filtered(r.h, fp(queries), C.int64_t(nq), C.int32_t(k), C.int32_t(nprobes), C.int32_t(scan), bp(mask), C.int64_t(maskStride), up(ids), fp(sc), nil)
	return ids, sc, hipctx.Err(int32(st))
}

// Search: the whole of Segment.Search for a segment opened from its file: scan type / beam search by what the
// file holds.
func (r *Resident) Search(queries []float32, nq, k, nprobes int) ([]uint32, []float32, error) {
	ids, sc := r.out(nq, k)
	st := C.vg_segment_search(r.seg, fp(queries), C.int64_t(nq), C.int32_t(k), C.int32_t(nprobes), up(ids), fp(sc), nil)
	return ids, sc, hipctx.Err(int32(st))
}

// SearchHNSW: hnsw.KNNSearch (hnsw.go:1650-1755).  stats may be nil; otherwise len(stats) >= nq.
func (r *Resident) SearchHNSW(queries []float32, nq, k, ef int, stats []Stats) ([]uint32, []float32, error) {
	ids, sc := r.out(nq, k)
	var sp *C.vg_search_stats
	if len(stats) >= nq && nq > 0 {
		sp = (*C.vg_search_stats)(unsafe.Pointer(&stats[0]))
	}
	st := C.vg_search_hnsw(r.h, fp(queries), C.int64_t(nq), C.int32_t(k), C.int32_t(ef), up(ids), fp(sc), sp, nil)
	return ids, sc, hipctx.Err(int32(st))
}

// SearchHNSWPQ: the graph walked on PQ codes (distFunc = pq.ComputeAsymmetricDistance, diskann/segment.go:536-557);
// follow with Rerank (engine/search.go:914-965).
func (r *Resident) SearchHNSWPQ(queries []float32, nq, k, ef int, stats []Stats) ([]uint32, []float32, error) {
	ids, sc := r.out(nq, k)
	var sp *C.vg_search_stats
	if len(stats) >= nq && nq > 0 {
		sp = (*C.vg_search_stats)(unsafe.Pointer(&stats[0]))
	}
	st := C.vg_search_hnsw_pq(r.h, fp(queries), C.int64_t(nq), C.int32_t(k), C.int32_t(ef), up(ids), fp(sc), sp, nil)
	return ids, sc, hipctx.Err(int32(st))
}

// SearchHNSWFiltered: searchExecute with a filter and a selectivity hint above 0.3 — searchLayerWithPostFilter
// (hnsw.go:1159-1218).  mask: bit i of byte i/8 = row i passes (filter.Matches and not tombstoned), len(mask) ==
// ceil(n/8) for one mask (maskStride 0) or nq*maskStride; ef is what determineEF returned.  A selectivity at or below
// 0.3 (or unknown, < 0) is routed by the library to the predicate-aware walk, as searchExecute does (hnsw.go:1086-1157):
// the same as calling SearchHNSWPredicate.
func (r *Resident) SearchHNSWFiltered(queries []float32, nq, k, ef int, mask []byte, maskStride int, selectivity float64, stats []Stats) ([]uint32, []float32, error) {
	ids, sc := r.out(nq, k)
	need := (r.rows + 7) / 8
	if maskStride != 0 {
		if maskStride < need {
			return nil, nil, fmt.Errorf("SearchHNSWFiltered: maskStride %d is shorter than a mask (%d bytes)", maskStride, need)
		}
		need += (nq - 1) * maskStride
	}
	if len(mask) < need {
		return nil, nil, fmt.Errorf("SearchHNSWFiltered: mask holds %d bytes, %d needed", len(mask), need)
	}
	var sp *C.vg_search_stats
	if len(stats) >= nq && nq > 0 {
		sp = (*C.vg_search_stats)(unsafe.Pointer(&stats[0]))
	}
	st := C.vg_search_hnsw_filtered(r.h, fp(queries), C.int64_t(nq), C.int32_t(k), C.int32_t(ef), bp(mask), C.int64_t(maskStride), C.double(selectivity), up(ids), fp(sc), sp, nil)
	return ids, sc, hipctx.Err(int32(st))
}

// SearchHNSWPredicate: searchExecute with a filter whose selectivity hint is at or below 0.3, or unknown —
// searchLayerPredicateAware (hnsw.go:1406-1558).  mask as for SearchHNSWFiltered (filter.Matches only); deleted: the
// tombstone bitmap (ceil(n/8) bytes) or nil.  stats[i].ShortCircuits = ExpansionsSkipped.
func (r *Resident) SearchHNSWPredicate(queries []float32, nq, k, ef int, mask []byte, maskStride int, deleted []byte, stats []Stats) ([]uint32, []float32, error) {
	ids, sc := r.out(nq, k)
	need := (r.rows + 7) / 8
	if len(deleted) > 0 && len(deleted) < need {
		return nil, nil, fmt.Errorf("SearchHNSWPredicate: deleted holds %d bytes, %d needed", len(deleted), need)
	}
	if maskStride != 0 {
		if maskStride < need {
			return nil, nil, fmt.Errorf("SearchHNSWPredicate: maskStride %d is shorter than a mask (%d bytes)", maskStride, need)
		}
		need += (nq - 1) * maskStride
	}
	if len(mask) < need {
		return nil, nil, fmt.Errorf("SearchHNSWPredicate: mask holds %d bytes, %d needed", len(mask), need)
	}
	var sp *C.vg_search_stats
	if len(stats) >= nq && nq > 0 {
		sp = (*C.vg_search_stats)(unsafe.Pointer(&stats[0]))
	}
	var dp *C.uint8_t
	if len(deleted) > 0 {
		dp = bp(deleted)
	}
	st := C.vg_search_hnsw_predicate(r.h, fp(queries), C.int64_t(nq), C.int32_t(k), C.int32_t(ef), bp(mask), C.int64_t(maskStride), dp, up(ids), fp(sc), sp, nil)
	return ids, sc, hipctx.Err(int32(st))
}

// SetHNSWTombstones: g.tombstones (hnsw.go:95) as a bitmap, ceil(rows/8) bytes, nil = none: deleted nodes are walked
// through but never returned by the HNSW searches (hnsw.go:1381-1390).
func (r *Resident) SetHNSWTombstones(deleted []byte) error {
	if len(deleted) == 0 {
		return hipctx.Err(int32(C.vg_index_set_hnsw_tombstones(r.h, nil, nil)))
	}
	if need := (r.rows + 7) / 8; len(deleted) < need {
		return fmt.Errorf("SetHNSWTombstones: bitmap holds %d bytes, %d needed", len(deleted), need)
	}
	return hipctx.Err(int32(C.vg_index_set_hnsw_tombstones(r.h, bp(deleted), nil)))
}

// SetHNSWEdgeDistances: the layer-0 lists' cached Neighbor.Dist (node.go:62-80), rows*m0 values slot for slot with the uploaded
// lists; nil = recompute them from the fp32 rows.
func (r *Resident) SetHNSWEdgeDistances(l0Dist []float32) error {
	var p *C.float
	if len(l0Dist) > 0 {
		p = fp(l0Dist)
	}
	return hipctx.Err(int32(C.vg_index_set_hnsw_edge_distances(r.h, p, nil)))
}

// BruteMode selects which of the HNSW index's exhaustive paths SearchHNSWBrute replays.
type BruteMode int32

const (
	BruteScan   BruteMode = 0 // hnsw.BruteSearch + scanSegment (hnsw.go:2021-2101): PopItem + PushItem
	BruteBitmap BruteMode = 1 // searchBitmap (hnsw.go:2240-2263): TryPushBounded(k)
)

// SearchHNSWBrute: BruteSearch / searchBitmap over the rows whose bit is set in mask (bit i of byte i/8; nil =
// every row; len(mask) == ceil(n/8) for one mask, nq*maskStride for one per query), in ascending id through the
// reference's PriorityQueue: ids AND their order among equal distances are the reference's.  Distances are the
// index's (L2, -dot, 0.5*L2), best first.
func (r *Resident) SearchHNSWBrute(queries []float32, nq, k int, mode BruteMode, mask []byte, maskStride int) ([]uint32, []float32, error) {
	ids, sc := r.out(nq, k)
	var mp *C.uint8_t
	if len(mask) > 0 {
		// the library reads ceil(rows/8) bytes per mask, query q's at q*maskStride: a shorter slice would be read past its end
		need := (r.rows + 7) / 8
		if maskStride != 0 {
			if maskStride < need {
				return nil, nil, fmt.Errorf("SearchHNSWBrute: maskStride %d is shorter than a mask (%d bytes)", maskStride, need)
			}
			need += (nq - 1) * maskStride
		}
		if len(mask) < need {
			return nil, nil, fmt.Errorf("SearchHNSWBrute: mask holds %d bytes, %d needed", len(mask), need)
		}
		mp = (*C.uint8_t)(unsafe.Pointer(&mask[0]))
	}
	st := C.vg_search_hnsw_brute(r.h, fp(queries), C.int64_t(nq), C.int32_t(k), C.int32_t(mode), mp, C.int64_t(maskStride), up(ids), fp(sc), nil)
	return ids, sc, hipctx.Err(int32(st))
}

// SearchVamana: diskann searchInternal (diskann/segment.go:503-706) with the given node scorer.
func (r *Resident) SearchVamana(queries []float32, nq, k, kind int, stats []Stats) ([]uint32, []float32, error) {
	ids, sc := r.out(nq, k)
	var sp *C.vg_search_stats
	if len(stats) >= nq && nq > 0 {
		sp = (*C.vg_search_stats)(unsafe.Pointer(&stats[0]))
	}
	st := C.vg_search_vamana(r.h, fp(queries), C.int64_t(nq), C.int32_t(k), C.int32_t(kind), up(ids), fp(sc), sp, nil)
	return ids, sc, hipctx.Err(int32(st))
}

// SearchVamanaFiltered: searchInternal with `filter` set (diskann/segment.go:616-627): a row whose mask bit is clear is walked
// through but never enters the result heap.  mask as for SearchFiltered.
func (r *Resident) SearchVamanaFiltered(queries []float32, nq, k, kind int, mask []byte, maskStride int, stats []Stats) ([]uint32, []float32, error) {
	ids, sc := r.out(nq, k)
	need := (r.rows + 7) / 8
	if maskStride != 0 {
		if maskStride < need {
			return nil, nil, fmt.Errorf("SearchVamanaFiltered: maskStride %d is shorter than a mask (%d bytes)", maskStride, need)
		}
		need += (nq - 1) * maskStride
	}
	if len(mask) < need {
		return nil, nil, fmt.Errorf("SearchVamanaFiltered: mask holds %d bytes, %d needed", len(mask), need)
	}
	var sp *C.vg_search_stats
	if len(stats) >= nq && nq > 0 {
		sp = (*C.vg_search_stats)(unsafe.Pointer(&stats[0]))
	}
	st := C.vg_search_vamana_filtered(r.h, fp(queries), C.int64_t(nq), C.int32_t(k), C.int32_t(kind), bp(mask), C.int64_t(maskStride), up(ids), fp(sc), sp, nil)
	return ids, sc, hipctx.Err(int32(st))
}

// Rerank: Segment.Rerank + top-k (flat/segment.go:754-780, engine/search.go:914-965): nc candidate rows per query.
func (r *Resident) Rerank(queries []float32, nq int, candidates []uint32, nc, k int) ([]uint32, []float32, error) {
	ids, sc := r.out(nq, k)
	st := C.vg_rerank(r.h, fp(queries), C.int64_t(nq), up(candidates), C.int32_t(nc), C.int32_t(k), up(ids), fp(sc), nil)
	return ids, sc, hipctx.Err(int32(st))
}

// MergeTopK: the engine's fan-in over segments (engine/search.go:904-908) for `lists` resident segments' results:
// ties by (score, RowID + idOffsets[list]).
func MergeTopK(idsIn []uint32, scoresIn []float32, lists, nq, k int, metric distance.Metric, idOffsets []uint32) ([]uint32, []float32, error) {
	p, err := hipctx.Ptr()
	if err != nil {
		return nil, nil, err
	}
	ids, sc := make([]uint32, nq*k), make([]float32, nq*k)
	var off *C.uint32_t
	if len(idOffsets) > 0 {
		off = up(idOffsets)
	}
	st := C.vg_merge_topk((*C.vg_ctx)(p), up(idsIn), fp(scoresIn), C.int32_t(lists), C.int64_t(nq), C.int32_t(k), C.int32_t(metric), off,
		up(ids), fp(sc), nil)
	return ids, sc, hipctx.Err(int32(st))
}
