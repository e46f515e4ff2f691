"""ctypes binding of the CPU oracle (oracle/vg_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg — never by the product package (vecgo_amd).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent

METRIC_L2, METRIC_COSINE, METRIC_DOT, METRIC_HAMMING = 0, 1, 2, 3
VAMANA_F32, VAMANA_PQ, VAMANA_RABITQ, VAMANA_INT4 = 0, 1, 2, 3

_f32p = C.POINTER(C.c_float)
_u8p = C.POINTER(C.c_uint8)
_i8p = C.POINTER(C.c_int8)
_u32p = C.POINTER(C.c_uint32)
_i32p = C.POINTER(C.c_int32)
_u64p = C.POINTER(C.c_uint64)


def _cpu_flags() -> set:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("flags"):
                return set(line.split(":", 1)[1].split())
    except OSError:
        pass
    return set()


def _load() -> C.CDLL:
    want = "liboracle_v4.so" if "avx512f" in _cpu_flags() else "liboracle_v3.so"
    path = _HERE / want
    if not path.exists():
        subprocess.check_call(["make", "-C", str(_HERE), want], stdout=subprocess.DEVNULL)
    return C.CDLL(str(path))


class PQ(C.Structure):
    _fields_ = [("dim", C.c_int32), ("m", C.c_int32), ("k", C.c_int32), ("subdim", C.c_int32),
                ("codebooks", _i8p), ("scales", _f32p), ("offsets", _f32p)]


class HnswGraph(C.Structure):
    _fields_ = [("n", C.c_int64), ("dim", C.c_int32), ("metric", C.c_int32), ("base", _f32p),
                ("m0", C.c_int32), ("l0", _u32p), ("max_level", C.c_int32), ("m", C.c_int32),
                ("slot", C.POINTER(_u32p)), ("adj", C.POINTER(_u32p)),
                ("entry_point", C.c_uint32), ("pq", C.POINTER(PQ)), ("codes", _u8p), ("tombstones", _u8p)]


class Vamana(C.Structure):
    _fields_ = [("n", C.c_int64), ("dim", C.c_int32), ("r", C.c_int32), ("graph", _u32p),
                ("entry_point", C.c_uint32), ("kind", C.c_int32), ("metric", C.c_int32),
                ("base", _f32p), ("pq", C.POINTER(PQ)), ("codes", _u8p), ("int4_table", _f32p)]


class FlatSeg(C.Structure):
    _fields_ = [("n", C.c_int64), ("dim", C.c_int32), ("metric", C.c_int32), ("base", _f32p),
                ("pq", C.POINTER(PQ)), ("codes", _u8p), ("sq_mins", _f32p), ("sq_inv_scales", _f32p),
                ("num_partitions", C.c_int32), ("centroids", _f32p), ("part_offsets", _u32p)]


class SearchStats(C.Structure):
    _fields_ = [("nodes_visited", C.c_int64), ("distance_computations", C.c_int64),
                ("distance_short_circuits", C.c_int64), ("pops", C.c_int64)]


lib = _load()


def _sig(name, restype, *argtypes):
    f = getattr(lib, name)
    f.restype = restype
    f.argtypes = list(argtypes)
    return f


_sig("vgo_dot_avx512", C.c_float, _f32p, _f32p, C.c_int64)
_sig("vgo_l2_avx512", C.c_float, _f32p, _f32p, C.c_int64)
_sig("vgo_l2_batch_avx512", None, _f32p, _f32p, C.c_int64, C.c_int64, _f32p)
_sig("vgo_dot_batch_avx512", None, _f32p, _f32p, C.c_int64, C.c_int64, _f32p)
_sig("vgo_l2_bounded_avx512", None, _f32p, _f32p, C.c_int64, C.c_float, _f32p, _i32p)
_sig("vgo_adc_avx512", C.c_float, _f32p, _u8p, C.c_int64)
_sig("vgo_adc_generic", C.c_float, _f32p, _u8p, C.c_int64)
_sig("vgo_hamming", C.c_int64, _u8p, _u8p, C.c_int64)
_sig("vgo_scale", None, _f32p, C.c_int64, C.c_float)
_sig("vgo_sqrt", C.c_float, C.c_float)
_sig("vgo_l2_int8_deq", C.c_float, _f32p, _i8p, C.c_int64, C.c_float, C.c_float)
_sig("vgo_build_table_int8", None, _f32p, _i8p, C.c_int64, C.c_int64, C.c_float, C.c_float, _f32p)
_sig("vgo_nearest_centroid_int8", C.c_int64, _f32p, _i8p, C.c_int64, C.c_int64, C.c_float, C.c_float)
_sig("vgo_pq_build_table", None, C.POINTER(PQ), _f32p, _f32p)
_sig("vgo_pq_encode", None, C.POINTER(PQ), _f32p, _u8p)
_sig("vgo_pq_decode", None, C.POINTER(PQ), _u8p, _f32p)
_sig("vgo_pq_asym_distance", C.c_float, C.POINTER(PQ), _f32p, _u8p)
_sig("vgo_pq_quantize_centroids", None, _f32p, C.c_int64, _i8p, _f32p, _f32p)
_sig("vgo_pq_train", C.c_int, _f32p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
     C.c_uint64, _i8p, _f32p, _f32p, _f32p)
_sig("vgo_rabitq_code_bytes", C.c_int64, C.c_int32)
_sig("vgo_rabitq_encode", None, _f32p, C.c_int32, _u8p)
_sig("vgo_rabitq_distance", C.c_float, _f32p, C.c_int32, _u8p)
_sig("vgo_binary_encode_u64", None, _f32p, C.c_int32, C.c_float, _u64p)
_sig("vgo_binary_train", C.c_float, _f32p, C.c_int64, C.c_int32)
_sig("vgo_binary_decode", None, _u8p, C.c_int32, C.c_int32, C.c_float, _f32p)
_sig("vgo_normalize_l2", C.c_int32, _f32p, C.c_int32)
_sig("vgo_kmeans_train", C.c_int, _f32p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
     C.c_uint64, _f32p)
_sig("vgo_assign_partition", C.c_int32, _f32p, _f32p, C.c_int32, C.c_int32, C.c_int32)
_sig("vgo_find_closest_centroids", C.c_int, _f32p, _f32p, C.c_int32, C.c_int32, C.c_int32,
     C.c_int32, _i32p)
_sig("vgo_rng_u64", C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64)
_sig("vgo_flat_search_f32", C.c_int32, _f32p, C.c_int64, C.c_int32, C.c_int32, _f32p, C.c_int32,
     _u32p, _f32p)
_sig("vgo_flat_search_pq", C.c_int32, C.POINTER(PQ), _u8p, C.c_int64, _f32p, C.c_int32, _u32p, _f32p)
_sig("vgo_flat_search_rabitq", C.c_int32, _u8p, C.c_int64, C.c_int32, _f32p, C.c_int32, _u32p, _f32p)
_sig("vgo_sq8u_l2_batch", None, _f32p, _u8p, _f32p, _f32p, C.c_int64, C.c_int64, _f32p)
_sig("vgo_sq8_train", None, _f32p, C.c_int64, C.c_int32, _f32p, _f32p, _f32p, _f32p)
_sig("vgo_sq8_encode", None, _f32p, C.c_int32, _f32p, _f32p, _f32p, _u8p)
_sig("vgo_sq8_decode", None, _u8p, C.c_int32, _f32p, _f32p, _f32p)
_sig("vgo_flat_search_sq8", C.c_int32, _u8p, C.c_int64, C.c_int32, _f32p, _f32p, _f32p, C.c_int32, _u32p, _f32p)
_sig("vgo_sq8_dot", C.c_float, _f32p, _u8p, C.c_int32, _f32p, _f32p)
_sig("vgo_flat_segment_search", C.c_int32, C.POINTER(FlatSeg), _f32p, C.c_int32, C.c_int32, _u32p, _f32p)
_sig("vgo_flat_segment_search_filtered", C.c_int32, C.POINTER(FlatSeg), _f32p, C.c_int32, C.c_int32, _u8p, _u32p, _f32p)
_sig("vgo_int4_l2", C.c_float, _f32p, _u8p, C.c_int64, _f32p, _f32p)
_sig("vgo_int4_l2_batch", None, _f32p, _u8p, C.c_int64, C.c_int64, _f32p, _f32p, _f32p)
_sig("vgo_int4_build_lut", None, _f32p, _f32p, C.c_int32, _f32p)
_sig("vgo_int4_l2_precomputed", C.c_float, _f32p, _u8p, C.c_int64, _f32p)
_sig("vgo_int4_train", None, _f32p, C.c_int64, C.c_int32, _f32p, _f32p)
_sig("vgo_int4_encode", None, _f32p, C.c_int32, _f32p, _f32p, _u8p)
_sig("vgo_int4_decode", None, _u8p, C.c_int32, _f32p, _f32p, _f32p)
_sig("vgo_robust_prune", C.c_int32, _f32p, C.c_int64, C.c_int32, C.c_int32, C.c_uint32, _u32p, C.c_int32, C.c_int32,
     C.c_float, _u32p)
_sig("vgo_hnsw_select_neighbors", C.c_int32, _f32p, C.c_int64, C.c_int32, C.c_int32, _u32p, _f32p, C.c_int32,
     C.c_int32, _u32p)
_sig("vgo_rerank_f32", None, _f32p, C.c_int32, C.c_int32, _f32p, _u32p, C.c_int32, _f32p)
_sig("vgo_hnsw_search", C.c_int32, C.POINTER(HnswGraph), _f32p, C.c_int32, C.c_int32, _u32p,
     _f32p, C.POINTER(SearchStats))
_sig("vgo_hnsw_brute_search", C.c_int32, C.POINTER(HnswGraph), _f32p, C.c_int32, C.c_int32, _u8p, _u32p, _f32p)
_sig("vgo_vamana_search", C.c_int32, C.POINTER(Vamana), _f32p, C.c_int32, _u32p, _f32p,
     C.POINTER(SearchStats))
_sig("vgo_vamana_search_filtered", C.c_int32, C.POINTER(Vamana), _f32p, C.c_int32, _u8p, _u32p, _f32p,
     C.POINTER(SearchStats))
_i64p = C.POINTER(C.c_int64)
_sig("vgo_opq_block_size", C.c_int32, C.c_int32, C.c_int32)
_sig("vgo_opq_rotate", None, _f32p, C.c_int32, C.c_int32, _f32p, _f32p)
_sig("vgo_opq_unrotate", None, _f32p, C.c_int32, C.c_int32, _f32p, _f32p)
_sig("vgo_procrustes", None, _f32p, C.c_int32, _f32p)
_sig("vgo_opq_accumulate_m", None, _f32p, _f32p, C.c_int64, C.c_int32, C.c_int32, _f32p)
_sig("vgo_opq_train", C.c_int32, _f32p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
     C.c_uint64, _f32p, _i8p, _f32p, _f32p)
_sig("vgo_hnsw_level_for_id", C.c_int32, C.c_uint64, C.c_int32)
_sig("vgo_hnsw_build_layout", C.c_int32, C.c_int64, C.c_int32, _i32p, _i64p)
_sig("vgo_hnsw_build_batch", C.c_int64, C.c_int64, C.c_int64, C.c_int32, C.c_int32)
_sig("vgo_hnsw_build", C.c_int32, _f32p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
     C.c_int32, _u32p, _u32p, _u32p, _u32p, _i32p)


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(_f32p)


def _u8(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    return a, a.ctypes.data_as(_u8p)


def _i8(a):
    a = np.ascontiguousarray(a, dtype=np.int8)
    return a, a.ctypes.data_as(_i8p)


def _u32(a):
    a = np.ascontiguousarray(a, dtype=np.uint32)
    return a, a.ctypes.data_as(_u32p)


# ---- L0 -------------------------------------------------------------------
def dot(a, b):
    a, pa = _f(a); b, pb = _f(b)
    return np.float32(lib.vgo_dot_avx512(pa, pb, a.size))


def l2(a, b):
    a, pa = _f(a); b, pb = _f(b)
    return np.float32(lib.vgo_l2_avx512(pa, pb, a.size))


def l2_batch(q, targets, dim):
    q, pq_ = _f(q); t, pt = _f(targets)
    n = t.size // dim if dim else 0
    out = np.empty(n, np.float32)
    lib.vgo_l2_batch_avx512(pq_, pt, dim, n, out.ctypes.data_as(_f32p))
    return out


def dot_batch(q, targets, dim):
    q, pq_ = _f(q); t, pt = _f(targets)
    n = t.size // dim if dim else 0
    out = np.empty(n, np.float32)
    lib.vgo_dot_batch_avx512(pq_, pt, dim, n, out.ctypes.data_as(_f32p))
    return out


def l2_bounded(a, b, bound):
    a, pa = _f(a); b, pb = _f(b)
    r = C.c_float(); e = C.c_int32()
    lib.vgo_l2_bounded_avx512(pa, pb, a.size, C.c_float(bound), C.byref(r), C.byref(e))
    return np.float32(r.value), bool(e.value)


def adc(table, codes, m):
    t, pt = _f(table); c, pc = _u8(codes)
    return np.float32(lib.vgo_adc_avx512(pt, pc, m))


def adc_generic(table, codes, m):
    t, pt = _f(table); c, pc = _u8(codes)
    return np.float32(lib.vgo_adc_generic(pt, pc, m))


def hamming(a, b):
    a, pa = _u8(a); b, pb = _u8(b)
    return int(lib.vgo_hamming(pa, pb, a.size))


def l2_int8_deq(q, code, scale, offset):
    q, pq_ = _f(q); c, pc = _i8(code)
    return np.float32(lib.vgo_l2_int8_deq(pq_, pc, q.size, scale, offset))


def build_table_int8(q, codebook, subdim, scale, offset):
    q, pq_ = _f(q); cb, pcb = _i8(codebook)
    k = cb.size // subdim
    out = np.empty(k, np.float32)
    lib.vgo_build_table_int8(pq_, pcb, subdim, k, scale, offset, out.ctypes.data_as(_f32p))
    return out


def nearest_centroid_int8(q, codebook, subdim, scale, offset):
    q, pq_ = _f(q); cb, pcb = _i8(codebook)
    return int(lib.vgo_nearest_centroid_int8(pq_, pcb, subdim, cb.size // subdim, scale, offset))


# ---- PQ ---------------------------------------------------------------------
class ProductQuantizer:
    """Oracle-side PQ state (internal/quantization/pq.go:20-29)."""

    def __init__(self, dim, m, k=256):
        assert dim % m == 0 and 0 < k <= 256
        self.dim, self.m, self.k, self.subdim = dim, m, k, dim // m
        self.codebooks = np.zeros(m * k * self.subdim, np.int8)
        self.scales = np.zeros(m, np.float32)
        self.offsets = np.zeros(m, np.float32)
        self.centroids_f32 = None
        self.trained = False

    def _c(self):
        return PQ(self.dim, self.m, self.k, self.subdim, self.codebooks.ctypes.data_as(_i8p),
                  self.scales.ctypes.data_as(_f32p), self.offsets.ctypes.data_as(_f32p))

    def set_codebooks(self, codebooks, scales, offsets):
        self.codebooks = np.ascontiguousarray(codebooks, np.int8).reshape(-1)
        self.scales = np.ascontiguousarray(scales, np.float32)
        self.offsets = np.ascontiguousarray(offsets, np.float32)
        self.trained = True

    def train(self, vectors, iters=20, seed=1):
        v, pv = _f(vectors)
        n = v.size // self.dim
        self.centroids_f32 = np.zeros(self.m * self.k * self.subdim, np.float32)
        rc = lib.vgo_pq_train(pv, n, self.dim, self.m, self.k, iters, seed,
                              self.codebooks.ctypes.data_as(_i8p),
                              self.scales.ctypes.data_as(_f32p),
                              self.offsets.ctypes.data_as(_f32p),
                              self.centroids_f32.ctypes.data_as(_f32p))
        if rc != 0:
            raise ValueError("pq train failed")
        self.trained = True

    def encode(self, vec):
        v, pv = _f(vec)
        out = np.empty(self.m, np.uint8)
        c = self._c()
        lib.vgo_pq_encode(C.byref(c), pv, out.ctypes.data_as(_u8p))
        return out

    def encode_batch(self, vecs):
        v = np.ascontiguousarray(vecs, np.float32).reshape(-1, self.dim)
        out = np.empty((v.shape[0], self.m), np.uint8)
        c = self._c()
        for i in range(v.shape[0]):
            lib.vgo_pq_encode(C.byref(c), v[i].ctypes.data_as(_f32p), out[i].ctypes.data_as(_u8p))
        return out

    def decode(self, codes):
        cd, pc = _u8(codes)
        out = np.empty(self.dim, np.float32)
        c = self._c()
        lib.vgo_pq_decode(C.byref(c), pc, out.ctypes.data_as(_f32p))
        return out

    def build_table(self, query):
        q, pq_ = _f(query)
        out = np.empty(self.m * self.k, np.float32)
        c = self._c()
        lib.vgo_pq_build_table(C.byref(c), pq_, out.ctypes.data_as(_f32p))
        return out

    def asym_distance(self, query, codes):
        q, pq_ = _f(query); cd, pc = _u8(codes)
        c = self._c()
        return np.float32(lib.vgo_pq_asym_distance(C.byref(c), pq_, pc))


def pq_quantize_centroids(centroids):
    c, pc = _f(centroids)
    out = np.empty(c.size, np.int8)
    s = C.c_float(); o = C.c_float()
    lib.vgo_pq_quantize_centroids(pc, c.size, out.ctypes.data_as(_i8p), C.byref(s), C.byref(o))
    return out, np.float32(s.value), np.float32(o.value)


# ---- RaBitQ -----------------------------------------------------------------
def rabitq_code_bytes(dim):
    return int(lib.vgo_rabitq_code_bytes(dim))


def rabitq_encode(v):
    v, pv = _f(v)
    out = np.empty(rabitq_code_bytes(v.size), np.uint8)
    lib.vgo_rabitq_encode(pv, v.size, out.ctypes.data_as(_u8p))
    return out


def rabitq_encode_batch(vecs, dim):
    v = np.ascontiguousarray(vecs, np.float32).reshape(-1, dim)
    out = np.empty((v.shape[0], rabitq_code_bytes(dim)), np.uint8)
    for i in range(v.shape[0]):
        lib.vgo_rabitq_encode(v[i].ctypes.data_as(_f32p), dim, out[i].ctypes.data_as(_u8p))
    return out


def rabitq_distance(query, code):
    q, pq_ = _f(query); c, pc = _u8(code)
    return np.float32(lib.vgo_rabitq_distance(pq_, q.size, pc))


def binary_train(vectors, dim):
    v, pv = _f(vectors)
    return np.float32(lib.vgo_binary_train(pv, v.size // dim, dim))


def binary_decode(code, dim, threshold):
    c = np.ascontiguousarray(code, np.uint8)
    out = np.empty(dim, np.float32)
    lib.vgo_binary_decode(c.ctypes.data_as(_u8p), c.size, dim, threshold, out.ctypes.data_as(_f32p))
    return out


def normalize_l2(v):
    """(normalized copy, ok) — distance.NormalizeL2Copy"""
    a = np.array(v, np.float32, copy=True)
    ok = lib.vgo_normalize_l2(a.ctypes.data_as(_f32p), a.size)
    return a, bool(ok)


def binary_encode_u64(v, threshold=0.0):
    v, pv = _f(v)
    out = np.zeros((v.size + 63) // 64, np.uint64)
    lib.vgo_binary_encode_u64(pv, v.size, threshold, out.ctypes.data_as(_u64p))
    return out


# ---- kmeans -------------------------------------------------------------------
def kmeans_train(vectors, dim, k, metric=METRIC_L2, max_iter=10, seed=1):
    v, pv = _f(vectors)
    n = v.size // dim
    out = np.zeros(k * dim, np.float32)
    rc = lib.vgo_kmeans_train(pv, n, dim, k, metric, max_iter, seed, out.ctypes.data_as(_f32p))
    if rc == 1:
        return None
    if rc < 0:
        raise ValueError("unsupported metric")
    return out


def assign_partition(vec, centroids, dim, metric=METRIC_L2):
    v, pv = _f(vec); c, pc = _f(centroids)
    return int(lib.vgo_assign_partition(pv, pc, dim, c.size // dim, metric))


def find_closest_centroids(query, centroids, dim, n, metric=METRIC_L2):
    q, pq_ = _f(query); c, pc = _f(centroids)
    k = c.size // dim
    out = np.empty(max(min(n, k), 1), np.int32)
    r = lib.vgo_find_closest_centroids(pq_, pc, dim, k, n, metric, out.ctypes.data_as(_i32p))
    if r < 0:
        raise ValueError("unsupported metric")
    return out[:r]


def rng_u64(seed, a, b, c):
    return int(lib.vgo_rng_u64(seed, a, b, c))


# ---- SQ8 ----------------------------------------------------------------------
def sq8u_l2_batch(query, codes, mins, inv_scales, dim):
    q, pq_ = _f(query); c, pc = _u8(codes); mn, pm = _f(mins); iv, pi = _f(inv_scales)
    n = c.size // dim if dim else 0
    out = np.zeros(n, np.float32)
    if n and dim:  # kernels_amd64.go:363-375: nothing happens for an empty batch
        lib.vgo_sq8u_l2_batch(pq_, pc, pm, pi, dim, n, out.ctypes.data_as(_f32p))
    return out


class ScalarQuantizer:
    """Oracle-side quantization.ScalarQuantizer (internal/quantization/quantizer.go:27-250)."""

    def __init__(self, dim):
        self.dim = dim
        self.mins = np.zeros(dim, np.float32); self.maxs = np.zeros(dim, np.float32)
        self.scales = np.zeros(dim, np.float32); self.inv_scales = np.zeros(dim, np.float32)
        self.trained = False

    def train(self, vectors):
        v, pv = _f(vectors)
        lib.vgo_sq8_train(pv, v.size // self.dim, self.dim, self.mins.ctypes.data_as(_f32p),
                          self.maxs.ctypes.data_as(_f32p), self.scales.ctypes.data_as(_f32p),
                          self.inv_scales.ctypes.data_as(_f32p))
        self.trained = True

    def encode(self, vec):
        v, pv = _f(vec)
        out = np.empty(self.dim, np.uint8)
        lib.vgo_sq8_encode(pv, self.dim, self.mins.ctypes.data_as(_f32p), self.maxs.ctypes.data_as(_f32p),
                           self.scales.ctypes.data_as(_f32p), out.ctypes.data_as(_u8p))
        return out

    def encode_batch(self, vecs):
        v = np.ascontiguousarray(vecs, np.float32).reshape(-1, self.dim)
        return np.stack([self.encode(r) for r in v]) if len(v) else np.zeros((0, self.dim), np.uint8)

    def decode(self, code):
        c, pc = _u8(code)
        out = np.empty(self.dim, np.float32)
        lib.vgo_sq8_decode(pc, self.dim, self.mins.ctypes.data_as(_f32p), self.inv_scales.ctypes.data_as(_f32p),
                           out.ctypes.data_as(_f32p))
        return out


def flat_search_sq8(sq: ScalarQuantizer, codes, query, k):
    c, pc = _u8(codes); q, pq_ = _f(query)
    n = c.size // sq.dim
    ids = np.empty(k, np.uint32); sc = np.empty(k, np.float32)
    r = lib.vgo_flat_search_sq8(pc, n, sq.dim, sq.mins.ctypes.data_as(_f32p), sq.inv_scales.ctypes.data_as(_f32p),
                                pq_, k, ids.ctypes.data_as(_u32p), sc.ctypes.data_as(_f32p))
    return ids[:r], sc[:r]


# ---- INT4 ---------------------------------------------------------------------
class Int4Quantizer:
    """Oracle-side quantization.Int4Quantizer (internal/quantization/int4.go)."""

    def __init__(self, dim):
        self.dim = dim
        self.min = np.zeros(dim, np.float32); self.diff = np.ones(dim, np.float32)
        self.table = None

    def set_params(self, min_val, diff):  # UnmarshalBinary (int4.go:190-219)
        self.min = np.ascontiguousarray(min_val, np.float32); self.diff = np.ascontiguousarray(diff, np.float32)
        self._build()

    def _build(self):
        self.table = np.empty(self.dim * 16, np.float32)
        lib.vgo_int4_build_lut(self.min.ctypes.data_as(_f32p), self.diff.ctypes.data_as(_f32p), self.dim,
                               self.table.ctypes.data_as(_f32p))

    def train(self, vectors):
        v, pv = _f(vectors)
        lib.vgo_int4_train(pv, v.size // self.dim, self.dim, self.min.ctypes.data_as(_f32p),
                           self.diff.ctypes.data_as(_f32p))
        self._build()

    @property
    def code_size(self):
        return (self.dim + 1) // 2

    def encode(self, vec):
        v, pv = _f(vec)
        out = np.empty(self.code_size, np.uint8)
        lib.vgo_int4_encode(pv, self.dim, self.min.ctypes.data_as(_f32p), self.diff.ctypes.data_as(_f32p),
                            out.ctypes.data_as(_u8p))
        return out

    def encode_batch(self, vecs):
        v = np.ascontiguousarray(vecs, np.float32).reshape(-1, self.dim)
        return np.stack([self.encode(r) for r in v]) if len(v) else np.zeros((0, self.code_size), np.uint8)

    def decode(self, code):
        c, pc = _u8(code)
        out = np.empty(self.dim, np.float32)
        lib.vgo_int4_decode(pc, self.dim, self.min.ctypes.data_as(_f32p), self.diff.ctypes.data_as(_f32p),
                            out.ctypes.data_as(_f32p))
        return out

    def l2_distance(self, query, code):
        """L2Distance (int4.go:133-147): the table exists after Train / UnmarshalBinary."""
        q, pq_ = _f(query); c, pc = _u8(code)
        return np.float32(lib.vgo_int4_l2_precomputed(pq_, pc, self.dim, self.table.ctypes.data_as(_f32p)))

    def l2_distance_batch(self, query, codes):
        """L2DistanceBatch (int4.go:150-164) = simd.Int4L2DistanceBatch."""
        q, pq_ = _f(query); c, pc = _u8(codes)
        n = c.size // self.code_size
        out = np.zeros(n, np.float32)
        if n:
            lib.vgo_int4_l2_batch(pq_, pc, self.dim, n, self.min.ctypes.data_as(_f32p),
                                  self.diff.ctypes.data_as(_f32p), out.ctypes.data_as(_f32p))
        return out


# ---- construction-time neighbour selection -------------------------------------
def robust_prune(base, dim, node, cands, r, alpha, metric=METRIC_L2):
    b, pb = _f(base)
    c = np.ascontiguousarray(cands, np.uint32)
    out = np.empty(r, np.uint32)
    k = lib.vgo_robust_prune(pb, b.size // dim, dim, metric, node, c.ctypes.data_as(_u32p), c.size, r,
                             C.c_float(alpha), out.ctypes.data_as(_u32p))
    return out[:k]


def hnsw_select_neighbors(base, dim, cand_ids, cand_dists, m, metric=METRIC_L2):
    b, pb = _f(base)
    ci = np.ascontiguousarray(cand_ids, np.uint32); cd = np.ascontiguousarray(cand_dists, np.float32)
    out = np.empty(m, np.uint32)
    k = lib.vgo_hnsw_select_neighbors(pb, b.size // dim, dim, metric, ci.ctypes.data_as(_u32p),
                                      cd.ctypes.data_as(_f32p), ci.size, m, out.ctypes.data_as(_u32p))
    return out[:k]


class OptimizedProductQuantizer:
    """quantization.OptimizedProductQuantizer (opq.go): block-diagonal rotation + ProductQuantizer."""

    def __init__(self, dim, m, k=256, num_iterations=2):
        self.dim, self.m, self.k, self.num_iterations = dim, m, k, num_iterations
        self.block = int(lib.vgo_opq_block_size(dim, m))
        self.nblocks = dim // self.block
        self.rotations = np.tile(np.eye(self.block, dtype=np.float32), (self.nblocks, 1, 1))
        self.pq = ProductQuantizer(dim, m, k)
        self.trained = False

    def train(self, vectors, pq_iters=20, seed=1):
        v, pv = _f(vectors)
        n = v.size // self.dim
        sd = self.dim // self.m
        cb = np.zeros(self.m * self.k * sd, np.int8); sc = np.zeros(self.m, np.float32); of = np.zeros(self.m, np.float32)
        rot = np.zeros((self.nblocks, self.block, self.block), np.float32)
        r = lib.vgo_opq_train(pv, n, self.dim, self.m, self.k, self.num_iterations, pq_iters, seed,
                              rot.ctypes.data_as(_f32p), cb.ctypes.data_as(_i8p), sc.ctypes.data_as(_f32p),
                              of.ctypes.data_as(_f32p))
        if r != 0:
            raise ValueError("vgo_opq_train failed")
        self.rotations = rot
        self.pq.set_codebooks(cb, sc, of)
        self.trained = True

    def rotate(self, vec):
        v, pv = _f(vec)
        out = np.empty(self.dim, np.float32)
        lib.vgo_opq_rotate(self.rotations.ctypes.data_as(_f32p), self.dim, self.block, pv, out.ctypes.data_as(_f32p))
        return out

    def encode(self, vec):
        return self.pq.encode(self.rotate(vec))

    def decode(self, codes):
        r = np.ascontiguousarray(self.pq.decode(codes), np.float32)
        out = np.empty(self.dim, np.float32)
        lib.vgo_opq_unrotate(self.rotations.ctypes.data_as(_f32p), self.dim, self.block, r.ctypes.data_as(_f32p),
                             out.ctypes.data_as(_f32p))
        return out

    def asym_distance(self, query, codes):
        return self.pq.asym_distance(self.rotate(query), codes)


def procrustes(m_matrix):
    m = np.array(m_matrix, np.float32, copy=True)
    n = m.shape[0]
    r = np.empty((n, n), np.float32)
    lib.vgo_procrustes(m.ctypes.data_as(_f32p), n, r.ctypes.data_as(_f32p))
    return r


def opq_accumulate_m(x, y, dim, block):
    a, pa = _f(x); b, pb = _f(y)
    out = np.empty((dim // block, block, block), np.float32)
    lib.vgo_opq_accumulate_m(pa, pb, a.size // dim, dim, block, out.ctypes.data_as(_f32p))
    return out


def hnsw_layout(n, m):
    """levels[n] of ApplyInsert's ids (layerForApplyInsert hnsw.go:2103-2116), rows per upper level, top level."""
    levels = np.empty(n, np.int32)
    rows = np.zeros(63, np.int64)
    top = lib.vgo_hnsw_build_layout(n, m, levels.ctypes.data_as(_i32p), rows.ctypes.data_as(_i64p))
    return levels, rows[:top].copy(), top


def hnsw_build(base, dim, m=32, ef=300, metric=METRIC_L2, max_batch=1, growth_div=32):
    """hnsw insert loop (hnsw.go:713-984) over rows 0..n-1; returns (l0[n,2m], upper, entry_point) with
    upper = [(slot[n], adj[rows, m]) per level 1..top] — the layout HnswIndex / the C-ABI take."""
    b, pb = _f(base)
    n = b.size // dim
    _, rows, top = hnsw_layout(n, m)
    l0 = np.empty((n, 2 * m), np.uint32)
    slots = np.empty((max(top, 1), n), np.uint32)
    adj = np.empty((max(int(rows.sum()), 1), m), np.uint32)
    entry = C.c_uint32(0)
    mx = C.c_int32(0)
    r = lib.vgo_hnsw_build(pb, n, dim, metric, m, ef, max_batch, growth_div, l0.ctypes.data_as(_u32p),
                           slots.ctypes.data_as(_u32p), adj.ctypes.data_as(_u32p), C.byref(entry), C.byref(mx))
    if r != 0:
        raise ValueError("vgo_hnsw_build: bad arguments")
    upper, off = [], 0
    for l in range(top):
        upper.append((slots[l].copy(), adj[off:off + int(rows[l])].copy()))
        off += int(rows[l])
    return l0, upper, int(entry.value)


# ---- scans --------------------------------------------------------------------
def flat_search_f32(base, dim, query, k, metric=METRIC_L2):
    b, pb = _f(base); q, pq_ = _f(query)
    n = b.size // dim
    ids = np.empty(k, np.uint32); sc = np.empty(k, np.float32)
    r = lib.vgo_flat_search_f32(pb, n, dim, metric, pq_, k, ids.ctypes.data_as(_u32p),
                                sc.ctypes.data_as(_f32p))
    return ids[:r], sc[:r]


def flat_search_pq(pq: ProductQuantizer, codes, query, k):
    cd, pc = _u8(codes); q, pq_ = _f(query)
    n = cd.size // pq.m
    ids = np.empty(k, np.uint32); sc = np.empty(k, np.float32)
    c = pq._c()
    r = lib.vgo_flat_search_pq(C.byref(c), pc, n, pq_, k, ids.ctypes.data_as(_u32p),
                               sc.ctypes.data_as(_f32p))
    return ids[:r], sc[:r]


def flat_search_rabitq(codes, dim, query, k):
    cd, pc = _u8(codes); q, pq_ = _f(query)
    n = cd.size // rabitq_code_bytes(dim)
    ids = np.empty(k, np.uint32); sc = np.empty(k, np.float32)
    r = lib.vgo_flat_search_rabitq(pc, n, dim, pq_, k, ids.ctypes.data_as(_u32p),
                                   sc.ctypes.data_as(_f32p))
    return ids[:r], sc[:r]


def rerank_f32(base, dim, query, ids, metric=METRIC_L2):
    b, pb = _f(base); q, pq_ = _f(query); i, pi = _u32(ids)
    out = np.empty(i.size, np.float32)
    lib.vgo_rerank_f32(pb, dim, metric, pq_, pi, i.size, out.ctypes.data_as(_f32p))
    return out


class HnswIndex:
    """Host-side graph in the layout the C-ABI uploads (see include/vecgo_hip.h)."""

    def __init__(self, base, dim, l0, upper=(), entry_point=0, metric=METRIC_L2, m=None, pq=None, codes=None):
        self.pq = pq  # score nodes from PQ codes (ComputeAsymmetricDistance) instead of fp32 rows
        self.codes = None if codes is None else np.ascontiguousarray(codes, np.uint8)
        self.base = np.ascontiguousarray(base, np.float32).reshape(-1, dim)
        self.dim = dim
        self.n = self.base.shape[0]
        self.l0 = np.ascontiguousarray(l0, np.uint32)
        self.m0 = self.l0.shape[1]
        # upper: list of (slot[n] u32, adj[count, m] u32) for levels 1..L
        self.upper = [(np.ascontiguousarray(s, np.uint32), np.ascontiguousarray(a, np.uint32))
                      for s, a in upper]
        self.m = m if m is not None else (self.upper[0][1].shape[1] if self.upper else self.m0 // 2)
        self.entry_point = entry_point
        self.metric = metric

    def _c(self):
        nl = len(self.upper)
        self._slots = (_u32p * max(nl, 1))(*[s.ctypes.data_as(_u32p) for s, _ in self.upper])
        self._adjs = (_u32p * max(nl, 1))(*[a.ctypes.data_as(_u32p) for _, a in self.upper])
        self._pqc = self.pq._c() if self.pq is not None else None
        return HnswGraph(self.n, self.dim, self.metric, self.base.ctypes.data_as(_f32p), self.m0,
                         self.l0.ctypes.data_as(_u32p), nl, self.m, self._slots, self._adjs,
                         self.entry_point, C.pointer(self._pqc) if self._pqc is not None else None,
                         self.codes.ctypes.data_as(_u8p) if self.codes is not None else None,
                         self._tomb.ctypes.data_as(_u8p) if getattr(self, "_tomb", None) is not None else None)

    def set_tombstones(self, deleted):
        """g.tombstones (hnsw.go:95): bool[n], None clears — deleted nodes are walked through, never returned"""
        self._tomb = None if deleted is None else np.packbits(np.asarray(deleted, np.bool_).reshape(self.n), bitorder="little")

    def search(self, query, k, ef):
        q, pq_ = _f(query)
        ids = np.empty(max(k, 1), np.uint32); sc = np.empty(max(k, 1), np.float32)
        st = SearchStats()
        g = self._c()
        r = lib.vgo_hnsw_search(C.byref(g), pq_, k, ef, ids.ctypes.data_as(_u32p),
                                sc.ctypes.data_as(_f32p), C.byref(st))
        return ids[:r], sc[:r], st


BRUTE_SCAN, BRUTE_BITMAP = 0, 1

_sig("vgo_prioq_replay", C.c_int32, C.c_int32, _i32p, C.c_int32, _i32p, C.POINTER(C.c_uint64), C.c_int32)
HEAP_OPS = {"push": 0, "pop": 1, "push_bounded": 2, "try_push_bounded": 3, "top": 4, "min_item": 5, "reset": 6, "len": 7}


def heap_script_array(script):
    """[{op, node, dist, cap}, ...] (tests/golden/reference_kats.json searcher_priority_queue) or rows of
    (op, node, dist, arg) -> int32[n, 4] in the encoding vgo_prioq_replay and vg_debug_heap_replay share."""
    ops = np.zeros((len(script), 4), np.int32)
    for i, st in enumerate(script):
        if isinstance(st, dict):
            st = (HEAP_OPS[st["op"]], st.get("node", 0), st.get("dist", 0.0), st.get("cap", 0))
        ops[i] = (st[0], int(np.array(st[1], np.uint32).view(np.int32)),
                  int(np.array(st[2], np.float32).view(np.int32)), st[3])
    return ops


def prioq_replay(is_max, script, cap=4096):
    """-> (out[n, 3] int32 {flag, node, dist bits}, final heap items as (node[len] u32, dist[len] f32))."""
    ops = heap_script_array(script)
    out = np.zeros((ops.shape[0], 3), np.int32)
    items = np.zeros(cap, np.uint64)
    n = lib.vgo_prioq_replay(int(bool(is_max)), ops.ctypes.data_as(_i32p), ops.shape[0], out.ctypes.data_as(_i32p),
                             items.ctypes.data_as(C.POINTER(C.c_uint64)), cap)
    items = items[:n]
    return out, ((items & 0xFFFFFFFF).astype(np.uint32), (items >> 32).astype(np.uint32).view(np.float32))


_sig("vgo_visited_replay", C.c_int32, C.c_int32, _i64p, C.c_int32, _i64p)
VISITED_OPS = {"visit": 0, "visited": 1, "check_and_visit": 2, "reset": 3, "ensure_capacity": 4, "capacity": 5}


def visited_replay(capacity, script):
    """[(op name, id)] through the literal restatement of searcher.VisitedSet -> list of results"""
    ops = np.array([[VISITED_OPS[op], arg] for op, arg in script], np.int64).reshape(-1, 2)
    out = np.zeros(ops.shape[0], np.int64)
    lib.vgo_visited_replay(capacity, ops.ctypes.data_as(_i64p), ops.shape[0], out.ctypes.data_as(_i64p))
    return out.tolist()


class _Cand(C.Structure):
    _fields_ = [("segment_id", C.c_uint32), ("row_id", C.c_uint32), ("score", C.c_float)]


class _CandHeap(C.Structure):
    _fields_ = [("c", C.POINTER(_Cand)), ("len", C.c_int32), ("cap", C.c_int32), ("descending", C.c_int32)]


_sig("vgo_candheap_init", None, C.POINTER(_CandHeap), C.c_int32, C.c_int32)
_sig("vgo_candheap_free", None, C.POINTER(_CandHeap))
_sig("vgo_cand_better", C.c_int, _Cand, _Cand, C.c_int)
_sig("vgo_candheap_try_push_bounded", C.c_int, C.POINTER(_CandHeap), _Cand, C.c_int32)
_sig("vgo_candheap_pop", C.c_int, C.POINTER(_CandHeap), C.POINTER(_Cand))
_sig("vgo_candheap_replace_top", C.c_int, C.POINTER(_CandHeap), _Cand)


class CandidateHeap:
    """searcher.CandidateHeap (candidate_queue.go) as the oracle restates it; Push = TryPushBounded with k = inf."""

    def __init__(self, descending, cap=16):
        self.h = _CandHeap()
        lib.vgo_candheap_init(C.byref(self.h), cap, int(bool(descending)))

    def __len__(self):
        return self.h.len

    def push(self, score, segment_id=0, row_id=0, k=1 << 30):
        return bool(lib.vgo_candheap_try_push_bounded(C.byref(self.h), _Cand(segment_id, row_id, score), k))

    def top(self):
        c = self.h.c[0]
        return (c.score, c.segment_id, c.row_id)

    def pop(self):
        c = _Cand()
        ok = lib.vgo_candheap_pop(C.byref(self.h), C.byref(c))
        return (c.score, c.segment_id, c.row_id) if ok else None

    def replace_top(self, score, segment_id=0, row_id=0):
        return bool(lib.vgo_candheap_replace_top(C.byref(self.h), _Cand(segment_id, row_id, score)))

    def close(self):
        lib.vgo_candheap_free(C.byref(self.h))


def cand_better(a, b, descending):
    """a, b = (score, segment_id, row_id)"""
    return bool(lib.vgo_cand_better(_Cand(a[1], a[2], a[0]), _Cand(b[1], b[2], b[0]), int(bool(descending))))


def _brute_search(self, query, k, mode=BRUTE_SCAN, mask=None):
    """hnsw.BruteSearch / scanSegment (mode BRUTE_SCAN) or searchBitmap + extraction (BRUTE_BITMAP) over the rows
    whose bit in `mask` (bool[n] or packed little-endian bits, None = all) is set."""
    q, pq_ = _f(query)
    ids = np.empty(max(k, 1), np.uint32); sc = np.empty(max(k, 1), np.float32)
    mp = None
    if mask is not None:
        mask = np.asarray(mask)
        if mask.dtype == np.bool_:
            mask = np.packbits(mask, bitorder="little")
        mask, mp = _u8(mask)
    g = self._c()
    r = lib.vgo_hnsw_brute_search(C.byref(g), pq_, k, mode, mp, ids.ctypes.data_as(_u32p), sc.ctypes.data_as(_f32p))
    return ids[:r], sc[:r]


HnswIndex.brute_search = _brute_search

_sig("vgo_hnsw_search_filtered", C.c_int32, C.POINTER(HnswGraph), _f32p, C.c_int32, C.c_int32, _u8p, C.c_double, _u32p, _f32p,
     C.POINTER(SearchStats))


def _search_filtered(self, query, k, ef, mask, selectivity):
    """searchExecute with a filter and a selectivity hint > 0.3: searchLayerWithPostFilter (hnsw.go:1159-1218).  mask:
    bool[n] or packed little-endian bits.  Returns (ids, scores, stats) or None when the selectivity is <= 0.3."""
    q, pq_ = _f(query)
    ids = np.empty(max(k, ef, 1), np.uint32); sc = np.empty(max(k, ef, 1), np.float32)
    mask = np.asarray(mask)
    if mask.dtype == np.bool_:
        mask = np.packbits(mask, bitorder="little")
    mask, mp = _u8(mask)
    st = SearchStats()
    g = self._c()
    r = lib.vgo_hnsw_search_filtered(C.byref(g), pq_, k, ef, mp, float(selectivity), ids.ctypes.data_as(_u32p),
                                     sc.ctypes.data_as(_f32p), C.byref(st))
    if r < 0:
        return None
    return ids[:r], sc[:r], st


HnswIndex.search_filtered = _search_filtered

_sig("vgo_hnsw_search_predicate", C.c_int32, C.POINTER(HnswGraph), _f32p, C.c_int32, C.c_int32, _u8p, _u8p, _f32p, _u32p, _f32p,
     C.POINTER(SearchStats))


def _search_predicate(self, query, k, ef, mask, deleted=None, l0_dist=None):
    """searchExecute with a filter whose selectivity hint is <= 0.3 or unknown: searchLayerPredicateAware (hnsw.go:1406-1558).
    mask / deleted: bool[n]; l0_dist: the cached Neighbor.Dist of the layer-0 lists [n, m0] (None = recomputed from the rows)."""
    q, pq_ = _f(query)
    ids = np.empty(max(k, ef, 1), np.uint32); sc = np.empty(max(k, ef, 1), np.float32)
    mb = np.packbits(np.asarray(mask, np.bool_).reshape(self.n), bitorder="little")
    db = None if deleted is None else np.packbits(np.asarray(deleted, np.bool_).reshape(self.n), bitorder="little")
    ld = None if l0_dist is None else np.ascontiguousarray(l0_dist, np.float32)
    st = SearchStats()
    g = self._c()
    r = lib.vgo_hnsw_search_predicate(C.byref(g), pq_, k, ef, mb.ctypes.data_as(_u8p),
                                      db.ctypes.data_as(_u8p) if db is not None else None,
                                      ld.ctypes.data_as(_f32p) if ld is not None else None,
                                      ids.ctypes.data_as(_u32p), sc.ctypes.data_as(_f32p), C.byref(st))
    return ids[:r], sc[:r], st


HnswIndex.search_predicate = _search_predicate


class VamanaIndex:
    def __init__(self, graph, entry_point, dim, kind=VAMANA_F32, metric=METRIC_L2, base=None,
                 pq: ProductQuantizer | None = None, codes=None, int4_table=None):
        self.int4_table = None if int4_table is None else np.ascontiguousarray(int4_table, np.float32)
        self.graph = np.ascontiguousarray(graph, np.uint32)
        self.n, self.r = self.graph.shape
        self.entry_point, self.dim, self.kind, self.metric = entry_point, dim, kind, metric
        self.base = None if base is None else np.ascontiguousarray(base, np.float32)
        self.pq = pq
        self.codes = None if codes is None else np.ascontiguousarray(codes, np.uint8)

    def _c(self):
        self._pqc = self.pq._c() if self.pq is not None else None
        return Vamana(self.n, self.dim, self.r, self.graph.ctypes.data_as(_u32p), self.entry_point,
                      self.kind, self.metric,
                      self.base.ctypes.data_as(_f32p) if self.base is not None else None,
                      C.pointer(self._pqc) if self._pqc is not None else None,
                      self.codes.ctypes.data_as(_u8p) if self.codes is not None else None,
                      self.int4_table.ctypes.data_as(_f32p) if self.int4_table is not None else None)

    def search(self, query, k, mask=None):
        """mask: bool[n] = filter.Matches per row (diskann/segment.go:616-627); None = no filter"""
        q, pq_ = _f(query)
        ids = np.empty(max(k, 1), np.uint32); sc = np.empty(max(k, 1), np.float32)
        st = SearchStats()
        v = self._c()
        pm = None
        if mask is not None:
            bits_ = np.packbits(np.asarray(mask, np.bool_).reshape(self.n), bitorder="little")
            pm = bits_.ctypes.data_as(_u8p)
        r = lib.vgo_vamana_search_filtered(C.byref(v), pq_, k, pm, ids.ctypes.data_as(_u32p),
                                           sc.ctypes.data_as(_f32p), C.byref(st))
        return ids[:r], sc[:r], st


class FlatSegment:
    """flat.Segment.Search (flat/segment.go:447-751) over a whole flat segment: scan type by
    quantization, IVF partitions (centroids [P, dim], part_offsets [P + 1]) probed when P > 1."""

    def __init__(self, base, dim, metric=METRIC_L2, pq: "ProductQuantizer | None" = None, codes=None,
                 sq: "ScalarQuantizer | None" = None, centroids=None, part_offsets=None):
        self.base = np.ascontiguousarray(base, np.float32).reshape(-1, dim)
        self.n, self.dim, self.metric = self.base.shape[0], dim, metric
        self.pq, self.sq = pq, sq
        self.codes = None if codes is None else np.ascontiguousarray(codes, np.uint8)
        self.centroids = None if centroids is None else np.ascontiguousarray(centroids, np.float32).reshape(-1, dim)
        self.part_offsets = None if part_offsets is None else np.ascontiguousarray(part_offsets, np.uint32)
        self.num_partitions = 0 if self.centroids is None else self.centroids.shape[0]

    def search(self, query, k, nprobes=0, mask=None):
        """mask: bool[n] = filter.Matches per row (segment.go:631-635); None = no filter"""
        q, pq_ = _f(query)
        ids = np.empty(max(k, 1), np.uint32); sc = np.empty(max(k, 1), np.float32)
        pqc = self.pq._c() if self.pq is not None else None
        seg = FlatSeg(self.n, self.dim, self.metric, self.base.ctypes.data_as(_f32p),
                      C.pointer(pqc) if pqc is not None else None,
                      self.codes.ctypes.data_as(_u8p) if self.codes is not None else None,
                      self.sq.mins.ctypes.data_as(_f32p) if self.sq is not None else None,
                      self.sq.inv_scales.ctypes.data_as(_f32p) if self.sq is not None else None,
                      self.num_partitions,
                      self.centroids.ctypes.data_as(_f32p) if self.centroids is not None else None,
                      self.part_offsets.ctypes.data_as(_u32p) if self.part_offsets is not None else None)
        pm = None
        if mask is not None:
            bits_ = np.packbits(np.asarray(mask, np.bool_).reshape(self.n), bitorder="little")
            pm = bits_.ctypes.data_as(_u8p)
        r = lib.vgo_flat_segment_search_filtered(C.byref(seg), pq_, k, nprobes, pm, ids.ctypes.data_as(_u32p),
                                                 sc.ctypes.data_as(_f32p))
        return ids[:r], sc[:r]


# ---- the compiled reference objects (only where oracle/_ref was built) -------
def load_ref():
    """oracle/_ref/libvecgo_ref_avx512.so: the reference's own AVX-512 C kernels compiled
    from /root/reference (never copied). Returns None when absent or the CPU lacks AVX-512."""
    path = _HERE / "_ref" / "libvecgo_ref_avx512.so"
    need = {"avx512f", "avx512dq", "avx512bw", "avx512vl", "avx512_vpopcntdq"}
    if not path.exists() or not need <= _cpu_flags():
        return None
    r = C.CDLL(str(path))
    vp = C.c_void_p
    r.dotProductAvx512.argtypes = [vp, vp, C.c_int64, vp]
    r.squaredL2Avx512.argtypes = [vp, vp, C.c_int64, vp]
    r.squaredL2BatchAvx512.argtypes = [vp, vp, C.c_int64, C.c_int64, vp]
    r.dotBatchAvx512.argtypes = [vp, vp, C.c_int64, C.c_int64, vp]
    r.squaredL2BoundedAvx512.argtypes = [vp, vp, C.c_int64, C.c_float, vp, vp]
    r.pqAdcLookupAvx512.argtypes = [vp, vp, C.c_int64, vp, vp]
    r.hammingAvx512.argtypes = [vp, vp, C.c_int64]
    r.hammingAvx512.restype = C.c_longlong
    for n in ("dotProductAvx512", "squaredL2Avx512", "squaredL2BatchAvx512", "dotBatchAvx512",
              "squaredL2BoundedAvx512", "pqAdcLookupAvx512"):
        getattr(r, n).restype = None
    if hasattr(r, "int4L2DistanceBatchAvx512"):
        r.int4L2DistanceAvx512.argtypes = [vp, vp, C.c_int64, vp, vp, vp]
        r.int4L2DistancePrecomputedAvx512.argtypes = [vp, vp, C.c_int64, vp, vp]
        r.int4L2DistanceBatchAvx512.argtypes = [vp, vp, C.c_int64, C.c_int64, vp, vp, vp]
        for n in ("int4L2DistanceAvx512", "int4L2DistancePrecomputedAvx512", "int4L2DistanceBatchAvx512"):
            getattr(r, n).restype = None
    if hasattr(r, "sq8uL2BatchPerDimensionAvx512"):  # sq8_avx512.c joined the recipe later
        r.sq8uL2BatchPerDimensionAvx512.argtypes = [vp, vp, vp, vp, C.c_int64, C.c_int64, vp]
        r.sq8uL2BatchPerDimensionAvx512.restype = None
    return r


class Ref:
    """numpy-friendly calls into the compiled reference kernels."""

    def __init__(self):
        self.lib = load_ref()
        # kernels_amd64.go:38-44 pqAdcOffsets[i] = i*256
        self._off = (np.arange(256, dtype=np.int32) * 256)

    @property
    def ok(self):
        return self.lib is not None

    def dot(self, a, b):
        a = np.ascontiguousarray(a, np.float32); b = np.ascontiguousarray(b, np.float32)
        r = np.zeros(1, np.float32)
        if a.size:  # kernels_amd64.go:291-297: empty → 0
            self.lib.dotProductAvx512(a.ctypes.data, b.ctypes.data, a.size, r.ctypes.data)
        return r[0]

    def l2(self, a, b):
        a = np.ascontiguousarray(a, np.float32); b = np.ascontiguousarray(b, np.float32)
        r = np.zeros(1, np.float32)
        if a.size:
            self.lib.squaredL2Avx512(a.ctypes.data, b.ctypes.data, a.size, r.ctypes.data)
        return r[0]

    def l2_batch(self, q, t, dim):
        q = np.ascontiguousarray(q, np.float32); t = np.ascontiguousarray(t, np.float32)
        n = t.size // dim
        out = np.zeros(n, np.float32)
        if n:
            self.lib.squaredL2BatchAvx512(q.ctypes.data, t.ctypes.data, dim, n, out.ctypes.data)
        return out

    def dot_batch(self, q, t, dim):
        q = np.ascontiguousarray(q, np.float32); t = np.ascontiguousarray(t, np.float32)
        n = t.size // dim
        out = np.zeros(n, np.float32)
        if n:
            self.lib.dotBatchAvx512(q.ctypes.data, t.ctypes.data, dim, n, out.ctypes.data)
        return out

    def l2_bounded(self, a, b, bound):
        a = np.ascontiguousarray(a, np.float32); b = np.ascontiguousarray(b, np.float32)
        if a.size == 0:
            return np.float32(0), False  # kernels_amd64.go:308-310
        r = np.zeros(1, np.float32); e = np.zeros(1, np.int32)
        self.lib.squaredL2BoundedAvx512(a.ctypes.data, b.ctypes.data, a.size, C.c_float(bound),
                                        r.ctypes.data, e.ctypes.data)
        return r[0], bool(e[0])

    def adc(self, table, codes, m):
        t = np.ascontiguousarray(table, np.float32); c = np.ascontiguousarray(codes, np.uint8)
        r = np.zeros(1, np.float32)
        if m > 0:
            self.lib.pqAdcLookupAvx512(t.ctypes.data, c.ctypes.data, m, r.ctypes.data,
                                       self._off.ctypes.data)
        return r[0]

    def int4_l2(self, query, code, min_val, diff):
        q = np.ascontiguousarray(query, np.float32); c = np.ascontiguousarray(code, np.uint8)
        mn = np.ascontiguousarray(min_val, np.float32); df = np.ascontiguousarray(diff, np.float32)
        r = np.zeros(1, np.float32)
        if q.size:  # kernels_amd64.go: empty query -> 0
            self.lib.int4L2DistanceAvx512(q.ctypes.data, c.ctypes.data, q.size, mn.ctypes.data, df.ctypes.data,
                                          r.ctypes.data)
        return r[0]

    def int4_l2_precomputed(self, query, code, table):
        q = np.ascontiguousarray(query, np.float32); c = np.ascontiguousarray(code, np.uint8)
        t = np.ascontiguousarray(table, np.float32)
        r = np.zeros(1, np.float32)
        if q.size:
            self.lib.int4L2DistancePrecomputedAvx512(q.ctypes.data, c.ctypes.data, q.size, t.ctypes.data, r.ctypes.data)
        return r[0]

    def int4_l2_batch(self, query, codes, dim, min_val, diff):
        q = np.ascontiguousarray(query, np.float32); c = np.ascontiguousarray(codes, np.uint8)
        mn = np.ascontiguousarray(min_val, np.float32); df = np.ascontiguousarray(diff, np.float32)
        n = c.size // ((dim + 1) // 2) if dim else 0
        out = np.zeros(n, np.float32)
        if n and dim:
            self.lib.int4L2DistanceBatchAvx512(q.ctypes.data, c.ctypes.data, dim, n, mn.ctypes.data, df.ctypes.data,
                                               out.ctypes.data)
        return out

    def sq8u_l2_batch(self, query, codes, mins, inv_scales, dim):
        q = np.ascontiguousarray(query, np.float32); c = np.ascontiguousarray(codes, np.uint8)
        mn = np.ascontiguousarray(mins, np.float32); iv = np.ascontiguousarray(inv_scales, np.float32)
        n = c.size // dim if dim else 0
        out = np.zeros(n, np.float32)
        if n and dim:
            self.lib.sq8uL2BatchPerDimensionAvx512(q.ctypes.data, c.ctypes.data, mn.ctypes.data, iv.ctypes.data,
                                                   dim, n, out.ctypes.data)
        return out

    def hamming(self, a, b):
        a = np.ascontiguousarray(a, np.uint8); b = np.ascontiguousarray(b, np.uint8)
        if a.size == 0:
            return 0
        return int(self.lib.hammingAvx512(a.ctypes.data, b.ctypes.data, a.size))


# ---- timed CPU baseline (oracle/vg_cpu_bench.c): C threads, one query per thread ------------------------
BENCH_FLAT, BENCH_HNSW, BENCH_ADC, BENCH_RABITQ, BENCH_VAMANA, BENCH_SQ8, BENCH_HNSW_PQ_RERANK = 0, 1, 2, 3, 4, 5, 6


class KernelHooks(C.Structure):
    _fields_ = [("l2", C.c_void_p), ("dot", C.c_void_p), ("l2_bounded", C.c_void_p), ("l2_batch", C.c_void_p),
                ("adc", C.c_void_p), ("hamming", C.c_void_p)]


class BenchJob(C.Structure):
    _fields_ = [("kind", C.c_int32), ("base", _f32p), ("n", C.c_int64), ("dim", C.c_int32), ("metric", C.c_int32),
                ("hnsw", C.POINTER(HnswGraph)), ("vamana", C.POINTER(Vamana)), ("pq", C.POINTER(PQ)),
                ("codes", _u8p), ("queries", _f32p), ("nq", C.c_int64), ("k", C.c_int32), ("ef", C.c_int32),
                ("ids", _u32p), ("dist_comps", _i64p), ("scores", _f32p), ("sq_mins", _f32p), ("sq_inv_scales", _f32p)]


_sig("vgo_set_kernel_hooks", None, C.POINTER(KernelHooks))
_sig("vgo_bench_interleaved_copy", C.c_void_p, C.c_void_p, C.c_size_t, _i32p)
_sig("vgo_bench_free", None, C.c_void_p, C.c_size_t)
_sig("vgo_bench_run", C.c_int64, C.POINTER(BenchJob), C.c_int32, C.c_double, C.POINTER(C.c_double))


def use_reference_kernels(on=True) -> bool:
    """Route the oracle's L2 / Dot / bounded / ADC / Hamming calls through the reference's compiled AVX-512
    kernels (oracle/_ref) — for the timed CPU baseline.  Returns False (and leaves the scalar port active) when
    _ref is absent or the host has no AVX-512."""
    if not on:
        lib.vgo_set_kernel_hooks(None)
        return False
    r = load_ref()
    if r is None:
        lib.vgo_set_kernel_hooks(None)
        return False
    addr = lambda name: C.cast(getattr(r, name), C.c_void_p).value
    h = KernelHooks(addr("squaredL2Avx512"), addr("dotProductAvx512"), addr("squaredL2BoundedAvx512"),
                    addr("squaredL2BatchAvx512"), addr("pqAdcLookupAvx512"), addr("hammingAvx512"))
    lib.vgo_set_kernel_hooks(C.byref(h))
    return True


class InterleavedCopy:
    """A page-interleaved (MPOL_INTERLEAVE over the online NUMA nodes) copy of a host array."""

    def __init__(self, arr):
        a = np.ascontiguousarray(arr)
        il = C.c_int32(0)
        self.nbytes = a.nbytes
        self.ptr = lib.vgo_bench_interleaved_copy(a.ctypes.data, a.nbytes, C.byref(il))
        if not self.ptr:
            raise MemoryError("vgo_bench_interleaved_copy")
        self.numa_nodes = int(il.value)
        self.array = np.ctypeslib.as_array(C.cast(self.ptr, C.POINTER(C.c_uint8)), (a.nbytes,)).view(a.dtype).reshape(a.shape)

    def close(self):
        if self.ptr:
            self.array = None
            lib.vgo_bench_free(self.ptr, self.nbytes)
            self.ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def bench_run(kind, queries, k, threads, budget_s, base=None, dim=None, metric=METRIC_L2, hnsw: "HnswIndex" = None,
              ef=0, pq: "ProductQuantizer" = None, codes=None, n=None, vamana: "VamanaIndex" = None, want_ids=False,
              sq_mins=None, sq_inv_scales=None):
    """One query per C thread for ~budget_s seconds (budget 0: every thread runs exactly one query).  Returns
    dict(queries, seconds, qps, ids, scores, dist_comps); ids / scores / dist_comps cover the first pass over the
    queries (rows a thread did not reach keep 0xFFFFFFFF / -1)."""
    q = np.ascontiguousarray(queries, np.float32)
    nq, d = q.shape
    job = BenchJob()
    job.kind = kind
    job.dim = d if dim is None else dim
    job.metric = metric
    job.queries = q.ctypes.data_as(_f32p)
    job.nq, job.k, job.ef = nq, k, ef
    keep = [q]
    if base is not None:
        job.base = base.ctypes.data_as(_f32p)
        job.n = base.shape[0]
    if n is not None:
        job.n = n
    if hnsw is not None:
        g = hnsw._c(); keep.append(g)
        job.hnsw = C.pointer(g)
    if pq is not None:
        pc = pq._c(); keep.append(pc)
        job.pq = C.pointer(pc)
    if codes is not None:
        cd = np.ascontiguousarray(codes, np.uint8); keep.append(cd)
        job.codes = cd.ctypes.data_as(_u8p)
    if vamana is not None:
        vc = vamana._c(); keep.append(vc)
        job.vamana = C.pointer(vc)
    if sq_mins is not None:
        mn = np.ascontiguousarray(sq_mins, np.float32); iv = np.ascontiguousarray(sq_inv_scales, np.float32)
        keep += [mn, iv]
        job.sq_mins, job.sq_inv_scales = mn.ctypes.data_as(_f32p), iv.ctypes.data_as(_f32p)
    ids = np.full((nq, k), 0xFFFFFFFF, np.uint32) if want_ids else None
    scs = np.zeros((nq, k), np.float32) if want_ids else None
    dc = np.full(nq, -1, np.int64) if want_ids else None
    if want_ids:
        job.ids = ids.ctypes.data_as(_u32p)
        job.scores = scs.ctypes.data_as(_f32p)
        job.dist_comps = dc.ctypes.data_as(_i64p)
    secs = C.c_double(0)
    done = lib.vgo_bench_run(C.byref(job), threads, budget_s, C.byref(secs))
    return dict(queries=int(done), seconds=float(secs.value), qps=done / max(secs.value, 1e-9), ids=ids, scores=scs,
                dist_comps=dc)


# ---- build-side twins (oracle/vg_cpu_bench.c "build-side twins"): one unit per C thread ----------------------
BUILD_KM_ASSIGN, BUILD_PQ_ENCODE, BUILD_PQ_LUT, BUILD_RERANK, BUILD_BRUTE, BUILD_PQ_TRAIN_SUB = 0, 1, 2, 3, 4, 5


class BuildJob(C.Structure):
    _fields_ = [("kind", C.c_int32), ("units", _f32p), ("n_units", C.c_int64), ("dim", C.c_int32), ("metric", C.c_int32),
                ("centroids", _f32p), ("k", C.c_int32), ("pq", C.POINTER(PQ)), ("base", _f32p), ("cand", _u32p),
                ("nc", C.c_int32), ("topk", C.c_int32), ("hnsw", C.POINTER(HnswGraph)), ("mode", C.c_int32),
                ("train_n", C.c_int64), ("pq_m", C.c_int32), ("pq_k", C.c_int32), ("iters", C.c_int32),
                ("seed", C.c_uint64), ("out_assign", _i32p), ("out_codes", _u8p), ("out_ids", _u32p),
                ("out_scores", _f32p), ("out_cent", _f32p)]


_sig("vgo_bench_build_run", C.c_int64, C.POINTER(BuildJob), C.c_int32, C.c_double, C.POINTER(C.c_double))


def bench_build_run(kind, units, threads, budget_s, metric=METRIC_L2, centroids=None, pq: "ProductQuantizer" = None,
                    base=None, cand=None, topk=10, hnsw: "HnswIndex" = None, mode=0, pq_m=0, pq_k=0, iters=0, seed=0,
                    want_out=False):
    """One unit per C thread for ~budget_s seconds (0: every thread runs exactly one unit).  `units` = rows
    (KM_ASSIGN, PQ_ENCODE), queries (PQ_LUT, RERANK, BRUTE) or the training rows (PQ_TRAIN_SUB: units are the pq_m
    sub-quantizers).  Returns dict(units, seconds, rate, out) — `out` holds the first pass's results when want_out."""
    u = np.ascontiguousarray(units, np.float32)
    job = BuildJob()
    job.kind = kind
    job.units = u.ctypes.data_as(_f32p)
    job.n_units = pq_m if kind == BUILD_PQ_TRAIN_SUB else u.shape[0]
    job.dim = u.shape[1]
    job.metric = metric
    job.topk, job.mode = topk, mode
    keep = [u]
    out = {}
    if centroids is not None:
        c = np.ascontiguousarray(centroids, np.float32); keep.append(c)
        job.centroids = c.ctypes.data_as(_f32p)
        job.k = c.shape[0]
    if pq is not None:
        pc = pq._c(); keep.append(pc)
        job.pq = C.pointer(pc)
    if base is not None:
        assert base.dtype == np.float32 and base.flags.c_contiguous
        job.base = base.ctypes.data_as(_f32p)
    if cand is not None:
        cd = np.ascontiguousarray(cand, np.uint32); keep.append(cd)
        job.cand = cd.ctypes.data_as(_u32p)
        job.nc = cd.shape[1]
    if hnsw is not None:
        g = hnsw._c(); keep.append(g)
        job.hnsw = C.pointer(g)
    if kind == BUILD_PQ_TRAIN_SUB:
        job.train_n, job.pq_m, job.pq_k, job.iters, job.seed = u.shape[0], pq_m, pq_k, iters, seed
    if want_out:
        if kind == BUILD_KM_ASSIGN:
            out["assign"] = np.full(u.shape[0], -1, np.int32)
            job.out_assign = out["assign"].ctypes.data_as(_i32p)
        elif kind == BUILD_PQ_ENCODE:
            out["codes"] = np.zeros((u.shape[0], pq.m), np.uint8)
            job.out_codes = out["codes"].ctypes.data_as(_u8p)
        elif kind in (BUILD_RERANK, BUILD_BRUTE):
            out["ids"] = np.full((u.shape[0], topk), 0xFFFFFFFF, np.uint32)
            out["scores"] = np.zeros((u.shape[0], topk), np.float32)
            job.out_ids = out["ids"].ctypes.data_as(_u32p)
            job.out_scores = out["scores"].ctypes.data_as(_f32p)
        elif kind == BUILD_PQ_TRAIN_SUB:
            out["cent"] = np.zeros((pq_m, pq_k, u.shape[1] // pq_m), np.float32)
            job.out_cent = out["cent"].ctypes.data_as(_f32p)
    secs = C.c_double(0)
    done = lib.vgo_bench_build_run(C.byref(job), threads, budget_s, C.byref(secs))
    return dict(units=int(done), seconds=float(secs.value), rate=done / max(secs.value, 1e-9), out=out)


def assign_partition_batch(vectors, centroids, metric=METRIC_L2, threads=8):
    """kmeans.AssignPartition for every row (vgo_bench_build_run, one pass, `threads` C threads): int32[n]."""
    r = bench_build_run(BUILD_KM_ASSIGN, vectors, threads, -1.0, metric=metric, centroids=centroids, want_out=True)
    assert r["units"] == np.asarray(vectors).shape[0]
    return r["out"]["assign"]


def replay(kind, queries, k, **kw):
    """Whole queries through the oracle's loops, one C thread per query, the reference's compiled kernels when
    oracle/_ref is present (bit-identical to the scalar restatement, tests/test_oracle_golden.py): what the
    full-size parity tests use to replay a few queries over 1M..10M rows in seconds.  Returns (ids, scores)."""
    q = np.ascontiguousarray(queries, np.float32)
    use_reference_kernels(True)
    try:
        r = bench_run(kind, q, k, q.shape[0], 0.0, want_ids=True, **kw)
    finally:
        use_reference_kernels(False)
    assert (r["dist_comps"] >= 0).all()
    return r["ids"], r["scores"]
