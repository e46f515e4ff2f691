"""Python mirror of the reference interfaces over the C ABI (tests + bench driver).

Arrays may be numpy (host) or torch CUDA tensors (HBM); the library detects which.
"""
from __future__ import annotations

import ctypes as C
import enum

import numpy as np

from . import _lib
from ._lib import VecgoHipError, check

try:  # torch is plumbing only: device buffers and streams
    import torch
except Exception:  # pragma: no cover
    torch = None


class Metric(enum.IntEnum):
    """distance.Metric (distance/distance.go:66-73)."""
    L2 = 0
    COSINE = 1
    DOT = 2
    HAMMING = 3


def _is_torch(x) -> bool:
    return torch is not None and isinstance(x, torch.Tensor)


def _ptr(x, dtype, count=None):
    """(keepalive, void*) of a contiguous numpy array or torch tensor of `dtype`."""
    if x is None:
        return None, None
    if _is_torch(x):
        tdt = {np.float32: torch.float32, np.uint8: torch.uint8, np.int8: torch.int8,
               np.uint32: torch.int32, np.int32: torch.int32, np.uint64: torch.int64,
               np.int64: torch.int64}[dtype]
        if x.dtype != tdt:
            raise TypeError(f"expected torch dtype {tdt}, got {x.dtype}")
        if not x.is_contiguous():
            x = x.contiguous()
        if count is not None and x.numel() < count:
            raise ValueError(f"buffer has {x.numel()} elements, need {count}")
        return x, C.c_void_p(x.data_ptr())
    a = np.ascontiguousarray(x, dtype=dtype)
    if count is not None and a.size < count:
        raise ValueError(f"buffer has {a.size} elements, need {count}")
    return a, C.c_void_p(a.ctypes.data)


def _stream_ptr(stream):
    if stream is None:
        # Device tensors handed to the library are produced on torch's current stream; run on
        # it by default so that producer and consumer are ordered.  (The C ABI's own default,
        # NULL = the context's private stream, is for host-buffer callers such as cgo.)
        if torch is not None and torch.cuda.is_available() and torch.cuda.is_initialized():
            return _handle(torch.cuda.current_stream().cuda_stream)
        return None
    if torch is not None and isinstance(stream, torch.cuda.Stream):
        return _handle(stream.cuda_stream)
    return _handle(int(stream))


def _handle(h: int):
    # torch's default stream is HIP's legacy default stream, handle 0; the C ABI reserves NULL
    # for "the context's own stream", so the legacy stream travels as VG_STREAM_LEGACY (= 1)
    return C.c_void_p(h if h != 0 else 1)


class Context:
    """One per (process, GPU)."""

    def __init__(self, device: int = 0):
        self._lib = _lib.load()
        h = C.c_void_p()
        check(self._lib.vg_ctx_create(C.c_int32(device), C.byref(h)))
        self._h = h
        self.device = device

    def close(self):
        if getattr(self, "_h", None):
            self._lib.vg_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def synchronize(self, stream=None):
        check(self._lib.vg_ctx_synchronize(self._h, _stream_ptr(stream)))

    def profile_enable(self, on: bool = True):
        check(self._lib.vg_profile_enable(self._h, C.c_int32(1 if on else 0)))

    def profile_read(self, kernel: str):
        """(launches, total_ms) of `kernel` since the last read (HIP events on the launch stream)."""
        n = C.c_int64(); ms = C.c_double()
        check(self._lib.vg_profile_read(self._h, kernel.encode(), C.byref(n), C.byref(ms)))
        return n.value, ms.value

    def device_info(self):
        arch = C.create_string_buffer(64)
        cus = C.c_int32()
        hbm = C.c_int64()
        check(self._lib.vg_ctx_device_info(self._h, arch, 64, C.byref(cus), C.byref(hbm)))
        return {"arch": arch.value.decode(), "compute_units": cus.value, "hbm_bytes": hbm.value}


def squared_l2_batch(ctx: Context, query, targets, dim: int, out=None, stream=None):
    """simd.SquaredL2Batch (internal/simd/kernels.go:66-68)."""
    return _batch(ctx, ctx._lib.vg_squared_l2_batch, query, targets, dim, out, stream)


def dot_batch(ctx: Context, query, targets, dim: int, out=None, stream=None):
    """simd.DotBatch (internal/simd/kernels.go:61-63)."""
    return _batch(ctx, ctx._lib.vg_dot_batch, query, targets, dim, out, stream)


def _batch(ctx, fn, query, targets, dim, out, stream):
    n = (targets.numel() if _is_torch(targets) else np.asarray(targets).size) // dim if dim > 0 else 0
    q, pq_ = _ptr(query, np.float32)
    t, pt = _ptr(targets, np.float32)
    if out is None:
        out = _empty_like(targets, (n,), np.float32)
    o, po = _ptr(out, np.float32, n)
    check(fn(ctx._h, pq_, pt, C.c_int64(dim), C.c_int64(n), po, _stream_ptr(stream)))
    return out


def merge_topk(ctx: Context, ids_in, scores_in, k: int, metric=0, id_offsets=None, out=None,
               stream=None):
    """engine fan-in (engine/search.go:904-908): ids_in/scores_in are [lists, nq, k]."""
    shape = tuple(ids_in.shape)
    lists, nq = shape[0], shape[1]
    assert shape[2] == k
    i, pi = _ptr(ids_in, np.uint32)
    s, ps = _ptr(scores_in, np.float32)
    o, po = _ptr(id_offsets, np.uint32, lists) if id_offsets is not None else (None, None)
    if out is None:
        out = (_empty_like(ids_in, (nq, k), np.uint32), _empty_like(ids_in, (nq, k), np.float32))
    oi, poi = _ptr(out[0], np.uint32, nq * k)
    os_, pos = _ptr(out[1], np.float32, nq * k)
    check(ctx._lib.vg_merge_topk(ctx._h, pi, ps, C.c_int32(lists), C.c_int64(nq), C.c_int32(k),
                                 C.c_int32(int(metric)), po, poi, pos, _stream_ptr(stream)))
    return out


class OptimizedProductQuantizer:
    """quantization.OptimizedProductQuantizer (opq.go): block-diagonal rotation + ProductQuantizer."""

    def __init__(self, ctx: Context, dimension: int, num_subvectors: int, num_centroids: int = 256, num_iterations: int = 2):
        self._lib = ctx._lib
        self._lib.vg_opq_pq.restype = C.c_void_p
        self.ctx, self.dimension, self.num_subvectors, self.num_centroids = ctx, dimension, num_subvectors, num_centroids
        h = C.c_void_p()
        check(self._lib.vg_opq_create(ctx._h, C.c_int32(dimension), C.c_int32(num_subvectors), C.c_int32(num_centroids),
                                      C.c_int32(num_iterations), C.byref(h)))
        self._h = h
        self.pq = ProductQuantizer._borrowed(ctx, C.c_void_p(self._lib.vg_opq_pq(h)), dimension, num_subvectors, num_centroids, self)
        b, nb = C.c_int32(), C.c_int32()
        check(self._lib.vg_opq_get_rotations(h, C.byref(b), C.byref(nb), None))
        self.block, self.nblocks = b.value, nb.value

    def close(self):
        if getattr(self, "_h", None):
            self._lib.vg_opq_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def is_trained(self) -> bool:
        return bool(self._lib.vg_opq_is_trained(self._h))

    def rotations(self):
        r = np.empty((self.nblocks, self.block, self.block), np.float32)
        check(self._lib.vg_opq_get_rotations(self._h, None, None, C.c_void_p(r.ctypes.data)))
        return r

    def set_rotations(self, rotations):
        r = np.ascontiguousarray(rotations, np.float32)
        assert r.size == self.nblocks * self.block * self.block
        check(self._lib.vg_opq_set_rotations(self._h, C.c_void_p(r.ctypes.data)))

    def train(self, vectors, pq_iters: int = 20, seed: int = 1, stream=None):
        n = _rows(vectors, self.dimension)
        v, pv = _ptr(vectors, np.float32)
        check(self._lib.vg_opq_train(self._h, pv, C.c_int64(n), C.c_int32(pq_iters), C.c_uint64(seed), _stream_ptr(stream)))

    def rotate(self, vectors, out=None, stream=None):
        n = _rows(vectors, self.dimension)
        v, pv = _ptr(vectors, np.float32)
        if out is None:
            out = _empty_like(vectors, (n, self.dimension), np.float32)
        o, po = _ptr(out, np.float32, n * self.dimension)
        check(self._lib.vg_opq_rotate(self._h, pv, C.c_int64(n), po, _stream_ptr(stream)))
        return out

    def encode(self, vectors, out=None, stream=None):
        n = _rows(vectors, self.dimension)
        v, pv = _ptr(vectors, np.float32)
        if out is None:
            out = _empty_like(vectors, (n, self.num_subvectors), np.uint8)
        o, po = _ptr(out, np.uint8, n * self.num_subvectors)
        check(self._lib.vg_opq_encode(self._h, pv, C.c_int64(n), po, _stream_ptr(stream)))
        return out

    def decode(self, codes, out=None, stream=None):
        c, pc = _ptr(codes, np.uint8)
        n = (c.numel() if _is_torch(c) else c.size) // self.num_subvectors
        if out is None:
            out = _empty_like(codes, (n, self.dimension), np.float32)
        o, po = _ptr(out, np.float32, n * self.dimension)
        check(self._lib.vg_opq_decode(self._h, pc, C.c_int64(n), po, _stream_ptr(stream)))
        return out

    def asymmetric_distance(self, query, codes, out=None, stream=None):
        c, pc = _ptr(codes, np.uint8)
        n = (c.numel() if _is_torch(c) else c.size) // self.num_subvectors
        q, pq_ = _ptr(query, np.float32, self.dimension)
        if out is None:
            out = _empty_like(codes, (n,), np.float32)
        o, po = _ptr(out, np.float32, n)
        check(self._lib.vg_opq_asymmetric_distance_batch(self._h, pq_, pc, C.c_int64(n), po, _stream_ptr(stream)))
        return out


class BinaryQuantizer:
    """quantization.BinaryQuantizer (binary.go:23-262): 1 bit per dimension against a threshold."""

    def __init__(self, ctx: Context, dimension: int):
        self._lib = ctx._lib
        self._lib.vg_binary_code_bytes.restype = C.c_int64
        self.ctx, self.dimension = ctx, dimension
        self.threshold, self.trained = np.float32(0.0), False

    def with_threshold(self, threshold: float):
        self.threshold, self.trained = np.float32(threshold), True
        return self

    @property
    def words(self) -> int:
        return (self.dimension + 63) // 64

    def bytes_total(self) -> int:            # BytesTotal (binary.go:198-200)
        return (self.dimension + 7) // 8

    def train(self, vectors, stream=None):
        n = _rows(vectors, self.dimension)
        if n == 0:
            raise ValueError("no vectors provided for training")
        v, pv = _ptr(vectors, np.float32)
        th = np.zeros(1, np.float32)
        check(self._lib.vg_binary_train(self.ctx._h, C.c_int32(self.dimension), pv, C.c_int64(n),
                                        C.c_void_p(th.ctypes.data), _stream_ptr(stream)))
        self.threshold, self.trained = th[0], True

    def encode(self, vectors, out=None, stream=None):
        """[n, dim] float32 -> [n, words*8] uint8 (the uint64 words of EncodeUint64Into, little endian)."""
        n = _rows(vectors, self.dimension)
        v, pv = _ptr(vectors, np.float32)
        if out is None:
            out = _empty_like(vectors, (n, self.words * 8), np.uint8)
        o, po = _ptr(out, np.uint8, n * self.words * 8)
        check(self._lib.vg_binary_encode(self.ctx._h, C.c_int32(self.dimension), C.c_float(float(self.threshold)), pv,
                                         C.c_int64(n), po, _stream_ptr(stream)))
        return out

    def decode(self, codes, out=None, stream=None):
        c, pc = _ptr(codes, np.uint8)
        total = c.numel() if _is_torch(c) else c.size
        cb = c.shape[-1] if len(c.shape) > 1 else total
        n = total // max(cb, 1)
        if out is None:
            out = _empty_like(codes, (n, self.dimension), np.float32)
        o, po = _ptr(out, np.float32, n * self.dimension)
        check(self._lib.vg_binary_decode(self.ctx._h, C.c_int32(self.dimension), C.c_float(float(self.threshold)), pc,
                                         C.c_int64(n), C.c_int32(cb), po, _stream_ptr(stream)))
        return out

    def compute_hamming_distance(self, query, codes, out=None, stream=None):
        """ComputeHammingDistance (binary.go:158-171) of one float query against [n, words*8] codes."""
        c, pc = _ptr(codes, np.uint8)
        n = (c.numel() if _is_torch(c) else c.size) // (self.words * 8)
        q, pq_ = _ptr(query, np.float32, self.dimension)
        if out is None:
            out = _empty_like(codes, (n,), np.int32)
        o, po = _ptr(out, np.int32, n)
        check(self._lib.vg_binary_hamming_batch(self.ctx._h, C.c_int32(self.dimension), C.c_float(float(self.threshold)),
                                                pq_, pc, C.c_int64(n), po, _stream_ptr(stream)))
        return out


def normalize_l2(ctx: Context, vectors, dim: int, stream=None):
    """distance.NormalizeL2InPlace over the rows of `vectors` (in place); returns ok[n] uint8."""
    n = _rows(vectors, dim)
    v, pv = _ptr(vectors, np.float32)
    ok = _empty_like(vectors, (n,), np.uint8)
    o, po = _ptr(ok, np.uint8, n)
    check(ctx._lib.vg_normalize_l2(ctx._h, pv, C.c_int64(n), C.c_int32(dim), po, _stream_ptr(stream)))
    return ok


class debug_hook:
    """with vecgo_amd.api.debug_hook("VG_FLAT_FORCE_EXACT"): ... — a test hook switched on for the block
    (vg_debug_set_hook)."""

    def __init__(self, *names):
        self.names = names

    def __enter__(self):
        lib = _lib.load()
        for n in self.names:
            check(lib.vg_debug_set_hook(n.encode(), 1))
        return self

    def __exit__(self, *exc):
        lib = _lib.load()
        for n in self.names:
            lib.vg_debug_set_hook(n.encode(), 0)
        return False


class Comm:
    """vg_comm: the exchange step of a row-sharded search through the C ABI (direct ncclAllGather on the
    caller's stream + merge).  Rank 0 makes the id (Comm.unique_id()), the host distributes the bytes."""

    ID_BYTES = 128

    @staticmethod
    def unique_id() -> bytes:
        lib = _lib.load()
        buf = (C.c_uint8 * Comm.ID_BYTES)()
        check(lib.vg_comm_unique_id(buf))
        return bytes(buf)

    @staticmethod
    def probe() -> str:
        """Load RCCL (not a collective) and return the file it came from; raises when it cannot be loaded.  Every
        rank calls this and the ranks agree on the outcome BEFORE any of them enters Comm(), which is a collective."""
        lib = _lib.load()
        buf = C.create_string_buffer(1024)
        check(lib.vg_comm_probe(buf, C.c_int32(1024)))
        return buf.value.decode()

    def describe(self) -> dict:
        """What RCCL says about this communicator (ncclCommCount / UserRank / CuDevice) and which librccl is in use."""
        n, r, d, reused = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        buf = C.create_string_buffer(1024)
        check(self._lib.vg_comm_describe(self._h, C.byref(n), C.byref(r), C.byref(d), C.byref(reused), buf, C.c_int32(1024)))
        return {"rccl_ranks": n.value, "rccl_rank": r.value, "rccl_device": d.value,
                "reused_mapped_rccl": bool(reused.value), "rccl_path": buf.value.decode()}

    def __init__(self, ctx: Context, world: int, rank: int, unique_id: bytes):
        self._lib = ctx._lib
        self.ctx, self.world, self.rank = ctx, world, rank
        h = C.c_void_p()
        buf = (C.c_uint8 * Comm.ID_BYTES).from_buffer_copy(unique_id)
        check(self._lib.vg_comm_create(ctx._h, C.c_int32(world), C.c_int32(rank), buf, C.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._lib.vg_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def all_gather_topk(self, local_ids, local_scores, k: int, metric=0, id_offsets=None, out=None, stream=None):
        """local_ids/local_scores [nq, k] (device) -> merged global (ids, scores) [nq, k]."""
        nq = local_ids.shape[0]
        i, pi = _ptr(local_ids, np.uint32, nq * k)
        s_, ps = _ptr(local_scores, np.float32, nq * k)
        o, po = _ptr(id_offsets, np.uint32, self.world) if id_offsets is not None else (None, None)
        if out is None:
            out = (_empty_like(local_ids, (nq, k), np.uint32), _empty_like(local_ids, (nq, k), np.float32))
        oi, poi = _ptr(out[0], np.uint32, nq * k)
        os_, pos = _ptr(out[1], np.float32, nq * k)
        check(self._lib.vg_comm_all_gather_topk(self._h, pi, ps, C.c_int64(nq), C.c_int32(k), C.c_int32(int(metric)),
                                                po, poi, pos, _stream_ptr(stream)))
        return out

    def all_gather_bytes(self, send, recv, stream=None):
        """send: uint8 device tensor of this rank; recv: uint8 device tensor, world times as long."""
        a, pa = _ptr(send, np.uint8)
        b, pb = _ptr(recv, np.uint8, a.numel() * self.world)
        check(self._lib.vg_comm_all_gather(self._h, pa, pb, C.c_int64(a.numel()), _stream_ptr(stream)))
        return recv


def heap_replay(ctx: Context, is_max: bool, ops, unsigned_keys: bool = False, cap: int = 4096, stream=None):
    """vg_debug_heap_replay: ops int32[n, 4] = {op (VG_HEAP_*), node, float32 bits, arg}; returns (out int32[n, 3]
    {flag, node, bits}, nodes uint32[len], dists float32[len]) — the device heap of csrc/vg_heap.hpp after the script."""
    ops = np.ascontiguousarray(ops, np.int32).reshape(-1, 4)
    out = np.zeros((ops.shape[0], 3), np.int32)
    items = np.zeros(max(cap, 1), np.uint64)
    flen = np.zeros(1, np.int32)
    check(ctx._lib.vg_debug_heap_replay(ctx._h, C.c_int32(int(bool(is_max))), C.c_int32(int(bool(unsigned_keys))),
                                        C.c_void_p(ops.ctypes.data), C.c_int32(ops.shape[0]), C.c_void_p(out.ctypes.data),
                                        C.c_void_p(flen.ctypes.data), C.c_void_p(items.ctypes.data), C.c_int32(cap),
                                        _stream_ptr(stream)))
    items = items[:int(flen[0])]
    return out, (items & 0xFFFFFFFF).astype(np.uint32), (items >> 32).astype(np.uint32).view(np.float32)


def merge_topk_packed(ctx: Context, packed, lists: int, nq: int, k: int, metric=0, id_offsets=None, out=None, stream=None):
    """vg_merge_topk_packed: packed is a device int32/uint32 tensor [lists, 2, nq, k]."""
    p_, pp = _ptr(packed, np.uint32, lists * 2 * nq * k)
    o, po = _ptr(id_offsets, np.uint32, lists) if id_offsets is not None else (None, None)
    if out is None:
        out = (_empty_like(packed, (nq, k), np.uint32), _empty_like(packed, (nq, k), np.float32))
    oi, poi = _ptr(out[0], np.uint32, nq * k)
    os_, pos = _ptr(out[1], np.float32, nq * k)
    check(ctx._lib.vg_merge_topk_packed(ctx._h, pp, C.c_int32(lists), C.c_int64(nq), C.c_int32(k), C.c_int32(int(metric)),
                                        po, poi, pos, _stream_ptr(stream)))
    return out


def squared_l2_bounded_batch(ctx: Context, query, targets, dim: int, bounds, stream=None):
    """simd.SquaredL2Bounded (kernels.go:173), one query vs n targets: (dist[n], exceeded[n])."""
    n = (targets.numel() if _is_torch(targets) else np.asarray(targets).size) // dim if dim > 0 else 0
    q, pq_ = _ptr(query, np.float32)
    t, pt = _ptr(targets, np.float32)
    b = np.atleast_1d(np.asarray(bounds, np.float32)) if not _is_torch(bounds) else bounds
    b_, pb = _ptr(b, np.float32)
    nb = b_.numel() if _is_torch(b_) else b_.size
    dist = _empty_like(targets, (n,), np.float32)
    exc = _empty_like(targets, (n,), np.int32)
    d, pd = _ptr(dist, np.float32, n)
    e, pe = _ptr(exc, np.int32, n)
    check(ctx._lib.vg_squared_l2_bounded_batch(ctx._h, pq_, pt, C.c_int64(dim), C.c_int64(n), pb,
                                               C.c_int64(nb), pd, pe, _stream_ptr(stream)))
    return dist, exc


def pq_adc_lookup_batch(ctx: Context, table, codes, m: int, stream=None):
    """simd.PqAdcLookup (kernels.go:56), one table vs n codes."""
    n = (codes.numel() if _is_torch(codes) else np.asarray(codes).size) // m if m > 0 else 0
    t, pt = _ptr(table, np.float32, m * 256)
    c, pc = _ptr(codes, np.uint8)
    out = _empty_like(codes, (n,), np.float32)
    o, po = _ptr(out, np.float32, n)
    check(ctx._lib.vg_pq_adc_lookup_batch(ctx._h, pt, pc, C.c_int64(m), C.c_int64(n), po, _stream_ptr(stream)))
    return out


def kmeans_train(ctx: Context, vectors, dim: int, k: int, metric=0, max_iter: int = 10, seed: int = 1, stream=None):
    """kmeans.TrainKMeans (kmeans.go:16-138): returns [k, dim] centroids or None when n < k."""
    n = _rows(vectors, dim)
    v, pv = _ptr(vectors, np.float32)
    out = _empty_like(vectors, (k, dim), np.float32)
    o, po = _ptr(out, np.float32)
    produced = C.c_int32(0)
    check(ctx._lib.vg_kmeans_train(ctx._h, pv, C.c_int64(n), C.c_int32(dim), C.c_int32(k), C.c_int32(int(metric)),
                                   C.c_int32(max_iter), C.c_uint64(seed), po, C.byref(produced),
                                   _stream_ptr(stream)))
    return out if produced.value else None


def kmeans_assign(ctx: Context, vectors, centroids, dim: int, metric=0, stream=None):
    """kmeans.AssignPartition (kmeans.go:142-196), batched."""
    n = _rows(vectors, dim)
    k = _rows(centroids, dim)
    v, pv = _ptr(vectors, np.float32)
    c, pc = _ptr(centroids, np.float32)
    out = _empty_like(vectors, (n,), np.int32)
    o, po = _ptr(out, np.int32, n)
    check(ctx._lib.vg_kmeans_assign(ctx._h, pv, C.c_int64(n), C.c_int32(dim), pc, C.c_int32(k),
                                    C.c_int32(int(metric)), po, _stream_ptr(stream)))
    return out


def find_closest_centroids(ctx: Context, query, centroids, dim: int, nprobe: int, metric=0, stream=None):
    """kmeans.FindClosestCentroids (kmeans.go:217-280)."""
    k = _rows(centroids, dim)
    q = np.ascontiguousarray(query, np.float32)
    c = np.ascontiguousarray(centroids, np.float32)
    out = np.empty(max(min(nprobe, k), 1), np.int32)
    n_out = C.c_int32(0)
    check(ctx._lib.vg_find_closest_centroids(ctx._h, C.c_void_p(q.ctypes.data), C.c_void_p(c.ctypes.data),
                                             C.c_int32(dim), C.c_int32(k), C.c_int32(nprobe),
                                             C.c_int32(int(metric)), C.c_void_p(out.ctypes.data),
                                             C.byref(n_out), _stream_ptr(stream)))
    return out[:n_out.value]


class RaBitQuantizer:
    """quantization.RaBitQuantizer (internal/quantization/rabitq.go:26-49)."""

    def __init__(self, ctx: Context, dimension: int):
        if dimension <= 0:
            raise VecgoHipError(-1, "dimension must be positive")
        self.ctx, self.dimension = ctx, dimension
        self._lib = ctx._lib
        self._lib.vg_rabitq_code_bytes.restype = C.c_int64

    def bytes_total(self) -> int:
        """BytesTotal (rabitq.go:187-190)."""
        return int(self._lib.vg_rabitq_code_bytes(C.c_int32(self.dimension)))

    def encode(self, vectors, out=None, stream=None):
        n = _rows(vectors, self.dimension)
        v, pv = _ptr(vectors, np.float32)
        if out is None:
            out = _empty_like(vectors, (n, self.bytes_total()), np.uint8)
        c, pc = _ptr(out, np.uint8, n * self.bytes_total())
        check(self._lib.vg_rabitq_encode(self.ctx._h, C.c_int32(self.dimension), pv, C.c_int64(n), pc,
                                         _stream_ptr(stream)))
        return out

    def distance(self, query, codes, out=None, stream=None):
        """Distance (rabitq.go:119-176) of one query against n codes."""
        if _rows(query, self.dimension) != 1:
            raise VecgoHipError(-2, "vector dimension mismatch")
        total = codes.numel() if _is_torch(codes) else np.asarray(codes).size
        if total % self.bytes_total():
            raise VecgoHipError(-4, "invalid code length")
        n = total // self.bytes_total()
        q, pq_ = _ptr(query, np.float32)
        c, pc = _ptr(codes, np.uint8)
        if out is None:
            out = _empty_like(codes, (n,), np.float32)
        o, po = _ptr(out, np.float32, n)
        check(self._lib.vg_rabitq_distance_batch(self.ctx._h, C.c_int32(self.dimension), pq_, pc,
                                                 C.c_int64(n), po, _stream_ptr(stream)))
        return out


class ScalarQuantizer:
    """quantization.ScalarQuantizer (internal/quantization/quantizer.go:27-39): 8 bits per
    dimension, per-dimension min / max."""

    def __init__(self, ctx: Context, dimension: int):
        self.ctx, self.dimension = ctx, dimension
        self._lib = ctx._lib
        h = C.c_void_p()
        check(self._lib.vg_sq8_create(ctx._h, C.c_int32(dimension), C.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._lib.vg_sq8_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def is_trained(self) -> bool:
        return bool(self._lib.vg_sq8_is_trained(self._h))

    def bytes_per_dimension(self) -> int:
        return 1

    def compression_ratio(self) -> float:
        return 4.0

    def train(self, vectors, stream=None):
        """Train (quantizer.go:127-180)."""
        n = _rows(vectors, self.dimension)
        v, pv = _ptr(vectors, np.float32)
        check(self._lib.vg_sq8_train(self._h, pv, C.c_int64(n), _stream_ptr(stream)))

    def set_bounds(self, mins, maxs):
        """SetBounds (quantizer.go:52-78)."""
        a, pa = _ptr(mins, np.float32, self.dimension)
        b, pb = _ptr(maxs, np.float32, self.dimension)
        check(self._lib.vg_sq8_set_bounds(self._h, pa, pb))

    def params(self):
        """(mins, maxs, scales, inv_scales) as numpy arrays."""
        out = [np.empty(self.dimension, np.float32) for _ in range(4)]
        check(self._lib.vg_sq8_get_params(self._h, *[C.c_void_p(a.ctypes.data) for a in out]))
        return tuple(out)

    def encode(self, vectors, out=None, stream=None):
        """Encode (quantizer.go:183-222), batched: returns [n, dim] uint8."""
        n = _rows(vectors, self.dimension)
        v, pv = _ptr(vectors, np.float32)
        if out is None:
            out = _empty_like(vectors, (n, self.dimension), np.uint8)
        c, pc = _ptr(out, np.uint8, n * self.dimension)
        check(self._lib.vg_sq8_encode(self._h, pv, C.c_int64(n), pc, _stream_ptr(stream)))
        return out

    def decode(self, codes, out=None, stream=None):
        """Decode (quantizer.go:225-250), batched: returns [n, dim] float32."""
        total = codes.numel() if _is_torch(codes) else np.asarray(codes).size
        if total % self.dimension:
            raise VecgoHipError(-2, "vector dimension mismatch")
        n = total // self.dimension
        c, pc = _ptr(codes, np.uint8)
        if out is None:
            out = _empty_like(codes, (n, self.dimension), np.float32)
        o, po = _ptr(out, np.float32, n * self.dimension)
        check(self._lib.vg_sq8_decode(self._h, pc, C.c_int64(n), po, _stream_ptr(stream)))
        return out

    def l2_distance_batch(self, query, codes, out=None, stream=None):
        """L2DistanceBatch (quantizer.go:93-106): one query against n codes."""
        if _rows(query, self.dimension) != 1:
            raise VecgoHipError(-2, "query dimension mismatch")
        total = codes.numel() if _is_torch(codes) else np.asarray(codes).size
        n = total // self.dimension
        q, pq_ = _ptr(query, np.float32)
        c, pc = _ptr(codes, np.uint8)
        if out is None:
            out = _empty_like(codes, (n,), np.float32)
        o, po = _ptr(out, np.float32, n)
        check(self._lib.vg_sq8_l2_distance_batch(self._h, pq_, pc, C.c_int64(n), po, _stream_ptr(stream)))
        return out


class Int4Quantizer:
    """quantization.Int4Quantizer (internal/quantization/int4.go:12-20): 4 bits per dimension."""

    def __init__(self, ctx: Context, dimension: int):
        self.ctx, self.dimension = ctx, dimension
        self._lib = ctx._lib
        self._lib.vg_int4_code_bytes.restype = C.c_int64
        h = C.c_void_p()
        check(self._lib.vg_int4_create(ctx._h, C.c_int32(dimension), C.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._lib.vg_int4_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def is_trained(self) -> bool:
        return bool(self._lib.vg_int4_is_trained(self._h))

    @property
    def code_bytes(self) -> int:
        return (self.dimension + 1) // 2

    def bytes_per_dimension(self) -> int:
        return 0  # sub-byte (int4.go:167-169)

    def train(self, vectors, stream=None):
        """Train (int4.go:29-62)."""
        n = _rows(vectors, self.dimension)
        v, pv = _ptr(vectors, np.float32)
        check(self._lib.vg_int4_train(self._h, pv, C.c_int64(n), _stream_ptr(stream)))

    def set_params(self, min_val, diff):
        """UnmarshalBinary (int4.go:190-219)."""
        a, pa = _ptr(min_val, np.float32, self.dimension)
        b, pb = _ptr(diff, np.float32, self.dimension)
        check(self._lib.vg_int4_set_params(self._h, pa, pb))

    def params(self):
        """(min, diff, lookup table[dim*16]) as numpy arrays."""
        mn, df = np.empty(self.dimension, np.float32), np.empty(self.dimension, np.float32)
        tb = np.empty(self.dimension * 16, np.float32)
        check(self._lib.vg_int4_get_params(self._h, C.c_void_p(mn.ctypes.data), C.c_void_p(df.ctypes.data),
                                           C.c_void_p(tb.ctypes.data)))
        return mn, df, tb

    def encode(self, vectors, out=None, stream=None):
        """Encode (int4.go:65-105), batched: returns [n, ceil(dim/2)] uint8."""
        n = _rows(vectors, self.dimension)
        v, pv = _ptr(vectors, np.float32)
        if out is None:
            out = _empty_like(vectors, (n, self.code_bytes), np.uint8)
        c, pc = _ptr(out, np.uint8, n * self.code_bytes)
        check(self._lib.vg_int4_encode(self._h, pv, C.c_int64(n), pc, _stream_ptr(stream)))
        return out

    def decode(self, codes, out=None, stream=None):
        """Decode (int4.go:108-130), batched: returns [n, dim] float32."""
        total = codes.numel() if _is_torch(codes) else np.asarray(codes).size
        if total % self.code_bytes:
            raise VecgoHipError(-2, "dimension mismatch")
        n = total // self.code_bytes
        c, pc = _ptr(codes, np.uint8)
        if out is None:
            out = _empty_like(codes, (n, self.dimension), np.float32)
        o, po = _ptr(out, np.float32, n * self.dimension)
        check(self._lib.vg_int4_decode(self._h, pc, C.c_int64(n), po, _stream_ptr(stream)))
        return out

    def _dist(self, query, codes, precomputed, out, stream):
        if _rows(query, self.dimension) != 1:
            raise VecgoHipError(-2, "dimension mismatch")
        total = codes.numel() if _is_torch(codes) else np.asarray(codes).size
        n = total // self.code_bytes
        q, pq_ = _ptr(query, np.float32)
        c, pc = _ptr(codes, np.uint8)
        if out is None:
            out = _empty_like(codes, (n,), np.float32)
        o, po = _ptr(out, np.float32, n)
        check(self._lib.vg_int4_l2_distance_batch(self._h, pq_, pc, C.c_int64(n), C.c_int32(precomputed), po,
                                                  _stream_ptr(stream)))
        return out

    def l2_distance_batch(self, query, codes, out=None, stream=None):
        """L2DistanceBatch (int4.go:150-164)."""
        return self._dist(query, codes, 0, out, stream)

    def l2_distance(self, query, codes, out=None, stream=None):
        """L2Distance (int4.go:133-147) of one query against each of n codes."""
        return self._dist(query, codes, 1, out, stream)


def hamming_batch(ctx: Context, a, codes, out=None, stream=None):
    """simd.Hamming (kernels.go:71) of one byte string against n contiguous ones."""
    a_, pa = _ptr(a, np.uint8)
    nbytes = a_.numel() if _is_torch(a_) else a_.size
    total = codes.numel() if _is_torch(codes) else np.asarray(codes).size
    n = total // nbytes if nbytes else 0
    c, pc = _ptr(codes, np.uint8)
    if out is None:
        out = _empty_like(codes, (n,), np.int32)
    o, po = _ptr(out, np.int32, n)
    check(ctx._lib.vg_hamming_batch(ctx._h, pa, pc, C.c_int64(nbytes), C.c_int64(n), po, _stream_ptr(stream)))
    return out


class ProductQuantizer:
    """quantization.ProductQuantizer (internal/quantization/pq.go:20-29)."""

    def __init__(self, ctx: Context, dimension: int, num_subvectors: int, num_centroids: int = 256):
        self._lib = ctx._lib
        self.ctx = ctx
        h = C.c_void_p()
        check(self._lib.vg_pq_create(ctx._h, dimension, num_subvectors, num_centroids, C.byref(h)))
        self._h = h
        self.dimension, self.num_subvectors, self.num_centroids = dimension, num_subvectors, num_centroids
        self.subvector_dim = dimension // num_subvectors

    @classmethod
    def _borrowed(cls, ctx: Context, handle, dimension: int, num_subvectors: int, num_centroids: int, owner):
        """A view of a quantizer another object owns (the inner PQ of an OptimizedProductQuantizer)."""
        self = cls.__new__(cls)
        self._lib, self.ctx, self._h, self._owner = ctx._lib, ctx, handle, owner
        self.dimension, self.num_subvectors, self.num_centroids = dimension, num_subvectors, num_centroids
        self.subvector_dim = dimension // num_subvectors
        return self

    def close(self):
        if getattr(self, "_h", None) and getattr(self, "_owner", None) is None:
            self._lib.vg_pq_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def is_trained(self) -> bool:
        return bool(self._lib.vg_pq_is_trained(self._h))

    def set_codebooks(self, codebooks, scales, offsets):
        n = self.num_subvectors * self.num_centroids * self.subvector_dim
        a, pa = _ptr(codebooks, np.int8, n)
        s, ps = _ptr(scales, np.float32, self.num_subvectors)
        o, po = _ptr(offsets, np.float32, self.num_subvectors)
        check(self._lib.vg_pq_set_codebooks(self._h, pa, ps, po))

    def codebooks(self):
        n = self.num_subvectors * self.num_centroids * self.subvector_dim
        cb = np.empty(n, np.int8); s = np.empty(self.num_subvectors, np.float32)
        o = np.empty(self.num_subvectors, np.float32)
        check(self._lib.vg_pq_get_codebooks(self._h, C.c_void_p(cb.ctypes.data),
                                            C.c_void_p(s.ctypes.data), C.c_void_p(o.ctypes.data)))
        return cb, s, o

    def train(self, vectors, iters: int = 20, seed: int = 1, stream=None):
        """Train (pq.go:68-143); the reference uses 20 Lloyd iterations."""
        n = _rows(vectors, self.dimension)
        v, pv = _ptr(vectors, np.float32)
        check(self._lib.vg_pq_train(self._h, pv, C.c_int64(n), C.c_int32(iters), C.c_uint64(seed),
                                    _stream_ptr(stream)))

    def train_subset(self, vectors, sub_begin: int, sub_count: int, iters: int = 20, seed: int = 1, stream=None):
        """Train only the sub-quantizers [sub_begin, sub_begin + sub_count) (the reference trains
        them as independent goroutines, pq.go:83-138); see vecgo_amd.sharded.train_pq_sharded."""
        n = _rows(vectors, self.dimension)
        v, pv = _ptr(vectors, np.float32)
        check(self._lib.vg_pq_train_subset(self._h, pv, C.c_int64(n), C.c_int32(iters), C.c_uint64(seed),
                                           C.c_int32(sub_begin), C.c_int32(sub_count), _stream_ptr(stream)))

    def codebooks_range(self, sub_begin: int, sub_count: int):
        per = self.num_centroids * self.subvector_dim
        cb = np.empty(sub_count * per, np.int8); s = np.empty(sub_count, np.float32)
        o = np.empty(sub_count, np.float32)
        check(self._lib.vg_pq_get_codebooks_range(self._h, C.c_int32(sub_begin), C.c_int32(sub_count),
                                                  C.c_void_p(cb.ctypes.data), C.c_void_p(s.ctypes.data),
                                                  C.c_void_p(o.ctypes.data)))
        return cb, s, o

    def encode(self, vectors, out=None, stream=None):
        """Encode (pq.go:147-176), batched: returns [n, m] uint8."""
        n = _rows(vectors, self.dimension)
        v, pv = _ptr(vectors, np.float32)
        if out is None:
            out = _empty_like(vectors, (n, self.num_subvectors), np.uint8)
        c, pc = _ptr(out, np.uint8, n * self.num_subvectors)
        check(self._lib.vg_pq_encode(self._h, pv, C.c_int64(n), pc, _stream_ptr(stream)))
        return out

    def decode(self, codes, out=None, stream=None):
        """Decode (pq.go:185-229), batched: returns [n, dim] float32."""
        total = codes.numel() if _is_torch(codes) else np.asarray(codes).size
        if total % self.num_subvectors:
            raise VecgoHipError(-4, "invalid code length")
        n = total // self.num_subvectors
        c, pc = _ptr(codes, np.uint8)
        if out is None:
            out = _empty_like(codes, (n, self.dimension), np.float32)
        o, po = _ptr(out, np.float32, n * self.dimension)
        check(self._lib.vg_pq_decode(self._h, pc, C.c_int64(n), po, _stream_ptr(stream)))
        return out

    def asymmetric_distance(self, query, codes, out=None, stream=None):
        """ComputeAsymmetricDistance (pq.go:234-260) of one query against n codes."""
        if _rows(query, self.dimension) != 1:
            raise VecgoHipError(-2, "vector dimension mismatch")
        total = codes.numel() if _is_torch(codes) else np.asarray(codes).size
        if total % self.num_subvectors:
            raise VecgoHipError(-4, "codes length mismatch")
        n = total // self.num_subvectors
        q, pq_ = _ptr(query, np.float32)
        c, pc = _ptr(codes, np.uint8)
        if out is None:
            out = _empty_like(codes, (n,), np.float32)
        o, po = _ptr(out, np.float32, n)
        check(self._lib.vg_pq_asymmetric_distance_batch(self._h, pq_, pc, C.c_int64(n), po,
                                                        _stream_ptr(stream)))
        return out

    def build_distance_table(self, queries, out=None, stream=None):
        """BuildDistanceTable (pq.go:468-491), batched: returns [nq, m*k]."""
        nq = _rows(queries, self.dimension)
        q, pq_ = _ptr(queries, np.float32)
        if out is None:
            out = _empty_like(queries, (nq, self.num_subvectors * self.num_centroids), np.float32)
        t, pt = _ptr(out, np.float32, nq * self.num_subvectors * self.num_centroids)
        check(self._lib.vg_pq_build_distance_table(self._h, pq_, C.c_int64(nq), pt, _stream_ptr(stream)))
        return out


def _rows(x, dim) -> int:
    n = x.numel() if _is_torch(x) else np.asarray(x).size
    if dim <= 0 or n % dim:
        raise VecgoHipError(-2, "vector dimension mismatch")
    return n // dim


def _empty_like(ref, shape, dtype):
    if _is_torch(ref):
        tdt = {np.float32: torch.float32, np.uint32: torch.int32, np.uint8: torch.uint8,
               np.int32: torch.int32}[dtype]
        return torch.empty(shape, dtype=tdt, device=ref.device)
    return np.empty(shape, dtype)


class Index:
    """Device-resident rows / codes / graph of one segment."""

    def __init__(self, ctx: Context, n: int, dim: int, metric: Metric = Metric.L2):
        self._lib = ctx._lib
        self.ctx = ctx
        h = C.c_void_p()
        check(self._lib.vg_index_create(ctx._h, C.c_int64(n), dim, int(metric), C.byref(h)))
        self._h = h
        self.n, self.dim, self.metric = n, dim, Metric(metric)
        self._keep = []

    @classmethod
    def _borrowed(cls, ctx: Context, handle, n: int, dim: int, metric: int, owner):
        """An Index view of a handle owned by something else (a Segment)."""
        self = cls.__new__(cls)
        self._lib, self.ctx, self._h = ctx._lib, ctx, handle
        self.n, self.dim, self.metric = n, dim, Metric(metric)
        self._keep = [owner]
        self._borrowed_handle = True
        return self

    def close(self):
        if getattr(self, "_h", None):
            if not getattr(self, "_borrowed_handle", False):
                self._lib.vg_index_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_pq_codes(self, pq: ProductQuantizer, codes, stream=None):
        c, pc = _ptr(codes, np.uint8, self.n * pq.num_subvectors)
        self._keep.append(pq)
        check(self._lib.vg_index_set_pq_codes(self._h, pq._h, pc, _stream_ptr(stream)))

    def set_sq8_codes(self, sq: "ScalarQuantizer", codes, stream=None):
        c, pc = _ptr(codes, np.uint8, self.n * self.dim)
        self._keep.append(sq)
        check(self._lib.vg_index_set_sq8_codes(self._h, sq._h, pc, _stream_ptr(stream)))

    def set_int4_codes(self, iq: "Int4Quantizer", codes, stream=None):
        c, pc = _ptr(codes, np.uint8, self.n * iq.code_bytes)
        self._keep.append(iq)
        check(self._lib.vg_index_set_int4_codes(self._h, iq._h, pc, _stream_ptr(stream)))

    def set_partitions(self, centroids, part_offsets, stream=None):
        """IVF partitions of a flat segment (flat/segment.go:187-207): centroids [P, dim] fp32 and the
        first row of every partition [P + 1] uint32; None / empty removes them."""
        if centroids is None or len(part_offsets) == 0:
            check(self._lib.vg_index_set_partitions(self._h, None, None, C.c_int32(0), _stream_ptr(stream)))
            return
        c = np.ascontiguousarray(centroids, np.float32).reshape(-1, self.dim)
        o = np.ascontiguousarray(part_offsets, np.uint32)
        if o.size != c.shape[0] + 1:
            raise ValueError("part_offsets must have one entry more than there are centroids")
        check(self._lib.vg_index_set_partitions(self._h, C.c_void_p(c.ctypes.data), C.c_void_p(o.ctypes.data),
                                                C.c_int32(c.shape[0]), _stream_ptr(stream)))

    SCAN_F32, SCAN_PQ, SCAN_SQ8 = 0, 1, 2

    def search_flat_probed(self, queries, k, nprobes=0, scan=0, out=None, stream=None):
        """flat.Segment.Search over the nprobes closest IVF partitions (flat/segment.go:727-749)."""
        return self._search(self._lib.vg_search_flat_probed, queries, k, extra=(C.c_int32(nprobes), C.c_int32(scan)),
                            out=out, stream=stream)

    def search_flat_filtered(self, queries, k, mask, nprobes=0, scan=0, out=None, stream=None):
        """flat.Segment.Search with a row filter (flat/segment.go:631-635, :559-561): the k best (score, row id) among the
        rows whose mask entry is set, over the probed partitions or the whole segment.  mask: bool[n] / packed bits for the
        batch, or bool[nq, n] / packed [nq, ceil(n/8)] for one filter per query; None = search_flat_probed."""
        if mask is None:
            return self.search_flat_probed(queries, k, nprobes, scan, out=out, stream=stream)
        m, pm, stride = self._packed_mask(mask, _rows(queries, self.dim), "search_flat_filtered")
        return self._search(self._lib.vg_search_flat_filtered, queries, k,
                            extra=(C.c_int32(nprobes), C.c_int32(scan), pm, C.c_int64(stride)),
                            out=out, stream=stream)

    def enable_pq_nomination(self, on: bool = True, stream=None):
        """vg_index_enable_pq_nomination: batches of search_pq_adc (queries x rows >= 24M, k <= 256, K = 256) are nominated by a bfloat16
        MFMA GEMM over the decoded rows (+ n * dim * 2 bytes), re-scored from the codes against the query's distance table in
        the reference's order and proven: ids and scores stay bit-identical."""
        check(self._lib.vg_index_enable_pq_nomination(self._h, C.c_int32(1 if on else 0), _stream_ptr(stream)))

    def enable_sq8_nomination(self, on: bool = True, stream=None):
        """vg_index_enable_sq8_nomination: batches of search_sq8 (L2 or Dot, 5 queries up, k <= 256, any dim) are nominated by a
        bfloat16 MFMA GEMM over the dequantised rows (+ n * dim * 2 bytes), re-scored exactly from the codes and proven: ids and
        scores stay bit-identical."""
        check(self._lib.vg_index_enable_sq8_nomination(self._h, C.c_int32(1 if on else 0), _stream_ptr(stream)))

    def search_sq8(self, queries, k, out=None, stream=None):
        """flat.Segment.Search SQ8 branch (flat/segment.go:517-604)."""
        return self._search(self._lib.vg_search_sq8, queries, k, out=out, stream=stream)

    def set_rabitq_codes(self, codes, stream=None):
        lib = self._lib
        lib.vg_rabitq_code_bytes.restype = C.c_int64
        cb = int(lib.vg_rabitq_code_bytes(C.c_int32(self.dim)))
        c, pc = _ptr(codes, np.uint8, self.n * cb)
        check(lib.vg_index_set_rabitq_codes(self._h, pc, _stream_ptr(stream)))

    def search_rabitq(self, queries, k, out=None, stream=None):
        """Exhaustive RaBitQ scan (rabitq.go:119-176 per row), top-k by (Score, RowID)."""
        return self._search(self._lib.vg_search_rabitq, queries, k, out=out, stream=stream)

    def set_hnsw_graph(self, l0, upper=(), entry_point=0, m=None, stream=None):
        """l0: [n, m0] uint32 (0xFFFFFFFF terminates a list); upper: list of (slot[n], adj[rows, m])
        for levels 1..L."""
        l0a = np.ascontiguousarray(l0, np.uint32)
        m0 = l0a.shape[1]
        L = len(upper)
        if L:
            m_ = upper[0][1].shape[1]
            slots = np.ascontiguousarray(np.stack([np.asarray(s_, np.uint32) for s_, _ in upper]))
            adj = np.ascontiguousarray(np.concatenate([np.asarray(a, np.uint32).reshape(-1, m_) for _, a in upper]))
            rows = np.array([np.asarray(a).reshape(-1, m_).shape[0] for _, a in upper], np.int64)
            ps, pa, pr = (C.c_void_p(slots.ctypes.data), C.c_void_p(adj.ctypes.data),
                          C.c_void_p(rows.ctypes.data))
        else:
            m_ = m if m is not None else max(1, m0 // 2)
            ps = pa = pr = None
        check(self._lib.vg_index_set_hnsw_graph(self._h, C.c_int32(m0), C.c_void_p(l0a.ctypes.data),
                                                C.c_int32(L), C.c_int32(m_), ps, pa, pr,
                                                C.c_uint32(entry_point), _stream_ptr(stream)))

    def build_hnsw(self, m=32, ef_construction=300, max_batch=8192, growth_div=32, stream=None):
        """hnsw.Insert over rows 0..n-1 on the GPU (hnsw.go:713-984; ids and levels of ApplyInsert); the
        graph becomes the index's HNSW graph.  max_batch=1 is the reference's sequential loop."""
        check(self._lib.vg_hnsw_build(self._h, C.c_int32(m), C.c_int32(ef_construction), C.c_int32(max_batch),
                                      C.c_int32(growth_div), _stream_ptr(stream)))

    def get_hnsw_graph(self, stream=None):
        """(l0[n, m0], upper=[(slot[n], adj[rows, m])...], entry_point) — set_hnsw_graph's arguments."""
        m0, m_, L, ep = C.c_int32(), C.c_int32(), C.c_int32(), C.c_uint32()
        rows = np.zeros(64, np.int64)
        sp = _stream_ptr(stream)
        check(self._lib.vg_index_get_hnsw_graph(self._h, C.byref(m0), C.byref(m_), C.byref(L), C.byref(ep),
                                                C.c_void_p(rows.ctypes.data), None, None, None, sp))
        L = L.value
        rows = rows[:L]
        l0 = np.empty((self.n, m0.value), np.uint32)
        slots = np.empty((max(L, 1), self.n), np.uint32)
        adj = np.empty((max(int(rows.sum()), 1), m_.value), np.uint32)
        check(self._lib.vg_index_get_hnsw_graph(self._h, None, None, None, None, None, C.c_void_p(l0.ctypes.data),
                                                C.c_void_p(slots.ctypes.data), C.c_void_p(adj.ctypes.data), sp))
        upper, off = [], 0
        for l in range(L):
            upper.append((slots[l].copy(), adj[off:off + int(rows[l])].copy()))
            off += int(rows[l])
        return l0, upper, int(ep.value)

    def set_vamana_graph(self, graph, entry_point, stream=None):
        g = np.ascontiguousarray(graph, np.uint32)
        check(self._lib.vg_index_set_vamana_graph(self._h, C.c_int32(g.shape[1]), C.c_void_p(g.ctypes.data),
                                                  C.c_uint32(entry_point), _stream_ptr(stream)))

    def _graph_search(self, fn, queries, k, mid_arg, want_stats, stream):
        nq = _rows(queries, self.dim)
        q, pq_ = _ptr(queries, np.float32)
        ids = _empty_like(queries, (nq, k), np.uint32)
        scores = _empty_like(queries, (nq, k), np.float32)
        i, pi = _ptr(ids, np.uint32)
        s_, ps = _ptr(scores, np.float32)
        stats = np.zeros((nq, 5), np.int64) if want_stats else None
        pst = C.c_void_p(stats.ctypes.data) if want_stats else None
        check(fn(self._h, pq_, C.c_int64(nq), C.c_int32(k), C.c_int32(mid_arg), pi, ps, pst, _stream_ptr(stream)))
        if want_stats and want_stats != "full":
            stats = stats[:, :4]     # the reference's FilterGateStats columns; "full" adds descent_distance_computations
        return (ids, scores, stats) if want_stats else (ids, scores)

    def search_hnsw(self, queries, k, ef, stats=False, stream=None):
        """hnsw.KNNSearch (hnsw.go:1650-1755); stats columns: nodes_visited,
        distance_computations, distance_short_circuits, pops (stats="full": + rows scored by the descent)."""
        return self._graph_search(self._lib.vg_search_hnsw, queries, k, ef, stats, stream)

    def search_hnsw_pq(self, queries, k, ef, stats=False, stream=None):
        """searchLayer with distFunc = pq.ComputeAsymmetricDistance over the nodes' PQ codes (the candidate
        stage of graph -> PQ -> exact rerank); scores are PQ distances."""
        return self._graph_search(self._lib.vg_search_hnsw_pq, queries, k, ef, stats, stream)

    def _packed_mask(self, mask, nq, what):
        """bool[n] / bool[nq, n] / packed little-endian bits -> (keepalive, void*, stride in bytes; 0 = one mask).  A torch
        uint8 tensor (host or device) is taken as packed bits where it lies."""
        if _is_torch(mask):
            if mask.dtype != torch.uint8:
                raise TypeError(f"{what}: a torch mask holds packed bits (uint8), got {mask.dtype}")
            m = mask if mask.is_contiguous() else mask.contiguous()
            if m.shape[-1] < (self.n + 7) // 8:
                raise ValueError(f"{what}: a packed mask holds ceil(n / 8) = {(self.n + 7) // 8} bytes, got {m.shape[-1]}")
            stride = 0
            if m.ndim > 1 and m.shape[0] > 1:
                if m.shape[0] != nq:
                    raise ValueError(f"{what}: one mask per query ({nq}), got {m.shape[0]}")
                stride = m.shape[1]
            return m, C.c_void_p(m.data_ptr()), stride
        m = np.asarray(mask)
        if m.dtype == np.bool_:
            if m.shape[-1] != self.n:   # the C side reads ceil(n / 8) bytes per mask: a short one would be read past its end
                raise ValueError(f"{what}: a bool mask has one entry per row ({self.n}), got {m.shape[-1]}")
            m = np.packbits(m.reshape(-1, self.n) if m.ndim > 1 else m, axis=-1, bitorder="little")
        m = np.ascontiguousarray(m, np.uint8)
        if m.shape[-1] < (self.n + 7) // 8:
            raise ValueError(f"{what}: a packed mask holds ceil(n / 8) = {(self.n + 7) // 8} bytes, got {m.shape[-1]}")
        stride = 0
        if m.ndim > 1 and m.shape[0] > 1:
            if m.shape[0] != nq:
                raise ValueError(f"{what}: one mask per query ({nq}), got {m.shape[0]}")
            stride = m.shape[1]
        return m, C.c_void_p(m.ctypes.data), stride

    def search_hnsw_filtered(self, queries, k, ef, mask, selectivity, stats=False, stream=None):
        """searchExecute with a filter whose selectivity hint is above 0.3: searchLayerWithPostFilter (hnsw.go:1159-1218) —
        the walk with an expanded ef, the results re-filtered through `mask` (bool[n] / packed bits, one for the batch or
        one per query) and capped at ef.  `ef` = what determineEF returned."""
        nq = _rows(queries, self.dim)
        m, pm, stride = self._packed_mask(mask, nq, "search_hnsw_filtered")
        q, pq_ = _ptr(queries, np.float32)
        ids = _empty_like(queries, (nq, k), np.uint32)
        scores = _empty_like(queries, (nq, k), np.float32)
        i, pi = _ptr(ids, np.uint32)
        s_, ps = _ptr(scores, np.float32)
        st = np.zeros((nq, 5), np.int64) if stats else None
        pst = C.c_void_p(st.ctypes.data) if stats else None
        check(self._lib.vg_search_hnsw_filtered(self._h, pq_, C.c_int64(nq), C.c_int32(k), C.c_int32(ef), pm,
                                                C.c_int64(stride), C.c_double(float(selectivity)), pi, ps, pst, _stream_ptr(stream)))
        if stats and stats != "full":
            st = st[:, :4]
        return (ids, scores, st) if stats else (ids, scores)

    def set_hnsw_edge_distances(self, l0_dist=None, stream=None):
        """The layer-0 lists' cached Neighbor.Dist [n, m0] (node.go:62-80) for search_hnsw_predicate; None = recompute them
        from the fp32 rows (the distance between the two nodes, what the insert stored)."""
        if l0_dist is None:
            check(self._lib.vg_index_set_hnsw_edge_distances(self._h, None, _stream_ptr(stream)))
            return
        d, pd = _ptr(l0_dist, np.float32)
        check(self._lib.vg_index_set_hnsw_edge_distances(self._h, pd, _stream_ptr(stream)))

    def set_hnsw_tombstones(self, deleted=None, stream=None):
        """g.tombstones (hnsw.go:95): bool[n] / packed bits, None clears.  Deleted nodes are walked through, never returned
        (hnsw.go:1381-1390); read by search_hnsw, search_hnsw_pq, search_hnsw_filtered, search_hnsw_predicate."""
        if deleted is None:
            check(self._lib.vg_index_set_hnsw_tombstones(self._h, None, _stream_ptr(stream)))
            return
        d, pd, _ = self._packed_mask(deleted, 1, "set_hnsw_tombstones")
        check(self._lib.vg_index_set_hnsw_tombstones(self._h, pd, _stream_ptr(stream)))

    def search_hnsw_predicate(self, queries, k, ef, mask, deleted=None, stats=False, stream=None):
        """searchExecute with a filter whose selectivity hint is <= 0.3 or unknown: searchLayerPredicateAware
        (hnsw.go:1406-1558).  mask: filter.Matches (bool[n] / packed bits, one for the batch or one per query); deleted: the
        tombstone bitmap (bool[n] / packed bits) or None.  stats columns: nodes_visited, distance_computations,
        ExpansionsSkipped, pops."""
        nq = _rows(queries, self.dim)
        m, pm, stride = self._packed_mask(mask, nq, "search_hnsw_predicate")
        d, pdl, _ = (None, None, 0) if deleted is None else self._packed_mask(deleted, 1, "search_hnsw_predicate (deleted)")
        q, pq_ = _ptr(queries, np.float32)
        ids = _empty_like(queries, (nq, k), np.uint32)
        scores = _empty_like(queries, (nq, k), np.float32)
        i, pi = _ptr(ids, np.uint32)
        s_, ps = _ptr(scores, np.float32)
        st = np.zeros((nq, 5), np.int64) if stats else None
        pst = C.c_void_p(st.ctypes.data) if stats else None
        check(self._lib.vg_search_hnsw_predicate(self._h, pq_, C.c_int64(nq), C.c_int32(k), C.c_int32(ef), pm, C.c_int64(stride), pdl,
                                                 pi, ps, pst, _stream_ptr(stream)))
        if stats and stats != "full":
            st = st[:, :4]
        return (ids, scores, st) if stats else (ids, scores)

    BRUTE_SCAN, BRUTE_BITMAP = 0, 1

    def search_hnsw_brute(self, queries, k, mode=0, mask=None, stream=None):
        """hnsw.BruteSearch + scanSegment (hnsw.go:2021-2101; mode BRUTE_SCAN) or searchBitmap + extraction
        (:2240-2263, :1732-1751; BRUTE_BITMAP): every row whose mask bit is set, in id order, through the
        reference's PriorityQueue.  mask: None, bool[n] / packed bits for the whole batch, or bool[nq, n] / packed
        [nq, ceil(n/8)] for one mask per query.  HNSW distances (L2, -dot, 0.5 * L2), best first."""
        nq = _rows(queries, self.dim)
        q, pq_ = _ptr(queries, np.float32)
        ids = _empty_like(queries, (nq, k), np.uint32)
        scores = _empty_like(queries, (nq, k), np.float32)
        i, pi = _ptr(ids, np.uint32)
        s_, ps = _ptr(scores, np.float32)
        m, pm, stride = (None, None, 0) if mask is None else self._packed_mask(mask, nq, "search_hnsw_brute")
        check(self._lib.vg_search_hnsw_brute(self._h, pq_, C.c_int64(nq), C.c_int32(k), C.c_int32(mode), pm,
                                             C.c_int64(stride), pi, ps, _stream_ptr(stream)))
        return ids, scores

    def search_vamana(self, queries, k, kind=0, stats=False, stream=None):
        """diskann searchInternal (diskann/segment.go:503-706); kind 0 fp32, 1 PQ, 2 RaBitQ, 3 INT4."""
        return self._graph_search(self._lib.vg_search_vamana, queries, k, kind, stats, stream)

    def search_vamana_filtered(self, queries, k, mask, kind=0, stats=False, stream=None):
        """searchInternal with `filter` set (diskann/segment.go:616-627): rows whose mask entry is clear are walked through
        but never enter the result heap.  mask: bool[n] / packed bits for the batch, or one per query."""
        if mask is None:
            return self.search_vamana(queries, k, kind, stats, stream)
        nq = _rows(queries, self.dim)
        m, pm, stride = self._packed_mask(mask, nq, "search_vamana_filtered")
        q, pq_ = _ptr(queries, np.float32)
        ids = _empty_like(queries, (nq, k), np.uint32)
        scores = _empty_like(queries, (nq, k), np.float32)
        i, pi = _ptr(ids, np.uint32)
        s_, ps = _ptr(scores, np.float32)
        st = np.zeros((nq, 5), np.int64) if stats else None
        pst = C.c_void_p(st.ctypes.data) if stats else None
        check(self._lib.vg_search_vamana_filtered(self._h, pq_, C.c_int64(nq), C.c_int32(k), C.c_int32(kind), pm, C.c_int64(stride),
                                                  pi, ps, pst, _stream_ptr(stream)))
        if stats and stats != "full":
            st = st[:, :4]
        return (ids, scores, st) if stats else (ids, scores)

    def set_vectors(self, base, stream=None):
        """fp32 rows, n*dim row-major (vectorstore/columnar.go:21-24)."""
        b, pb = _ptr(base, np.float32, self.n * self.dim)
        check(self._lib.vg_index_set_vectors(self._h, pb, _stream_ptr(stream)))

    def rerank(self, queries, cand_ids, k, out=None, stream=None):
        """Segment.Rerank + top-k (flat/segment.go:754-780): cand_ids is [nq, nc]."""
        nq = _rows(queries, self.dim)
        nc = (cand_ids.numel() if _is_torch(cand_ids) else np.asarray(cand_ids).size) // max(nq, 1)
        q, pq_ = _ptr(queries, np.float32)
        c, pc = _ptr(cand_ids, np.uint32, nq * nc)
        if out is None:
            out = (_empty_like(queries, (nq, k), np.uint32), _empty_like(queries, (nq, k), np.float32))
        i, pi = _ptr(out[0], np.uint32, nq * k)
        s, ps = _ptr(out[1], np.float32, nq * k)
        check(self._lib.vg_rerank(self._h, pq_, C.c_int64(nq), pc, C.c_int32(nc), C.c_int32(k), pi, ps,
                                  _stream_ptr(stream)))
        return out

    def robust_prune(self, nodes, cands, r, alpha=1.2, stream=None):
        """Vamana robustPrune (diskann/writer.go:571-625) for a batch of nodes: cands is [n_nodes, nc]
        uint32 (VG_INVALID_ID padded); returns (kept[n_nodes, r], counts[n_nodes])."""
        nd = np.ascontiguousarray(nodes, np.uint32)
        cd = np.ascontiguousarray(cands, np.uint32).reshape(nd.size, -1)
        out = np.empty((nd.size, r), np.uint32); cnt = np.empty(nd.size, np.int32)
        check(self._lib.vg_robust_prune(self._h, C.c_void_p(nd.ctypes.data), C.c_int64(nd.size),
                                        C.c_void_p(cd.ctypes.data), C.c_int32(cd.shape[1]), C.c_int32(r),
                                        C.c_float(alpha), C.c_void_p(out.ctypes.data), C.c_void_p(cnt.ctypes.data),
                                        _stream_ptr(stream)))
        return out, cnt

    def hnsw_select_neighbors(self, cand_ids, cand_dists, m, stream=None):
        """HNSW selectNeighborsHeuristic (hnsw.go:1009-1106) for a batch of nodes: cand_ids / cand_dists
        are [n_nodes, nc], nearest first; returns (kept[n_nodes, m], counts[n_nodes])."""
        ci = np.ascontiguousarray(cand_ids, np.uint32)
        cdst = np.ascontiguousarray(cand_dists, np.float32)
        nn, nc = ci.shape
        out = np.empty((nn, m), np.uint32); cnt = np.empty(nn, np.int32)
        check(self._lib.vg_hnsw_select_neighbors(self._h, C.c_int64(nn), C.c_void_p(ci.ctypes.data),
                                                 C.c_void_p(cdst.ctypes.data), C.c_int32(nc), C.c_int32(m),
                                                 C.c_void_p(out.ctypes.data), C.c_void_p(cnt.ctypes.data),
                                                 _stream_ptr(stream)))
        return out, cnt

    def score_candidates(self, queries, cand_ids, out=None, stream=None):
        nq = _rows(queries, self.dim)
        nc = (cand_ids.numel() if _is_torch(cand_ids) else np.asarray(cand_ids).size) // max(nq, 1)
        q, pq_ = _ptr(queries, np.float32)
        c, pc = _ptr(cand_ids, np.uint32, nq * nc)
        if out is None:
            out = _empty_like(queries, (nq, nc), np.float32)
        s, ps = _ptr(out, np.float32, nq * nc)
        check(self._lib.vg_score_candidates(self._h, pq_, C.c_int64(nq), pc, C.c_int32(nc), ps,
                                            _stream_ptr(stream)))
        return out

    def _search(self, fn, queries, k, extra=(), out=None, stream=None):
        nq = _rows(queries, self.dim)
        q, pq_ = _ptr(queries, np.float32)
        if out is None:
            ids = _empty_like(queries, (nq, k), np.uint32)
            scores = _empty_like(queries, (nq, k), np.float32)
        else:
            ids, scores = out
        i, pi = _ptr(ids, np.uint32, nq * k)
        s, ps = _ptr(scores, np.float32, nq * k)
        check(fn(self._h, pq_, C.c_int64(nq), C.c_int32(k), *extra, pi, ps, _stream_ptr(stream)))
        return ids, scores

    def search_flat(self, queries, k, out=None, stream=None):
        """flat.Segment.Search fp32 branch / hnsw.BruteSearch: exact brute force."""
        return self._search(self._lib.vg_search_flat, queries, k, out=out, stream=stream)

    def enable_bf16_filter(self, on: bool = True, stream=None):
        """vg_index_enable_bf16_filter: nominate with a bfloat16 MFMA GEMM over a bf16 copy of the rows; the exact fp32
        re-score and the (widened) proof keep ids and scores bit-identical."""
        check(self._lib.vg_index_enable_bf16_filter(self._h, C.c_int32(1 if on else 0), _stream_ptr(stream)))

    def flat_stats(self, stream=None):
        """(queries searched, queries answered by the exhaustive kernel) since set_vectors."""
        q, e = C.c_int64(0), C.c_int64(0)
        check(self._lib.vg_index_flat_stats(self._h, C.byref(q), C.byref(e), _stream_ptr(stream)))
        return q.value, e.value

    def search_pq_adc(self, queries, k, out=None, stream=None):
        """flat.Segment.Search PQ branch (flat/segment.go:476-483,678-689,714-721)."""
        return self._search(self._lib.vg_search_pq_adc, queries, k, out=out, stream=stream)


class SegmentInfo(C.Structure):
    _fields_ = [("segment_id", C.c_uint64), ("rows", C.c_int64), ("dim", C.c_int32), ("metric", C.c_int32),
                ("kind", C.c_int32), ("quantization", C.c_int32), ("pq_m", C.c_int32), ("pq_k", C.c_int32),
                ("max_degree", C.c_int32), ("search_list_size", C.c_int32), ("entrypoint", C.c_uint32),
                ("num_partitions", C.c_int32)]


class Segment:
    """A flat or DiskANN segment file of the reference (internal/segment/flat/format.go,
    internal/segment/diskann/format.go) opened straight onto the GPU: `image` is the whole file
    as bytes / numpy uint8 / np.memmap.  `index` is searched like any other Index."""

    def __init__(self, ctx: Context, image, kind: str = "flat", verify_checksum: bool = True, stream=None):
        self._lib, self.ctx = ctx._lib, ctx
        buf = np.frombuffer(image, np.uint8) if isinstance(image, (bytes, bytearray, memoryview)) else \
            np.ascontiguousarray(image, np.uint8)
        fn = {"flat": self._lib.vg_segment_open_flat, "diskann": self._lib.vg_segment_open_diskann}[kind]
        h = C.c_void_p()
        check(fn(ctx._h, C.c_void_p(buf.ctypes.data), C.c_int64(buf.size), C.c_int32(int(verify_checksum)),
                 C.byref(h), _stream_ptr(stream)))
        self._h = h
        info = SegmentInfo()
        check(self._lib.vg_segment_get_info(self._h, C.byref(info)))
        self.info = info
        self._lib.vg_segment_index.restype = C.c_void_p
        ih = C.c_void_p(self._lib.vg_segment_index(self._h))
        self.index = Index._borrowed(ctx, ih, int(info.rows), int(info.dim), int(info.metric), self)

    def search(self, queries, k, nprobes=0, out=None, stream=None):
        """flat.Segment.Search (flat/segment.go:447-751): scan type by the segment's quantization,
        IVF partitions probed when the segment has more than one."""
        seg = self._h

        def fn(_index_handle, *args):  # same argument list as the index searches, segment handle first
            return self._lib.vg_segment_search(seg, *args)
        return self.index._search(fn, queries, k, extra=(C.c_int32(nprobes),), out=out, stream=stream)

    def search_filtered(self, queries, k, mask, nprobes=0, out=None, stream=None):
        """Segment.Search with a row filter (flat/segment.go:631-635, diskann/segment.go:616-627): mask bool[n] / packed bits
        for the batch or one per query; None = search."""
        if mask is None:
            return self.search(queries, k, nprobes, out=out, stream=stream)
        seg = self._h
        m, pm, stride = self.index._packed_mask(mask, _rows(queries, self.index.dim), "Segment.search_filtered")

        def fn(_index_handle, *args):
            return self._lib.vg_segment_search_filtered(seg, *args)
        return self.index._search(fn, queries, k, extra=(C.c_int32(nprobes), pm, C.c_int64(stride)), out=out, stream=stream)

    def close(self):
        if getattr(self, "_h", None):
            self.index._h = None
            self._lib.vg_segment_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def crc32c(data) -> int:
    """hash.CRC32C (internal/hash/crc32c.go:15-17) as computed by the library."""
    from . import _lib
    lib = _lib.load()
    lib.vg_crc32c.restype = C.c_uint32
    buf = np.frombuffer(data, np.uint8) if isinstance(data, (bytes, bytearray, memoryview)) else \
        np.ascontiguousarray(data, np.uint8)
    return int(lib.vg_crc32c(C.c_void_p(buf.ctypes.data), C.c_int64(buf.size)))
