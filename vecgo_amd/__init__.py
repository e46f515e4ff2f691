"""vecgo_amd — MI355X (gfx950) implementation of vecgo's distance + quantization hot path.

The product is libvecgo_hip.so (C ABI: include/vecgo_hip.h).  This package is the thin
Python binding the tests and bench.py drive it with; names mirror the reference's Go
interfaces (distance.Metric, quantization.ProductQuantizer, ...).
"""
from .api import (BinaryQuantizer, Comm, Context, Index, Int4Quantizer, Metric, OptimizedProductQuantizer, ProductQuantizer, RaBitQuantizer, ScalarQuantizer, Segment, VecgoHipError, crc32c,  # noqa: F401
                  dot_batch, find_closest_centroids, hamming_batch, heap_replay, kmeans_assign, kmeans_train,
                  merge_topk, merge_topk_packed, normalize_l2, pq_adc_lookup_batch, squared_l2_batch,
                  squared_l2_bounded_batch)

__all__ = ["BinaryQuantizer", "Comm", "Context", "Index", "Int4Quantizer", "Metric", "OptimizedProductQuantizer", "ProductQuantizer", "RaBitQuantizer", "ScalarQuantizer", "Segment", "VecgoHipError", "crc32c",
           "dot_batch", "find_closest_centroids", "hamming_batch", "heap_replay", "kmeans_assign", "kmeans_train",
           "merge_topk", "merge_topk_packed", "normalize_l2", "pq_adc_lookup_batch", "squared_l2_batch", "squared_l2_bounded_batch"]
