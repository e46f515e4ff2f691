"""vecgo_amd — MI355X (gfx950) implementation of vecgo's distance + quantization hot path.

The product is libvecgo_hip.so (C ABI: include/vecgo_hip.h).  This package is the thin
Python binding the tests and bench.py drive it with; names mirror the reference's Go
interfaces (distance.Metric, quantization.ProductQuantizer, ...).
"""
from .api import (Context, Index, Metric, ProductQuantizer, RaBitQuantizer, VecgoHipError,  # noqa: F401
                  dot_batch, hamming_batch, merge_topk, squared_l2_batch)

__all__ = ["Context", "Index", "Metric", "ProductQuantizer", "RaBitQuantizer", "VecgoHipError",
           "dot_batch", "hamming_batch", "merge_topk", "squared_l2_batch"]
