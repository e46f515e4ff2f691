"""Generate tests/golden/sq8_ref.npz from the reference's own AVX-512 SQ8 kernel.

Run in the build container only (needs /root/reference, clang and an AVX-512 CPU):
    make -C oracle ref && python tests/golden/make_golden_sq8.py
Seeded numpy inputs; expected outputs from oracle/_ref/libvecgo_ref_avx512.so, i.e.
internal/simd/src/sq8_avx512.c (sq8uL2BatchPerDimensionAvx512) compiled in place with the
generator's flags.  Only inputs and outputs are stored.  Dims/batch sizes are those of the
reference's own boundary test (internal/simd/floats_test.go:471-500) plus the BASELINE dims.
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from oracle import oracle as o  # noqa: E402

ref = o.Ref()
assert ref.ok, "oracle/_ref missing or CPU lacks AVX-512"
rng = np.random.default_rng(20260203)
Q, CD, MN, IV, DIM, N, OUT = [], [], [], [], [], [], []
for dim in [1, 7, 8, 15, 16, 17, 31, 32, 33, 100, 128, 768]:
    for n in [1, 2, 5]:
        q = (rng.random(dim) * 2 - 1).astype(np.float32)
        mn = (rng.random(dim) * 2 - 1).astype(np.float32)
        iv = (rng.random(dim) * 2 - 1).astype(np.float32)
        codes = rng.integers(0, 256, n * dim).astype(np.uint8)
        Q.append(q); MN.append(mn); IV.append(iv); CD.append(codes); DIM.append(dim); N.append(n)
        OUT.append(ref.sq8u_l2_batch(q, codes, mn, iv, dim))
np.savez_compressed(Path(__file__).with_name("sq8_ref.npz"), q=np.concatenate(Q), mins=np.concatenate(MN),
                    inv=np.concatenate(IV), codes=np.concatenate(CD), dim=np.array(DIM, np.int64),
                    n=np.array(N, np.int64), out=np.concatenate(OUT))
print("wrote sq8_ref.npz", len(DIM), "cases")
