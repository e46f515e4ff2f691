"""Generate tests/golden/int4_ref.npz from the reference's own AVX-512 INT4 kernels.

Run in the build container only (needs /root/reference, clang and an AVX-512 CPU):
    make -C oracle ref && python tests/golden/make_golden_int4.py
Seeded numpy inputs; expected outputs from oracle/_ref/libvecgo_ref_avx512.so, i.e.
internal/simd/src/int4_avx512.c (int4L2DistanceAvx512, int4L2DistancePrecomputedAvx512,
int4L2DistanceBatchAvx512) compiled in place with the generator's flags.  Only inputs and
outputs are stored.  The lookup table fed to the precomputed kernel is the oracle's restatement
of simd.BuildInt4LookupTable (Go, kernels.go:94-103) and is stored with the inputs.
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from oracle import oracle as o  # noqa: E402

ref = o.Ref()
assert ref.ok, "oracle/_ref missing or CPU lacks AVX-512"
rng = np.random.default_rng(20260204)
Q, MN, DF, CD, TB, DIM, N, BATCH, SINGLE, PRE = [], [], [], [], [], [], [], [], [], []
for dim in [1, 2, 3, 15, 16, 17, 31, 32, 33, 63, 64, 65, 96, 100, 128, 768]:
    n = 3
    q = rng.standard_normal(dim).astype(np.float32)
    mn = rng.standard_normal(dim).astype(np.float32)
    df = (rng.random(dim) * 3 + 0.1).astype(np.float32)
    iq = o.Int4Quantizer(dim); iq.set_params(mn, df)
    codes = rng.integers(0, 256, n * iq.code_size).astype(np.uint8)
    Q.append(q); MN.append(mn); DF.append(df); CD.append(codes); TB.append(iq.table); DIM.append(dim); N.append(n)
    BATCH.append(ref.int4_l2_batch(q, codes, dim, mn, df))
    SINGLE.append(ref.int4_l2(q, codes[:iq.code_size], mn, df))
    PRE.append(ref.int4_l2_precomputed(q, codes[:iq.code_size], iq.table))
np.savez_compressed(Path(__file__).with_name("int4_ref.npz"), q=np.concatenate(Q), min=np.concatenate(MN),
                    diff=np.concatenate(DF), codes=np.concatenate(CD), table=np.concatenate(TB),
                    dim=np.array(DIM, np.int64), n=np.array(N, np.int64), batch=np.concatenate(BATCH),
                    single=np.array(SINGLE, np.float32), precomputed=np.array(PRE, np.float32))
print("wrote int4_ref.npz", len(DIM), "cases")
