"""Generate tests/golden/l0_ref.npz from the reference's own AVX-512 kernels.

Run in the build container only (needs /root/reference, clang and an AVX-512 CPU):
    make -C oracle ref && python tests/golden/make_golden.py
Inputs are seeded numpy draws; expected outputs come from
oracle/_ref/libvecgo_ref_avx512.so, i.e. the reference's C sources
(internal/simd/src/{floats,batch,bounded_l2,popcount}_avx512.c) compiled in place
with its generator's flags.  Only inputs and outputs are stored.
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from oracle import oracle as o  # noqa: E402

ref = o.Ref()
assert ref.ok, "oracle/_ref missing or CPU lacks AVX-512"
rng = np.random.default_rng(20260130)
out = {}

# pair kernels: dot / l2 / bounded over the reference's boundary sizes
# (floats_test.go:195-216) plus the BASELINE dims
sizes = [0, 1, 3, 4, 7, 8, 15, 16, 17, 31, 32, 33, 63, 64, 65, 128, 129, 200, 768, 777, 1536]
A, B, off = [], [], [0]
dots, l2s, bnd_in, bnd_d, bnd_e = [], [], [], [], []
for n in sizes:
    for rep in range(3):
        a = rng.standard_normal(n).astype(np.float32)
        b = rng.standard_normal(n).astype(np.float32)
        A.append(a); B.append(b); off.append(off[-1] + n)
        dots.append(ref.dot(a, b)); l2s.append(ref.l2(a, b))
        full = float(l2s[-1])
        for bound in (full * 0.5, full, full * 1.25, 0.0):
            d, e = ref.l2_bounded(a, b, bound)
            bnd_in.append(bound); bnd_d.append(d); bnd_e.append(e)
out.update(pair_a=np.concatenate(A), pair_b=np.concatenate(B), pair_off=np.array(off, np.int64),
           pair_dot=np.array(dots, np.float32), pair_l2=np.array(l2s, np.float32),
           bounded_bound=np.array(bnd_in, np.float32), bounded_dist=np.array(bnd_d, np.float32),
           bounded_exceeded=np.array(bnd_e, np.uint8))

# batch kernels (floats_test.go:218-276 dims x n, plus d=768)
bq, bt, bdims, bns, bl2, bdot = [], [], [], [], [], []
for dim in [1, 3, 7, 16, 33, 64, 80, 100, 768]:
    for n in [1, 5, 17]:
        q = rng.standard_normal(dim).astype(np.float32)
        t = rng.standard_normal(n * dim).astype(np.float32)
        bq.append(q); bt.append(t); bdims.append(dim); bns.append(n)
        bl2.append(ref.l2_batch(q, t, dim)); bdot.append(ref.dot_batch(q, t, dim))
out.update(batch_q=np.concatenate(bq), batch_t=np.concatenate(bt),
           batch_dim=np.array(bdims, np.int64), batch_n=np.array(bns, np.int64),
           batch_l2=np.concatenate(bl2), batch_dot=np.concatenate(bdot))

# ADC lookups (floats_test.go:420-447 m values, plus 96/100)
at, ac, ams, ares = [], [], [], []
for m in [1, 2, 7, 8, 9, 15, 16, 17, 32, 33, 96, 100]:
    for rep in range(4):
        table = (rng.standard_normal(m * 256) ** 2).astype(np.float32)
        codes = rng.integers(0, 256, m).astype(np.uint8)
        at.append(table); ac.append(codes); ams.append(m); ares.append(ref.adc(table, codes, m))
out.update(adc_table=np.concatenate(at), adc_codes=np.concatenate(ac),
           adc_m=np.array(ams, np.int64), adc_out=np.array(ares, np.float32))

# Hamming
ha, hb, hn, hr = [], [], [], []
for n in [1, 7, 8, 63, 64, 65, 96, 100, 128, 192]:
    for rep in range(3):
        a = rng.integers(0, 256, n).astype(np.uint8); b = rng.integers(0, 256, n).astype(np.uint8)
        ha.append(a); hb.append(b); hn.append(n); hr.append(ref.hamming(a, b))
out.update(ham_a=np.concatenate(ha), ham_b=np.concatenate(hb), ham_n=np.array(hn, np.int64),
           ham_out=np.array(hr, np.int64))

np.savez_compressed(Path(__file__).with_name("l0_ref.npz"), **out)
print("wrote l0_ref.npz:", {k: v.shape for k, v in out.items()})
