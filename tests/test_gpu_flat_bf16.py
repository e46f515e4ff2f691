"""vg_index_enable_bf16_filter: the flat search with its nomination GEMMs in bfloat16 must return the same ids and the
same fp32 scores as without the filter — i.e. the reference's (flat/segment.go:691-721 order and tie-break), which is
what the oracle gives.  The filter only changes WHICH rows are nominated; the exact re-score and the widened proof (or
the exhaustive fallback) decide the result."""
import numpy as np
import pytest

from oracle import oracle as o

pytestmark = pytest.mark.gpu


def bits(x):
    return np.asarray(x, np.float32).view(np.uint32)


@pytest.fixture(scope="module")
def vg():
    import vecgo_amd
    return vecgo_amd


@pytest.fixture(scope="module")
def ctx(vg):
    return vg.Context(0)


@pytest.mark.parametrize("n,dim,nq,k,metric", [
    (20000, 768, 130, 10, 0),     # BASELINE shape, two query tiles (one ragged)
    (20000, 768, 70, 1, 0),
    (5000, 128, 200, 10, 0),
    (9000, 1024, 65, 32, 0),
    (20000, 768, 130, 10, 2),     # Dot
    (20000, 768, 130, 10, 1),     # Cosine (the reference scores it as Dot on the rows as given)
    (3000, 64, 129, 64, 0),       # k = the candidate budget
    (30000, 192, 100, 100, 0),    # k > 64: every appended row is re-scored
    (20000, 768, 5, 10, 0),       # one block of 32 queries (the HBM-bound tile)
    (20000, 768, 33, 10, 0),      # two blocks
    (20000, 768, 64, 10, 2),
])
def test_filter_is_bit_identical(vg, ctx, n, dim, nq, k, metric):
    rng = np.random.default_rng(n + dim + nq + k + metric)
    base = rng.standard_normal((n, dim)).astype(np.float32)
    base[n // 3] = base[7]                      # duplicates: ties broken by row id
    base[n // 2] = base[7]
    q = rng.standard_normal((nq, dim)).astype(np.float32)
    q[3] = base[7]                              # a query that is a corpus row
    idx = vg.Index(ctx, n, dim, vg.Metric(metric))
    idx.set_vectors(base)
    ids0, sc0 = idx.search_flat(q, k)
    idx.enable_bf16_filter(True)
    ids1, sc1 = idx.search_flat(q, k)
    assert np.array_equal(ids0, ids1) and np.array_equal(bits(sc0), bits(sc1))
    for qi in (0, 3, nq - 1):
        eid, esc = o.flat_search_f32(base, dim, q[qi], k, metric)
        assert np.array_equal(ids1[qi], eid) and np.array_equal(bits(sc1[qi]), bits(esc)), qi
    idx.enable_bf16_filter(False)
    ids2, sc2 = idx.search_flat(q, k)
    assert np.array_equal(ids0, ids2) and np.array_equal(bits(sc0), bits(sc2))
    idx.close()


def test_filter_survives_values_bf16_cannot_tell_apart(vg, ctx):
    """Rows that differ only below bfloat16's 8 bits of mantissa get equal filter scores; the exact re-score must still
    order them, and clusters larger than the candidate budget must fall back to the exhaustive kernel."""
    rng = np.random.default_rng(5)
    n, dim, nq, k = 8192, 128, 96, 10
    centre = rng.standard_normal(dim).astype(np.float32)
    base = (centre + 1e-4 * rng.standard_normal((n, dim))).astype(np.float32)     # one tight cluster: all rows alike in bf16
    q = (centre + 1e-4 * rng.standard_normal((nq, dim))).astype(np.float32)
    idx = vg.Index(ctx, n, dim)
    idx.set_vectors(base)
    idx.enable_bf16_filter(True)
    ids, sc = idx.search_flat(q, k)
    for qi in range(0, nq, 13):
        eid, esc = o.flat_search_f32(base, dim, q[qi], k, 0)
        assert np.array_equal(ids[qi], eid) and np.array_equal(bits(sc[qi]), bits(esc)), qi
    searched, exhaustive = idx.flat_stats()
    assert searched == nq and exhaustive > 0          # the proof cannot pass here: the fallback did the work
    idx.close()


def test_filter_needs_rows_and_follows_them(vg, ctx):
    idx = vg.Index(ctx, 100, 64)
    with pytest.raises(vg.VecgoHipError) as e:
        idx.enable_bf16_filter(True)
    assert e.value.status == -9
    rng = np.random.default_rng(1)
    a = rng.standard_normal((100, 64)).astype(np.float32)
    b = rng.standard_normal((100, 64)).astype(np.float32)
    q = rng.standard_normal((70, 64)).astype(np.float32)
    idx.set_vectors(a)
    idx.enable_bf16_filter(True)
    idx.set_vectors(b)                                 # drops the copy of the old rows
    ids, sc = idx.search_flat(q, 5)
    eid, esc = o.flat_search_f32(b, 64, q[0], 5, 0)
    assert np.array_equal(ids[0], eid) and np.array_equal(bits(sc[0]), bits(esc))
    idx.close()
