"""The C ABI is called from many OS threads on the same handles (SURVEY.md §8b "Threading":
read-only search concurrent, the engine fans out one goroutine per segment).  Four threads, each
on its own HIP stream, hammer one index; every result must equal the single-threaded one."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def vg():
    import vecgo_amd
    return vecgo_amd


def test_concurrent_searches_on_one_index(vg):
    ctx = vg.Context(0)
    n, dim, m, k = 60000, 128, 16, 10
    g = torch.Generator(device="cuda"); g.manual_seed(3)
    base = torch.randn(n, dim, device="cuda", generator=g)
    rng = np.random.default_rng(0)
    pq = vg.ProductQuantizer(ctx, dim, m, 256)
    pq.set_codebooks(rng.integers(-128, 128, m * 256 * (dim // m)).astype(np.int8),
                     (rng.random(m) * 0.02 + 0.005).astype(np.float32), np.zeros(m, np.float32))
    codes = pq.encode(base)
    idx = vg.Index(ctx, n, dim)
    idx.set_vectors(base); idx.set_pq_codes(pq, codes)
    torch.cuda.synchronize()
    batches = [torch.randn(nq, dim, device="cuda", generator=g) for nq in (3, 40, 200, 7, 130, 64, 1, 300)]
    want = [(idx.search_flat(q, k), idx.search_pq_adc(q, k)) for q in batches]
    torch.cuda.synchronize()
    errors = []

    def worker(t):
        try:
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                for rep in range(6):
                    for j in range(t, len(batches), 2):
                        q = batches[j]
                        (fi, fs), (ai, as_) = want[j]
                        gi, gs = idx.search_flat(q, k, stream=s)
                        pi, ps = idx.search_pq_adc(q, k, stream=s)
                        s.synchronize()
                        if not (torch.equal(gi, fi) and torch.equal(gs.view(torch.int32), fs.view(torch.int32))
                                and torch.equal(pi, ai) and torch.equal(ps.view(torch.int32), as_.view(torch.int32))):
                            errors.append((t, rep, j))
        except Exception as e:  # noqa: BLE001
            errors.append((t, repr(e)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors[:5]
