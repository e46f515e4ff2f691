"""Runs the reference's own heap tests (tests/golden/reference_kats.json: searcher_priority_queue from
internal/searcher/queue_test.go, searcher_candidate_heap from candidate_queue_test.go) against ANY replay function
`replay(is_max, script) -> (out[n,3] int32 {flag, node, dist bits}, (nodes, dists))` — the oracle's heap on the CPU,
vg_heap.hpp on the GPU — plus seeded random scripts for cross-implementation comparison."""
import json
from pathlib import Path

import numpy as np

KATS = json.loads((Path(__file__).resolve().parent / "golden" / "reference_kats.json").read_text())
OPS = {"push": 0, "pop": 1, "push_bounded": 2, "try_push_bounded": 3, "top": 4, "min_item": 5, "reset": 6, "len": 7}


def f32(bits):
    return np.array(bits, np.int32).view(np.float32)


def check_priority_queue_kats(replay):
    seen = 0
    for case in KATS["searcher_priority_queue"]["cases"]:
        if "script" in case:
            out, _ = replay(case["is_max"], case["script"])
            for st, (flag, node, bits) in zip(case["script"], out.tolist()):
                if "expect_ok" in st:
                    assert bool(flag) == st["expect_ok"], (case["name"], st)
                if "expect_dist" in st:
                    assert flag == 1 and f32(bits) == np.float32(st["expect_dist"]), (case["name"], st, f32(bits))
                if "expect_len" in st:
                    assert flag == st["expect_len"], (case["name"], st)
                seen += 1
        if case["name"] == "ZeroAllocations/Reset":     # queue_test.go:141-147
            script = [(OPS["push"], i, float(i), 0) for i in range(1000)] + [(OPS["reset"], 0, 0.0, 0), (OPS["len"], 0, 0.0, 0)]
            out, (nodes, _) = replay(case["is_max"], script)
            assert out[-1, 0] == case["then"]["expect_len"] and nodes.size == 0
        if case["name"] == "Stress":                    # queue_test.go:150-167 (seeded here; the property is the KAT)
            rng = np.random.default_rng(20260401)
            d = rng.random(1000, dtype=np.float32)
            script = [(OPS["push"], i, float(d[i]), 0) for i in range(1000)] + [(OPS["pop"], 0, 0.0, 0)] * 1000
            out, (nodes, _) = replay(case["is_max"], script)
            pops = f32(out[1000:, 2])
            assert np.all(out[1000:, 0] == 1) and np.all(np.diff(pops) >= 0) and nodes.size == 0
            assert sorted(out[1000:, 1].tolist()) == list(range(1000))
    assert seen >= 30


def random_scripts(seed, n_scripts=24):
    """Seeded scripts mixing every operation, distances from small integer grids (ties) and continuous draws, both
    heap kinds; non-negative distances so that the unsigned-key sifts of the device heap apply."""
    rng = np.random.default_rng(seed)
    for s in range(n_scripts):
        is_max = bool(s & 1)
        grid = (2, 3, 6, 50, 0)[s % 5]               # 0 = continuous
        cap = int(rng.integers(1, 70))
        n = int(rng.integers(50, 900))
        script = []
        for _ in range(n):
            r = rng.random()
            d = float(rng.integers(0, grid)) if grid else float(rng.random(dtype=np.float32))
            node = int(rng.integers(0, 1 << 32, dtype=np.uint64))
            if r < 0.35:
                script.append((OPS["push"], node, d, 0))
            elif r < 0.55:
                script.append((OPS["push_bounded"], node, d, cap))
            elif r < 0.75:
                script.append((OPS["try_push_bounded"], node, d, cap))
            elif r < 0.90:
                script.append((OPS["pop"], 0, 0.0, 0))
            elif r < 0.94:
                script.append((OPS["top"], 0, 0.0, 0))
            elif r < 0.97:
                script.append((OPS["min_item"], 0, 0.0, 0))
            elif r < 0.995:
                script.append((OPS["len"], 0, 0.0, 0))
            else:
                script.append((OPS["reset"], 0, 0.0, 0))
        script += [(OPS["pop"], 0, 0.0, 0)] * 8
        yield is_max, script


def python_replay(is_max, script):
    """The same scripts through tests/prioq_py.PrioQ (a second reading of queue.go)."""
    from tests.prioq_py import PrioQ
    q = PrioQ(is_max)
    out = np.zeros((len(script), 3), np.int32)
    for i, st in enumerate(script):
        if isinstance(st, dict):
            st = (OPS[st["op"]], st.get("node", 0), st.get("dist", 0.0), st.get("cap", 0))
        op, node, dist, arg = st
        flag, res = 0, None
        if op == 0:
            q.push(node, dist); flag = 1
        elif op == 1:
            res = q.pop(); flag = int(res is not None)
        elif op == 2:
            flag = int(q.push_bounded(node, dist, arg)) if (len(q) < arg or len(q) > 0) else 0
        elif op == 3:
            flag = int(q.try_push_bounded(node, dist, arg)) if (len(q) < arg or len(q) > 0) else 0
        elif op == 4:
            res = q.top(); flag = int(res is not None)
        elif op == 5:
            res = q.min_item(); flag = int(res is not None)
        elif op == 6:
            q.reset(); flag = 1
        elif op == 7:
            flag = len(q)
        out[i, 0] = flag
        if res is not None:
            out[i, 1] = int(np.array(res[0], np.uint32).view(np.int32))
            out[i, 2] = int(np.array(res[1], np.float32).view(np.int32))
    nodes = np.array([x[0] for x in q.items], np.uint32)
    dists = np.array([x[1] for x in q.items], np.float32)
    return out, (nodes, dists)


def same(a, b):
    (oa, (na, da)), (ob, (nb, db)) = a, b
    return (np.array_equal(oa, ob) and np.array_equal(na, nb)
            and np.array_equal(np.asarray(da, np.float32).view(np.uint32), np.asarray(db, np.float32).view(np.uint32)))
