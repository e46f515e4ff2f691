"""bench.py --gpus 2 without a launcher: it must start two ranks itself (never report a 1-GPU run as n_gpus = 2) and run
BASELINE configs[4]'s code — `rabitq_sharded` (row shards + one all-gather of per-shard top-k, engine/search.go:835-908's
fan-out/merge) and `pq_train_sharded` (sub-quantizer ranges, pq.go:83-138) — here with gloo, two ranks sharing GPU 0 and
reduced sizes (the line says so).  No scaling curve comes out of this: it proves the N > 1 code runs."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]


@pytest.mark.parametrize("world", [2, 8])   # 8: the node's size — 12 sub-quantizers per rank, an 8-list merge, eight writers of one stdout
def test_bench_gpus2_self_launch_gloo(world):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", VECGO_BENCH_ROWS="200000", VECGO_BENCH_SCAN_ROWS="200000")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, str(ROOT / "bench.py"), "--gpus", str(world), "--backend", "gloo", "--steps", "3", "--warmup", "1"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = r.stdout.splitlines()
    assert len(lines) == 1 and lines[0].startswith("{"), r.stdout[-3000:]   # nothing but the line (Gloo's own prints go to stderr)
    d = json.loads(lines[0])
    assert len(lines[0]) < 8192
    assert d["n_gpus"] == world and d["exchange"]["world_size"] == world and d["exchange"]["torch_backend"] == "gloo"
    assert d["reduced_sizes"] is True and "multi_gpu_legs_error" not in d
    assert d["recall_at_10"] == 1.0                   # the sharded exact path against the fp64 ground truth
    names = {c["config"]: c for c in d["configs"]}
    assert names["configs[4] rabitq sharded (strong)"]["qps"] > 0
    assert names["configs[4] rabitq sharded (weak)"]["rows_per_gpu"] == 200000
    assert names["configs[4] pq kmeans train, sharded by sub-quantizer"]["codebooks_identical_on_all_ranks"] is True
    rep = names["metric pipeline over query-sharded replicas (structured corpus)"]   # graph search = replicas
    assert rep["replicas"] == world and rep["ids_equal_single_gpu"] is True and rep["qps"] > 0 and rep["recall_at_10"] > 0.5
    assert "not measured" in d["scaling_curve"]

