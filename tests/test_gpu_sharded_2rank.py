"""Two ranks (gloo) sharing GPU 0: the row-sharded searches with the real HIP local scorers and vg_merge_topk_packed
equal the single-index search (tests/sharded_2rank_worker.py).  The N > 1 path on the hardware a 1-GPU box has; the
RCCL all-gather itself needs two GPUs and is covered by world = 1 (tests/test_gpu_comm.py) and by bench.py's
cross-check of the two exchange paths before it times anything."""
import os
import socket
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]


def test_two_ranks_on_one_gpu():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(ROOT / "tests" / "sharded_2rank_worker.py")]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "OK 7" in r.stdout, r.stdout[-2000:]
