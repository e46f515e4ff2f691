"""bench.py's stdout contract: ONE JSON line the driver can parse.  r03's line was 22.7 KB and the driver
recorded `parsed: null`; the line is now a compact function of the full record (bench.compact_line), which goes to
bench_full.json.  Checked here on the committed full records of earlier rounds and on this round's captured line."""
import json
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline")


def _check(line: str):
    assert "\n" not in line and len(line) < 8192, len(line)
    d = json.loads(line)
    for k in CONTRACT:
        assert k in d, k
    assert d["metric"] == json.loads((ROOT / "BASELINE.json").read_text())["metric"]
    assert isinstance(d["roofline"]["frac"], float) and 0 < d["roofline"]["frac"] <= 1
    assert d["roofline"]["bound"] in ("hbm", "mfma") and "traffic" in d["roofline"]
    assert d["cpu_baseline"]["value"] > 0 and d["cpu_baseline"]["cores"] >= 1 and d["cpu_baseline"]["kind"] in ("reference", "port")
    assert "usable_cpus" in d["cpu_baseline"] and "sample" in d["cpu_baseline"]
    assert "workload" in d["config"] and "operating_point" in d["config"] and "model" not in d["config"]
    for c in d["configs"]:
        assert "config" in c and not any(isinstance(v, (dict, list)) for v in c.values()), c   # flat summaries only
    return d


@pytest.mark.parametrize("record", ["r03_bench_line.json", "r02_bench_line.json"])
def test_compact_line_of_committed_full_records(record):
    import bench
    full = json.loads((ROOT / "profiles" / record).read_text())
    full["metric"] = bench.BASELINE_METRIC          # earlier rounds reworded the metric
    line = json.dumps(bench.compact_line(full), ensure_ascii=True, separators=(",", ":"))
    assert len(line) < bench.LINE_LIMIT, len(line)
    d = _check(line)
    assert d["value"] == pytest.approx(full["value"], rel=1e-5)
    assert d["roofline"]["frac"] == pytest.approx(full["roofline"]["frac"], rel=1e-5)
    names = " ".join(c["config"] for c in d["configs"])
    for want in ("configs[1]", "configs[3]", "configs[4]"):
        assert want in names


def test_compact_line_survives_failed_legs():
    import bench
    full = json.loads((ROOT / "profiles" / "r03_bench_line.json").read_text())
    full["metric"] = bench.BASELINE_METRIC
    for k in ("adc_scan", "int4_scan", "structured_corpus", "flat_small_batch"):
        full[k] = {"error": "RuntimeError: " + "x" * 5000}
    del full["hnsw_pq"], full["hnsw_layer0"]
    line = json.dumps(bench.compact_line(full), separators=(",", ":"))
    _check(line)
    assert len(line) < bench.LINE_LIMIT


def _check_fractions(d):
    """Every configs[] row that quotes a fraction quotes it against the PHYSICAL bound of the kernel that runs:
    a named bound and 0 < frac <= 1; ratios against arithmetic the kernel does not execute live under other keys."""
    for c in d["configs"]:
        if "frac" in c:
            assert c.get("bound"), c
            assert 0 < c["frac"] <= 1, c
            assert "valu_reference" not in str(c["bound"]), c


def test_captured_line_of_this_round():
    """The line bench.py printed on the GPU box this round (copied from gpurun_out/ into profiles/)."""
    cap = sorted((ROOT / "profiles").glob("r0[4-9]_bench_stdout.txt"))
    if not cap:
        pytest.skip("no captured line committed yet this round")
    lines = [ln for ln in cap[-1].read_text().splitlines() if ln.strip()]
    assert len(lines) == 1, "bench.py must print exactly one stdout line"
    d = _check(lines[0])
    if cap[-1].name >= "r06":
        _check_fractions(d)


def test_every_fraction_is_against_a_physical_bound():
    """compact_line over the newest committed full record: rows built by THIS bench.py carry bound + 0 < frac <= 1."""
    import bench
    full_files = sorted((ROOT / "profiles").glob("r0[6-9]_bench_full.json"))
    if not full_files:
        pytest.skip("no full record of this round committed yet")
    full = json.loads(full_files[-1].read_text())
    _check_fractions(bench.compact_line(full))


def _run_bench(args, **env_over):
    import os
    import subprocess
    import sys
    env = dict(os.environ, **env_over)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        if k not in env_over:
            env.pop(k, None)
    return subprocess.run([sys.executable, str(ROOT / "bench.py")] + args, cwd=ROOT, env=env, capture_output=True, text=True,
                          timeout=600)


def test_bench_refuses_world_mismatch():
    """WORLD_SIZE = 1 with --gpus 2 (a launcher misconfiguration) is an error, not a silent 1-GPU line (r03:
    `assert world == args.gpus or world == 1` let exactly that through)."""
    r = _run_bench(["--gpus", "2"], WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr and not r.stdout.strip()


def test_bench_gpus_n_launches_ranks_itself():
    """No launcher in the environment: --gpus 2 starts two ranks as a child process.  Without a GPU the ranks exit with
    "needs an MI355X" and the parent relays the failure — there is no CPU fallback and no 1-GPU line."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("CPU-side check; the GPU box runs tests/test_gpu_bench_2rank.py")
    r = _run_bench(["--gpus", "2", "--backend", "gloo", "--steps", "1", "--warmup", "0"])
    assert r.returncode != 0 and not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert "torch.distributed.run" in r.stderr and "--nproc-per-node=2" in r.stderr
    assert r.stderr.count("needs an MI355X") >= 1


def test_stdout_carries_only_the_line_whatever_libraries_print():
    """Gloo ("[Gloo] Rank 1 is connected to 7 peer ranks"), RCCL banners and stray prints go to descriptor 1 of a rank
    process; bench.guard_stdout points that at stderr and keeps the real stdout for the ONE line (seen with 8 gloo ranks
    in r04: seven such lines in front of the JSON)."""
    import subprocess
    import sys
    code = ("import os, sys, json, bench\n"
            "bench.guard_stdout()\n"
            "os.write(1, b'[Gloo] Rank 1 is connected to 7 peer ranks\\n')\n"
            "print('a stray print')\n"
            "os.system('echo a child process writing to stdout')\n"
            "full = json.loads(open(sys.argv[1]).read())\n"
            "bench.FULL_RECORD = os.devnull\n"
            "bench.emit(full)\n")
    full = str(ROOT / "profiles" / "r04_bench_full.json")
    r = subprocess.run([sys.executable, "-c", code, full], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = r.stdout.splitlines()
    assert len(lines) == 1 and json.loads(lines[0])["metric"], r.stdout[:500]
    assert "[Gloo]" in r.stderr and "a stray print" in r.stderr and "child process" in r.stderr


def test_design_table_is_the_committed_record():
    """DESIGN.md section 8's current-state table is generated from profiles/r06_bench_full.json (tools/design_table.py): the text
    between its markers must be what the generator writes from the committed record — a number typed by hand, or a record
    refreshed without the table, fails here."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("design_table", ROOT / "tools" / "design_table.py")
    dt = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(dt)
    text = (ROOT / "DESIGN.md").read_text()
    a, b = text.index(dt.BEGIN) + len(dt.BEGIN), text.index(dt.END)
    assert text[a:b].strip() == dt.table("r06").strip()
    assert text[a:b].count("\n") >= 20            # every kernel family has its row
