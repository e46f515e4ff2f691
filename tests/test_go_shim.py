"""The cgo shim under go/ is source only (no Go toolchain in the build image): check it against the C ABI it binds.
Every `C.vg_*(...)` call must name a function include/vecgo_hip.h declares and pass the declared number of
arguments; every file carries the `hip && cgo` build tag and includes the header; the status -> error map names
every vg_status.  With a Go toolchain on PATH, `go vet` runs over the tree as well."""
import re
import shutil
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
GO = sorted((ROOT / "go").rglob("*.go"))


def header_functions():
    text = (ROOT / "include" / "vecgo_hip.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    fns = {}
    for m in re.finditer(r"\b(vg_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S):
        args = " ".join(m.group(2).split())
        fns[m.group(1)] = 0 if args in ("", "void") else args.count(",") + 1
    return fns, text


def header_types():
    _, text = header_functions()
    return set(re.findall(r"\}\s*(vg_[a-z0-9_]+)\s*;", text)) | set(re.findall(r"typedef struct (vg_[a-z0-9_]+) ", text))


def c_calls(src: str):
    """(name, argument count) of every C.vg_xxx( ... ) call; arguments counted at parenthesis depth 0."""
    out = []
    for m in re.finditer(r"\bC\.(vg_[a-z0-9_]+)\(", src):
        i, depth, args, cur = m.end(), 1, 0, ""
        while depth:
            ch = src[i]
            if ch in "([{":
                depth += 1
            elif ch in ")]}":
                depth -= 1
            if depth == 1 and ch == ",":
                args += 1
                cur = ""
            elif depth:
                cur += ch
            i += 1
        if cur.strip():
            args += 1
        out.append((m.group(1), args))
    return out


def test_go_files_exist():
    names = {str(p.relative_to(ROOT / "go")) for p in GO}
    want = {"simd/kernels_hip.go", "quantization/pq_hip.go", "quantization/rabitq_hip.go", "quantization/sq8_hip.go",
            "quantization/int4_hip.go", "quantization/binary_hip.go", "quantization/opq_hip.go", "segment/resident.go",
            "kmeans/kmeans_hip.go", "hipctx/ctx.go"}
    assert want <= names, want - names


@pytest.mark.parametrize("path", GO, ids=lambda p: str(p.relative_to(ROOT / "go")))
def test_calls_match_the_header(path):
    fns, _ = header_functions()
    src = path.read_text()
    assert src.startswith("//go:build hip && cgo\n"), "build tag"
    assert '#include "vecgo_hip.h"' in src and 'import "C"' in src
    calls = [(n, a) for n, a in c_calls(src) if n not in header_types()]   # C.vg_status(x) is a conversion
    assert calls, "no C.vg_* call"
    for name, n in calls:
        assert name in fns, f"{path.name}: {name} is not declared in include/vecgo_hip.h"
        assert fns[name] == n, f"{path.name}: {name} takes {fns[name]} arguments, the call passes {n}"


def test_shim_binds_every_reference_facing_entry_point():
    fns, _ = header_functions()
    used = {n for p in GO for n, _ in c_calls(p.read_text())} - header_types()
    # not bound on purpose: profiling / debug hooks, the multi-process exchange (a Go host would bind RCCL itself
    # or use these from its own launcher), the builders' batched primitives, accessors the shim does not need
    unbound_ok = {"vg_profile_enable", "vg_profile_read", "vg_debug_set_hook", "vg_debug_heap_replay", "vg_ctx_device_info", "vg_comm_unique_id",
                  "vg_comm_create", "vg_comm_destroy", "vg_comm_info", "vg_comm_probe", "vg_comm_describe", "vg_comm_all_gather", "vg_comm_all_gather_topk",
                  "vg_merge_topk_packed", "vg_robust_prune", "vg_hnsw_select_neighbors", "vg_score_candidates",
                  "vg_index_enable_bf16_filter", "vg_index_flat_stats", "vg_index_get_hnsw_graph", "vg_hnsw_level_for_id",
                  "vg_segment_pq", "vg_segment_sq8", "vg_segment_int4", "vg_opq_pq", "vg_rabitq_code_bytes", "vg_int4_code_bytes",
                  "vg_normalize_l2", "vg_crc32c"}
    missing = set(fns) - used - unbound_ok
    assert not missing, f"declared in the header, bound nowhere under go/: {sorted(missing)}"


def test_status_map_covers_every_status():
    _, text = header_functions()
    statuses = set(re.findall(r"\b(VG_(?:OK|ERR_[A-Z_]+))\s*=", text))
    src = (ROOT / "go" / "hipctx" / "ctx.go").read_text()
    for s in statuses:
        assert f"C.{s}" in src, f"hipctx.Err does not mention {s}"
    for msg in ("vector dimension mismatch", "ProductQuantizer not trained", "codes length mismatch", "invalid code length"):
        assert msg in src


def test_stats_mirror_matches_the_struct():
    _, text = header_functions()
    body = re.search(r"typedef struct vg_search_stats \{(.*?)\} vg_search_stats;", text, flags=re.S).group(1)
    n_fields = len(re.findall(r"\b[a-z_]+(?=\s*[,;])", re.sub(r"int64_t", "", body)))
    src = (ROOT / "go" / "segment" / "resident.go").read_text()
    go_body = re.search(r"type Stats struct \{(.*?)\n\}", src, flags=re.S).group(1)
    go_fields = re.findall(r"\b([A-Z][A-Za-z]+)\b(?=[\s,]*(?:,|int64))", go_body)
    assert n_fields == len(go_fields) == 5, (n_fields, go_fields)


@pytest.mark.skipif(shutil.which("go") is None, reason="no Go toolchain in this image")
def test_go_vet():
    subprocess.run(["gofmt", "-l", "."], cwd=ROOT / "go", check=True)
