"""The cgo shim under go/ is source only (no Go toolchain in the build image): check it against the C ABI it binds.
Every `C.vg_*(...)` call must name a function include/vecgo_hip.h declares, pass the declared number of
arguments AND pass arguments of the declared C types, position by position (the cgo rules: `C.int32_t(x)` for an
int32_t, `*C.float` for a `const float *`, unsafe.Pointer for `void *`, nil for any pointer, an untyped constant for a
number) — a `C.int32_t` where the header has `int64_t` is a `go build` error this test reports without a Go
toolchain; every file carries the `hip && cgo` build tag and includes the header; the status -> error map names
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


# ---- type-level check ----------------------------------------------------------------------------------------
SCALARS = {"int32_t", "int64_t", "uint32_t", "uint64_t", "uint8_t", "int8_t", "float", "double", "size_t"}


def header_param_types():
    """name -> [normalised C type per parameter]: `const float *q` -> `float*`, `vg_pq **out` -> `vg_pq**`"""
    _, text = header_functions()
    out = {}
    for m in re.finditer(r"\b(vg_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S):
        args = " ".join(m.group(2).split())
        types = []
        if args not in ("", "void"):
            for a in args.split(","):
                a = re.sub(r"\bconst\b", "", a).strip()
                t = re.match(r"^(.*?)([A-Za-z_][A-Za-z0-9_]*)$", a)
                assert t, a
                types.append(re.sub(r"\s+", "", t.group(1)))
        out[m.group(1)] = types
    return out


def _go_type(t: str):
    """`*C.float` -> `float*`, `C.int32_t` -> `int32_t`, `unsafe.Pointer` -> `void*`; None for non-C types"""
    t = t.strip()
    if t == "unsafe.Pointer":
        return "void*"
    m = re.match(r"^(\**)C\.([A-Za-z_][A-Za-z0-9_]*)$", t)
    return (m.group(2) + m.group(1)) if m else None


def _split_args(src, i):
    depth, cur, args = 1, "", []
    while depth:
        ch = src[i]
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if depth == 1 and ch == ",":
            args.append(cur.strip())
            cur = ""
        elif depth:
            cur += ch
        i += 1
    if cur.strip():
        args.append(cur.strip())
    return args, i


class GoPackage:
    """What the checker needs to know about one directory of the shim: helper functions that return C types, struct
    fields of C types, untyped numeric constants."""

    def __init__(self, sources):
        self.helpers, self.fields, self.consts, self.globals = {}, {}, set(), {}
        for src in sources:
            for m in re.finditer(r"^var (\w+)\s+(\**C\.\w+|unsafe\.Pointer)\b", src, flags=re.M):      # var hipCtx *C.vg_ctx
                self.globals[m.group(1)] = _go_type(m.group(2))
            for m in re.finditer(r"^var \((.*?)^\)", src, flags=re.S | re.M):                          # var ( ctx *C.vg_ctx ... )
                for line in m.group(1).splitlines():
                    vm = re.match(r"^\s*(\w+)\s+(\**C\.\w+|unsafe\.Pointer)\s*(//.*)?$", line)
                    if vm:
                        self.globals[vm.group(1)] = _go_type(vm.group(2))
            for m in re.finditer(r"^func (\w+)\([^)]*\)\s*(\**C\.\w+|unsafe\.Pointer)\s*\{", src, flags=re.M):
                self.helpers[m.group(1)] = _go_type(m.group(2))
            for m in re.finditer(r"^type (\w+) struct \{(.*?)^\}", src, flags=re.S | re.M):
                f = self.fields.setdefault(m.group(1), {})
                for line in m.group(2).splitlines():
                    fm = re.match(r"^\s*([\w\s,]+?)\s+(\**C\.\w+|unsafe\.Pointer)\s*(//.*)?$", line)
                    if fm:
                        for name in fm.group(1).split(","):
                            f[name.strip()] = _go_type(fm.group(2))
            for m in re.finditer(r"^\s*(?:const\s+)?(\w+)\s*(?:\w+\s*)?=\s*-?[0-9][0-9a-fx_.]*\s*(//.*)?$", src, flags=re.M):
                self.consts.add(m.group(1))


def _function_chunks(src):
    """(header line, body text, offset) per top-level func"""
    starts = [m.start() for m in re.finditer(r"^func ", src, flags=re.M)] + [len(src)]
    return [(src[a:b], a) for a, b in zip(starts, starts[1:])]


def _env(chunk, pkg):
    """local name -> type inside one function: receiver and parameters, `var x *C.T`, `x := <typed expression>`"""
    env, structs = dict(pkg.globals), {}
    head = chunk.split("{", 1)[0]
    for m in re.finditer(r"(\w+)\s+\*?(\w+)\b", head):
        if m.group(2) in pkg.fields:
            structs[m.group(1)] = m.group(2)
    for m in re.finditer(r"\bvar\s+([\w\s,]+?)\s+(\**C\.\w+|unsafe\.Pointer)\b", chunk):
        for name in m.group(1).split(","):
            env[name.strip()] = _go_type(m.group(2))      # a local shadows a package-level name
    for m in re.finditer(r"\b(\w+)\s*:=\s*&(\w+)\{", chunk):          # r := &Resident{...}
        if m.group(2) in pkg.fields:
            structs[m.group(1)] = m.group(2)
    for m in re.finditer(r"\b(\w+)\s*:=\s*([^\n]+)", chunk):
        t = _expr_type(m.group(2).strip(), env, structs, pkg, quiet=True)
        if t and t not in ("nil", "untyped"):
            env.setdefault(m.group(1), t)
    for m in re.finditer(r"\b(\w+)\s*=\s*(\(\*+C\.\w+\)\([^\n]+)", chunk):   # sp = (*C.T)(...)
        t = _expr_type(m.group(2).strip(), env, structs, pkg, quiet=True)
        if t:
            env.setdefault(m.group(1), t)
    return env, structs


def _expr_type(e, env, structs, pkg, quiet=False):
    e = e.strip()
    if e == "nil":
        return "nil"
    if re.match(r"^-?[0-9][0-9a-fx_.]*$", e) or e in pkg.consts:
        return "untyped"
    m = re.match(r"^C\.(\w+)\(", e)
    if m:
        return m.group(1)
    m = re.match(r"^\((\*+)C\.(\w+)\)\(", e)
    if m:
        return m.group(2) + m.group(1)
    if e.startswith("unsafe.Pointer("):
        return "void*"
    if e.startswith("&"):
        t = _expr_type(e[1:], env, structs, pkg, quiet)
        return t + "*" if t and t not in ("nil", "untyped") else None
    m = re.match(r"^(\w+)\(", e)
    if m and m.group(1) in pkg.helpers:
        return pkg.helpers[m.group(1)]
    m = re.match(r"^(\w+)\.(\w+)$", e)
    if m and m.group(1) in structs:
        return pkg.fields[structs[m.group(1)]].get(m.group(2))
    if re.match(r"^\w+$", e) and e in env:
        return env[e]
    return None


def _compatible(param, arg):
    if arg == "nil":
        return param.endswith("*")
    if arg == "untyped":
        return param in SCALARS
    return param == arg


def type_errors(sources_by_path):
    """[(path, function, argument index, expression, got, want)] over every C.vg_* call of one package"""
    protos = header_param_types()
    types = header_types()
    pkg = GoPackage(list(sources_by_path.values()))
    errs = []
    for path, src in sources_by_path.items():
        for chunk, _ in _function_chunks(src):
            env, structs = _env(chunk, pkg)
            for m in re.finditer(r"\bC\.(vg_[a-z0-9_]+)\(", chunk):
                name = m.group(1)
                if name in types or name not in protos:
                    continue
                args, _ = _split_args(chunk, m.end())
                for i, (a, want) in enumerate(zip(args, protos[name])):
                    got = _expr_type(a, env, structs, pkg)
                    if got is None or not _compatible(want, got):
                        errs.append((path, name, i, a, got, want))
    return errs


def _packages():
    by_dir = {}
    for p in GO:
        by_dir.setdefault(p.parent, {})[str(p.relative_to(ROOT / "go"))] = p.read_text()
    return by_dir


@pytest.mark.parametrize("pkg_dir", sorted(_packages()), ids=lambda d: d.name)
def test_argument_types_match_the_header(pkg_dir):
    errs = type_errors(_packages()[pkg_dir])
    assert not errs, "\n".join(f"{p}: {fn} argument {i} `{a}` is {got}, the header wants {want}" for p, fn, i, a, got, want in errs)


def test_type_check_catches_a_wrong_cast():
    """the done-criterion of the check: scratch copies with one deliberately wrong cast each are reported"""
    pk = _packages()[ROOT / "go" / "segment"]
    src = pk["segment/resident.go"]
    line = "C.vg_search_hnsw(r.h, fp(queries), C.int64_t(nq), C.int32_t(k), C.int32_t(ef), up(ids), fp(sc), sp, nil)"
    assert line in src
    for bad, (idx, got, want) in {
        line.replace("C.int64_t(nq)", "C.int32_t(nq)"): (2, "int32_t", "int64_t"),          # narrower integer
        line.replace("up(ids)", "fp(ids)"): (5, "float*", "uint32_t*"),                       # wrong pointee
        line.replace("C.int32_t(k)", "C.int(k)"): (3, "int", "int32_t"),                      # C.int is not int32_t to cgo
        line.replace("fp(queries)", "unsafe.Pointer(&queries[0])"): (1, "void*", "float*"),   # untyped pointer
        line.replace("r.h", "r.seg"): (0, "vg_segment*", "vg_index*"),                        # wrong handle
        line.replace("sp, nil", "sp, 0"): (8, "untyped", "void*"),                            # number for a pointer
    }.items():
        errs = type_errors({**pk, "segment/resident.go": src.replace(line, bad)})
        assert [(e[1], e[2], e[4], e[5]) for e in errs] == [("vg_search_hnsw", idx, got, want)], (bad, errs)


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


# ---- lexical checks a `go build` would fail on (no toolchain here) -----------------------------------------------
def _strip_go(src: str) -> str:
    """source with comments, string / rune / raw-string literals and the cgo preamble blanked out"""
    out, i, n = [], 0, len(src)
    while i < n:
        two = src[i:i + 2]
        if two == "//":
            j = src.find("\n", i)
            i = n if j < 0 else j
        elif two == "/*":
            j = src.find("*/", i + 2)
            assert j >= 0, "unterminated comment"
            out.append("\n" * src.count("\n", i, j))
            i = j + 2
        elif src[i] in "\"'`":
            q = src[i]
            j = i + 1
            while j < n and src[j] != q:
                j += 2 if (src[j] == "\\" and q != "`") else 1
            assert j < n, "unterminated literal"
            out.append(q + q)
            i = j + 1
        else:
            out.append(src[i])
            i += 1
    return "".join(out)


@pytest.mark.parametrize("path", GO, ids=lambda p: str(p.relative_to(ROOT / "go")))
def test_go_file_is_lexically_sound(path):
    raw = path.read_text()
    src = _strip_go(raw)
    # balanced delimiters
    stack, pairs = [], {")": "(", "]": "[", "}": "{"}
    for ln, line in enumerate(src.splitlines(), 1):
        for ch in line:
            if ch in "([{":
                stack.append((ch, ln))
            elif ch in ")]}":
                assert stack and stack[-1][0] == pairs[ch], f"{path.name}:{ln}: unbalanced {ch!r}"
                stack.pop()
    assert not stack, f"{path.name}: unclosed {stack[-1]}"
    # one package clause, the directory's name
    pk = re.findall(r"^package (\w+)$", src, flags=re.M)
    assert pk == [path.parent.name], pk
    # every import is used (an unused import is a compile error in Go); "C" is the cgo pseudo-package
    imports = []
    for m in re.finditer(r'^import\s+(?:(\w+)\s+)?"([^"]+)"$', raw, flags=re.M):
        imports.append((m.group(1), m.group(2)))
    for blk in re.finditer(r"^import \((.*?)^\)", raw, flags=re.S | re.M):
        for m in re.finditer(r'^\s*(?:(\w+)\s+)?"([^"]+)"', blk.group(1), flags=re.M):
            imports.append((m.group(1), m.group(2)))
    assert ("", "C") in [(a or "", p_) for a, p_ in imports]
    for alias, ipath in imports:
        name = alias or ipath.rsplit("/", 1)[-1]
        if name in ("C", "_"):
            continue
        assert re.search(rf"\b{re.escape(name)}\.", src), f"{path.name}: import {ipath!r} is not used"
    # every function has a body or is a method expression; no `:=` at package level
    for m in re.finditer(r"^func [^\n]*$", src, flags=re.M):
        assert m.group(0).rstrip().endswith(("{", "}")) or "{" in m.group(0), f"{path.name}: {m.group(0)!r} has no body"
