"""Build-time invariants of kernels whose correctness leans on what the compiler emits (no GPU needed: hipcc cross-
compiles gfx950 here).  int4_scan_tab_kernel retires its LDS table lookups with COUNTED waits (`s_waitcnt lgkmcnt(N)`
in inline asm: LDS operations of a wave return in order, so "at most N outstanding" names which lookups are back).  That
accounting holds only if nothing else that counts on lgkmcnt is in flight inside the lookup loop: a compiler-generated
scalar load (it returns out of order) or an extra LDS access would make a wait return too early and the kernel read a
register before its load landed — silently wrong distances.  The bit-exact GPU tests would catch the symptom; this test
catches the cause, at build time, for both instantiations."""
import re
import shutil
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.fixture(scope="module")
def sq8_asm(tmp_path_factory):
    if not Path(HIPCC).exists():
        pytest.skip("no hipcc")
    out = tmp_path_factory.mktemp("isa") / "k_sq8.s"
    src = ROOT / "vecgo_amd" / "csrc" / "k_sq8.hip"
    flags = re.search(r"^FLAGS\s*:=\s*(.*?)(?:\n\S|\Z)", (src.parent / "Makefile").read_text(), flags=re.S | re.M).group(1)
    flags = [f for f in flags.replace("\\\n", " ").split() if f not in ("-fPIC",) and not f.startswith("$(")]
    cmd = [HIPCC, "--offload-arch=gfx950", *[f for f in flags if not f.startswith("--offload-arch")],
           f"-I{ROOT / 'include'}", "--cuda-device-only", "-S", "-o", str(out), str(src)]
    subprocess.run(cmd, check=True, capture_output=True, cwd=src.parent, timeout=900)
    return out.read_text()


def _kernel(asm, mangled_prefix):
    m = re.search(rf"^({re.escape(mangled_prefix)}\w*):.*?s_endpgm", asm, flags=re.S | re.M)
    assert m, mangled_prefix
    tail = asm[m.end():m.end() + 6000]
    return m.group(0), tail


@pytest.mark.parametrize("inst", ["ILb1E", "ILb0E"])   # PRE = true / false
def test_int4_lookup_loop_has_nothing_else_on_lgkmcnt(sq8_asm, inst):
    body, meta = _kernel(sq8_asm, f"_ZN2vg20int4_scan_tab_kernel{inst}")
    assert re.search(r"; ScratchSize: 0\b", meta), "int4_scan_tab_kernel spills: scratch traffic inside the counted waits"
    lines = body.splitlines()
    start = next(i for i, ln in enumerate(lines) if "Inner Loop Header: Depth=2" in ln)
    end = next(i for i in range(start, len(lines)) if re.search(r"s_cbranch_\w+\s+\.LBB\d+_\d+", lines[i])
               and lines[i].split()[-1] == lines[start - 1 if lines[start - 1].startswith(".LBB") else start].split(":")[0])
    loop = lines[start:end + 1]
    ops = [ln.split()[0] for ln in loop if ln.startswith("\t") and not ln.strip().startswith((";", "."))]
    mem = [o for o in ops if o.startswith(("s_load", "s_buffer", "s_memtime", "s_memrealtime", "ds_", "global_", "flat_",
                                            "scratch_", "buffer_"))]
    counts = {o: mem.count(o) for o in set(mem)}
    # one 128-byte piece: 256 table lookups, the next 8 blocks' code bytes, the staging writes, the next piece's rows
    assert counts == {"ds_read_b32": 256, "ds_read_b128": 8, "ds_write_b128": 8, "global_load_dwordx4": 8}, counts
    # the counted waits themselves: per block 9, 9, 8, 8 (7 blocks) and 8, 8, 8, 0 (the last)
    waits = [int(m.group(1)) for ln in loop for m in [re.search(r"s_waitcnt lgkmcnt\((\d+)\)\s*$", ln)] if m]
    asm_waits = [w for w in waits]
    assert asm_waits.count(9) == 14 and asm_waits.count(8) >= 17, asm_waits
