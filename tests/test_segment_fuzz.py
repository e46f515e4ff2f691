"""The segment-image parsers under AddressSanitizer + UBSan with seeded mutations.  vg_segment_open_flat /
_diskann read untrusted file images; everything they read before a section goes to the device lives in
vecgo_amd/csrc/vg_segment_layout.hpp (host only), which tests/cpp/segment_fuzz.cpp compiles with plain g++
-fsanitize=address,undefined: 10^5 mutations of tests/segfile.py's images (bytes, bits, header fields set to extremes and to the
image size +- 16, truncations); every accepted parse must declare sections inside the image, and every byte of them is read from a
heap block of exactly the image's size.  Formats: internal/segment/diskann/format.go:49-79, internal/segment/flat/format.go.  No GPU."""
import shutil
import subprocess
from pathlib import Path

import numpy as np
import pytest

from tests import segfile

ROOT = Path(__file__).resolve().parents[1]


def _images(tmp):
    rng = np.random.default_rng(5)
    n, dim = 40, 16
    x = rng.standard_normal((n, dim)).astype(np.float32)
    g = rng.integers(0, n, (n, 8)).astype(np.uint32)
    m, k = 4, 256
    pq = (m, k, rng.random(m).astype(np.float32), rng.random(m).astype(np.float32), rng.integers(-128, 128, m * k * (dim // m)).astype(np.int8))
    imgs = {
        "flat_plain": segfile.write_flat(x),
        "flat_sq8": segfile.write_flat(x, sq=(x.min(0), x.max(0)), codes=rng.integers(0, 256, (n, dim)).astype(np.uint8)),
        "flat_pq": segfile.write_flat(x, pq=pq, codes=rng.integers(0, 256, (n, m)).astype(np.uint8)),
        "flat_parts": segfile.write_flat(x, partitions=(x[:4], np.array([0, 10, 20, 30, 40], np.uint32))),
        "flat_empty": segfile.write_flat(x[:0]),
        "disk_plain": segfile.write_diskann(x, g, 3),
        "disk_pq": segfile.write_diskann(x, g, 3, pq=pq, pq_codes=rng.integers(0, 256, (n, m)).astype(np.uint8)),
        "disk_rabitq": segfile.write_diskann(x, g, 3, rabitq_codes=rng.integers(0, 256, (n, ((dim + 63) // 64) * 8 + 4)).astype(np.uint8)),
        "disk_int4": segfile.write_diskann(x, g, 3, int4=(x.min(0), x.max(0) - x.min(0), rng.integers(0, 256, (n, (dim + 1) // 2)).astype(np.uint8))),
    }
    paths = []
    for name, data in imgs.items():
        p = tmp / (name + ".seg")
        p.write_bytes(data)
        paths.append(str(p))
    return paths


@pytest.mark.skipif(shutil.which("g++") is None, reason="no g++")
def test_segment_parsers_under_asan_ubsan_with_mutations(tmp_path):
    exe = tmp_path / "segment_fuzz"
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer",
           "-I", str(ROOT / "include"), "-I", str(ROOT / "vecgo_amd" / "csrc"), str(ROOT / "tests" / "cpp" / "segment_fuzz.cpp"), "-o", str(exe)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    paths = _images(tmp_path)
    r = subprocess.run([str(exe), "100000", "20261005"] + paths, capture_output=True, text=True, timeout=600,
                       env={"ASAN_OPTIONS": "detect_leaks=1:abort_on_error=1", "UBSAN_OPTIONS": "print_stacktrace=1"})
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    out = r.stdout.strip().splitlines()[-1]
    import re
    mo = re.search(r"(\d+) parses accepted \((\d+) of unmodified images\), (\d+) rejected", out)
    assert mo, out
    accepted, pristine, rejected = (int(x) for x in mo.groups())
    assert pristine == 2 * len(paths)              # every unmodified image parses, with and without the checksum test
    assert accepted > 10000 and rejected > 10000   # the mutations reach both sides of the checks
