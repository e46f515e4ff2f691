"""The C++ host-side mirror of the reference's Go interfaces (include/vecgo_hip.hpp): its test
program restates the reference's quantizer / kmeans / distance / flat-segment tests in C++.  And the C ABI from plain
C99 (tests/cpp/abi_c_test.c) — the language cgo compiles include/vecgo_hip.h as: the calls the Go shim makes, checked
against the reference's known answers."""
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
EXE = ROOT / "tests" / "cpp" / "host_mirror_test"


def _build():
    if not EXE.exists():
        import __graft_entry__ as g
        g.build()


def test_cpp_mirror_refuses_to_run_without_a_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    _build()
    r = subprocess.run([str(EXE)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 77, r.stdout + r.stderr     # VG_ERR_NO_DEVICE surfaced as vecgo::Error
    assert "no CPU fallback" in r.stdout


@pytest.mark.gpu
def test_cpp_mirror_passes_reference_style_checks():
    _build()
    r = subprocess.run([str(EXE)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "all checks passed" in r.stdout


C_EXE = ROOT / "tests" / "cpp" / "abi_c_test"


def _build_c():
    if not C_EXE.exists():
        import __graft_entry__ as g
        g.build()


def test_c99_host_refuses_to_run_without_a_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    _build_c()
    r = subprocess.run([str(C_EXE)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 77 and "no CPU fallback" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
def test_c99_host_passes_reference_known_answers():
    _build_c()
    r = subprocess.run([str(C_EXE)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "all checks passed" in r.stdout
