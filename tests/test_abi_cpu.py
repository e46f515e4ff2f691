"""No-GPU checks of the drop-in boundary: the C-ABI library loads, exports every symbol the
public header declares, reports errors as statuses (never aborts), and refuses to run without a
gfx950 device (there is no CPU fallback)."""
import ctypes as C

import pytest

from vecgo_amd import _lib


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    declared = _lib.declared_symbols()
    assert len(declared) >= 40
    missing = [s for s in declared if not hasattr(lib, s)]
    assert not missing, missing


def test_abi_version_and_status_strings():
    lib = _lib.load()
    import re
    want = int(re.search(r"#define VG_ABI_VERSION (\d+)", _lib.HEADER_PATH.read_text()).group(1))
    assert lib.vg_abi_version() == want == _lib.ABI_VERSION == 2   # 2: vg_search_stats has five int64
    minor = int(re.search(r"#define\s+VG_ABI_MINOR\s+(\d+)", _lib.HEADER_PATH.read_text()).group(1))
    assert lib.vg_abi_minor() == minor >= 2                        # additions only: the minor says which symbols exist
    assert lib.vg_status_string(0) == b"ok"
    # the Go error strings the shim maps to (pq.go:148-153, rabitq.go:53)
    assert lib.vg_status_string(-2) == b"vector dimension mismatch"
    assert lib.vg_status_string(-3) == b"ProductQuantizer not trained"
    assert lib.vg_status_string(-4) == b"codes length mismatch"


def test_rabitq_code_bytes_is_pure():
    lib = _lib.load()
    lib.vg_rabitq_code_bytes.restype = C.c_int64
    assert lib.vg_rabitq_code_bytes(C.c_int32(768)) == 100   # rabitq.go:187-190
    assert lib.vg_rabitq_code_bytes(C.c_int32(128)) == 20
    assert lib.vg_rabitq_code_bytes(C.c_int32(100)) == 20


def test_no_device_means_error_not_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    import vecgo_amd
    with pytest.raises(vecgo_amd.VecgoHipError) as e:
        vecgo_amd.Context(0)
    assert e.value.status == -8 and "no CPU fallback" in e.value.message
    lib = _lib.load()
    # NULL handles are rejected with a status, not a crash
    assert lib.vg_search_flat(None, None, C.c_int64(1), 1, None, None, None) == -1
    assert lib.vg_pq_encode(None, None, C.c_int64(1), None, None) == -1
    assert lib.vg_ctx_destroy(None) == 0 and lib.vg_index_destroy(None) == 0


def test_product_package_does_not_import_the_oracle():
    """The oracle is test infrastructure: nothing under vecgo_amd/ may reference it."""
    import pathlib
    root = pathlib.Path(_lib.__file__).resolve().parent
    for p in list(root.rglob("*.py")) + list(root.rglob("*.hip")) + list(root.rglob("*.hpp")):
        text = p.read_text()
        assert "import oracle" not in text and "from oracle" not in text and "vg_oracle" not in text, p
