"""ctypes loader for libvecgo_hip.so (the C-ABI declared in include/vecgo_hip.h).

There is no CPU fallback: if the shared library is missing or no gfx950 device is
visible, every entry point of this package raises.
"""
from __future__ import annotations

import ctypes as C
import os
import re
from pathlib import Path

_PKG = Path(__file__).resolve().parent
# VECGO_HIP_LIB: another build of the same library (tools/build_variant.sh: kernel experiments)
LIB_PATH = Path(os.environ["VECGO_HIP_LIB"]) if os.environ.get("VECGO_HIP_LIB") else _PKG / "libvecgo_hip.so"
HEADER_PATH = _PKG.parent / "include" / "vecgo_hip.h"

# the ABI this binding was written against; tests/test_abi_cpu.py asserts it equals include/vecgo_hip.h's
# VG_ABI_VERSION, so the package does not need the source tree's include/ directory at import time
ABI_VERSION = 2

_lib = None


class VecgoHipError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"vecgo_hip status {status}: {message}")
        self.status = status
        self.message = message


def declared_symbols() -> list[str]:
    """Every function the public header declares (used by the symbol-export test)."""
    text = HEADER_PATH.read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vg_[a-z0-9_]+)\s*\(", text)))


def load() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` (hipcc --offload-arch=gfx950). vecgo_amd has no CPU fallback.")
    lib = C.CDLL(str(LIB_PATH), mode=C.RTLD_GLOBAL)
    lib.vg_last_error.restype = C.c_char_p
    lib.vg_status_string.restype = C.c_char_p
    lib.vg_status_string.argtypes = [C.c_int32]
    if lib.vg_abi_version() != ABI_VERSION:
        raise ImportError(f"{LIB_PATH} was built for ABI version {lib.vg_abi_version()}, this binding expects {ABI_VERSION}: rebuild")
    _lib = lib
    return lib


def check(status: int) -> None:
    if status != 0:
        lib = load()
        msg = lib.vg_last_error().decode() or lib.vg_status_string(status).decode()
        raise VecgoHipError(status, msg)
