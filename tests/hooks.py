"""Test hooks of the library (vg_debug_set_hook): force an alternative device path for a comparison."""


def set_hook(name: str, on) -> None:
    from vecgo_amd import _lib
    lib = _lib.load()
    _lib.check(lib.vg_debug_set_hook(name.encode(), 1 if str(on) == "1" or on is True or on == 1 else 0))
