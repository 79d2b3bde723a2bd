"""Dispatch switches of libvtx.so (csrc/options.h, include/vtx.h: vtx_set_option / vtx_get_option).

The library reads each switch once per call (atomic table, initialised from the VTX_* environment variables at load):
a test or a benchmark can flip a kernel variant in-process -- ``with options.override(GLDS_BM=64): ...`` -- and compare
the variants bit for bit.  Names = the environment variables without the ``VTX_`` prefix.
"""
import contextlib

from . import _lib


_IDS = None


def _ids():
    """name -> id of the loaded library's switches (one library per process: built once; get() sits on per-launch paths)."""
    global _IDS
    if _IDS is None:
        lib = _lib.load()
        _IDS = {lib.vtx_option_name(i).decode()[4:]: i for i in range(lib.vtx_option_count())}
    return _IDS


def names():
    return sorted(_ids())


def get(name):
    return _lib.load().vtx_get_option(_ids()[name])


def set(name, value):
    _lib.check(_lib.load().vtx_set_option(_ids()[name], int(value)), f"vtx_set_option({name})")


@contextlib.contextmanager
def override(**kw):
    """Temporarily set switches, e.g. ``override(GLDS_BM=128, GLDS_WAVES=4)``; restored on exit."""
    old = {k: get(k) for k in kw}
    try:
        for k, v in kw.items():
            set(k, v)
        yield
    finally:
        for k, v in old.items():
            set(k, v)
