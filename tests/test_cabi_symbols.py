"""CPU: the C-ABI library loads and exports every symbol declared in include/dvae_hip.h
(no compute calls -- there is no GPU here)."""
import ctypes
import os
import re

from disvae_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "dvae_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(dvae_[a-zA-Z0-9_]+)\s*\(", txt)))


def test_header_symbols_exported():
    names = _declared()
    assert len(names) >= 24
    h = ctypes.CDLL(os.path.abspath(_lib.LIB_PATH))
    for n in names:
        assert hasattr(h, n), "missing export: " + n


def test_ctypes_table_matches_header():
    assert sorted(_lib.SIGNATURES) == _declared()
    h = _lib.lib()
    assert h.dvae_version() == 103
    assert h.dvae_conv_wgrad_ws_floats() > 4_000_000


def test_argument_errors_are_reported():
    # a NULL pointer is rejected before any launch (no GPU needed), message retrievable
    try:
        _lib.call("dvae_add", None, None, None, 4, None)
    except _lib.DvaeHipError as e:
        assert "invalid argument" in str(e)
    else:
        raise AssertionError("expected DvaeHipError")


def test_graft_entry_build_runs():
    """__graft_entry__.build() is what the driver calls on the CPU box: it must compile (no-op when the
    objects are current), load the library and agree with the header's DVAE_VERSION."""
    import importlib
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    g = importlib.import_module("__graft_entry__")
    g.build()
    hdr = open(os.path.join(root, "include", "dvae_hip.h")).read()
    assert int(re.search(r"#define DVAE_VERSION (\d+)", hdr).group(1)) == _lib.lib().dvae_version()
