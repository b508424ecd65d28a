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
    assert h.dvae_version() == 109
    assert h.dvae_conv_wgrad_ws_floats() > 4_000_000


def test_argument_errors_are_reported():
    # a NULL pointer is rejected before any launch (no GPU needed), message retrievable
    try:
        _lib.call("dvae_add", None, None, None, 4, None)
    except _lib.DvaeHipError as e:
        assert "invalid argument" in str(e)
    else:
        raise AssertionError("expected DvaeHipError")


def test_plan_ops_cover_the_launching_entry_points():
    """dvae_plan_op resolves every entry point the training step records; dvae_plan_run rejects malformed entries before any
    launch (no GPU needed) and runs an argument-checked call through its trampoline (the NULL-pointer error comes back)."""
    h = _lib.lib()
    not_replayable = {"dvae_version", "dvae_last_error", "dvae_conv_wgrad_ws_floats", "dvae_latent_entropy_ws_floats",
                      "dvae_reparam_kl_blocks", "dvae_fc_chain_rows", "dvae_u8_fused_supported", "dvae_plan_op", "dvae_plan_run", "dvae_comm_load",
                      "dvae_comm_unique_id", "dvae_comm_init", "dvae_comm_destroy", "dvae_comm_world", "dvae_comm_rank", "dvae_stream_create",
                      "dvae_adam_step"}        # (step count and learning rate change every iteration: issued directly)
    for name in _lib.SIGNATURES:
        assert (h.dvae_plan_op(name.encode()) >= 0) == (name not in not_replayable), name
    assert h.dvae_plan_op(b"no_such_entry_point") == -1
    arr = (_lib.PlanEntry * 1)()
    arr[0].op, arr[0].nargs = h.dvae_plan_op(b"dvae_add"), 2              # dvae_add takes 5 arguments
    assert h.dvae_plan_run(ctypes.addressof(arr), 1) != 0 and b"arguments recorded" in h.dvae_last_error()
    arr[0].nargs = 5                                                        # all-NULL pointers: dvae_add's own check fires
    assert h.dvae_plan_run(ctypes.addressof(arr), 1) != 0 and b"invalid argument" in h.dvae_last_error()
    arr[0].op = 10 ** 6
    assert h.dvae_plan_run(ctypes.addressof(arr), 1) != 0 and b"unknown op" in h.dvae_last_error()
    # the recorder packs what the trampolines unpack
    assert _lib._pack(ctypes.c_float, 1.5) == 0x3FC00000 and _lib._pack(ctypes.c_int, -1) == 2 ** 64 - 1
    assert _lib._pack(ctypes.c_void_p, None) == 0


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


def test_argument_structs_have_the_headers_layout(tmp_path):
    """The by-pointer argument blocks (dvae_*_image_desc, dvae_fc_chain_*_args) as the C header lays them out (gcc, the
    header is plain C) == the ctypes mirrors in disvae_amd/_lib.py: size, and name / offset / size of every field in order.
    A silent mismatch here would hand a kernel the wrong pointers."""
    import ctypes
    import shutil
    import subprocess
    from disvae_amd import _lib
    if shutil.which("gcc") is None:
        pytest.skip("needs gcc")
    pairs = [("dvae_conv_image_desc", _lib.ConvImageDesc), ("dvae_fc_image_desc", _lib.FcImageDesc),
             ("dvae_thin_image_desc", _lib.ThinImageDesc), ("dvae_fc_chain_fwd_args", _lib.FcChainFwdArgs),
             ("dvae_fc_chain_bwd_args", _lib.FcChainBwdArgs), ("dvae_plan_entry", _lib.PlanEntry),
             ("dvae_adam_tensor", _lib.AdamTensor)]
    body = []
    for cname, cls in pairs:
        body.append('printf("%s %%zu\\n", sizeof(%s));' % (cname, cname))
        for fname, _ in cls._fields_:
            body.append('printf("%s.%s %%zu %%zu\\n", offsetof(%s, %s), sizeof(((%s*)0)->%s));' % (cname, fname, cname, fname, cname, fname))
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "dvae_hip.h"\nint main(void) {\n' + "\n".join(body) + "\nreturn 0;\n}\n")
    exe = tmp_path / "layout"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(root, "include"), str(src), "-o", str(exe)], check=True)
    got = dict((l.split()[0], [int(x) for x in l.split()[1:]]) for l in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    for cname, cls in pairs:
        assert got[cname] == [ctypes.sizeof(cls)], (cname, got[cname], ctypes.sizeof(cls))
        for fname, ftype in cls._fields_:
            f = getattr(cls, fname)
            assert got["%s.%s" % (cname, fname)] == [f.offset, f.size], (cname, fname, got["%s.%s" % (cname, fname)], f.offset, f.size)
    # ... and the header declares no further members (sizes match and the last field ends at the padded size)
    hdr = open(os.path.join(root, "include", "dvae_hip.h")).read()
    for cname, cls in pairs:
        decl = hdr[:hdr.index("} %s;" % cname)]
        decl = decl[decl.rindex("typedef struct"):]
        decl = re.sub(r"\[[^\]]*\]", "", decl)                # array members: name[N]
        names = re.findall(r"[\*\s,]([A-Za-z_][A-Za-z_0-9]*)\s*(?=[,;])", re.sub(r"/\*.*?\*/", "", decl, flags=re.S))
        assert names == [n for n, _ in cls._fields_], (cname, names)


def test_latent_layout_macros_match_the_python_helpers(tmp_path):
    """The D-dependent buffer layouts of include/dvae_hip.h (DVAE_ROWSTATS_STRIDE, DVAE_BTCVAE_TMP_FLOATS, DVAE_NPACK_D,
    DVAE_NSCAL_D, DVAE_WIDE_KL0: latent dimensions above DVAE_MAX_D use the "wide" layouts) evaluated by gcc == the helpers of
    disvae_amd/_lib.py that size the buffers.  A mismatch would make a kernel write past a buffer the Python side allocated."""
    import shutil
    import subprocess
    from disvae_amd import _lib
    if shutil.which("gcc") is None:
        pytest.skip("needs gcc")
    dims = [1, 3, 10, 16, 17, 20, 28, 29, 32, 33, 64, 100, 1000]
    body = ['printf("max %d kl0 %d\\n", DVAE_MAX_D, DVAE_WIDE_KL0);']
    for D in dims:
        body.append('printf("%d %%d %%zu %%d %%d\\n", DVAE_ROWSTATS_STRIDE(%d), DVAE_BTCVAE_TMP_FLOATS(1024, 128, %d), '
                    'DVAE_NPACK_D(%d), DVAE_NSCAL_D(%d));' % (D, D, D, D, D))
    src = tmp_path / "macros.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "dvae_hip.h"\nint main(void) {\n' + "\n".join(body) + "\nreturn 0;\n}\n")
    exe = tmp_path / "macros"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(root, "include"), str(src), "-o", str(exe)], check=True)
    lines = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines()
    assert lines[0].split() == ["max", str(_lib.MAX_LATENT_DIM), "kl0", str(_lib.WIDE_KL0)]
    for D, line in zip(dims, lines[1:]):
        got = [int(x) for x in line.split()]
        assert got == [D, _lib.rowstats_stride(D), _lib.btcvae_tmp_floats(1024, 128, D), _lib.npack(D), _lib.nscal(D)], (D, got)
        assert _lib.rowstats_stride(D) >= 4 + D and _lib.kl0(D) + D <= _lib.nscal(D)
