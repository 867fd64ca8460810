"""CPU-side checks of the C-ABI shared library: it loads, exports every symbol the header declares,
and refuses to compute without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def capi():
    import __graft_entry__ as ge
    ge.build()
    from vorbispizza_amd import capi as m
    m.lib()
    return m


def header_symbols(name="vorbispizza_synth.h", prefix="vpz_"):
    text = open(os.path.join(ROOT, "include", name)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(%s[a-z0-9_]+)\s*\(" % prefix, text)))


def test_header_symbols_are_exported(capi):
    syms = header_symbols()
    assert len(syms) >= 20
    L = C.CDLL(capi.LIB_PATH)
    for s in syms:
        assert hasattr(L, s), "missing export " + s
    assert sorted(capi.EXPORTED_SYMBOLS) == syms
    # the test-only header: exported too, bound by the ctypes layer, and NOT part of the product header
    dbg = header_symbols("vorbispizza_synth_debug.h")
    assert dbg and sorted(capi.DEBUG_SYMBOLS) == dbg and not set(dbg) & set(syms)
    for s in dbg:
        assert hasattr(L, s), "missing export " + s


def test_host_library_headers_are_exported_and_plain_c(capi, tmp_path):
    """include/vorbispizza_front.h (vpzh_*) and include/vorbispizza_reader.h (vpzr_*) -- the C interfaces of
    libvorbispizza_host.so the drop-in C# reader binds: every declared symbol is exported, the
    headers compile as C99 on their own, and vpzh_info has the layout the ctypes / C# mirror assumes."""
    import shutil
    import subprocess
    from vorbispizza_amd import front
    L = front.lib()
    for header, prefix, least in (("vorbispizza_front.h", "vpzh_", 14), ("vorbispizza_reader.h", "vpzr_", 14)):
        syms = header_symbols(header, prefix)
        assert len(syms) >= least, (header, syms)
        for s in syms:
            assert hasattr(L, s), "missing export " + s
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    fields = [n for n, _ in front.Info._fields_]
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "vorbispizza_reader.h"', '#include "vorbispizza_front.h"',
             '#include "vorbispizza_synth_debug.h"', 'int main(void) {', 'printf("size %zu\\n", sizeof(vpzh_info));']
    lines += ['printf("%s %%zu\\n", offsetof(vpzh_info, %s));' % (f, f) for f in fields] + ['return 0; }']
    src = tmp_path / "host_layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "host_layout"
    subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src),
                    "-o", str(exe)], check=True, capture_output=True)
    got = dict(l.split() for l in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    assert int(got["size"]) == C.sizeof(front.Info)
    for f in fields:
        assert int(got[f]) == getattr(front.Info, f).offset, f


def test_abi_version_and_error_strings(capi):
    L = capi.lib()
    assert L.vpz_abi_version() == 6
    assert L.vpz_error_string(0) == b"ok"
    for code in range(-7, 0):
        assert L.vpz_error_string(code) not in (b"ok", b"unknown status")


def test_struct_layouts_match_header(capi):
    assert C.sizeof(capi.Packet) == 24
    assert C.sizeof(capi.Floor1Config) == 4 * (2 + 65)
    assert C.sizeof(capi.MappingConfig) == 4 + 256 + 256 + 256 + 16  # (ABI v4: residue_begin[2], residue_end[2])


def test_no_cpu_fallback(capi):
    """Without a GPU the library must fail loudly instead of computing on the host."""
    L = capi.lib()
    if L.vpz_device_count() > 0:
        pytest.skip("a GPU is visible here")
    h = C.c_void_p()
    assert L.vpz_context_create(0, C.byref(h)) == capi.E_NO_DEVICE
    assert not h.value
    with pytest.raises(capi.SynthError):
        capi.Context(0)
    x = np.zeros((1, 1024), dtype=np.float32)
    y = np.zeros((1, 2048), dtype=np.float32)
    assert L.vpz_imdct_batch(None, 2048, 1, x.ctypes.data, y.ctypes.data, 0, 0) == capi.E_INVALID_ARG


def test_product_does_not_touch_oracle():
    """The shipped package must not import, link or call anything under oracle/."""
    pkg = os.path.join(ROOT, "vorbispizza_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "orc_" not in text and "import oracle" not in text and "liboracle" not in text, f


def test_header_is_plain_c_and_struct_layouts_match_the_bindings(tmp_path):
    """include/vorbispizza_synth.h must compile as C99 (the boundary is a C ABI) and the ctypes mirrors in
    vorbispizza_amd/capi.py -- and therefore the C# LayoutKind.Sequential structs of INTEGRATION.md, which have
    the same fields in the same order -- must have the C compiler's sizes and offsets."""
    import ctypes as C
    import shutil
    import subprocess
    from vorbispizza_amd import capi
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    structs = {"vpz_floor1_config": capi.Floor1Config, "vpz_floor0_config": capi.Floor0Config,
               "vpz_mapping_config": capi.MappingConfig, "vpz_stream_config": capi.StreamConfig,
               "vpz_packet": capi.Packet}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "vorbispizza_synth.h"', 'int main(void) {']
    for cname, cls in structs.items():
        lines.append('printf("%s %%zu\\n", sizeof(%s));' % (cname, cname))
        for fname, _ in cls._fields_:
            lines.append('printf("%s.%s %%zu\\n", offsetof(%s, %s));' % (cname, fname, cname, fname))
    lines += ['return 0; }']
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I", os.path.join(root, "include"), str(src),
                    "-o", str(exe)], check=True, capture_output=True)
    got = dict(l.split() for l in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    for cname, cls in structs.items():
        assert int(got[cname]) == C.sizeof(cls), cname
        for fname, _ in cls._fields_:
            assert int(got["%s.%s" % (cname, fname)]) == getattr(cls, fname).offset, (cname, fname)
