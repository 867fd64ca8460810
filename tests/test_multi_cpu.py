"""include/vorbispizza_multi.h (the in-process multi-device dispatcher): plain C, every declared symbol exported by
libvorbispizza_host.so, struct layouts of the ctypes mirror (and of the C# structs, which have the same fields in the
same order) as the C compiler has them.  No compute here: the dispatcher's GPU tests are tests/test_multi_gpu.py."""
import ctypes as C
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_declared_symbol_is_exported():
    from vorbispizza_amd import multi
    L = multi.lib()
    header = open(os.path.join(ROOT, "include", "vorbispizza_multi.h")).read()
    declared = sorted(set(re.findall(r"\b(vpzm_[a-z_]+)\s*\(", header)))
    assert declared == sorted(multi.EXPORTED_SYMBOLS)
    for name in declared:
        assert hasattr(L, name), name


def test_header_is_plain_c_and_layouts_match(tmp_path):
    from vorbispizza_amd import multi
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    structs = {"vpzm_options": multi.Options, "vpzm_stream_result": multi.StreamResult, "vpzm_stats": multi.Stats}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "vorbispizza_multi.h"', 'int main(void) {']
    for cname, cls in structs.items():
        lines.append('printf("%s %%zu\\n", sizeof(%s));' % (cname, cname))
        for fname, _ in cls._fields_:
            lines.append('printf("%s.%s %%zu\\n", offsetof(%s, %s));' % (cname, fname, cname, fname))
    lines += ['return 0; }']
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)],
                   check=True, capture_output=True)
    got = dict(l.split() for l in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    for cname, cls in structs.items():
        assert int(got[cname]) == C.sizeof(cls), cname
        for fname, _ in cls._fields_:
            assert int(got["%s.%s" % (cname, fname)]) == getattr(cls, fname).offset, (cname, fname)


def test_bad_arguments_are_refused_without_a_device():
    from vorbispizza_amd import multi
    L = multi.lib()
    h = C.c_void_p()
    assert L.vpzm_create(None, 1, None, C.byref(h)) == multi.E_ARG
    ids = (C.c_int32 * 1)(0)
    assert L.vpzm_create(ids, 0, None, C.byref(h)) == multi.E_ARG
    assert L.vpzm_create(ids, 1, None, None) == multi.E_ARG
    assert L.vpzm_device_count(None) == 0
    L.vpzm_destroy(None)
