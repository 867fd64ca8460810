"""bindings/csharp/*.cs against include/*.h.

No .NET toolchain exists in this pipeline, so the C# files a maintainer adds to the reference cannot be
compiled here.  What can drift silently -- a DllImport whose name, argument count or argument kinds no longer match the C
declaration, a LayoutKind.Sequential struct whose fields are not the header's in order, a constant that is not the
`#define` -- is checked by parsing both sides.  The style being mirrored is NVorbis.Tests/Bindings/Vorbisfile.cs:43-107.
"""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INC = os.path.join(ROOT, "include")
CS = os.path.join(ROOT, "bindings", "csharp")


def strip_c_comments(text):
    return re.sub(r"/\*.*?\*/", " ", text, flags=re.S)


def strip_cs_comments(text):
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    return re.sub(r"//[^\n]*", " ", text)


# ---- kinds: what has to agree between a C parameter and its C# counterpart for the call to be ABI-correct
def c_kind(decl):
    decl = decl.strip()
    if "*" in decl:
        return "ptr"
    base = re.sub(r"\bconst\b", "", decl).split()
    base = [t for t in base if t]
    ty = base[0] if len(base) == 1 else " ".join(base[:-1])  # drop the parameter name
    return {"int": "i32", "int32_t": "i32", "int64_t": "i64", "uint64_t": "u64", "float": "f32", "void": "void",
            "uint8_t": "u8", "uint16_t": "u16", "int16_t": "i16", "double": "f64"}[ty]


def cs_kind(decl):
    decl = decl.strip()
    toks = decl.split()
    if toks[0] in ("out", "ref"):
        return "ptr"
    ty = toks[0]
    if ty.endswith("*") or (len(toks) > 1 and toks[1].startswith("*")):
        return "ptr"
    if ty in ("IntPtr", "ContextHandle", "DecoderHandle"):
        return "ptr"
    return {"int": "i32", "long": "i64", "ulong": "u64", "float": "f32", "void": "void", "byte": "u8", "short": "i16",
            "ushort": "u16", "double": "f64"}[ty]


def c_functions(header, prefix):
    text = strip_c_comments(open(os.path.join(INC, header)).read())
    text = re.sub(r"^\s*#.*$", " ", text, flags=re.M)
    out = {}
    for m in re.finditer(r"([A-Za-z_][\w \t\*]*?)\b(%s\w+)\s*\(([^;{]*?)\)\s*;" % prefix, text, flags=re.S):
        ret, name, args = m.group(1), m.group(2), " ".join(m.group(3).split())
        params = [] if args in ("void", "") else [a.strip() for a in args.split(",")]
        out[name] = (c_kind(ret.strip() + " x") if "*" not in ret else "ptr", [c_kind(p) for p in params])
    return out


def cs_imports(path):
    text = strip_cs_comments(open(path).read())
    out = {}
    pat = r"\[DllImport\((\w+)[^\]]*\)\]\s*(?:public|private|internal)?\s*static\s+extern\s+([\w\*]+)\s+(\w+)\s*\(([^;]*?)\)\s*;"
    for m in re.finditer(pat, text, flags=re.S):
        lib, ret, name, args = m.group(1), m.group(2), m.group(3), " ".join(m.group(4).split())
        params = [] if not args else [a.strip() for a in args.split(",")]
        out[name] = (lib, cs_kind(ret + " x"), [cs_kind(p) for p in params])
    return out


def test_every_dllimport_of_the_synth_binding_matches_the_header():
    c = c_functions("vorbispizza_synth.h", "vpz_")
    cs = cs_imports(os.path.join(CS, "VorbisPizzaSynth.cs"))
    assert len(c) >= 24 and "vpz_decoder_synth" in c and len(c["vpz_decoder_synth"][1]) == 15
    # the binding covers the whole product header, and nothing that is not in it
    assert sorted(cs) == sorted(c), (sorted(set(c) - set(cs)), sorted(set(cs) - set(c)))
    for name, (ret, params) in c.items():
        lib, cs_ret, cs_params = cs[name]
        assert lib == "Lib"
        assert len(cs_params) == len(params), (name, params, cs_params)
        assert cs_params == params, (name, params, cs_params)
        assert cs_ret == ret, (name, ret, cs_ret)
    text = open(os.path.join(CS, "VorbisPizzaSynth.cs")).read()
    assert re.search(r'Lib\s*=\s*"vorbispizza_synth"', text)
    assert text.count("CallingConvention.Cdecl") >= len(c)


def test_every_dllimport_of_the_reader_binding_matches_its_header():
    c = dict(c_functions("vorbispizza_reader.h", "vpzr_"))
    c.update(c_functions("vorbispizza_synth.h", "vpz_"))
    cs = cs_imports(os.path.join(CS, "GpuVorbisReader.cs"))
    reader_syms = [n for n in c if n.startswith("vpzr_")]
    assert len(reader_syms) >= 20
    assert sorted(n for n in cs if n.startswith("vpzr_")) == sorted(reader_syms)
    for name, (lib, cs_ret, cs_params) in cs.items():
        ret, params = c[name]
        assert lib == ("Host" if name.startswith("vpzr_") else "Synth"), name
        assert cs_params == params and cs_ret == ret, (name, (ret, params), (cs_ret, cs_params))
    text = open(os.path.join(CS, "GpuVorbisReader.cs")).read()
    assert re.search(r'Host\s*=\s*"vorbispizza_host"', text) and re.search(r'Synth\s*=\s*"vorbispizza_synth"', text)


# ---- structs
def c_structs(header):
    text = strip_c_comments(open(os.path.join(INC, header)).read())
    defines = {k: int(v, 0) for k, v in re.findall(r"#define\s+(\w+)\s+(\(?-?\w+\)?)\s*$", text, flags=re.M)
               if re.fullmatch(r"\(?-?(0x[0-9a-fA-F]+|\d+)\)?", v) for v in [v.strip("()")]}
    out = {}
    for m in re.finditer(r"typedef\s+struct\s+(\w+)\s*\{(.*?)\}\s*\1\s*;", text, flags=re.S):
        fields = []
        for stmt in m.group(2).split(";"):
            stmt = " ".join(stmt.split())
            if not stmt:
                continue
            # "int32_t a, b" / "const T *p" / "uint8_t x[EXPR]"
            mm = re.match(r"^(?:const\s+)?(\w+)\s*(\*?)\s*(.*)$", stmt)
            ty, star, names = mm.group(1), mm.group(2), mm.group(3)
            for n in [x.strip() for x in names.split(",")]:
                arr = re.match(r"(\w+)\[(.+)\]", n)
                if arr:
                    expr = arr.group(2)
                    for k, v in defines.items():
                        expr = re.sub(r"\b%s\b" % k, str(v), expr)
                    fields.append((arr.group(1), c_kind(ty + " x"), int(eval(expr))))
                else:
                    ptr = bool(star) or n.startswith("*")
                    fields.append((n.lstrip("*").strip(), "ptr" if ptr else c_kind(ty + " x"), 0))
        out[m.group(1)] = fields
    return out, defines


def cs_structs(path):
    text = strip_cs_comments(open(path).read())
    out = {}
    for m in re.finditer(r"\[StructLayout\(LayoutKind\.Sequential\)\]\s*public\s+struct\s+(\w+)\s*\{(.*?)\}", text, flags=re.S):
        fields = []
        for stmt in m.group(2).split(";"):
            stmt = " ".join(stmt.split())
            if not stmt:
                continue
            stmt = re.sub(r"^public\s+", "", stmt)
            fixed = stmt.startswith("fixed ")
            stmt = re.sub(r"^fixed\s+", "", stmt)
            ty, names = stmt.split(" ", 1)
            for n in [x.strip() for x in names.split(",")]:
                arr = re.match(r"(\w+)\[(\d+)\]", n)
                kind = "u8" if ty == "PacketFlags" else cs_kind(ty + " x")
                if arr:
                    assert fixed
                    fields.append((arr.group(1), kind, int(arr.group(2))))
                else:
                    fields.append((n, kind, 0))
        out[m.group(1)] = fields
    return out


def norm(name):
    return name.replace("_", "").lower()


def test_sequential_structs_have_the_headers_fields_in_order():
    c, _ = c_structs("vorbispizza_synth.h")
    cs = cs_structs(os.path.join(CS, "VorbisPizzaSynth.cs"))
    pairs = {"vpz_floor1_config": "Floor1Config", "vpz_floor0_config": "Floor0Config", "vpz_mapping_config": "MappingConfig",
             "vpz_stream_config": "StreamConfig", "vpz_packet": "Packet"}
    assert sorted(c) == sorted(pairs) and sorted(cs) == sorted(pairs.values())
    for cname, csname in pairs.items():
        cf, sf = c[cname], cs[csname]
        assert len(cf) == len(sf), (cname, cf, sf)
        for (n0, k0, a0), (n1, k1, a1) in zip(cf, sf):
            assert norm(n0) == norm(n1), (cname, n0, n1)
            assert k0 == k1 and a0 == a1, (cname, n0, (k0, a0), (k1, a1))


def test_constants_equal_the_defines():
    _, defines = c_structs("vorbispizza_synth.h")
    text = strip_cs_comments(open(os.path.join(CS, "VorbisPizzaSynth.cs")).read())
    consts = {}
    for m in re.finditer(r"public\s+const\s+int\s+([^;]+);", text):
        for part in m.group(1).split(","):
            k, v = part.split("=")
            consts[norm(k.strip())] = int(v.strip(), 0)
    enum = re.search(r"enum\s+PacketFlags\s*:\s*byte\s*\{(.*?)\}", text, flags=re.S).group(1)
    flags = {norm(k): int(v, 0) for k, v in re.findall(r"(\w+)\s*=\s*(0x[0-9a-fA-F]+|\d+)", enum)}
    seen = 0
    for name, value in defines.items():
        if not name.startswith("VPZ_"):
            continue
        key = norm(name[4:])
        if name.startswith("VPZ_PKT_"):
            assert flags[norm(name[8:])] == value, name
            seen += 1
        elif key in consts:
            assert consts[key] == value, (name, value, consts[key])
            seen += 1
        else:
            # array bounds appear as literals in the struct definitions (checked by the struct test)
            assert name in ("VPZ_MAX_FLOOR1_POSTS", "VPZ_MAX_CHANNELS", "VPZ_MAX_COUPLING"), name
    assert len(flags) == 8 and seen >= 8 + 8 + 2 + 4 + 2 + 2


def test_batch_class_passes_extents_and_reads_the_per_packet_status():
    """The glue class hands the queued extents over (ABI v3) and does not treat a window mismatch as a failed call."""
    text = strip_cs_comments(open(os.path.join(CS, "GpuSynthesisBatch.cs")).read())
    call = re.search(r"vpz_decoder_synth\((.*?)\);", text, flags=re.S).group(1)
    assert len([a for a in call.split(",")]) == 15
    assert "_residueUsed" in call and "_count * _channels" in call
    assert "vpz_decoder_last_packet_status" in text and "EWindowMismatch" not in text


def test_the_multi_device_binding_matches_its_header():
    """bindings/csharp/VorbisPizzaMulti.cs against include/vorbispizza_multi.h: every DllImport, struct and status."""
    c = c_functions("vorbispizza_multi.h", "vpzm_")
    path = os.path.join(CS, "VorbisPizzaMulti.cs")
    cs = {}
    text = strip_cs_comments(open(path).read()).replace("DispatcherHandle", "IntPtr")
    pat = r"\[DllImport\((\w+)[^\]]*\)\]\s*(?:public|private|internal)?\s*static\s+extern\s+([\w\*]+)\s+(\w+)\s*\(([^;]*?)\)\s*;"
    for m in re.finditer(pat, text, flags=re.S):
        lib, ret, name, args = m.group(1), m.group(2), m.group(3), " ".join(m.group(4).split())
        params = [] if not args else [a.strip() for a in args.split(",")]
        cs[name] = (lib, cs_kind(ret + " x"), [cs_kind(p) for p in params])
    assert sorted(cs) == sorted(c) and len(c) == 5
    for name, (ret, params) in c.items():
        lib, cs_ret, cs_params = cs[name]
        assert lib == "Host" and cs_params == params and cs_ret == ret, (name, (ret, params), (cs_ret, cs_params))
    assert re.search(r'Host\s*=\s*"vorbispizza_host"', open(path).read())
    cstructs, defines = c_structs("vorbispizza_multi.h")
    css = cs_structs_with_doubles(path)
    pairs = {"vpzm_options": "Options", "vpzm_stream_result": "StreamResult", "vpzm_stats": "Stats"}
    assert sorted(cstructs) == sorted(pairs)
    for cname, csname in pairs.items():
        cf, sf = cstructs[cname], css[csname]
        assert len(cf) == len(sf), (cname, cf, sf)
        for (n0, k0, a0), (n1, k1, a1) in zip(cf, sf):
            assert norm(n0) == norm(n1) and k0 == k1 and a0 == a1, (cname, (n0, k0, a0), (n1, k1, a1))
    consts = {}
    for m in re.finditer(r"public\s+const\s+int\s+([^;]+);", strip_cs_comments(open(path).read())):
        for part in m.group(1).split(","):
            k, v = part.split("=")
            consts[norm(k.strip())] = int(v.strip(), 0)
    for name, value in defines.items():
        if name.startswith("VPZM_"):
            assert consts[norm(name[5:])] == value, name


def cs_structs_with_doubles(path):
    return cs_structs(path)
