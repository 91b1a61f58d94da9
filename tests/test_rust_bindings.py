"""bindings/rust cannot be compiled here (no rustc/cargo in the image), so its declarations are checked
structurally against include/tinyrt.h: same functions with the same parameter counts and types, same struct fields in
the same order with matching scalar types, same enum values."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = open(os.path.join(ROOT, "include", "tinyrt.h")).read()
RUST = open(os.path.join(ROOT, "bindings", "rust", "tinyrt-sys", "src", "lib.rs")).read()

C2RUST = {"float": "f32", "double": "f64", "uint32_t": "u32", "uint64_t": "u64", "int32_t": "i32", "uint8_t": "u8", "int": "c_int",
          "char": "c_char", "void": "c_void"}


def strip_comments(src):
    return re.sub(r"/\*.*?\*/", " ", src, flags=re.S)


def c_type_to_rust(ctype):
    """'const trt_vec3 *' -> '*const trt_vec3' ; 'trt_world **' -> '*mut *mut trt_world' ; 'float' -> 'f32'"""
    ctype = ctype.strip()
    const = "const" in ctype.split()
    base = [t for t in ctype.replace("*", " ").split() if t not in ("const", "struct")][0]
    stars = ctype.count("*")
    out = C2RUST.get(base, base)
    for k in range(stars):
        out = ("*const " if (const and k == 0) else "*mut ") + out
    return out


def header_functions():
    src = strip_comments(HEADER)
    fns = {}
    for m in re.finditer(r"\n\s*([A-Za-z_][\w\s\*]*?)\b(trt_\w+)\s*\(([^;{}]*?)\)\s*;", src):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        params = [] if args in ("", "void") else [a.strip() for a in args.split(",")]
        types = []
        for p in params:
            mm = re.match(r"(.*?)(\w+)$", p)
            types.append(c_type_to_rust(mm.group(1)))
        fns[name] = (None if ret == "void" else c_type_to_rust(ret), types)
    return fns


def rust_functions():
    block = re.search(r'extern "C" \{(.*?)\n\}', RUST, flags=re.S).group(1)
    fns = {}
    for m in re.finditer(r"pub fn (trt_\w+)\s*\((.*?)\)\s*(?:->\s*([^;]+))?;", block, flags=re.S):
        name, args, ret = m.group(1), m.group(2), m.group(3)
        types = [a.split(":", 1)[1].strip() for a in re.split(r",\s*(?=\w+\s*:)", args.strip()) if a.strip()]
        fns[name] = (ret.strip() if ret else None, [re.sub(r"\s+", " ", t) for t in types])
    return fns


def header_structs():
    src = strip_comments(HEADER)
    out = {}
    for m in re.finditer(r"typedef struct \{(.*?)\}\s*(trt_\w+)\s*;", src, flags=re.S):
        fields = []
        for decl in m.group(1).split(";"):
            decl = decl.strip()
            if not decl:
                continue
            ptr = re.match(r"(.*\*)\s*(\w+)$", decl)                     # 'const trt_tuning *tuning'
            if ptr:
                fields.append((ptr.group(2), c_type_to_rust(ptr.group(1))))
                continue
            ctype, names = decl.split(None, 1)
            for nm in names.split(","):
                nm = nm.strip()
                arr = re.match(r"(\w+)\[(\d+)\]", nm)
                rt = C2RUST.get(ctype, ctype)
                fields.append((arr.group(1), f"[{rt}; {arr.group(2)}]") if arr else (nm, rt))
        out[m.group(2)] = fields
    return out


def rust_structs():
    out = {}
    for m in re.finditer(r"#\[repr\(C\)\]\s*(?:#\[derive\([^)]*\)\]\s*)?pub struct (trt_\w+) \{(.*?)\n\}", RUST, flags=re.S):
        fields = [(f.group(1), f.group(2).strip()) for f in re.finditer(r"pub (\w+): ([^,\n]+),", m.group(2))]
        out[m.group(1)] = fields
    return out


def test_every_header_function_is_declared_in_rust_with_the_same_signature():
    h, r = header_functions(), rust_functions()
    assert len(h) >= 24
    assert set(h) == set(r), (sorted(set(h) - set(r)), sorted(set(r) - set(h)))
    for name in h:
        assert h[name] == r[name], (name, h[name], r[name])


def test_repr_c_structs_match_the_header_field_for_field():
    h, r = header_structs(), rust_structs()
    assert set(h) <= set(r), sorted(set(h) - set(r))
    for name, fields in h.items():
        assert r[name] == fields, (name, fields, r[name])


def test_enum_constants_carry_the_header_values():
    src = strip_comments(HEADER)
    consts = dict(re.findall(r"\b(TRT_[A-Z_0-9]+)\s*=\s*(-?\d+)", src))
    consts["TRT_ABI_VERSION"] = re.search(r"#define TRT_ABI_VERSION (\d+)", src).group(1)
    rust = dict(re.findall(r"pub const (TRT_[A-Z_0-9]+): \w+ = (-?\d+);", RUST))
    assert set(consts) == set(rust), (sorted(set(consts) ^ set(rust)))
    for k, v in consts.items():
        assert int(rust[k]) == int(v), k


def test_safe_wrapper_uses_only_declared_symbols():
    wrapper = open(os.path.join(ROOT, "bindings", "rust", "tinyrt", "src", "lib.rs")).read()
    used = set(re.findall(r"sys::(trt_\w+|TRT_\w+)", wrapper))
    declared = set(rust_functions()) | set(rust_structs()) | set(re.findall(r"pub const (TRT_\w+)", RUST))
    assert used and used <= declared, sorted(used - declared)
