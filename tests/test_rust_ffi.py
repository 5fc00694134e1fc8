"""rust/bevy_mi355x/src/ffi.rs against include/bevy_mi355x.h.

There is no Rust toolchain in this image, so bindgen cannot regenerate the binding and rustc cannot type-check it; this test is what
keeps the committed Rust source honest.  It parses the C header and the Rust file with parsers of its own (NOT the generator's, so a
bug in tools/gen_rust_ffi.py cannot hide behind itself) and checks, item by item:
  * every `mi_*` prototype has exactly one `pub fn` of the same name, arity, parameter names, parameter types and return type;
  * every `#define MI_*` has a `pub const` of the same value;
  * every `typedef struct` has a `#[repr(C)]` struct with the same fields in the same order, and the same size as ctypes computes
    for the mirror in bevy_amd/api.py where one exists;
  * lib.rs only calls functions ffi.rs declares, with the declared number of arguments."""
import ctypes
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "bevy_mi355x.h")
FFI = os.path.join(ROOT, "rust", "bevy_mi355x", "src", "ffi.rs")
LIB = os.path.join(ROOT, "rust", "bevy_mi355x", "src", "lib.rs")

C_TO_RUST_SCALAR = {"int32_t": "i32", "uint32_t": "u32", "uint64_t": "u64", "uint8_t": "u8", "float": "f32", "double": "f64", "void": "c_void",
                    "char": "c_char"}


def camel(c_struct_name):
    return "".join(p.capitalize() for p in c_struct_name.split("_"))


def c_type_to_rust(t):
    """`const mi_view*` -> `*const MiView`; `void* const*` -> `*const *mut c_void`: a `const` left of the first `*` qualifies
    the base type, a `const` right of a `*` qualifies that pointer, and a Rust pointer is `*const` iff what it POINTS TO is const."""
    toks = re.findall(r"\*|\w+", t)
    base, pointee_const, rust = None, False, None
    for tok in toks:
        if tok == "struct":
            continue
        if tok == "const":
            pointee_const = True
        elif tok == "*":
            assert base is not None, t
            rust = ("*const " if pointee_const else "*mut ") + rust
            pointee_const = False
        else:
            assert base is None, t
            base = tok
            rust = C_TO_RUST_SCALAR.get(base) or camel(base)
    return rust


def parse_c(src):
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    funcs = {}
    for m in re.finditer(r"\b(int32_t|const\s+char\s*\*)\s*(mi_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", src):
        params = []
        body = " ".join(m.group(3).split())
        if body not in ("", "void"):
            for p in body.split(","):
                pm = re.fullmatch(r"\s*(.+?)\s*\b([A-Za-z_]\w*)\s*(\[\d*\])?\s*", p)
                ctype = pm.group(1) + ("*" if pm.group(3) else "")
                params.append((pm.group(2), c_type_to_rust(ctype)))
        funcs[m.group(2)] = ("i32" if m.group(1) == "int32_t" else "*const c_char", params)
    defines = {}
    for m in re.finditer(r"^#define[ \t]+(MI_\w+)[ \t]+(.+)$", src, flags=re.M):
        defines[m.group(1)] = m.group(2).strip()
    structs = {}
    for m in re.finditer(r"typedef\s+struct\s+(\w+)\s*\{([^}]*)\}\s*(\w+)\s*;", src):
        fields = []
        for decl in m.group(2).split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            parts = [x.strip() for x in decl.split(",")]
            head = re.fullmatch(r"(.+?)\s*\b(\w+)\s*(?:\[(\d+)\])?", parts[0])
            ctype = head.group(1)
            entries = [(head.group(2), head.group(3))]
            for more in parts[1:]:
                mm = re.fullmatch(r"(\w+)\s*(?:\[(\d+)\])?", more)
                entries.append((mm.group(1), mm.group(2)))
            for name, count in entries:
                rt = c_type_to_rust(ctype)
                fields.append((name, f"[{rt}; {count}]" if count else rt))
        structs[m.group(3)] = fields
    return funcs, defines, structs


def parse_rust(src):
    src = re.sub(r"//[^\n]*", "", src)
    ext = re.search(r'unsafe extern "C" \{(.*?)\n\}', src, flags=re.S).group(1)
    funcs = {}
    for m in re.finditer(r"pub fn (\w+)\s*\((.*?)\)\s*->\s*([^;]+);", ext, flags=re.S):
        params = []
        for p in [x.strip() for x in m.group(2).split(",") if x.strip()]:
            name, ty = p.split(":", 1)
            params.append((name.strip().removeprefix("r#"), " ".join(ty.split())))
        assert m.group(1) not in funcs, f"duplicate {m.group(1)}"
        funcs[m.group(1)] = (m.group(3).strip(), params)
    consts = {m.group(1): (m.group(2), m.group(3).strip()) for m in re.finditer(r"pub const (\w+): (\w+) = ([^;]+);", src)}
    structs = {}
    for m in re.finditer(r"#\[repr\(C\)\]\s*(?:#\[derive\([^)]*\)\]\s*)?pub struct (\w+) \{(.*?)\n\}", src, flags=re.S):
        fields = []
        for f in re.finditer(r"pub (\S+): ([^,\n]+),", m.group(2)):
            fields.append((f.group(1).removeprefix("r#"), f.group(2).strip()))
        structs[m.group(1)] = fields
    return funcs, consts, structs


@pytest.fixture(scope="module")
def both():
    return parse_c(open(HEADER).read()), parse_rust(open(FFI).read())


def test_parsers_see_the_whole_header(both):
    (c_funcs, c_defs, c_structs), _ = both
    from bevy_amd import api

    assert set(c_funcs) == set(api.ABI_SYMBOLS)  # the C parser above did not silently skip a prototype
    assert len(c_funcs) >= 60 and len(c_structs) >= 10 and len(c_defs) >= 50


def test_every_function_is_bound_with_the_same_signature(both):
    (c_funcs, _, _), (r_funcs, _, _) = both
    assert set(r_funcs) == set(c_funcs)
    for name, (ret, params) in c_funcs.items():
        r_ret, r_params = r_funcs[name]
        assert r_ret == ret, name
        assert [p for p, _ in r_params] == [p for p, _ in params], name
        assert [t for _, t in r_params] == [t for _, t in params], name


def _value(expr, table):
    expr = expr.strip("() ")
    if "|" in expr:
        v = 0
        for part in expr.split("|"):
            v |= _value(table[part.strip()] if part.strip() in table else part, table)
        return v
    expr = expr.rstrip("u")
    return int(expr, 0)


def test_every_constant_has_the_same_value(both):
    (_, c_defs, _), (_, r_consts, _) = both
    assert set(r_consts) == set(c_defs)
    r_exprs = {k: v for k, (_, v) in r_consts.items()}
    for name, expr in c_defs.items():
        ty, r_expr = r_consts[name]
        assert _value(r_expr, r_exprs) == _value(expr, c_defs), name
        unsigned = expr.strip("() ").endswith("u") or "|" in expr
        assert ty == ("u32" if unsigned else "i32"), name


def test_every_struct_has_the_same_fields(both):
    (_, _, c_structs), (_, _, r_structs) = both
    assert set(r_structs) - {"MiCtx"} == {camel(n) for n in c_structs}
    for name, fields in c_structs.items():
        assert r_structs[camel(name)] == fields, name


RUST_SIZES = {"u8": 1, "u32": 4, "i32": 4, "f32": 4, "u64": 8}


def _rust_layout(fields):
    """size/align of a #[repr(C)] struct of scalars, arrays and pointers."""
    off, align = 0, 1
    for _, ty in fields:
        am = re.fullmatch(r"\[(\w+); (\d+)\]", ty)
        if ty.startswith("*"):
            size = a = 8
        elif am:
            a = RUST_SIZES[am.group(1)]
            size = a * int(am.group(2))
        else:
            size = a = RUST_SIZES[ty]
        off = (off + a - 1) // a * a + size
        align = max(align, a)
    return (off + align - 1) // align * align


def test_struct_sizes_match_the_ctypes_mirrors(both):
    _, (_, _, r_structs) = both
    from bevy_amd import api

    mirrors = {"MiView": api.View, "MiClusterView": api.ClusterView, "MiClusterConfig": api.ClusterConfig,
               "MiClusterHistory": api.ClusterHistory, "MiClusterResolved": api.ClusterResolved, "MiFrameResults": api.FrameResults, "MiVisibleList": api.VisibleList, "MiUploadWindow": api.UploadWindow}
    for name, cls in mirrors.items():
        assert _rust_layout(r_structs[name]) == ctypes.sizeof(cls), name


def test_generator_output_is_committed():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_rust_ffi.py"), "--check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


def test_lib_rs_calls_only_declared_functions_with_declared_arity(both):
    _, (r_funcs, r_consts, _) = both
    src = re.sub(r"//[^\n]*", "", open(LIB).read())
    calls = list(re.finditer(r"ffi::(mi_\w+)\s*\(", src))
    assert len({m.group(1) for m in calls}) >= 15  # the three systems, setup and teardown really are written out
    for m in calls:
        name = m.group(1)
        assert name in r_funcs, name
        depth, i, args, cur = 1, m.end(), 0, ""
        while depth:
            ch = src[i]
            if ch in "([{":
                depth += 1
            elif ch in ")]}":
                depth -= 1
            if depth == 1 and ch == ",":
                args += 1 if cur.strip() else 0
                cur = ""
            elif depth >= 1:
                cur += ch
            i += 1
        args += 1 if cur.strip() else 0
        assert args == len(r_funcs[name][1]), f"{name}: {args} arguments in lib.rs, {len(r_funcs[name][1])} declared"
    for m in re.finditer(r"ffi::(MI_\w+)", src):
        assert m.group(1) in r_consts, m.group(1)
    for needle in ("impl Plugin for Mi355xRenderPrepPlugin", "fn mi_propagate_transforms", "fn mi_check_visibility",
                   "fn mi_assign_objects_to_clusters", "contiguous_iter", "MI_ERR_MALFORMED_HIERARCHY", "CpuFallback"):
        assert needle in src, needle


def test_integration_md_names_every_function(both):
    (c_funcs, _, _), _ = both
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    section = text[text.index("## 1. FFI declarations"):text.index("## 2. Plugins and systems")]
    missing = [f for f in c_funcs if f"`{f}`" not in section]
    assert not missing, missing
    assert f"**all {len(c_funcs)}**" in section
