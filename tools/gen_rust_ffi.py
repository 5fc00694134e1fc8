#!/usr/bin/env python
"""include/bevy_mi355x.h -> rust/bevy_mi355x/src/ffi.rs

    python tools/gen_rust_ffi.py            # rewrites ffi.rs
    python tools/gen_rust_ffi.py --check    # exits 1 if ffi.rs is out of date

Every `mi_*` prototype, `#define` constant and `typedef struct` of the C header becomes the corresponding `extern "C"` item,
`pub const` and `#[repr(C)] struct`.  tests/test_rust_ffi.py re-parses BOTH files independently (its own C and Rust parsers, its
own type table) and checks them against each other, so the binding cannot drift from the header unnoticed -- there is no Rust
toolchain in this image to do it with bindgen."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "bevy_mi355x.h")
OUT = os.path.join(ROOT, "rust", "bevy_mi355x", "src", "ffi.rs")

SCALARS = {"int32_t": "i32", "uint32_t": "u32", "uint64_t": "u64", "uint8_t": "u8", "float": "f32", "double": "f64", "void": "c_void",
           "char": "c_char", "int": "c_int"}
STRUCT_RUST = {"mi_ctx": "MiCtx", "mi_view": "MiView", "mi_cluster_view": "MiClusterView", "mi_cluster_config": "MiClusterConfig",
               "mi_cluster_history": "MiClusterHistory", "mi_cluster_resolved": "MiClusterResolved", "mi_bin_metadata": "MiBinMetadata",
               "mi_preprocess_work_item": "MiPreprocessWorkItem", "mi_indirect_parameters_metadata": "MiIndirectParametersMetadata",
               "mi_indirect_batch_set": "MiIndirectBatchSet", "mi_batch_set_record": "MiBatchSetRecord", "mi_batch_initial": "MiBatchInitial",
               "mi_batch_totals": "MiBatchTotals", "mi_sorted_item": "MiSortedItem", "mi_sorted_batch": "MiSortedBatch", "mi_unbatchable_index": "MiUnbatchableIndex",
               "mi_frame_results": "MiFrameResults", "mi_visible_list": "MiVisibleList", "mi_upload_window": "MiUploadWindow",
               "mi_hierarchy_advice": "MiHierarchyAdvice"}


def strip_comments(src):
    return re.sub(r"/\*.*?\*/", " ", src, flags=re.S)


def rust_type(ctype):
    """C parameter / field type (without the name) -> Rust."""
    t = " ".join(ctype.replace("*", " * ").split())
    toks = t.split(" ")
    out, const_next, base = None, False, None
    i = 0
    while i < len(toks):
        tok = toks[i]
        if tok == "const":
            const_next = True
        elif tok == "*":
            out = ("*const " if const_next_for_ptr else "*mut ") + out
        elif tok == "struct":
            pass
        else:
            base = tok
            out = STRUCT_RUST.get(tok) or SCALARS[tok]
            const_next_for_ptr = const_next
            const_next = False
            i += 1
            # pointer levels follow; a `const` between stars applies to the next pointer level
            while i < len(toks):
                if toks[i] == "*":
                    out = ("*const " if const_next_for_ptr else "*mut ") + out
                    const_next_for_ptr = False
                elif toks[i] == "const":
                    const_next_for_ptr = True
                i += 1
            return out
        i += 1
    return out


def parse_header(src):
    src = strip_comments(src)
    consts = re.findall(r"^#define\s+(MI_[A-Z0-9_]+)\s+(.+?)\s*$", src, flags=re.M)
    structs = []
    for m in re.finditer(r"typedef struct (\w+) \{(.*?)\}\s*(\w+);", src, flags=re.S):
        fields = []
        for decl in m.group(2).split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            # `uint32_t a, b[2], c` or `const float* x_planes`
            first, *rest = [d.strip() for d in decl.split(",")]
            fm = re.match(r"(.*?)(\w+)\s*(\[\d+\])?$", first)
            ctype = fm.group(1).strip()
            names = [(fm.group(2), fm.group(3))]
            for r in rest:
                rm = re.match(r"(\w+)\s*(\[\d+\])?$", r)
                names.append((rm.group(1), rm.group(2)))
            for name, arr in names:
                fields.append((name, ctype, int(arr[1:-1]) if arr else None))
        structs.append((m.group(3), fields))
    funcs = []
    for m in re.finditer(r"^(const char\*|int32_t)\s+(mi_\w+)\s*\((.*?)\)\s*;", src, flags=re.S | re.M):
        ret, name, args = m.group(1), m.group(2), " ".join(m.group(3).split())
        params = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                am = re.match(r"(.*?)(\w+)\s*(\[\d*\])?$", a)
                ctype, pname, arr = am.group(1).strip(), am.group(2), am.group(3)
                if arr is not None:  # `float out[16]` is a pointer in a parameter list
                    ctype += "*"
                params.append((pname, ctype))
        funcs.append((name, ret, params))
    return consts, structs, funcs


RUST_KEYWORDS = {"fn", "type", "ref", "in", "loop", "move", "box", "where", "match", "mod", "use", "impl", "self", "super"}


def field_name(n):
    return "r#" + n if n in RUST_KEYWORDS else n


def emit(consts, structs, funcs):
    o = ["//! Raw binding of `include/bevy_mi355x.h` (C ABI version 1) -- GENERATED by tools/gen_rust_ffi.py, do not edit.",
         "//! tests/test_rust_ffi.py checks every item below against the header.",
         "#![allow(non_camel_case_types, dead_code, clippy::too_many_arguments)]", "",
         "use core::ffi::{c_char, c_void};", "",
         "/// Opaque context handle (`mi_ctx`).", "#[repr(C)]", "pub struct MiCtx {", "    _private: [u8; 0],", "}", ""]
    for name, val in consts:
        val = val.strip()
        if name == "MI_ABI_VERSION" or re.fullmatch(r"\(?-?\d+\)?", val):
            o.append(f"pub const {name}: i32 = {val.strip('()')};")
        elif re.fullmatch(r"(0x[0-9A-Fa-f]+|\d+)u", val):
            o.append(f"pub const {name}: u32 = {val[:-1]};")
        elif re.fullmatch(r"\((MI_[A-Z_]+\s*\|\s*)+MI_[A-Z_]+\)", val):
            o.append(f"pub const {name}: u32 = {val.strip('()')};")
        else:
            raise SystemExit(f"cannot translate #define {name} {val}")
    o.append("")
    for name, fields in structs:
        o += ["#[repr(C)]", "#[derive(Clone, Copy, Debug)]", f"pub struct {STRUCT_RUST[name]} {{"]
        for fname, ctype, arr in fields:
            rt = rust_type(ctype)
            o.append(f"    pub {field_name(fname)}: {'[' + rt + '; ' + str(arr) + ']' if arr else rt},")
        o += ["}", ""]
    o += ['#[link(name = "bevy_mi355x")]', 'unsafe extern "C" {']
    for name, ret, params in funcs:
        ps = ", ".join(f"{field_name(p)}: {rust_type(t)}" for p, t in params)
        r = "*const c_char" if ret.startswith("const char") else "i32"
        line = f"    pub fn {name}({ps}) -> {r};"
        if len(line) > 120:
            line = f"    pub fn {name}(\n        " + ",\n        ".join(f"{field_name(p)}: {rust_type(t)}" for p, t in params) + f",\n    ) -> {r};"
        o.append(line)
    o += ["}", ""]
    return "\n".join(o)


def main():
    text = emit(*parse_header(open(HEADER).read()))
    if "--check" in sys.argv:
        ok = os.path.exists(OUT) and open(OUT).read() == text
        print("ffi.rs is up to date" if ok else "ffi.rs is OUT OF DATE: run python tools/gen_rust_ffi.py")
        sys.exit(0 if ok else 1)
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    open(OUT, "w").write(text)
    print(f"wrote {OUT}")


if __name__ == "__main__":
    main()
