#!/usr/bin/env python3
"""Name resolution of the uncompiled Rust sources against the reference checkout -- the part of `cargo check` that grep can do.

There is no rustc in this image, so `rust/bevy_mi355x/src/lib.rs` and `tools/golden_dump/src/main.rs` have never been compiled.
This tool checks what can be checked without a compiler:

  1. every leaf of every `use bevy_*::...;` tree names an item the reference crate defines (`pub struct|enum|fn|trait|type|const|
     mod NAME`), an enum variant (`Enum::Variant` -> the variant is listed in `pub enum Enum { .. }`), or comes through a
     re-export the crate declares (`pub use glam::*` in bevy_math, `pub use tracing::{..}` in bevy_log);
  2. every CamelCase identifier used in type position is imported, defined in the file, in std's prelude or in bevy_ecs's prelude
     (read from crates/bevy_ecs/src/lib.rs);
  3. every method the file calls with `.name(` exists as `fn name` somewhere in the reference, in the file itself, or in the
     std list below (a weak check: it catches misspelt and renamed methods, not a method called on the wrong type);
  4. the traits whose methods the file calls are in scope (TRAIT_METHODS below: method -> trait that must be imported or come
     with a prelude);
  5. no system function has more than 16 parameters;
  6. every struct literal / struct pattern of a reference type (`Camera { is_active: .. }`, `ClusterConfig::XYZ { dimensions, .. }`)
     names fields the reference's definition has, and all of them unless it ends in `..`;
  7. every field the file reads with `.name` is a field some struct of the reference, of ffi.rs or of the file itself declares
     (weak in the same way as 3: it catches renamed fields).

    python tools/check_rust_names.py            # checks against /root/reference, rewrites tests/golden/reference_api_names.json
    python tools/check_rust_names.py --check    # the same, fails if the fixture would change

The fixture (name -> defining file:line of the reference) travels; the reference does not.  tests/test_rust_names.py runs part 1
against the fixture everywhere and the whole tool where the reference checkout exists.
"""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/crates"
FILES = ["rust/bevy_mi355x/src/lib.rs", "rust/bevy_mi355x/src/sharded.rs", "tools/golden_dump/src/main.rs"]
FIXTURE = os.path.join(ROOT, "tests", "golden", "reference_api_names.json")

STD_TYPES = set("""Option Some None Ok Err Vec String Box Default Clone Copy PartialEq Eq Debug Send Sync Sized Drop Fn FnMut FnOnce
Iterator IntoIterator From Into Self ToString AsRef Result Ordering""".split())
STD_METHODS = set("""len iter map collect push clear is_empty unwrap unwrap_or get get_mut insert remove contains contains_key extend
as_ptr as_mut_ptr as_slice as_mut_slice iter_mut enumerate zip filter filter_map for_each min max clone clone_from copied cloned into
to_bits from_bits abs sqrt is_some is_none is_ok is_err ok err expect take replace resize reserve with_capacity entry or_insert
or_insert_with or_default sort sort_unstable sort_unstable_by_key sort_by_key retain drain truncate first last any all sum count rev
chunks chunks_exact windows swap to_vec as_ref as_mut map_or map_or_else and_then then then_some unwrap_or_default unwrap_or_else
to_string to_str to_owned into_owned as_str wrapping_add wrapping_sub saturating_sub saturating_add position find fold
copy_from_slice fill keys values values_mut set cast add offset read write is_null eq ne cmp partial_cmp powf ln exp ceil floor round
clamp to_array to_cols_array is_finite flatten flat_map skip step_by next dedup try_into try_from into_iter chain extend_from_slice
is_some_and ok_or ok_or_else to_string_lossy or back reverse length write_all flush to_le_bytes as_bytes exit args nth parse join display
create unwrap_or_else trailing_zeros and_then map_err""".split())
# glam methods the files call (bevy_math re-exports glam; glam itself is not in the checkout)
GLAM_METHODS = set("""to_cols_array to_array length from_cols_array from_array truncate extend normalize dot cross mul_vec3 transform_point3
transform_point3a transform_vector3 inverse abs max_element min_element splat from_rotation_y from_rotation_x from_rotation_z
from_axis_angle from_euler looking_at from_xyz from_scale_rotation_translation to_scale_rotation_translation as_vec3 as_uvec2 xyz""".split())
# methods of the `tracing` crate (bevy_log re-exports its macros; the crate itself is not in the checkout): `Span::entered`, as the
# reference calls it at crates/bevy_transform/src/systems.rs:592
TRACING_METHODS = {"entered", "in_scope", "instrument"}
# a method that only resolves with its trait in scope -> the trait
TRAIT_METHODS = {
    "intern": "ScheduleLabel",
    "set_visible": "SetViewVisibility",
    "run_if": "IntoScheduleConfigs",
    "in_set": "IntoScheduleConfigs",
    "before": "IntoScheduleConfigs",
    "after": "IntoScheduleConfigs",
    "last_changed": "DetectChanges",
    "is_changed": "DetectChanges",
    "is_added": "DetectChanges",
    "set_if_neq": "DetectChangesMut",
    "bypass_change_detection": "DetectChangesMut",
}
# (CameraProjection's methods: callable on `&Projection` without the trait -- it derefs to `dyn CameraProjection`, whose methods are
#  inherent to the trait object -- but on a concrete PerspectiveProjection / OrthographicProjection only with the trait imported)
CONCRETE_PROJECTIONS = ("PerspectiveProjection", "OrthographicProjection")


def strip(src):
    s = re.sub(r"//[^\n]*", "", src)
    return re.sub(r'"(\\.|[^"\\])*"', '""', s)


def use_leaves(s):
    out = []

    def expand(prefix, body):
        depth, cur, parts = 0, "", []
        for ch in body:
            depth += ch == "{"
            depth -= ch == "}"
            if ch == "," and depth == 0:
                parts.append(cur)
                cur = ""
            else:
                cur += ch
        if cur.strip():
            parts.append(cur)
        for p in parts:
            p = p.strip()
            m = re.match(r"^([\w:]*?)(?:::)?\{(.*)\}$", p, re.S)
            if m:
                expand(prefix + [x for x in m.group(1).split("::") if x], m.group(2))
            else:
                out.append(prefix + [x for x in re.sub(r"\s+as\s+\w+", "", p).split("::") if x])

    for m in re.finditer(r"\buse\s+([^;]+);", s):
        expand([], m.group(1))
    return out


def grep(pattern, where):
    r = subprocess.run(["grep", "-rnE", pattern, where, "--include=*.rs"], capture_output=True, text=True).stdout.splitlines()
    return ":".join(r[0].split(":", 2)[:2]).replace("/root/reference/", "") if r else None


def resolve_use(path):
    """file:line of the reference that defines the leaf of a `use` path, or None"""
    crate, name = path[0], path[-1]
    src = os.path.join(REF, crate, "src")
    if name in ("*", "prelude", "self"):
        return "glob"
    hit = grep(r"\bpub(\([a-z]+\))?\s+(unsafe\s+)?(struct|enum|fn|trait|type|const|mod|static)\s+" + name + r"\b", src)
    if hit:
        return hit
    if len(path) >= 3 and path[-2][0].isupper():  # Enum::Variant
        enum = path[-2]
        loc = grep(r"\bpub enum " + enum + r"\b", src)
        if loc:
            f, line = loc.rsplit(":", 1)
            body = open("/root/reference/" + f).read().split("\n")[int(line) - 1:]
            depth, seen = 0, False
            for i, ln in enumerate(body):
                if re.match(r"\s*" + name + r"\b\s*[,({=]?", ln) and seen and depth == 1:
                    return "%s:%d" % (f, int(line) + i)
                depth += ln.count("{") - ln.count("}")
                seen = seen or "{" in ln
                if seen and depth == 0:
                    break
    # re-exports of an external crate
    loc = grep(r"\bpub trait " + name + r"\b|define_label!\([^)]*\b" + name + r"\b", src)
    if loc:
        return loc
    for ext in ("glam", "tracing", "bevy_ecs_macros"):
        loc = grep(r"pub use " + ext + r"::(\*|prelude::\*|\{[^}]*\b" + name + r"\b|[\w:]*\b" + name + r"\b)", src)
        if loc:
            return loc + " (re-export of " + ext + ")"
    # tracing's macros come through a braces list that may span lines
    if crate == "bevy_log":
        text = open(os.path.join(src, "lib.rs")).read()
        m = re.search(r"pub use tracing::\{([^}]*)\}", text, re.S)
        if m and re.search(r"\b" + name + r"\b", m.group(1)):
            return "crates/bevy_log/src/lib.rs (re-export of tracing)"
    return None


def ecs_prelude():
    text = open(os.path.join(REF, "bevy_ecs", "src", "lib.rs")).read()
    m = re.search(r"pub mod prelude \{(.*?)\n\}", text, re.S)
    return set(re.findall(r"\b([A-Z][A-Za-z0-9]+)\b", m.group(1)))


def check_file(rel, names, problems):
    check_text(rel, open(os.path.join(ROOT, rel)).read(), names, problems)


def check_text(rel, text, names, problems):
    s = strip(text)
    leaves = use_leaves(s)
    for u in leaves:
        if not u or not u[0].startswith("bevy"):
            continue
        where = resolve_use(u)
        key = "::".join(u)
        if where is None:
            problems.append("%s: `use %s` names nothing the reference defines" % (rel, key))
        elif where != "glob":
            names[key] = where
    imported = set()
    for m in re.finditer(r"\buse\s+([^;]+);", s):
        imported |= set(re.findall(r"\b([A-Z][A-Za-z0-9]+)\b", m.group(1)))
    local = set(re.findall(r"\b(?:struct|enum|trait|type|const|static|union)\s+([A-Z][A-Za-z0-9_]+)", s))
    generics = set(re.findall(r"<\s*([A-Z])\s*[:>,]", s))
    globbed = ecs_prelude() if re.search(r"use bevy_ecs::\{[^;]*prelude::\*|use bevy_ecs::prelude::\*", s) else set()
    for ident in sorted(set(re.findall(r"(?<![\w:])([A-Z][A-Za-z0-9]+)\b", s))):
        if ident.isupper() or ident in imported | local | generics | globbed | STD_TYPES:
            continue
        problems.append("%s: `%s` is used but neither imported, defined here nor in a prelude" % (rel, ident))
    own = set(re.findall(r"\bfn\s+([a-z_][a-z0-9_]*)", s))
    methods = set(re.findall(r"(?<!\.)\.([a-z_][a-z0-9_]*)\s*(?:::<[^>]*>)?\(", s))
    for m in sorted(methods):
        if m in own or m in STD_METHODS or m in GLAM_METHODS or m in TRACING_METHODS:
            continue
        if not grep(r"\bfn\s+" + m + r"\b", REF):
            problems.append("%s: no `fn %s` anywhere in the reference" % (rel, m))
    # a system function takes at most 16 parameters (all_tuples!(impl_system_function, 0, 16, F), function_system.rs:950)
    for m in re.finditer(r"\bfn (\w+)(?:<[^>]*>)?\(", s):
        j, depth, angle, n, cur = m.end(), 1, 0, 0, ""
        while depth:
            c = s[j]
            depth += c in "([{"
            depth -= c in ")]}"
            angle += c == "<"
            angle -= c == ">" and s[j - 1] != "-"
            if c == "," and depth == 1 and angle == 0:
                n += bool(cur.strip())
                cur = ""
            elif depth >= 1:
                cur += c
            j += 1
        n += bool(cur.strip(" )\n"))
        params = s[m.end():j]
        if n > 16 and re.search(r"\b(Query|Res|ResMut|Commands)\b", params):
            problems.append("%s: system `%s` has %d parameters, the limit is 16 (group some into a tuple)" % (rel, m.group(1), n))
    check_struct_literals(rel, s, problems)
    check_field_names(rel, s, problems)
    for m, trait in TRAIT_METHODS.items():
        if m in methods and trait not in imported | globbed:
            problems.append("%s: `.%s()` needs the trait `%s` in scope" % (rel, m, trait))
    if methods & {"get_clip_from_view", "compute_frustum"} and imported & set(CONCRETE_PROJECTIONS) and "CameraProjection" not in imported:
        problems.append("%s: CameraProjection's methods on a concrete projection need the trait in scope" % rel)
    if re.search(r"CameraProjection::(get_clip_from_view|compute_frustum)\(", s):
        problems.append("%s: `CameraProjection::f(&Projection)` does not resolve (Projection only DEREFS to dyn CameraProjection): use method syntax" % rel)


def reference_struct_fields(name, variant=None):
    """[(file, {fields})] of `pub struct name { .. }` (or of `variant { .. }` inside `pub enum name`) in the reference"""
    hits = subprocess.run(["grep", "-rnE", r"\bpub (struct|enum) " + name + r"\b", REF, "--include=*.rs"], capture_output=True, text=True).stdout.splitlines()
    out = []
    for hit in hits:
        f, line, _ = hit.split(":", 2)
        text = "\n".join(open(f).read().split("\n")[int(line) - 1:int(line) + 400])
        m = re.search(r"\b" + variant + r"\s*\{", text) if variant else re.search(r"\{", text)
        if not m or (not variant and ";" in text[:m.start()]):
            continue
        j = k = m.end()
        depth = 1
        while depth and k < len(text):
            depth += text[k] == "{"
            depth -= text[k] == "}"
            k += 1
        body = re.sub(r"#\[[^\]]*\]", "", re.sub(r"//[^\n]*", "", text[j:k - 1]))
        out.append((f.replace("/root/reference/", ""), set(re.findall(r"(?:pub(?:\([a-z]+\))?\s+)?([a-z_][a-z0-9_]*)\s*:", body))))
    return out


def check_struct_literals(rel, s, problems):
    for m in re.finditer(r"(?<![\w:])((?:[a-z_]+::)*)([A-Z][A-Za-z0-9]+)(?:::([A-Z][A-Za-z0-9]+))?\s*\{", s):
        if re.search(r"\b(struct|enum|impl|trait|for|mod|union|fn|->)\s*$", s[max(0, m.start() - 12):m.start()]) or m.group(1).startswith("ffi::"):
            continue
        name, variant = m.group(2), m.group(3)
        j = k = m.end()
        depth = 1
        while depth:
            depth += s[k] == "{"
            depth -= s[k] == "}"
            k += 1
        parts, d, cur = [], 0, ""
        for ch in s[j:k - 1]:
            d += ch in "([{<"
            d -= ch in ")]}>"
            if ch == "," and d == 0:
                parts.append(cur)
                cur = ""
            else:
                cur += ch
        parts.append(cur)
        fields, rest, ok = [], False, True
        for part in (x.strip() for x in parts):
            if part.startswith(".."):
                rest = True
            elif part:
                fm = re.match(r"^([a-z_][a-z0-9_]*)\s*(:|$)", part)
                if not fm:
                    ok = False
                    break
                fields.append(fm.group(1))
        if not ok or not fields:
            continue  # a block, not a field list
        defs = reference_struct_fields(name, variant)
        if not defs:
            continue  # not a reference type
        line = s[:m.start()].count("\n") + 1
        label = name + ("::" + variant if variant else "")
        if not any(set(fields) <= have for _, have in defs):
            problems.append("%s:%d: `%s { .. }` names fields the reference's definition does not have: %s" % (rel, line, label, sorted(set(fields) - defs[0][1])))
        elif not rest and not any(set(fields) == have for _, have in defs):
            problems.append("%s:%d: `%s { .. }` leaves out %s and does not end in `..`" % (rel, line, label, sorted(defs[0][1] - set(fields))))


_reference_fields = None


def check_field_names(rel, s, problems):
    global _reference_fields
    if _reference_fields is None:
        out = subprocess.run(["grep", "-rhoE", r"^\s+pub(\([a-z]+\))?\s+[a-z_][a-z0-9_]*\s*:", REF, "--include=*.rs"], capture_output=True, text=True).stdout
        _reference_fields = set(re.findall(r"([a-z_][a-z0-9_]*)\s*:", out))
    ffi = open(os.path.join(ROOT, "rust", "bevy_mi355x", "src", "ffi.rs")).read()
    local = set(re.findall(r"^\s+(?:pub(?:\([a-z]+\))?\s+)?([a-z_][a-z0-9_]*)\s*:", s, re.M)) | set(re.findall(r"pub ([a-z_][a-z0-9_]*):", ffi))
    if rel.startswith("rust/bevy_mi355x/src/"):  # a module of the crate reads the pub(crate) fields its siblings declare
        crate_dir = os.path.join(ROOT, "rust", "bevy_mi355x", "src")
        for other in os.listdir(crate_dir):
            if other.endswith(".rs") and other != "ffi.rs":
                local |= set(re.findall(r"^\s+pub(?:\([a-z]+\))?\s+([a-z_][a-z0-9_]*)\s*:", open(os.path.join(crate_dir, other)).read(), re.M))
    glam = set("x y z w x_axis y_axis z_axis w_axis matrix3 translation".split())
    for f in sorted(set(re.findall(r"(?<=[\w)\]])\.([a-z_][a-z0-9_]*)\b(?!\s*(?:\(|::|!))", s))):
        if f not in _reference_fields | local | glam and not f.isdigit():
            problems.append("%s: `.%s` is read, but no struct of the reference, of ffi.rs or of the file has such a field" % (rel, f))


def main():
    if not os.path.isdir(REF):
        print("no reference checkout at %s: nothing to check against" % REF)
        return 0
    names, problems = {}, []
    for rel in FILES:
        check_file(rel, names, problems)
    for p in problems:
        print("PROBLEM", p)
    text = json.dumps(names, indent=1, sort_keys=True) + "\n"
    if "--check" in sys.argv:
        if not os.path.exists(FIXTURE) or open(FIXTURE).read() != text:
            print("PROBLEM tests/golden/reference_api_names.json is stale: run tools/check_rust_names.py")
            return 1
    else:
        open(FIXTURE, "w").write(text)
        print("%d imported names resolved -> %s" % (len(names), os.path.relpath(FIXTURE, ROOT)))
    return 1 if problems else 0


if __name__ == "__main__":
    sys.exit(main())
