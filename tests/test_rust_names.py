"""The uncompiled Rust sources against the reference's API names (tools/check_rust_names.py).

No rustc here, so this is the part of `cargo check` grep can do: imported paths, CamelCase identifiers, method names and the traits
their methods need.  Where the reference checkout exists the whole tool runs; everywhere (GPU box included) the `use` leaves of the
two files are checked against the committed fixture of names the tool resolved, so an import nobody resolved cannot be added
unnoticed."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import check_rust_names as crn  # noqa: E402


def test_every_bevy_import_is_a_name_the_tool_resolved():
    known = json.load(open(crn.FIXTURE))
    for rel in crn.FILES:
        s = crn.strip(open(os.path.join(ROOT, rel)).read())
        for u in crn.use_leaves(s):
            if u and u[0].startswith("bevy") and u[-1] not in ("*", "prelude", "self"):
                assert "::".join(u) in known, "%s imports %s: run tools/check_rust_names.py where /root/reference exists" % (rel, "::".join(u))


def test_fixture_cites_reference_lines():
    known = json.load(open(crn.FIXTURE))
    assert len(known) >= 60
    for name, where in known.items():
        assert where.startswith("crates/" + name.split("::")[0] + "/src/"), (name, where)


@pytest.mark.skipif(not os.path.isdir(crn.REF), reason="no reference checkout on this box")
def test_names_resolve_against_the_reference_checkout():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_rust_names.py"), "--check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


@pytest.mark.skipif(not os.path.isdir(crn.REF), reason="no reference checkout on this box")
@pytest.mark.parametrize("snippet, expected", [
    # the five kinds of error the round-4 desk check found in lib.rs, each as the checker must report it
    ("use bevy_ecs::schedule::RemoveSystemsOnly;\n", "names nothing the reference defines"),
    ("use bevy_app::PostUpdate;\nfn f() { let s = PostUpdate.intern(); }\n", "needs the trait `ScheduleLabel` in scope"),
    ("fn f(p: &u32) { let c = bevy_camera::CameraProjection::get_clip_from_view(p); }\n", "use method syntax"),
    ("use bevy_ecs::prelude::*;\nfn sys(" + ", ".join(f"q{i}: Query<()>" for i in range(17)) + ") {}\n", "has 17 parameters"),
    ("use bevy_camera::PerspectiveProjection;\nfn f(p: PerspectiveProjection) { p.get_clip_from_view(); }\n", "need the trait in scope"),
    ("fn f(c: bevy_camera::Camera) { c.physical_viewport_sizes(); }\n", "no `fn physical_viewport_sizes`"),
    ("fn f() -> Option<Frustum> { None }\n", "`Frustum` is used but neither imported"),
])
def test_the_checker_reports_what_it_is_there_for(snippet, expected):
    names, problems = {}, []
    crn.check_text("snippet.rs", snippet, names, problems)
    assert any(expected in p for p in problems), problems


@pytest.mark.skipif(not os.path.isdir(crn.REF), reason="no reference checkout on this box")
def test_the_checker_accepts_what_compiles():
    ok = ("use bevy_ecs::{prelude::*, schedule::{ScheduleCleanupPolicy::RemoveSystemsOnly, ScheduleLabel}};\nuse bevy_app::{App, PostUpdate};\n"
          "use bevy_camera::Projection;\n"
          "fn f(app: &mut App, p: &Projection) { let s = PostUpdate.intern(); let m = p.get_clip_from_view(); let _ = (s, m, RemoveSystemsOnly); }\n")
    names, problems = {}, []
    crn.check_text("snippet.rs", ok, names, problems)
    assert not problems, problems
    assert "bevy_ecs::schedule::ScheduleCleanupPolicy::RemoveSystemsOnly" in names


@pytest.mark.skipif(not os.path.isdir(crn.REF), reason="no reference checkout on this box")
@pytest.mark.parametrize("snippet, expected", [
    ("use bevy_camera::Camera;\nfn f() { let c = Camera { is_actve: true, ..Default::default() }; }\n", "names fields the reference's definition does not have: ['is_actve']"),
    ("use bevy_light::cluster::ClusterZConfig;\nfn f(d: f32) { let z = ClusterZConfig { first_slice_depth: d }; }\n", "leaves out ['far_z_mode']"),
    ("use bevy_light::cluster::ClusterConfig;\nfn f(c: ClusterConfig) { if let ClusterConfig::XYZ { dimensions, z_config } = c {} }\n", "leaves out ['dynamic_resizing']"),
    ("use bevy_camera::Camera;\nfn f(c: &Camera) -> bool { c.computed.is_actve }\n", "`.is_actve` is read"),
])
def test_the_checker_knows_the_reference_structs(snippet, expected):
    names, problems = {}, []
    crn.check_text("snippet.rs", snippet, names, problems)
    assert any(expected in p for p in problems), problems
