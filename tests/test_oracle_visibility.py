"""The oracle's restatements around check_visibility that have no golden vector of their own: checked through properties of the
reference's rule (CPU only)."""
def test_render_layers_64_restatement():
    """orc_check_visibility_layers64: RenderLayers::intersects over the first u64 word (render_layers.rs:121-135).  With no high
    word it IS the 32-layer function; a row is visible iff it is visible through the low words or through the high words (every
    other test of the closure is independent of the layers)."""
    import numpy as np
    from bevy_amd import api, workloads as W
    import oracle_lib as O
    n = 5000
    sc = W.many_cubes(n, radius=60.0, ragged_flags=True)
    g, _ = O.sync_simple_transforms(sc["translation"], sc["rotation"], sc["scale"])
    rng = np.random.default_rng(2)
    lo = rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32) & rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32)
    hi = rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32) & rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32) & 0x0F0F0F0F
    cfv = api.perspective_clip_from_view(W.CAMERA_FOV, W.CAMERA_ASPECT, W.CAMERA_NEAR)
    frusta = np.concatenate([api.compute_frustum(cfv, W.many_cubes_camera(0, yaw=v), W.CAMERA_FAR) for v in range(3)])
    vm_lo, vm_hi = np.array([1, 0x10, 0], np.uint32), np.array([0, 0x0F000000, 0x00000F00], np.uint32)
    vv0 = np.zeros(n, np.uint8)
    a = O.check_visibility(g, sc["aabb_center"], sc["aabb_half"], sc["flags"], lo, vv0, frusta, view_masks=vm_lo)
    b = O.check_visibility_layers64(g, sc["aabb_center"], sc["aabb_half"], sc["flags"], lo, None, vv0, frusta, vm_lo, None)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    both = O.check_visibility_layers64(g, sc["aabb_center"], sc["aabb_half"], sc["flags"], lo, hi, vv0, frusta, vm_lo, vm_hi)
    only_hi = O.check_visibility(g, sc["aabb_center"], sc["aabb_half"], sc["flags"], hi, vv0, frusta, view_masks=vm_hi)
    assert np.array_equal(both[1], a[1] | only_hi[1]) and both[1][2].any() and (both[1] != a[1]).any()
