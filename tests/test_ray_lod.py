"""Per-ray adaptive LOD (EXTENSION, BASELINE C5): the host build of vrc_pixel_ray_lod / vrc_build_lod_tables
(libre_amd/csrc) against the oracle's raycast_pixel_ray_lod, plus the properties that pin the
definition to the reference's per-brick rule (SelectVisibles.cpp:52-68).  CPU only; the gfx950
kernel is checked against the same oracle in test_gpu_parity.py."""
import numpy as np
import pytest

import orc
import scenes


def _parity(got, want, what):
    scenes.assert_parity(got, want, what, allow_frac=scenes.RAY_LOD_ALLOW)


def _close_to_per_brick(got, want, what):
    # against the reference's per-brick march of the same bricks: the runs start eps (1 % of a voxel)
    # inside a brick where the per-brick segment starts on its face, so every sample sits that much
    # further along the ray and a few more nearest-voxel picks differ
    mx, mean, over = orc.compare(got, want)
    assert mx <= 5 * scenes.MAX_ABS and mean <= 8 * scenes.MEAN_ABS and over <= 0.06, \
        "%s: max|d|=%.3g mean|d|=%.3g over=%.4f" % (what, mx, mean, over)


def _hierarchy(voxels=(64, 64, 64), block=16, levels=None, **kw):
    vi = orc.mem_volume_info(voxels[0], voxels[1], voxels[2], block)
    return orc.build_scene(voxels=voxels, block=block, ids=orc.all_level_ids(vi, levels), **kw)


@pytest.mark.parametrize("sse", [0.5, 1.3, 1.7, 2.5, 3.3, 6.0])
@pytest.mark.parametrize("kernel", [1, 3])
def test_harness_matches_oracle(sse, kernel):
    s = _hierarchy(viewport=(48, 40), volume="hash", spin=(0.4, 0.3))
    lod = (sse, orc.world_space_per_pixel(s))
    want, n_want = orc.oracle_render(s, ray_lod=lod)
    got, n_got, ok = orc.harness_render_ray_lod(s, lod, kernel=kernel)
    assert ok
    _parity(got, want, "sse %g kernel %d" % (sse, kernel))
    assert abs(n_got - n_want) <= 3e-4 * n_want + 16


@pytest.mark.parametrize("kernel,filter_mode,dtype", [(5, 1, "u8"), (7, 0, "u8"), (5, 1, "u16"), (7, 0, "u16"),
                                                      (9, 1, "u8"), (9, 1, "u16"), (11, 1, "u16")])  # 9 / 11: tap-packed atlas
def test_per_sample_classification_modes(kernel, filter_mode, dtype):
    s = _hierarchy(viewport=(40, 32), volume="hash", spin=(-0.7, 0.2), dtype=dtype, alpha=0.3)
    lod = (1.6, orc.world_space_per_pixel(s))
    want, n_want = orc.oracle_render(s, ray_lod=lod, filter_mode=filter_mode)
    got, n_got, _ = orc.harness_render_ray_lod(s, lod, kernel=kernel)
    _parity(got, want, "kernel %d %s" % (kernel, dtype))
    assert abs(n_got - n_want) <= 3e-4 * n_want + 16


def test_levels_are_actually_mixed_along_a_ray():
    # at sse 1.5 the front of the volume needs level 0 and the back takes level 1: fewer samples
    # than the leaves alone, more than level 1 alone
    s = _hierarchy(viewport=(48, 48), volume="hash")
    wpp = orc.world_space_per_pixel(s)
    n = {sse: orc.oracle_render(s, ray_lod=(sse, wpp))[1] for sse in (0.5, 1.5, 2.5)}
    assert n[0.5] > n[1.5] > n[2.5]
    assert n[1.5] < 0.95 * n[0.5] and n[1.5] > 1.05 * n[2.5]


def test_small_error_bound_renders_the_leaves():
    # a bound no level but the finest meets: every run is a leaf brick, and runs are exactly the
    # reference's per-brick segments -> the frame and the sample count of the leaf-only render
    s = _hierarchy(viewport=(48, 40), volume="hash", spin=(0.5, -0.2))
    leaves = orc.build_scene(voxels=(64, 64, 64), block=16, viewport=(48, 40), volume="hash", spin=(0.5, -0.2),
                             spr=s.render.samplesPerRay)
    assert leaves.render.samplesPerRay == s.render.samplesPerRay
    lod = (0.01, orc.world_space_per_pixel(s))
    want, n_want = orc.oracle_render(leaves)
    for got, n_got in (orc.oracle_render(s, ray_lod=lod), orc.harness_render_ray_lod(s, lod, kernel=1)[:2]):
        _close_to_per_brick(got, want, "leaves")
        assert abs(n_got - n_want) <= 2e-3 * n_want + 16  # a run is eps shorter than the brick segment


def test_large_error_bound_renders_the_root_with_scaled_steps():
    # everything is fine enough: the root brick alone, 2^(depth-1) times fewer samples
    s = _hierarchy(viewport=(40, 40), volume="hash")
    lod = (1e3, orc.world_space_per_pixel(s))
    _, n_root = orc.oracle_render(s, ray_lod=lod)
    _, n_leaf = orc.oracle_render(s, ray_lod=(0.01, lod[1]))
    scale = 2 ** (s.vi.depth - 1)
    assert abs(n_root * scale - n_leaf) <= 0.1 * n_leaf


def test_opacity_of_a_homogeneous_volume_does_not_depend_on_the_level():
    # the point of scaling the opacity exponent with the step: a constant volume composites to the
    # same colour whichever level it is sampled at (up to the +-1 sample a run's length rounds to)
    vol = np.full((64, 64, 64), 100, dtype=np.uint8)
    s = _hierarchy(viewport=(32, 32), volume=vol, alpha=0.3, spin=(0.3, 0.2))
    wpp = orc.world_space_per_pixel(s)
    fine, n_fine = orc.oracle_render(s, ray_lod=(0.01, wpp))
    mixed, n_mixed = orc.oracle_render(s, ray_lod=(1.5, wpp))
    root, n_root = orc.oracle_render(s, ray_lod=(1e3, wpp))
    assert n_fine > n_mixed > n_root
    assert np.abs(fine - mixed).max() < 2e-2 and np.abs(fine - root).max() < 4e-2
    # without the scaling the root would be far more transparent: plain quarter-rate sampling
    assert fine[..., 3].max() > 0.2


def test_missing_levels_fall_back_to_coarser_then_finer():
    # levels {0, 2} only: level 1 wanted -> level 2 taken (coarser first); levels {0} only with a
    # large bound -> the leaves (finer, the only ones there)
    s = _hierarchy(levels=[0, 2], viewport=(40, 32), volume="hash")
    wpp = orc.world_space_per_pixel(s)
    want, n_want = orc.oracle_render(s, ray_lod=(2.5, wpp))
    got, n_got, ok = orc.harness_render_ray_lod(s, (2.5, wpp), kernel=1)
    assert ok
    _parity(got, want, "levels 0+2")
    assert abs(n_got - n_want) <= 3e-4 * n_want + 16
    only_leaves = orc.build_scene(voxels=(64, 64, 64), block=16, viewport=(40, 32), volume="hash")
    a, n_a = orc.oracle_render(only_leaves, ray_lod=(1e3, wpp))
    b, n_b = orc.oracle_render(only_leaves)
    assert abs(n_a - n_b) <= 2e-3 * n_b + 16
    _close_to_per_brick(a, b, "leaves only")


def test_partial_hierarchy_with_holes():
    # a cut that is finer in one octant only, plus the ancestors of everything in it
    vi = orc.mem_volume_info(64, 64, 64, 16)
    ids = orc.all_level_ids(vi, [0, 1])
    ids += [orc.pack(2, x, y, z, 0) for x in range(2) for y in range(2) for z in range(2, 4)]
    s = orc.build_scene(voxels=(64, 64, 64), block=16, viewport=(40, 40), volume="hash", ids=ids, spin=(0.3, 0.6))
    lod = (0.8, orc.world_space_per_pixel(s))
    want, n_want = orc.oracle_render(s, ray_lod=lod)
    got, n_got, ok = orc.harness_render_ray_lod(s, lod, kernel=3)
    assert ok
    _parity(got, want, "partial hierarchy")
    assert abs(n_got - n_want) <= 3e-4 * n_want + 16


def test_clip_planes_and_eye_inside():
    s = _hierarchy(viewport=(40, 40), volume="hash", eye=(0.1, -0.2, 0.35), spin=(0.2, 0.1),
                   planes=[[0.0, 0.0, 1.0, 0.3], [0.6, 0.8, 0.0, 0.25]])
    lod = (1.2, orc.world_space_per_pixel(s))
    want, n_want = orc.oracle_render(s, ray_lod=lod)
    got, n_got, _ = orc.harness_render_ray_lod(s, lod, kernel=1)
    _parity(got, want, "clip planes, eye inside")
    assert abs(n_got - n_want) <= 3e-4 * n_want + 16


def test_two_bricks_of_one_level_over_a_cell_are_refused():
    vi = orc.mem_volume_info(64, 64, 64, 16)
    ids = orc.all_level_ids(vi, [2]) + [orc.pack(2, 0, 0, 0, 0)]
    s = orc.build_scene(voxels=(64, 64, 64), block=16, viewport=(16, 16), ids=ids)
    lod = (1.0, orc.world_space_per_pixel(s))
    with pytest.raises(RuntimeError):
        orc.oracle_render(s, ray_lod=lod)
    with pytest.raises(RuntimeError):
        orc.harness_render_ray_lod(s, lod)


@pytest.mark.parametrize("seed", range(24 * scenes.FUZZ_SCALE))
def test_random_views(seed):
    rng = np.random.default_rng(7000 + seed)
    vox = [int(rng.choice([32, 64, 96])) for _ in range(3)]
    block = int(rng.choice([16, 32]))
    vox = [max(v, block) // block * block for v in vox]
    kw = dict(voxels=tuple(vox), block=block, viewport=(int(rng.integers(9, 40)), int(rng.integers(9, 40))),
              volume=str(rng.choice(["hash", "mem"])),
              spin=(float(rng.uniform(-3.1, 3.1)), float(rng.uniform(-1.5, 1.5))),
              alpha=float(rng.choice([0.05, 0.3, 1.0])))
    if rng.random() < 0.3:
        kw["eye"] = (float(rng.uniform(-0.4, 0.4)), float(rng.uniform(-0.4, 0.4)), float(rng.uniform(0.1, 0.9)))
    s = _hierarchy(**kw)
    # error bounds around the value at which this view switches levels inside the volume
    lod = (float(rng.uniform(0.3, 4.0)) * vox[0] / 64.0 * 48.0 / s.H, orc.world_space_per_pixel(s))
    want, n_want = orc.oracle_render(s, threads=4, ray_lod=lod)
    for kernel in (1, 3):
        got, n_got, ok = orc.harness_render_ray_lod(s, lod, kernel=kernel)
        assert ok
        _parity(got, want, "seed %d k%d %r lod %r" % (seed, kernel, kw, lod))
        assert abs(n_got - n_want) <= 3e-4 * n_want + 16, (seed, kernel, kw)
    # the trilinear filter through the tap-packed atlas around the same walk (round 4)
    want_lin, n_lin = orc.oracle_render(s, threads=4, ray_lod=lod, filter_mode=1)
    got, n_got, ok = orc.harness_render_ray_lod(s, lod, kernel=9)
    assert ok
    _parity(got, want_lin, "seed %d tap-packed trilinear %r lod %r" % (seed, kw, lod))
    assert abs(n_got - n_lin) <= 3e-4 * n_lin + 16, (seed, kw)


# ---- ragged trees: the reference's UVF fixture (75x75x138 voxels, bricks of 28^3, two levels whose
# ---- brick grids do not align: 38 voxels at the coarse level stand for 75) -------------------------------

def _uvf_hierarchy(viewport, **kw):
    import os
    from libre_amd import driver
    driver.load_library()
    uri = "uvf://" + os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mouse_reduced.uvf")
    info = driver.datasource_info(uri)
    ids = []
    for level in range(info["depth"]):
        for x in range(info["root_blocks"][0] << level):
            for y in range(info["root_blocks"][1] << level):
                for z in range(info["root_blocks"][2] << level):
                    nid = orc.pack(level, x, y, z, 0)
                    if driver.datasource_node(uri, nid)["valid"]:
                        ids.append(nid)
    return orc.scene_from_datasource(driver, uri, ids, viewport, **kw), ids


@pytest.mark.parametrize("sse", [0.4, 1.0, 1.6, 4.0])
def test_ragged_uvf_tree(sse):
    s, ids = _uvf_hierarchy((56, 48), spin=(0.6, 0.3), alpha=0.3)
    assert len(ids) == 45 + 12
    lod = (sse, orc.world_space_per_pixel(s))
    want, n_want = orc.oracle_render(s, ray_lod=lod)
    assert want[..., 3].max() > 0.3
    for kernel in (1, 3, 5, 9):  # classified tables (float / fixed-point stepping), trilinear by gathers, tap-packed atlas
        if kernel == 5:
            want, n_want = orc.oracle_render(s, ray_lod=lod, filter_mode=1)
        got, n_got, ok = orc.harness_render_ray_lod(s, lod, kernel=kernel)
        assert ok
        _parity(got, want, "uvf sse %g kernel %d" % (sse, kernel))
        assert abs(n_got - n_want) <= 3e-4 * n_want + 16


def test_ragged_uvf_tree_level_extremes():
    # a bound only the fine level meets = the 45 leaves rendered per brick; a bound both meet = the
    # 12 coarse bricks with doubled step
    s, ids = _uvf_hierarchy((56, 48), spin=(0.6, 0.3), alpha=0.3)
    import os
    from libre_amd import driver
    uri = "uvf://" + os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mouse_reduced.uvf")
    leaves = orc.scene_from_datasource(driver, uri, [i for i in ids if (i & 0xF) == 1], (56, 48), spin=(0.6, 0.3),
                                       alpha=0.3)
    wpp = orc.world_space_per_pixel(s)
    fine, n_fine = orc.oracle_render(s, ray_lod=(0.01, wpp))
    want, n_want = orc.oracle_render(leaves)
    _close_to_per_brick(fine, want, "uvf leaves")
    assert abs(n_fine - n_want) <= 2e-3 * n_want + 16
    coarse, n_coarse = orc.oracle_render(s, ray_lod=(1e3, wpp))
    assert abs(2 * n_coarse - n_fine) <= 0.1 * n_fine


def test_oracle_still_matches_the_committed_ray_lod_frames():
    # tests/golden/frames_ray_lod.npz pins the extension's definition between rounds
    import importlib.util
    import os
    gdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    spec = importlib.util.spec_from_file_location("make_golden_ray_lod", os.path.join(gdir, "make_golden_ray_lod.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    golden = np.load(os.path.join(gdir, "frames_ray_lod.npz"))
    n_cases = 0
    for name, s, sse in gen.cases():
        lod = (sse, orc.world_space_per_pixel(s))
        fb, n = orc.oracle_render(s, threads=8, ray_lod=lod)
        assert n == int(golden[name + "__samples"][0]), name
        assert np.allclose(fb, golden[name], atol=1e-6), name
        got, n_got, ok = orc.harness_render_ray_lod(s, lod, kernel=3)
        scenes.assert_parity(got, golden[name], name + " host build vs golden", budget=orc.budget_of(fb))
        # ... and with the trilinear filter (the frames the LDS-staged kernel is held to on the GPU)
        fb, n = orc.oracle_render(s, threads=8, ray_lod=lod, filter_mode=1)
        assert n == int(golden[name + "__trilinear_samples"][0]), name
        assert np.allclose(fb, golden[name + "__trilinear"], atol=1e-6), name
        n_cases += 1
    assert n_cases == 3
