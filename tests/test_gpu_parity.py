"""Parity tests proper: the gfx950 path driven through the C ABI (ctypes) against the CPU
oracle and the committed golden frames.  Run on the GPU box with `pytest -m gpu`."""
import ctypes as C
import os

import numpy as np
import pytest

import orc
import scenes

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "frames.npz")


@pytest.fixture(scope="module")
def vrc():
    from libre_amd import vrc as v
    v.load_library()  # fails loudly when the HIP extension is missing
    return v


@pytest.fixture(scope="module")
def golden():
    return np.load(GOLDEN)


def _gpu(s):
    from gpu_run import GpuScene
    return GpuScene(s)


def test_oracle_still_matches_golden(golden):
    for name in sorted(scenes.SCENES):
        fb, n = orc.oracle_render(scenes.get(name), threads=8)
        assert n == int(golden[name + "__samples"][0])
        assert np.allclose(fb, golden[name], atol=1e-6), name


def test_atlas_layout_roundtrip(vrc):
    # upload through the micro-block repack, read back logical regions: bit-exact
    s = scenes.get("hash64_spin")
    with _gpu(s) as g:
        info = g.info()
        assert info["atlas_dim"] == s.atlas_dim and info["slots"] == s.slots
        for nid in s.ids:
            assert g.slots[nid] == s.slot_of[nid]
        out = np.zeros_like(s.atlas)
        vrc.check(g.L, g.L.vrc_pool_read_region(g.pool, vrc.u32x3(0, 0, 0), vrc.u32x3(*s.atlas_dim),
                                                out.ctypes.data))
        assert (out == s.atlas).all()
        # a ragged sub-region that straddles micro-blocks and slots
        o, sz = (13, 5, 3), (29, 17, 11)
        sub = np.zeros((sz[2], sz[1], sz[0]), dtype=np.uint8)
        vrc.check(g.L, g.L.vrc_pool_read_region(g.pool, vrc.u32x3(*o), vrc.u32x3(*sz), sub.ctypes.data))
        assert (sub == s.atlas[o[2]:o[2] + sz[2], o[1]:o[1] + sz[1], o[0]:o[0] + sz[0]]).all()


@pytest.mark.parametrize("name", sorted(scenes.SCENES))
def test_scene_parity_both_kernels(vrc, golden, name):
    s = scenes.get(name)
    want, n_want = orc.oracle_render(s, threads=8)
    # the oracle's "every brick-entry tie the other way" frame, for the bias check (tests/scenes.py)
    flipped, _ = orc.oracle_render(s, threads=8, entry_bias=orc.ENTRY_BIAS)
    with _gpu(s) as g:
        ref, n_ref, st = g.render(kernel=vrc.KERNEL_REFERENCE_ORDER)
        assert st.kernel_variant == vrc.KERNEL_REFERENCE_ORDER
        scenes.assert_parity(ref, want, name + " ref-order vs oracle")
        scenes.assert_no_tie_bias(ref, want, flipped, name + " ref-order")
        scenes.assert_parity(ref, golden[name], name + " ref-order vs golden", budget=orc.budget_of(want))
        assert abs(n_ref - n_want) <= 1e-4 * n_want + 8
        dda, n_dda, st = g.render(kernel=vrc.KERNEL_GRID_DDA)
        assert st.kernel_variant == vrc.KERNEL_GRID_DDA
        scenes.assert_parity(dda, want, name + " dda vs oracle")
        scenes.assert_no_tie_bias(dda, want, flipped, name + " dda")
        # the grid walk composites the reference's samples, one for one (opaque noise scene: a ray may cross
        # the early-exit threshold a sample sooner or later where a tie falls the other way)
        assert n_dda == n_want or (name == "hash64_ert" and abs(n_dda - n_want) <= 1e-4 * n_want)
        auto, _, st = g.render(kernel=vrc.KERNEL_AUTO, count=False)
        assert st.kernel_variant == vrc.KERNEL_GRID_DDA
        used = C.c_int64(-1)  # the same answer without the synchronisation of vrc_get_stats
        vrc.check(g.L, g.L.vrc_get_option(g.ctx, vrc.OPT_KERNEL_USED, C.byref(used)))
        assert used.value == vrc.KERNEL_GRID_DDA
        assert g.L.vrc_set_option(g.ctx, vrc.OPT_KERNEL_USED, 1) != 0  # read-only
        # counting samples must not change a single bit of the frame
        assert (auto == dda).all()
        # idempotence: same inputs, same bits
        again, _, _ = g.render(kernel=vrc.KERNEL_GRID_DDA)
        assert (again == dda).all()
        # the reference's float position accumulation (VRC_OPT_STEPPING = 0) is kept and tested
        flt, n_flt, _ = g.render(kernel=vrc.KERNEL_GRID_DDA, stepping=0)
        scenes.assert_parity(flt, want, name + " dda float stepping vs oracle")
        scenes.assert_no_tie_bias(flt, want, flipped, name + " dda float stepping")
        assert n_flt == n_dda


@pytest.mark.parametrize("name", sorted(scenes.SCENES))
def test_trilinear_extension_parity(vrc, name):
    # VRC_OPT_FILTER = 1 against the oracle's fetch_trilinear (extension; the reference is
    # point-sampled)
    s = scenes.get(name)
    want, n_want = orc.oracle_render(s, threads=8, filter_mode=1)
    with _gpu(s) as g:
        for k in (vrc.KERNEL_REFERENCE_ORDER, vrc.KERNEL_GRID_DDA):
            got, n_got, st = g.render(kernel=k, filter_mode=vrc.FILTER_TRILINEAR)
            assert st.kernel_variant == k
            scenes.assert_parity(got, want, name + " trilinear k%d" % k)
            assert abs(n_got - n_want) <= 2e-4 * n_want + 8
        # switching back rebuilds the classified table
        near, _, _ = g.render(kernel=vrc.KERNEL_GRID_DDA)
        scenes.assert_parity(near, orc.oracle_render(s, threads=8)[0], name + " nearest after trilinear")


@pytest.mark.parametrize("name", sorted(scenes.SCENES))
def test_lds_staged_kernel_parity(vrc, name):
    # VRC_KERNEL_LDS: same sample sequence as the gather kernel with fixed-point stepping,
    # voxels read from LDS-staged boxes; point-sampled and trilinear
    s = scenes.get(name)
    want, n_want = orc.oracle_render(s, threads=8)
    want_lin, n_want_lin = orc.oracle_render(s, threads=8, filter_mode=1)
    with _gpu(s) as g:
        got, n_got, st = g.render(kernel=vrc.KERNEL_LDS)
        assert st.kernel_variant == vrc.KERNEL_LDS
        scenes.assert_parity(got, want, name + " lds nearest")
        assert abs(n_got - n_want) <= 2e-4 * n_want + 8
        dda, n_dda, _ = g.render(kernel=vrc.KERNEL_GRID_DDA)
        assert n_got == n_dda  # identical sample sequence ...
        assert np.abs(got - dda).max() <= 1e-6  # ... and (up to blend order: none) identical frame
        lin, n_lin, st = g.render(kernel=vrc.KERNEL_LDS, filter_mode=vrc.FILTER_TRILINEAR)
        assert st.kernel_variant == vrc.KERNEL_LDS
        scenes.assert_parity(lin, want_lin, name + " lds trilinear")
        assert abs(n_lin - n_want_lin) <= 2e-4 * n_want_lin + 8
        # AUTO + trilinear: the tap-packed atlas where the bricks have an overlap (the staged form's frame, bit for
        # bit), else -- or with VRC_OPT_PACKED_ATLAS off -- the staged form itself
        auto, _, st = g.render(filter_mode=vrc.FILTER_TRILINEAR, count=False)
        assert st.kernel_variant == vrc.KERNEL_PACKED
        assert (auto == lin).all()
        vrc.check(g.L, g.L.vrc_set_option(g.ctx, vrc.OPT_PACKED_ATLAS, 0))
        auto, _, st = g.render(filter_mode=vrc.FILTER_TRILINEAR, count=False)
        assert st.kernel_variant == vrc.KERNEL_LDS
        assert (auto == lin).all()


def test_the_judged_kernel_runs_five_workgroups_per_cu(vrc):
    # The grey table form in groups of 14 (what bench.py times) needs 80 registers: six waves per SIMD would fit and
    # six thrash the L1.  Rounds 2-3 capped it with 20 KiB of LDS the kernel never touched; now the kernel says
    # amdgpu_waves_per_eu( 5, 5 ) and the runtime's own occupancy calculator -- no profiler -- must agree: five
    # workgroups of four waves per compute unit.  Frames are what they were: the parity tests.
    # (a frame of more than 6144 tiles: smaller launches take the latency-bound instance with 24 samples in flight)
    s = orc.build_scene(voxels=(64, 64, 64), block=16, viewport=(704, 640), volume="hash", spin=(0.5, 0.35))
    with _gpu(s) as g:
        g.render(count=False)
        assert b"<true,false,false,true,3,unsigned char,14,false>" in g.L.vrc_last_kernel(), g.L.vrc_last_kernel()
        wgs, threads = C.c_int(), C.c_int()
        vrc.check(g.L, g.L.vrc_last_kernel_occupancy(C.byref(wgs), C.byref(threads)))
        assert (wgs.value, threads.value) == (5, 256), (wgs.value, threads.value)
        # the four-float form: five by its registers
        vrc.check(g.L, g.L.vrc_set_option(g.ctx, vrc.OPT_GREY_TABLE, 0))
        g.render(count=False)
        assert b",0,unsigned char,8,false>" in g.L.vrc_last_kernel()
        vrc.check(g.L, g.L.vrc_last_kernel_occupancy(C.byref(wgs), C.byref(threads)))
        assert wgs.value == 5
        # the LDS-staged kernel does not report
        g.render(kernel=vrc.KERNEL_LDS, count=False)
        assert g.L.vrc_last_kernel_occupancy(C.byref(wgs), C.byref(threads)) == vrc.VRC_EINVAL


@pytest.mark.parametrize("name", sorted(scenes.SCENES))
def test_tap_packed_trilinear_parity(vrc, name):
    # VRC_KERNEL_PACKED: the trilinear filter through the pool's tap-packed atlas (16-bit texels holding a voxel and its
    # z neighbour, two 4-byte gathers per sample; vrc_core.h: vrc_march_segment_packed) against the oracle's
    # fetch_trilinear and against the LDS-staged form, whose positions, weights and arithmetic it shares
    s = scenes.get(name)
    L = None
    with _gpu(s) as g:
        L = g.L
        if min(s.vi.overlap[a] for a in range(3)) < 1:
            with pytest.raises(vrc.VrcError) as e:
                g.render(kernel=vrc.KERNEL_PACKED, filter_mode=vrc.FILTER_TRILINEAR)
            assert e.value.code == vrc.VRC_EINVAL
            return
        want_lin, n_want_lin = orc.oracle_render(s, threads=8, filter_mode=1)
        # the packed atlas is built here, from bricks uploaded before anyone asked for it
        got, n_got, st = g.render(kernel=vrc.KERNEL_PACKED, filter_mode=vrc.FILTER_TRILINEAR)
        assert st.kernel_variant == vrc.KERNEL_PACKED
        grey = (s.tf[:, 0] == s.tf[:, 1]).all() and (s.tf[:, 0] == s.tf[:, 2]).all()
        assert (b",%d,unsigned int," % (6 if grey else 5)) in L.vrc_last_kernel(), L.vrc_last_kernel()
        scenes.assert_parity(got, want_lin, name + " trilinear, tap-packed atlas")
        assert abs(n_got - n_want_lin) <= 2e-4 * n_want_lin + 8
        staged, n_staged, _ = g.render(kernel=vrc.KERNEL_LDS, filter_mode=vrc.FILTER_TRILINEAR)
        assert n_got == n_staged
        scenes.assert_same_frame(got, staged, name + ": tap-packed atlas vs LDS-staged", tol=1e-6)
        # the four-float colours of a coloured transfer function composite the same bits
        vrc.check(L, L.vrc_set_option(g.ctx, vrc.OPT_GREY_TABLE, 0))
        four, n_four, _ = g.render(kernel=vrc.KERNEL_PACKED, filter_mode=vrc.FILTER_TRILINEAR)
        assert b",5,unsigned int," in L.vrc_last_kernel()
        assert n_four == n_got and (four == got).all()
        vrc.check(L, L.vrc_set_option(g.ctx, vrc.OPT_GREY_TABLE, 1))
        # point sampling is refused, and leaves the context usable
        with pytest.raises(vrc.VrcError):
            g.render(kernel=vrc.KERNEL_PACKED)
        near, _, _ = g.render(kernel=vrc.KERNEL_GRID_DDA)
        scenes.assert_parity(near, orc.oracle_render(s, threads=8)[0], name + " nearest after the packed form")


def test_tap_packed_atlas_follows_uploads(vrc):
    # bricks uploaded AFTER the packed atlas exists are packed by their upload; a released and re-used slot too
    s = orc.build_scene(voxels=(64, 64, 64), block=16, viewport=(96, 80), volume="hash", spin=(0.4, 0.3))
    want_lin, _ = orc.oracle_render(s, threads=8, filter_mode=1)
    with _gpu(s) as g:
        L = g.L
        first, _, _ = g.render(kernel=vrc.KERNEL_PACKED, filter_mode=vrc.FILTER_TRILINEAR)
        scenes.assert_parity(first, want_lin, "packed atlas, before the re-uploads")
        # overwrite every slot with zeros, then put the bricks back (same slots: the free list is a stack)
        ids = list(s.ids)
        for nid in ids:
            vrc.check(L, L.vrc_pool_release_slot(g.pool, vrc.f32x3(*g.slots[nid])))
        zero = np.zeros_like(s.bricks[ids[0]])
        taken = []
        for nid in ids:
            slot = vrc.f32x3()
            b = s.bricks[nid]
            vrc.check(L, L.vrc_pool_copy_to_slot(g.pool, zero.ctypes.data, vrc.u32x3(b.shape[2], b.shape[1], b.shape[0]), slot))
            taken.append((slot[0], slot[1], slot[2]))
        black, _, _ = g.render(kernel=vrc.KERNEL_PACKED, filter_mode=vrc.FILTER_TRILINEAR)
        assert (s.tf[0] == 0).all() and not black.any(), "zero bricks under a ramp render nothing"
        # give the slots back in the order they were taken: the bricks then return to their own slots
        for t in taken:
            vrc.check(L, L.vrc_pool_release_slot(g.pool, vrc.f32x3(*t)))
        for nid in ids:
            slot = vrc.f32x3()
            b = s.bricks[nid]
            vrc.check(L, L.vrc_pool_copy_to_slot(g.pool, b.ctypes.data, vrc.u32x3(b.shape[2], b.shape[1], b.shape[0]), slot))
            g.slots[nid] = (slot[0], slot[1], slot[2])
        assert all(g.slots[nid] == s.slot_of[nid] for nid in ids)
        again, _, _ = g.render(kernel=vrc.KERNEL_PACKED, filter_mode=vrc.FILTER_TRILINEAR)
        assert (again == first).all()


@pytest.mark.parametrize("spin", [(0.0, 0.0), (1.5708, 0.0), (1.40, 0.12), (-1.5, 0.3), (0.0, 1.5708),
                                  (0.3, 1.45), (0.7854, 0.7854)])
def test_lds_region_shapes_by_view_axis(vrc, spin):
    # the LDS kernel picks its region shape (32x24x11 / 32x20x16) from the view direction: views
    # along each volume axis and near the switch-over, against the gather kernel (same sample
    # sequence, so same sample count and frame) and the oracle; boxes deeper than 11 slices take
    # the two-halves staging path of the deep shape
    s = orc.build_scene(voxels=(96, 80, 112), block=16, viewport=(160, 128), volume="hash", spin=spin)
    want, n_want = orc.oracle_render(s, threads=8)
    want_lin, n_want_lin = orc.oracle_render(s, threads=8, filter_mode=1)
    with _gpu(s) as g:
        got, n_got, _ = g.render(kernel=vrc.KERNEL_LDS)
        dda, n_dda, _ = g.render(kernel=vrc.KERNEL_GRID_DDA)
        assert n_got == n_dda
        assert np.abs(got - dda).max() <= 1e-6
        scenes.assert_parity(got, want, "lds nearest spin %s" % (spin,))
        lin, n_lin, _ = g.render(kernel=vrc.KERNEL_LDS, filter_mode=vrc.FILTER_TRILINEAR)
        gat, n_gat, _ = g.render(kernel=vrc.KERNEL_GRID_DDA, filter_mode=vrc.FILTER_TRILINEAR)
        assert n_lin == n_gat
        assert np.abs(lin - gat).max() <= 1e-5  # same taps, fma contraction may differ
        scenes.assert_parity(lin, want_lin, "lds trilinear spin %s" % (spin,))
        assert abs(n_lin - n_want_lin) <= 2e-4 * n_want_lin + 8


def test_lds_kernel_refuses_clamped_sampler(vrc):
    s = scenes.nucleon_scene()
    with _gpu(s) as g:
        with pytest.raises(Exception):
            g.render(kernel=vrc.KERNEL_LDS)
        got, _, st = g.render(filter_mode=vrc.FILTER_TRILINEAR)  # AUTO falls back to the gather form
        assert st.kernel_variant == vrc.KERNEL_GRID_DDA


@pytest.mark.parametrize("case", ["axis", "x", "oblique", "near_y", "diagonal", "c3_shape", "c3_shape_opaque"])
def test_lds_staged_trilinear_uint16(vrc, case):
    # 16-bit voxels through the LDS-staged trilinear kernel: views along each axis of a small-brick volume and the
    # 136^3-slot shape of BASELINE C3's bricks, against the gather form (same samples: same count, same frame up to
    # the contraction of multiply-adds) and the oracle
    spins = dict(axis=(0.0, 0.0), x=(1.5708, 0.0), oblique=(-1.5, 0.3), near_y=(0.3, 1.45), diagonal=(0.7854, 0.7854))
    if case.startswith("c3_shape"):
        s = orc.build_scene(voxels=(256, 256, 256), block=128, viewport=(256, 256), volume="hash", spin=(0.5236, 0.349),
                            alpha=1.0 if case.endswith("opaque") else 0.3, dtype="u16")
    else:
        s = orc.build_scene(voxels=(96, 80, 112), block=16, viewport=(160, 128), volume="hash", spin=spins[case], dtype="u16")
    want_lin, n_want_lin = orc.oracle_render(s, threads=16, filter_mode=1)
    with _gpu(s) as g:
        lin, n_lin, st = g.render(kernel=vrc.KERNEL_LDS, filter_mode=vrc.FILTER_TRILINEAR)
        assert st.kernel_variant == vrc.KERNEL_LDS and g.L.vrc_last_kernel().decode().endswith(",false,unsigned short,false>")
        gat, n_gat, st = g.render(kernel=vrc.KERNEL_GRID_DDA, filter_mode=vrc.FILTER_TRILINEAR)
        assert st.kernel_variant == vrc.KERNEL_GRID_DDA
        # AUTO: the tap-packed atlas of 16-bit voxels (32-bit texels, two 8-byte gathers per sample) -- the staged form's
        # positions, weights and arithmetic: the same frame, bit for bit
        packed, n_packed, st = g.render(filter_mode=vrc.FILTER_TRILINEAR)
        assert st.kernel_variant == vrc.KERNEL_PACKED and ",unsigned long," in g.L.vrc_last_kernel().decode(), g.L.vrc_last_kernel()
        assert n_packed == n_lin and (packed == lin).all()
        vrc.check(g.L, g.L.vrc_set_option(g.ctx, vrc.OPT_GREY_TABLE, 0))
        four, n_four, _ = g.render(kernel=vrc.KERNEL_LDS, filter_mode=vrc.FILTER_TRILINEAR)
        packed4, n_packed4, st = g.render(kernel=vrc.KERNEL_PACKED, filter_mode=vrc.FILTER_TRILINEAR)
        assert st.kernel_variant == vrc.KERNEL_PACKED and n_packed4 == n_lin and (packed4 == lin).all()
    scenes.assert_parity(lin, want_lin, "u16 staged trilinear, " + case)
    scenes.assert_parity(gat, want_lin, "u16 gathered trilinear, " + case)
    assert abs(n_lin - n_want_lin) <= 2e-4 * n_want_lin + 8
    assert abs(n_lin - n_gat) <= 2e-4 * n_gat + 8
    assert np.abs(lin - gat).max() <= (2e-3 if case.endswith("opaque") else 2e-5)
    assert n_four == n_lin and (four == lin).all()  # the four-float form of the staged kernel: the same bits


@pytest.mark.parametrize("name", ["hash64_spin_u16", "hash64_ert_u16", "mem_ragged_u16"])
def test_uint16_extension_parity(vrc, name):
    # 16-bit voxels: point sampled and trilinear, classified per sample through the data range
    from test_cpu_harness import U16_SCENES
    s = orc.build_scene(**U16_SCENES[name])
    want, n_want = orc.oracle_render(s, threads=8)
    want_lin, n_lin = orc.oracle_render(s, threads=8, filter_mode=1)
    with _gpu(s) as g:
        sub = np.zeros((8, 8, 8), dtype=np.uint16)  # the u16 atlas round-trips
        vrc.check(g.L, g.L.vrc_pool_read_region(g.pool, vrc.u32x3(8, 8, 8), vrc.u32x3(8, 8, 8), sub.ctypes.data))
        assert (sub == s.atlas[8:16, 8:16, 8:16]).all()
        for k in (vrc.KERNEL_REFERENCE_ORDER, vrc.KERNEL_GRID_DDA):
            got, n_got, st = g.render(kernel=k)
            assert st.kernel_variant == k
            scenes.assert_parity(got, want, name + " k%d" % k)
            assert abs(n_got - n_want) <= 2e-4 * n_want + 8
            lin, n_got, _ = g.render(kernel=k, filter_mode=vrc.FILTER_TRILINEAR)
            scenes.assert_parity(lin, want_lin, name + " trilinear k%d" % k)
            assert abs(n_got - n_lin) <= 2e-4 * n_lin + 8
        with pytest.raises(Exception):
            g.render(kernel=vrc.KERNEL_LDS)  # point sampling through LDS reads the classified table: 8-bit only
        # the trilinear filter is staged through LDS for 16-bit voxels too; AUTO takes the tap-packed atlas (same bits)
        staged, n_got, st = g.render(kernel=vrc.KERNEL_LDS, filter_mode=vrc.FILTER_TRILINEAR)
        assert st.kernel_variant == vrc.KERNEL_LDS and g.L.vrc_last_kernel().decode().endswith(",false,unsigned short,false>")
        scenes.assert_parity(staged, want_lin, name + " trilinear, staged")
        assert abs(n_got - n_lin) <= 2e-4 * n_lin + 8
        auto, _, st = g.render(filter_mode=vrc.FILTER_TRILINEAR)
        assert st.kernel_variant == vrc.KERNEL_PACKED and (auto == staged).all()
        vrc.check(g.L, g.L.vrc_set_option(g.ctx, vrc.OPT_PACKED_ATLAS, 0))
        auto, _, st = g.render(filter_mode=vrc.FILTER_TRILINEAR)
        assert st.kernel_variant == vrc.KERNEL_LDS and (auto == staged).all()
        vrc.check(g.L, g.L.vrc_set_option(g.ctx, vrc.OPT_PACKED_ATLAS, 1))
        uncounted, _, _ = g.render(filter_mode=vrc.FILTER_TRILINEAR, count=False)
        assert (uncounted == staged).all()


def test_pixel_buffers_of_2_pow_32_pixels_are_refused(vrc):
    # pixels are indexed in 32 bits; a 70000 x 70000 viewport (78 GB of RGBA32F would fit this GPU) is refused before
    # anything is allocated
    s = scenes.get("hash64_spin")
    L = vrc.load_library()
    ctx = C.c_void_p()
    vrc.check(L, L.vrc_ctx_create(0, C.byref(ctx)))
    try:
        view = vrc.ViewData.from_buffer_copy(bytes(s.view))
        view.glViewport[2], view.glViewport[3] = 70000, 70000
        assert L.vrc_pre_render(ctx, C.byref(view)) != 0
        assert b"2^32" in L.vrc_last_error()
        view.glViewport[2], view.glViewport[3] = s.W, s.H
        vrc.check(L, L.vrc_pre_render(ctx, C.byref(view)))
    finally:
        L.vrc_ctx_destroy(ctx)


def test_unsupported_voxel_types_are_refused(vrc):
    L = vrc.load_library()
    ctx = C.c_void_p()
    vrc.check(L, L.vrc_ctx_create(0, C.byref(ctx)))
    pool = C.c_void_p()
    for bpv, signed, flt in ((4, 0, 0), (4, 0, 1), (1, 1, 0), (2, 1, 0)):
        assert L.vrc_pool_create(ctx, bpv, signed, flt, 1, vrc.u32x3(24, 24, 24), 1 << 20,
                                 C.byref(pool)) == vrc.VRC_EUNSUPPORTED
    L.vrc_ctx_destroy(ctx)


@pytest.mark.parametrize("name", sorted(scenes.SCENES))
def test_glraycaster_variant_parity(vrc, name):
    # VRC_OPT_VARIANT = glRaycaster: the GLSL twin's frame (second parity target)
    s = scenes.get(name)
    want, n_want = orc.oracle_render(s, threads=8, variant=1)
    with _gpu(s) as g:
        for k in (vrc.KERNEL_REFERENCE_ORDER, vrc.KERNEL_GRID_DDA, vrc.KERNEL_LDS):
            got, n_got, st = g.render(kernel=k, variant=vrc.VARIANT_GLRAYCASTER)
            assert st.kernel_variant == k
            scenes.assert_parity(got, want, name + " gl k%d" % k)
            assert abs(n_got - n_want) <= 3e-4 * n_want + 8
        back, _, _ = g.render()  # and back to the CUDA variant
        scenes.assert_parity(back, orc.oracle_render(s, threads=8)[0], name + " cuda after gl")


@pytest.mark.parametrize("name,spp", [("hash64_spin", 2), ("hash64_axis", 4), ("mem_ragged", 3), ("hash_clip", 2)])
def test_glraycaster_supersampling_parity(vrc, name, spp):
    # glRaycaster with nSamplesPerPixel > 1 (fragRaycast.glsl:121-129, :212-214): spp jittered rays per pixel and
    # brick, averaged brick by brick; rendered by the reference-order kernel (AUTO selects it), refused by the others
    s = scenes.get(name)
    s.render = orc.RenderData(s.render.samplesPerRay, spp, s.render.maxSamplesPerRay, s.render.datatype,
                              (C.c_float * 2)(*s.render.dataSourceRange))
    want, n_want = orc.oracle_render(s, threads=8, variant=1)
    with _gpu(s) as g:
        got, n_got, st = g.render(variant=vrc.VARIANT_GLRAYCASTER)
        assert st.kernel_variant == vrc.KERNEL_REFERENCE_ORDER
        scenes.assert_parity(got, want, "%s gl spp %d" % (name, spp))
        assert abs(n_got - n_want) <= 3e-4 * n_want + 8
        for k in (vrc.KERNEL_GRID_DDA, vrc.KERNEL_LDS):
            with pytest.raises(vrc.VrcError):
                g.render(kernel=k, variant=vrc.VARIANT_GLRAYCASTER)
        # the CUDA variant ignores samplesPerPixel (quirk Q3)
        cuda, _, _ = g.render()
        one = scenes.get(name)
        scenes.assert_parity(cuda, orc.oracle_render(one, threads=8)[0], name + " cuda ignores spp")


def test_brick_histogram_side_kernel(vrc):
    # tests/lib/cache.cpp:103-120 (known answer): mem://#1024,1024,512,32, first child of the
    # root: every interior voxel is 17, 32^3 voxels x scale 8^3 = 2^24 in bin 17, nothing else
    L = vrc.load_library()
    vi = orc.mem_volume_info(1024, 1024, 512, 32)
    nid = orc.pack(1, 0, 0, 0)
    assert vi.depth == 5
    brick = np.full((40, 40, 40), orc.lib().orc_mem_brick_value_u8(nid), dtype=np.uint8)
    assert brick[0, 0, 0] == 17
    ctx, pool = C.c_void_p(), C.c_void_p()
    vrc.check(L, L.vrc_ctx_create(0, C.byref(ctx)))
    vrc.check(L, L.vrc_pool_create(ctx, 1, 0, 0, 1, vrc.u32x3(40, 40, 40), 4 * 40 ** 3, C.byref(pool)))
    slot = vrc.f32x3()
    vrc.check(L, L.vrc_pool_copy_to_slot(pool, brick.ctypes.data, vrc.u32x3(40, 40, 40), slot))
    bins = np.zeros(256, dtype=np.uint64)
    vrc.check(L, L.vrc_pool_histogram(pool, slot, vrc.u32x3(4, 4, 4), vrc.u32x3(32, 32, 32), 256, 8 ** 3,
                                      bins.ctypes.data))
    assert int(bins.argmax()) == 17 and int(bins[17]) == 1 << 24 and int(bins.sum()) == 1 << 24
    # a noise brick against numpy, interior only, and the reference's 1024 bins for uint16
    rng = np.random.default_rng(5)
    noise = rng.integers(0, 256, size=(40, 40, 40), dtype=np.uint8)
    vrc.check(L, L.vrc_pool_copy_to_slot(pool, noise.ctypes.data, vrc.u32x3(40, 40, 40), slot))
    vrc.check(L, L.vrc_pool_histogram(pool, slot, vrc.u32x3(4, 4, 4), vrc.u32x3(32, 32, 32), 256, 1,
                                      bins.ctypes.data))
    assert (bins == np.bincount(noise[4:36, 4:36, 4:36].ravel(), minlength=256)).all()
    assert L.vrc_pool_histogram(pool, slot, vrc.u32x3(4, 4, 4), vrc.u32x3(40, 40, 40), 256, 1,
                                bins.ctypes.data) == vrc.VRC_EINVAL  # region leaves the slot
    L.vrc_pool_destroy(pool)
    pool16 = C.c_void_p()
    vrc.check(L, L.vrc_pool_create(ctx, 2, 0, 0, 1, vrc.u32x3(24, 24, 24), 4 * 2 * 24 ** 3, C.byref(pool16)))
    n16 = rng.integers(0, 65536, size=(24, 24, 24), dtype=np.uint16)
    vrc.check(L, L.vrc_pool_copy_to_slot(pool16, n16.ctypes.data, vrc.u32x3(24, 24, 24), slot))
    bins16 = np.zeros(1024, dtype=np.uint64)
    vrc.check(L, L.vrc_pool_histogram(pool16, slot, vrc.u32x3(4, 4, 4), vrc.u32x3(16, 16, 16), 1024, 8,
                                      bins16.ctypes.data))
    want = np.bincount((n16[4:20, 4:20, 4:20].ravel() // 64).astype(np.int64), minlength=1024) * 8
    assert (bins16 == want.astype(np.uint64)).all()
    L.vrc_pool_destroy(pool16)
    L.vrc_ctx_destroy(ctx)


def test_contexts_and_pools_release_their_device_memory(vrc):
    # the HIP runtime libvrc_hip.so itself is linked against (not torch's bundled copy)
    hip = C.CDLL("libamdhip64.so")
    s = scenes.get("hash64_spin")
    free0 = None
    for i in range(12):
        with _gpu(s) as g:
            g.render()
            g.render(kernel=vrc.KERNEL_LDS, filter_mode=vrc.FILTER_TRILINEAR)
        free, total = C.c_size_t(), C.c_size_t()
        assert hip.hipDeviceSynchronize() == 0
        assert hip.hipMemGetInfo(C.byref(free), C.byref(total)) == 0
        if i == 1:
            free0 = free.value  # after the first cycles have warmed the runtime's own pools
    assert free0 - free.value < 32 << 20, (free0, free.value)


def test_trilinear_nucleon_clamped(vrc):
    s = scenes.nucleon_scene()
    want, n_want = orc.oracle_render(s, threads=8, filter_mode=1)
    with _gpu(s) as g:
        got, n_got, _ = g.render(filter_mode=vrc.FILTER_TRILINEAR)
    scenes.assert_parity(got, want, "nucleon trilinear")
    assert n_got == n_want


def test_large_launch_uses_groups_of_eight_and_matches_small_launch_rules(vrc):
    # launches above ~6144 tiles march in groups of 8 samples, smaller ones in groups of 16
    # (vrc_launch_raycast); both must reproduce the oracle, sample for sample
    s = orc.build_scene(voxels=(64, 64, 64), block=16, viewport=(1024, 512), volume="hash", spin=(0.5, 0.35))
    want, n_want = orc.oracle_render(s, threads=16)
    with _gpu(s) as g:
        got, n_got, st = g.render()
        assert st.kernel_variant == vrc.KERNEL_GRID_DDA
    scenes.assert_parity(got, want, "half a million rays through the noise volume")
    assert abs(n_got - n_want) <= 2e-4 * n_want + 8


def test_c2_full_size_rows_and_properties(vrc):
    # BASELINE C2 at full size (1024^3 mem://, block 128, 1024^2, 512 bricks): the oracle renders
    # every 64th row; the GPU frame must match those rows and their sample count, be idempotent,
    # and be the same frame from the reference-order kernel
    s = orc.build_scene(voxels=(1024, 1024, 1024), block=128, viewport=(1024, 1024))
    assert s.n_nodes == 512 and s.render.samplesPerRay == 1024
    rows = (0, 1024, 64)
    want, n_want = orc.oracle_render(s, threads=16, rows=rows)
    with _gpu(s) as g:
        got, n_got, st = g.render()
        assert st.kernel_variant == vrc.KERNEL_GRID_DDA and list(st.grid_dims) == [8, 8, 8]
        again, n_again, _ = g.render()
        assert (again == got).all() and n_again == n_got
        ref, n_ref, _ = g.render(kernel=vrc.KERNEL_REFERENCE_ORDER)
    scenes.assert_parity(got[::64], want[::64], "C2 rows")
    scenes.assert_same_frame(ref, got, "C2 reference order vs DDA")
    assert n_ref == n_got  # the grid walk composites the reference's samples, one for one
    assert 5.5e8 < n_got < 5.8e8  # SURVEY 8d estimate: ~5.6e8 samples per frame
    # the oracle's count over 16 of 1024 rows, scaled, agrees with the full-frame counter to 3 %
    assert abs(n_want * 64 - n_got) <= 0.03 * n_got


def test_c2_full_size_rows_of_the_other_modes(vrc):
    # BASELINE C2 at full size for the glRaycaster variant, the trilinear filter (LDS kernel) and
    # the gather form of the trilinear filter: every 128th row against the oracle
    s = orc.build_scene(voxels=(1024, 1024, 1024), block=128, viewport=(1024, 1024))
    rows = (0, 1024, 128)
    want_gl, _ = orc.oracle_render(s, threads=16, rows=rows, variant=1)
    want_lin, _ = orc.oracle_render(s, threads=16, rows=rows, filter_mode=1)
    with _gpu(s) as g:
        gl, _, st = g.render(variant=vrc.VARIANT_GLRAYCASTER)
        assert st.kernel_variant == vrc.KERNEL_GRID_DDA
        lin, n_lin, st = g.render(filter_mode=vrc.FILTER_TRILINEAR)
        assert st.kernel_variant == vrc.KERNEL_PACKED  # what AUTO takes: a 3 GB packed atlas
        staged, n_staged, st = g.render(kernel=vrc.KERNEL_LDS, filter_mode=vrc.FILTER_TRILINEAR)
        assert st.kernel_variant == vrc.KERNEL_LDS
        lin_gather, n_gather, st = g.render(kernel=vrc.KERNEL_GRID_DDA, filter_mode=vrc.FILTER_TRILINEAR)
        assert st.kernel_variant == vrc.KERNEL_GRID_DDA
    scenes.assert_parity(gl[::128], want_gl[::128], "C2 rows, glRaycaster variant")
    scenes.assert_parity(lin[::128], want_lin[::128], "C2 rows, trilinear (tap-packed atlas)")
    assert n_staged == n_lin and (staged == lin).all(), "C2 trilinear: the staged form composites the packed form's numbers"
    scenes.assert_same_frame(lin_gather, lin, "C2 trilinear: gather form vs packed form", tol=2e-5)
    assert abs(n_lin - n_gather) <= 2e-4 * n_lin


def test_auto_keeps_the_reference_order_for_mixed_brick_sizes(vrc):
    # a coarse brick among fine ones (an LOD cut): the reference composites in the host's centre-distance
    # order (cuda/Renderer.cu:172-199, CudaRaycastRenderer.cpp:160-163), which is not a visibility order for
    # every ray once brick sizes differ; AUTO reproduces it with the literal loop, GRID_DDA walks along the ray
    vi = orc.mem_volume_info(64, 64, 64, 16)
    coarse = orc.pack(1, 0, 0, 0)
    ids = [coarse] + [i for i in orc.leaf_ids(vi) if orc.lib().orc_nodeid_parent(i) != coarse]
    s = orc.build_scene(voxels=(64, 64, 64), block=16, viewport=(96, 80), spin=(0.4, 0.3), ids=ids, volume="hash")
    want, n_want = orc.oracle_render(s, threads=8)
    with _gpu(s) as g:
        auto, n_auto, st = g.render()
        assert st.kernel_variant == vrc.KERNEL_REFERENCE_ORDER
        scenes.assert_parity(auto, want, "mixed brick sizes, AUTO")
        assert n_auto == n_want
        dda, n_dda, st = g.render(kernel=vrc.KERNEL_GRID_DDA)
        assert st.kernel_variant == vrc.KERNEL_GRID_DDA and n_dda == n_want  # the same samples ...
        scenes.assert_close_frames(dda, auto, "grid walk vs reference order")  # ... in along-ray order
    # bricks of one size: AUTO is the grid walk, and it is the reference's frame
    s1 = scenes.get("hash64_spin")
    with _gpu(s1) as g:
        _, _, st = g.render()
        assert st.kernel_variant == vrc.KERNEL_GRID_DDA


# ---- the judged shape on data that can tell voxels apart -----------------------------------------------
# BASELINE C2/C4 use block 128 -> slots of 136^3 voxels (17^3 micro-blocks, address tables indexed up to
# 135).  mem:// bricks are constant including their overlap, so the tests above cannot see a wrong voxel
# inside such a slot; these render the seeded-noise volume ("Volume N") in the same slots.
# Reference: cuda/Renderer.cu:208-216 (sample fetch), cuda/TexturePool.cu:187-201 (copyToSlot).

@pytest.mark.parametrize("spin", [(0.0, 0.0), (0.5236, 0.349)])
def test_noise_in_136_cubed_slots_every_kernel_form(vrc, spin):
    # 256^3 noise, block 128: 8 bricks of 136^3, full 256^2 frame against the oracle
    s = orc.build_scene(voxels=(256, 256, 256), block=128, viewport=(256, 256), volume="hash", spin=spin)
    assert s.slot_dim == [136, 136, 136] and s.n_nodes == 8
    want, n_want = orc.oracle_render(s, threads=16)
    want_lin, n_lin = orc.oracle_render(s, threads=16, filter_mode=1)
    want_gl, _ = orc.oracle_render(s, threads=16, variant=1)
    with _gpu(s) as g:
        # one whole 136^3 brick read back through the layout transform, bit-exact
        nid = s.ids[5]
        o = [int(round(s.slot_of[nid][a] * s.atlas_dim[a])) for a in range(3)]
        back = np.zeros((136, 136, 136), dtype=np.uint8)
        vrc.check(g.L, g.L.vrc_pool_read_region(g.pool, vrc.u32x3(*o), vrc.u32x3(136, 136, 136), back.ctypes.data))
        assert (back == s.bricks[nid]).all()
        ref, n_ref, st = g.render(kernel=vrc.KERNEL_REFERENCE_ORDER)
        assert st.kernel_variant == vrc.KERNEL_REFERENCE_ORDER
        scenes.assert_parity(ref, want, "136^3 noise, reference order")
        flipped, _ = orc.oracle_render(s, threads=16, entry_bias=orc.ENTRY_BIAS)
        scenes.assert_no_tie_bias(ref, want, flipped, "136^3 noise, reference order")
        assert n_ref == n_want
        dda, n_dda, st = g.render(kernel=vrc.KERNEL_GRID_DDA)
        assert st.kernel_variant == vrc.KERNEL_GRID_DDA
        scenes.assert_parity(dda, want, "136^3 noise, grid DDA, fixed-point stepping")
        scenes.assert_no_tie_bias(dda, want, flipped, "136^3 noise, grid DDA, fixed-point stepping")
        assert n_dda == n_want
        flt, n_flt, _ = g.render(kernel=vrc.KERNEL_GRID_DDA, stepping=0)
        scenes.assert_parity(flt, want, "136^3 noise, grid DDA, float stepping")
        assert n_flt == n_dda
        lds, n_lds, st = g.render(kernel=vrc.KERNEL_LDS)
        assert st.kernel_variant == vrc.KERNEL_LDS
        scenes.assert_parity(lds, want, "136^3 noise, LDS-staged")
        scenes.assert_no_tie_bias(lds, want, flipped, "136^3 noise, LDS-staged")
        assert n_lds == n_dda
        gl, _, _ = g.render(variant=vrc.VARIANT_GLRAYCASTER)
        scenes.assert_parity(gl, want_gl, "136^3 noise, glRaycaster variant")
        lin, n_got_lin, st = g.render(filter_mode=vrc.FILTER_TRILINEAR)
        assert st.kernel_variant == vrc.KERNEL_PACKED
        scenes.assert_parity(lin, want_lin, "136^3 noise, trilinear (tap-packed atlas)")
        assert abs(n_got_lin - n_lin) <= 2e-4 * n_lin + 8
        staged, n_staged, st = g.render(kernel=vrc.KERNEL_LDS, filter_mode=vrc.FILTER_TRILINEAR)
        assert st.kernel_variant == vrc.KERNEL_LDS
        scenes.assert_parity(staged, want_lin, "136^3 noise, trilinear (LDS kernel)")
        assert n_staged == n_got_lin and (staged == lin).all()
        ling, _, _ = g.render(kernel=vrc.KERNEL_GRID_DDA, filter_mode=vrc.FILTER_TRILINEAR)
        scenes.assert_parity(ling, want_lin, "136^3 noise, trilinear (gather form)")


@pytest.fixture(scope="module")
def c2_noise_scene():
    # "Volume N" at BASELINE C2's size: hash 1024^3, block 128 (512 bricks of 136^3), default camera
    return orc.build_scene(voxels=(1024, 1024, 1024), block=128, viewport=(1024, 1024), volume="hash")


def test_c2_noise_volume_rows_at_1024_and_2048(vrc, c2_noise_scene):
    # the kernel instance bench.py times (grid DDA, fixed-point stepping, groups of 8) on the judged shape
    # with noise data: every 64th row of the 1024^2 frame (C2) and every 128th of the 2048^2 frame (C4's)
    # against the oracle; plus the reference-order kernel and the float stepping on the same frame
    s = c2_noise_scene
    assert s.n_nodes == 512 and s.render.samplesPerRay == 1024 and s.slot_dim == [136, 136, 136]
    rows = (0, 1024, 64)
    want, n_want = orc.oracle_render(s, threads=16, rows=rows)
    s2 = orc.with_viewport(s, 2048, 2048)
    rows2 = (0, 2048, 128)
    want2, n_want2 = orc.oracle_render(s2, threads=16, rows=rows2)
    with _gpu(s) as g:
        nid = s.ids[137]
        o = [int(round(s.slot_of[nid][a] * s.atlas_dim[a])) for a in range(3)]
        back = np.zeros((136, 136, 136), dtype=np.uint8)
        vrc.check(g.L, g.L.vrc_pool_read_region(g.pool, vrc.u32x3(*o), vrc.u32x3(136, 136, 136), back.ctypes.data))
        assert (back == s.bricks[nid]).all()
        got, n_got, st = g.render()
        assert st.kernel_variant == vrc.KERNEL_GRID_DDA and list(st.grid_dims) == [8, 8, 8]
        ref, n_ref, _ = g.render(kernel=vrc.KERNEL_REFERENCE_ORDER)
        flt, n_flt, _ = g.render(kernel=vrc.KERNEL_GRID_DDA, stepping=0)
        g.s = s2
        got2, n_got2, st2 = g.render()
        assert st2.kernel_variant == vrc.KERNEL_GRID_DDA
    scenes.assert_parity(got[::64], want[::64], "C2 noise rows")
    flipped, _ = orc.oracle_render(s, threads=16, rows=rows, entry_bias=orc.ENTRY_BIAS)
    scenes.assert_no_tie_bias(got[::64], want[::64], flipped[::64], "C2 noise rows (the kernel instance the bench times)")
    scenes.assert_parity(ref[::64], want[::64], "C2 noise rows, reference order")
    scenes.assert_parity(flt[::64], want[::64], "C2 noise rows, float stepping")
    assert abs(n_want * 64 - n_got) <= 0.03 * n_got and n_flt == n_got and n_ref == n_got
    scenes.assert_parity(got2[::128], want2[::128], "C4 frame (2048^2) noise rows")
    assert abs(n_want2 * 128 - n_got2) <= 0.03 * n_got2


@pytest.mark.parametrize("seed", range(32 * scenes.FUZZ_SCALE))
def test_random_views_all_gpu_kernels_match_the_oracle(vrc, seed):
    # the fuzz of tests/test_cpu_harness.py on the device: random volumes, cameras (also inside
    # the volume), clip planes, viewports, step sizes; gather and LDS kernels, both filters,
    # both reference variants
    from test_cpu_harness import _fuzz_scene, _fuzz_parity
    rng = np.random.default_rng(1000 + seed)
    kw = _fuzz_scene(rng)
    s = orc.build_scene(**kw)
    want, n_want = orc.oracle_render(s, threads=8)
    want_lin, _ = orc.oracle_render(s, threads=8, filter_mode=1)
    want_gl, _ = orc.oracle_render(s, threads=8, variant=1)
    with _gpu(s) as g:
        for k in (vrc.KERNEL_REFERENCE_ORDER, vrc.KERNEL_GRID_DDA, vrc.KERNEL_LDS):
            got, n_got, _ = g.render(kernel=k)
            _fuzz_parity(got, want, "seed %d k%d %r" % (seed, k, kw))
            assert n_got == n_want or (kw.get("alpha", 0.05) >= 0.3 and abs(n_got - n_want) <= 1e-4 * n_want + 8), \
                (seed, k, kw, n_got, n_want)
            got, _, _ = g.render(kernel=k, filter_mode=vrc.FILTER_TRILINEAR)
            _fuzz_parity(got, want_lin, "seed %d k%d trilinear %r" % (seed, k, kw))
            got, _, _ = g.render(kernel=k, variant=vrc.VARIANT_GLRAYCASTER)
            _fuzz_parity(got, want_gl, "seed %d k%d glRaycaster %r" % (seed, k, kw))
        if min(s.vi.overlap[a] for a in range(3)) >= 1 and max(s.slot_dim) <= 248:
            got, _, st = g.render(kernel=vrc.KERNEL_PACKED, filter_mode=vrc.FILTER_TRILINEAR)
            assert st.kernel_variant == vrc.KERNEL_PACKED
            _fuzz_parity(got, want_lin, "seed %d trilinear, tap-packed atlas %r" % (seed, kw))


@pytest.mark.parametrize("seed", range(16 * scenes.FUZZ_SCALE))
def test_random_views_uint16_and_multipass(vrc, seed):
    # the same fuzz for 16-bit volumes (gather kernels, per-sample classification) and for frames
    # rendered in several passes over random splits of the brick list (accumulating pixel buffer)
    from test_cpu_harness import _fuzz_scene, _fuzz_parity
    rng = np.random.default_rng(5000 + seed)
    kw = _fuzz_scene(rng)
    kw["volume"] = "hash"
    kw16 = dict(kw, dtype="u16")
    s16 = orc.build_scene(**kw16)
    want16, n16 = orc.oracle_render(s16, threads=8)
    want16_lin, _ = orc.oracle_render(s16, threads=8, filter_mode=1)
    with _gpu(s16) as g:
        for k in (vrc.KERNEL_REFERENCE_ORDER, vrc.KERNEL_GRID_DDA):
            got, n_got, _ = g.render(kernel=k)
            _fuzz_parity(got, want16, "seed %d u16 k%d %r" % (seed, k, kw))
            assert abs(n_got - n16) <= 3e-4 * n16 + 16
            got, _, _ = g.render(kernel=k, filter_mode=vrc.FILTER_TRILINEAR)
            _fuzz_parity(got, want16_lin, "seed %d u16 trilinear k%d %r" % (seed, k, kw))
        got, _, st = g.render(kernel=vrc.KERNEL_LDS, filter_mode=vrc.FILTER_TRILINEAR)
        assert st.kernel_variant == vrc.KERNEL_LDS
        _fuzz_parity(got, want16_lin, "seed %d u16 trilinear staged %r" % (seed, kw))
        if min(s16.vi.overlap[a] for a in range(3)) >= 1:
            got, _, st = g.render(kernel=vrc.KERNEL_PACKED, filter_mode=vrc.FILTER_TRILINEAR)
            assert st.kernel_variant == vrc.KERNEL_PACKED
            _fuzz_parity(got, want16_lin, "seed %d u16 trilinear tap-packed %r" % (seed, kw))
    s = orc.build_scene(**kw)
    want, n_want = orc.oracle_render(s, threads=8)
    cuts = sorted(set(int(c) for c in rng.integers(1, max(2, s.n_nodes), size=3)) | {0, s.n_nodes})
    passes = [(a, b) for a, b in zip(cuts[:-1], cuts[1:]) if b > a]
    with _gpu(s) as g:
        for k in (vrc.KERNEL_REFERENCE_ORDER, vrc.KERNEL_GRID_DDA, vrc.KERNEL_LDS):
            got, n_got, _ = g.render(kernel=k, passes=passes)
            _fuzz_parity(got, want, "seed %d multipass %r k%d %r" % (seed, passes, k, kw))
        if min(s.vi.overlap[a] for a in range(3)) >= 1:
            # the tap-packed march over the same passes (the grey form on the first pass only: later passes start from
            # the pixel's colour so far) and in one pass
            want_lin, n_lin = orc.oracle_render(s, threads=8, filter_mode=1)
            one, n_one, _ = g.render(kernel=vrc.KERNEL_PACKED, filter_mode=vrc.FILTER_TRILINEAR)
            got, n_got, _ = g.render(kernel=vrc.KERNEL_PACKED, filter_mode=vrc.FILTER_TRILINEAR, passes=passes)
            _fuzz_parity(got, want_lin, "seed %d multipass %r tap-packed %r" % (seed, passes, kw))
            assert n_got == n_one and np.abs(got - one).max() <= 1e-6


def test_c1_config_parity(vrc):
    # mem://#128,128,128,32, 512^2 viewport, 512 samples/ray, 64 leaf bricks
    s = orc.build_scene(voxels=(128, 128, 128), block=32, viewport=(512, 512))
    assert s.n_nodes == 64 and s.render.samplesPerRay == 512
    want, n_want = orc.oracle_render(s, threads=16)
    with _gpu(s) as g:
        got, n_got, st = g.render()
        assert st.kernel_variant == vrc.KERNEL_GRID_DDA and list(st.grid_dims) == [4, 4, 4]
    scenes.assert_parity(got, want, "C1")
    assert abs(n_got - n_want) <= 2e-4 * n_want
    assert 6.0e7 < n_want < 8.0e7  # SURVEY 8d estimate: ~7.0e7 samples/frame


def test_nucleon_raw_single_brick(vrc, golden):
    s = scenes.nucleon_scene()
    want, n_want = orc.oracle_render(s, threads=8)
    with _gpu(s) as g:
        for k in (vrc.KERNEL_REFERENCE_ORDER, vrc.KERNEL_GRID_DDA):
            got, n_got, _ = g.render(kernel=k)
            scenes.assert_parity(got, want, "nucleon")
            scenes.assert_parity(got, golden["nucleon"], "nucleon golden", budget=orc.budget_of(want))
            assert n_got == n_want


def test_multipass_equals_single_pass(vrc):
    s = scenes.get("hash64_spin")
    with _gpu(s) as g:
        one, n1, _ = g.render()
        h = s.n_nodes // 3
        many, n3, _ = g.render(passes=[(0, h), (h, 2 * h), (2 * h, s.n_nodes)])
    want, _ = orc.oracle_render(s, threads=8)
    scenes.assert_parity(many, want, "multipass")
    scenes.assert_same_frame(many, one, "multipass vs single")


def test_early_ray_termination(vrc):
    s = scenes.get("hash64_ert")
    want, n_want = orc.oracle_render(s, threads=8)
    assert (want[..., 3] > 0.999).mean() > 0.3  # most rays terminate early
    with _gpu(s) as g:
        got, n_got, _ = g.render()
    scenes.assert_parity(got, want, "ert")
    assert abs(n_got - n_want) <= 1e-3 * n_want
    # no ray composites past the threshold by more than one sample's alpha
    assert got[..., 3].max() <= 1.0


def test_sort_first_tile_is_crop_of_full_frame(vrc):
    # Equalizer-style tile: origin-0 viewport + off-axis frustum (livre/eq/Channel.cpp:151-157)
    kw = dict(scenes.SCENES["hash64_spin"])
    kw["viewport"] = (48, 48)
    fulls = orc.build_scene(**kw)
    t = orc.build_scene(tile=(12, 24, 24, 12, 48, 48), **kw)
    with _gpu(fulls) as g:
        full, _, _ = g.render()
    with _gpu(t) as g:
        tile, _, _ = g.render()
    want, _ = orc.oracle_render(t, threads=8)
    scenes.assert_parity(tile, want, "tile vs oracle")
    scenes.assert_close_frames(tile, full[24:36, 12:36], "tile vs crop of the full frame")


def test_pool_exhaustion_and_slot_reuse(vrc):
    L = vrc.load_library()
    ctx = C.c_void_p()
    vrc.check(L, L.vrc_ctx_create(0, C.byref(ctx)))
    pool = C.c_void_p()
    mb = vrc.u32x3(24, 24, 24)
    vrc.check(L, L.vrc_pool_create(ctx, 1, 0, 0, 1, mb, 3 * 24 ** 3, C.byref(pool)))
    brick = np.arange(24 ** 3, dtype=np.uint32).astype(np.uint8)
    slots = []
    for _ in range(3):
        slot = vrc.f32x3()
        vrc.check(L, L.vrc_pool_copy_to_slot(pool, brick.ctypes.data, mb, slot))
        slots.append(tuple(slot))
    assert len(set(slots)) == 3
    slot = vrc.f32x3()
    rc = L.vrc_pool_copy_to_slot(pool, brick.ctypes.data, mb, slot)
    assert rc == vrc.VRC_EFULL and tuple(slot) == (-1.0, -1.0, -1.0)  # TexturePool.cu:180-181
    vrc.check(L, L.vrc_pool_release_slot(pool, vrc.f32x3(*slots[1])))
    vrc.check(L, L.vrc_pool_copy_to_slot(pool, brick.ctypes.data, mb, slot))
    assert tuple(slot) == slots[1]
    # brick larger than a slot is refused
    assert L.vrc_pool_copy_to_slot(pool, brick.ctypes.data, vrc.u32x3(25, 24, 24), slot) == vrc.VRC_EINVAL
    L.vrc_pool_destroy(pool)
    # unsupported formats are reported, not mis-rendered (quirk Q2)
    assert L.vrc_pool_create(ctx, 4, 0, 0, 1, mb, 1 << 20, C.byref(pool)) == vrc.VRC_EUNSUPPORTED
    assert L.vrc_pool_create(ctx, 1, 0, 0, 5, mb, 1 << 20, C.byref(pool)) == vrc.VRC_EUNSUPPORTED
    assert b"Channel number" in L.vrc_last_error()
    L.vrc_ctx_destroy(ctx)


def test_error_paths(vrc):
    L = vrc.load_library()
    ctx = C.c_void_p()
    assert L.vrc_ctx_create(99, C.byref(ctx)) == vrc.VRC_EINVAL
    vrc.check(L, L.vrc_ctx_create(0, C.byref(ctx)))
    s = scenes.get("mem64_axis")
    view = C.cast(C.byref(s.view), C.POINTER(vrc.ViewData))
    render = C.cast(C.byref(s.render), C.POINTER(vrc.RenderData))
    nodes = C.cast(s.nodes, C.POINTER(vrc.NodeData))
    pool = C.c_void_p()
    vrc.check(L, L.vrc_pool_create(ctx, 1, 0, 0, 1, vrc.u32x3(24, 24, 24), 1 << 20, C.byref(pool)))
    # render before pre_render
    assert L.vrc_render(ctx, view, nodes, s.n_nodes, render, pool) == vrc.VRC_EINVAL
    assert L.vrc_update(ctx, None, None, 7) == vrc.VRC_EINVAL
    assert L.vrc_set_option(ctx, vrc.OPT_FILTER, 2) == vrc.VRC_EINVAL
    assert L.vrc_set_option(ctx, vrc.OPT_KERNEL, 4) == vrc.VRC_EINVAL
    assert L.vrc_set_option(ctx, 999, 1) == vrc.VRC_EINVAL
    vrc.check(L, L.vrc_pre_render(ctx, view))
    # empty node list renders nothing (CudaRaycastRenderer.cpp:157-158)
    vrc.check(L, L.vrc_render(ctx, view, None, 0, render, pool))
    fb = np.ones((s.H, s.W, 4), dtype=np.float32)
    vrc.check(L, L.vrc_post_render(ctx, fb.ctypes.data))
    assert (fb == 0).all()  # cleared by pre_render (PixelBufferObject.cu:80)
    L.vrc_pool_destroy(pool)
    L.vrc_ctx_destroy(ctx)


def test_concurrent_uploads_are_thread_safe(vrc):
    # the reference calls copyToSlot from 3 threads (CudaRaycastPipeline.cpp:60-63)
    import threading
    s = scenes.get("hash64_spin")
    L = vrc.load_library()
    ctx = C.c_void_p()
    vrc.check(L, L.vrc_ctx_create(0, C.byref(ctx)))
    pool = C.c_void_p()
    mb = vrc.u32x3(24, 24, 24)
    vrc.check(L, L.vrc_pool_create(ctx, 1, 0, 0, 1, mb, s.pool_bytes, C.byref(pool)))
    got = {}
    lock = threading.Lock()

    def work(ids):
        for nid in ids:
            slot = vrc.f32x3()
            rc = L.vrc_pool_copy_to_slot(pool, s.bricks[nid].ctypes.data, mb, slot)
            with lock:
                got[nid] = (rc, tuple(slot))

    ths = [threading.Thread(target=work, args=(s.ids[i::3],)) for i in range(3)]
    [t.start() for t in ths]
    [t.join() for t in ths]
    assert all(rc == 0 for rc, _ in got.values())
    assert len({sl for _, sl in got.values()}) == len(s.ids)
    vrc.check(L, L.vrc_pool_synchronize(pool))
    for nid in s.ids:  # every brick landed intact in its slot
        sl = got[nid][1]
        o = [int(round(sl[a] * s.atlas_dim[a])) for a in range(3)]
        out = np.zeros((24, 24, 24), dtype=np.uint8)
        vrc.check(L, L.vrc_pool_read_region(pool, vrc.u32x3(*o), mb, out.ctypes.data))
        assert (out == s.bricks[nid]).all()
    L.vrc_pool_destroy(pool)
    L.vrc_ctx_destroy(ctx)


# ---- per-ray adaptive LOD (EXTENSION, BASELINE C5; vrc_set_ray_lod) ---------------------------------------

def _hierarchy(voxels=(64, 64, 64), block=16, levels=None, **kw):
    vi = orc.mem_volume_info(voxels[0], voxels[1], voxels[2], block)
    return orc.build_scene(voxels=voxels, block=block, ids=orc.all_level_ids(vi, levels), **kw)


def _lod_parity(got, want, what):
    scenes.assert_parity(got, want, what, allow_frac=scenes.RAY_LOD_ALLOW)


@pytest.mark.parametrize("sse", [0.5, 1.3, 1.7, 2.5, 6.0])
def test_ray_lod_matches_oracle(vrc, sse):
    s = _hierarchy(viewport=(160, 120), volume="hash", spin=(0.4, 0.3))
    lod = (sse, orc.world_space_per_pixel(s))
    want, n_want = orc.oracle_render(s, ray_lod=lod)
    with _gpu(s) as g:
        for stepping in (1, 0):
            got, n_got, st = g.render(ray_lod=lod, stepping=stepping)
            assert st.kernel_variant == vrc.KERNEL_RAY_LOD
            _lod_parity(got, want, "sse %g stepping %d" % (sse, stepping))
            assert abs(n_got - n_want) <= 3e-4 * n_want + 16
        uncounted, _, _ = g.render(ray_lod=lod, count=False)
        assert (uncounted == g.render(ray_lod=lod)[0]).all()
        # switching the mode off again renders the list as the reference would: every brick of it
        with pytest.raises(Exception):
            g.render(kernel=vrc.KERNEL_GRID_DDA)  # nested boxes are no partition


def _slabs_of(s, axis, cuts, descending):
    """[(planes, node indices)] front to back: the slabs of space between consecutive `cuts` across `axis`, each with
    the scene's own planes + its two, and the bricks of s.nodes that reach into it."""
    out = []
    pairs = list(zip(cuts[:-1], cuts[1:]))
    if descending:
        pairs.reverse()
    for a, b in pairs:
        pa, pb = [0.0] * 4, [0.0] * 4
        pa[axis], pa[3] = 1.0, -a   # kept: n.x + d >= 0 (cuda/Renderer.cu:132-146)
        pb[axis], pb[3] = -1.0, b
        planes = [list(p) for p in s.planes] + [pa, pb]
        idx = [i for i in range(s.n_nodes)
               if s.nodes[i].aabbMin[axis] + s.nodes[i].aabbSize[axis] > a + 1e-6 and s.nodes[i].aabbMin[axis] < b - 1e-6]
        out.append((np.asarray(planes, dtype=np.float32), idx))
    return out


@pytest.mark.parametrize("spin,axis,descending", [((0.0, 0.0), 2, True), ((0.4, 0.3), 2, True), ((1.45, 0.1), 0, False)])
def test_ray_lod_in_slabs_matches_the_oracle(vrc, spin, axis, descending):
    # per-ray LOD over a hierarchy that does not fit the atlas is rendered in SLABS of space across the view's main axis
    # (libre_amd/host: HipRaycastPipeline::renderRayLodInSlabs): per slab the bricks of every level that reach into it,
    # the rays confined to it by two clip planes, front to back into the accumulating pixel buffer.  Kernel against
    # oracle through the C ABI on the same slabs; and against the single pass, which it equals up to the sampling
    # restarts at the slab faces.
    import copy
    s = _hierarchy(viewport=(120, 96), volume="hash", spin=spin)
    lod = (1.7, orc.world_space_per_pixel(s))
    slabs = _slabs_of(s, axis, [-0.5, -0.25, 0.0, 0.25, 0.5], descending)
    assert all(0 < len(idx) < s.n_nodes for _, idx in slabs)
    want = np.zeros((s.H, s.W, 4), dtype=np.float32)
    n_want = 0
    for planes, idx in slabs:
        part = copy.copy(s)
        part.planes = planes
        part.nodes = (orc.NodeData * len(idx))(*[s.nodes[i] for i in idx])
        part.n_nodes = len(idx)
        want, n = orc.oracle_render(part, ray_lod=lod, fb=want)
        n_want += n
    whole, n_whole = orc.oracle_render(s, ray_lod=lod)
    with _gpu(s) as g:
        got, n_got, st = g.render(ray_lod=lod, slabs=slabs)
        assert st.kernel_variant == vrc.KERNEL_RAY_LOD
        _lod_parity(got, want, "slabs across axis %d" % axis)
        assert abs(n_got - n_want) <= 3e-4 * n_want + 16
        single, n_single, _ = g.render(ray_lod=lod)
    # the slab frame is the per-ray LOD frame with extra sampling restarts at three faces
    scenes.assert_close_frames(got, single, "slabs against the single pass", max_abs=2e-2, mean_abs=1e-3)
    assert abs(n_got - n_single) <= 0.02 * n_single
    assert np.abs(want - whole).max() < 2e-2


@pytest.mark.parametrize("filter_mode,dtype", [(1, "u8"), (1, "u16"), (0, "u16")])
def test_ray_lod_per_sample_classification_modes(vrc, filter_mode, dtype):
    s = _hierarchy(viewport=(96, 80), volume="hash", spin=(-0.7, 0.2), dtype=dtype, alpha=0.3)
    lod = (1.6, orc.world_space_per_pixel(s))
    want, n_want = orc.oracle_render(s, ray_lod=lod, filter_mode=filter_mode)
    with _gpu(s) as g:
        got, n_got, st = g.render(ray_lod=lod, filter_mode=filter_mode)
        assert st.kernel_variant == vrc.KERNEL_RAY_LOD
        _lod_parity(got, want, "filter %d %s" % (filter_mode, dtype))
        assert abs(n_got - n_want) <= 3e-4 * n_want + 16


def _ran(g):
    return g.L.vrc_last_kernel().decode()


@pytest.mark.parametrize("sse", [0.5, 1.3, 2.5, 6.0])
@pytest.mark.parametrize("tf", ["grey", "rgb", "grey_u16"])
def test_ray_lod_trilinear_staged_through_lds(vrc, sse, tf):
    # trilinear filter on 8-bit bricks: AUTO stages the voxels through LDS (vrc_k_raycast_lds<.,true,.,true>), GRID_DDA
    # asks for the gather form (vrc_k_raycast_raylod); both are held to the oracle, and to each other
    voxel = "unsigned short" if tf == "grey_u16" else "unsigned char"
    s = _hierarchy(viewport=(160, 120), volume="hash", spin=(0.4, 0.3), alpha=0.3, dtype="u16" if tf == "grey_u16" else "u8")
    if tf == "rgb":
        i = np.arange(256, dtype=np.float32) / np.float32(255.0)
        s.tf = np.ascontiguousarray(np.stack([i, i * i, np.float32(1.0) - i, np.float32(0.3) * i], axis=1))
    lod = (sse, orc.world_space_per_pixel(s))
    want, n_want = orc.oracle_render(s, ray_lod=lod, filter_mode=1)
    with _gpu(s) as g:
        # AUTO: the tap-packed atlas (round 4; tag unsigned int: of 8-bit voxels, unsigned long: of 16-bit voxels)
        tag = ",unsigned int," if voxel == "unsigned char" else ",unsigned long,"
        auto, n_auto, st = g.render(ray_lod=lod, filter_mode=1)
        assert st.kernel_variant == vrc.KERNEL_RAY_LOD
        assert _ran(g).startswith("vrc_k_raycast_raylod<") and tag in _ran(g), _ran(g)
        staged, n_staged, st = g.render(ray_lod=lod, filter_mode=1, kernel=vrc.KERNEL_LDS)
        assert st.kernel_variant == vrc.KERNEL_RAY_LOD
        assert _ran(g).startswith("vrc_k_raycast_lds<true,true,") and _ran(g).endswith(",true,%s,false>" % voxel), _ran(g)
        assert n_auto == n_staged
        scenes.assert_same_frame(auto, staged, "per-ray LOD trilinear: AUTO vs LDS-staged", tol=1e-6)
        gathered, n_gathered, _ = g.render(ray_lod=lod, filter_mode=1, kernel=vrc.KERNEL_GRID_DDA)
        assert _ran(g).startswith("vrc_k_raycast_raylod<"), _ran(g)
        uncounted, _, _ = g.render(ray_lod=lod, filter_mode=1, count=False)
        assert (uncounted == auto).all()
        with pytest.raises(Exception):
            g.render(ray_lod=lod, filter_mode=1, kernel=vrc.KERNEL_REFERENCE_ORDER)
        # ... and through the tap-packed atlas asked for by name: the staged form's positions, weights and arithmetic
        packed, n_packed, st = g.render(ray_lod=lod, filter_mode=1, kernel=vrc.KERNEL_PACKED)
        assert st.kernel_variant == vrc.KERNEL_RAY_LOD and _ran(g).startswith("vrc_k_raycast_raylod<") and tag in _ran(g), _ran(g)
        assert n_packed == n_staged
        scenes.assert_same_frame(packed, staged, "per-ray LOD, tap-packed atlas vs LDS-staged", tol=1e-6)
    _lod_parity(staged, want, "staged sse %g" % sse)
    _lod_parity(gathered, want, "gathers sse %g" % sse)
    assert abs(n_staged - n_want) <= 3e-4 * n_want + 16
    assert abs(n_gathered - n_want) <= 3e-4 * n_want + 16


@pytest.mark.parametrize("seed", range(10 * scenes.FUZZ_SCALE))
def test_ray_lod_trilinear_staged_random_views(vrc, seed):
    rng = np.random.default_rng(7300 + seed)
    vox = [int(rng.choice([32, 64, 96])) for _ in range(3)]
    block = int(rng.choice([16, 32]))
    vox = [max(v, block) // block * block for v in vox]
    kw = dict(voxels=tuple(vox), block=block, viewport=(int(rng.integers(40, 200)), int(rng.integers(40, 160))),
              volume=str(rng.choice(["hash", "mem"])),
              spin=(float(rng.uniform(-3.1, 3.1)), float(rng.uniform(-1.5, 1.5))),
              alpha=float(rng.choice([0.05, 0.3, 1.0])))
    if rng.random() < 0.3:
        kw["eye"] = (float(rng.uniform(-0.4, 0.4)), float(rng.uniform(-0.4, 0.4)), float(rng.uniform(0.1, 0.9)))
    if rng.random() < 0.3:
        kw["planes"] = [[0.0, 0.6, 0.8, 0.2]]
    s = _hierarchy(**kw)
    lod = (float(rng.uniform(0.3, 4.0)) * vox[0] / 64.0 * 48.0 / s.H, orc.world_space_per_pixel(s))
    want, n_want = orc.oracle_render(s, threads=8, ray_lod=lod, filter_mode=1)
    with _gpu(s) as g:
        got, n_got, _ = g.render(ray_lod=lod, filter_mode=1, kernel=vrc.KERNEL_LDS)
        assert _ran(g).startswith("vrc_k_raycast_lds<"), _ran(g)
        auto, n_auto, _ = g.render(ray_lod=lod, filter_mode=1)  # the tap-packed atlas: the same numbers
        assert ",unsigned int," in _ran(g), _ran(g)
        assert n_auto == n_got
        scenes.assert_same_frame(auto, got, "seed %d: tap-packed atlas vs staged under per-ray LOD" % seed, tol=1e-6)
    _lod_parity(got, want, "seed %d %r lod %r" % (seed, kw, lod))
    assert abs(n_got - n_want) <= 3e-4 * n_want + 16


def test_ray_lod_small_bound_equals_the_leaf_render(vrc):
    s = _hierarchy(viewport=(128, 96), volume="hash", spin=(0.5, -0.2))
    leaves = orc.build_scene(voxels=(64, 64, 64), block=16, viewport=(128, 96), volume="hash", spin=(0.5, -0.2),
                             spr=s.render.samplesPerRay)
    with _gpu(leaves) as g:
        want, n_want, _ = g.render(kernel=vrc.KERNEL_GRID_DDA)
    with _gpu(s) as g:
        got, n_got, _ = g.render(ray_lod=(0.01, orc.world_space_per_pixel(s)))
    # the runs start 1 % of a voxel inside a brick, the per-brick segments on its face
    mx, mean, over = orc.compare(got, want)
    assert mx <= 5 * scenes.MAX_ABS and mean <= 8 * scenes.MEAN_ABS and over <= 0.06, (mx, mean, over)
    assert abs(n_got - n_want) <= 2e-3 * n_want + 16


def test_ray_lod_partial_hierarchy_clip_planes_eye_inside(vrc):
    vi = orc.mem_volume_info(64, 64, 64, 16)
    ids = orc.all_level_ids(vi, [0, 1])
    ids += [orc.pack(2, x, y, z, 0) for x in range(2) for y in range(2) for z in range(2, 4)]
    s = orc.build_scene(voxels=(64, 64, 64), block=16, viewport=(96, 96), volume="hash", ids=ids,
                        eye=(0.1, -0.2, 0.45), spin=(0.3, 0.6), planes=[[0.0, 0.0, 1.0, 0.3], [0.6, 0.8, 0.0, 0.25]])
    lod = (0.8, orc.world_space_per_pixel(s))
    want, n_want = orc.oracle_render(s, ray_lod=lod)
    with _gpu(s) as g:
        got, n_got, _ = g.render(ray_lod=lod)
    _lod_parity(got, want, "partial hierarchy")
    assert abs(n_got - n_want) <= 3e-4 * n_want + 16


def test_ray_lod_refusals(vrc):
    s = _hierarchy(viewport=(32, 32), volume="hash")
    lod = (1.0, orc.world_space_per_pixel(s))
    with _gpu(s) as g:
        with pytest.raises(Exception):
            g.render(ray_lod=lod, variant=1)  # cudaRaycaster rules only
        with pytest.raises(Exception):
            g.render(ray_lod=lod, kernel=vrc.KERNEL_LDS)
        assert g.L.vrc_set_ray_lod(g.ctx, 1, 0.0, 1.0) != 0
        assert g.L.vrc_set_ray_lod(g.ctx, 1, 1.0, -1.0) != 0
        g.render(ray_lod=lod)
    vi = orc.mem_volume_info(64, 64, 64, 16)
    dup = orc.build_scene(voxels=(64, 64, 64), block=16, viewport=(16, 16),
                          ids=orc.all_level_ids(vi, [2]) + [orc.pack(2, 0, 0, 0, 0)])
    with _gpu(dup) as g:
        with pytest.raises(Exception):
            g.render(ray_lod=lod)


@pytest.mark.parametrize("seed", range(12 * scenes.FUZZ_SCALE))
def test_ray_lod_random_views(vrc, seed):
    rng = np.random.default_rng(7100 + seed)
    vox = [int(rng.choice([32, 64, 96])) for _ in range(3)]
    block = int(rng.choice([16, 32]))
    vox = [max(v, block) // block * block for v in vox]
    kw = dict(voxels=tuple(vox), block=block, viewport=(int(rng.integers(40, 200)), int(rng.integers(40, 160))),
              volume=str(rng.choice(["hash", "mem"])),
              spin=(float(rng.uniform(-3.1, 3.1)), float(rng.uniform(-1.5, 1.5))),
              alpha=float(rng.choice([0.05, 0.3, 1.0])))
    if rng.random() < 0.3:
        kw["eye"] = (float(rng.uniform(-0.4, 0.4)), float(rng.uniform(-0.4, 0.4)), float(rng.uniform(0.1, 0.9)))
    s = _hierarchy(**kw)
    lod = (float(rng.uniform(0.3, 4.0)) * vox[0] / 64.0 * 48.0 / s.H, orc.world_space_per_pixel(s))
    want, n_want = orc.oracle_render(s, threads=8, ray_lod=lod)
    with _gpu(s) as g:
        got, n_got, _ = g.render(ray_lod=lod)
    _lod_parity(got, want, "seed %d %r lod %r" % (seed, kw, lod))
    assert abs(n_got - n_want) <= 3e-4 * n_want + 16


def test_atlas_of_more_than_2_pow_32_voxels(vrc):
    # the atlas is sized for the GPU's memory, not for a 32-bit index (the reference truncates, quirk Q12):
    # a 6 GB pool (4080 x 4080 x 360 voxels); bricks land below and above the 4 Gi-voxel line
    # (the free list hands out slots z-fastest), are read back bit-exactly and render the frame
    # of the same scene in a small pool, bit for bit
    s = scenes.get("hash64_spin")
    with _gpu(s) as g:
        want, n_want, _ = g.render(kernel=vrc.KERNEL_GRID_DDA, stepping=0)
        want_lin, _, _ = g.render(kernel=vrc.KERNEL_GRID_DDA, filter_mode=vrc.FILTER_TRILINEAR)
        want_lds, n_lds, _ = g.render(kernel=vrc.KERNEL_LDS, filter_mode=vrc.FILTER_TRILINEAR)
        want_ref, _, _ = g.render(kernel=vrc.KERNEL_REFERENCE_ORDER, stepping=0)
    L = vrc.load_library()
    ctx, pool = C.c_void_p(), C.c_void_p()
    vrc.check(L, L.vrc_ctx_create(0, C.byref(ctx)))
    try:
        mb = [s.vi.maximumBlockSize[a] for a in range(3)]
        vrc.check(L, L.vrc_pool_create(ctx, 1, 0, 0, 1, vrc.u32x3(*mb), 6 * 1000 ** 3, C.byref(pool)))
        sb, ab, fs = C.c_size_t(), C.c_size_t(), C.c_uint32()
        ad, sl = vrc.u32x3(), vrc.u32x3()
        vrc.check(L, L.vrc_pool_info(pool, C.byref(sb), ad, C.byref(ab), sl, C.byref(fs)))
        atlas_dim = list(ad)
        assert atlas_dim[0] * atlas_dim[1] * atlas_dim[2] > 2 ** 32
        slot_dim = [atlas_dim[a] // sl[a] for a in range(3)]
        slots, high = {}, 0
        for nid in s.ids:
            brick = s.bricks[nid]
            slot = vrc.f32x3()
            vrc.check(L, L.vrc_pool_copy_to_slot(pool, brick.ctypes.data,
                                                 vrc.u32x3(brick.shape[2], brick.shape[1], brick.shape[0]), slot))
            slots[nid] = (slot[0], slot[1], slot[2])
            idx = [int(round(slot[a] * sl[a])) for a in range(3)]
            base = ((idx[2] * sl[1] + idx[1]) * sl[0] + idx[0]) * slot_dim[0] * slot_dim[1] * slot_dim[2]
            high += base >= 2 ** 32
            # read the brick back through the logical atlas coordinates
            origin = [idx[a] * slot_dim[a] for a in range(3)]
            out = np.zeros_like(brick)
            vrc.check(L, L.vrc_pool_read_region(pool, vrc.u32x3(*origin),
                                                vrc.u32x3(brick.shape[2], brick.shape[1], brick.shape[0]),
                                                out.ctypes.data))
            assert (out == brick).all()
        assert 0 < high < len(s.ids)
        # node list in the scene's (sorted) order, texture coordinates for THIS atlas
        nodes = (vrc.NodeData * s.n_nodes)()
        for k, nid in enumerate(s.sorted_ids):
            tp, ts = orc.f32x3(), orc.f32x3()
            orc.lib().orc_texture_object(C.byref(s.vi), C.byref(s.lod[nid]), orc.f32x3(*slots[nid]),
                                         orc.u32x3(*atlas_dim), tp, ts)
            for a in range(3):
                nodes[k].textureMin[a] = tp[a]
                nodes[k].textureSize[a] = ts[a]
                nodes[k].aabbMin[a] = s.nodes[k].aabbMin[a]
                nodes[k].aabbSize[a] = s.nodes[k].aabbSize[a]
        view = C.cast(C.byref(s.view), C.POINTER(vrc.ViewData))
        render = C.cast(C.byref(s.render), C.POINTER(vrc.RenderData))
        vrc.check(L, L.vrc_update(ctx, s.tf.ctypes.data, None, 0))
        vrc.check(L, L.vrc_set_option(ctx, vrc.OPT_COUNT_SAMPLES, 1))

        def frame(kernel, flt):
            vrc.check(L, L.vrc_set_option(ctx, vrc.OPT_KERNEL, kernel))
            vrc.check(L, L.vrc_set_option(ctx, vrc.OPT_FILTER, flt))
            vrc.check(L, L.vrc_pre_render(ctx, view))
            vrc.check(L, L.vrc_render(ctx, view, nodes, s.n_nodes, render, pool))
            fb = np.zeros((s.H, s.W, 4), dtype=np.float32)
            vrc.check(L, L.vrc_post_render(ctx, fb.ctypes.data))
            st = vrc.Stats()
            vrc.check(L, L.vrc_get_stats(ctx, C.byref(st)))
            return fb, st

        got, st = frame(vrc.KERNEL_AUTO, vrc.FILTER_NEAREST)
        assert st.kernel_variant == vrc.KERNEL_GRID_DDA and st.samples == n_want
        assert (got == want).all()
        got, st = frame(vrc.KERNEL_REFERENCE_ORDER, vrc.FILTER_NEAREST)
        assert (got == want_ref).all()
        got, st = frame(vrc.KERNEL_GRID_DDA, vrc.FILTER_TRILINEAR)  # the gather form
        assert st.kernel_variant == vrc.KERNEL_GRID_DDA
        assert np.abs(got - want_lin).max() <= 1e-6
        # AUTO takes the tap-packed atlas here too (64-bit lane pointers: the ...,true> instance) and, without it, stages
        # the voxels through LDS (64-bit slot bases): the staged frame of the small pool, bit for bit, both times
        got, st = frame(vrc.KERNEL_AUTO, vrc.FILTER_TRILINEAR)
        assert st.kernel_variant == vrc.KERNEL_PACKED and L.vrc_last_kernel().decode().endswith(",unsigned int,12,true>"), L.vrc_last_kernel()
        assert (got == want_lds).all() and st.samples == n_lds
        vrc.check(L, L.vrc_set_option(ctx, vrc.OPT_PACKED_ATLAS, 0))
        got, st = frame(vrc.KERNEL_AUTO, vrc.FILTER_TRILINEAR)
        assert st.kernel_variant == vrc.KERNEL_LDS and L.vrc_last_kernel().decode().endswith(",false,unsigned char,true>")
        assert (got == want_lds).all() and st.samples == n_lds
        assert L.vrc_set_option(ctx, vrc.OPT_KERNEL, vrc.KERNEL_LDS) == 0
        vrc.check(L, L.vrc_set_option(ctx, vrc.OPT_FILTER, vrc.FILTER_NEAREST))
        vrc.check(L, L.vrc_pre_render(ctx, view))
        assert L.vrc_render(ctx, view, nodes, s.n_nodes, render, pool) != 0  # its point-sampling form: refused, loudly
    finally:
        if pool:
            L.vrc_pool_destroy(pool)
        L.vrc_ctx_destroy(ctx)


def _frames_in_a_large_pool(vrc, s, pool_bytes, line_voxels, cases):
    """Upload the scene's bricks into a pool of pool_bytes (bricks must land on both sides of voxel offset line_voxels),
    render every case = (options {OPT: value}, ray_lod or None) and return [(frame, samples, kernel name)]."""
    L = vrc.load_library()
    ctx, pool = C.c_void_p(), C.c_void_p()
    vrc.check(L, L.vrc_ctx_create(0, C.byref(ctx)))
    out = []
    try:
        mb = [s.vi.maximumBlockSize[a] for a in range(3)]
        vrc.check(L, L.vrc_pool_create(ctx, s.atlas.dtype.itemsize, 0, 0, 1, vrc.u32x3(*mb), pool_bytes, C.byref(pool)))
        sb, ab, fs = C.c_size_t(), C.c_size_t(), C.c_uint32()
        ad, sl = vrc.u32x3(), vrc.u32x3()
        vrc.check(L, L.vrc_pool_info(pool, C.byref(sb), ad, C.byref(ab), sl, C.byref(fs)))
        atlas_dim = list(ad)
        assert atlas_dim[0] * atlas_dim[1] * atlas_dim[2] > line_voxels
        slot_dim = [atlas_dim[a] // sl[a] for a in range(3)]
        slots, high = {}, 0
        for nid in s.ids:
            brick = s.bricks[nid]
            slot = vrc.f32x3()
            vrc.check(L, L.vrc_pool_copy_to_slot(pool, brick.ctypes.data,
                                                 vrc.u32x3(brick.shape[2], brick.shape[1], brick.shape[0]), slot))
            slots[nid] = (slot[0], slot[1], slot[2])
            idx = [int(round(slot[a] * sl[a])) for a in range(3)]
            high += ((idx[2] * sl[1] + idx[1]) * sl[0] + idx[0]) * slot_dim[0] * slot_dim[1] * slot_dim[2] >= line_voxels
        assert 0 < high < len(s.ids)
        nodes = (vrc.NodeData * s.n_nodes)()
        for k, nid in enumerate(s.sorted_ids):
            tp, ts = orc.f32x3(), orc.f32x3()
            orc.lib().orc_texture_object(C.byref(s.vi), C.byref(s.lod[nid]), orc.f32x3(*slots[nid]),
                                         orc.u32x3(*atlas_dim), tp, ts)
            for a in range(3):
                nodes[k].textureMin[a] = tp[a]
                nodes[k].textureSize[a] = ts[a]
                nodes[k].aabbMin[a] = s.nodes[k].aabbMin[a]
                nodes[k].aabbSize[a] = s.nodes[k].aabbSize[a]
        view = C.cast(C.byref(s.view), C.POINTER(vrc.ViewData))
        render = C.cast(C.byref(s.render), C.POINTER(vrc.RenderData))
        vrc.check(L, L.vrc_update(ctx, s.tf.ctypes.data, None, 0))
        vrc.check(L, L.vrc_set_option(ctx, vrc.OPT_COUNT_SAMPLES, 1))
        for options, lod in cases:
            for o, v in options.items():
                vrc.check(L, L.vrc_set_option(ctx, o, v))
            vrc.check(L, L.vrc_set_ray_lod(ctx, 1 if lod else 0, lod[0] if lod else 0.0, lod[1] if lod else 0.0))
            vrc.check(L, L.vrc_pre_render(ctx, view))
            vrc.check(L, L.vrc_render(ctx, view, nodes, s.n_nodes, render, pool))
            fb = np.zeros((s.H, s.W, 4), dtype=np.float32)
            vrc.check(L, L.vrc_post_render(ctx, fb.ctypes.data))
            st = vrc.Stats()
            vrc.check(L, L.vrc_get_stats(ctx, C.byref(st)))
            out.append((fb, st.samples, st.kernel_variant, L.vrc_last_kernel().decode()))
    finally:
        if pool:
            L.vrc_pool_destroy(pool)
        L.vrc_ctx_destroy(ctx)
    return out


def test_ray_lod_in_an_atlas_of_more_than_2_pow_32_voxels(vrc):
    # per-ray adaptive LOD in a pool of more than 2^32 voxels (BASELINE C3's full-size atlas is one): round 3 refused it
    # (32-bit slot bases in the hierarchy walk); now the walk has instances with 64-bit slot bases and float positions
    # (vrc_k_raycast_raylod<...,true>).  The whole hierarchy of a 64^3 volume in a 6 GB pool, bricks on both sides of the
    # 4 Gi-voxel line: the frame of the same hierarchy in a small pool with float stepping, bit for bit, point-sampled
    # and with the trilinear filter (gather form).  Reference hook: CudaRaycastPipeline.cpp:236-301 (what renders C3).
    vi = orc.mem_volume_info(64, 64, 64, 16)
    s = orc.build_scene(voxels=(64, 64, 64), block=16, viewport=(96, 80), volume="hash", spin=(0.5, 0.35),
                        ids=orc.all_level_ids(vi))
    lod = (0.8, orc.world_space_per_pixel(s))
    with _gpu(s) as g:
        want, n_want, st = g.render(kernel=vrc.KERNEL_GRID_DDA, stepping=0, ray_lod=lod)
        assert st.kernel_variant == vrc.KERNEL_RAY_LOD
        want_lin, n_lin, _ = g.render(kernel=vrc.KERNEL_GRID_DDA, filter_mode=vrc.FILTER_TRILINEAR, ray_lod=lod)
        want_pk, n_pk, _ = g.render(kernel=vrc.KERNEL_PACKED, filter_mode=vrc.FILTER_TRILINEAR, ray_lod=lod)
    # ... and the tap-packed atlas of such a pool (13.5 GB here): 64-bit lane pointers, the small pool's packed frame
    got = _frames_in_a_large_pool(vrc, s, 6 * 1000 ** 3, 2 ** 32,
                                  [({vrc.OPT_FILTER: vrc.FILTER_NEAREST}, lod),
                                   ({vrc.OPT_FILTER: vrc.FILTER_TRILINEAR, vrc.OPT_PACKED_ATLAS: 0}, lod),
                                   ({vrc.OPT_FILTER: vrc.FILTER_TRILINEAR, vrc.OPT_PACKED_ATLAS: 1}, lod)])
    for (fb, n, variant, name), frame_want, n_w in zip(got, (want, want_lin, want_pk), (n_want, n_lin, n_pk)):
        assert variant == vrc.KERNEL_RAY_LOD and name.endswith(",true>"), name
        assert n == n_w and (fb == frame_want).all(), name
    assert ",unsigned int," in got[2][3] and ",unsigned int," not in got[1][3], (got[1][3], got[2][3])


def test_tap_packed_atlas_of_more_than_4_gib(vrc):
    # the packed kernels address the packed atlas as scalar base + 32-bit byte offset per lane; a packed atlas of more
    # than 4 GiB (a byte atlas past 1.9 Gi voxels: the reference's default 3 GB texture cache is one) takes instances
    # with 64-bit lane pointers (<..., true>).  A 64^3 volume's hierarchy in a 3 GB pool, bricks on both sides of the
    # packed atlas's 4 GiB line: the frames of the small pool, bit for bit, with and without per-ray LOD
    vi = orc.mem_volume_info(64, 64, 64, 16)
    s = orc.build_scene(voxels=(64, 64, 64), block=16, viewport=(96, 80), volume="hash", spin=(0.5, 0.35),
                        ids=orc.all_level_ids(vi))
    lod = (0.8, orc.world_space_per_pixel(s))
    with _gpu(s) as g:
        want_lod, n_lod, st = g.render(kernel=vrc.KERNEL_PACKED, filter_mode=vrc.FILTER_TRILINEAR, ray_lod=lod)
        assert st.kernel_variant == vrc.KERNEL_RAY_LOD and _ran(g).endswith(",unsigned int,false>"), _ran(g)
        want, n_want, st = g.render(kernel=vrc.KERNEL_PACKED, filter_mode=vrc.FILTER_TRILINEAR)
        assert st.kernel_variant == vrc.KERNEL_PACKED and _ran(g).endswith(",false>"), _ran(g)
    line = (2 ** 32) * 4 // 9  # voxels whose packed texels (2.25 bytes each) fill 4 GiB
    opts = {vrc.OPT_FILTER: vrc.FILTER_TRILINEAR, vrc.OPT_KERNEL: vrc.KERNEL_PACKED}
    got = _frames_in_a_large_pool(vrc, s, 3 * 1000 ** 3, line, [(opts, lod), (opts, None)])
    for (fb, n, variant, name), frame_want, n_w, v in zip(got, (want_lod, want), (n_lod, n_want), (vrc.KERNEL_RAY_LOD, vrc.KERNEL_PACKED)):
        assert variant == v and ",unsigned int," in name and name.endswith(",true>"), name
        assert n == n_w and (fb == frame_want).all(), name
    # the same for 16-bit voxels (32-bit texels: the 4 GiB line lies at 2^32 / 4.5 voxels)
    s16 = orc.build_scene(voxels=(64, 64, 64), block=16, viewport=(96, 80), volume="hash", spin=(0.5, 0.35),
                          ids=orc.all_level_ids(vi), dtype="u16")
    with _gpu(s16) as g:
        want16, n16, st = g.render(kernel=vrc.KERNEL_PACKED, filter_mode=vrc.FILTER_TRILINEAR)
        assert st.kernel_variant == vrc.KERNEL_PACKED and _ran(g).endswith(",unsigned long,12,false>"), _ran(g)
    (fb, n, variant, name), = _frames_in_a_large_pool(vrc, s16, 6 * 1000 ** 3, (2 ** 32) * 2 // 9, [(opts, None)])
    assert variant == vrc.KERNEL_PACKED and name.endswith(",unsigned long,12,true>"), name
    assert n == n16 and (fb == want16).all(), name


def test_first_uploads_into_a_fresh_large_pool_survive_its_clear(vrc):
    # a new atlas is cleared; the clear is queued on the upload stream, BEFORE the first uploads (round 4: it was a
    # hipMemset on the null stream, which a non-blocking stream does not wait for -- in one cold run of four the clear of
    # a 6 GB pool overtook the first bricks written into it).  Several fresh 6 GB pools, bricks uploaded at once from
    # four threads, every brick read back
    import threading
    L = vrc.load_library()
    rng = np.random.default_rng(11)
    bricks = [rng.integers(1, 256, size=(24, 24, 24), dtype=np.uint8) for _ in range(32)]
    ctx = C.c_void_p()
    vrc.check(L, L.vrc_ctx_create(0, C.byref(ctx)))
    try:
        for _ in range(4):
            pool = C.c_void_p()
            vrc.check(L, L.vrc_pool_create(ctx, 1, 0, 0, 1, vrc.u32x3(24, 24, 24), 6 * 1000 ** 3, C.byref(pool)))
            try:
                slots = [None] * len(bricks)

                def up(t):
                    for i in range(t, len(bricks), 4):
                        slot = vrc.f32x3()
                        vrc.check(L, L.vrc_pool_copy_to_slot(pool, bricks[i].ctypes.data, vrc.u32x3(24, 24, 24), slot))
                        slots[i] = (slot[0], slot[1], slot[2])
                ths = [threading.Thread(target=up, args=(t,)) for t in range(4)]
                for t in ths:
                    t.start()
                for t in ths:
                    t.join()
                sb, ab, fs = C.c_size_t(), C.c_size_t(), C.c_uint32()
                ad, sl = vrc.u32x3(), vrc.u32x3()
                vrc.check(L, L.vrc_pool_info(pool, C.byref(sb), ad, C.byref(ab), sl, C.byref(fs)))
                for i, b in enumerate(bricks):
                    origin = [int(round(slots[i][a] * sl[a])) * (ad[a] // sl[a]) for a in range(3)]
                    out = np.zeros_like(b)
                    vrc.check(L, L.vrc_pool_read_region(pool, vrc.u32x3(*origin), vrc.u32x3(24, 24, 24), out.ctypes.data))
                    assert (out == b).all(), "brick %d of a fresh pool" % i
            finally:
                L.vrc_pool_destroy(pool)
    finally:
        L.vrc_ctx_destroy(ctx)


def test_slot_longer_than_255_voxels_marches_with_float_positions(vrc):
    # pool creation bounds a slot's volume (2^24 voxels), not its edges: a 300 x 16 x 16 brick (overlap 1) has slot-local
    # coordinates that do not fit the 8.24 fixed-point positions of VRC_OPT_STEPPING = 1, the address tables and the LDS
    # kernel's boxes.  Such pools march with float positions whatever the option says (same frame, bit for bit), the
    # trilinear filter takes the gather form and the LDS kernel is refused
    s = orc.build_scene(voxels=(64, 64, 64), block=64, viewport=(128, 96), spin=(0.5, 0.3), alpha=0.3)  # camera only
    rng = np.random.default_rng(77)
    interior, ov = (300, 16, 16), 1
    size = tuple(d + 2 * ov for d in interior)
    brick = rng.integers(0, 256, size=(size[2], size[1], size[0]), dtype=np.uint8)
    L = vrc.load_library()
    ctx, pool = C.c_void_p(), C.c_void_p()
    vrc.check(L, L.vrc_ctx_create(0, C.byref(ctx)))
    try:
        vrc.check(L, L.vrc_pool_create(ctx, 1, 0, 0, 1, vrc.u32x3(*size), 64 << 20, C.byref(pool)))
        sb, ab, fs = C.c_size_t(), C.c_size_t(), C.c_uint32()
        ad, sl = vrc.u32x3(), vrc.u32x3()
        vrc.check(L, L.vrc_pool_info(pool, C.byref(sb), ad, C.byref(ab), sl, C.byref(fs)))
        slot = vrc.f32x3()
        vrc.check(L, L.vrc_pool_copy_to_slot(pool, brick.ctypes.data, vrc.u32x3(*size), slot))
        node = (vrc.NodeData * 1)()
        for a in range(3):
            node[0].textureMin[a] = slot[a] + ov / ad[a]        # CudaTextureObject.cpp:61-84
            node[0].textureSize[a] = interior[a] / ad[a]
            node[0].aabbSize[a] = interior[a] / 300.0
            node[0].aabbMin[a] = -0.5 * interior[a] / 300.0
        view = C.cast(C.byref(s.view), C.POINTER(vrc.ViewData))
        render = C.cast(C.byref(s.render), C.POINTER(vrc.RenderData))
        vrc.check(L, L.vrc_update(ctx, s.tf.ctypes.data, None, 0))
        vrc.check(L, L.vrc_set_option(ctx, vrc.OPT_COUNT_SAMPLES, 1))

        def frame(kernel, flt, stepping):
            vrc.check(L, L.vrc_set_option(ctx, vrc.OPT_KERNEL, kernel))
            vrc.check(L, L.vrc_set_option(ctx, vrc.OPT_FILTER, flt))
            vrc.check(L, L.vrc_set_option(ctx, vrc.OPT_STEPPING, stepping))
            vrc.check(L, L.vrc_pre_render(ctx, view))
            vrc.check(L, L.vrc_render(ctx, view, node, 1, render, pool))
            fb = np.zeros((s.H, s.W, 4), dtype=np.float32)
            vrc.check(L, L.vrc_post_render(ctx, fb.ctypes.data))
            st = vrc.Stats()
            vrc.check(L, L.vrc_get_stats(ctx, C.byref(st)))
            return fb, st

        for kernel in (vrc.KERNEL_REFERENCE_ORDER, vrc.KERNEL_GRID_DDA):
            flt_frame, st0 = frame(kernel, vrc.FILTER_NEAREST, 0)
            fix_frame, st1 = frame(kernel, vrc.FILTER_NEAREST, 1)
            assert flt_frame[..., 3].max() > 0.2 and st0.samples > 2000
            assert (fix_frame == flt_frame).all() and st1.samples == st0.samples, kernel
        lin0, st = frame(vrc.KERNEL_AUTO, vrc.FILTER_TRILINEAR, 0)
        assert st.kernel_variant == vrc.KERNEL_GRID_DDA  # not the LDS kernel
        lin1, _ = frame(vrc.KERNEL_AUTO, vrc.FILTER_TRILINEAR, 1)
        assert (lin1 == lin0).all()
        assert L.vrc_set_option(ctx, vrc.OPT_KERNEL, vrc.KERNEL_LDS) == 0
        vrc.check(L, L.vrc_pre_render(ctx, view))
        assert L.vrc_render(ctx, view, node, 1, render, pool) != 0
    finally:
        if pool:
            L.vrc_pool_destroy(pool)
        L.vrc_ctx_destroy(ctx)


def test_ray_lod_matches_the_committed_frames(vrc):
    # tests/golden/frames_ray_lod.npz: the per-ray LOD definition pinned between rounds (regular tree at two
    # error bounds, the ragged UVF fixture)
    import importlib.util
    gdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    spec = importlib.util.spec_from_file_location("make_golden_ray_lod", os.path.join(gdir, "make_golden_ray_lod.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    golden_lod = np.load(os.path.join(gdir, "frames_ray_lod.npz"))
    for name, s, sse in gen.cases():
        with _gpu(s) as g:
            got, n_got, st = g.render(ray_lod=(sse, orc.world_space_per_pixel(s)))
        assert st.kernel_variant == vrc.KERNEL_RAY_LOD
        want, _ = orc.oracle_render(s, threads=8, ray_lod=(sse, orc.world_space_per_pixel(s)))
        assert np.allclose(want, golden_lod[name], atol=1e-6), name
        scenes.assert_parity(got, golden_lod[name], name + " gpu vs golden", budget=orc.budget_of(want))
        n_want = int(golden_lod[name + "__samples"][0])
        assert abs(n_got - n_want) <= 3e-4 * n_want + 16


def test_ray_lod_trilinear_on_the_pinned_hierarchies(vrc):
    # the hierarchies of tests/golden/frames_ray_lod.npz (a regular tree at two error bounds, the ragged UVF fixture
    # whose levels do not align) with the trilinear filter: staged through LDS where the bricks have an overlap, by
    # gathers otherwise, both against the oracle
    import importlib.util
    gdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    spec = importlib.util.spec_from_file_location("make_golden_ray_lod", os.path.join(gdir, "make_golden_ray_lod.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    staged_cases = 0
    golden_lod = np.load(os.path.join(gdir, "frames_ray_lod.npz"))
    for name, s, sse in gen.cases():
        lod = (sse, orc.world_space_per_pixel(s))
        want, n_want = orc.oracle_render(s, threads=8, ray_lod=lod, filter_mode=1)
        assert np.allclose(want, golden_lod[name + "__trilinear"], atol=1e-6) and n_want == int(golden_lod[name + "__trilinear_samples"][0])
        with _gpu(s) as g:
            # AUTO: the tap-packed atlas where the bricks have an overlap (round 4), else the gather form; with
            # VRC_OPT_PACKED_ATLAS off the staged form as in round 3
            auto, n_auto, st = g.render(ray_lod=lod, filter_mode=1)
            assert st.kernel_variant == vrc.KERNEL_RAY_LOD
            packed = ",unsigned int," in _ran(g)
            vrc.check(g.L, g.L.vrc_set_option(g.ctx, vrc.OPT_PACKED_ATLAS, 0))
            plain, n_plain, _ = g.render(ray_lod=lod, filter_mode=1)
            staged = _ran(g).startswith("vrc_k_raycast_lds<")
            assert staged == packed  # both need an overlap
            staged_cases += staged
            vrc.check(g.L, g.L.vrc_set_option(g.ctx, vrc.OPT_PACKED_ATLAS, 1))
            gathered, n_gathered, _ = g.render(ray_lod=lod, filter_mode=1, kernel=vrc.KERNEL_GRID_DDA)
            assert _ran(g).startswith("vrc_k_raycast_raylod<") and ",unsigned int," not in _ran(g)
        _lod_parity(auto, want, name + " trilinear, " + ("tap-packed atlas" if packed else "gathers (AUTO)"))
        _lod_parity(plain, want, name + " trilinear, " + ("staged" if staged else "gathers (AUTO)"))
        _lod_parity(gathered, want, name + " trilinear, gathers")
        assert abs(n_auto - n_want) <= 3e-4 * n_want + 16 and abs(n_gathered - n_want) <= 3e-4 * n_want + 16
    assert staged_cases >= 1


def test_more_bricks_than_the_reference_node_table_holds(vrc):
    # quirk Q8: the reference kernel's node table has 16384 entries and no bounds check (cuda/Renderer.cu:237,
    # 261-264); here the table is sized to the list: 32768 bricks of 8^3 voxels, both brick enumerations
    from test_cpu_harness import _fuzz_parity
    s = orc.build_scene(voxels=(256, 256, 256), block=8, viewport=(40, 32), volume="hash", spin=(0.4, 0.3), alpha=0.3)
    assert s.n_nodes == 32768
    want, n_want = orc.oracle_render(s, threads=8)
    with _gpu(s) as g:
        for k in (vrc.KERNEL_GRID_DDA, vrc.KERNEL_REFERENCE_ORDER):
            got, n_got, st = g.render(kernel=k)
            assert st.kernel_variant == k
            # 8-voxel bricks: a brick entry (the reference's sample on the face) every few samples
            _fuzz_parity(got, want, "32768 bricks k%d" % k)
            assert abs(n_got - n_want) <= 3e-4 * n_want + 16
        # the two enumerations composite the same samples: where a ray leaves a cell through an edge or a
        # corner of the brick grid, the cell walk hands the bricks around it to the reference's slab test too
        dda, n_dda, _ = g.render(kernel=vrc.KERNEL_GRID_DDA)
        ref, n_ref, _ = g.render(kernel=vrc.KERNEL_REFERENCE_ORDER)
        assert n_dda == n_ref and np.abs(dda - ref).max() <= 1e-6


def test_gather_tiles_places_bands_at_their_rows(vrc):
    # vrc_gather_tiles (the sort-first exchange of livre/eq/Channel.cpp:519-523 behind the C ABI) for a world
    # of one rank: every band of the rank's stacked buffer lands at its rows of the frame, for a batch of frames;
    # argument errors are reported before anything is queued.  N > 1 needs N GPUs (the driver's scaling run).
    import torch
    L = vrc.load_library()
    ctx, comm = C.c_void_p(), C.c_void_p()
    vrc.check(L, L.vrc_ctx_create(0, C.byref(ctx)))
    vrc.check(L, L.vrc_comm_create(ctx, 0, 1, None, C.byref(comm)))
    r, w = C.c_int(-1), C.c_int(-1)
    vrc.check(L, L.vrc_comm_info(comm, C.byref(r), C.byref(w)))
    assert (r.value, w.value) == (0, 1)
    W, H, F = 24, 20, 3
    bands = [(0, 3), (7, 5), (12, 1), (15, 4)]  # (frame_row, rows): 13 of 20 rows, the rest stay untouched
    arr = (vrc.Band * len(bands))(*[vrc.Band(0, y0, h) for y0, h in bands])
    rows = sum(h for _, h in bands)
    local = torch.arange(F * rows * W * 4, dtype=torch.float32, device="cuda").reshape(F, rows, W, 4)
    frame = torch.full((F, H, W, 4), -1.0, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    vrc.check(L, L.vrc_gather_tiles(ctx, comm, arr, len(bands), W, H, F, local.data_ptr(), rows * W * 16,
                                    frame.data_ptr(), H * W * 16, 0, None))
    vrc.check(L, L.vrc_synchronize(ctx))
    want = torch.full((F, H, W, 4), -1.0)
    off = 0
    for y0, h in bands:
        want[:, y0:y0 + h] = local[:, off:off + h].cpu()
        off += h
    assert (frame.cpu() == want).all()
    # errors: band of a rank outside the communicator, missing frame on the display rank, bad root
    bad = (vrc.Band * 1)(vrc.Band(1, 0, 1))
    assert L.vrc_gather_tiles(ctx, comm, bad, 1, W, H, 1, local.data_ptr(), 0, frame.data_ptr(), 0, 0, None) == vrc.VRC_EINVAL
    assert L.vrc_gather_tiles(ctx, comm, arr, len(bands), W, H, 1, local.data_ptr(), 0, None, 0, 0, None) == vrc.VRC_EINVAL
    assert L.vrc_gather_tiles(ctx, comm, arr, len(bands), W, H, 1, local.data_ptr(), 0, frame.data_ptr(), 0, 1, None) == vrc.VRC_EINVAL
    # a band that reaches past the frame -- also when frame_row + rows wraps in 32 bits -- and a frame stride that
    # does not hold a frame are refused before anything is queued (round-2 advisor finding)
    for y0, h in ((H - 1, 2), (0xFFFFFFFF, 2), (H, 0xFFFFFFFF)):
        past = (vrc.Band * 1)(vrc.Band(0, y0, h))
        assert L.vrc_gather_tiles(ctx, comm, past, 1, W, H, 1, local.data_ptr(), 0, frame.data_ptr(), 0, 0, None) == vrc.VRC_EINVAL
    assert L.vrc_gather_tiles(ctx, comm, arr, len(bands), W, H, F, local.data_ptr(), rows * W * 16,
                              frame.data_ptr(), H * W * 16 - 16, 0, None) == vrc.VRC_EINVAL
    assert (frame.cpu() == want).all()  # untouched by the refused calls
    # on a caller's stream the exchange is ordered behind the context's render stream
    st = torch.cuda.Stream()
    frame.fill_(-1.0)
    torch.cuda.synchronize()
    vrc.check(L, L.vrc_gather_tiles(ctx, comm, arr, len(bands), W, H, F, local.data_ptr(), rows * W * 16,
                                    frame.data_ptr(), H * W * 16, 0, C.c_void_p(st.cuda_stream)))
    st.synchronize()
    assert (frame.cpu() == want).all()
    # a communicator of more ranks needs an id (and RCCL)
    c2 = C.c_void_p()
    assert L.vrc_comm_create(ctx, 0, 2, None, C.byref(c2)) == vrc.VRC_ECOMM
    uid = C.create_string_buffer(vrc.COMM_ID_BYTES)
    vrc.check(L, L.vrc_comm_unique_id(uid))  # RCCL is on the box: the id comes from ncclGetUniqueId
    assert any(uid.raw)
    L.vrc_comm_destroy(comm)
    L.vrc_ctx_destroy(ctx)


def test_kernel_timing_can_be_switched_off(vrc):
    # VRC_OPT_KERNEL_TIMING = 0: no event pair around the launch; the frame and the sample counter are unaffected
    s = scenes.get("hash64_spin")
    with _gpu(s) as g:
        fb, n, st = g.render()
        assert st.kernel_launches == 1 and st.kernel_ms > 0.0
        vrc.check(g.L, g.L.vrc_set_option(g.ctx, vrc.OPT_KERNEL_TIMING, 0))
        fb2, n2, st2 = g.render()
        assert (fb2 == fb).all() and n2 == n and st2.kernel_launches == 0 and st2.kernel_ms_sum == 0.0
        vrc.check(g.L, g.L.vrc_set_option(g.ctx, vrc.OPT_KERNEL_TIMING, 1))
        _, _, st3 = g.render()
        assert st3.kernel_launches == 1 and st3.kernel_ms > 0.0


def test_depth_split_composites_the_same_samples(vrc):
    # VRC_OPT_DEPTH_SPLIT: two waves per tile (near / far half of every ray) composited with `over`: the sample set
    # is the plain kernel's (counts equal), the frame equal up to the regrouped float additions, the oracle rule
    # holds; with an opaque transfer function (early termination possible) and on later passes of a multipass
    # frame the option falls back to the plain kernel, bit for bit
    for name in ("hash64_spin", "mem64_axis", "hash_clip", "mem_inside"):
        s = scenes.get(name)
        want, n_want = orc.oracle_render(s, threads=8)
        with _gpu(s) as g:
            plain, n_plain, _ = g.render()
            vrc.check(g.L, g.L.vrc_set_option(g.ctx, vrc.OPT_DEPTH_SPLIT, 1))
            split, n_split, st = g.render()
            assert st.kernel_variant == vrc.KERNEL_GRID_DDA and n_split == n_plain == n_want
            assert np.abs(split - plain).max() <= 2e-6 and not (split == plain).all()  # really another kernel
            scenes.assert_parity(split, want, name + " depth split")
            # multipass: the first pass may be split (the frame starts from zero), later ones accumulate in
            h = s.n_nodes // 2
            two, n_two, _ = g.render(passes=[(0, h), (h, s.n_nodes)])
            scenes.assert_parity(two, want, name + " depth split, two passes")
            assert n_two == n_want
    s = scenes.get("hash64_ert")  # alpha 1.0: early termination happens, the split would not be exact
    with _gpu(s) as g:
        plain, n_plain, _ = g.render()
        vrc.check(g.L, g.L.vrc_set_option(g.ctx, vrc.OPT_DEPTH_SPLIT, 1))
        again, n_again, _ = g.render()
        assert (again == plain).all() and n_again == n_plain
    # the judged shape: C2's 136^3 slots with noise, one whole frame
    s = orc.build_scene(voxels=(256, 256, 256), block=128, viewport=(256, 256), volume="hash", spin=(0.5236, 0.349))
    want, n_want = orc.oracle_render(s, threads=16)
    with _gpu(s) as g:
        vrc.check(g.L, g.L.vrc_set_option(g.ctx, vrc.OPT_DEPTH_SPLIT, 1))
        split, n_split, _ = g.render()
        scenes.assert_parity(split, want, "136^3 noise, depth split")
        assert n_split == n_want


def test_depth_split_at_the_opacity_bound(vrc):
    # the split is taken only where early ray termination cannot occur: (1 - largest classified alpha)^(most samples of
    # a ray) above 1 - 0.999 with a margin of 0.01 in the logarithm (the kernel accumulates in float, round-2 advisor
    # finding).  Transfer functions just inside the bound, between the bound with and without the margin, and just
    # outside: the first renders with the split kernel (same samples), the other two with the plain kernel, bit for bit
    import math
    kw = dict(scenes.SCENES["hash64_spin"])
    probe = orc.build_scene(**kw)
    spr, grid = probe.render.samplesPerRay, 64 // 16
    n_max = 1.7320508 * spr + 3.0 * (3 * grid) + 8.0

    def alpha_at(log_bound):  # the transfer-function opacity whose classified alpha sits exactly at the bound
        m = 1.0 - math.exp(log_bound / n_max)
        return 1.0 - (1.0 - m) ** (spr / 32.0)

    with_margin, without = alpha_at(math.log(1.0 - 0.999) + 0.01), alpha_at(math.log(1.0 - 0.999))
    assert 0.05 < with_margin < without < 0.2
    for alpha, split_expected in ((0.99 * with_margin, True), (0.5 * (with_margin + without), False), (1.01 * without, False)):
        s = orc.build_scene(**dict(kw, alpha=alpha))
        with _gpu(s) as g:
            plain, n_plain, _ = g.render()
            assert _ran(g).startswith("vrc_k_raycast<")
            vrc.check(g.L, g.L.vrc_set_option(g.ctx, vrc.OPT_DEPTH_SPLIT, 1))
            frame, n_frame, _ = g.render()
            assert _ran(g).startswith("vrc_k_raycast_split<") == split_expected, (alpha, _ran(g))
            assert n_frame == n_plain
            if split_expected:
                assert np.abs(frame - plain).max() <= 2e-6
                assert plain[..., 3].max() < 0.999  # no ray near the threshold: the bound is conservative
            else:
                assert (frame == plain).all()


def _ray_counts(g):
    from libre_amd import vrc
    counts = (C.c_uint32 * 8)()
    parts = C.c_int()
    vrc.check(g.L, g.L.vrc_get_ray_counts(g.ctx, C.byref(counts), C.byref(parts)))
    return list(counts), parts.value


def test_ray_compaction_is_bit_identical(vrc):
    # VRC_OPT_ERT_COMPACTION = P: the march in P launches; after each a wave ballot packs the rays that early
    # termination (Renderer.cu:219-226) has not ended into a list, the next launch marches 64 live rays per wave.
    # Same bricks in the same order with the same arithmetic per ray: the frame and the sample count are the plain
    # kernel's, bit for bit -- with early termination (alpha 1.0), without it, with clip planes, from inside the
    # volume, on a multipass frame
    for name in ("hash64_ert", "mem64_ert", "hash64_spin", "hash_clip", "mem_inside", "mem_ragged"):
        s = scenes.get(name)
        with _gpu(s) as g:
            plain, n_plain, _ = g.render()
            assert _ray_counts(g) == ([0] * 8, 0)
            h = s.n_nodes // 2
            plain2, n_plain2, _ = g.render(passes=[(0, h), (h, s.n_nodes)])
            for parts in (2, 3, 8):
                vrc.check(g.L, g.L.vrc_set_option(g.ctx, vrc.OPT_ERT_COMPACTION, parts))
                fb, n, st = g.render()
                assert st.kernel_variant == vrc.KERNEL_GRID_DDA
                assert (fb == plain).all() and n == n_plain, "%s, %d launches" % (name, parts)
                counts, used = _ray_counts(g)
                assert used == parts and counts[parts - 1:] == [0] * (9 - parts)
                assert all(counts[p] >= counts[p + 1] for p in range(parts - 2)), counts  # rays only end
                # a ray whose final opacity is below the threshold was alive after every launch
                never_ended = int(((plain[..., 3] > 0) & (plain[..., 3] <= 0.999)).sum())
                assert counts[parts - 2] >= never_ended and counts[0] <= s.W * s.H
                if name.endswith("_ert"):  # opaque transfer function: rays did end before the last launch
                    assert counts[parts - 2] < int((plain[..., 3] > 0).sum())
                fb2, n2, _ = g.render(passes=[(0, h), (h, s.n_nodes)])
                assert (fb2 == plain2).all() and n2 == n_plain2, "%s, %d launches, two passes" % (name, parts)
            vrc.check(g.L, g.L.vrc_set_option(g.ctx, vrc.OPT_ERT_COMPACTION, 0))
            fb, n, _ = g.render()
            assert (fb == plain).all() and _ray_counts(g)[1] == 0
            assert g.L.vrc_set_option(g.ctx, vrc.OPT_ERT_COMPACTION, 9) != 0  # more launches than the list has counters
    # other kernels ignore the option: reference order, trilinear filter
    s = scenes.get("hash64_ert")
    with _gpu(s) as g:
        vrc.check(g.L, g.L.vrc_set_option(g.ctx, vrc.OPT_ERT_COMPACTION, 4))
        for kw in (dict(kernel=vrc.KERNEL_REFERENCE_ORDER), dict(filter_mode=1)):
            g.render(**kw)
            assert _ray_counts(g)[1] == 0
    # the judged shape (136^3 slots, noise), an opaque transfer function: rays end inside the first bricks
    s = orc.build_scene(voxels=(256, 256, 256), block=128, viewport=(256, 256), volume="hash", spin=(0.5236, 0.349),
                        alpha=1.0)
    want, n_want = orc.oracle_render(s, threads=16)
    with _gpu(s) as g:
        plain, n_plain, _ = g.render()
        vrc.check(g.L, g.L.vrc_set_option(g.ctx, vrc.OPT_ERT_COMPACTION, 4))
        fb, n, _ = g.render()
        counts, _ = _ray_counts(g)
        # (against the oracle: a ray that crosses the threshold within float rounding of 0.999 ends a sample apart)
        assert (fb == plain).all() and n == n_plain and abs(n - n_want) <= 1e-5 * n_want + 8
        scenes.assert_parity(fb, want, "136^3 noise, 4 launches")
        ended, hits = int((plain[..., 3] > 0.999).sum()), int((plain[..., 3] > 0).sum())
        assert ended > 0 and counts[0] >= counts[2] and counts[2] <= hits - ended // 2, (counts, ended, hits)


def test_grey_table_form_is_bit_identical(vrc):
    # VRC_OPT_GREY_TABLE (default on): a transfer function with r == g == b in every entry lets the point-sampling
    # grid-walk kernel keep (grey, alpha) instead of four floats per table entry and colour; the three colour
    # channels of the reference's blend are then the same operations on the same numbers, so every bit of the frame
    # is the four-float kernel's -- with early termination, clip planes, from inside the volume, in a multipass frame
    # (first pass grey, later passes four floats), and at the judged 136^3-slot shape
    cases = [scenes.get(n) for n in ("hash64_spin", "hash64_ert", "mem64_axis", "hash_clip", "mem_inside", "mem_ragged")]
    cases.append(orc.build_scene(voxels=(256, 256, 256), block=128, viewport=(256, 256), volume="hash",
                                 spin=(0.5236, 0.349), alpha=0.3))
    for s in cases:
        assert (s.tf[:, 0] == s.tf[:, 1]).all() and (s.tf[:, 0] == s.tf[:, 2]).all()  # the linear ramp is grey
        with _gpu(s) as g:
            vrc.check(g.L, g.L.vrc_set_option(g.ctx, vrc.OPT_GREY_TABLE, 0))
            four, n_four, _ = g.render()
            h = s.n_nodes // 2
            four2, n_four2, _ = g.render(passes=[(0, h), (h, s.n_nodes)])
            vrc.check(g.L, g.L.vrc_set_option(g.ctx, vrc.OPT_GREY_TABLE, 1))
            grey, n_grey, st = g.render()
            assert st.kernel_variant == vrc.KERNEL_GRID_DDA
            assert (grey == four).all() and n_grey == n_four
            assert (grey[..., 0] == grey[..., 1]).all() and (grey[..., 0] == grey[..., 2]).all()
            grey2, n_grey2, _ = g.render(passes=[(0, h), (h, s.n_nodes)])
            assert (grey2 == four2).all() and n_grey2 == n_four2
            # the reference-order kernel has the grey form too
            ref_grey, n_ref_grey, st = g.render(kernel=vrc.KERNEL_REFERENCE_ORDER)
            assert st.kernel_variant == vrc.KERNEL_REFERENCE_ORDER
            vrc.check(g.L, g.L.vrc_set_option(g.ctx, vrc.OPT_GREY_TABLE, 0))
            ref_four, n_ref_four, _ = g.render(kernel=vrc.KERNEL_REFERENCE_ORDER)
            assert (ref_grey == ref_four).all() and n_ref_grey == n_ref_four
    # a coloured transfer function: the option changes nothing, the frame is the oracle's
    s = scenes.get("hash64_spin")
    i = np.arange(256, dtype=np.float32) / np.float32(255.0)
    s.tf = np.ascontiguousarray(np.stack([i, i * i, np.float32(1.0) - i, np.float32(0.3) * i], axis=1))
    want, n_want = orc.oracle_render(s, threads=8)
    with _gpu(s) as g:
        got, n_got, _ = g.render()
        scenes.assert_parity(got, want, "coloured transfer function")
        assert n_got == n_want and not (got[..., 0] == got[..., 1]).all()
        vrc.check(g.L, g.L.vrc_set_option(g.ctx, vrc.OPT_GREY_TABLE, 0))
        again, _, _ = g.render()
        assert (again == got).all()


def test_uint16_point_sampling_grouped_march_and_its_grey_form(vrc):
    # 16-bit voxels cannot index the 257-entry classified table: every sample is classified on its own.  With overlap
    # (no clamped sampler) and fixed-point stepping the march is the table form's (whole groups, address tables) with
    # the table read replaced by the classification; a grey transfer function takes (grey, alpha) pairs.  The frame
    # matches the oracle, the grey form equals the four-float form bit for bit, and the float-stepping form (the
    # general per-sample march) composites the same samples
    for kw in (dict(voxels=(64, 64, 64), block=16, viewport=(48, 40), volume="hash", spin=(0.5, 0.35), dtype="u16"),
               dict(voxels=(64, 64, 64), block=16, viewport=(40, 40), volume="hash", spin=(0.3, -0.2), alpha=1.0,
                    dtype="u16"),
               dict(voxels=(256, 256, 256), block=128, viewport=(160, 160), volume="hash", spin=(0.5236, 0.349),
                    alpha=0.3, dtype="u16")):
        s = orc.build_scene(**kw)
        want, n_want = orc.oracle_render(s, threads=16)
        with _gpu(s) as g:
            for k in (vrc.KERNEL_GRID_DDA, vrc.KERNEL_REFERENCE_ORDER):
                grey, n_grey, _ = g.render(kernel=k)
                scenes.assert_parity(grey, want, "u16 %r kernel %d" % (kw, k))
                assert abs(n_grey - n_want) <= 3e-4 * n_want + 16
                vrc.check(g.L, g.L.vrc_set_option(g.ctx, vrc.OPT_GREY_TABLE, 0))
                four, n_four, _ = g.render(kernel=k)
                vrc.check(g.L, g.L.vrc_set_option(g.ctx, vrc.OPT_GREY_TABLE, 1))
                assert (four == grey).all() and n_four == n_grey
                flt, n_flt, _ = g.render(kernel=k, stepping=0)
                scenes.assert_parity(flt, want, "u16 %r kernel %d, float stepping" % (kw, k))
                assert n_flt == n_grey


def test_lds_kernel_grey_form_is_bit_identical(vrc):
    # the LDS-staged kernel (the fast path of the trilinear filter) with (grey, alpha) pairs for a grey transfer
    # function: the same bits as its four-float form, point-sampled and trilinear, first pass only
    for name in ("hash64_spin", "hash64_ert", "mem_inside", "hash_clip"):
        s = scenes.get(name)
        with _gpu(s) as g:
            for flt in (0, 1):
                vrc.check(g.L, g.L.vrc_set_option(g.ctx, vrc.OPT_GREY_TABLE, 0))
                four, n_four, st = g.render(kernel=vrc.KERNEL_LDS, filter_mode=flt)
                assert st.kernel_variant == vrc.KERNEL_LDS
                h = s.n_nodes // 2
                four2, n_four2, _ = g.render(kernel=vrc.KERNEL_LDS, filter_mode=flt, passes=[(0, h), (h, s.n_nodes)])
                vrc.check(g.L, g.L.vrc_set_option(g.ctx, vrc.OPT_GREY_TABLE, 1))
                grey, n_grey, _ = g.render(kernel=vrc.KERNEL_LDS, filter_mode=flt)
                assert (grey == four).all() and n_grey == n_four, (name, flt)
                grey2, n_grey2, _ = g.render(kernel=vrc.KERNEL_LDS, filter_mode=flt, passes=[(0, h), (h, s.n_nodes)])
                assert (grey2 == four2).all() and n_grey2 == n_four2, (name, flt)


def test_per_ray_lod_grey_form_is_bit_identical(vrc):
    # the per-ray LOD kernel with (grey, alpha) level tables: the same bits as its four-float form
    from test_ray_lod import _hierarchy
    for kw, sse in ((dict(voxels=(64, 64, 64), block=16, viewport=(48, 48), volume="hash", spin=(0.5, 0.35)), 2.0),
                    (dict(voxels=(96, 64, 64), block=16, viewport=(40, 44), volume="hash", spin=(-0.4, 0.6), alpha=1.0),
                     1.0)):
        s = _hierarchy(**kw)
        lod = (sse, orc.world_space_per_pixel(s))
        with _gpu(s) as g:
            vrc.check(g.L, g.L.vrc_set_option(g.ctx, vrc.OPT_GREY_TABLE, 0))
            four, n_four, st = g.render(ray_lod=lod)
            assert st.kernel_variant == vrc.KERNEL_RAY_LOD
            vrc.check(g.L, g.L.vrc_set_option(g.ctx, vrc.OPT_GREY_TABLE, 1))
            grey, n_grey, st = g.render(ray_lod=lod)
            assert st.kernel_variant == vrc.KERNEL_RAY_LOD
            assert (grey == four).all() and n_grey == n_four
