"""Named small scenes shared by the CPU-harness tests, the GPU parity tests and the golden
fixture generator.  Sizes are chosen so the oracle finishes each in well under a second."""
import ctypes as C
import os

import numpy as np

import orc

# tolerance of the parity contract: per-channel max-abs <= 2e-3, mean-abs
# <= 5e-5 on RGBA in [0,1], at most 0.1 % of pixels over 2e-3.  The residual comes from
# (a) brick slivers the DDA steps over (<= 1 sample per brick crossing), (b) nearest-voxel
# flips for samples within ~1e-4 voxel of a voxel face, (c) FMA contraction on the GPU.
# VRC_FUZZ_SCALE=n multiplies the number of seeds of every randomized test (soak runs)
FUZZ_SCALE = max(1, int(os.environ.get("VRC_FUZZ_SCALE", "1")))

# A soak run meets the rare seeds whose first sample of a brick -- which the reference puts exactly on the
# brick's face -- sits on a voxel boundary to the last bit in several pixels at once: the oracle's and the
# kernel's rounding pick different voxels there (one sample's weight, up to ~2e-2 with an opaque transfer
# function on noise).  The randomized checks widen by this factor when VRC_FUZZ_SCALE > 1.
SOAK_SLACK = 1.0 if FUZZ_SCALE == 1 else 4.0

MAX_ABS = 2e-3
MEAN_ABS = 5e-5
MAX_OVER = 1e-3

SCENES = {
    # name: kwargs of orc.build_scene
    "mem64_axis": dict(voxels=(64, 64, 64), block=16, viewport=(48, 48)),
    "mem64_spin": dict(voxels=(64, 64, 64), block=16, viewport=(48, 48), spin=(0.5, 0.35)),
    "hash64_axis": dict(voxels=(64, 64, 64), block=16, viewport=(48, 48), volume="hash"),
    "hash64_spin": dict(voxels=(64, 64, 64), block=16, viewport=(40, 56), volume="hash", spin=(0.5, 0.35)),
    "hash64_ert": dict(voxels=(64, 64, 64), block=16, viewport=(48, 48), volume="hash", spin=(0.3, -0.2), alpha=1.0),
    "mem64_ert": dict(voxels=(64, 64, 64), block=16, viewport=(48, 48), spin=(0.3, -0.2), alpha=1.0),
    "mem_ragged": dict(voxels=(96, 64, 32), block=16, viewport=(37, 29), spin=(0.2, 0.9)),
    "hash_spr300": dict(voxels=(64, 64, 64), block=32, viewport=(33, 31), volume="hash", spin=(0.1, 0.2), spr=300),
    "hash_clip": dict(voxels=(64, 64, 64), block=16, viewport=(48, 48), volume="hash", spin=(0.5, 0.35),
                      planes=[[-1, 0, 0, 0.2], [0, 1, 0, 0.3], [0.6, 0, 0.8, 0.35]]),
    "mem_inside": dict(voxels=(64, 64, 64), block=16, viewport=(32, 32), eye=(0.1, 0.05, 0.2)),
}


def get(name):
    return orc.build_scene(**SCENES[name])


def nucleon_scene(viewport=(48, 48), spin=(0.4, 0.3), alpha=0.3):
    """raw:// style single brick: the 41^3 u8 fixture of the reference's tests
    (tests/lib/nucleon.raw; datasources/raw/RawDataSource.cpp:85-87: depth 1, overlap 0, one
    brick = whole volume).  Exercises overlap 0, a block that is not a multiple of 8 and the
    clamped sampler."""
    import os
    raw = np.fromfile(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "nucleon.raw"),
                      dtype=np.uint8)
    assert raw.size == 41 ** 3
    vol = raw.reshape(41, 41, 41)
    L = orc.lib()
    s = orc.Scene()
    vi = orc.VolumeInfo()
    for a in range(3):
        vi.voxels[a] = 41
        vi.maximumBlockSize[a] = 41
        vi.overlap[a] = 0
    L.orc_fill_regular_volume_info(C.byref(vi))
    assert vi.depth == 1
    s.vi = vi
    nid = orc.pack(0, 0, 0, 0)
    s.ids = [nid]
    s.sorted_ids = [nid]
    s.slot_dim = [48, 48, 48]
    s.slots = [1, 1, 1]
    s.pool_bytes = 48 ** 3
    s.atlas_dim = [48, 48, 48]
    s.atlas = np.zeros((48, 48, 48), dtype=np.uint8)
    s.atlas[:41, :41, :41] = vol
    # clamp addressing at the brick border for the oracle's atlas == what the clamped
    # sampler reads (replicate the last voxel into the slot padding)
    s.atlas[41:, :, :] = s.atlas[40:41, :, :]
    s.atlas[:, 41:, :] = s.atlas[:, 40:41, :]
    s.atlas[:, :, 41:] = s.atlas[:, :, 40:41]
    s.bricks = {nid: np.ascontiguousarray(vol)}
    s.slot_of = {nid: (0.0, 0.0, 0.0)}
    node = orc.lod_node(vi, nid)
    s.lod = {nid: node}
    s.W, s.H = viewport
    s.mv = orc.default_mv(spin)
    s.proj = orc.default_proj()
    s.view = orc.ViewData()
    L.orc_make_view_data(s.mv, s.proj, (C.c_uint32 * 4)(0, 0, s.W, s.H), C.byref(vi), C.byref(s.view))
    s.nodes = (orc.NodeData * 1)()
    tp, ts = orc.f32x3(), orc.f32x3()
    L.orc_texture_object(C.byref(vi), C.byref(node), orc.f32x3(0, 0, 0), orc.u32x3(48, 48, 48), tp, ts)
    for a in range(3):
        s.nodes[0].textureMin[a] = tp[a]
        s.nodes[0].textureSize[a] = ts[a]
        s.nodes[0].aabbMin[a] = node.worldBoxMin[a]
        s.nodes[0].aabbSize[a] = node.worldBoxMax[a] - node.worldBoxMin[a]
    s.n_nodes = 1
    s.render = orc.RenderData(512, 1, 32, 0, (C.c_float * 2)(0.0, 255.0))
    s.tf = orc.linear_ramp_tf(alpha)
    s.planes = np.zeros((0, 4), dtype=np.float32)
    return s


def assert_parity(got, want, what=""):
    mx, mean, over = orc.compare(got, want)
    assert mx <= MAX_ABS and mean <= MEAN_ABS and over <= MAX_OVER, \
        "%s: max|d|=%.3g mean|d|=%.3g over=%.4f" % (what, mx, mean, over)
    return mx, mean, over
