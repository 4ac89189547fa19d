"""Known-answer tests that pin the oracle's host-side restatement to the values the
reference's own unit tests hold.  Pixels are unpinned by the reference."""
import ctypes as C

import numpy as np
import pytest

import orc


def test_nodeid_bit_layout():
    # tests/lib/lodSelection.cpp:83-194: 17 = level 1,x 1; 262145 = level 1,y 1;
    # 4294967297 = level 1,z 1
    assert orc.pack(1, 0, 0, 0) == 1
    assert orc.pack(1, 1, 0, 0) == 17
    assert orc.pack(1, 0, 1, 0) == 262145
    assert orc.pack(1, 1, 1, 0) == 262161
    assert orc.pack(1, 0, 0, 1) == 4294967297
    assert orc.pack(1, 1, 0, 1) == 4294967313
    assert orc.pack(1, 0, 1, 1) == 4295229441
    assert orc.pack(1, 1, 1, 1) == 4295229457
    # level-2 and level-3 ids from the same golden lists
    assert orc.pack(2, 0, 0, 2) == 8589934594
    assert orc.pack(2, 3, 3, 3) == 12885688370
    assert orc.pack(3, 0, 0, 6) == 25769803779
    assert orc.unpack(30066344035) == (3, 6, 6, 7, 0)
    for i in (0, 1, 17, 262161, 12885688370, 30066344035):
        assert orc.pack(*orc.unpack(i)) == i


def test_nodeid_children_parent():
    L = orc.lib()
    kids = (C.c_uint64 * 8)()
    L.orc_nodeid_children(orc.pack(0, 0, 0, 0), kids)
    # NodeId.cpp:92-113: x outer, y middle, z inner; golden level-1 ids
    assert list(kids) == [1, 4294967297, 262145, 4295229441, 17, 4294967313, 262161, 4295229457]
    for k in kids:
        assert L.orc_nodeid_parent(k) == 0
    assert L.orc_nodeid_parent(0) == 2**64 - 1  # INVALID_NODE_ID


def _info(voxels, max_block, overlap):
    vi = orc.VolumeInfo()
    for a in range(3):
        vi.voxels[a] = voxels[a]
        vi.maximumBlockSize[a] = max_block
        vi.overlap[a] = overlap
    orc.lib().orc_fill_regular_volume_info(C.byref(vi))
    return vi


def test_fill_regular_volume_info():
    # tests/core/volumeInformation.cpp:57-108
    vi = _info((2048, 2048, 2048), 64, 0)
    assert vi.worldSpacePerVoxel == np.float32(1.0 / 2048.0)
    assert list(vi.worldSize) == [1.0, 1.0, 1.0]
    assert vi.depth == 6 and list(vi.rootBlocks) == [1, 1, 1]
    vi = _info((2048, 2099, 2048), 64, 0)
    assert vi.depth == 6 and list(vi.rootBlocks) == [1, 2, 1]
    assert vi.worldSpacePerVoxel == np.float32(1.0) / np.float32(2099.0)
    assert list(vi.worldSize) == [np.float32(2048.0) * np.float32(np.float32(1.0) / np.float32(2099.0)),
                                  np.float32(2099.0) * np.float32(np.float32(1.0) / np.float32(2099.0)),
                                  np.float32(2048.0) * np.float32(np.float32(1.0) / np.float32(2099.0))]
    assert abs(vi.worldSize[1] - 1.0) < 1e-6 and abs(vi.worldSize[0] - 2048.0 / 2099.0) < 1e-6
    vi = _info((2048, 2037, 2048), 64, 0)
    assert vi.depth == 6 and list(vi.rootBlocks) == [1, 1, 1]
    assert vi.worldSpacePerVoxel == np.float32(1.0 / 2048.0)
    assert abs(vi.worldSize[1] - 2037.0 / 2048.0) < 1e-7
    vi = _info((2048, 2049, 2048), 64, 0)
    assert vi.depth == 6 and list(vi.rootBlocks) == [1, 1, 1]
    assert abs(vi.worldSize[0] - 2048.0 / 2049.0) < 1e-6


def test_mem_data_source():
    # tests/data/dataSource.cpp:45-70: mem://#1024,1024,512,32 -> depth 5, voxel box 32^3,
    # brick = 40^3 bytes
    vi = orc.mem_volume_info(1024, 1024, 512, 32)
    assert vi.depth == 5
    assert list(vi.maximumBlockSize) == [40, 40, 40] and list(vi.overlap) == [4, 4, 4]
    kids = (C.c_uint64 * 8)()
    orc.lib().orc_nodeid_children(orc.pack(0, 0, 0, 0), kids)
    node = orc.lod_node(vi, kids[0])
    assert [node.voxelBoxMax[a] - node.voxelBoxMin[a] for a in range(3)] == [32, 32, 32]
    assert [node.blockSize[a] + 2 * vi.overlap[a] for a in range(3)] == list(vi.maximumBlockSize)
    # tests/lib/cache.cpp:97-119: every voxel of NodeId(1,(0,0,0)) is 17 (histogram has the
    # single bin 17)
    assert orc.lib().orc_mem_brick_value_u8(kids[0]) == 17
    buf = np.zeros(40 * 40 * 40, dtype=np.uint8)
    orc.lib().orc_mem_brick_fill_u8(C.byref(vi), kids[0], buf.ctypes.data)
    assert (buf == 17).all()


def test_mem_default_volume_depths():
    # tests/lib/lodSelection.cpp: mem://#4096,4096,4096,256 has levels 0..3 in its golden lists
    vi = orc.mem_volume_info(4096, 4096, 4096, 256)
    assert vi.depth == 5 and list(vi.rootBlocks) == [1, 1, 1]
    # BASELINE.md C1/C2
    vi = orc.mem_volume_info(128, 128, 128, 32)
    assert vi.depth == 3 and len(orc.leaf_ids(vi)) == 64
    vi = orc.mem_volume_info(1024, 1024, 1024, 128)
    assert vi.depth == 4 and len(orc.leaf_ids(vi)) == 512
    # z-invariance of the mem:// value at level 3 (quirk Q13): ((3 | x<<4) ^ (y<<2)) + 16
    for x in range(8):
        for y in range(8):
            for z in (0, 3, 7):
                assert orc.lib().orc_mem_brick_value_u8(orc.pack(3, x, y, z)) == ((3 | x << 4) ^ (y << 2)) + 16


def test_world_boxes_tile_the_volume():
    vi = orc.mem_volume_info(128, 128, 128, 32)
    n = orc.lod_node(vi, orc.pack(2, 0, 0, 0))
    assert list(n.worldBoxMin) == [-0.5, -0.5, -0.5] and list(n.worldBoxMax) == [-0.25, -0.25, -0.25]
    n = orc.lod_node(vi, orc.pack(2, 3, 1, 2))
    assert list(n.worldBoxMin) == [0.25, -0.25, 0.0] and list(n.worldBoxMax) == [0.5, 0.0, 0.25]
    assert list(n.voxelBoxMin) == [96, 32, 64]


def test_projection_matrix_matches_lodselection_fixture():
    # tests/lib/lodSelection.cpp:38-41
    want = [2.0, 0, 0, 0, 0, 2.0, 0, 0, 0, 0, -1.01342285, -1, 0, 0, -0.201342285, 0]
    got = list(orc.default_proj())
    assert np.allclose(got, want, rtol=1e-6, atol=1e-7)


TOL = 1e-5  # tests/eq/settings/cameraSettings.cpp uses BOOST_CHECK_CLOSE at 0.001 %


def test_camera_spin_model():
    # tests/eq/settings/cameraSettings.cpp:44-57
    m = orc.f32x16()
    orc.lib().orc_mat4_identity(m)
    orc.lib().orc_spin_model(m, 20.0, 20.0)
    want = [0.408082, 0.0, 0.912945, 0.0, 0.833469, 0.408082, -0.372557, 0.0,
            -0.372557, 0.912945, 0.166531, 0.0, 0.0, 0.0, 0.0, 1.0]
    assert np.allclose(list(m), want, rtol=TOL, atol=1e-6)


def test_camera_look_at():
    # tests/eq/settings/cameraSettings.cpp:99-117
    m = orc.f32x16()
    orc.lib().orc_look_at(orc.f32x3(0, 0, 0), orc.f32x3(20, 20, 20), orc.f32x3(0, 1, 0), m)
    want = [-0.707107, -0.408248, -0.57735, 0.0, 0.0, 0.816496, -0.57735, 0.0,
            0.707107, -0.408248, -0.57735, 0.0, 0.0, 0.0, 0.0, 1.0]
    assert np.allclose(list(m), want, rtol=TOL, atol=1e-6)


def test_camera_everything():
    # tests/eq/settings/cameraSettings.cpp:119-144: setCameraPosition, setCameraLookAt,
    # spinModel, moveCamera
    L = orc.lib()
    pos = (0.5, 1.17, 6.78)
    m = orc.f32x16()
    L.orc_look_at(orc.f32x3(*pos), orc.f32x3(13.52, 123.53, 21.12), orc.f32x3(0, 1, 0), m)
    L.orc_spin_model(m, 13.54, 21.49)
    m[12] += 13.54
    m[13] += 21.49
    m[14] += 33.25
    want = [0.413936, -0.460246, -0.785385, 0.0, 0.328941, -0.728848, 0.600482, 0.0,
            -0.848796, -0.506907, -0.150303, 0.0, 9.3526, 26.597, 35.243, 1.0]
    assert np.allclose(list(m), want, rtol=1e-4, atol=1e-4)


def test_default_view_data():
    vi = orc.mem_volume_info(128, 128, 128, 32)
    s = orc.build_scene(voxels=(128, 128, 128), block=32, viewport=(8, 8))
    assert np.allclose(list(s.view.eyePosition), [0, 0, 1.5], atol=1e-7)
    assert abs(s.view.nearPlane - 0.1) < 1e-6
    assert list(s.view.aabbMin) == [-0.5, -0.5, -0.5] and list(s.view.aabbMax) == [0.5, 0.5, 0.5]
    assert s.render.samplesPerRay == 512  # CudaRaycastRenderer.cpp:113-129, BASELINE.md C1
    assert vi.depth == 3


def test_samples_per_ray_auto():
    s = orc.build_scene(voxels=(64, 64, 64), block=16, viewport=(4, 4))
    assert s.render.samplesPerRay == 512
    vi = orc.mem_volume_info(1024, 1024, 1024, 128)
    ids = orc.leaf_ids(vi)
    arr = (C.c_uint64 * len(ids))(*ids)
    assert orc.lib().orc_computed_samples_per_ray(C.byref(vi), arr, len(ids), 0) == 1024  # C2
    assert orc.lib().orc_computed_samples_per_ray(C.byref(vi), arr, len(ids), 300) == 300


def test_pool_slot_order_and_texture_object():
    L = orc.lib()
    slots = orc.u32x3()
    L.orc_pool_slots(orc.u32x3(40, 40, 40), 64000, 64 * 64000, orc.u32x3(4096, 4096, 4096), slots)
    assert list(slots) == [64, 1, 1]
    L.orc_pool_slots(orc.u32x3(136, 136, 136), 136 ** 3, 600 * 136 ** 3,
                     orc.u32x3(4096, 4096, 4096), slots)
    assert list(slots) == [30, 20, 1]  # cuda/TexturePool.cu:128-135
    slot = orc.f32x3()
    L.orc_pool_kth_slot(slots, 0, slot)
    assert list(slot) == [0, 0, 0]
    L.orc_pool_kth_slot(slots, 1, slot)  # k (z) innermost, then j
    assert list(slot) == [0.0, np.float32(1) / np.float32(20), 0.0]
    L.orc_pool_kth_slot(slots, 20, slot)
    assert list(slot) == [np.float32(1) / np.float32(30), 0.0, 0.0]
    origin = orc.u32x3()
    L.orc_pool_slot_voxel_origin(slots, orc.u32x3(136, 136, 136), slot, origin)
    assert list(origin) == [136, 0, 0]


def test_tf_fetch_and_composite():
    L = orc.lib()
    tf = orc.linear_ramp_tf(1.0)
    out = (C.c_float * 4)()
    # u = d/255 with the (0,255) range: texel d is hit (almost) exactly
    L.orc_tf_fetch(tf.ctypes.data, C.c_float(0.0), 8, out)
    assert list(out) == [0, 0, 0, 0]
    L.orc_tf_fetch(tf.ctypes.data, C.c_float(1.0), 8, out)
    assert list(out) == [1, 1, 1, 1]
    L.orc_tf_fetch(tf.ctypes.data, C.c_float(128.0 / 255.0), 8, out)
    assert abs(out[0] - 128.0 / 255.0) < 1e-4
    # 8-bit weight vs exact float weight differ by at most one weight step * slope
    a, b = (C.c_float * 4)(), (C.c_float * 4)()
    for d in range(256):
        L.orc_tf_fetch(tf.ctypes.data, C.c_float(d / 255.0), 8, a)
        L.orc_tf_fetch(tf.ctypes.data, C.c_float(d / 255.0), 0, b)
        assert abs(a[0] - b[0]) <= (1.0 / 255.0) / 256.0 + 1e-7
    # composite: Renderer.cu:83-93; alpha clamped at 255/256 before correction
    src = (C.c_float * 4)(1.0, 0.5, 0.25, 1.0)
    dst = (C.c_float * 4)(0, 0, 0, 0)
    L.orc_composite(src, dst, C.c_float(1.0))
    assert abs(dst[3] - 255.0 / 256.0) < 1e-6 and abs(dst[0] - 255.0 / 256.0) < 1e-6
    assert abs(dst[1] - 0.5 * 255.0 / 256.0) < 1e-6


@pytest.mark.parametrize("spin", [(0.0, 0.0), (0.5, 0.35)])
def test_oracle_frame_sanity(spin):
    s = orc.build_scene(voxels=(64, 64, 64), block=16, viewport=(48, 48), spin=spin)
    fb, n = orc.oracle_render(s, threads=4)
    assert n > 0 and np.isfinite(fb).all()
    assert fb[..., 3].max() < 0.999 and fb[..., 3].min() >= 0.0
    # rays that miss leave the cleared pixel untouched (quirk Q17)
    if spin != (0.0, 0.0):
        assert (fb[0, 0] == 0).all()
    # single-thread and multi-thread agree bit for bit
    fb1, n1 = orc.oracle_render(s, threads=1)
    assert n1 == n and (fb1 == fb).all()
