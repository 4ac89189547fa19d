"""ctypes binding of the CPU oracle (oracle/livre_oracle.c) + scene builders for the tests.

TEST INFRASTRUCTURE.  Everything here goes through the oracle's restatement of the
reference's host-side derivation (NodeId, LODNode, texture pool slots, texture objects,
sorting, view data), so the parity tests feed the HIP path and the oracle the same inputs.
"""
import ctypes as C
import os
import subprocess
import weakref

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "_build", "liblivre_oracle.so")
HARNESS_DIR = os.path.join(ROOT, "tests", "cpu_harness")
HARNESS_SO = os.path.join(HARNESS_DIR, "libharness.so")

u32x3 = C.c_uint32 * 3
f32x3 = C.c_float * 3
f32x16 = C.c_float * 16


class NodeData(C.Structure):  # cuda/Renderer.cuh:35-41
    _fields_ = [("textureMin", f32x3), ("textureSize", f32x3),
                ("aabbMin", f32x3), ("aabbSize", f32x3)]


class ViewData(C.Structure):  # cuda/Renderer.cuh:46-56
    _fields_ = [("eyePosition", f32x3), ("glViewport", C.c_uint32 * 4),
                ("invProjMatrix", f32x16), ("modelViewMatrix", f32x16),
                ("invViewMatrix", f32x16), ("aabbMin", f32x3), ("aabbMax", f32x3),
                ("nearPlane", C.c_float)]


class RenderData(C.Structure):  # cuda/Renderer.cuh:59-66
    _fields_ = [("samplesPerRay", C.c_uint32), ("samplesPerPixel", C.c_uint32),
                ("maxSamplesPerRay", C.c_uint32), ("datatype", C.c_uint32),
                ("dataSourceRange", C.c_float * 2)]


class VolumeInfo(C.Structure):
    _fields_ = [("voxels", u32x3), ("maximumBlockSize", u32x3), ("overlap", u32x3),
                ("worldSize", f32x3), ("worldSpacePerVoxel", C.c_float),
                ("depth", C.c_uint32), ("rootBlocks", u32x3)]


class LODNode(C.Structure):
    _fields_ = [("nodeId", C.c_uint64), ("blockSize", u32x3), ("voxelBoxMin", u32x3),
                ("voxelBoxMax", u32x3), ("worldBoxMin", f32x3), ("worldBoxMax", f32x3)]


class Options(C.Structure):
    _fields_ = [("tfFracBits", C.c_int), ("filter", C.c_int), ("nThreads", C.c_int),
                ("rowBegin", C.c_uint32), ("rowEnd", C.c_uint32), ("rowStride", C.c_uint32),
                ("voxelBytes", C.c_int), ("variant", C.c_int), ("rayLod", C.c_int),
                ("lodScreenSpaceError", C.c_float), ("lodWorldSpacePerPixel", C.c_float),
                ("tieBudget", C.c_void_p), ("tieDelta", C.c_float), ("entryBias", C.c_float)]


def build_oracle():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])
    return ORACLE_SO


def build_harness(sanitize=False):
    if sanitize == "fma":
        # emulate the GPU build's floating-point contraction on the host: clang, FMA enabled,
        # contraction "fast" but honouring the pragmas in vrc_core.h (hipcc's default mode)
        out = os.path.join(HARNESS_DIR, "libharness_fma.so")
        src = os.path.join(HARNESS_DIR, "harness.cpp")
        tmp = "%s.%d.tmp" % (out, os.getpid())  # several test workers may build at once: rename is atomic
        subprocess.check_call(["/opt/rocm/lib/llvm/bin/clang++", "-O2", "-std=c++17", "-fPIC",
                               "-shared", "-mfma", "-ffp-contract=fast-honor-pragmas",
                               "-Wno-unknown-pragmas", "-o", tmp, src])
        os.replace(tmp, out)
        return out
    if sanitize == "bias":
        # negative control of the parity rule: a build whose brick-entry samples always read the near-side voxel
        out = os.path.join(HARNESS_DIR, "libharness_bias.so")
        src = os.path.join(HARNESS_DIR, "harness.cpp")
        tmp = "%s.%d.tmp" % (out, os.getpid())
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
                               "-Wno-unknown-pragmas", "-DVRC_DEV_BUILD", "-DVRC_TEST_BIAS_ENTRY", "-o", tmp, src])
        os.replace(tmp, out)
        return out
    out = HARNESS_SO if not sanitize else os.path.join(HARNESS_DIR, "libharness_asan.so")
    src = os.path.join(HARNESS_DIR, "harness.cpp")
    deps = [src] + [os.path.join(ROOT, "libre_amd", "csrc", f) for f in ("vrc_core.h", "vrc_tables.h")]
    if os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(d) for d in deps):
        return out
    tmp = "%s.%d.tmp" % (out, os.getpid())  # several test workers may build at once: rename is atomic
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
           "-Wno-unknown-pragmas", "-o", tmp, src]
    if sanitize:
        cmd[1:1] = ["-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"]
    subprocess.check_call(cmd)
    os.replace(tmp, out)
    return out


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    L = C.CDLL(build_oracle())
    L.orc_nodeid_pack.restype = C.c_uint64
    L.orc_nodeid_pack.argtypes = [C.c_uint32] * 5
    L.orc_nodeid_unpack.argtypes = [C.c_uint64, C.POINTER(C.c_uint32)]
    L.orc_nodeid_parent.restype = C.c_uint64
    L.orc_nodeid_parent.argtypes = [C.c_uint64]
    L.orc_nodeid_children.argtypes = [C.c_uint64, C.POINTER(C.c_uint64)]
    L.orc_fill_regular_volume_info.argtypes = [C.POINTER(VolumeInfo)]
    L.orc_mem_volume_info.argtypes = [C.c_uint32] * 4 + [C.POINTER(VolumeInfo)]
    L.orc_lod_node_from_id.argtypes = [C.POINTER(VolumeInfo), C.c_uint64, C.POINTER(LODNode)]
    L.orc_mem_brick_value_u8.restype = C.c_uint8
    L.orc_mem_brick_value_u8.argtypes = [C.c_uint64]
    L.orc_mem_brick_fill_u8.argtypes = [C.POINTER(VolumeInfo), C.c_uint64, C.c_void_p]
    L.orc_pool_slots.argtypes = [u32x3, C.c_size_t, C.c_size_t, u32x3, u32x3]
    L.orc_pool_kth_slot.argtypes = [u32x3, C.c_uint32, f32x3]
    L.orc_pool_slot_voxel_origin.argtypes = [u32x3, u32x3, f32x3, u32x3]
    L.orc_pool_copy_to_slot_u8.argtypes = [C.c_void_p, u32x3, u32x3, C.c_void_p, u32x3]
    L.orc_texture_object.argtypes = [C.POINTER(VolumeInfo), C.POINTER(LODNode), f32x3, u32x3,
                                     f32x3, f32x3]
    L.orc_node_distance.restype = C.c_float
    L.orc_node_distance.argtypes = [f32x16, C.POINTER(LODNode)]
    L.orc_sort_nodes_front_to_back.argtypes = [C.POINTER(VolumeInfo), f32x16,
                                               C.POINTER(C.c_uint64), C.c_uint32]
    L.orc_computed_samples_per_ray.restype = C.c_uint32
    L.orc_computed_samples_per_ray.argtypes = [C.POINTER(VolumeInfo), C.POINTER(C.c_uint64),
                                               C.c_uint32, C.c_uint32]
    L.orc_mat4_identity.argtypes = [f32x16]
    L.orc_mat4_mul.argtypes = [f32x16, f32x16, f32x16]
    L.orc_mat4_inverse.restype = C.c_int
    L.orc_mat4_inverse.argtypes = [f32x16, f32x16]
    L.orc_look_at.argtypes = [f32x3, f32x3, f32x3, f32x16]
    L.orc_spin_model.argtypes = [f32x16, C.c_float, C.c_float]
    L.orc_perspective_frustum.argtypes = [C.c_float] * 6 + [f32x16]
    L.orc_make_view_data.argtypes = [f32x16, f32x16, C.c_uint32 * 4, C.POINTER(VolumeInfo),
                                     C.POINTER(ViewData)]
    L.orc_raycast.restype = C.c_uint64
    L.orc_raycast.argtypes = [C.c_void_p, u32x3, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p,
                              C.c_uint32, C.c_void_p, C.POINTER(ViewData), C.c_uint32,
                              C.POINTER(NodeData), C.POINTER(RenderData), C.POINTER(Options)]
    L.orc_tf_fetch.argtypes = [C.c_void_p, C.c_float, C.c_int, C.c_float * 4]
    L.orc_composite.argtypes = [C.c_float * 4, C.c_float * 4, C.c_float]
    _lib = L
    return L


_harness = {}


def harness(sanitize=False):
    if sanitize in _harness:
        return _harness[sanitize]
    H = C.CDLL(build_harness(sanitize))
    H.harness_render.restype = C.c_int
    H.harness_render.argtypes = [C.c_void_p, u32x3, u32x3, C.c_void_p, C.c_uint32, C.c_uint32,
                                 C.c_void_p, C.c_uint32, C.c_void_p, C.POINTER(ViewData),
                                 C.c_uint32, C.POINTER(NodeData), C.POINTER(RenderData), C.c_int,
                                 C.c_int, C.c_int, C.c_int, C.POINTER(C.c_uint64),
                                 C.POINTER(C.c_int)]
    _harness[sanitize] = H
    return H


# ---------------------------------------------------------------------------------------------
def mat16(a):
    return f32x16(*[float(v) for v in a])


def pack(level, x, y, z, t=0):
    return int(lib().orc_nodeid_pack(level, x, y, z, t))


def unpack(i):
    out = (C.c_uint32 * 5)()
    lib().orc_nodeid_unpack(i, out)
    return tuple(out)


def mem_volume_info(vx, vy, vz, block):
    vi = VolumeInfo()
    lib().orc_mem_volume_info(vx, vy, vz, block, C.byref(vi))
    return vi


def lod_node(vi, node_id):
    n = LODNode()
    lib().orc_lod_node_from_id(C.byref(vi), node_id, C.byref(n))
    return n


def leaf_ids(vi):
    """All NodeIds of the finest level (what --min-lod = --max-lod = depth-1 selects when the
    whole volume is inside the frustum)."""
    level = vi.depth - 1
    dims = [vi.rootBlocks[a] << level for a in range(3)]
    ids = []
    for x in range(dims[0]):
        for y in range(dims[1]):
            for z in range(dims[2]):
                ids.append(pack(level, x, y, z, 0))
    return ids


def linear_ramp_tf(alpha_scale=0.05):
    """BASELINE.md TF: rgba[i] = (i/255, i/255, i/255, alpha*i/255)."""
    i = np.arange(256, dtype=np.float32) / np.float32(255.0)
    tf = np.stack([i, i, i, np.float32(alpha_scale) * i], axis=1).astype(np.float32)
    return np.ascontiguousarray(tf)


def default_proj():
    """near 0.1, far 15, l/r/b/t = -/+0.05 (livre/eq/Channel.cpp:61-62 +
    tests/lib/lodSelection.cpp:38-41)."""
    m = f32x16()
    lib().orc_perspective_frustum(-0.05, 0.05, -0.05, 0.05, 0.1, 15.0, m)
    return m


def tile_proj(x0, y0, w, h, W, H):
    """Off-axis sub-frustum of pixel tile (x0,y0,w,h) of a W x H frame (the per-channel
    frustum Equalizer hands to livre/eq/Channel.cpp:151-157)."""
    l, r, b, t = -0.05, 0.05, -0.05, 0.05
    tl = l + (r - l) * x0 / W
    tr = l + (r - l) * (x0 + w) / W
    tb = b + (t - b) * y0 / H
    tt = b + (t - b) * (y0 + h) / H
    m = f32x16()
    lib().orc_perspective_frustum(tl, tr, tb, tt, 0.1, 15.0, m)
    return m


def default_mv(spin=(0.0, 0.0), eye=(0.0, 0.0, 1.5)):
    """lookAt(eye (0,0,1.5) -> origin, up +y) (ApplicationParameters.cpp:54-55), optionally
    spun (CameraSettings.cpp:35-59)."""
    m = f32x16()
    lib().orc_look_at(f32x3(*eye), f32x3(0, 0, 0), f32x3(0, 1, 0), m)
    if spin[0] != 0.0 or spin[1] != 0.0:
        lib().orc_spin_model(m, spin[0], spin[1])
    return m


def _hash_raw(vx, vy, zs, seed):
    """lowbias32(x + vx*(y + vy*z) + seed) >> 24 as float32 for the z slices zs."""
    plane = np.arange(vx * vy, dtype=np.uint64).reshape(1, vy, vx)
    idx = plane + (np.asarray(zs, dtype=np.uint64) * np.uint64(vx * vy)).reshape(-1, 1, 1)
    h = ((idx + np.uint64(seed)) & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    h ^= h >> np.uint32(16)
    h *= np.uint32(0x7FEB352D)
    h ^= h >> np.uint32(15)
    h *= np.uint32(0x846CA68B)
    h ^= h >> np.uint32(16)
    return (h >> np.uint32(24)).astype(np.float32)


def hash_volume(vx, vy, vz, seed=0x5EED, slab=16):
    """'Volume N' of SURVEY 8d: v = hash32(x + vx*(y + vy*z) + seed) >> 24, then a 3-tap box
    filter per axis (z, then y, then x; periodic; ((prev + cur) + next) / 3 in float32).
    Deterministic, build-defined (not a reference input).  Computed slab by slab in z so that the
    1024^3 volume of BASELINE C2 needs no more than its own gigabyte."""
    out = np.empty((vz, vy, vx), dtype=np.uint8)
    for z0 in range(0, vz, slab):
        z1 = min(vz, z0 + slab)
        raw = _hash_raw(vx, vy, [(z % vz) for z in range(z0 - 1, z1 + 1)], seed)
        v = (raw[:-2] + raw[1:-1] + raw[2:]) / np.float32(3.0)
        for ax in (1, 2):
            v = (np.roll(v, 1, axis=ax) + v + np.roll(v, -1, axis=ax)) / np.float32(3.0)
        out[z0:z1] = np.clip(np.floor(v), 0, 255).astype(np.uint8)
    return out


def brick_from_volume(vol, vi, node):
    """Cut brick + overlap out of a full-resolution volume (z,y,x order), clamping at the
    volume border (what a bricking data source does).  A node above the finest level takes
    every 2^k-th voxel (the hash:// source of the host library does the same)."""
    ov = [vi.overlap[a] for a in range(3)]
    shift = int(vi.depth) - 1 - int(unpack(node.nodeId)[0])
    lo = [int(node.voxelBoxMin[a]) - ov[a] for a in range(3)]
    hi = [int(node.voxelBoxMax[a]) + ov[a] for a in range(3)]
    ix = [np.clip(np.arange(lo[a], hi[a]) << shift, 0, vol.shape[2 - a] - 1) for a in range(3)]
    return np.ascontiguousarray(vol[np.ix_(ix[2], ix[1], ix[0])])


class Scene:
    pass


def build_scene(voxels=(64, 64, 64), block=16, viewport=(64, 64), spr=0, alpha=0.05,
                spin=(0.0, 0.0), volume="mem", max_slots=None, planes=None, ids=None,
                tile=None, eye=(0.0, 0.0, 1.5), max_tex3d=4096, pad8=True, dtype="u8",
                data_range=None, order=None):
    """Everything the integrator needs, derived through the oracle's restatement of the
    reference host code.  `pad8`: slot = maxBlock rounded up to 8 (the HIP atlas layout)."""
    L = lib()
    s = Scene()
    vi = mem_volume_info(voxels[0], voxels[1], voxels[2], block)
    s.vi = vi
    s.ids = list(ids) if ids is not None else leaf_ids(vi)
    n = len(s.ids)

    mb = [vi.maximumBlockSize[a] for a in range(3)]
    slot_dim = [(m + 7) // 8 * 8 for m in mb] if pad8 else mb
    s.slot_dim = slot_dim
    slot_bytes = slot_dim[0] * slot_dim[1] * slot_dim[2]
    want = max_slots if max_slots is not None else n
    # smallest budget whose slot grid (TexturePool.cu:128-135) holds `want` bricks
    slots = u32x3()
    blocks = want
    while True:
        L.orc_pool_slots(u32x3(*slot_dim), slot_bytes, blocks * slot_bytes,
                         u32x3(max_tex3d, max_tex3d, max_tex3d), slots)
        if slots[0] * slots[1] * slots[2] >= want:
            break
        blocks += 1
    s.pool_bytes = blocks * slot_bytes
    s.slots = [slots[a] for a in range(3)]
    s.atlas_dim = [s.slots[a] * slot_dim[a] for a in range(3)]
    # dtype "u16" is an EXTENSION (the reference CUDA kernel fetches unsigned char only)
    np_dtype = np.uint16 if dtype == "u16" else np.uint8
    s.atlas = np.zeros((s.atlas_dim[2], s.atlas_dim[1], s.atlas_dim[0]), dtype=np_dtype)

    vol = None
    if isinstance(volume, np.ndarray):  # caller-provided (z, y, x) volume of the scene's dtype
        vol = volume
        assert vol.dtype == np_dtype and list(vol.shape) == [voxels[2], voxels[1], voxels[0]]
    elif volume == "hash":
        vol = hash_volume(*voxels)
        if dtype == "u16":  # spread over 16 bits, keep the low byte busy
            vol = vol.astype(np.uint16) * np.uint16(257) ^ (vol.astype(np.uint16) >> np.uint16(3))
    s.bricks = {}
    s.slot_of = {}
    s.lod = {}
    size = u32x3(*mb)
    for k, nid in enumerate(s.ids):
        node = lod_node(vi, nid)
        s.lod[nid] = node
        if vol is None:
            # MemoryDataSource.cpp:54-57 computes the value in the volume's type T
            brick = np.full((mb[2], mb[1], mb[0]), L.orc_mem_brick_value_u8(nid), dtype=np_dtype)
        else:
            brick = brick_from_volume(vol, vi, node)
        s.bricks[nid] = brick
        slot = f32x3()
        L.orc_pool_kth_slot(slots, k, slot)
        origin = u32x3()
        L.orc_pool_slot_voxel_origin(slots, u32x3(*slot_dim), slot, origin)
        if dtype == "u16":
            s.atlas[origin[2]:origin[2] + mb[2], origin[1]:origin[1] + mb[1],
                    origin[0]:origin[0] + mb[0]] = brick
        else:
            L.orc_pool_copy_to_slot_u8(s.atlas.ctypes.data, u32x3(*s.atlas_dim), origin,
                                       brick.ctypes.data, size)
        s.slot_of[nid] = (slot[0], slot[1], slot[2])

    W, H = viewport
    s.W, s.H = W, H
    s.mv = default_mv(spin, eye)
    if tile is None:
        s.proj = default_proj()
        vp = (C.c_uint32 * 4)(0, 0, W, H)
    else:
        x0, y0, w, h, FW, FH = tile
        s.proj = tile_proj(x0, y0, w, h, FW, FH)
        vp = (C.c_uint32 * 4)(0, 0, w, h)
        s.W, s.H = w, h
    s.view = ViewData()
    L.orc_make_view_data(s.mv, s.proj, vp, C.byref(vi), C.byref(s.view))

    ids_arr = (C.c_uint64 * n)(*s.ids)
    L.orc_sort_nodes_front_to_back(C.byref(vi), s.mv, ids_arr, n)
    s.sorted_ids = list(ids_arr)
    if order is not None:
        # a host's own front-to-back list (bricks at nearly equal centre distance come in an order the
        # reference leaves to std::sort and to the rounding of vmmlib's transform): take it as it is, after
        # checking that it IS a front-to-back order of the oracle's distances up to that rounding
        assert sorted(order) == sorted(s.ids)
        dist = [float(L.orc_node_distance(s.mv, C.byref(s.lod[i]))) for i in order]
        assert all(b >= a - 1e-5 * max(1.0, a) for a, b in zip(dist, dist[1:])), "not a front-to-back order"
        s.sorted_ids = list(order)
    s.nodes = (NodeData * n)()
    for k, nid in enumerate(s.sorted_ids):
        node = s.lod[nid]
        tp, ts = f32x3(), f32x3()
        L.orc_texture_object(C.byref(vi), C.byref(node), f32x3(*s.slot_of[nid]),
                             u32x3(*s.atlas_dim), tp, ts)
        nd = s.nodes[k]
        for a in range(3):
            nd.textureMin[a] = tp[a]
            nd.textureSize[a] = ts[a]
            nd.aabbMin[a] = node.worldBoxMin[a]
            nd.aabbSize[a] = node.worldBoxMax[a] - node.worldBoxMin[a]
    s.n_nodes = n
    spr_eff = L.orc_computed_samples_per_ray(C.byref(vi), ids_arr, n, spr)
    if data_range is None:  # CudaRaycastRenderer.cpp:205 hard-codes (0,255); u16: the type's range
        data_range = (0.0, 65535.0) if dtype == "u16" else (0.0, 255.0)
    s.render = RenderData(spr_eff, 1, 32, 0, (C.c_float * 2)(*data_range))
    s.tf = linear_ramp_tf(alpha)
    if planes is None:
        s.planes = np.zeros((0, 4), dtype=np.float32)
    else:
        s.planes = np.ascontiguousarray(np.asarray(planes, dtype=np.float32).reshape(-1, 4))
    return s


def with_viewport(s, W, H):
    """The same scene (volume, atlas, bricks, camera) seen through another viewport."""
    import copy
    t = copy.copy(s)
    t.W, t.H = W, H
    t.view = ViewData()
    lib().orc_make_view_data(s.mv, s.proj, (C.c_uint32 * 4)(0, 0, W, H), C.byref(s.vi), C.byref(t.view))
    return t


def world_space_per_pixel(s, top=0.05, bottom=-0.05):
    """SelectVisibles.cpp:55-57: (frustum.top - frustum.bottom) / window height, for default_proj."""
    return (top - bottom) / float(s.H)


#: samples within this many voxels of a voxel face (+ the drift of the reference's own position chain, see
#: orc_options.tieBudget) count towards a pixel's tie budget
TIE_DELTA = 2.0 ** -11


#: world units by which the "every brick-entry tie the other way" frame starts its brick segments early
ENTRY_BIAS = 2e-7


def oracle_render(s, frac_bits=8, threads=8, rows=None, fb=None, filter_mode=0, variant=0, ray_lod=None,
                  budget=False, entry_bias=0.0):
    """Run the oracle integrator on a scene. Returns (rgba[H,W,4], samples), with budget=True
    (rgba, samples, tie_budget[H,W]) (orc_options.tieBudget, the per-pixel part of the parity tolerance).
    ray_lod = (screenSpaceError, worldSpacePerPixel): per-ray adaptive LOD over a node hierarchy."""
    global _LAST_SCENE
    _LAST_SCENE = s
    L = lib()
    if fb is None:
        fb = np.zeros((s.H, s.W, 4), dtype=np.float32)
    first_pass = not fb.any()  # a frame that accumulates over passes keeps adding to its budget
    prev = budget_of(fb)
    tb = prev if (prev is not None and not first_pass) else np.zeros((s.H, s.W), dtype=np.float32)
    opt = Options(frac_bits, filter_mode, threads, 0, s.H, 1, s.atlas.dtype.itemsize, variant,
                  1 if ray_lod else 0, ray_lod[0] if ray_lod else 0.0, ray_lod[1] if ray_lod else 0.0,
                  tb.ctypes.data, TIE_DELTA, entry_bias)
    if rows is not None:
        opt.rowBegin, opt.rowEnd, opt.rowStride = rows
    n = L.orc_raycast(s.atlas.ctypes.data, u32x3(*s.atlas_dim), fb.ctypes.data, s.W, s.H,
                      s.planes.ctypes.data if len(s.planes) else None, len(s.planes),
                      s.tf.ctypes.data, C.byref(s.view), s.n_nodes, s.nodes, C.byref(s.render),
                      C.byref(opt))
    if int(n) == 2 ** 64 - 1:
        raise RuntimeError("orc_raycast: the node list is not a brick hierarchy")
    if len(_BUDGETS) > 64:  # frames that are gone
        for k in [k for k, v in _BUDGETS.items() if v[0]() is None]:
            del _BUDGETS[k]
    _BUDGETS[fb.__array_interface__["data"][0]] = (weakref.ref(fb), tb)
    return (fb, int(n), tb) if budget else (fb, int(n))


def harness_render(s, kernel=2, frac_bits=8, fb=None, sanitize=False, pixel_off=(0, 0), variant=0, parts=0):
    """Run the host build of the HIP kernel's per-ray code (vrc_core.h).  parts > 1 (kernel 4 only): every ray is
    marched in that many slices, as the launches of VRC_OPT_ERT_COMPACTION do."""
    H = harness(sanitize)
    H.harness_set_parts.argtypes = [C.c_int]
    H.harness_set_parts.restype = None
    H.harness_set_parts(parts)
    if fb is None:
        fb = np.zeros((s.H, s.W, 4), dtype=np.float32)
    samples = C.c_uint64(0)
    grid_ok = C.c_int(0)
    rc = H.harness_render(s.atlas.ctypes.data, u32x3(*s.atlas_dim), u32x3(*s.slot_dim),
                          fb.ctypes.data, s.W, s.H,
                          s.planes.ctypes.data if len(s.planes) else None, len(s.planes),
                          s.tf.ctypes.data, C.byref(s.view), s.n_nodes, s.nodes,
                          C.byref(s.render), frac_bits, kernel, pixel_off[0], pixel_off[1],
                          C.byref(samples), C.byref(grid_ok), s.atlas.dtype.itemsize, variant)
    if rc != 0:
        raise RuntimeError("harness_render failed: %d" % rc)
    return fb, int(samples.value), bool(grid_ok.value)


def harness_render_ray_lod(s, ray_lod, kernel=1, frac_bits=8, sanitize=False):
    """Host build of vrc_pixel_ray_lod (vrc_core.h) + vrc_build_lod_tables (vrc_tables.h)."""
    H = harness(sanitize)
    fb = np.zeros((s.H, s.W, 4), dtype=np.float32)
    samples = C.c_uint64(0)
    ok = C.c_int(0)
    H.harness_render_ray_lod.restype = C.c_int
    rc = H.harness_render_ray_lod(C.c_void_p(s.atlas.ctypes.data), u32x3(*s.atlas_dim), u32x3(*s.slot_dim),
                                  C.c_void_p(fb.ctypes.data), C.c_uint32(s.W), C.c_uint32(s.H),
                                  C.c_void_p(s.planes.ctypes.data if len(s.planes) else None),
                                  C.c_uint32(len(s.planes)), C.c_void_p(s.tf.ctypes.data), C.byref(s.view),
                                  C.c_uint32(s.n_nodes), s.nodes, C.byref(s.render), C.c_int(frac_bits),
                                  C.c_int(kernel), C.byref(samples), C.byref(ok),
                                  C.c_int(s.atlas.dtype.itemsize), C.c_float(ray_lod[0]),
                                  C.c_float(ray_lod[1]))
    if rc != 0:
        raise RuntimeError("harness_render_ray_lod failed: %d" % rc)
    return fb, int(samples.value), bool(ok.value)


def all_level_ids(vi, levels=None):
    """NodeIds of every brick of the given levels (default: the whole tree): the resident
    hierarchy per-ray LOD renders from."""
    ids = []
    for level in (range(vi.depth) if levels is None else levels):
        dims = [vi.rootBlocks[a] << level for a in range(3)]
        for x in range(dims[0]):
            for y in range(dims[1]):
                for z in range(dims[2]):
                    ids.append(pack(level, x, y, z, 0))
    return ids


#: the scene of the last oracle_render call: compare() takes the frame's largest single-sample weight from it
_LAST_SCENE = None
#: with VRC_PARITY_STATS=<file> every comparison is appended to <file> as one JSON line (profiles/*_parity_errors.json)
_STATS_FILE = os.environ.get("VRC_PARITY_STATS")
#: tie budgets of the frames oracle_render returned, by the address of the frame's memory
_BUDGETS = {}


def flip_weight(s, level_scale=1.0):
    """Largest change one sample can make to a channel of a pixel: the largest classified alpha,
    1 - (1 - min(a, 255/256))^(maxSamplesPerRay / samplesPerRay) (cuda/Renderer.cu:88-89, 167-168); colours
    are <= 1."""
    a = np.minimum(np.asarray(s.tf, dtype=np.float64).reshape(-1, 4)[:, 3], 255.0 / 256.0)
    k = float(s.render.maxSamplesPerRay) / float(s.render.samplesPerRay) * level_scale
    return float((1.0 - (1.0 - a) ** k).max())


def budget_of(frame):
    """The tie budget (orc_options.tieBudget) of an oracle frame or of a row/column slice of one
    (frame[::64], frame[a:b]); None if `frame` did not come from oracle_render."""
    base = frame
    while isinstance(base, np.ndarray) and base.base is not None:
        base = base.base
    ent = _BUDGETS.get(base.__array_interface__["data"][0]) if isinstance(base, np.ndarray) else None
    if ent is None or ent[0]() is not base:
        return None
    tb = ent[1]
    if frame is base:
        return tb
    # the same rows / columns of the budget as the view takes of the frame
    off = frame.__array_interface__["data"][0] - base.__array_interface__["data"][0]
    r0, rem = divmod(off, base.strides[0])
    c0 = rem // base.strides[1]
    rs, cs = frame.strides[0] // base.strides[0], frame.strides[1] // base.strides[1]
    out = tb[r0::rs, c0::cs][:frame.shape[0], :frame.shape[1]]
    assert out.shape == frame.shape[:2], (out.shape, frame.shape)
    return out


def note_stat(**kw):
    """Append a record of the calling test to the VRC_PARITY_STATS file (tools/parity_summary.py)."""
    if _STATS_FILE:
        import inspect
        import json
        fr = inspect.stack()
        who = next((f.function for f in fr[1:] if f.function.startswith("test_")), fr[1].function)
        with open(_STATS_FILE, "a") as f:
            f.write(json.dumps(dict(test=who, note=True, **kw)) + "\n")


def compare(a, b):
    """(max-abs, mean-abs, fraction of pixels with any channel over 2e-3)."""
    d = np.abs(a.astype(np.float64) - b.astype(np.float64))
    over = (d.max(axis=-1) > 2e-3).mean()
    if _STATS_FILE:
        import inspect
        import json
        pix = d.max(axis=-1)
        small = pix <= 1e-4
        fr = inspect.stack()
        who = next((f.function for f in fr[1:] if f.function.startswith("test_")), fr[1].function)
        tb = budget_of(b)
        if tb is None:
            tb = budget_of(a)
        rec = dict(test=who, case=os.environ.get("PYTEST_CURRENT_TEST", "").split("::")[-1].split(" ")[0],
                   shape=list(a.shape[:2]), max=float(d.max()), mean=float(d.mean()),
                   frac_over_1e4=float(1.0 - small.mean()), frac_over_2e3=float(over),
                   flip_weight=flip_weight(_LAST_SCENE) if _LAST_SCENE is not None else None)
        if tb is not None:
            ex = pix - 2.0 * tb
            rec.update(pixel_mean=float(pix.mean()), needs_budget=float((pix > 5e-5).mean()),
                       budget_use=float(max(pix.mean() - 2e-5, 0.0) / tb.mean()) if tb.mean() > 0 else 0.0,
                       budget_mean=float(tb.mean()), budget_max=float(tb.max()),
                       excess_max=float(ex.max()), n_excess_over_1e5=int((ex > 1e-5).sum()),
                       n_excess_over_2e5=int((ex > 2e-5).sum()), n_excess_over_5e5=int((ex > 5e-5).sum()),
                       n_excess_over_1e4=int((ex > 1e-4).sum()),
                       n_excess_over_2e4=int((ex > 2e-4).sum()), n_pixels=int(pix.size))
        with open(_STATS_FILE, "a") as f:
            f.write(json.dumps(rec) + "\n")
    return float(d.max()), float(d.mean()), float(over)


def scene_from_datasource(drv, uri, ids, viewport, spin=(0.0, 0.0), alpha=0.05, eye=(0.0, 0.0, 1.5),
                          spr=0, data_range=(0.0, 255.0)):
    """Oracle scene for an irregular tree (uvf://): brick geometry and payloads come from the
    host library's data source (libre_amd.driver.datasource_*), everything downstream -- atlas
    emulation, texture coordinates, view data, brick order, the integrator -- is the oracle's."""
    L = lib()
    info = drv.datasource_info(uri)
    assert info["data_type"] == 1  # uint8
    s = Scene()
    vi = VolumeInfo()
    for a in range(3):
        vi.voxels[a] = info["voxels"][a]
        vi.maximumBlockSize[a] = info["max_block"][a]
        vi.overlap[a] = info["overlap"][a]
        vi.worldSize[a] = info["world_size"][a]
    vi.depth = info["depth"]
    s.vi = vi
    n = len(ids)
    mb = info["max_block"]
    ov = info["overlap"]
    slot_dim = [(m + 7) // 8 * 8 for m in mb]
    s.slot_dim = slot_dim
    slot_bytes = slot_dim[0] * slot_dim[1] * slot_dim[2]
    slots = u32x3()
    blocks = n
    while True:
        L.orc_pool_slots(u32x3(*slot_dim), slot_bytes, blocks * slot_bytes, u32x3(4096, 4096, 4096), slots)
        if slots[0] * slots[1] * slots[2] >= n:
            break
        blocks += 1
    s.slots = [slots[a] for a in range(3)]
    s.atlas_dim = [s.slots[a] * slot_dim[a] for a in range(3)]
    s.atlas = np.zeros((s.atlas_dim[2], s.atlas_dim[1], s.atlas_dim[0]), dtype=np.uint8)
    s.W, s.H = viewport
    s.mv = default_mv(spin, eye)
    s.proj = default_proj()
    s.view = ViewData()
    L.orc_make_view_data(s.mv, s.proj, (C.c_uint32 * 4)(0, 0, s.W, s.H), C.byref(vi), C.byref(s.view))
    mv = np.array(list(s.mv), dtype=np.float32).reshape(4, 4).T  # column-major -> rows
    recs = []
    bricks = {}
    for k, nid in enumerate(ids):
        node = drv.datasource_node(uri, nid)
        assert node["valid"]
        bs = node["block_size"]
        full = [bs[a] + 2 * ov[a] for a in range(3)]
        brick = drv.datasource_brick(uri, nid).reshape(full[2], full[1], full[0])
        bricks[nid] = np.ascontiguousarray(brick)
        slot = f32x3()
        L.orc_pool_kth_slot(slots, k, slot)
        origin = u32x3()
        L.orc_pool_slot_voxel_origin(slots, u32x3(*slot_dim), slot, origin)
        o = [origin[a] for a in range(3)]
        # brick + border replication into the slot padding (what the upload does)
        pad = [(0, slot_dim[2 - a] - full[2 - a]) for a in range(3)]
        s.atlas[o[2]:o[2] + slot_dim[2], o[1]:o[1] + slot_dim[1], o[0]:o[0] + slot_dim[0]] = \
            np.pad(brick, pad, mode="edge")
        wb = node["world_box"]
        centre = np.array([(wb[a] + wb[3 + a]) * 0.5 for a in range(3)] + [1.0], dtype=np.float32)
        dist = float(np.linalg.norm((mv @ centre)[:3]))
        recs.append((dist, k, nid, o, bs, wb))
    recs.sort(key=lambda r: r[0])  # CudaRaycastRenderer.cpp:160-163
    s.nodes = (NodeData * n)()
    for i, (_, k, nid, o, bs, wb) in enumerate(recs):
        nd = s.nodes[i]
        for a in range(3):
            nd.textureMin[a] = np.float32(o[a] + ov[a]) / np.float32(s.atlas_dim[a])
            nd.textureSize[a] = np.float32(bs[a]) / np.float32(s.atlas_dim[a])
            nd.aabbMin[a] = wb[a]
            nd.aabbSize[a] = np.float32(wb[3 + a]) - np.float32(wb[a])
    s.n_nodes = n
    # what tests/gpu_run.GpuScene needs to upload the same bricks into the same slots through the C ABI
    s.ids = list(ids)                      # upload order = slot order
    s.sorted_ids = [r[2] for r in recs]    # order of s.nodes
    s.bricks = bricks
    s.pool_bytes = blocks * slot_bytes
    max_level = max(unpack(nid)[0] for nid in ids)
    if spr == 0:  # CudaRaycastRenderer.cpp:113-129
        max_dim = float(max(info["voxels"]))
        spr = int(max(max_dim / float(1 << (info["depth"] - max_level - 1)), 512.0))
    s.render = RenderData(spr, 1, 32, 0, (C.c_float * 2)(*data_range))
    s.tf = linear_ramp_tf(alpha)
    s.planes = np.zeros((0, 4), dtype=np.float32)
    return s
