"""ctypes binding of include/livre_hip_driver.h (libre_amd/lib/libLivreHipRaycastPipeline.so):
the headless driver around the C++ plugin surface.  No algorithm lives here."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# LIVRE_HIP_HOST_LIB: another build of the same library (the sanitizer run of the host tests)
LIB_PATH = os.environ.get("LIVRE_HIP_HOST_LIB") or os.path.join(_HERE, "lib", "libLivreHipRaycastPipeline.so")


class Params(C.Structure):  # lvh_params
    _fields_ = [("device", C.c_int), ("width", C.c_uint32), ("height", C.c_uint32),
                ("tile", C.c_uint32 * 4), ("synchronous", C.c_int),
                ("samples_per_ray", C.c_uint32), ("min_lod", C.c_uint32), ("max_lod", C.c_uint32),
                ("sse", C.c_float), ("gpu_cache_mb", C.c_uint32), ("cpu_cache_mb", C.c_uint32)]


class FrameStats(C.Structure):  # lvh_frame_stats
    _fields_ = [("n_available", C.c_uint64), ("n_not_available", C.c_uint64),
                ("n_render_available", C.c_uint64), ("n_passes", C.c_uint32),
                ("kernel_ms", C.c_float), ("samples", C.c_uint64), ("samples_per_ray", C.c_uint32),
                ("kernel_ms_sum", C.c_double), ("kernel_launches", C.c_uint32), ("ray_lod", C.c_uint32)]


EXPORTS = [
    "lvh_last_error", "lvh_app_create", "lvh_app_destroy", "lvh_app_set_camera",
    "lvh_app_set_modelview", "lvh_app_set_time_step", "lvh_datasource_frame_range", "lvh_app_set_colormap", "lvh_app_set_clip_planes",
    "lvh_app_set_bands", "lvh_app_set_frames_in_flight", "lvh_app_select_slot", "lvh_app_set_option", "lvh_app_set_data_range", "lvh_app_set_ray_lod", "lvh_app_set_stream", "lvh_app_set_framebuffer", "lvh_app_render_frame",
    "lvh_app_get_stats", "lvh_app_wait_uploads", "lvh_app_synchronize", "lvh_app_volume_info",
    "lvh_comm_unique_id", "lvh_app_comm_create", "lvh_app_set_layout", "lvh_app_gather_tiles",
    "lvh_app_visible_set", "lvh_app_node_order", "lvh_app_view_matrices", "lvh_app_cache_stats", "lvh_select_visibles",
    "lvh_selftest_cache", "lvh_selftest_plugin_factory", "lvh_selftest_camera",
    "lvh_selftest_clip_planes", "lvh_selftest_renderer_parameters",
    "lvh_datasource_brick", "lvh_datasource_info", "lvh_datasource_node",
]

_lib = None


def load_library():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FileNotFoundError("%s not found: run __graft_entry__.build(). There is no CPU fallback." % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    L.lvh_last_error.restype = C.c_char_p
    L.lvh_app_create.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(Params), C.POINTER(vp)]
    L.lvh_app_destroy.argtypes = [vp]
    L.lvh_app_destroy.restype = None
    L.lvh_app_set_camera.argtypes = [vp, C.c_float * 3, C.c_float * 3, C.c_float, C.c_float]
    L.lvh_app_set_modelview.argtypes = [vp, C.c_float * 16]
    L.lvh_app_set_colormap.argtypes = [vp, vp]
    L.lvh_app_set_clip_planes.argtypes = [vp, vp, C.c_uint32]
    L.lvh_app_set_bands.argtypes = [vp, vp, vp, C.c_uint32]
    L.lvh_app_set_frames_in_flight.argtypes = [vp, C.c_uint32]
    L.lvh_app_select_slot.argtypes = [vp, C.c_uint32]
    L.lvh_app_set_option.argtypes = [vp, C.c_int, C.c_int64]
    L.lvh_app_set_data_range.argtypes = [vp, C.c_float, C.c_float]
    L.lvh_app_set_ray_lod.argtypes = [vp, C.c_int]
    L.lvh_app_set_stream.argtypes = [vp, vp]
    L.lvh_app_set_framebuffer.argtypes = [vp, vp]
    L.lvh_app_render_frame.argtypes = [vp, vp, C.POINTER(FrameStats)]
    L.lvh_app_get_stats.argtypes = [vp, C.POINTER(FrameStats)]
    L.lvh_app_wait_uploads.argtypes = [vp]
    L.lvh_app_synchronize.argtypes = [vp]
    L.lvh_app_volume_info.argtypes = [vp, C.c_uint32 * 3, C.c_uint32 * 3, C.c_uint32 * 3,
                                      C.c_float * 3, C.POINTER(C.c_uint32), C.c_uint32 * 3]
    L.lvh_app_visible_set.argtypes = [vp, C.POINTER(C.c_uint64), C.c_size_t, C.POINTER(C.c_size_t)]
    L.lvh_app_node_order.argtypes = [vp, C.POINTER(C.c_uint64), C.c_size_t, C.POINTER(C.c_size_t)]
    L.lvh_comm_unique_id.argtypes = [C.c_char_p]
    L.lvh_app_comm_create.argtypes = [vp, C.c_int, C.c_int, C.c_char_p]
    L.lvh_app_set_layout.argtypes = [vp, vp, vp, vp, C.c_uint32]
    L.lvh_app_gather_tiles.argtypes = [vp, C.c_uint32, vp, C.c_size_t, vp, C.c_size_t, C.c_int, vp]
    L.lvh_app_view_matrices.argtypes = [vp, C.c_float * 16, C.c_float * 16]
    L.lvh_app_cache_stats.argtypes = [vp, C.c_uint64 * 4, C.c_uint64 * 4]
    L.lvh_select_visibles.argtypes = [C.c_char_p, C.c_float * 16, C.c_float * 16, C.c_uint32,
                                      C.c_float, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64),
                                      C.c_size_t, C.POINTER(C.c_size_t)]
    L.lvh_selftest_camera.argtypes = [(C.c_float * 16) * 4]
    L.lvh_datasource_brick.argtypes = [C.c_char_p, C.c_uint64, vp, C.c_size_t, C.POINTER(C.c_size_t)]
    u3, f3 = C.POINTER(C.c_uint32), C.POINTER(C.c_float)
    L.lvh_datasource_info.argtypes = [C.c_char_p, u3, u3, u3, f3, u3, u3, u3, u3]
    L.lvh_datasource_node.argtypes = [C.c_char_p, C.c_uint64, C.POINTER(C.c_int), u3, u3, f3]
    _lib = L
    return L


class DriverError(RuntimeError):
    pass


def check(L, rc):
    if rc != 0:
        raise DriverError((L.lvh_last_error() or b"").decode("utf-8", "replace"))


class App:
    """One rendering process: data source + RenderPipeline("hip") + per-frame RenderInputs."""

    def __init__(self, volume_uri, width, height, device=0, renderer="hip", tile=None,
                 synchronous=True, samples_per_ray=0, min_lod=0, max_lod=0, sse=0.0,
                 gpu_cache_mb=0, cpu_cache_mb=0):
        self.L = L = load_library()
        p = Params()
        p.device, p.width, p.height = device, width, height
        if tile is not None:
            for i in range(4):
                p.tile[i] = tile[i]
        p.synchronous = 1 if synchronous else 0
        p.samples_per_ray, p.min_lod, p.max_lod, p.sse = samples_per_ray, min_lod, max_lod, sse
        p.gpu_cache_mb, p.cpu_cache_mb = gpu_cache_mb, cpu_cache_mb
        self.width, self.height = (tile[2], tile[3]) if tile is not None else (width, height)
        self.h = C.c_void_p()
        check(L, L.lvh_app_create(volume_uri.encode(), renderer.encode(), C.byref(p), C.byref(self.h)))

    def set_camera(self, position=(0.0, 0.0, 1.5), lookat=(0.0, 0.0, 0.0), spin=(0.0, 0.0)):
        check(self.L, self.L.lvh_app_set_camera(self.h, (C.c_float * 3)(*position),
                                                (C.c_float * 3)(*lookat), spin[0], spin[1]))

    def set_time_step(self, t):
        check(self.L, self.L.lvh_app_set_time_step(self.h, C.c_uint32(t)))

    def set_colormap(self, rgba256):
        a = np.ascontiguousarray(rgba256, dtype=np.float32).reshape(1024)
        check(self.L, self.L.lvh_app_set_colormap(self.h, a.ctypes.data))

    def set_clip_planes(self, planes):
        a = np.ascontiguousarray(planes, dtype=np.float32).reshape(-1, 4)
        check(self.L, self.L.lvh_app_set_clip_planes(self.h, a.ctypes.data if len(a) else None, len(a)))

    def set_bands(self, bands):
        """bands: list of (y0, h) rows of the full frame rendered by this process (one launch)."""
        n = len(bands)
        y0 = (C.c_uint32 * max(1, n))(*[b[0] for b in bands])
        h = (C.c_uint32 * max(1, n))(*[b[1] for b in bands])
        check(self.L, self.L.lvh_app_set_bands(self.h, y0, h, n))
        self.height = sum(b[1] for b in bands) if n else self.height

    def set_frames_in_flight(self, n):
        check(self.L, self.L.lvh_app_set_frames_in_flight(self.h, n))

    def select_slot(self, slot):
        check(self.L, self.L.lvh_app_select_slot(self.h, slot))

    def set_option(self, option, value):
        check(self.L, self.L.lvh_app_set_option(self.h, option, value))

    def set_data_range(self, lo, hi):
        """dataSourceRange of a volume that is not uint8 (extension)."""
        check(self.L, self.L.lvh_app_set_data_range(self.h, lo, hi))

    def set_ray_lod(self, enable=True):
        """Per-ray adaptive LOD (extension): ancestors of the visible set resident, LOD chosen along the ray."""
        check(self.L, self.L.lvh_app_set_ray_lod(self.h, 1 if enable else 0))

    def set_stream(self, stream_handle):
        check(self.L, self.L.lvh_app_set_stream(self.h, stream_handle))

    def set_framebuffer(self, device_ptr):
        check(self.L, self.L.lvh_app_set_framebuffer(self.h, device_ptr))

    def render_frame(self, readback=True):
        st = FrameStats()
        fb = np.zeros((self.height, self.width, 4), dtype=np.float32) if readback else None
        check(self.L, self.L.lvh_app_render_frame(self.h, fb.ctypes.data if readback else None, C.byref(st)))
        return fb, st

    def stats(self):
        st = FrameStats()
        check(self.L, self.L.lvh_app_get_stats(self.h, C.byref(st)))
        return st

    def wait_uploads(self):
        check(self.L, self.L.lvh_app_wait_uploads(self.h))

    def synchronize(self):
        check(self.L, self.L.lvh_app_synchronize(self.h))

    def volume_info(self):
        v, mb, ov, rb = (C.c_uint32 * 3)(), (C.c_uint32 * 3)(), (C.c_uint32 * 3)(), (C.c_uint32 * 3)()
        ws, depth = (C.c_float * 3)(), C.c_uint32()
        check(self.L, self.L.lvh_app_volume_info(self.h, v, mb, ov, ws, C.byref(depth), rb))
        return dict(voxels=list(v), max_block=list(mb), overlap=list(ov), world_size=list(ws),
                    depth=depth.value, root_blocks=list(rb))

    def visible_set(self):
        n = C.c_size_t()
        check(self.L, self.L.lvh_app_visible_set(self.h, None, 0, C.byref(n)))
        ids = (C.c_uint64 * max(1, n.value))()
        check(self.L, self.L.lvh_app_visible_set(self.h, ids, n.value, C.byref(n)))
        return list(ids)[:n.value]

    def comm_create(self, rank, world, unique_id):
        """Collective: the RCCL communicator of the sort-first tile exchange (unique_id: 128 bytes from
        comm_unique_id() of one rank; None for a world of one)."""
        check(self.L, self.L.lvh_app_comm_create(self.h, rank, world, unique_id))

    def set_layout(self, layout):
        """layout: per rank a list of (y0, h) bands (sortfirst.band_layout); bands in frame order."""
        flat = sorted((y0, h, r) for r, bands in enumerate(layout) for (y0, h) in bands)
        n = len(flat)
        rk = (C.c_uint32 * max(1, n))(*[b[2] for b in flat])
        y0 = (C.c_uint32 * max(1, n))(*[b[0] for b in flat])
        hh = (C.c_uint32 * max(1, n))(*[b[1] for b in flat])
        check(self.L, self.L.lvh_app_set_layout(self.h, rk, y0, hh, n))

    def gather_tiles(self, n_frames, local_ptr, local_stride, frame_ptr, frame_stride, root=0, stream=None):
        check(self.L, self.L.lvh_app_gather_tiles(self.h, n_frames, local_ptr, local_stride, frame_ptr,
                                                   frame_stride, root, stream))

    def node_order(self):
        """ids of the bricks of the last frame in the renderer's front-to-back order."""
        n = C.c_size_t()
        check(self.L, self.L.lvh_app_node_order(self.h, None, 0, C.byref(n)))
        ids = (C.c_uint64 * max(1, n.value))()
        check(self.L, self.L.lvh_app_node_order(self.h, ids, n.value, C.byref(n)))
        return list(ids)[:n.value]

    def view_matrices(self):
        mv, proj = (C.c_float * 16)(), (C.c_float * 16)()
        check(self.L, self.L.lvh_app_view_matrices(self.h, mv, proj))
        return list(mv), list(proj)

    def cache_stats(self):
        t, d = (C.c_uint64 * 4)(), (C.c_uint64 * 4)()
        check(self.L, self.L.lvh_app_cache_stats(self.h, t, d))
        keys = ("used", "max", "count", "misses")
        return dict(zip(keys, t)), dict(zip(keys, d))

    def close(self):
        if self.h:
            self.L.lvh_app_destroy(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def comm_unique_id():
    """128 bytes that identify a new RCCL communicator (one rank calls this, all ranks get the bytes)."""
    L = load_library()
    buf = C.create_string_buffer(128)
    check(L, L.lvh_comm_unique_id(buf))
    return buf.raw


def select_visibles(volume_uri, mv, proj, window_height, sse, min_lod, max_lod):
    L = load_library()
    n = C.c_size_t()
    m, p = (C.c_float * 16)(*mv), (C.c_float * 16)(*proj)
    check(L, L.lvh_select_visibles(volume_uri.encode(), m, p, window_height, sse, min_lod, max_lod,
                                   None, 0, C.byref(n)))
    ids = (C.c_uint64 * max(1, n.value))()
    check(L, L.lvh_select_visibles(volume_uri.encode(), m, p, window_height, sse, min_lod, max_lod,
                                   ids, n.value, C.byref(n)))
    return list(ids)[:n.value]


def datasource_info(volume_uri):
    L = load_library()
    v, mb, ov, rb = (C.c_uint32 * 3)(), (C.c_uint32 * 3)(), (C.c_uint32 * 3)(), (C.c_uint32 * 3)()
    ws, depth, dt, cc = (C.c_float * 3)(), C.c_uint32(), C.c_uint32(), C.c_uint32()
    check(L, L.lvh_datasource_info(volume_uri.encode(), v, mb, ov, ws, C.byref(depth), rb, C.byref(dt),
                                   C.byref(cc)))
    return dict(voxels=list(v), max_block=list(mb), overlap=list(ov), world_size=list(ws), depth=depth.value,
                root_blocks=list(rb), data_type=dt.value, comp_count=cc.value)


def datasource_frame_range(volume_uri):
    L = load_library()
    r = (C.c_uint32 * 2)()
    check(L, L.lvh_datasource_frame_range(volume_uri.encode(), r))
    return (int(r[0]), int(r[1]))


def datasource_node(volume_uri, node_id):
    L = load_library()
    valid, bs, vb, wb = C.c_int(), (C.c_uint32 * 3)(), (C.c_uint32 * 6)(), (C.c_float * 6)()
    check(L, L.lvh_datasource_node(volume_uri.encode(), C.c_uint64(node_id), C.byref(valid), bs, vb, wb))
    return dict(valid=bool(valid.value), block_size=list(bs), voxel_box=list(vb), world_box=list(wb))


def datasource_brick(volume_uri, node_id):
    L = load_library()
    n = C.c_size_t()
    check(L, L.lvh_datasource_brick(volume_uri.encode(), node_id, None, 0, C.byref(n)))
    out = np.zeros(n.value, dtype=np.uint8)
    check(L, L.lvh_datasource_brick(volume_uri.encode(), node_id, out.ctypes.data, n.value, C.byref(n)))
    return out
