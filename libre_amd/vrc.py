"""ctypes binding of the C ABI in include/vrc_hip.h (libre_amd/lib/libvrc_hip.so).

This is the reference-side binding stub an integrator would add; it holds
no algorithm.  The library must exist: there is no CPU fallback and no silent degradation.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libvrc_hip.so")

VRC_OK, VRC_EINVAL, VRC_EHIP, VRC_EFULL, VRC_ENOMEM, VRC_EUNSUPPORTED, VRC_EHIERARCHY, VRC_ECOMM = range(8)
OPT_KERNEL, OPT_FILTER, OPT_TF_FRAC_BITS, OPT_COUNT_SAMPLES, OPT_TILE_ORDER, OPT_STEPPING, OPT_VARIANT, OPT_KERNEL_USED, OPT_KERNEL_TIMING, OPT_DEPTH_SPLIT, OPT_ERT_COMPACTION, OPT_GREY_TABLE, OPT_PACKED_ATLAS = range(1, 14)
VARIANT_CUDARAYCASTER, VARIANT_GLRAYCASTER = 0, 1
FILTER_NEAREST, FILTER_TRILINEAR = 0, 1
KERNEL_AUTO, KERNEL_REFERENCE_ORDER, KERNEL_GRID_DDA, KERNEL_LDS, KERNEL_RAY_LOD, KERNEL_PACKED = 0, 1, 2, 3, 4, 5

ABI_VERSION = 4  # VRC_ABI_VERSION of include/vrc_hip.h this binding was written against

f32x3 = C.c_float * 3
u32x3 = C.c_uint32 * 3


class NodeData(C.Structure):  # vrc_node_data
    _fields_ = [("textureMin", f32x3), ("textureSize", f32x3),
                ("aabbMin", f32x3), ("aabbSize", f32x3)]


class ViewData(C.Structure):  # vrc_view_data
    _fields_ = [("eyePosition", f32x3), ("glViewport", C.c_uint32 * 4),
                ("invProjMatrix", C.c_float * 16), ("modelViewMatrix", C.c_float * 16),
                ("invViewMatrix", C.c_float * 16), ("aabbMin", f32x3), ("aabbMax", f32x3),
                ("nearPlane", C.c_float)]


class RenderData(C.Structure):  # vrc_render_data
    _fields_ = [("samplesPerRay", C.c_uint32), ("samplesPerPixel", C.c_uint32),
                ("maxSamplesPerRay", C.c_uint32), ("datatype", C.c_uint32),
                ("dataSourceRange", C.c_float * 2)]


class Stats(C.Structure):  # vrc_stats
    _fields_ = [("kernel_ms", C.c_float), ("samples", C.c_uint64),
                ("kernel_variant", C.c_uint32), ("grid_dims", C.c_uint32 * 3),
                ("kernel_ms_sum", C.c_double), ("kernel_launches", C.c_uint32)]


class VrcError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("vrc error %d: %s" % (code, msg))
        self.code = code


#: every symbol include/vrc_hip.h declares; tests check the library exports all of them
EXPORTS = [
    "vrc_ctx_create", "vrc_ctx_destroy", "vrc_ctx_set_stream", "vrc_set_option", "vrc_get_option", "vrc_set_ray_lod",
    "vrc_pool_create", "vrc_pool_destroy", "vrc_pool_copy_to_slot", "vrc_pool_copy_to_slot_device",
    "vrc_pool_release_slot", "vrc_pool_info", "vrc_pool_synchronize", "vrc_pool_read_region",
    "vrc_pool_histogram",
    "vrc_update", "vrc_pre_render", "vrc_set_row_map", "vrc_set_framebuffer", "vrc_get_framebuffer", "vrc_render",
    "vrc_post_render", "vrc_synchronize", "vrc_get_stats", "vrc_get_ray_counts", "vrc_last_error", "vrc_last_kernel", "vrc_last_kernel_occupancy", "vrc_abi_version", "vrc_is_dev_build",
    "vrc_comm_unique_id", "vrc_comm_create", "vrc_comm_destroy", "vrc_comm_info", "vrc_gather_tiles",
]
COMM_ID_BYTES = 128


class Band(C.Structure):  # vrc_band
    _fields_ = [("rank", C.c_uint32), ("frame_row", C.c_uint32), ("rows", C.c_uint32)]


_lib = None


def load_library(path=None):
    """Load libvrc_hip.so and declare its prototypes.  Raises if the library is missing."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("VRC_HIP_LIB") or LIB_PATH  # VRC_HIP_LIB: developer A/B builds
    if not os.path.exists(p):
        raise FileNotFoundError(
            "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback." % p)
    L = C.CDLL(p)
    vp = C.c_void_p
    L.vrc_last_error.restype = C.c_char_p
    L.vrc_abi_version.restype = C.c_int
    L.vrc_is_dev_build.restype = C.c_int
    L.vrc_last_kernel.restype = C.c_char_p
    L.vrc_last_kernel_occupancy.argtypes = [C.POINTER(C.c_int), C.POINTER(C.c_int)]
    # a developer build (-DVRC_DEV_BUILD) reports the negated version
    if abs(L.vrc_abi_version()) != ABI_VERSION:
        raise RuntimeError("%s has ABI version %d, this binding needs %d: rebuild it (__graft_entry__.build())"
                           % (p, L.vrc_abi_version(), ABI_VERSION))
    L.vrc_ctx_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.vrc_ctx_destroy.argtypes = [vp]
    L.vrc_ctx_destroy.restype = None
    L.vrc_ctx_set_stream.argtypes = [vp, vp]
    L.vrc_set_option.argtypes = [vp, C.c_int, C.c_int64]
    L.vrc_set_ray_lod.argtypes = [vp, C.c_int, C.c_float, C.c_float]
    L.vrc_get_option.argtypes = [vp, C.c_int, C.POINTER(C.c_int64)]
    L.vrc_pool_create.argtypes = [vp, C.c_size_t, C.c_int, C.c_int, C.c_size_t, u32x3, C.c_size_t,
                                  C.POINTER(vp)]
    L.vrc_pool_destroy.argtypes = [vp]
    L.vrc_pool_destroy.restype = None
    L.vrc_pool_copy_to_slot.argtypes = [vp, vp, u32x3, f32x3]
    L.vrc_pool_copy_to_slot_device.argtypes = [vp, vp, u32x3, f32x3]
    L.vrc_pool_release_slot.argtypes = [vp, f32x3]
    L.vrc_pool_info.argtypes = [vp, C.POINTER(C.c_size_t), u32x3, C.POINTER(C.c_size_t), u32x3,
                                C.POINTER(C.c_uint32)]
    L.vrc_pool_synchronize.argtypes = [vp]
    L.vrc_pool_read_region.argtypes = [vp, u32x3, u32x3, vp]
    L.vrc_pool_histogram.argtypes = [vp, f32x3, u32x3, u32x3, C.c_uint32, C.c_uint64, vp]
    L.vrc_update.argtypes = [vp, vp, vp, C.c_uint32]
    L.vrc_pre_render.argtypes = [vp, C.POINTER(ViewData)]
    L.vrc_set_row_map.argtypes = [vp, vp, C.c_uint32]
    L.vrc_set_framebuffer.argtypes = [vp, vp, C.c_uint32, C.c_uint32]
    L.vrc_get_framebuffer.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.vrc_render.argtypes = [vp, C.POINTER(ViewData), C.POINTER(NodeData), C.c_uint32,
                             C.POINTER(RenderData), vp]
    L.vrc_post_render.argtypes = [vp, vp]
    L.vrc_synchronize.argtypes = [vp]
    L.vrc_get_stats.argtypes = [vp, C.POINTER(Stats)]
    L.vrc_get_ray_counts.argtypes = [vp, C.POINTER(C.c_uint32 * 8), C.POINTER(C.c_int)]
    L.vrc_comm_unique_id.argtypes = [C.c_char_p]
    L.vrc_comm_create.argtypes = [vp, C.c_int, C.c_int, C.c_char_p, C.POINTER(vp)]
    L.vrc_comm_destroy.argtypes = [vp]
    L.vrc_comm_destroy.restype = None
    L.vrc_comm_info.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.vrc_gather_tiles.argtypes = [vp, vp, C.POINTER(Band), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, vp, C.c_size_t,
                                   vp, C.c_size_t, C.c_int, vp]
    if path is None:
        _lib = L
    return L


def check(L, rc):
    if rc != VRC_OK:
        raise VrcError(rc, (L.vrc_last_error() or b"").decode("utf-8", "replace"))
