"""Drive the HIP path through the C ABI (ctypes) with a scene built by tests/orc.py."""
import ctypes as C

import numpy as np

from libre_amd import vrc


class GpuScene:
    """Uploads a Scene's bricks through vrc_pool_copy_to_slot and renders it."""

    def __init__(self, s, device=0, lib=None):
        self.L = L = lib if lib is not None else vrc.load_library()  # lib: a developer A/B build
        self.s = s
        self.ctx = C.c_void_p()
        vrc.check(L, L.vrc_ctx_create(device, C.byref(self.ctx)))
        self.pool = C.c_void_p()
        mb = vrc.u32x3(*[s.vi.maximumBlockSize[a] for a in range(3)])
        self.voxel_bytes = s.atlas.dtype.itemsize  # 2: uint16 volume (extension)
        vrc.check(L, L.vrc_pool_create(self.ctx, self.voxel_bytes, 0, 0, 1, mb,
                                       s.pool_bytes * self.voxel_bytes, C.byref(self.pool)))
        self.slots = {}
        for nid in s.ids:  # same order as the oracle's k-th slot
            brick = s.bricks[nid]
            slot = vrc.f32x3()
            size = vrc.u32x3(brick.shape[2], brick.shape[1], brick.shape[0])
            vrc.check(L, L.vrc_pool_copy_to_slot(self.pool, brick.ctypes.data, size, slot))
            self.slots[nid] = (slot[0], slot[1], slot[2])

    def info(self):
        L = self.L
        sb, ab, fs = C.c_size_t(), C.c_size_t(), C.c_uint32()
        ad, sl = vrc.u32x3(), vrc.u32x3()
        vrc.check(L, L.vrc_pool_info(self.pool, C.byref(sb), ad, C.byref(ab), sl, C.byref(fs)))
        return dict(slot_bytes=sb.value, atlas_dim=list(ad), atlas_bytes=ab.value,
                    slots=list(sl), free=fs.value)

    def render(self, kernel=vrc.KERNEL_AUTO, frac_bits=8, count=True, passes=None, stepping=1,
               filter_mode=0, variant=0, ray_lod=None, slabs=None):
        L, s = self.L, self.s
        # ray_lod = (screenSpaceError, worldSpacePerPixel): per-ray adaptive LOD over the hierarchy s.nodes
        vrc.check(L, L.vrc_set_ray_lod(self.ctx, 1 if ray_lod else 0, ray_lod[0] if ray_lod else 0.0,
                                       ray_lod[1] if ray_lod else 0.0))
        vrc.check(L, L.vrc_set_option(self.ctx, vrc.OPT_KERNEL, kernel))
        vrc.check(L, L.vrc_set_option(self.ctx, vrc.OPT_TF_FRAC_BITS, frac_bits))
        vrc.check(L, L.vrc_set_option(self.ctx, vrc.OPT_COUNT_SAMPLES, 1 if count else 0))
        vrc.check(L, L.vrc_set_option(self.ctx, vrc.OPT_STEPPING, stepping))
        vrc.check(L, L.vrc_set_option(self.ctx, vrc.OPT_FILTER, filter_mode))
        vrc.check(L, L.vrc_set_option(self.ctx, vrc.OPT_VARIANT, variant))
        planes = s.planes
        vrc.check(L, L.vrc_update(self.ctx, s.tf.ctypes.data,
                                  planes.ctypes.data if len(planes) else None, len(planes)))
        view = C.cast(C.byref(s.view), C.POINTER(vrc.ViewData))
        render = C.cast(C.byref(s.render), C.POINTER(vrc.RenderData))
        nodes = C.cast(s.nodes, C.POINTER(vrc.NodeData))
        vrc.check(L, L.vrc_pre_render(self.ctx, view))
        samples = 0
        stats = vrc.Stats()
        for planes_i, idx in (slabs or []):
            # one pass per slab of space: its own clip planes and the bricks that reach into it (what
            # HipRaycastPipeline does for a per-ray LOD hierarchy larger than the atlas), accumulating
            pl = np.ascontiguousarray(planes_i, dtype=np.float32).reshape(-1, 4)
            vrc.check(L, L.vrc_update(self.ctx, s.tf.ctypes.data, pl.ctypes.data if len(pl) else None, len(pl)))
            sub_nodes = (vrc.NodeData * max(1, len(idx)))()
            for k, i in enumerate(idx):
                C.memmove(C.byref(sub_nodes, k * C.sizeof(vrc.NodeData)), C.byref(s.nodes, i * C.sizeof(vrc.NodeData)),
                          C.sizeof(vrc.NodeData))
            vrc.check(L, L.vrc_render(self.ctx, view, C.cast(sub_nodes, C.POINTER(vrc.NodeData)), len(idx), render, self.pool))
            vrc.check(L, L.vrc_get_stats(self.ctx, C.byref(stats)))
            samples += stats.samples
        if passes is None:
            passes = [(0, s.n_nodes)] if not slabs else []
        for (a, b) in passes:  # multipass: CudaRaycastPipeline.cpp:149-185
            sub = C.cast(C.byref(s.nodes, a * C.sizeof(vrc.NodeData)), C.POINTER(vrc.NodeData)) if a else nodes
            vrc.check(L, L.vrc_render(self.ctx, view, sub, b - a, render, self.pool))
            vrc.check(L, L.vrc_get_stats(self.ctx, C.byref(stats)))
            samples += stats.samples
        fb = np.zeros((s.H, s.W, 4), dtype=np.float32)
        vrc.check(L, L.vrc_post_render(self.ctx, fb.ctypes.data))
        return fb, samples, stats

    def close(self):
        if self.pool:
            self.L.vrc_pool_destroy(self.pool)
            self.pool = None
        if self.ctx:
            self.L.vrc_ctx_destroy(self.ctx)
            self.ctx = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()
