"""Developer micro-bench of the raycast kernel on oracle-derived inputs (NOT bench.py: the
judged benchmark drives the C++ host; this exists to iterate on the kernel)."""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import orc  # noqa: E402
from gpu_run import GpuScene  # noqa: E402
from libre_amd import vrc  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--voxels", type=int, default=1024)
    ap.add_argument("--block", type=int, default=128)
    ap.add_argument("--viewport", type=int, default=1024)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--volume", default="mem")
    ap.add_argument("--spin", type=float, nargs=2, default=(0.0, 0.0))
    ap.add_argument("--alpha", type=float, default=0.05)
    ap.add_argument("--ref-order", action="store_true")
    ap.add_argument("--no-tile-order", action="store_true")
    ap.add_argument("--stepping", type=int, default=1)
    ap.add_argument("--kernels", type=int, nargs="*", default=None,
                    help="VRC_KERNEL_* codes to time (default: 2 = grid DDA gather kernel)")
    ap.add_argument("--filters", type=int, nargs="*", default=[0], help="0 nearest, 1 trilinear")
    ap.add_argument("--dtype", default="u8", help="u8 | u16")
    ap.add_argument("--tile-offset", type=int, default=-1,
                    help="render the sub-frame that starts at this pixel offset (x and y) and is 8 pixels smaller: "
                         "moves the 8x8 pixel tiles against the 8^3 micro-blocks of the atlas")
    ap.add_argument("--ray-lod", type=float, default=0.0,
                    help="screen-space error: time the per-ray LOD kernel on the same node list (add --levels)")
    ap.add_argument("--levels", type=int, nargs="*", default=None,
                    help="tree levels in the node list (default: the leaves; e.g. 0 1 2 3 = whole pyramid)")
    ap.add_argument("--spr", type=int, default=0, help="samples per ray (0 = the automatic value)")
    ap.add_argument("--ert-parts", type=int, nargs="*", default=[0],
                    help="VRC_OPT_ERT_COMPACTION values to time (0 = one launch)")
    a = ap.parse_args()
    t0 = time.time()
    ids = None
    if a.levels is not None:
        ids = orc.all_level_ids(orc.mem_volume_info(a.voxels, a.voxels, a.voxels, a.block), a.levels)
    s = orc.build_scene(voxels=(a.voxels,) * 3, block=a.block, viewport=(a.viewport,) * 2,
                        volume=a.volume, spin=tuple(a.spin), alpha=a.alpha, ids=ids, dtype=a.dtype, spr=a.spr,
                        tile=None if a.tile_offset < 0 else (a.tile_offset, a.tile_offset, a.viewport - 8,
                                                             a.viewport - 8, a.viewport, a.viewport))
    print("scene built in %.1fs: %d nodes spr %d atlas %s" % (time.time() - t0, s.n_nodes,
          s.render.samplesPerRay, s.atlas_dim), flush=True)
    with GpuScene(s) as g:
        print("uploaded in %.1fs" % (time.time() - t0), g.info(), flush=True)
        lod = (a.ray_lod, orc.world_space_per_pixel(s)) if a.ray_lod > 0 else None
        fb, n, st = g.render(count=True, ray_lod=lod)
        print("samples/frame %d, counted-kernel %.3f ms, variant %d grid %s alpha max %.3f" %
              (n, st.kernel_ms, st.kernel_variant, list(st.grid_dims), fb[..., 3].max()), flush=True)
        L = g.L
        view = C.cast(C.byref(s.view), C.POINTER(vrc.ViewData))
        render = C.cast(C.byref(s.render), C.POINTER(vrc.RenderData))
        nodes = C.cast(s.nodes, C.POINTER(vrc.NodeData))
        vrc.check(L, L.vrc_set_option(g.ctx, vrc.OPT_COUNT_SAMPLES, 0))
        vrc.check(L, L.vrc_set_option(g.ctx, vrc.OPT_TILE_ORDER, 0 if a.no_tile_order else 1))
        vrc.check(L, L.vrc_set_option(g.ctx, vrc.OPT_STEPPING, a.stepping))
        kernels = [vrc.KERNEL_GRID_DDA] + ([vrc.KERNEL_REFERENCE_ORDER] if a.ref_order else [])
        if a.kernels:
            kernels = a.kernels
        if lod and not a.kernels:
            kernels = [vrc.KERNEL_AUTO]  # (under per-ray LOD: 0 = staged where it applies, 2 = gathers)
        for k, flt, parts in [(k, flt, parts) for flt in a.filters for k in kernels for parts in a.ert_parts]:
            vrc.check(L, L.vrc_set_option(g.ctx, vrc.OPT_KERNEL, k))
            vrc.check(L, L.vrc_set_option(g.ctx, vrc.OPT_FILTER, flt))
            vrc.check(L, L.vrc_set_option(g.ctx, vrc.OPT_ERT_COMPACTION, parts))
            ms = []
            stt = vrc.Stats()
            for i in range(a.steps + 2):
                vrc.check(L, L.vrc_pre_render(g.ctx, view))
                vrc.check(L, L.vrc_render(g.ctx, view, nodes, s.n_nodes, render, g.pool))
                vrc.check(L, L.vrc_get_stats(g.ctx, C.byref(stt)))
                if i >= 2:
                    ms.append(stt.kernel_ms)
            ms = np.array(ms)
            if parts:
                counts = (C.c_uint32 * 8)()
                used = C.c_int()
                vrc.check(L, L.vrc_get_ray_counts(g.ctx, C.byref(counts), C.byref(used)))
                print("ray compaction in %d launches: rays alive after each %s of %d pixels"
                      % (used.value, list(counts)[:max(used.value - 1, 0)], a.viewport ** 2), flush=True)
            A = a.voxels ** 3 + a.viewport ** 2 * 16 + s.n_nodes * 48 + 4096
            print("kernel %d filter %d: median %.3f ms min %.3f ms -> %.1f Msamples/s, %.1f fps, algorithmic %.1f GB/s"
                  % (k, flt, np.median(ms), ms.min(), n / np.median(ms) / 1e3, 1e3 / np.median(ms),
                     A / np.median(ms) / 1e6), flush=True)


if __name__ == "__main__":
    main()
