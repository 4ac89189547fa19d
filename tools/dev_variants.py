"""Developer A/B of libvrc_hip.so builds (tools/dev_layouts.sh) on one scene: every build renders the
same frame (checked against the first build's frame), then the gather kernel is timed.
usage: python tools/dev_variants.py [--volume mem|hash] [--steps 20] variants/a.so variants/b.so ..."""
import argparse
import ctypes as C
import hashlib
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
    ap.add_argument("libs", nargs="+")
    ap.add_argument("--voxels", type=int, default=1024)
    ap.add_argument("--block", type=int, default=128)
    ap.add_argument("--viewport", type=int, default=1024)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--volume", default="mem")
    ap.add_argument("--spin", type=float, nargs=2, default=(0.0, 0.0))
    ap.add_argument("--alpha", type=float, default=0.05)
    ap.add_argument("--rounds", type=int, default=2, help="interleaved timing rounds over all builds")
    ap.add_argument("--no-tile-order", action="store_true", help="row-major tile order instead of heaviest-first")
    ap.add_argument("--kernel", type=int, default=vrc.KERNEL_GRID_DDA, help="VRC_KERNEL_* code")
    ap.add_argument("--filter", type=int, default=0, help="0 nearest, 1 trilinear")
    ap.add_argument("--dtype", default="u8", help="u8 | u16")
    a = ap.parse_args()
    t0 = time.time()
    s = orc.build_scene(voxels=(a.voxels,) * 3, block=a.block, viewport=(a.viewport,) * 2,
                        volume=a.volume, spin=tuple(a.spin), alpha=a.alpha, dtype=a.dtype)
    print("scene %s built in %.1fs: %d nodes spr %d" % (a.volume, time.time() - t0, s.n_nodes,
          s.render.samplesPerRay), flush=True)
    view = C.cast(C.byref(s.view), C.POINTER(vrc.ViewData))
    render = C.cast(C.byref(s.render), C.POINTER(vrc.RenderData))
    nodes = C.cast(s.nodes, C.POINTER(vrc.NodeData))
    first = None
    scenes = []
    for path in a.libs:
        L = vrc.load_library(path)
        g = GpuScene(s, lib=L)
        fb, n, st = g.render(kernel=a.kernel, filter_mode=a.filter, count=True)
        if first is None:
            first = fb
        d = np.abs(fb - first)
        print("%-28s samples %d sha1 %s max|d| vs first %.3g mean %.3g atlas %.2f GB" % (
            os.path.basename(path), n, hashlib.sha1(fb.tobytes()).hexdigest()[:12], d.max(), d.mean(),
            g.info()["atlas_bytes"] / 1e9), flush=True)
        vrc.check(L, L.vrc_set_option(g.ctx, vrc.OPT_COUNT_SAMPLES, 0))
        vrc.check(L, L.vrc_set_option(g.ctx, vrc.OPT_TILE_ORDER, 0 if a.no_tile_order else 1))
        scenes.append((path, L, g, n))
    res = {p: [] for p in a.libs}
    for r in range(a.rounds):
        for path, L, g, n in scenes:
            stt = vrc.Stats()
            for i in range(a.steps + 3):
                vrc.check(L, L.vrc_pre_render(g.ctx, view))
                vrc.check(L, L.vrc_render(g.ctx, view, nodes, s.n_nodes, render, g.pool))
                vrc.check(L, L.vrc_get_stats(g.ctx, C.byref(stt)))
                if i >= 3:
                    res[path].append(stt.kernel_ms)
    for path, L, g, n in scenes:
        ms = np.array(res[path])
        print("%-28s median %.4f ms  min %.4f ms  -> %.0f Gsamples/s" % (
            os.path.basename(path), np.median(ms), ms.min(), n / np.median(ms) / 1e6), flush=True)
        g.close()


if __name__ == "__main__":
    main()
