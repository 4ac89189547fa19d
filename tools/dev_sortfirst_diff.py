"""Developer probe: how do the frames of a random sort-first layout differ from the full frame (LDS trilinear kernel)?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import orc
from libre_amd import driver as drv, sortfirst, vrc

for seed in [int(x) for x in sys.argv[1:]] or [8, 11, 21]:
    rng = np.random.default_rng(9000 + seed)
    W, H = int(rng.integers(16, 56)), int(rng.integers(24, 72))
    world = int(rng.choice([2, 3, 4, 8]))
    bpr = int(rng.choice([1, 2, 4]))
    spin = (float(rng.uniform(-3.1, 3.1)), float(rng.uniform(-1.2, 1.2)))
    mode = int(rng.integers(0, 3))
    kw = dict(synchronous=True, min_lod=2, max_lod=2, gpu_cache_mb=8)
    frames = {}
    for kernel in (vrc.KERNEL_AUTO, vrc.KERNEL_GRID_DDA):
        with drv.App("hash://#64,64,64,16", W, H, **kw) as app:
            app.set_camera(spin=spin)
            app.set_colormap(orc.linear_ramp_tf(0.3))
            app.set_option(vrc.OPT_FILTER, vrc.FILTER_TRILINEAR)
            app.set_option(vrc.OPT_KERNEL, kernel)
            full, _ = app.render_frame()
            out = np.full_like(full, -1.0)
            for bands in sortfirst.band_layout(H, world, bpr):
                if not bands:
                    continue
                app.set_bands(bands)
                fb, _ = app.render_frame()
                off = 0
                for (y0, h) in bands:
                    out[y0:y0 + h] = fb[off:off + h]
                    off += h
        d = np.abs(out - full)
        frames[kernel] = full
        print("seed %d %dx%d world %d bpr %d mode %d kernel %d: %d pixels differ, max %.3g" %
              (seed, W, H, world, bpr, mode, kernel, int((d.max(-1) > 0).sum()), d.max()))
    print("   LDS vs gather full frames: max %.3g" % np.abs(frames[vrc.KERNEL_AUTO] - frames[vrc.KERNEL_GRID_DDA]).max())
