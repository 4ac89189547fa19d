"""Developer experiment: kernel time of ONE rank's share of a sort-first frame (N ranks emulated
on one GPU: only rank r's row bands are rendered).  Shows the per-rank latency floor."""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from libre_amd import driver, sortfirst, vrc

ap = argparse.ArgumentParser()
ap.add_argument("--world", type=int, nargs="+", default=[1, 2, 4, 8])
ap.add_argument("--bands", type=int, default=4)
ap.add_argument("--steps", type=int, default=30)
ap.add_argument("--viewport", type=int, default=1024)
ap.add_argument("--depth-split", type=int, default=0, help="VRC_OPT_DEPTH_SPLIT: two waves per tile (near / far half)")
a = ap.parse_args()
W = H = a.viewport
i = np.arange(256, dtype=np.float32) / np.float32(255.0)
tf = np.ascontiguousarray(np.stack([i, i, i, np.float32(0.05) * i], axis=1))
app = driver.App("mem://#1024,1024,1024,128", W, H, synchronous=True, min_lod=3, max_lod=3, gpu_cache_mb=3072)
app.set_colormap(tf)
app.set_option(vrc.OPT_DEPTH_SPLIT, a.depth_split)
app.render_frame(readback=False)
for world in a.world:
    lay = sortfirst.band_layout(H, world, a.bands)
    res = []
    for r in sorted(set([0, world // 2, world - 1])):
        app.set_bands(lay[r] if world > 1 else [])
        for _ in range(3):
            app.render_frame(readback=False)
        app.stats()
        for _ in range(a.steps):
            app.render_frame(readback=False)
        st = app.stats()
        res.append((r, st.kernel_ms_sum / st.kernel_launches))
    print("world %d bands/rank %d: kernel ms per rank-frame:" % (world, a.bands),
          ", ".join("rank %d: %.3f" % x for x in res), flush=True)
app.close()
