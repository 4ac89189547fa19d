"""Developer probe: one seed of tests/test_gpu_host.py::test_random_lod_cuts_through_the_plugin_match_the_oracle,
rendered through the plugin and through the C ABI with the oracle's node list; prints what differs where."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import orc  # noqa: E402
from gpu_run import GpuScene  # noqa: E402
from libre_amd import driver as drv, vrc  # noqa: E402

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 6
rng = np.random.default_rng(7000 + seed)
vox = int(rng.choice([64, 128]))
block = 16
volume = str(rng.choice(["mem", "hash"]))
W, H = int(rng.integers(24, 64)), int(rng.integers(24, 64))
eye = (float(rng.uniform(-0.4, 0.4)), float(rng.uniform(-0.4, 0.4)), float(rng.uniform(0.3, 1.8)))
spin = (float(rng.uniform(-3.1, 3.1)), float(rng.uniform(-1.2, 1.2)))
sse = float(rng.choice([0.5, 1.0, 2.0, 4.0]))
uri = "%s://#%d,%d,%d,%d" % (volume, vox, vox, vox, block)
with drv.App(uri, W, H, synchronous=True, sse=sse, gpu_cache_mb=32) as app:
    app.set_camera(position=eye, spin=spin)
    app.set_colormap(orc.linear_ramp_tf(0.3))
    app.set_option(vrc.OPT_COUNT_SAMPLES, 1)
    ids = app.visible_set()
    fb, st = app.render_frame()
    n_plugin = int(app.stats().samples)
    order = app.node_order()
    app.set_option(vrc.OPT_KERNEL, vrc.KERNEL_GRID_DDA)
    fb_dda, _ = app.render_frame()
s = orc.build_scene(voxels=(vox, vox, vox), block=block, viewport=(W, H), ids=ids, spin=spin, eye=eye,
                    volume=volume, alpha=0.3, order=order)
want, n_want = orc.oracle_render(s, threads=8)
tb = orc.budget_of(want)


def report(name, got, n):
    d = np.abs(got - want).max(-1)
    ex = d - (5e-5 + 2 * tb)
    ys, xs = np.nonzero(ex > 0)
    print("%-28s samples %d (oracle %d) violations %d %s" % (name, n, n_want, len(ys), list(zip(xs.tolist(), ys.tolist()))[:6]))
    for x, y in list(zip(xs.tolist(), ys.tolist()))[:3]:
        print("    (%d,%d) got %s want %s" % (x, y, got[y, x], want[y, x]))


report("plugin AUTO", fb, n_plugin)
report("plugin GRID_DDA", fb_dda, -1)
with GpuScene(s) as g:
    for k, nm in ((vrc.KERNEL_AUTO, "C ABI AUTO"), (vrc.KERNEL_REFERENCE_ORDER, "C ABI REFERENCE_ORDER"), (vrc.KERNEL_GRID_DDA, "C ABI GRID_DDA")):
        for stepping in (1, 0):
            got, n, stt = g.render(kernel=k, stepping=stepping)
            report("%s stepping %d (variant %d)" % (nm, stepping, stt.kernel_variant), got, n)
print("sorted ids equal to visible-set order?", s.sorted_ids == list(ids))
