"""Developer probe of BASELINE configuration C5's kernel side on one GPU: per-ray adaptive LOD +
early ray termination against the reference's per-brick LOD cut, same volume, same camera, same
screen-space error.  (C5's input format, UVF, is covered by tests; its fixture is 1.5 MB, so the
timing uses the 1024^3 hash:// volume with its LOD pyramid.)
usage: python tools/dev_c5.py [N=1024] [block=128] [viewport=1024]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from libre_amd import driver, vrc  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
B = int(sys.argv[2]) if len(sys.argv) > 2 else 128
V = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
uri = "hash://#%d,%d,%d,%d" % (N, N, N, B)


def tf(alpha):
    return [[i / 255.0, i / 255.0, i / 255.0, alpha * i / 255.0] for i in range(256)]


def measure(app, frames=12):
    app.set_option(vrc.OPT_COUNT_SAMPLES, 1)
    _, st = app.render_frame(readback=False)
    n = app.stats().samples
    app.set_option(vrc.OPT_COUNT_SAMPLES, 0)
    for _ in range(3):
        app.render_frame(readback=False)
    app.stats()
    for _ in range(frames):
        _, st = app.render_frame(readback=False)
    torch.cuda.synchronize()
    s = app.stats()
    return s.kernel_ms_sum / max(1, s.kernel_launches), n, st


views = [("outside, eye (0,0,1.5)", (0.0, 0.0, 1.5), (0.0, 0.0)),
         ("outside, spun 30/20 deg", (0.0, 0.0, 1.5), (0.5236, 0.349)),
         ("close, eye (0,0,0.7)", (0.0, 0.0, 0.7), (0.3, 0.2)),
         ("inside, eye (0.1,0,0.3)", (0.1, 0.0, 0.3), (0.4, 0.1))]
print("%s, viewport %d^2; kernel ms per frame / Gsamples per frame / bricks" % (uri, V))
for alpha, what in ((1.0, "ERT (alpha 1.0)"), (0.05, "no ERT (alpha 0.05)")):
    for sse in (1.0, 2.0, 4.0):
        with driver.App(uri, V, V, synchronous=True, sse=sse, gpu_cache_mb=3072, cpu_cache_mb=16384) as app:
            app.set_colormap(tf(alpha))
            for name, eye, spin in views:
                app.set_camera(position=eye, spin=spin)
                app.set_ray_lod(False)
                ms0, n0, st0 = measure(app)
                app.set_ray_lod(True)
                ms1, n1, st1 = measure(app)
                print("%-20s sse %.0f %-24s per-brick %.3f ms %6.3f Gs %4d bricks | per-ray %.3f ms %6.3f Gs %4d bricks (ray_lod %d) | x%.2f" % (
                    what, sse, name, ms0, n0 / 1e9, st0.n_available, ms1, n1 / 1e9, st1.n_available, st1.ray_lod,
                    ms0 / ms1), flush=True)
