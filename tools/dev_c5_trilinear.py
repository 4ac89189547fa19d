"""Developer probe: per-ray LOD with the trilinear filter, samples by gathers (vrc_k_raycast_raylod) against the
LDS-staged form (vrc_k_raycast_lds<.,true,.,true>) and the tap-packed atlas (vrc_k_raycast_raylod<...,5|6,unsigned int,.>), same frames.
usage: python tools/dev_c5_trilinear.py [N=1024] [block=128] [viewport=1024]"""
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


def measure(app, frames=10):
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
print("%s, viewport %d^2, trilinear; kernel ms per frame / Gsamples per frame" % (uri, V))
L = vrc.load_library()
for alpha, what in ((1.0, "ERT (alpha 1.0)"), (0.05, "no ERT (alpha 0.05)")):
    for sse in (1.0, 4.0):
        with driver.App(uri, V, V, synchronous=True, sse=sse, gpu_cache_mb=3072, cpu_cache_mb=16384) as app:
            app.set_colormap(tf(alpha))
            app.set_option(vrc.OPT_FILTER, 1)
            app.set_ray_lod(True)
            for name, eye, spin in views:
                app.set_camera(position=eye, spin=spin)
                app.set_option(vrc.OPT_KERNEL, vrc.KERNEL_GRID_DDA)
                ms0, n0, st0 = measure(app)
                k0 = L.vrc_last_kernel().decode()
                app.set_option(vrc.OPT_KERNEL, vrc.KERNEL_LDS)
                ms1, n1, st1 = measure(app)
                k1 = L.vrc_last_kernel().decode()
                app.set_option(vrc.OPT_KERNEL, vrc.KERNEL_PACKED)
                ms2, n2, st2 = measure(app)
                k2 = L.vrc_last_kernel().decode()
                print("%-20s sse %.0f %-24s gathers %.3f ms %6.3f Gs | staged %.3f ms (x%.2f) | packed %.3f ms %6.3f Gs (x%.2f; ray_lod %d)  [%s | %s | %s]" % (
                    what, sse, name, ms0, n0 / 1e9, ms1, ms0 / ms1, ms2, n2 / 1e9, ms0 / ms2, st2.ray_lod, k0.split("<")[0], k1.split("<")[0], k2[:60]),
                    flush=True)
