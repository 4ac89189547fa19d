"""Developer tool: round statistics of the LDS-staged kernel (needs the VRC_LDS_STATS build:
tools/build_variants.sh ldsstats -DVRC_LDS_STATS; VRC_HIP_LIB=variants/libvrc_hip_ldsstats.so)."""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import orc  # noqa: E402
from gpu_run import GpuScene  # noqa: E402
from libre_amd import vrc  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--voxels", type=int, default=1024)
ap.add_argument("--block", type=int, default=128)
ap.add_argument("--viewport", type=int, default=1024)
ap.add_argument("--spin", type=float, nargs=2, default=(0.0, 0.0))
ap.add_argument("--filter", type=int, default=0)
a = ap.parse_args()
s = orc.build_scene(voxels=(a.voxels,) * 3, block=a.block, viewport=(a.viewport,) * 2, spin=tuple(a.spin))
with GpuScene(s) as g:
    fn = g.L.vrc_debug_lds_stats
    fn.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
    out = (C.c_ulonglong * 8)()
    fn(out, 1)
    fb, n, st = g.render(kernel=vrc.KERNEL_LDS, filter_mode=a.filter, count=True)
    fn(out, 1)
    rounds, sumg, shrunk, fits, aloop, lanes, vol, dz = list(out)
    print("samples %d kernel_ms %.3f" % (n, st.kernel_ms))
    print("rounds %d  mean g %.2f  rounds with shrunk lane set %d  fit iterations/round %.2f" %
          (rounds, sumg / max(rounds, 1), shrunk, fits / max(rounds, 1)))
    print("A-loop iterations %d  mean participating lanes %.1f  samples/round %.1f" %
          (aloop, lanes / max(rounds, 1), n / max(rounds, 1)))
    print("mean box volume %.0f B  mean dz %.2f" % (vol / max(rounds, 1), dz / max(rounds, 1)))
    lg = g.L.vrc_debug_lds_log
    buf = (C.c_uint * (64 * 16))()
    lg(buf)
    for i in range(40):
        print("fit %2d:" % i, list(buf[i * 16:(i + 1) * 16]))
