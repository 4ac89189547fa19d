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
    passes, todo0, gathers, fast, cand0, lanes, rounds, part0 = list(out)
    print("samples %d kernel_ms %.3f" % (n, st.kernel_ms))
    print("rounds %d  box passes %d (%.2f per round, %d of them fast)  rounds that ended in the gather path %d" %
          (rounds, passes, passes / max(rounds, 1), fast, gathers))
    print("first pass of a round, lanes: with steps to take %.1f, of them in the lead's brick %.1f, of them in the window %.1f"
          % (todo0 / max(rounds, 1), cand0 / max(rounds, 1), part0 / max(rounds, 1)))
    print("mean participating lanes per pass %.1f  samples/pass %.1f  samples/round %.1f" %
          (lanes / max(passes, 1), n / max(passes, 1), n / max(rounds, 1)))
