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
    out = (C.c_ulonglong * 8)()
    fn = g.L.vrc_debug_lds_stats if hasattr(g.L, 'vrc_debug_lds_stats') else (lambda *a: 0)
    if hasattr(g.L, 'vrc_debug_lds_stats'):
        fn.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
    fn(out, 1)
    pout = (C.c_ulonglong * 8)()
    ph = getattr(g.L, "vrc_debug_lds_phases", None) if hasattr(g.L, "vrc_debug_lds_phases") else None
    if ph is not None:
        ph.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
        ph(pout, 1)
    fb, n, st = g.render(kernel=vrc.KERNEL_LDS, filter_mode=a.filter, count=False if ph is not None else True)
    if ph is not None:
        ph(pout, 1)
    fn(out, 1)
    passes, todo0, gathers, walks, brick0, dy0, dz0, part0 = list(out)
    print("samples %d kernel_ms %.3f" % (n, st.kernel_ms))
    print("box passes %d  walk iterations %d  rounds that ended in the gather path %d  samples/pass %.1f" %
          (passes, walks, gathers, n / max(passes, 1)))
    r = max(passes, 1)  # one pass per round in the product build
    print("first pass of a round, lanes: with steps to take %.1f, in the lead's brick %.1f, in the box %.1f; "
          "mean box %.1f rows x %.1f slices" % (todo0 / r, brick0 / r, part0 / r, dy0 / r, dz0 / r))
    if os.environ.get("VRC_STATS2"):  # -DVRC_LDS_STATS2: slots 5 and 6 count unrolled groups and general batches (as lane 0 saw them)
        print("unrolled groups %d (%.2f per pass), general batches %d (%.2f per pass)" % (dy0, dy0 / r, dz0, dz0 / r))
    tot = float(sum(pout)) or 1.0
    names = ["walks", "box", "stage (incl. load wait)", "march", "ERT check/replay", "gather path", "round bookkeeping", "set-up/other"]
    print("wave cycles by phase: " + ", ".join("%s %.1f%%" % (nm, 100.0 * v / tot) for nm, v in zip(names, pout)))
