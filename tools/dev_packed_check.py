"""Developer check: the trilinear frame of the tap-packed kernel against the LDS-staged and the gather form at
BASELINE C2's size (noise volume), per band of rows -- is any slot range of the packed atlas read wrongly?"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import orc  # noqa: E402
from gpu_run import GpuScene  # noqa: E402
from libre_amd import vrc  # noqa: E402

lib = vrc.load_library(sys.argv[1]) if len(sys.argv) > 1 else None
s = orc.build_scene(voxels=(1024,) * 3, block=128, viewport=(1024, 1024), volume="hash")
with GpuScene(s, lib=lib) as g:
    print(g.info())
    p, n_p, _ = g.render(kernel=vrc.KERNEL_PACKED, filter_mode=1)
    l, n_l, _ = g.render(kernel=vrc.KERNEL_LDS, filter_mode=1)
    d, n_d, _ = g.render(kernel=vrc.KERNEL_GRID_DDA, filter_mode=1)
    print("samples packed %d staged %d gathers %d" % (n_p, n_l, n_d))
    for name, a, b in (("packed vs staged", p, l), ("packed vs gathers", p, d), ("staged vs gathers", l, d)):
        e = np.abs(a - b).max(axis=2)
        print("%-18s max %.3g mean %.3g; per 128-row band max: %s" % (name, e.max(), e.mean(),
              " ".join("%.1e" % e[r:r + 128].max() for r in range(0, 1024, 128))))
        print("%-18s per 128-column band max: %s" % ("", " ".join("%.1e" % e[:, c:c + 128].max() for c in range(0, 1024, 128))))
