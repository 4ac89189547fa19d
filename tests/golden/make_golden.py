"""Regenerates tests/golden/frames.npz: oracle frames of the named small scenes.

The reference itself cannot be built or run here (needs nvcc, Boost, GLEW and
ten un-vendored subprojects), and it holds no golden frame of its own, so these vectors are
outputs of the CPU restatement in oracle/ ("parity unpinned" by the reference).  They pin the
oracle against regressions and give the GPU tests a committed target.
Run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import orc  # noqa: E402
import scenes  # noqa: E402


def main():
    out = {}
    for name in sorted(scenes.SCENES):
        s = scenes.get(name)
        fb, n = orc.oracle_render(s, threads=8)
        out[name] = fb.astype(np.float32)
        out[name + "__samples"] = np.array([n], dtype=np.uint64)
    s = scenes.nucleon_scene()
    fb, n = orc.oracle_render(s, threads=8)
    out["nucleon"] = fb
    out["nucleon__samples"] = np.array([n], dtype=np.uint64)
    np.savez_compressed(os.path.join(HERE, "frames.npz"), **out)
    print("wrote", len(out) // 2, "frames")


if __name__ == "__main__":
    main()
