"""Regenerates tests/golden/frames_ray_lod.npz: oracle frames of the per-ray adaptive LOD extension (BASELINE C5).
The extension has no counterpart in the reference; these vectors pin its definition (oracle/livre_oracle.c,
raycast_pixel_ray_lod) against drift between rounds.  Run:  python tests/golden/make_golden_ray_lod.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import orc  # noqa: E402
import test_ray_lod  # noqa: E402


def cases():
    for sse in (1.5, 3.0):
        s = test_ray_lod._hierarchy(viewport=(48, 40), volume="hash", spin=(0.4, 0.3))
        yield "hash64_sse%g" % sse, s, sse
    s, _ = test_ray_lod._uvf_hierarchy((56, 48), spin=(0.6, 0.3), alpha=0.3)
    yield "uvf_mouse_sse1.2", s, 1.2


def main():
    out = {}
    for name, s, sse in cases():
        fb, n = orc.oracle_render(s, threads=8, ray_lod=(sse, orc.world_space_per_pixel(s)))
        out[name] = fb.astype(np.float32)
        out[name + "__samples"] = np.array([n], dtype=np.uint64)
        # the trilinear filter on the same hierarchies (round 3: the LDS-staged kernel renders these)
        fb, n = orc.oracle_render(s, threads=8, ray_lod=(sse, orc.world_space_per_pixel(s)), filter_mode=1)
        out[name + "__trilinear"] = fb.astype(np.float32)
        out[name + "__trilinear_samples"] = np.array([n], dtype=np.uint64)
    np.savez_compressed(os.path.join(HERE, "frames_ray_lod.npz"), **out)
    print("wrote", len(out) // 2, "frames")  # (point-sampled and trilinear per case)


if __name__ == "__main__":
    main()
