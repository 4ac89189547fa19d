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


def variants():
    """(scene, key, oracle options): the other frames this build defines or mirrors -- the glRaycaster twin
    (fragRaycast.glsl), its jittered supersampling (round 3), the trilinear filter (extension)"""
    for name in ("hash64_spin", "hash_clip", "mem_inside"):
        yield name, "gl", dict(variant=1)
        yield name, "trilinear", dict(filter_mode=1)
    yield "hash64_spin", "gl_spp4", dict(variant=1, spp=4)
    yield "mem_ragged", "gl_spp3", dict(variant=1, spp=3)


def variant_scene(name, kw):
    import ctypes as C
    s = scenes.get(name)
    if "spp" in kw:
        s.render = orc.RenderData(s.render.samplesPerRay, kw["spp"], s.render.maxSamplesPerRay, s.render.datatype,
                                  (C.c_float * 2)(*s.render.dataSourceRange))
    return s


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
    for name, key, kw in variants():
        fb, n = orc.oracle_render(variant_scene(name, kw), threads=8, **{k: v for k, v in kw.items() if k != "spp"})
        out[name + "__" + key] = fb.astype(np.float32)
        out[name + "__" + key + "_samples"] = np.array([n], dtype=np.uint64)
    np.savez_compressed(os.path.join(HERE, "frames.npz"), **out)
    print("wrote", len(out) // 2, "frames")


if __name__ == "__main__":
    main()
