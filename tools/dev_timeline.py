"""Developer look at what ONE launch of the judged kernel leaves idle: a -DVRC_WG_TIMELINE build
stamps every wave's start and end with the chip-wide 100 MHz clock and its XCD / CU; this prints how many waves are
resident over the launch's life, per XCD, and where the wave-slot time goes.
usage: python tools/dev_timeline.py variants/timeline.so [--volume mem|hash] [--spin a b]"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import orc  # noqa: E402
from gpu_run import GpuScene  # noqa: E402
from libre_amd import vrc  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("lib")
    ap.add_argument("--volume", default="mem")
    ap.add_argument("--spin", type=float, nargs=2, default=(0.0, 0.0))
    ap.add_argument("--kernel", type=int, default=vrc.KERNEL_GRID_DDA)
    ap.add_argument("--filter", type=int, default=0)
    ap.add_argument("--bins", type=int, default=40)
    a = ap.parse_args()
    L = vrc.load_library(a.lib)
    s = orc.build_scene(voxels=(1024,) * 3, block=128, viewport=(1024, 1024), volume=a.volume, spin=tuple(a.spin))
    dt = np.dtype([("start", "<u8"), ("end", "<u8"), ("hw", "<u4"), ("xcc", "<u4")])
    with GpuScene(s, lib=L) as g:
        for _ in range(5):
            fb, n, st = g.render(kernel=a.kernel, filter_mode=a.filter, count=False)
        print("kernel %s: %.3f ms by HIP events" % (L.vrc_last_kernel().decode(), st.kernel_ms))
        buf = np.zeros(1 << 16, dtype=dt)
        L.vrc_dev_read_timeline.argtypes = [C.c_void_p, C.c_size_t]
        assert L.vrc_dev_read_timeline(buf.ctypes.data, buf.nbytes) == 0
    w = buf[buf["end"] > 0]
    t0, t1 = int(w["start"].min()), int(w["end"].max())
    span = (t1 - t0) * 10e-6  # ms (100 MHz)
    cu = (w["hw"] >> 8) & 0xF
    se = (w["hw"] >> 13) & 0x7
    sh = (w["hw"] >> 12) & 0x1
    xcc = w["xcc"] & 0xF
    simd = (w["hw"] >> 4) & 0x3
    where = xcc.astype(np.int64) * 1024 + se * 64 + sh * 32 + cu * 2
    print("%d waves stamped, span %.3f ms; XCDs %s; distinct (xcd, se, sh, cu) %d" % (
        len(w), span, sorted(set(xcc.tolist())), len(set(where.tolist()))))
    dur = (w["end"] - w["start"]) * 10e-6
    print("wave life: min %.3f median %.3f max %.3f ms; sum of lives %.1f ms = %.1f %% of %d wave slots x span" % (
        dur.min(), np.median(dur), dur.max(), dur.sum(), 100 * dur.sum() / (256 * 4 * 5 * span), 256 * 4 * 5))
    last_start = (int(w["start"].max()) - t0) * 10e-6
    print("last wave starts at %.3f ms (%.0f %% of the span): from then on the queue is dry" % (last_start, 100 * last_start / span))
    edges = np.linspace(t0, t1, a.bins + 1)
    print("resident waves over time (bin centre in ms: all | per XCD)")
    for b in range(a.bins):
        lo, hi = edges[b], edges[b + 1]
        mid = 0.5 * (lo + hi)
        live = (w["start"] <= mid) & (w["end"] > mid)
        per = [int((live & (xcc == x)).sum()) for x in range(8)]
        print("%6.3f  %5d | %s" % ((mid - t0) * 10e-6, int(live.sum()), " ".join("%4d" % p for p in per)))
    # per XCD: when does it run dry and when does it end
    for x in range(8):
        m = xcc == x
        if m.any():
            print("XCD %d: %5d waves, lives sum %.1f ms, last start %.3f ms, last end %.3f ms" % (
                x, int(m.sum()), dur[m].sum(), (int(w["start"][m].max()) - t0) * 10e-6, (int(w["end"][m].max()) - t0) * 10e-6))


if __name__ == "__main__":
    main()
