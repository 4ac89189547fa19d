"""Developer probe of the brick upload path (run on the GPU box): where does the first frame's
time go -- brick generation (mem://), staging copy, DMA + repack."""
import ctypes as C
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from libre_amd import vrc  # noqa: E402

L = vrc.load_library()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
S = 136
ctx = C.c_void_p()
vrc.check(L, L.vrc_ctx_create(0, C.byref(ctx)))
pool = C.c_void_p()
vrc.check(L, L.vrc_pool_create(ctx, 1, 0, 0, 1, vrc.u32x3(S, S, S), (N + 64) * S ** 3, C.byref(pool)))
bricks = [np.full((S, S, S), i & 255, dtype=np.uint8) for i in range(8)]

t0 = time.perf_counter()
for i in range(64):
    b = np.empty((S, S, S), dtype=np.uint8)
    b.fill(i & 255)
dt = time.perf_counter() - t0
print("host fill of a fresh 2.4 MiB buffer: %.3f ms per brick (%.1f GB/s)" % (dt / 64 * 1e3, 64 * S ** 3 / dt / 1e9))


def upload(n, tid, nthreads, slots):
    size = vrc.u32x3(S, S, S)
    for i in range(tid, n, nthreads):
        slot = vrc.f32x3()
        vrc.check(L, L.vrc_pool_copy_to_slot(pool, bricks[i & 7].ctypes.data, size, slot))
        slots.append((slot[0], slot[1], slot[2]))


for nthreads in (1, 2, 4, 8):
    slots = []
    ths = [threading.Thread(target=upload, args=(N, t, nthreads, slots)) for t in range(nthreads)]
    t0 = time.perf_counter()
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    vrc.check(L, L.vrc_pool_synchronize(pool))
    dt = time.perf_counter() - t0
    print("copy_to_slot x%d, %d thread(s): %.1f ms = %.2f GB/s" % (N, nthreads, dt * 1e3, N * S ** 3 / dt / 1e9))
    for s in slots:
        L.vrc_pool_release_slot(pool, vrc.f32x3(*s))
