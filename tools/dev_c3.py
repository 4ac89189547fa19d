"""Developer probe of BASELINE configuration C3 in the size that fits a gpurun call: a raw://
uint16 volume file, bricked on demand with an LOD octree (extension of the reference's
single-brick raw source), asynchronous brick upload overlapped with the march.
usage: python tools/dev_c3.py [N=1024] [block=128]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from libre_amd import driver, vrc  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
B = int(sys.argv[2]) if len(sys.argv) > 2 else 128
# texture cache that holds the finest level (bricks with overlap, 2 bytes per voxel) and a CPU cache of the same size:
# 2048^3 -> 24 GiB, an atlas of more than 2^32 voxels (64-bit slot bases)
GPU_MB = int(sys.argv[3]) if len(sys.argv) > 3 else max(6144, int(1.2 * 2 * (N / B) ** 3 * (B + 8) ** 3 / 2 ** 20))
CPU_MB = max(8192, GPU_MB + 2048)
path = "/tmp/vol_u16_%d.raw" % N
if not os.path.exists(path) or os.path.getsize(path) != 2 * N ** 3:
    t0 = time.perf_counter()
    with open(path, "wb") as f:
        xy = (np.arange(N, dtype=np.uint64)[None, :] + np.uint64(N) * np.arange(N, dtype=np.uint64)[:, None])
        for z0 in range(0, N, 16):
            idx = xy[None, :, :] + np.uint64(N * N) * np.arange(z0, z0 + 16, dtype=np.uint64)[:, None, None]
            h = (idx & np.uint64(0xFFFFFFFF)).astype(np.uint32)
            h ^= h >> np.uint32(16)
            h *= np.uint32(0x7FEB352D)
            h ^= h >> np.uint32(15)
            h *= np.uint32(0x846CA68B)
            h ^= h >> np.uint32(16)
            (h >> np.uint32(16)).astype(np.uint16).tofile(f)
    print("wrote %s (%.1f GB) in %.1f s" % (path, 2 * N ** 3 / 1e9, time.perf_counter() - t0), flush=True)
uri = "raw://%s#%d,%d,%d,uint16,%d" % (path, N, N, N, B)
tf = [[i / 255.0, i / 255.0, i / 255.0, 0.05 * i / 255.0] for i in range(256)]
brick_bytes = 2 * (B + 8) ** 3

# 1. synchronous, leaves only: the whole finest level through the out-of-core path
probe = driver.App(uri, 1024, 1024)
depth = probe.volume_info()["depth"]
probe.close()
with driver.App(uri, 1024, 1024, synchronous=True, min_lod=depth - 1, max_lod=depth - 1,
                gpu_cache_mb=GPU_MB, cpu_cache_mb=CPU_MB) as app:
    app.set_colormap(tf)
    t0 = time.perf_counter()
    _, st = app.render_frame(readback=False)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    nb = st.n_available
    print("sync leaves: %d bricks (%.2f GB) cut from the file and uploaded in %.1f ms = %.2f GB/s" %
          (nb, nb * brick_bytes / 1e9, dt * 1e3, nb * brick_bytes / dt / 1e9), flush=True)
    app.set_option(vrc.OPT_COUNT_SAMPLES, 1)
    app.render_frame(readback=False)
    n = app.stats().samples
    app.set_option(vrc.OPT_COUNT_SAMPLES, 0)
    for flt, name in ((0, "point"), (1, "trilinear")):
        app.set_option(vrc.OPT_FILTER, flt)
        for _ in range(3):
            app.render_frame(readback=False)
        app.stats()
        for _ in range(20):
            app.render_frame(readback=False)
        torch.cuda.synchronize()
        s = app.stats()
        ms = s.kernel_ms_sum / max(1, s.kernel_launches)
        print("uint16 %s kernel: %.3f ms per frame = %.1f Gsamples/s" % (name, ms, n / ms / 1e6), flush=True)

# 2. asynchronous, LOD cut by screen-space error, camera orbiting: upload overlapped with the march
with driver.App(uri, 1024, 1024, synchronous=False, sse=4.0, gpu_cache_mb=GPU_MB, cpu_cache_mb=CPU_MB) as app:
    app.set_colormap(tf)
    t0 = time.perf_counter()
    frames = 0
    first_complete = None
    while time.perf_counter() - t0 < 20.0:
        app.set_camera(spin=(0.002 * frames, 0.001 * frames))
        _, st = app.render_frame(readback=False)
        frames += 1
        if st.n_not_available == 0 and first_complete is None and frames > 1:
            first_complete = (time.perf_counter() - t0, frames, st.n_available)
            break
    torch.cuda.synchronize()
    if first_complete:
        t, fr, nb = first_complete
        print("async + LOD cut (sse 4): %d bricks resident after %.1f ms; %d frames drawn meanwhile (%.0f fps)" %
              (nb, t * 1e3, fr, fr / t), flush=True)
    else:
        print("async: not complete after 20 s (%d frames)" % frames)
    tex, data = app.cache_stats()
    print("texture cache:", tex, "data cache:", data)
