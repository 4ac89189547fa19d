"""Developer experiment: throughput of ONE rank's share of a sort-first frame (N ranks emulated on
one GPU, only rank 0's row bands rendered, no gather) against the number of frames in flight."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from libre_amd import driver, sortfirst

ap = argparse.ArgumentParser()
ap.add_argument("--world", type=int, nargs="+", default=[2, 4, 8])
ap.add_argument("--inflight", type=int, nargs="+", default=[1, 2, 3, 4, 6, 8])
ap.add_argument("--bands", type=int, default=4)
ap.add_argument("--frames", type=int, default=300)
a = ap.parse_args()
W = H = 1024
torch.cuda.init()
i = np.arange(256, dtype=np.float32) / np.float32(255.0)
tf = np.ascontiguousarray(np.stack([i, i, i, np.float32(0.05) * i], axis=1))
for world in a.world:
    lay = sortfirst.band_layout(H, world, a.bands)
    rows = sum(h for _, h in lay[0])
    for K in a.inflight:
        app = driver.App("mem://#1024,1024,1024,128", W, H, synchronous=True, min_lod=3, max_lod=3, gpu_cache_mb=3072)
        app.set_colormap(tf)
        app.set_bands(lay[0])
        app.set_frames_in_flight(K)
        streams = [torch.cuda.Stream() for _ in range(K)]
        fbs = [torch.zeros((rows, W, 4), dtype=torch.float32, device="cuda") for _ in range(K)]
        for k in range(K):
            app.select_slot(k)
            app.set_stream(streams[k].cuda_stream)
            app.set_framebuffer(fbs[k].data_ptr())

        def frame(n):
            k = n % K
            with torch.cuda.stream(streams[k]):
                app.select_slot(k)
                app.render_frame(readback=False)

        for n in range(3 * K):
            frame(n)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for n in range(a.frames):
            frame(n)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("world %d (rank 0: %d rows) frames in flight %d: %.1f frames/s, %.3f ms per frame" %
              (world, rows, K, a.frames / dt, dt / a.frames * 1e3), flush=True)
        app.close()
