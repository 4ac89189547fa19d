"""Developer probe: frames/s with every frame copied to pinned host memory, for different ways of issuing the copy
(same stream as the kernel / one copy stream for all / more frames in flight)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402  (before the library: one HIP runtime)
from libre_amd import driver  # noqa: E402

W = H = 1024
i = np.arange(256, dtype=np.float32) / np.float32(255.0)
tf = np.ascontiguousarray(np.stack([i, i, i, np.float32(0.05) * i], axis=1))
app = driver.App("mem://#1024,1024,1024,128", W, H, synchronous=True, min_lod=3, max_lod=3, gpu_cache_mb=3072)
app.set_colormap(tf)
app.render_frame(readback=False)


def run(K, mode, n=150):
    streams = [torch.cuda.Stream() for _ in range(K)]
    fbs = [torch.zeros((H, W, 4), dtype=torch.float32, device="cuda") for _ in range(K)]
    host = [torch.zeros((H, W, 4), dtype=torch.float32).pin_memory() for _ in range(K)]
    copy_stream = torch.cuda.Stream()
    done = [None] * K
    app.set_frames_in_flight(K)
    for k in range(K):
        app.select_slot(k)
        app.set_stream(streams[k].cuda_stream)
        app.set_framebuffer(fbs[k].data_ptr())

    def frame(j):
        k = j % K
        with torch.cuda.stream(streams[k]):
            if mode == "copy_stream" and done[k] is not None:
                streams[k].wait_event(done[k])  # the copy of this slot's previous frame has read the buffer
            app.select_slot(k)
            app.render_frame(readback=False)
            if mode == "same_stream":
                host[k].copy_(fbs[k], non_blocking=True)
            elif mode == "copy_stream":
                ev = torch.cuda.Event()
                ev.record(streams[k])
                with torch.cuda.stream(copy_stream):
                    copy_stream.wait_event(ev)
                    host[k].copy_(fbs[k], non_blocking=True)
                    d = torch.cuda.Event()
                    d.record(copy_stream)
                    done[k] = d

    for j in range(2 * K):
        frame(j)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for j in range(n):
        frame(j)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("%-12s %d frames in flight: %.0f frames/s (%.3f ms per frame)" % (mode, K, n / dt, dt / n * 1e3), flush=True)


for K in (1, 2, 3, 4, 6):
    for mode in ("none", "same_stream", "copy_stream"):
        run(K, mode)
app.close()
