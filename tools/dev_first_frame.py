"""Developer probe: first frame (all bricks uploaded through the plugin path) and the
asynchronous mode's time-to-complete, for a number of loader threads.
usage: LIVRE_HIP_UPLOAD_THREADS=N python tools/dev_first_frame.py [sync|async]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401  (device runtime initialisation order as in bench.py)
from libre_amd import driver  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "sync"
uri = "mem://#1024,1024,1024,128"
app = driver.App(uri, 1024, 1024, synchronous=(mode == "sync"), min_lod=3, max_lod=3,
                 gpu_cache_mb=3072, cpu_cache_mb=4096)
app.set_colormap([[i / 255.0, i / 255.0, i / 255.0, 0.05 * i / 255.0] for i in range(256)])
t0 = time.perf_counter()
if mode == "sync":
    app.render_frame(readback=False)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("threads %s sync first frame %.1f ms = %.2f GB/s of bricks" % (
        os.environ.get("LIVRE_HIP_UPLOAD_THREADS", "default"), dt * 1e3, 512 * 136 ** 3 / dt / 1e9))
else:
    frames = 0
    while True:
        app.render_frame(readback=False)
        frames += 1
        s = app.stats()
        if s.n_not_available == 0 and frames > 1:
            break
        if time.perf_counter() - t0 > 60:
            break
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("threads %s async: all bricks resident after %.1f ms, %d frames rendered meanwhile (%.0f fps), "
          "%.2f GB/s of bricks" % (os.environ.get("LIVRE_HIP_UPLOAD_THREADS", "default"), dt * 1e3, frames,
                                   frames / dt, 512 * 136 ** 3 / dt / 1e9))
