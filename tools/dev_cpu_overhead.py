"""Developer experiment: host time of one lvh_app_render_frame call (enqueue only)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from libre_amd import driver
i = np.arange(256, dtype=np.float32) / np.float32(255.0)
tf = np.ascontiguousarray(np.stack([i, i, i, np.float32(0.05) * i], axis=1))
app = driver.App("mem://#1024,1024,1024,128", 1024, 1024, synchronous=True, min_lod=3, max_lod=3, gpu_cache_mb=3072)
app.set_colormap(tf)
app.render_frame(readback=False)
app.synchronize()
ts = []
for _ in range(50):
    app.synchronize()
    t0 = time.perf_counter()
    app.render_frame(readback=False)
    ts.append(time.perf_counter() - t0)
ts = np.array(ts) * 1e6
print("host time per render_frame call: median %.1f us, min %.1f us, max %.1f us" % (np.median(ts), ts.min(), ts.max()))
app.close()
