import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
torch.cuda.init()
from libre_amd import driver, sortfirst
lay = sortfirst.band_layout(1024, 8, 4)
app = driver.App("mem://#1024,1024,1024,128", 1024, 1024, synchronous=True, min_lod=3, max_lod=3, gpu_cache_mb=3072)
i = np.arange(256, dtype=np.float32) / np.float32(255.0)
app.set_colormap(np.ascontiguousarray(np.stack([i, i, i, np.float32(0.05) * i], axis=1)))
app.set_bands(lay[0])
for _ in range(5): app.render_frame(readback=False)
torch.cuda.synchronize()
os.environ["LIVRE_HIP_TRACE"]="1"
for _ in range(3):
    t0=time.perf_counter(); app.render_frame(readback=False); t1=time.perf_counter()
    print("render_frame call %.1f us" % ((t1-t0)*1e6), flush=True)
    torch.cuda.synchronize()
