import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import orc
from libre_amd import driver as drv, vrc
cases = {
 "inside_x": dict(position=(0.1, 0.0, 0.2), lookat=(1.0, 0.0, 0.2)),
 "flat": dict(position=(0.3, 0.0, 0.3), lookat=(0.3, 0.0, -1.0)),
 "diag": dict(position=(-0.2, -0.1, -0.25), lookat=(1.0, 1.0, 1.0)),
}
for name, uri in (("inside_x", "hash://#128,128,128,16"), ("inside_x", "hash://#256,256,256,32"), ("flat", "hash://#192,128,64,16"), ("flat", "hash://#96,64,32,8"),
                  ("diag", "hash://#128,128,128,16"), ("diag", "hash://#256,256,256,32")):
    for sse in (0.6, 0.25, 0.1):
        for mb in (256, 1, 2, 4):
            try:
                with drv.App(uri, 96, 80, synchronous=True, sse=sse, gpu_cache_mb=mb) as app:
                    app.set_camera(**cases[name])
                    app.set_colormap(orc.linear_ramp_tf(0.1))
                    app.set_ray_lod(True)
                    fb, st = app.render_frame()
                    print(name, uri, "sse", sse, "mb", mb, "avail", st.n_available, "passes", st.n_passes, "raylod", st.ray_lod, "amax %.3f" % fb[..., 3].max(), flush=True)
            except Exception as e:
                print(name, uri, sse, mb, "ERR", str(e)[:100], flush=True)
