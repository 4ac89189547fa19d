"""Sort-first exchange through the C ABI with more than one rank on ONE GPU: the ranks are threads of this process,
each with its own plugin instance (libre_amd.driver.App) and its own communicator, and RCCL is replaced by the test
double tests/host_san/fake_rccl.cpp (VRC_RCCL_LIBRARY) that moves the bands with device copies and fails on any
send without its receive.  Everything but RCCL itself is the product path bench.py runs at N > 1:
App.set_bands -> render -> sortfirst.AbiTileGather -> vrc_gather_tiles.  Run by tests/test_gpu_host.py in a process
of its own (the library binds its RCCL once).  usage: gpu_fake_rccl_gather.py WORLD BANDS_PER_RANK BATCH"""
import ctypes as C
import os
import sys
import threading

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    world, bands_per_rank, batch = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    assert os.environ.get("VRC_RCCL_LIBRARY"), "set VRC_RCCL_LIBRARY to the test double"
    import torch
    from libre_amd import driver, sortfirst
    W, H = 200, 168
    uri = "mem://#128,128,128,32"
    i = np.arange(256, dtype=np.float32) / np.float32(255.0)
    tf = np.ascontiguousarray(np.stack([i, i, i, np.float32(0.3) * i], axis=1))
    spins = [(0.1 * k, 0.05 * k) for k in range(2 * batch)]

    def make_app():
        app = driver.App(uri, W, H, synchronous=True, min_lod=2, max_lod=2, gpu_cache_mb=64)
        app.set_colormap(tf)
        return app

    # the frames one rank renders whole
    full = []
    app = make_app()
    for sp in spins:
        app.set_camera(spin=sp)
        fb, _ = app.render_frame(readback=True)
        full.append(fb.copy())
    app.close()
    assert all(f[..., 3].max() > 0.1 for f in full)

    layout = sortfirst.band_layout(H, world, bands_per_rank)
    uid = driver.comm_unique_id()
    errors, result = [], {}
    barrier = threading.Barrier(world)

    def rank_main(rank):
        try:
            app = make_app()
            app.comm_create(rank, world, uid)
            app.set_bands(layout[rank])
            g = sortfirst.AbiTileGather(app, layout, W, rank, "cuda", batch)
            out = []
            for half in range(2):
                for k in range(batch):
                    app.set_camera(spin=spins[half * batch + k])
                    app.set_framebuffer(g.send[half, k].data_ptr())
                    app.render_frame(readback=False)
                g.gather(half, batch)  # on the renderer's stream: ordered after the renders
                app.synchronize()
                barrier.wait(timeout=60)  # the display rank's receives are done: senders may reuse their buffers
                if rank == 0:
                    out += [f.cpu().numpy().copy() for f in g.assemble(batch)]
            if rank == 0:
                result["frames"] = out
            barrier.wait(timeout=60)
            app.close()
        except Exception as e:  # noqa: BLE001
            errors.append("rank %d: %r" % (rank, e))
            barrier.abort()

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=180)
    assert not any(t.is_alive() for t in threads), "a rank is stuck"
    assert not errors, errors
    for k, (got, want) in enumerate(zip(result["frames"], full)):
        # a band is the same rays as the rows of the full frame (vrc_set_row_map): bit for bit
        assert got.shape == want.shape and (got == want).all(), "frame %d differs, max |d| = %g" % (
            k, np.abs(got - want).max())
    fake = C.CDLL(os.environ["VRC_RCCL_LIBRARY"])
    assert fake.fake_rccl_leftovers() == 0, "sends without a receive"
    print("ok: %d ranks x %d bands, %d frames per exchange, %d frames bit-identical to the one-rank frames"
          % (world, bands_per_rank, batch, len(full)))


if __name__ == "__main__":
    main()
