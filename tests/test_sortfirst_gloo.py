"""The N > 1 path on CPU: world_size-2 (and 3, ragged) gloo process groups run the sort-first
decomposition + tile gather of libre_amd/sortfirst.py, with every rank's bands rendered by the
oracle through the per-tile off-axis frustum; the assembled frame must equal the single-rank
frame."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

import orc
import scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, height, width, bands_per_rank, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from libre_amd import sortfirst
    dist.init_process_group("gloo", rank=rank, world_size=world)
    layout = sortfirst.band_layout(height, world, bands_per_rank)
    kw = dict(scenes.SCENES["hash64_spin"])
    kw["viewport"] = (width, height)
    parts = []
    for (y0, h) in layout[rank]:
        s = orc.build_scene(tile=(0, y0, width, h, width, height), **kw)
        fb, _ = orc.oracle_render(s, threads=2)
        parts.append(torch.from_numpy(fb))
    local = torch.cat(parts, dim=0) if parts else torch.zeros((0, width, 4))
    g = sortfirst.TileGather(layout, width, rank, "cpu")
    g.gather(local.contiguous())
    if rank == 0:
        np.save(out_path, g.assemble().numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,height,bands", [(2, 48, 2), (3, 50, 2)])
def test_sort_first_assembly_equals_full_frame(tmp_path, world, height, bands):
    width = 40
    out = str(tmp_path / "frame.npy")
    mp.spawn(_worker, args=(world, _free_port(), height, width, bands, out), nprocs=world, join=True)
    got = np.load(out)
    kw = dict(scenes.SCENES["hash64_spin"])
    kw["viewport"] = (width, height)
    full, _ = orc.oracle_render(orc.build_scene(**kw), threads=4)
    assert got.shape == full.shape
    # the tile frusta are separate float matrices: equal up to rounding of the ray directions
    scenes.assert_close_frames(got, full, "assembled tiles vs full frame")


def test_band_layout_properties():
    from libre_amd import sortfirst
    for world in (1, 2, 4, 8):
        for height in (1024, 2048, 1000, 7):
            lay = sortfirst.band_layout(height, world, 4)
            rows = sorted((y0, h) for bands in lay for (y0, h) in bands)
            assert rows[0][0] == 0 and sum(h for _, h in rows) == height
            for (a, b) in zip(rows, rows[1:]):
                assert a[0] + a[1] == b[0]
    lay = sortfirst.band_layout(1024, 8, 4)
    assert all(len(b) == 4 and sum(h for _, h in b) == 128 for b in lay)
    # interleaving: every rank has a band in each quarter of the image
    assert [y0 // 256 for (y0, _) in lay[3]] == [0, 1, 2, 3]


def test_assembly_of_regular_and_ragged_layouts_without_a_process_group():
    # the display rank's placement of received bands: one strided copy for the regular
    # interleave, band by band otherwise; both must put every frame row where it belongs
    import torch
    from libre_amd import sortfirst
    for height, world, bpr in ((64, 4, 4), (1024, 8, 4), (50, 3, 4), (37, 2, 3)):
        layout = sortfirst.band_layout(height, world, bpr)
        tg = sortfirst.TileGather(layout, 8, 0, "cpu")
        assert tg.regular == (height % (world * bpr) == 0)
        for r, bands in enumerate(layout):
            rows = torch.cat([torch.arange(y0, y0 + h) for (y0, h) in bands]).float()
            tg.recv[r].copy_(rows[:, None, None].expand(-1, 8, 4))
        frame = tg.assemble()
        assert torch.equal(frame[:, 0, 0], torch.arange(height).float())


def _batched_worker(rank, world, port, height, width, bands_per_rank, batch, n_frames, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from libre_amd import sortfirst
    dist.init_process_group("gloo", rank=rank, world_size=world)
    layout = sortfirst.band_layout(height, world, bands_per_rank)
    g = sortfirst.BatchedTileGather(layout, width, rank, "cpu", batch)
    frames = []
    # frame f, row y, column x, channel c = f*1e6 + y*1e3 + x + c/10: every value names its place
    yy, xx, cc = torch.meshgrid(torch.arange(height), torch.arange(width), torch.arange(4), indexing="ij")
    base = (yy * 1000 + xx).float() + cc.float() / 10
    f = 0
    while f < n_frames:
        n = min(batch, n_frames - f)
        half = (f // batch) % 2
        for i in range(n):
            rows = torch.cat([base[y0:y0 + h] for (y0, h) in layout[rank]], dim=0) + (f + i) * 1.0e6
            g.send[half, i].copy_(rows)
        g.gather(half, n)
        if rank == 0:
            frames.append(g.assemble(n).clone())
        f += n
    if rank == 0:
        np.save(out_path, torch.cat(frames, dim=0).numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,height,bands,batch,n_frames", [(2, 48, 2, 3, 8), (3, 48, 4, 2, 5), (2, 40, 1, 4, 4)])
def test_batched_gather_places_every_band_of_every_frame(tmp_path, world, height, bands, batch, n_frames):
    # several frames per collective (BatchedTileGather): full and partial batches, both halves
    width = 12
    out = str(tmp_path / "frames.npy")
    mp.spawn(_batched_worker, args=(world, _free_port(), height, width, bands, batch, n_frames, out),
             nprocs=world, join=True)
    got = np.load(out)
    assert got.shape == (n_frames, height, width, 4)
    yy, xx, cc = np.meshgrid(np.arange(height), np.arange(width), np.arange(4), indexing="ij")
    for f in range(n_frames):
        want = (yy * 1000 + xx).astype(np.float32) + cc.astype(np.float32) / 10 + np.float32(f * 1.0e6)
        assert (got[f] == want).all(), f


def test_batched_gather_refuses_unequal_row_counts():
    from libre_amd import sortfirst
    with pytest.raises(ValueError):
        sortfirst.BatchedTileGather(sortfirst.band_layout(50, 3, 2), 8, 0, "cpu", 2)


def test_bench_starts_its_own_ranks():
    # `python bench.py --gpus N` with no launcher around it starts N ranks itself (RANK / LOCAL_RANK /
    # WORLD_SIZE / MASTER_* set, fresh processes) and relays their exit status; BENCH_DRY_RUN stops each
    # rank after the rendezvous and the control-plane exchanges (no GPU here)
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(BENCH_DRY_RUN="1", BENCH_FORCE_DEVICE="0")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "3"], env=env,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0]) == {"dry_run": True, "n_gpus": 3, "ok": True}
    # under a launcher (WORLD_SIZE set) the process is one rank: no second level of processes
    env1 = dict(env, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1"], env=env1,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and json.loads(out.stdout.strip().splitlines()[-1])["n_gpus"] == 1
    # more ranks than GPUs, no rehearsal knob: refused loudly, nothing started
    env2 = {k: v for k, v in env.items() if k != "BENCH_FORCE_DEVICE"}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], env=env2,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 2 and "GPU(s) visible" in out.stderr
