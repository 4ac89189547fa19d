#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric: Msamples/s (and frames/s) of the raycast hot path on a
1024^3 uint8 mem:// volume at a 1024^2 viewport (configs[1], "C2"), on N MI355X of one node.

One "step" = one frame: RenderPipeline("hip")::render with every brick resident in the HBM
atlas (steady state; the first, uploading frame is reported separately and never timed), i.e.
pre-render (clear) -> node-table/LUT reuse -> raycast kernel -> post-render.  The product path
is C++ host (libre_amd/host) -> C ABI (include/vrc_hip.h) -> gfx950 kernels; Python only
parses flags, barriers and prints.  N > 1: sort-first screen tiles, one process per GPU, every rank
holds the whole volume, tiles go to rank 0 over RCCL/xGMI inside the timed region through the C ABI's
vrc_gather_tiles (torch.distributed is the control plane only: barrier, timing, the communicator id).
Started by a launcher (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment) the process is one
rank; started plainly with --gpus N > 1 it starts the N ranks itself, before it touches a GPU.

Prints ONE JSON line (rank 0).  `roofline` is the HBM roofline of the raycast kernel with the
algorithmic bytes of SURVEY.md 8(d); `cpu_baseline` times the CPU oracle (oracle/, "port") on a
bounded sample of the same workload on the host cores (rank 0, N = 1 only).
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--voxels", type=int, default=1024)
    ap.add_argument("--block", type=int, default=128)
    ap.add_argument("--viewport", type=int, default=1024)
    ap.add_argument("--alpha", type=float, default=0.05, help="TF: rgba[i]=(i,i,i,alpha*i)/255")
    ap.add_argument("--spin", type=float, nargs=2, default=(0.0, 0.0))
    ap.add_argument("--bands", type=int, default=4, help="interleaved row bands per rank (N>1)")
    ap.add_argument("--frames-in-flight", type=int, default=0,
                    help="renderer instances per rank, each on its own stream; 0 = auto: 1 on one "
                         "GPU (the kernel fills the chip; the roofline is that of an undisturbed "
                         "launch), 3 when the frame is split over several GPUs and each has idle "
                         "capacity (Equalizer renders ahead too: its default latency is one frame)")
    ap.add_argument("--ray-lod-sse", type=float, default=0.0,
                    help="not the judged workload: render with the whole LOD tree and per-ray adaptive LOD at this "
                         "screen-space error (BASELINE C5's kernel side; combine with --alpha 1.0 for early ray "
                         "termination and --gpus N for its sort-first form)")
    ap.add_argument("--gather-batch", type=int, default=0,
                    help="N>1: frames per RCCL gather (sortfirst.BatchedTileGather); 0 = auto: 3 when every rank "
                         "has the same number of rows, else 1 (one gather per frame, sortfirst.TileGather)")
    ap.add_argument("--check-frames", action="store_true",
                    help="(always on for N>1) after the timing, compare the last assembled frame on rank 0 with the "
                         "full frame rendered by one application (must be bit-identical)")
    ap.add_argument("--gather", choices=["abi", "torch"], default="abi",
                    help="N>1: who moves the tiles: abi = vrc_gather_tiles (RCCL behind the C ABI, the product path); "
                         "torch = torch.distributed.gather (the round-1 path, kept as a cross-check)")
    ap.add_argument("--require-abi-gather", action="store_true",
                    help="N>1: exit non-zero instead of falling back to torch.distributed when the C-ABI tile "
                         "exchange (vrc_gather_tiles over RCCL) is unavailable or fails its pattern check")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the trilinear-extension measurement")
    ap.add_argument("--cpu-row-stride", type=int, default=1)
    ap.add_argument("--cpu-threads", type=int, default=16, help="oracle threads (GPU box CPU share)")
    return ap.parse_args()


def linear_ramp(alpha):
    import numpy as np
    i = np.arange(256, dtype=np.float32) / np.float32(255.0)
    return np.ascontiguousarray(np.stack([i, i, i, np.float32(alpha) * i], axis=1))


def cpu_baseline(a, samples_gpu_frame):
    """Oracle (CPU port of the reference algorithm) on every `stride`-th row of the same frame."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import orc
    s = orc.build_scene(voxels=(a.voxels,) * 3, block=a.block, viewport=(a.viewport,) * 2,
                        alpha=a.alpha, spin=tuple(a.spin))
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    cores = max(1, min(cores, a.cpu_threads))
    stride = max(1, a.cpu_row_stride)
    t0 = time.perf_counter()
    fb, n = orc.oracle_render(s, threads=cores, rows=(0, s.H, stride))
    dt = time.perf_counter() - t0
    del fb, np
    out = {"value": n / dt / 1e6, "unit": "Msamples/s", "cores": cores, "kind": "port",
           "sample": "oracle/livre_oracle.c on every %d-th row of the same %dx%d frame "
                     "(%d of ~%d samples, %.1f s)" % (stride, a.viewport, a.viewport, n,
                                                       samples_gpu_frame, dt)}
    try:
        out["c1_cpu"] = c1_cpu(cores)
    except Exception as e:  # noqa: BLE001
        out["c1_cpu"] = {"error": repr(e)}
    return out


def c1_cpu(threads):
    """BASELINE.json configs[0] ("C1"): memory:// 128^3 uint8, 512^2 viewport, single LOD, the CPU raycast (the reference
    has none: the oracle) -- the whole frame, timed on the host cores."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orc
    s = orc.build_scene(voxels=(128, 128, 128), block=32, viewport=(512, 512), alpha=0.05)
    orc.oracle_render(s, threads=threads, rows=(0, s.H, 64))  # (page the library in)
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        _, n = orc.oracle_render(s, threads=threads)
        dt = time.perf_counter() - t0
        best = dt if best is None or dt < best else best
    return {"workload": "C1: mem://#128,128,128,32 uint8, 512x512 viewport, leaves only (%d bricks), %d samples/ray, "
                        "linear-ramp TF alpha=0.05, default camera" % (s.n_nodes, s.render.samplesPerRay),
            "samples_per_frame": int(n), "ms_per_frame": best * 1e3, "Msamples_per_s": n / best / 1e6,
            "frames_per_s": 1.0 / best, "cores": threads, "kind": "port",
            "note": "oracle/livre_oracle.c, whole frame, best of 3"}


def self_launch(a):
    """`python bench.py --gpus N` with no launcher around it: start the N ranks here.  This process never
    initialises a GPU (torch.cuda.device_count() does not, on this image); the ranks are fresh processes."""
    import socket
    import subprocess
    import torch
    n_dev = torch.cuda.device_count()
    if "BENCH_FORCE_DEVICE" not in os.environ and n_dev < a.gpus:
        sys.stderr.write("bench.py: --gpus %d but only %d GPU(s) visible\n" % (a.gpus, n_dev))
        return 2
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    alive = list(procs)
    while alive:
        time.sleep(0.2)
        for pr in list(alive):
            code = pr.poll()
            if code is None:
                continue
            alive.remove(pr)
            if code != 0 and rc == 0:
                rc = code
                for other in alive:  # one rank failed: the others would wait for it forever
                    other.terminate()
    return rc


def _json_channel():
    """The contract is ONE JSON line on stdout.  Libraries print there too (gloo's "[Gloo] Rank 0 is connected to 1
    peer ranks", RCCL's version banner on the first communicator -- from C++, past sys.stdout): keep a private
    duplicate of the real stdout for the line and point file descriptor 1 at stderr for everybody else."""
    sys.stdout.flush()
    out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    return out


def main():
    a = parse()
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        sys.exit(self_launch(a))
    json_out = _json_channel()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world > 1 or a.gpus > 1:
            sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d: using the launcher's world\n" % (a.gpus, world))
        a.gpus = world

    import numpy as np
    import torch
    import torch.distributed as dist

    if os.environ.get("BENCH_DRY_RUN"):
        # launcher + control plane only (tests/test_sortfirst_gloo.py, no GPU): ranks rendezvous, agree on the
        # world size, exchange what the real run exchanges, and rank 0 says so
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("gloo")
            assert dist.get_world_size() == a.gpus == world
            uid = torch.full((128,), 7 if rank == 0 else 0, dtype=torch.uint8)
            dist.broadcast(uid, src=0)
            t = torch.tensor([float(rank)], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dist.barrier()
            ok = bool((uid == 7).all()) and int(t.item()) == world - 1
            dist.destroy_process_group()
        else:
            ok = True
        if rank == 0:
            print(json.dumps({"dry_run": True, "n_gpus": world, "ok": ok}), file=json_out, flush=True)
        sys.exit(0 if ok else 1)

    from libre_amd import driver, sortfirst, vrc

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (MI355X); there is no CPU fallback")
    if vrc.load_library().vrc_is_dev_build():
        # a library compiled with -DVRC_DEV_BUILD may carry timing ablations that render wrong pixels on purpose
        raise SystemExit("bench.py refuses a developer build of libvrc_hip.so (VRC_DEV_BUILD): build the product "
                         "with __graft_entry__.build()")
    # rehearsal knobs (1-GPU box): BENCH_FORCE_DEVICE=0 puts every rank on one card and
    # BENCH_BACKEND=gloo replaces RCCL; the judged runs use neither
    if "BENCH_FORCE_DEVICE" in os.environ:
        local_rank = int(os.environ["BENCH_FORCE_DEVICE"])
    backend = os.environ.get("BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    data_group = None  # torch.distributed group that moves tiles (--gather torch, or the fallback)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # control plane: barrier, max-over-ranks of the timing, the communicator id.  CPU tensors over gloo.
        dist.init_process_group("gloo")
        assert dist.get_world_size() == a.gpus == world, (dist.get_world_size(), a.gpus, world)

    def make_data_group():
        if backend == "nccl":
            return dist.new_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        return dist.group.WORLD  # rehearsal on one card: gloo moves the tiles too

    def all_max(x):
        t = torch.tensor([x], dtype=torch.float64)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def all_sum(x):
        t = torch.tensor([x], dtype=torch.int64)
        if world > 1:
            dist.all_reduce(t)
        return int(t.item())

    def all_ok(ok):
        t = torch.tensor([1 if ok else 0], dtype=torch.int64)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(t.item())

    W = H = a.viewport
    uri = "mem://#%d,%d,%d,%d" % (a.voxels, a.voxels, a.voxels, a.block)
    layout = sortfirst.band_layout(H, world, a.bands)
    bands = layout[rank]
    rows = sum(h for _, h in bands)
    K = a.frames_in_flight if a.frames_in_flight > 0 else (1 if world == 1 else 3)
    # N>1: B frames share one exchange (a collective costs its host issue time whatever it carries, and at
    # 8 ranks a rank's share of a frame is ~60 us of kernel); 2B renderer slots: batch n+1 is
    # rendered while batch n is on the wire
    equal_rows = len({sum(h for _, h in b) for b in layout}) == 1
    B = a.gather_batch if a.gather_batch > 0 else (3 if world > 1 else 1)
    if world == 1:
        B = 1
    # leaves only: --min-lod = --max-lod = depth-1 (BASELINE.md "single LOD")
    probe = driver.App(uri, W, H, device=local_rank)
    depth = probe.volume_info()["depth"]
    probe.close()
    # one application (one atlas, one pair of caches) per rank; its row bands are rendered by
    # ONE kernel launch per frame; K renderer instances give K frames in flight
    ray_lod_on = a.ray_lod_sse > 0.0
    if ray_lod_on:
        app = driver.App(uri, W, H, device=local_rank, synchronous=True, sse=a.ray_lod_sse, gpu_cache_mb=3072)
        app.set_ray_lod(True)
    else:
        app = driver.App(uri, W, H, device=local_rank, synchronous=True, min_lod=depth - 1,
                         max_lod=depth - 1, gpu_cache_mb=3072)
    if world > 1:
        app.set_bands(bands)
    app.set_camera(spin=tuple(a.spin))
    app.set_colormap(linear_ramp(a.alpha))

    # ---- who moves the tiles -------------------------------------------------------------------------
    # abi: vrc_gather_tiles behind the C ABI (RCCL sends/receives straight into the frame); checked once on a
    # known pattern before it is trusted.  torch: torch.distributed.gather + a placement copy (round 1).
    gather_kind, gather_note = ("none", None) if world == 1 else (a.gather, None)
    bgather = None
    if world > 1 and gather_kind == "abi":
        # a failing ncclCommInitRank should say why on stderr (RCCL prints its warnings there)
        os.environ.setdefault("NCCL_DEBUG", "WARN")
        try:
            uid = torch.zeros(128, dtype=torch.uint8)
            if rank == 0:
                uid = torch.frombuffer(bytearray(driver.comm_unique_id()), dtype=torch.uint8).clone()
            dist.broadcast(uid, src=0)
            app.comm_create(rank, world, bytes(uid.numpy().tobytes()))
            bgather = sortfirst.AbiTileGather(app, layout, W, rank, "cuda", B)
            bgather.send[0, 0].fill_(float(rank + 1))
            torch.cuda.synchronize()
            # The check runs on an explicit stream and its completion event is recorded on THAT stream: torch's
            # default stream has handle 0, which vrc_gather_tiles takes as "the context's own (non-blocking)
            # stream" -- an event on the default stream would not order against it, the watchdog below could
            # never fire and the pattern could be read before the receives land (round-2 advisor finding).
            cstream = torch.cuda.Stream()
            assert cstream.cuda_stream != 0
            with torch.cuda.stream(cstream):
                bgather.gather(0, 1, cstream.cuda_stream)
                done = torch.cuda.Event()
                done.record(cstream)
            # a first exchange that never completes (a link that does not come up) must not look like a hung
            # benchmark: give it a minute, then stop this rank with a message (the launcher stops the others)
            t_wait = time.perf_counter()
            while not done.query():
                if time.perf_counter() - t_wait > 60.0:
                    sys.stderr.write("bench.py rank %d: the first RCCL tile exchange did not complete within 60 s\n" % rank)
                    sys.stderr.flush()
                    os._exit(3)
                time.sleep(0.01)
            ok = True
            if rank == 0:
                with torch.cuda.stream(cstream):  # same stream as the exchange (the event has completed anyway)
                    got = bgather.frames[0, :, 0, 0].cpu().numpy()
                want = np.zeros(H, dtype=np.float32)
                for r_, y0_, h_ in sortfirst.flat_layout(layout):
                    want[y0_:y0_ + h_] = r_ + 1
                ok = bool((got == want).all()) and bool((bgather.frames[0] == bgather.frames[0, :, :1, :1]).all())
            err = None if ok else "pattern check failed"
        except Exception as e:  # noqa: BLE001
            ok, err = False, repr(e)
        if not all_ok(ok):
            if a.require_abi_gather:
                sys.stderr.write("bench.py rank %d: --require-abi-gather and the C-ABI tile gather is unavailable (%s)\n"
                                 % (rank, err))
                sys.stderr.flush()
                os._exit(4)
            # loud, reported (top-level "abi_gather_ok": false), and only for the plumbing around the kernels:
            # the round-1 gather takes over
            sys.stderr.write("bench.py rank %d: C-ABI tile gather unavailable (%s); using torch.distributed\n" % (rank, err))
            gather_kind, gather_note, bgather = "torch", "C-ABI gather unavailable on at least one rank (%s)" % err, None
    if world > 1 and gather_kind == "torch":
        data_group = make_data_group()
        if not equal_rows:
            B = 1
        if B > 1:
            bgather = sortfirst.BatchedTileGather(layout, W, rank, "cuda", B, group=data_group)
    batched = bgather is not None
    if batched:
        K = 2 * B
    a.warmup = max(a.warmup, K)
    # per in-flight frame: a pixel buffer of this rank's stacked bands (device memory owned by
    # torch), a stream and a tile-gather buffer set
    if batched:
        fbs = [bgather.send[k // B, k % B] for k in range(K)]
    else:
        fbs = [torch.zeros((rows, W, 4), dtype=torch.float32, device="cuda") for _ in range(K)]
    streams = [torch.cuda.Stream() for _ in range(K)]
    gstream = torch.cuda.Stream() if batched else None
    rendered = [torch.cuda.Event() for _ in range(K)] if batched else None
    consumed = [None, None]  # per half: event after the gather that last read it
    app.set_frames_in_flight(K)
    for k in range(K):
        app.select_slot(k)
        app.set_stream(streams[k].cuda_stream)
        app.set_framebuffer(fbs[k].data_ptr())
    gathers = ([sortfirst.TileGather(layout, W, rank, "cuda", group=data_group) for _ in range(K)]
               if (world > 1 and not batched) else None)
    counter = [0]
    last_frame = [None]  # rank 0: the most recently assembled frame (--check-frames)
    gather_events = []   # (start, end) event pairs around every tile exchange while timing[0] is set
    timing = [False]

    def flush(half, n):
        # one gather + one assembly for the n frames rendered into this half
        with torch.cuda.stream(gstream):
            for i in range(n):
                gstream.wait_event(rendered[half * B + i])
            if timing[0]:
                g0 = torch.cuda.Event(enable_timing=True)
                g0.record(gstream)
            if gather_kind == "abi":
                bgather.gather(half, n, gstream.cuda_stream)
            else:
                bgather.gather(half, n)
            if rank == 0:
                last_frame[0] = bgather.assemble(n)[n - 1]
            if timing[0]:
                g1 = torch.cuda.Event(enable_timing=True)
                g1.record(gstream)
                gather_events.append((g0, g1, n))
            ev = torch.cuda.Event()
            ev.record(gstream)
            consumed[half] = ev

    def frame():
        if batched:
            c = counter[0]
            counter[0] += 1
            i, half = c % B, (c // B) % 2
            k = half * B + i
            with torch.cuda.stream(streams[k]):
                if consumed[half] is not None:
                    streams[k].wait_event(consumed[half])  # the gather two batches ago read this buffer
                app.select_slot(k)
                app.render_frame(readback=False)
                rendered[k].record(streams[k])
            if i == B - 1:
                flush(half, B)
            return
        k = counter[0] % K
        counter[0] += 1
        with torch.cuda.stream(streams[k]):
            app.select_slot(k)
            app.render_frame(readback=False)
            if gathers is not None:  # sort-first assembly: tiles to rank 0 over RCCL/xGMI
                if timing[0]:
                    g0 = torch.cuda.Event(enable_timing=True)
                    g0.record(streams[k])
                gathers[k].gather(fbs[k])
                if rank == 0:
                    last_frame[0] = gathers[k].assemble()
                if timing[0]:
                    g1 = torch.cuda.Event(enable_timing=True)
                    g1.record(streams[k])
                    gather_events.append((g0, g1, 1))

    def drain():
        # a partial batch at the end of a run of frames
        if batched and counter[0] % B:
            n = counter[0] % B
            flush((counter[0] // B) % 2, n)
            counter[0] += B - n  # the next frame starts a new batch

    def all_slots(fn):
        out = []
        for k in range(K):
            app.select_slot(k)
            out.append(fn())
        return out

    # first frame: uploads every brick through the 2-thread upload path (not timed below)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    frame()
    drain()
    torch.cuda.synchronize()
    first_frame_ms = (time.perf_counter() - t0) * 1e3

    # samples per frame (deterministic for a fixed view): one counted frame, outside the timing
    app.select_slot(0)
    app.set_option(vrc.OPT_COUNT_SAMPLES, 1)
    with torch.cuda.stream(streams[0]):
        app.render_frame(readback=False)
    samples = app.stats().samples
    app.set_option(vrc.OPT_COUNT_SAMPLES, 0)
    samples_frame = all_sum(int(samples))

    # the GPU clocks up over the first ~20 ms of load: a short untimed run-in before the W warm-up steps,
    # so that a small --warmup / --steps pair measures the same machine state as the default one
    for _ in range(max(0, 40 - a.warmup)):
        frame()
    for _ in range(a.warmup):
        frame()
    drain()
    torch.cuda.synchronize()
    all_slots(app.stats)  # reset the kernel-time accumulators
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    timing[0] = True
    t0 = time.perf_counter()
    for _ in range(a.steps):
        frame()
    drain()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = all_max(time.perf_counter() - t0)
    timing[0] = False

    # HIP-event kernel time of the timed region (events recorded by the library around every raycast launch on
    # the render stream): mean over the launches, which must be the timed frames, one launch each
    ksum, klaunch = 0.0, 0
    for s_ in all_slots(app.stats):
        ksum += s_.kernel_ms_sum
        klaunch += s_.kernel_launches
    assert klaunch == a.steps, "kernel launches in the timed region: %d, frames: %d" % (klaunch, a.steps)
    kernel_ms_per_frame = all_max(ksum / klaunch)  # slowest rank's kernel per frame
    launched_kernel = (vrc.load_library().vrc_last_kernel() or b"").decode()  # the instance the timed frames ran

    # ---- per rank, so that a scaling line can tell throughput from latency (N > 1) ----------------------
    # kernel_ms: this rank's mean raycast kernel per frame (HIP events of the library); gather_ms: device time
    # between the start and the end of the tile exchange (+ assembly on rank 0) per frame, on the stream that
    # carries it; single_frame_latency_ms: wall time of ONE frame with nothing else in flight -- render of this
    # rank's bands, exchange, assembly, synchronised -- the figure an interactive viewer feels, which the
    # frames-in-flight throughput above hides
    my_gather_ms = None
    if gather_events:
        my_gather_ms = sum(g0.elapsed_time(g1) for g0, g1, _ in gather_events) / max(1, sum(n for _, _, n in gather_events))
    lat = []
    if world > 1:
        for _ in range(8):
            dist.barrier()
            torch.cuda.synchronize()
            tl = time.perf_counter()
            frame()
            drain()
            torch.cuda.synchronize()
            lat.append((time.perf_counter() - tl) * 1e3)
        all_slots(app.stats)
    mine = {"rank": rank, "kernel_ms": ksum / klaunch, "gather_ms": my_gather_ms,
            "single_frame_latency_ms": (sorted(lat)[len(lat) // 2] if lat else None), "rows": rows}
    per_rank = [mine]
    if world > 1:
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)

    # the frame assembled from the ranks' bands is the frame one application renders (outside the timing)
    frame_check = None
    if world > 1 and rank == 0 and last_frame[0] is not None:
        with driver.App(uri, W, H, device=local_rank, synchronous=True, gpu_cache_mb=3072,
                        **(dict(sse=a.ray_lod_sse) if ray_lod_on else dict(min_lod=depth - 1, max_lod=depth - 1))) as whole:
            whole.set_ray_lod(ray_lod_on)
            whole.set_camera(spin=tuple(a.spin))
            whole.set_colormap(linear_ramp(a.alpha))
            want, _ = whole.render_frame()
        got = last_frame[0].cpu().numpy()
        frame_check = {"max_abs_diff": float(np.abs(got - want).max()), "bit_identical": bool((got == want).all()),
                       "alpha_max": float(got[..., 3].max())}

    def time_kernel(the_app, n_warm, n_timed):
        # mean HIP-event kernel time per frame over n_timed frames of the_app on its current stream
        for _ in range(n_warm):
            the_app.render_frame(readback=False)
        torch.cuda.synchronize()
        the_app.stats()
        for _ in range(n_timed):
            the_app.render_frame(readback=False)
        torch.cuda.synchronize()
        st_ = the_app.stats()
        return st_.kernel_ms_sum / max(1, st_.kernel_launches)

    def count_samples(the_app):
        the_app.set_option(vrc.OPT_COUNT_SAMPLES, 1)
        the_app.render_frame(readback=False)
        n = int(the_app.stats().samples)
        the_app.set_option(vrc.OPT_COUNT_SAMPLES, 0)
        return n

    def trilinear_forms(the_app, n_warm=5, n_timed=20):
        # the trilinear filter (north star; extension: the reference's sampler is point-sampled) in the form
        # VRC_KERNEL_AUTO takes -- the pool's tap-packed atlas, two 4-byte gathers per sample -- and in the LDS-staged
        # form it took until round 3 (VRC_OPT_PACKED_ATLAS = 0); the same frame, bit for bit
        the_app.set_option(vrc.OPT_FILTER, vrc.FILTER_TRILINEAR)
        res = {}
        try:
            n = count_samples(the_app)
            for key, packed in (("auto", 1), ("lds_staged", 0)):
                the_app.set_option(vrc.OPT_PACKED_ATLAS, packed)
                ms = time_kernel(the_app, n_warm, n_timed)
                res[key] = {"kernel": (vrc.load_library().vrc_last_kernel() or b"").decode(), "kernel_ms_per_frame": ms,
                            "Msamples_per_s": n / ms / 1e3}
            res["samples_per_frame"] = n
        finally:
            the_app.set_option(vrc.OPT_PACKED_ATLAS, 1)
            the_app.set_option(vrc.OPT_FILTER, vrc.FILTER_NEAREST)
        return res

    def extra_trilinear():
        # extension, outside the judged number: the trilinear filter on the same workload (mem:// bricks),
        # kernel time from the library's HIP events
        app.select_slot(0)
        with torch.cuda.stream(streams[0]):
            res = trilinear_forms(app)
        auto = res["auto"]
        alg = a.voxels ** 3 + W * H * 16 + (a.voxels // a.block) ** 3 * 48 + 4096
        return {"kernel": auto["kernel"] + " (tap-packed atlas: 2.25 B per voxel next to the byte atlas, two 4-byte gathers per sample)",
                "kernel_ms_per_frame": auto["kernel_ms_per_frame"], "samples_per_frame": res["samples_per_frame"],
                "Msamples_per_s": auto["Msamples_per_s"],
                "roofline_frac": alg / (auto["kernel_ms_per_frame"] * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                "lds_staged_form": res["lds_staged"],
                "note": "roofline_frac = the judged workload's algorithmic bytes / kernel time / HBM peak; what the "
                        "kernel really moves is the packed atlas (DESIGN.md section 4.2: 5.4 GB of 128-byte line fills "
                        "per frame; arithmetic, fills and L1 look-ups are level)"}

    def extra_moving_camera():
        # outside the judged number too: frame rate while the camera moves (every frame re-derives the
        # frustum, the visible set, the brick order and the tile schedule; nothing is reusable)
        moving = None
        if True:
            app.select_slot(0)
            n_orbit = 100
            with torch.cuda.stream(streams[0]):
                for i in range(10):
                    app.set_camera(spin=(a.spin[0] + 0.002 * i, a.spin[1]))
                    app.render_frame(readback=False)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i in range(n_orbit):
                    app.set_camera(spin=(a.spin[0] + 0.002 * (10 + i), a.spin[1] + 0.001 * i))
                    app.render_frame(readback=False)
                torch.cuda.synchronize()
                moving = {"frames_per_s": n_orbit / (time.perf_counter() - t0), "frames": n_orbit,
                          "note": "camera orbits 0.002 rad per frame: tile schedule, brick order and "
                                  "visible set re-derived every frame"}
            app.set_camera(spin=tuple(a.spin))
        return moving

    def extra_pipelined_and_readback():
        # SURVEY 8(d) defines fps with the frame read back to pinned host memory: three frames in
        # flight, each on its own stream: kernel, then the 16 MiB device -> pinned copy behind it
        readback = None
        pipelined = None
        if True:
            KR = 3
            r_streams = [torch.cuda.Stream() for _ in range(KR)]
            r_fbs = [torch.zeros((rows, W, 4), dtype=torch.float32, device="cuda") for _ in range(KR)]
            r_host = [torch.zeros((rows, W, 4), dtype=torch.float32).pin_memory() for _ in range(KR)]
            app.set_frames_in_flight(KR)
            for k in range(KR):
                app.select_slot(k)
                app.set_stream(r_streams[k].cuda_stream)
                app.set_framebuffer(r_fbs[k].data_ptr())

            def rb_frame(i):
                k = i % KR
                with torch.cuda.stream(r_streams[k]):
                    app.select_slot(k)
                    app.render_frame(readback=False)
                    r_host[k].copy_(r_fbs[k], non_blocking=True)

            # the same three frames in flight without the read-back: consecutive frames' kernels overlap
            # and fill each other's tails.  Not the judged number: with overlapping launches the
            # HIP-event duration of a kernel is no longer its own, so the timed region above keeps one
            # frame in flight at N=1 and a clean roofline.
            def pl_frame(i):
                k = i % KR
                with torch.cuda.stream(r_streams[k]):
                    app.select_slot(k)
                    app.render_frame(readback=False)

            for i in range(2 * KR):
                pl_frame(i)
            torch.cuda.synchronize()
            n_pl = 200
            t0 = time.perf_counter()
            for i in range(n_pl):
                pl_frame(i)
            torch.cuda.synchronize()
            dt_pl = time.perf_counter() - t0
            pipelined = {"frames_per_s": n_pl / dt_pl, "Msamples_per_s": samples_frame * n_pl / dt_pl / 1e6,
                         "frames": n_pl, "frames_in_flight": KR,
                         "note": "three renderers on three streams over one atlas; kernels of consecutive "
                                 "frames overlap"}

            # read-back leg: six frames in flight (measured, tools/dev_readback.py: the 16 MiB copy behind each
            # kernel hides completely from four to six on: 1 / 2 / 3 / 4 / 6 in flight = 1231 / 1340 / 1623 / 2108 / 2154
            # frames/s)
            KR = 6
            while len(r_streams) < KR:
                r_streams.append(torch.cuda.Stream())
                r_fbs.append(torch.zeros((rows, W, 4), dtype=torch.float32, device="cuda"))
                r_host.append(torch.zeros((rows, W, 4), dtype=torch.float32).pin_memory())
            torch.cuda.synchronize()
            app.set_frames_in_flight(KR)
            for k in range(KR):
                app.select_slot(k)
                app.set_stream(r_streams[k].cuda_stream)
                app.set_framebuffer(r_fbs[k].data_ptr())
            for i in range(2 * KR):
                rb_frame(i)
            torch.cuda.synchronize()
            n_rb = 240
            t0 = time.perf_counter()
            for i in range(n_rb):
                rb_frame(i)
            torch.cuda.synchronize()
            dt_rb = time.perf_counter() - t0
            readback = {"frames_per_s": n_rb / dt_rb, "frames": n_rb, "frames_in_flight": KR,
                        "bytes_per_frame": rows * W * 16,
                        "d2h_GBps": n_rb * rows * W * 16 / dt_rb / 1e9,
                        "note": "kernel + RGBA32F frame copied to pinned host memory, 6 frames in flight"}
            ok = bool(torch.isfinite(r_host[0]).all()) and float(r_host[0][..., 3].max()) > 0.0
            readback["frame_ok"] = ok
        return (pipelined, readback)

    def extra_per_ray_lod():
        # BASELINE C5's kernel side on the same volume (extension): the SelectVisibles cut at sse 2 rendered
        # per brick (the reference's way: fewer bricks, finest step) and with per-ray LOD (the rule applied
        # along the ray over the cut and its ancestors, step and opacity scaled with the level)
        res = {"sse": 2.0, "note": "same camera and volume as the judged workload, LOD tree enabled; "
                                   "kernel time from HIP events, samples from the kernel's counter"}
        with driver.App(uri, W, H, synchronous=True, sse=2.0, gpu_cache_mb=3072, device=local_rank) as lod_app:
            lod_app.set_colormap(linear_ramp(a.alpha))
            lod_app.set_camera(spin=tuple(a.spin))
            for key, on in (("per_brick_cut", False), ("per_ray_lod", True)):
                lod_app.set_ray_lod(on)
                lod_app.set_option(vrc.OPT_COUNT_SAMPLES, 1)
                _, st = lod_app.render_frame(readback=False)
                n = int(lod_app.stats().samples)
                lod_app.set_option(vrc.OPT_COUNT_SAMPLES, 0)
                for _ in range(3):
                    lod_app.render_frame(readback=False)
                lod_app.stats()
                for _ in range(20):
                    lod_app.render_frame(readback=False)
                torch.cuda.synchronize()
                s2 = lod_app.stats()
                ms = s2.kernel_ms_sum / max(1, s2.kernel_launches)
                res[key] = {"kernel_ms": ms, "samples_per_frame": n, "bricks": int(st.n_available),
                            "Msamples_per_s": n / ms / 1e3}
        return res

    def extra_volume_n():
        # SURVEY 8(d) "Volume N" (bandwidth realism): mem:// bricks are constant, so every lane of a wave reads
        # the same classified-table entry; the seeded-noise volume of the same size, camera and transfer
        # function does not have that luck.  Same kernels, kernel time from the library's HIP events.  And the second
        # view BASELINE.md names for C2: the model spun by 30 / 20 degrees (livre/eq/settings/CameraSettings.cpp:35-59),
        # where the rays cross the brick grid and the micro-blocks at an angle.  Point-sampled (the reference's
        # sampler) and with the trilinear filter (extension).
        n_uri = "hash://#%d,%d,%d,%d" % (a.voxels, a.voxels, a.voxels, a.block)
        alg = a.voxels ** 3 + W * H * 16 + (a.voxels // a.block) ** 3 * 48 + 4096
        views = {}
        with driver.App(n_uri, W, H, device=local_rank, synchronous=True, min_lod=depth - 1, max_lod=depth - 1,
                        gpu_cache_mb=3072) as napp:
            napp.set_colormap(linear_ramp(a.alpha))
            for key, spin in (("default_camera", tuple(a.spin)), ("off_axis", (0.5235988, 0.3490659))):
                napp.set_camera(spin=spin)
                n_samples = count_samples(napp)
                ms = time_kernel(napp, 10, 40)
                point_kernel = (vrc.load_library().vrc_last_kernel() or b"").decode()
                tri = trilinear_forms(napp, 3, 12)
                views[key] = {"spin_rad": list(spin), "kernel": point_kernel,
                              "kernel_ms_per_frame": ms, "samples_per_frame": n_samples,
                              "Msamples_per_s": n_samples / ms / 1e3, "frames_per_s_kernel": 1e3 / ms,
                              "roofline_frac": alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                              "trilinear": {"kernel": tri["auto"]["kernel"],
                                            "kernel_ms_per_frame": tri["auto"]["kernel_ms_per_frame"],
                                            "samples_per_frame": tri["samples_per_frame"],
                                            "Msamples_per_s": tri["auto"]["Msamples_per_s"],
                                            "roofline_frac": alg / (tri["auto"]["kernel_ms_per_frame"] * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                                            "lds_staged_form": tri["lds_staged"]}}
        d = views["default_camera"]
        out = {"volume": n_uri + " (v = lowbias32(x + X*(y + Y*z) + 0x5EED) >> 24, 3-tap box filter per axis)",
               "kernel_ms_per_frame": d["kernel_ms_per_frame"], "samples_per_frame": d["samples_per_frame"],
               "Msamples_per_s": d["Msamples_per_s"], "frames_per_s_kernel": d["frames_per_s_kernel"],
               "roofline_frac": d["roofline_frac"],
               "vs_mem_volume_kernel_ms": d["kernel_ms_per_frame"] / kernel_ms_per_frame,
               "trilinear": d["trilinear"],
               "off_axis": dict(views["off_axis"], note="the model spun by 30 / 20 degrees (BASELINE.md's second C2 view)")}
        return out

    # the extras can never cost the judged line: a failure is reported in their place
    def guarded(fn, default):
        if world != 1 or a.no_extras or ray_lod_on:
            return default
        try:
            return fn()
        except Exception as e:  # noqa: BLE001
            sys.stderr.write("bench.py: extra measurement %s failed: %r\n" % (fn.__name__, e))
            return {"error": repr(e)} if not isinstance(default, tuple) else tuple({"error": repr(e)} for _ in default)

    trilinear = guarded(extra_trilinear, None)
    moving = guarded(extra_moving_camera, None)
    pipelined, readback = guarded(extra_pipelined_and_readback, (None, None))
    ray_lod = guarded(extra_per_ray_lod, None)
    volume_n = guarded(extra_volume_n, None)

    if rank == 0:
        # measured HBM traffic per launch, from the committed rocprofv3 PMC passes (bench.py cannot
        # run the profiler on itself); only quoted for the workload it was measured on
        traffic, valu, tnote = None, None, "no committed PMC passes for this workload"
        tpath = os.path.join(ROOT, "profiles", "r4_traffic_c2.json")
        if (world == 1 and a.voxels == 1024 and a.block == 128 and a.viewport == 1024
                and tuple(a.spin) == (0.0, 0.0) and not ray_lod_on and os.path.exists(tpath)):
            tj = json.load(open(tpath))
            # the counters are only quoted for the kernel instance they were collected on
            if launched_kernel and tj["kernel"].startswith(launched_kernel + " "):
                traffic = tj["traffic_bytes_per_launch"]
                tnote = tj.get("traffic_note", "profiles/r4_traffic_c2.json (same kernel instance)")
                if "sq_insts_valu_per_launch" in tj:
                    # a VALU figure next to the HBM fraction.  One wave-instruction holds its SIMD's
                    # issue for 4 cycles (2 for the dual-issue classes, profiles/r3_ubench_valu_lds_issue_costs.txt:
                    # the floor below prices every instruction at its measured class cost); 1024 SIMDs
                    steps = tj["wave_steps_per_launch"]
                    floor_ms = tj["valu_issue_ns_per_simd_per_launch"] * 1e-6
                    valu = {"insts_per_step": tj["sq_insts_valu_per_launch"] / steps,
                            "insts_per_launch": tj["sq_insts_valu_per_launch"], "wave_steps_per_launch": steps,
                            "issue_floor_ms": floor_ms, "frac": floor_ms / kernel_ms_per_frame,
                            "source": tj.get("valu_source")}
            else:
                tnote = ("profiles/r4_traffic_c2.json is a profile of %s, this run launched %s: not quoted"
                         % (tj["kernel"].split(" (")[0], launched_kernel or "?"))
        n_nodes = (a.voxels // a.block) ** 3
        # interior voxels of marched bricks + one RGBA32F write + node table + TF, each input
        # voxel once per frame.  With N ranks a rank's row bands sweep rows/H of the volume (the bands are
        # interleaved, every brick is crossed by rays of every rank, each voxel belongs to the rays of ONE
        # row): its launch reads voxels^3 * rows/H + its own part of the frame, the table and the TF.
        alg_bytes = a.voxels ** 3 + world * (n_nodes * 48 + 4096) + W * H * 16
        per_rank_alg = a.voxels ** 3 * rows // H + n_nodes * 48 + 4096 + rows * W * 16
        achieved = per_rank_alg / (kernel_ms_per_frame * 1e-3) / 1e9
        out = {
            "metric": "Msamples/s + frames/s, 1024^3 volume @ 1024^2 viewport",
            "value": samples_frame * a.steps / dt / 1e6,
            "unit": "Msamples/s",
            "frames_per_s": a.steps / dt,
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic (mem:// rule of datasources/memory/MemoryDataSource.cpp:54-57)",
            # N > 1: did the tiles move through the C ABI (vrc_gather_tiles over RCCL)?  false = this line was
            # measured on the torch.distributed fallback and is NOT a measurement of the C-ABI exchange
            "abi_gather_ok": (None if world == 1 else gather_kind == "abi"),
            "frames_in_flight": K, "frames_per_exchange": B,
            "single_frame_latency_ms": (max(r["single_frame_latency_ms"] for r in per_rank) if world > 1 else
                                        dt / a.steps * 1e3),
            "per_rank": per_rank,
            "config": {"workload": ("C2: %s uint8, %dx%d viewport, leaves only (%d bricks of %d^3), "
                                    "%d samples/ray, linear-ramp TF alpha=%.3g, default camera"
                                    % (uri, W, H, n_nodes, a.block + 8,
                                       app.stats().samples_per_ray, a.alpha)) if not ray_lod_on else
                                   ("NOT the judged workload -- C5 kernel side: %s uint8, %dx%d viewport, whole LOD "
                                    "tree, per-ray adaptive LOD at screen-space error %g, linear-ramp TF alpha=%.3g"
                                    % (uri, W, H, a.ray_lod_sse, a.alpha)),
                       "parallelism": "sort-first, %d rank(s) x %d interleaved row band(s) in one "
                                      "launch, RGBA32F tiles to rank 0 by %s (%d frame(s) per exchange), %d frames in flight"
                                      % (world, len(bands),
                                         {"abi": "vrc_gather_tiles (RCCL behind the C ABI, straight into the frame)",
                                          "torch": "torch.distributed.gather + placement copy",
                                          "none": "nothing (one rank)"}[gather_kind], B, K),
                       "tile_gather": {"abi": "abi", "torch": "torch-fallback" if a.gather == "abi" else "torch", "none": "none"}[gather_kind],
                       "tile_gather_note": gather_note,
                       "world_size_checked": world,
                       "sort_first_frame_check": frame_check,
                       "samples_per_frame": samples_frame,
                       "per_frame_work": "every timed frame is a full render_frame call and a full march (frame "
                                         "cleared, every ray and sample computed, frame written); the camera stands "
                                         "still, so the plugin keeps its brick list, node table and tile schedule "
                                         "between identical frames instead of deriving them again (host side only; "
                                         "moving_camera below is the same path with nothing to keep)",
                       "first_frame_with_upload_ms": first_frame_ms,
                       "extension_trilinear": trilinear, "moving_camera": moving,
                       "with_readback_to_pinned_host": readback,
                       "three_frames_in_flight": pipelined,
                       "extension_per_ray_lod": ray_lod,
                       "volume_n": volume_n},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_note": tnote,
                         "valu": valu,
                         # what the library says it launched (vrc_last_kernel): MODE 3 = the two-float table form of a
                         # grey transfer function (the linear ramp is one), bit-identical frames (VRC_OPT_GREY_TABLE)
                         "kernel": launched_kernel,
                         "kernel_ms_per_frame": kernel_ms_per_frame,
                         "algorithmic_bytes_per_launch": per_rank_alg,
                         "algorithmic_bytes_per_frame_all_ranks": alg_bytes,
                         "launches_per_frame": 1,
                         "note": "achieved = SURVEY 8(d)'s algorithmic bytes / kernel time.  What the kernel really moves is "
                                 "`traffic`: the L2s fill whole 128-byte lines (TCC_EA0_RDREQ_128B = every read request), "
                                 "and a line that straddles two tiles' footprints is filled once for each of them -- the "
                                 "tiles reach it at different times and an L2 remembers a few steps (DESIGN.md section 4).  "
                                 "The rest of the time is vector issue (valu.frac at the measured class costs; the build "
                                 "without any gather runs in 0.265 ms)"},
            # the judged number rides three best cases, each measured in this line: the default camera looks along a
            # volume axis (config.volume_n.off_axis: the 30/20-degree view), mem:// bricks are constant (config.volume_n:
            # noise data), the linear ramp is a grey transfer function (VRC_OPT_GREY_TABLE: a coloured one costs ~3 %)
            "best_case_riders": ["axis-aligned default camera", "constant mem:// bricks", "grey transfer function"],
        }
        if world == 1 and not a.no_cpu_baseline and not ray_lod_on:
            try:
                out["cpu_baseline"] = cpu_baseline(a, samples_frame)
            except Exception as e:  # noqa: BLE001  (the GPU line must survive a broken checker build)
                sys.stderr.write("bench.py: cpu_baseline failed: %r\n" % (e,))
                out["cpu_baseline"] = {"value": None, "unit": "Msamples/s", "cores": 0, "kind": "port",
                                       "sample": "failed: %r" % (e,)}
        print(json.dumps(out), file=json_out, flush=True)

    app.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
