"""Sort-first (screen-space) decomposition across the ranks of one node and the per-frame
assembly of the RGBA32F tiles on the display rank -- the role Equalizer's 2-D compound +
eq::Compositor::assembleFrame plays in the reference (livre/eq/Channel.cpp:272-290, 519-523),
here over torch.distributed (backend "nccl" = RCCL over xGMI on the GPU node, "gloo" in the CPU
tests).  No brick data moves between ranks: every rank owns its bricks and renders its tiles
with an off-axis sub-frustum (livre/eq/Channel.cpp:151-157).

Tiles are row bands.  Ray length varies strongly with the image row (rays through the image
centre cross the whole volume, rays near the top/bottom leave through a side face early), so a
frame of `world` equal strips is badly balanced; instead the frame is cut into
world*bands_per_rank bands and rank r takes bands r, r+world, r+2*world, ...
"""
import torch
import torch.distributed as dist


def flat_layout(layout):
    """-> [(rank, y0, h)] in frame order: the band list vrc_gather_tiles takes (include/vrc_hip.h)."""
    return [(r, y0, h) for (y0, h, r) in sorted((y0, h, r) for r, bands in enumerate(layout) for (y0, h) in bands)]


def band_layout(height, world, bands_per_rank):
    """-> per rank: list of (y0, h).  Bands tile [0, height) exactly."""
    nb = world * bands_per_rank if world > 1 else 1
    nb = max(1, min(nb, height))
    edges = [round(height * k / nb) for k in range(nb + 1)]
    out = [[] for _ in range(world)]
    for b in range(nb):
        if edges[b + 1] > edges[b]:
            out[b % world].append((edges[b], edges[b + 1] - edges[b]))
    return out


class TileGather:
    """Per-frame gather of every rank's stacked bands to rank `dst` and their placement into
    the full frame.  Buffers are allocated once."""

    def __init__(self, layout, width, rank, device, dst=0, group=None):
        self.layout, self.width, self.rank, self.dst, self.group = layout, width, rank, dst, group
        self.world = len(layout)
        self.counts = [sum(h for _, h in b) for b in layout]
        self.equal = len(set(self.counts)) == 1
        self.recv = None
        self.frame = None
        # regular interleave (band b = rows [b*h, (b+1)*h) on rank b % world, all bands h rows):
        # the frame is a permutation of the receive buffer, assembled by ONE strided copy
        heights = {h for b in layout for _, h in b}
        self.regular = (self.equal and len(heights) == 1 and all(
            bands == [((k * self.world + r) * h, h) for k in range(len(bands))]
            for r, bands in enumerate(layout) for h in heights))
        if rank == dst:
            if self.equal:
                self.recv_all = torch.empty((self.world, self.counts[0], width, 4), dtype=torch.float32,
                                            device=device)
                self.recv = [self.recv_all[r] for r in range(self.world)]
            else:
                self.recv = [torch.empty((c, width, 4), dtype=torch.float32, device=device)
                             for c in self.counts]
            self.frame = torch.zeros((sum(self.counts), width, 4), dtype=torch.float32, device=device)

    def gather(self, local):
        """local: this rank's bands stacked, (rows, W, 4) float32.  Collective."""
        if self.world == 1:
            if self.rank == self.dst:
                self.recv[0] = local  # no copy: the local frame is the frame
            return
        if self.equal:
            # equal tiles: one RCCL gather; each peer uses its own point-to-point xGMI link
            dist.gather(local, self.recv if self.rank == self.dst else None, dst=self.dst, group=self.group)
        elif self.rank == self.dst:
            self.recv[self.dst].copy_(local)
            reqs = [dist.irecv(self.recv[r], src=r, group=self.group) for r in range(self.world) if r != self.dst]
            for q in reqs:
                q.wait()
        else:
            dist.send(local, dst=self.dst, group=self.group)

    def assemble(self):
        """Display rank only: place the received bands at their rows (the frame assembly)."""
        assert self.rank == self.dst
        if self.regular and self.world > 1:
            h = self.layout[0][0][1]
            bpr = len(self.layout[0])
            src = self.recv_all.view(self.world, bpr, h, self.width, 4).permute(1, 0, 2, 3, 4)
            self.frame.view(bpr, self.world, h, self.width, 4).copy_(src)
            return self.frame
        for r, bands in enumerate(self.layout):
            off = 0
            for (y0, h) in bands:
                self.frame[y0:y0 + h].copy_(self.recv[r][off:off + h])
                off += h
        return self.frame


class BatchedTileGather:
    """The sort-first assembly of `batch` consecutive frames with ONE collective: every rank renders
    frame i of a batch into send[half, i]; after the batch one gather moves all of them to the
    display rank and one strided copy places them into `batch` full frames.  A collective costs
    its issue time on the host whatever it carries; at eight ranks a rank's share of a frame is
    ~60 us of kernel, less than that issue time, so the frame rate of the node is the rate at
    which collectives can be issued unless frames share one.  Two halves: batch n+1 is rendered
    while batch n is on the wire.  Equal row counts per rank (band_layout gives them whenever the
    frame height divides evenly) are required; otherwise use TileGather per frame."""

    def __init__(self, layout, width, rank, device, batch, dst=0, group=None):
        self.layout, self.width, self.rank, self.dst, self.batch, self.group = layout, width, rank, dst, batch, group
        self.world = len(layout)
        counts = [sum(h for _, h in b) for b in layout]
        if len(set(counts)) != 1:
            raise ValueError("BatchedTileGather needs the same number of rows on every rank")
        self.rows = counts[0]
        self.height = sum(counts)
        heights = {h for b in layout for _, h in b}
        self.regular = (len(heights) == 1 and all(
            bands == [((k * self.world + r) * h, h) for k in range(len(bands))]
            for r, bands in enumerate(layout) for h in heights))
        self.send = torch.zeros((2, batch, self.rows, width, 4), dtype=torch.float32, device=device)
        self.recv_all = None
        self.frames = None
        if rank == dst:
            self.recv_all = torch.empty((self.world, batch, self.rows, width, 4), dtype=torch.float32, device=device)
            self.frames = torch.zeros((batch, self.height, width, 4), dtype=torch.float32, device=device)

    def gather(self, half, n):
        """Collective: the first n frames of send[half] of every rank -> recv_all[:, :n] on the display rank."""
        local = self.send[half, :n]
        if self.world == 1:
            if self.rank == self.dst:
                self.recv_all[0, :n].copy_(local)
            return
        recv = [self.recv_all[r, :n] for r in range(self.world)] if self.rank == self.dst else None
        dist.gather(local, recv, dst=self.dst, group=self.group)

    def assemble(self, n):
        """Display rank only: the n gathered frames, each band at its rows.  -> frames[:n]"""
        assert self.rank == self.dst
        if self.regular:
            h = self.layout[0][0][1]
            bpr = len(self.layout[0])
            src = self.recv_all[:, :n].reshape(self.world, n, bpr, h, self.width, 4).permute(1, 2, 0, 3, 4, 5)
            self.frames[:n].view(n, bpr, self.world, h, self.width, 4).copy_(src)
            return self.frames[:n]
        for r, bands in enumerate(self.layout):
            off = 0
            for (y0, h) in bands:
                self.frames[:n, y0:y0 + h].copy_(self.recv_all[r, :n, off:off + h])
                off += h
        return self.frames[:n]



def check_one_hip_runtime():
    """PyTorch ships its own libamdhip64 / librccl (same sonames as ROCm's).  Imported FIRST, they satisfy
    libvrc_hip.so's dependencies and the process has one HIP runtime and one RCCL, shared by both; the other way round
    the process ends up with two runtimes, and a stream or an event of one is garbage to the other.  Raises if that
    has happened (Linux: /proc/self/maps)."""
    try:
        maps = open("/proc/self/maps").read()
    except OSError:
        return
    import re
    found = sorted(set(re.findall(r"(/\S*libamdhip64\S*)", maps)))
    if len(found) > 1:
        raise RuntimeError("two HIP runtimes in one process (%s): import torch before loading libvrc_hip.so / "
                           "libLivreHipRaycastPipeline.so" % ", ".join(found))


class AbiTileGather:
    """The sort-first assembly through the C ABI: vrc_gather_tiles (include/vrc_hip.h) -- RCCL sends and
    receives, one per band, that land every band directly at its rows of the frame on the display rank; no
    staging buffer and no placement copy.  PyTorch only owns the memory.  `batch` consecutive frames share
    one group of sends / receives (two halves: batch n+1 is rendered while batch n is on the wire); ranks
    may have different numbers of rows.  `app` is a libre_amd.driver.App whose communicator exists
    (app.comm_create)."""

    def __init__(self, app, layout, width, rank, device, batch, dst=0):
        check_one_hip_runtime()
        self.app, self.layout, self.width, self.rank, self.dst, self.batch = app, layout, width, rank, dst, batch
        self.world = len(layout)
        self.rows = sum(h for _, h in layout[rank])
        self.height = sum(h for b in layout for _, h in b)
        self.send = torch.zeros((2, batch, max(1, self.rows), width, 4), dtype=torch.float32, device=device)
        self.frames = None
        if rank == dst:
            self.frames = torch.zeros((batch, self.height, width, 4), dtype=torch.float32, device=device)
        app.set_layout(layout)

    def gather(self, half, n, stream=None):
        """Collective, asynchronous on `stream` (a raw hipStream_t handle; None: the renderer's stream): the first
        n frames of send[half] of every rank -> frames[:n] on the display rank."""
        self.app.gather_tiles(n, self.send[half].data_ptr(), self.send[half, 0].numel() * 4,
                              self.frames.data_ptr() if self.frames is not None else None,
                              self.height * self.width * 16, self.dst, stream)

    def assemble(self, n):
        """Display rank only: the n gathered frames (already in place)."""
        assert self.rank == self.dst
        return self.frames[:n]
