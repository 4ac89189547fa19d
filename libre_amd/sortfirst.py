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

    def __init__(self, layout, width, rank, device, dst=0):
        self.layout, self.width, self.rank, self.dst = layout, width, rank, dst
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
            dist.gather(local, self.recv if self.rank == self.dst else None, dst=self.dst)
        elif self.rank == self.dst:
            self.recv[self.dst].copy_(local)
            reqs = [dist.irecv(self.recv[r], src=r) for r in range(self.world) if r != self.dst]
            for q in reqs:
                q.wait()
        else:
            dist.send(local, dst=self.dst)

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
