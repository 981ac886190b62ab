"""Multi-GPU: one process per GPU, world sections sharded by contiguous key range, one exchange
step per frame -- the all-gather of every GPU's packed visible-instance buffer (RCCL over xGMI via
torch.distributed; backend "nccl" is RCCL on ROCm).  Sections are independent for cull and pack, so
there is no data-path collective before the exchange.

The functions work on whatever device the tensors live on, so the N>1 logic is covered by
world_size-2 gloo tests on CPU (tests/test_parallel_gloo.py).
"""
import numpy as np
import torch


def shard_bounds(n_units, world_size):
    """Contiguous, near-equal ranges of the key-sorted unit (section / entity index) space."""
    base, rem = divmod(n_units, world_size)
    out, lo = [], 0
    for r in range(world_size):
        hi = lo + base + (1 if r < rem else 0)
        out.append((lo, hi)); lo = hi
    return out


def allgather_packed(ids, mats, n_local, dist, group=None):
    """Variable-length all-gather of {entity id, 4x4 matrix} slabs.

    ids: int32/uint32-as-int32 tensor [cap]; mats: float32 [cap, 16]; n_local valid rows.
    Returns (ids_all [sum n], mats_all [sum n, 16], counts list) in rank order -- deterministic
    given the partition.  Two collectives: the counts (one int per rank), then slabs padded to the
    largest count (each GPU's slab crosses each xGMI link once in a direct all-gather).
    """
    world = dist.get_world_size(group)
    dev = ids.device
    cnt = torch.tensor([int(n_local)], dtype=torch.int64, device=dev)
    cnts = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(cnts, cnt, group=group)
    counts = [int(c.item()) for c in cnts]
    m = max(counts)
    if m == 0:
        return ids[:0].clone(), mats[:0].clone(), counts
    if m > ids.shape[0]:
        raise ValueError(f"all-gather slab of {m} instances exceeds the local buffer capacity {ids.shape[0]}")
    send_i, send_m = ids[:m].contiguous(), mats[:m].contiguous()
    recv_i = [torch.empty_like(send_i) for _ in range(world)]
    recv_m = [torch.empty_like(send_m) for _ in range(world)]
    dist.all_gather(recv_i, send_i, group=group)
    dist.all_gather(recv_m, send_m, group=group)
    ids_all = torch.cat([recv_i[r][:counts[r]] for r in range(world)])
    mats_all = torch.cat([recv_m[r][:counts[r]] for r in range(world)])
    return ids_all, mats_all, counts


class VisibleAllGather:
    """Binds a Pipeline's packed output to torch tensors (the all-gather send slab) and runs the
    per-frame exchange."""

    def __init__(self, pipeline, capacity, dist):
        self.p, self.dist, self.cap = pipeline, dist, capacity
        dev = torch.device("cuda", torch.cuda.current_device())
        self.ids = torch.zeros(capacity, dtype=torch.int32, device=dev)
        self.mats = torch.zeros(capacity, 16, dtype=torch.float32, device=dev)
        pipeline.set_output_buffers(self.ids.data_ptr(), self.mats.data_ptr(), capacity)
        self.last = None

    def exchange(self, n_written):
        self.last = allgather_packed(self.ids, self.mats, n_written, self.dist)
        torch.cuda.current_stream().synchronize()      # the send slab is rewritten by the next cull
        return self.last


class SlabAllGather:
    """The per-frame exchange without a host round trip: every rank packs its visible instances straight into a
    fixed-size slab  [count | pad to 64 B | ids[cap] | matrices[cap * 16]]  (the count is written by the pack kernel,
    re_set_output_count), and one all_gather_into_tensor of the slabs follows the pack in stream order.  Two slabs
    alternate by frame, so the collective of frame f overlaps the cull of frame f+1; a slab is rewritten only after the
    collective that read it (two frames earlier) has completed -- enforced on the stream, not on the host.
    A rank whose visible set exceeds the slab keeps its full result locally (Pipeline results are unaffected); its slab
    carries the truncated prefix and the header tells the receivers (count == cap means "possibly truncated").
    xGMI is point to point: the direct all-gather moves each rank's slab over each link once, so the slab is sized
    for the expected visible set (a few thousand instances), not for the worst case."""

    HEADER_WORDS = 16

    def __init__(self, pipeline, slab_instances, dist, group=None):
        self.p, self.dist, self.group, self.cap = pipeline, dist, group, int(slab_instances)
        self.world = dist.get_world_size(group)
        dev = torch.device("cuda", torch.cuda.current_device())
        words = self.HEADER_WORDS + self.cap * 17
        self.slab = [torch.zeros(words, dtype=torch.int32, device=dev) for _ in range(2)]
        self.recv = [torch.zeros(words * self.world, dtype=torch.int32, device=dev) for _ in range(2)]
        self.work = [None, None]
        self.frame = 0
        self._lag_pending = False
        self.stream = torch.cuda.ExternalStream(pipeline.stream()) if hasattr(torch.cuda, "ExternalStream") else torch.cuda.current_stream()

    def _bind(self, b):
        base = self.slab[b].data_ptr()
        self.p.set_output_count(base)
        self.p.set_output_buffers(base + 4 * self.HEADER_WORDS, base + 4 * (self.HEADER_WORDS + self.cap), self.cap)

    def begin_frame(self):
        """call before the frame's cull_and_pack: points the pack at this frame's slab"""
        b = self.frame & 1
        if self.work[b] is not None:
            with torch.cuda.stream(self.stream):
                self.work[b].wait()                      # stream-side: the slab's previous collective has finished
            self.work[b] = None
        self._bind(b)
        return b

    def exchange(self):
        """call after the frame's (asynchronous) cull_and_pack: enqueues the all-gather behind it"""
        b = self.frame & 1
        with torch.cuda.stream(self.stream):
            self.work[b] = self.dist.all_gather_into_tensor(self.recv[b], self.slab[b], group=self.group, async_op=True)
        self.frame += 1
        return b

    def _gather(self, b):
        with torch.cuda.stream(self.stream):
            self.work[b] = self.dist.all_gather_into_tensor(self.recv[b], self.slab[b], group=self.group, async_op=True)

    def exchange_lagged(self):
        """for frames issued with defer_pack=True (Pipeline.cull_and_pack): the launch just enqueued carries the pack of the PREVIOUS
        frame, so that frame's slab is complete in stream order now -- enqueue its all-gather.  Call after the frame's asynchronous
        cull_and_pack; returns the buffer gathered (None for the first frame).  Works unchanged when the library did not defer the
        pack (worlds with dynamic entities): the exchange then simply runs one frame late."""
        prev = None
        if self.frame >= 1:
            prev = (self.frame - 1) & 1
            self._gather(prev)
        self.frame += 1
        self._lag_pending = True
        return prev

    def finish_lagged(self):
        """after Pipeline.wait() (which sends a still deferred pack off on its own): the last frame's all-gather, then wait for all"""
        b = None
        if self._lag_pending and self.frame >= 1:
            b = (self.frame - 1) & 1
            self._gather(b)
            self._lag_pending = False
        self.finish()
        return b

    def finish(self):
        for b in (0, 1):
            if self.work[b] is not None:
                self.work[b].wait(); self.work[b] = None
        torch.cuda.synchronize()

    def gathered(self, b):
        """(ids, matrices, counts) of buffer b in rank order, after finish()"""
        words = self.HEADER_WORDS + self.cap * 17
        r = self.recv[b].view(self.world, words)
        counts = [int(c) for c in r[:, 0].tolist()]
        ids = torch.cat([r[k, self.HEADER_WORDS:self.HEADER_WORDS + min(counts[k], self.cap)] for k in range(self.world)])
        mats = torch.cat([r[k, self.HEADER_WORDS + self.cap:].view(torch.float32).view(self.cap, 16)[:min(counts[k], self.cap)] for k in range(self.world)])
        return ids, mats, counts
