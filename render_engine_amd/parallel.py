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


def key_ranges_from_cuts(cut_keys):
    """[(lo, hi)] per rank from the ascending first keys of ranks 1..N-1: rank r owns the world sections with cut[r-1] <= key < cut[r]"""
    edges = [0] + [int(k) for k in cut_keys] + [1 << 63]
    return [(edges[r], edges[r + 1]) for r in range(len(edges) - 1)]


def route_migrants(states, ranges, tree_outline_length=16384, tree_atomic_length=64):
    """The owner of every migrant record (ENTITY_DT array): the rank whose key range holds the smallest key of the entity's world section(s) --
    host arithmetic (re_section_keys), the same on every rank.  Returns one index array per rank; entities out of bounds (no section) go nowhere."""
    import numpy as np
    from .pipeline import first_section_keys
    if not len(states):
        return [np.zeros(0, np.int64) for _ in ranges]
    first = first_section_keys(states, tree_outline_length, tree_atomic_length)
    out = []
    for lo, hi in ranges:
        out.append(np.nonzero((first != 0) & (first >= np.uint64(lo)) & (first < np.uint64(hi)))[0])
    return out


def exchange_migrants(pipeline, dist, ranges, group=None, tree_outline_length=16384, tree_atomic_length=64):
    """The frame's second, sparse exchange: every rank hands over the entities whose section left its key range (take_migrants: list,
    export, remove), the records travel over the host channel (a few per frame: all_gather_object), and every rank registers the ones whose new
    section it owns (register_model_instances appends).  Call between frames, after the tick.  Returns (sent, received)."""
    import numpy as np
    from .pipeline import ENTITY_DT
    mine = pipeline.take_migrants()
    world = dist.get_world_size(group)
    box = [None] * world
    dist.all_gather_object(box, mine.tobytes(), group=group)
    rank = dist.get_rank(group)
    got = []
    for r, blob in enumerate(box):
        if r == rank or not blob:
            continue
        st = np.frombuffer(blob, ENTITY_DT)
        sel = route_migrants(st, ranges, tree_outline_length, tree_atomic_length)[rank]
        if len(sel):
            got.append(st[sel])
    if got:
        new = np.concatenate(got)
        pipeline.register_model_instances(new)
        return len(mine), len(new)
    return len(mine), 0


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


HEADER_WORDS = 16                 # slab = [4-word header written by the pack kernel | pad to 16 words | ids[cap] | matrices[cap * 16]]
SLAB_CANCELLED = -1               # header word 0 of a frame that cross-frame speculation cancelled (0xFFFFFFFF as int32)


class SlabCancelled(RuntimeError):
    """a rank's frame was cancelled by cross-frame speculation when its slab went out: settle (Pipeline.wait replays it into the same slab) and gather again"""


class SlabOverflow(RuntimeError):
    """a rank's visible set outgrew its slab: `totals` holds every rank's true count; gather the full buffers with allgather_packed"""

    def __init__(self, totals, cap):
        super().__init__(f"visible sets {totals} exceed the slab of {cap} instances")
        self.totals, self.cap = totals, cap


def slab_words(cap):
    return HEADER_WORDS + cap * 17


def fill_slab(slab, ids, mats, total, frame=0):
    """what the pack kernel writes (tests build slabs on the CPU with it): header {written, total, frame, 0}, ids, matrices"""
    cap = (slab.numel() - HEADER_WORDS) // 17
    n = min(int(total), cap)
    slab[0], slab[1], slab[2], slab[3] = n, int(total), int(frame), 0
    slab[HEADER_WORDS:HEADER_WORDS + n] = ids[:n].to(torch.int32)
    slab[HEADER_WORDS + cap:].view(torch.float32).view(cap, 16)[:n] = mats[:n]
    return slab


def parse_slabs(recv, world, cap):
    """(ids, matrices, counts) in rank order from the all-gathered slabs; raises SlabCancelled / SlabOverflow from the headers every rank sees alike"""
    words = slab_words(cap)
    r = recv.view(world, words)
    written = [int(c) for c in r[:, 0].tolist()]
    totals = [int(c) for c in r[:, 1].tolist()]
    if any(w == SLAB_CANCELLED for w in written):
        raise SlabCancelled([k for k, w in enumerate(written) if w == SLAB_CANCELLED])
    if any(t > cap for t in totals):
        raise SlabOverflow(totals, cap)
    ids = torch.cat([r[k, HEADER_WORDS:HEADER_WORDS + written[k]] for k in range(world)])
    mats = torch.cat([r[k, HEADER_WORDS + cap:].view(torch.float32).view(cap, 16)[:written[k]] for k in range(world)])
    return ids, mats, written


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
        # the packed ids / matrices are only stream-ordered behind a synchronous re_cull_pack (it returns when the group table is on the host, before
        # the last matrix store): the collective runs on the pipeline's own stream
        st = torch.cuda.ExternalStream(self.p.stream())
        with torch.cuda.stream(st):
            self.last = allgather_packed(self.ids, self.mats, n_written, self.dist)
        st.synchronize()                               # the send slab is rewritten by the next cull
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

    HEADER_WORDS = HEADER_WORDS

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
        """(ids, matrices, counts) of buffer b in rank order, after finish().  Raises SlabCancelled when a rank's frame had been cancelled by cross-frame
        speculation when its slab went out (call regather(b) after Pipeline.wait()) and SlabOverflow when a visible set outgrew the slab."""
        return parse_slabs(self.recv[b], self.world, self.cap)

    def regather(self, b):
        """after Pipeline.wait() has replayed a cancelled frame into slab b: the all-gather of that slab once more (every rank, the headers are the same everywhere)"""
        self._gather(b); self.finish()
        return self.gathered(b)


class SlabAllGatherLanes:
    """The exchange for frame loops issued with defer_pack=True, two_lanes=True (Pipeline.cull_and_pack): frames alternate between two
    HIP streams and the pack of frame g rides in the launch of frame g + 2 on the same stream, so the slab of frame g - 2 is complete,
    in the order of the stream frame g was just issued on, as soon as that call returns.  Eight slabs rotate (frame g packs into slab
    g % 8); the all-gather of the slab of frame g - 2 is enqueued behind launch g.  Also correct when the library keeps the frames on one
    stream or packs at once (worlds with dynamic entities, first frames): a slab is then simply complete earlier than assumed.
    Call order per frame: begin_frame(); pipeline.cull_and_pack(..., asynchronous=True, defer_pack=True, two_lanes=True);
    after_cull(); at the end pipeline.wait(); finish()."""

    HEADER_WORDS = SlabAllGather.HEADER_WORDS
    DEPTH = 8                                       # a slab is rewritten 8 frames after it was filled: its collective (handed over 6 frames earlier at the latest) has long finished

    def __init__(self, pipeline, slab_instances, dist, group=None):
        self.p, self.dist, self.group, self.cap = pipeline, dist, group, int(slab_instances)
        self.world = dist.get_world_size(group)
        dev = torch.device("cuda", torch.cuda.current_device())
        words = self.HEADER_WORDS + self.cap * 17
        self.slab = [torch.zeros(words, dtype=torch.int32, device=dev) for _ in range(self.DEPTH)]
        self.recv = [torch.zeros(words * self.world, dtype=torch.int32, device=dev) for _ in range(self.DEPTH)]
        self.work = [None] * self.DEPTH
        self.gathered_upto = -1                     # newest frame whose slab has been handed to the collective
        self.frame = 0
        self._streams = {}                          # raw stream handle -> torch ExternalStream

    def _stream(self):
        h = self.p.stream()
        if h not in self._streams:
            self._streams[h] = torch.cuda.ExternalStream(h)
        return self._streams[h]

    def begin_frame(self):
        g = self.frame
        # the launch about to be issued may write the slab of frame g (pack at once), g - 1 or g - 2 (deferred packs): the previous
        # collectives on those slabs (six or more frames old) must have finished -- waited for on every stream the pipeline has used so far
        for b in {g % self.DEPTH, (g - 1) % self.DEPTH, (g - 2) % self.DEPTH}:
            if self.work[b] is not None:
                for st in self._streams.values():
                    with torch.cuda.stream(st):
                        self.work[b].wait()
                self.work[b] = None
        base = self.slab[g % self.DEPTH].data_ptr()
        self.p.set_output_count(base)
        self.p.set_output_buffers(base + 4 * self.HEADER_WORDS, base + 4 * (self.HEADER_WORDS + self.cap), self.cap)

    def _gather(self, frame, st):
        b = frame % self.DEPTH
        with torch.cuda.stream(st):
            self.work[b] = self.dist.all_gather_into_tensor(self.recv[b], self.slab[b], group=self.group, async_op=True)
        self.gathered_upto = frame

    def after_cull(self):
        g = self.frame
        st = self._stream()                         # the stream frame g was issued on
        if g >= 2:
            self._gather(g - 2, st)
        self.frame += 1

    def finish(self):
        """after Pipeline.wait() (every pending pack has run): the slabs of the last frames, then wait for all collectives"""
        st = self._stream()
        for f in range(max(self.gathered_upto + 1, 0), self.frame):
            self._gather(f, st)
        for b in range(self.DEPTH):
            if self.work[b] is not None:
                self.work[b].wait(); self.work[b] = None
        torch.cuda.synchronize()
        return (self.frame - 1) % self.DEPTH

    def gathered(self, b):
        return parse_slabs(self.recv[b], self.world, self.cap)
