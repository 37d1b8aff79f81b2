"""
Multi-GPU host logic: environments are independent (the reference runs one env per process),
so a global batch is cut into contiguous per-rank shards with NO data-path collective.  The
only exchange the north star asks for is the end-of-step observation gather -- RCCL over xGMI
on GPUs (backend "nccl"), gloo on CPU in the tests.  Two forms:

* ChunkedFeedbackGather (what bench.py runs at N > 1): one byte per env-step, one all-gather per
  64 steps, overlapped with the next chunk's stepping -- sized for xGMI, where a collective costs
  tens of microseconds however small it is;
* ObservationGather / PipelinedGather: the plain per-step form, every rank's (obs, reward, done)
  record all-gathered once per step.

The record is ONE contiguous byte buffer laid out [obs int32 x N | reward float32 x N |
done uint8 x N | pad]; the step kernel writes straight into its three views, so the gather
needs no packing kernel and is a single collective of 9*N (+pad) bytes per rank.
"""
import logging

import torch


def shard_range(global_envs, world_size, rank):
    """Contiguous block of env ids owned by `rank` (first `rem` ranks get one extra)."""
    if not (0 <= rank < world_size):
        raise ValueError("rank %d outside world of %d" % (rank, world_size))
    base, rem = divmod(int(global_envs), int(world_size))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


_log = logging.getLogger("gymwipe_amd.sharding")


def pick_all_gather(dist, group=None):
    """The asynchronous all-gather-into-one-tensor entry every gather object of this module uses, chosen ONCE (at
    construction) and for good: `fn(out, inp) -> Work`.

    On a real c10d process group that is the backend object's `_allgather_base` -- the collective
    `all_gather_into_tensor` issues, minus ~5 us of argument checking per call (at a 6 us step every microsecond the
    host spends here is a microsecond the GPU may run dry).  It is a private entry, so it is taken only if its
    signature reads `(output, input, ...)`; otherwise -- and for the loop-back stand-ins the tests inject -- the public
    call is bound instead, and the choice is logged once.  A collective is NEVER re-issued after an exception from the
    backend call itself: a rank that enqueued two collectives against its peers' one would mismatch or hang."""
    pg = group
    if pg is None:
        try:
            pg = dist.distributed_c10d._get_default_group()
        except Exception:
            pg = getattr(getattr(dist, "group", None), "WORLD", None)
    base = getattr(pg, "_allgather_base", None)
    doc = getattr(base, "__doc__", None) or ""
    if base is not None and "output: torch.Tensor, input: torch.Tensor" in doc:
        return base, "ProcessGroup._allgather_base"
    if base is not None:
        _log.warning("gymwipe_amd.sharding: ProcessGroup._allgather_base has an unexpected signature; "
                     "using torch.distributed.all_gather_into_tensor")
    if group is not None:
        return (lambda out, inp: dist.all_gather_into_tensor(out, inp, group=group, async_op=True)), "all_gather_into_tensor"
    return (lambda out, inp: dist.all_gather_into_tensor(out, inp, async_op=True)), "all_gather_into_tensor"


class StepRecord:
    """Packed per-step output record of one rank: three typed views over one byte buffer."""
    ALIGN = 16

    def __init__(self, num_envs, device):
        n = int(num_envs)
        self.num_envs = n
        self.nbytes = (9 * n + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        self.buf = torch.zeros(self.nbytes, dtype=torch.uint8, device=device)
        self.obs = self.buf[0:4 * n].view(torch.int32)
        self.reward = self.buf[4 * n:8 * n].view(torch.float32)
        self.done = self.buf[8 * n:9 * n]

    @staticmethod
    def split(flat, num_envs):
        """Views (obs, reward, done) into a flat record buffer (1-D uint8 tensor)."""
        n = int(num_envs)
        return (flat[0:4 * n].view(torch.int32), flat[4 * n:8 * n].view(torch.float32), flat[8 * n:9 * n])


class PipelinedGather:
    """Double-buffered end-of-step gather: step k writes record k % 2 while the all-gather of step
    k-1 is still in flight on the RCCL stream, so the exchange overlaps the next step's kernel."""

    def __init__(self, num_envs, device, world_size=None, depth=2):
        self.records = [StepRecord(num_envs, device) for _ in range(depth)]
        self.gathers = [ObservationGather(r, world_size) for r in self.records]
        self.pending = [None] * depth
        self.depth = depth
        self.k = 0

    def current(self):
        """The record the NEXT step must write; waits for the gather that last used it."""
        i = self.k % self.depth
        if self.pending[i] is not None:
            self.pending[i].wait()
            self.pending[i] = None
        return self.records[i]

    def submit(self):
        """Start gathering the record just written."""
        i = self.k % self.depth
        self.pending[i] = self.gathers[i](async_op=True)
        self.k += 1
        return self.gathers[i]

    def drain(self):
        for i, w in enumerate(self.pending):
            if w is not None:
                w.wait()
                self.pending[i] = None


class ObservationGather:
    """The end-of-step observation gather the north star names: all-gather of equally sized StepRecords over the
    default process group (or `group`), once per env.step()."""

    def __init__(self, record, world_size=None, dist_module=None, group=None):
        if dist_module is None:                      # tests may inject a loop-back stand-in for the process group
            import torch.distributed as dist_module
        dist = self._dist = dist_module
        self.world = world_size if world_size is not None else dist.get_world_size()
        self.record = record
        self.out = torch.empty(self.world * record.nbytes, dtype=torch.uint8, device=record.buf.device)
        self._gather, self.entry = pick_all_gather(dist, group)
        self._buf = record.buf

    def __call__(self, async_op=False):
        """Gather every rank's record.  Synchronous form: returns the [world, nbytes] byte tensor, valid for work
        enqueued afterwards on the current stream (on RCCL `wait()` makes the current STREAM wait for the collective;
        the host does not block).  With async_op=True the collective runs on the backend's own stream and the Work
        handle is returned: the caller may launch the next env.step() right away and must wait() on the handle before
        this record's buffer (or `out`) is written again."""
        work = self._gather(self.out, self._buf)
        if async_op:
            return work
        work.wait()
        return self.out.view(self.world, self.record.nbytes)

    def step_done(self):
        """The per-step call of a stepping loop: gather this step's record and order the current stream behind it, so
        that whatever is launched next (the next env.step(), the agent's forward pass) sees every rank's feedback."""
        self._gather(self.out, self._buf).wait()

    def unpack(self):
        """(obs[W*N], reward[W*N], done[W*N]) of the whole job, rank-major (== global env order
        for equal shards)."""
        n = self.record.num_envs
        rows = self.out.view(self.world, self.record.nbytes)
        parts = [StepRecord.split(rows[r], n) for r in range(self.world)]
        return (torch.cat([p[0] for p in parts]), torch.cat([p[1] for p in parts]),
                torch.cat([p[2] for p in parts]))


class ChunkedFeedbackGather:
    """The end-of-step observation gather, batched and compressed for xGMI.

    A per-step all-gather of 9*N bytes is latency-bound (tens of microseconds per collective against a
    7 us step), so the exchange is made coarser and smaller instead: the step kernel writes the outputs
    of `chunk` consecutive steps into one [chunk][N] record, a streaming kernel packs them to ONE byte per
    env-step (gw_pack_feedback: lossless for the built-in interpreter), and one all-gather per chunk moves
    chunk*N bytes per rank while the next chunk is already being stepped (double-buffered; the collective
    runs on the backend's stream).  Nothing about the environments themselves is exchanged.

    `pack(obs, reward, done, out)` / `unpack(packed, obs, reward, done)` are the byte codec: on a GPU pass
    `env.pack_feedback` / `env.unpack_feedback` (the HIP kernels behind the C-ABI) -- or `pack=None` when the
    step itself writes its byte row (`env.feedback_bytes_into(gather.byte_slot())` before each step: the default
    step kernel stores the byte along with its outputs, so a chunk costs no packing launch at all).
    """

    def __init__(self, num_envs, device, pack, world_size=None, chunk=64, depth=2, dist_module=None, group=None):
        if dist_module is None:                      # tests may inject a loop-back stand-in for the process group
            import torch.distributed as dist_module
        dist = self._dist = dist_module
        self.world = world_size if world_size is not None else dist.get_world_size()
        n, g = int(num_envs), int(chunk)
        self.num_envs, self.chunk, self.depth = n, g, depth
        self._pack = pack
        self._gather, self.entry = pick_all_gather(dist, group)
        mk = lambda dt: [torch.zeros((g, n), dtype=dt, device=device) for _ in range(depth)]
        self.obs, self.reward, self.done = mk(torch.int32), mk(torch.float32), mk(torch.uint8)
        self.packed = mk(torch.uint8)
        self.gathered = [torch.zeros((self.world, g, n), dtype=torch.uint8, device=device) for _ in range(depth)]
        self.pending = [None] * depth
        self.filled = [0] * depth
        self.k = 0                                   # steps handed out so far
        # the per-step views, made once: slot() is on the host's critical path (one call per 7 us step)
        self._views = [[(self.obs[b][j], self.reward[b][j], self.done[b][j]) for j in range(g)] for b in range(depth)]
        self._byte_rows = [[self.packed[b][j] for j in range(g)] for b in range(depth)]
        self._slots = [self._views[b][j] + (self._byte_rows[b][j],) for b in range(depth) for j in range(g)]
        if pack is None and self.obs[0].is_cuda:     # steps write their own byte rows: hand out ready-made StepOutputs
            from .envs.counter_traffic import StepOutputs
            self._slots = [StepOutputs(*v) for v in self._slots]

    def slot(self):
        """(obs, reward, done) views the NEXT step must write into."""
        b, j = (self.k // self.chunk) % self.depth, self.k % self.chunk
        if j == 0 and self.pending[b] is not None:   # this buffer's previous gather must have landed
            self.pending[b].wait()
            self.pending[b] = None
        return self._views[b][j]

    # -- the same protocol with ONE call per step (the host has ~1 us to spare per 6-us step) --------------------------
    #    views = g.begin()            # (obs, reward, done, byte_row) of the next step; again after every drain()
    #    loop:  step into `views`;  views = g.advance()
    #    With pack=None on a GPU the views are `StepOutputs` objects: `env.step(action, out=views)`.
    def begin(self):
        """(obs, reward, done, byte_row) views the next step must write into."""
        b, j = (self.k // self.chunk) % self.depth, self.k % self.chunk
        if j == 0 and self.pending[b] is not None:   # this buffer's previous gather must have landed
            self.pending[b].wait()
            self.pending[b] = None
        return self._slots[b * self.chunk + j]

    def advance(self):
        """After a step: submits the chunk if that step filled it, and returns the views of the step after it."""
        k = self.k
        self.k = k + 1
        j = k % self.chunk
        if j + 1 == self.chunk:
            self._submit((k // self.chunk) % self.depth, self.chunk)
            return self.begin()
        return self._slots[((k // self.chunk) % self.depth) * self.chunk + j + 1]

    def byte_slot(self):
        """The uint8[N] row of the packed chunk record that belongs to the NEXT step (for steps that write it themselves)."""
        b, j = (self.k // self.chunk) % self.depth, self.k % self.chunk
        return self._byte_rows[b][j]

    def stepped(self):
        """Call after each step; starts the chunk's pack + all-gather when the chunk is full.
        Returns the buffer index that was submitted, or None."""
        b, j = (self.k // self.chunk) % self.depth, self.k % self.chunk
        self.k += 1
        if j + 1 < self.chunk:
            return None
        return self._submit(b, self.chunk)

    def _submit(self, b, steps):
        if self._pack is not None:                   # (None: every step of the chunk wrote its own row, see byte_slot)
            self._pack(self.obs[b][:steps], self.reward[b][:steps], self.done[b][:steps], self.packed[b][:steps])
        self.filled[b] = steps
        # only the rows that were filled travel: a partial chunk (the flush at the end of a short run) costs steps/chunk of a
        # full one on the links, not all of it
        cnt = steps * self.num_envs
        out = self.gathered[b].view(-1)[:self.world * cnt]
        self.pending[b] = self._gather(out, self.packed[b].view(-1)[:cnt])
        return b

    def drain(self):
        """Flush a partly filled chunk and wait for every gather in flight."""
        j = self.k % self.chunk
        if j:
            b = (self.k // self.chunk) % self.depth
            self._submit(b, j)
            self.k += self.chunk - j                 # the next step starts a fresh chunk
        for b, w in enumerate(self.pending):
            if w is not None:
                w.wait()
                self.pending[b] = None

    def result(self, b):
        """Packed feedback of the whole job for the chunk last gathered into buffer b: uint8[world][steps][N]."""
        steps = self.filled[b]
        return self.gathered[b].view(-1)[:self.world * steps * self.num_envs].view(self.world, steps, self.num_envs)
