"""
BASELINE config 5's SHAPE on one GPU: 524 288 envs of 4 devices cut into 8 shards of 65 536 (one handle per
shard, as one process per GPU would hold) must be indistinguishable from ONE handle of 524 288 envs -- outputs of
every step, final state -- with the shards' feedback travelling through both forms of the end-of-step observation
gather (gymwipe_amd/sharding.py).  The collective itself is a loop-back stand-in (one process cannot hold eight
ranks): it places every shard's send buffer where an all-gather would, so everything around it -- the step kernel
writing into the chunk records, the pack kernel, buffer rotation, rank-major layout -- is the production code.
More than one RCCL rank needs more than one GPU and is left to the driver's 8-GPU run.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

WORLD, SHARD, D, K = 8, 65536, 4, 64


class LoopbackWorld:
    """Stand-in for a process group inside ONE process: the k-th all_gather_into_tensor call of every rank forms
    collective number k; it completes when the last rank has called."""

    def __init__(self, world):
        self.world = world
        self.calls = {}                                   # sequence number -> {rank: (out, inp)}
        self.seq = [0] * world

    class _Work:
        def __init__(self, owner, k):
            self.owner, self.k = owner, k

        def wait(self):
            assert self.k not in self.owner.calls, "collective %d waited for before every rank joined it" % self.k

    class _Rank:
        def __init__(self, owner, rank):
            self.owner, self.rank = owner, rank

        def get_world_size(self):
            return self.owner.world

        def all_gather_into_tensor(self, out, inp, async_op=False):
            o = self.owner
            k = o.seq[self.rank]
            o.seq[self.rank] += 1
            slot = o.calls.setdefault(k, {})
            slot[self.rank] = (out, inp)
            if len(slot) == o.world:
                n = inp.numel()
                for r in range(o.world):
                    assert slot[r][1].numel() == n
                for out_r, _ in slot.values():
                    flat = out_r.view(-1)
                    for r in range(o.world):
                        flat[r * n:(r + 1) * n].copy_(slot[r][1].view(-1))
                del o.calls[k]
            return LoopbackWorld._Work(o, k)

    def rank(self, r):
        return LoopbackWorld._Rank(self, r)


def test_eight_shards_equal_one_handle_with_both_gathers():
    import torch
    import gymwipe_amd
    from gymwipe_amd.actions import actions_torch
    from gymwipe_amd.sharding import ChunkedFeedbackGather, ObservationGather, StepRecord, shard_range

    G = WORLD * SHARD
    whole = gymwipe_amd.VecCounterTrafficEnv(G, num_devices=D)
    shards = [gymwipe_amd.VecCounterTrafficEnv(SHARD, num_devices=D) for _ in range(WORLD)]
    lo_hi = [shard_range(G, WORLD, r) for r in range(WORLD)]
    assert lo_hi[0] == (0, SHARD) and lo_hi[-1] == (G - SHARD, G)

    a_dev, a_dur = actions_torch(1234, 0, G, 0, K, D, device="cuda")     # the job's global action stream
    w_obs = torch.empty((K, G), dtype=torch.int32, device="cuda")
    w_rew = torch.empty((K, G), dtype=torch.float32, device="cuda")
    w_done = torch.empty((K, G), dtype=torch.uint8, device="cuda")

    chunk = 16                                               # 4 chunks in 64 steps: the double buffers rotate twice
    lb_c, lb_s = LoopbackWorld(WORLD), LoopbackWorld(WORLD)
    # pack=None + begin()/advance(): bench.py's production protocol -- every step writes its own one-byte row (gw_step_fb)
    pipes = [ChunkedFeedbackGather(SHARD, "cuda", None, WORLD, chunk=chunk, dist_module=lb_c.rank(r)) for r in range(WORLD)]
    views = [p.begin() for p in pipes]
    recs = [StepRecord(SHARD, "cuda") for _ in range(WORLD)]
    gathers = [ObservationGather(recs[r], WORLD, dist_module=lb_s.rank(r)) for r in range(WORLD)]
    works = [None] * WORLD

    whole.reset()
    for s in shards:
        s.reset()
    for k in range(K):
        if k == 40:                                          # a reset in the middle, on every handle
            whole.reset()
            for s in shards:
                s.reset()
        whole._obs, whole._rew, whole._done = w_obs[k], w_rew[k], w_done[k]
        whole.step({"device": a_dev[k], "duration": a_dur[k]})
        for r, s in enumerate(shards):
            lo, hi = lo_hi[r]
            act = {"device": a_dev[k, lo:hi], "duration": a_dur[k, lo:hi]}
            if k < K // 2:                                   # first half: the chunked byte gather
                s.step(act, out=views[r])                    # a StepOutputs: typed outputs + the step's one-byte row
                views[r] = pipes[r].advance()
            else:                                            # second half: the literal per-step record gather
                s.feedback_bytes_into(None)
                s._obs, s._rew, s._done = recs[r].obs, recs[r].reward, recs[r].done
                s.step(act)
                works[r] = gathers[r](async_op=True)         # (one process plays all ranks: wait once every rank has joined)
        if k >= K // 2:
            for w in works:
                w.wait()
        if k < K // 2 and (k + 1) % chunk == 0:              # a chunk was gathered: every rank holds the whole job's bytes
            b = (k // chunk) % 2
            k0 = k + 1 - chunk
            want = whole.pack_feedback(w_obs[k0:k + 1].contiguous(), w_rew[k0:k + 1].contiguous(), w_done[k0:k + 1].contiguous(), check=True)
            for r in (0, WORLD - 1):
                got = pipes[r].result(b)                     # uint8[world][chunk][SHARD]
                for q in range(WORLD):
                    lo, hi = lo_hi[q]
                    assert torch.equal(got[q], want[:, lo:hi]), "chunk ending at step %d, rank %d's copy of shard %d" % (k, r, q)
        if k >= K // 2:
            for r in (0, WORLD - 1):
                o, rw, dn = gathers[r].unpack()              # rank-major == global env order
                assert torch.equal(o, w_obs[k]) and torch.equal(rw, w_rew[k]) and torch.equal(dn, w_done[k]), "step %d" % k
    for p in pipes:
        p.drain()
    assert not lb_c.calls and not lb_s.calls                 # every collective completed

    for f in ("now", "wake", "counter", "qlen", "received", "last_abs", "rx_power", "flags", "n_tx", "n_delivered",
              "n_appended", "n_popped", "n_dropped"):
        a = whole.get_state(f)
        b = np.concatenate([s.get_state(f) for s in shards])
        assert a.shape == b.shape and (a.view(np.uint8) == b.view(np.uint8)).all(), f
    tot = whole.check()
    parts = [s.check() for s in shards]
    for key in ("steps", "transmissions", "delivered", "appended", "popped", "dropped"):
        assert tot[key] == sum(p[key] for p in parts), key
    assert tot["steps"] == G * K and tot["delivered"] > 0
