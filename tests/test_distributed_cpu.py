"""
CPU tier: the N > 1 path with two gloo ranks.  Envs shard as contiguous blocks with no data-path
collective; the only exchange is the end-of-step all-gather of the packed (obs, reward, done)
record (gymwipe_amd/sharding.py).  Here each rank steps ITS shard with the C oracle (the HIP
kernel needs a GPU) and the gathered records must equal one oracle stepping the whole batch.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, total, D, K, seed, out_dir, pipelined=False):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gymwipe_amd.sharding import ObservationGather, PipelinedGather, StepRecord, shard_range
        from oracle.ct_oracle import CtOracle
        lo, hi = shard_range(total, world, rank)
        n = hi - lo
        rng = np.random.default_rng(seed)                       # same global action stream on every rank
        dev = rng.integers(0, D, (K, total), dtype=np.int32)
        dur = rng.integers(0, 20, (K, total), dtype=np.int32)
        shard = CtOracle(n, D)
        rec = StepRecord(n, "cpu")
        gather = ObservationGather(rec, world)
        pipe = PipelinedGather(n, "cpu", world) if pipelined else None
        shard.reset()
        got = []
        for k in range(K):
            o, r, d = shard.step(dev[k, lo:hi], dur[k, lo:hi])
            cur = pipe.current() if pipelined else rec       # waits for the gather that last used this buffer
            cur.obs.copy_(torch.from_numpy(o))
            cur.reward.copy_(torch.from_numpy(r))
            cur.done.copy_(torch.from_numpy(d))
            if pipelined:
                g = pipe.submit()                            # asynchronous all-gather of the record just written
                pipe.pending[(pipe.k - 1) % pipe.depth].wait()
            else:
                g = gather
                g()
            go, gr, gd = g.unpack()
            got.append((go.numpy().copy(), gr.numpy().copy(), gd.numpy().copy()))
        if pipelined:
            pipe.drain()
        if rank == 0:
            whole = CtOracle(total, D)
            whole.reset()
            for k in range(K):
                o, r, d = whole.step(dev[k], dur[k])
                assert (got[k][0] == o).all() and (got[k][1] == r).all() and (got[k][2] == d).all(), k
        dist.barrier()
        open(os.path.join(out_dir, "ok%d" % rank), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_two_rank_shard_and_gather(tmp_path):
    world, total, D, K = 2, 96, 4, 12
    port = _free_port()
    mp.spawn(_worker, args=(world, port, total, D, K, 31, str(tmp_path)), nprocs=world, join=True)
    assert sorted(os.listdir(tmp_path)) == ["ok0", "ok1"]


def test_two_rank_pipelined_gather(tmp_path):
    """The double-buffered asynchronous gather bench.py uses at N > 1."""
    world, total, D, K = 2, 64, 2, 9
    port = _free_port()
    mp.spawn(_worker, args=(world, port, total, D, K, 32, str(tmp_path), True), nprocs=world, join=True)
    assert sorted(os.listdir(tmp_path)) == ["ok0", "ok1"]


def test_step_record_layout():
    from gymwipe_amd.sharding import StepRecord
    rec = StepRecord(5, "cpu")
    assert rec.nbytes == 48 and rec.buf.numel() == 48            # 9*5 = 45 -> 16-byte multiple
    rec.obs[:] = torch.arange(5, dtype=torch.int32) + 65534
    rec.reward[:] = torch.tensor([-2.0, 0.0, 2.0, 0.0, -2.0])
    rec.done[:] = torch.tensor([0, 1, 0, 0, 1], dtype=torch.uint8)
    o, r, d = StepRecord.split(rec.buf, 5)
    assert o.tolist() == [65534, 65535, 65536, 65537, 65538]
    assert r.tolist() == [-2.0, 0.0, 2.0, 0.0, -2.0] and d.tolist() == [0, 1, 0, 0, 1]
    assert o.data_ptr() == rec.buf.data_ptr() and r.data_ptr() == rec.buf.data_ptr() + 20
