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


# ---- the chunked, one-byte-per-env-step gather (bench.py's default at N > 1) -------------------------
def byte_pack(obs, reward, done, out, bound=65536):
    """Host restatement of gw_pack_feedback's format, for the gloo ranks (the HIP kernel needs a GPU; the
    GPU tier checks the kernel against this same function)."""
    sgn = torch.sign(obs - bound).to(torch.int32)
    out.copy_(((sgn + 1) | ((reward.to(torch.int32) + 10) << 2) | (done.to(torch.int32) << 7)).to(torch.uint8))
    return out


def byte_unpack(packed, bound=65536, pv=2):
    b = packed.to(torch.int32)
    return (bound + pv * ((b & 3) - 1)).to(torch.int32), (((b >> 2) & 31) - 10).to(torch.float32), (b >> 7).to(torch.uint8)


def _chunk_worker(rank, world, port, total, D, K, chunk, seed, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gymwipe_amd.sharding import ChunkedFeedbackGather, shard_range
        from oracle.ct_oracle import CtOracle
        lo, hi = shard_range(total, world, rank)
        n = hi - lo
        rng = np.random.default_rng(seed)
        dev = rng.integers(0, D, (K, total), dtype=np.int32)
        dur = rng.integers(0, 20, (K, total), dtype=np.int32)
        shard = CtOracle(n, D)
        shard.reset()
        cg = ChunkedFeedbackGather(n, "cpu", byte_pack, world, chunk=chunk)
        got = []

        def collect(b):
            cg.pending[b].wait()
            got.append(cg.result(b).clone())                     # uint8[world][steps][n]

        one_call = (K + chunk) % 2 == 1                          # half of the cases through the one-call-per-step protocol
        views = cg.begin() if one_call else None
        for k in range(K):
            if one_call:                                         # begin() / advance(): what bench.py uses
                obs, rew, done = views[0], views[1], views[2]
            else:                                                # slot() / stepped()
                obs, rew, done = cg.slot()
            o, r, d = shard.step(dev[k, lo:hi], dur[k, lo:hi])
            obs.copy_(torch.from_numpy(o)); rew.copy_(torch.from_numpy(r)); done.copy_(torch.from_numpy(d))
            if one_call:
                views = cg.advance()
                b = ((k // chunk) % cg.depth) if (k + 1) % chunk == 0 else None
            else:
                b = cg.stepped()
            if b is not None:
                collect(b)
        tail = K % chunk
        if tail:
            b = (cg.k // chunk) % cg.depth
            cg.drain()
            got.append(cg.result(b).clone())
        else:
            cg.drain()
        if rank == 0:
            packed = torch.cat(got, dim=1)                       # [world][K][n]
            assert packed.shape == (world, K, n)
            o, r, d = byte_unpack(packed)
            whole = CtOracle(total, D)
            whole.reset()
            for k in range(K):
                wo, wr, wd = whole.step(dev[k], dur[k])
                assert (o[:, k].reshape(-1).numpy() == wo).all(), k   # rank-major == global env order
                assert (r[:, k].reshape(-1).numpy() == wr).all() and (d[:, k].reshape(-1).numpy() == wd).all(), k
        dist.barrier()
        open(os.path.join(out_dir, "ok%d" % rank), "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("K,chunk", [(12, 4), (10, 4), (3, 8), (13, 4), (9, 3)])
def test_two_rank_chunked_feedback_gather(tmp_path, K, chunk):
    world, total, D = 2, 64, 4
    port = _free_port()
    mp.spawn(_chunk_worker, args=(world, port, total, D, K, chunk, 33, str(tmp_path)), nprocs=world, join=True)
    assert sorted(os.listdir(tmp_path)) == ["ok0", "ok1"]


def test_feedback_byte_codec_round_trip():
    obs = torch.tensor([65534, 65536, 65538, 65536], dtype=torch.int32)
    rew = torch.tensor([-10.0, 0.0, 10.0, 2.0])
    done = torch.tensor([0, 1, 0, 1], dtype=torch.uint8)
    out = torch.empty(4, dtype=torch.uint8)
    o, r, d = byte_unpack(byte_pack(obs, rew, done, out))
    assert o.tolist() == obs.tolist() and r.tolist() == rew.tolist() and d.tolist() == done.tolist()
