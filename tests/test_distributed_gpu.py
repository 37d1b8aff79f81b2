"""
GPU tier, two processes: each rank steps ITS shard of the batch with the HIP kernels on the one GPU of the box, the
end-of-step feedback travels through ChunkedFeedbackGather (packed on the GPU, gathered by gloo on the host because
two RCCL ranks cannot share a device), and rank 0 checks the gathered job against one oracle stepping the whole batch.
The RCCL form of the same gather is covered in-process by test_chunked_feedback_gather_over_rccl_single_rank.
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


def _worker(rank, world, port, total, D, K, chunk, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gymwipe_amd import VecCounterTrafficEnv
        from gymwipe_amd.sharding import ChunkedFeedbackGather, shard_range
        from oracle.ct_oracle import CtOracle
        from test_distributed_cpu import byte_unpack
        lo, hi = shard_range(total, world, rank)
        n = hi - lo
        rng = np.random.default_rng(17)                      # the same global action stream on every rank
        dev = rng.integers(0, D, (K, total), dtype=np.int32)
        dur = rng.integers(0, 20, (K, total), dtype=np.int32)
        env = VecCounterTrafficEnv(n, D, device="cuda:0")
        stage = torch.empty((chunk, n), dtype=torch.uint8, device="cuda:0")

        def pack_to_host(o, r, d, out):
            env.pack_feedback(o, r, d, stage[:o.shape[0]])
            out.copy_(stage[:o.shape[0]])
        cg = ChunkedFeedbackGather(n, torch.device("cuda:0"), pack_to_host, world, chunk=chunk)
        cg.packed = [torch.zeros((chunk, n), dtype=torch.uint8) for _ in range(cg.depth)]
        cg.gathered = [torch.zeros((world, chunk, n), dtype=torch.uint8) for _ in range(cg.depth)]
        env.reset()
        got = []
        for k in range(K):
            env._obs, env._rew, env._done = cg.slot()
            env.step({"device": torch.from_numpy(dev[k, lo:hi]).cuda(), "duration": torch.from_numpy(dur[k, lo:hi]).cuda()})
            b = cg.stepped()
            if b is not None:
                cg.pending[b].wait()
                got.append(cg.result(b).clone())
        if K % chunk:
            b = (cg.k // chunk) % cg.depth
            cg.drain()
            got.append(cg.result(b).clone())
        env.check()
        if rank == 0:
            packed = torch.cat(got, dim=1)
            assert packed.shape == (world, K, n)
            o, r, d = byte_unpack(packed)
            whole = CtOracle(total, D)
            whole.reset()
            for k in range(K):
                wo, wr, wd = whole.step(dev[k], dur[k])
                assert (o[:, k].reshape(-1).numpy() == wo).all(), k
                assert (r[:, k].reshape(-1).numpy() == wr).all() and (d[:, k].reshape(-1).numpy() == wd).all(), k
        dist.barrier()
        open(os.path.join(out_dir, "ok%d" % rank), "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_two_ranks_shard_the_batch_on_the_gpu(tmp_path):
    world, total, D, K, chunk = 2, 4096, 4, 150, 64
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_worker, args=(world, port, total, D, K, chunk, str(tmp_path)), nprocs=world, join=True)
    assert sorted(os.listdir(tmp_path)) == ["ok0", "ok1"]
