#!/usr/bin/env python3
"""Where the host's time goes around one env.step(): per-call enqueue cost (Python + ctypes + hipLaunchKernel), and the
latency of torch.cuda.synchronize() after the last launch.  Prints one JSON line."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gymwipe_amd
from gymwipe_amd.actions import actions_torch

out = {}
for N in (64, 65536):
    env = gymwipe_amd.VecCounterTrafficEnv(N, num_devices=4)
    a_dev, a_dur = actions_torch(1234, 0, N, 0, 64, 4, device="cuda")
    acts = [{"device": a_dev[i], "duration": a_dur[i]} for i in range(64)]
    env.reset()
    for i in range(64):
        env.step(acts[i])
    torch.cuda.synchronize()
    # (a) enqueue cost: K calls, clock stopped BEFORE the synchronize
    best_enq, best_tot = 1e9, 1e9
    for rep in range(20):
        env.reset()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(20):
            env.step(acts[i + 5])
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        best_enq = min(best_enq, (t1 - t0) / 20)
        best_tot = min(best_tot, (t2 - t0) / 20)
    out["N=%d" % N] = {"enqueue_us_per_step": best_enq * 1e6, "wall_us_per_step_incl_sync": best_tot * 1e6}
    # (a') the same with preallocated outputs, env.step(action, out=StepOutputs): no data_ptr() calls for the outputs
    slot = gymwipe_amd.StepOutputs(torch.empty(N, dtype=torch.int32, device="cuda"), torch.empty(N, dtype=torch.float32, device="cuda"),
                                   torch.empty(N, dtype=torch.uint8, device="cuda"))
    best_out = 1e9
    for rep in range(20):
        env.reset()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(20):
            env.step(acts[i + 5], slot)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        best_out = min(best_out, (t1 - t0) / 20)
    out["N=%d" % N]["enqueue_us_per_step_with_out"] = best_out * 1e6
    # (b) one launch + synchronize
    lat = []
    for rep in range(50):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        env.step(acts[40])
        torch.cuda.synchronize()
        lat.append(time.perf_counter() - t0)
    lat.sort()
    out["N=%d" % N]["single_step_launch_to_sync_us_median"] = lat[len(lat) // 2] * 1e6
    out["N=%d" % N]["single_step_launch_to_sync_us_min"] = lat[0] * 1e6
    # (c) the C side alone: gw_step through ctypes with every argument resolved beforehand
    L, h = env._L, env._h
    ptrs = [(acts[i]["device"].data_ptr(), acts[i]["duration"].data_ptr()) for i in range(64)]
    o, r, d = env._obs.data_ptr(), env._rew.data_ptr(), env._done.data_ptr()
    stream = torch.cuda.current_stream().cuda_stream
    best_c = 1e9
    for rep in range(20):
        env.reset()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(5, 25):
            L.gw_step(h, ptrs[i][0], ptrs[i][1], o, r, d, stream)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        best_c = min(best_c, (t1 - t0) / 20)
    out["N=%d" % N]["c_abi_enqueue_us_per_step"] = best_c * 1e6
    # (d) the same through the CPython fast-call shim (what env.step() uses when it is built)
    f = env._fast
    if f is not None:
        hv = h.value
        best_f = 1e9
        for rep in range(20):
            env.reset()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(5, 25):
                f.step(hv, ptrs[i][0], ptrs[i][1], o, r, d, stream)
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            best_f = min(best_f, (t1 - t0) / 20)
        out["N=%d" % N]["fastcall_enqueue_us_per_step"] = best_f * 1e6
    env.close()
t0 = time.perf_counter()
for _ in range(1000):
    torch.cuda.synchronize()
out["empty_synchronize_us"] = (time.perf_counter() - t0) / 1000 * 1e6
print(json.dumps(out))
