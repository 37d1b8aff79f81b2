#!/usr/bin/env python3
"""Five-minute tour (needs an MI355X): the reference's scalar env, the vectorised env, a fused rollout, a custom
interpreter, the pendulum env (open and closed loop) and the PHY grid.  `python examples/quickstart.py`"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import gymwipe_amd
from gymwipe_amd import VecInterpreter

# 1. the reference's own usage, unchanged (tests/envs/test_counter_traffic.py:17-34)
env = gymwipe_amd.make("CounterTraffic-v0")
print("scalar env:", env.step({"device": 0, "duration": 3}), env.step({"device": 1, "duration": 12}))

# 2. the same environment 65 536 times on one GPU; actions and results are tensors that never leave it
N = 65536
venv = gymwipe_amd.VecCounterTrafficEnv(N, num_devices=4)
obs = venv.reset()
for _ in range(32):
    action = {"device": torch.randint(0, 4, (N,), dtype=torch.int32, device="cuda"),
              "duration": torch.randint(0, 20, (N,), dtype=torch.int32, device="cuda")}
    obs, reward, done, info = venv.step(action)
print("vectorised:", obs[:4].tolist(), reward[:4].tolist(), venv.stats())

# 2b. preallocated outputs (their device addresses are taken once): what a tight loop, or a multi-GPU gather record, steps into
slot = gymwipe_amd.StepOutputs(torch.empty(N, dtype=torch.int32, device="cuda"), torch.empty(N, dtype=torch.float32, device="cuda"),
                               torch.empty(N, dtype=torch.uint8, device="cuda"),
                               feedback_bytes=torch.empty(N, dtype=torch.uint8, device="cuda"))   # + the one-byte exchange format
obs, reward, done, info = venv.step(action, out=slot)
print("step(out=):", obs is slot.obs, slot.feedback_bytes[:4].tolist())

# 3. pre-staged actions: 64 steps in one persistent launch
dev = torch.randint(0, 4, (64, N), dtype=torch.int32, device="cuda")
dur = torch.randint(0, 20, (64, N), dtype=torch.int32, device="cuda")
o, r, d = venv.rollout(dev, dur)
print("rollout:", tuple(o.shape), float(r.float().mean()))


# 4. your own Interpreter (envs/core.py:59-159), fed with what the RRM sniffed each step
class CountDeliveries(VecInterpreter):
    def __init__(self, n):
        self.total = torch.zeros(n, dtype=torch.int64, device="cuda")
        self.last = torch.zeros(n, dtype=torch.int32, device="cuda")

    def onPacketReceived(self, senderIndex, receiverIndex, payload):
        self.last = payload.count
        self.total += payload.count

    def getObservation(self):
        return self.last

    def getReward(self):
        return self.last.float()

    def getDone(self):
        return torch.zeros_like(self.last, dtype=torch.bool)

    def reset(self):
        self.total.zero_()


cenv = gymwipe_amd.VecCounterTrafficEnv(1024, num_devices=2, interpreter=CountDeliveries(1024))
cenv.reset()
for k in range(16):
    fb = cenv.step({"device": torch.full((1024,), k % 2, dtype=torch.int32, device="cuda"),
                    "duration": torch.full((1024,), 10, dtype=torch.int32, device="cuda")})
print("custom interpreter: packets decoded by the RRM per env so far:", int(cenv.interpreter.total[0]))

# 5. the pendulum band-assignment env (builder-defined linear plant on the f64 matrix cores)
penv = gymwipe_amd.make("VecInvertedPendulum-v0", num_envs=4096)
for _ in range(8):
    pobs, prew, pdone, pinfo = penv.step({"device": torch.zeros(4096, dtype=torch.int32, device="cuda"),
                                          "duration": torch.full((4096,), 19, dtype=torch.int32, device="cuda")})
print("pendulum:", int(pobs[0]), float(prew[0]), float(pinfo["Sensor angle"][0]))

# 5b. the same env with its control loop closed over receive-mode MACs (sensor -> controller -> actuator)
loop = gymwipe_amd.make("VecControlLoop-v0", num_envs=4096)
for k in range(60):                                   # alternate the band between sensor and controller, 5 ms each
    lobs, lrew, ldone, linfo = loop.step({"device": torch.full((4096,), k % 2, dtype=torch.int32, device="cuda"),
                                          "duration": torch.full((4096,), 5, dtype=torch.int32, device="cuda")})
rec = loop.get_state("received")
print("control loop: angle %.3f deg, %d samples reached the controller, %d commands the actuator, motor velocity %.3f"
      % (float(linfo["Sensor angle"][0]), int(rec[0, 0]), int(rec[0, 1]), float(loop.get_state("u")[0])))

# 6. the reference's PHY-grid benchmark (tests/test_benchmark.py), 1 024 replicas of a 16-device grid
import numpy as np
grid = gymwipe_amd.VecPhyGrid(1024, 16, np.random.default_rng(0).uniform(0, 1e-2, (1024, 16)))
grid.runSimulation(0.1)
print("grid: events per replica", float(grid.get_state("events").mean()))
