#!/usr/bin/env python3
"""The reference's agents/dqn_counter_traffic.py, on the vectorised env: a torch DQN whose policy, the env
step and the replay memory all stay on the GPU.  python examples/dqn_counter_traffic.py [num_envs] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gymwipe_amd
from gymwipe_amd.agents import DqnCounterTrafficAgent

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 300
env = gymwipe_amd.make("VecCounterTraffic-v0", num_envs=N, num_devices=2)
agent = DqnCounterTrafficAgent(env)
agent.fit(20)                                   # warm-up
torch.cuda.synchronize()
t0 = time.perf_counter()
loss = agent.fit(STEPS)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("%d envs x %d steps with a GPU-resident DQN in the loop: %.3f s -> %.2f M env-steps/s (last loss %.4f)"
      % (N, STEPS, dt, N * STEPS / dt / 1e6, float(loss) if loss is not None else float("nan")))
env.check()
