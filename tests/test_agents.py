"""The caller-side adapter: flat-action processor parity with the reference's
arithmetic (agents/dqn_counter_traffic.py:23-33) and a GPU-resident policy loop."""
import pytest


def test_processor_matches_reference_arithmetic():
    import torch
    from gymwipe_amd.agents import CounterTrafficProcessor
    p = CounterTrafficProcessor()
    assert p.max_duration == 20                                 # CounterTrafficEnv.MAX_ASSIGN_DURATION
    for flat in range(0, 16 * 20):
        device = int(flat / 20)                                 # the reference's two lines, verbatim arithmetic
        duration = flat - (device * 20)
        assert p.process_action(flat) == {"device": device, "duration": duration}
        assert 0 <= duration < 20
    t = torch.arange(0, 320)
    out = p.process_action(t)
    assert out["device"].dtype == torch.int32 and out["duration"].dtype == torch.int32
    assert out["device"].tolist() == [int(f / 20) for f in range(320)]
    assert out["duration"].tolist() == [f - int(f / 20) * 20 for f in range(320)]
    with pytest.raises(AssertionError):
        p.process_action(None)


@pytest.mark.gpu
def test_gpu_resident_dqn_loop_runs():
    import torch
    import gymwipe_amd
    from gymwipe_amd.agents import DqnCounterTrafficAgent
    env = gymwipe_amd.make("VecCounterTraffic-v0", num_envs=512, num_devices=2)
    agent = DqnCounterTrafficAgent(env, warmup_steps=512, memory_limit=8192)
    assert agent.nb_actions == 2 * 20
    loss = agent.fit(24)
    assert loss is not None and torch.isfinite(loss).item()
    st = env.check()                                           # every sampled action was inside the action space
    assert st["steps"] == 24 * 512 and st["bad_actions"] == 0
    assert agent.m_len == min(24 * 512, agent.cap)


@pytest.mark.gpu
def test_quickstart_example_runs():
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "examples", "quickstart.py")], capture_output=True, text=True,
                         timeout=600, cwd=root)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-1500:]
    assert "scalar env: (65538, -2.0, False" in out.stdout and "(65536, 2.0, False" in out.stdout   # the reference's known answer
    for word in ("vectorised:", "rollout:", "custom interpreter:", "pendulum:", "control loop:", "grid:"):
        assert word in out.stdout
