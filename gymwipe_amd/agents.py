"""
The caller side of the hot path: what the reference's example agent
(agents/dqn_counter_traffic.py) needs between a flat-action DQN and the env, vectorised and kept on
the GPU so that ``policy -> env.step -> replay`` involves no host round trip.

``CounterTrafficProcessor``  the keras-rl ``Processor`` of the reference (:23-33): flat action ->
                             ``{"device", "duration"}``; Python ints in, ints out (same arithmetic),
                             or an int tensor in, int32 tensors out (used in place by the env).
``DqnCounterTrafficAgent``   a small torch DQN with the reference example's shape (:47-63):
                             Dense 16-16-16 with ReLU on the 1-d observation, Boltzmann policy, Adam
                             1e-3, soft target update 1e-2, sequential replay memory -- over N envs.
                             It exists to drive the env from a GPU-resident policy, not to be a
                             tuned learner (the reference's own reward shaping is "most likely far
                             away from being perfect", counter_traffic.py:86-92).
"""
from .envs.core import BaseEnv


class CounterTrafficProcessor:
    """agents/dqn_counter_traffic.py:23-33."""

    def __init__(self, max_duration=BaseEnv.MAX_ASSIGN_DURATION):
        self.max_duration = int(max_duration)

    def process_action(self, flat_action):
        assert flat_action is not None
        md = self.max_duration
        try:
            import torch
            is_tensor = isinstance(flat_action, torch.Tensor)
        except ImportError:                                   # pragma: no cover
            is_tensor = False
        if is_tensor:
            flat = flat_action.to(torch.int32)
            device = torch.div(flat, md, rounding_mode="trunc")          # int(flat_action / max_duration)
            return {"device": device, "duration": flat - device * md}
        device = int(flat_action / md)
        return {"device": device, "duration": flat_action - (device * md)}


class DqnCounterTrafficAgent:
    def __init__(self, env, hidden=16, lr=1e-3, gamma=0.99, tau=1.0, target_update=1e-2, memory_limit=50000,
                 batch_size=32, warmup_steps=1000, seed=123):
        import torch
        from torch import nn
        self.torch = torch
        self.env = env
        self.n = env.num_envs
        self.dev = env.device
        self.processor = CounterTrafficProcessor(env.MAX_ASSIGN_DURATION)
        self.nb_devices = env.action_space.spaces["device"].n
        self.nb_durations = env.action_space.spaces["duration"].n
        self.nb_actions = self.nb_devices * self.nb_durations      # agents/dqn_counter_traffic.py:41-44
        torch.manual_seed(seed)

        def net():
            return nn.Sequential(nn.Linear(1, hidden), nn.ReLU(), nn.Linear(hidden, hidden), nn.ReLU(),
                                 nn.Linear(hidden, hidden), nn.ReLU(), nn.Linear(hidden, self.nb_actions)).to(self.dev)
        self.q, self.q_target = net(), net()
        self.q_target.load_state_dict(self.q.state_dict())
        self.opt = torch.optim.Adam(self.q.parameters(), lr=lr)
        self.gamma, self.tau, self.target_update = gamma, tau, target_update
        self.batch_size, self.warmup = batch_size, warmup_steps
        # sequential memory of the last `memory_limit` transitions, on the GPU (rl.memory.SequentialMemory)
        cap = max(memory_limit, self.n)
        self.cap = cap // self.n * self.n
        self.m_obs = torch.zeros(self.cap, device=self.dev)
        self.m_next = torch.zeros(self.cap, device=self.dev)
        self.m_act = torch.zeros(self.cap, dtype=torch.int64, device=self.dev)
        self.m_rew = torch.zeros(self.cap, device=self.dev)
        self.m_done = torch.zeros(self.cap, device=self.dev)
        self.m_pos, self.m_len, self.steps = 0, 0, 0
        self.center = float(env.COUNTER_BOUND)

    def _features(self, obs):
        return (obs.to(self.torch.float32) - self.center).unsqueeze(-1)   # the observation is centred on COUNTER_BOUND

    def act(self, obs):
        """Boltzmann policy over Q (rl.policy.BoltzmannQPolicy: exp(clip(q / tau, -500, 500)))."""
        torch = self.torch
        with torch.no_grad():
            qv = self.q(self._features(obs))
            p = torch.softmax(torch.clamp(qv / self.tau, -500.0, 500.0), dim=-1)
            return torch.multinomial(p, 1).squeeze(-1)

    def remember(self, obs, act, rew, nxt, done):
        i = self.m_pos
        sl = slice(i, i + self.n)
        self.m_obs[sl] = obs.to(self.torch.float32)
        self.m_next[sl] = nxt.to(self.torch.float32)
        self.m_act[sl] = act
        self.m_rew[sl] = rew
        self.m_done[sl] = done.to(self.torch.float32)
        self.m_pos = (i + self.n) % self.cap
        self.m_len = min(self.m_len + self.n, self.cap)

    def learn(self):
        torch = self.torch
        idx = torch.randint(0, self.m_len, (self.batch_size,), device=self.dev)
        q = self.q(self._features(self.m_obs[idx])).gather(1, self.m_act[idx].unsqueeze(1)).squeeze(1)
        with torch.no_grad():
            target = self.m_rew[idx] + self.gamma * (1.0 - self.m_done[idx]) * self.q_target(self._features(self.m_next[idx])).max(dim=1).values
        loss = torch.nn.functional.mse_loss(q, target)
        self.opt.zero_grad(set_to_none=True)
        loss.backward()
        self.opt.step()
        with torch.no_grad():                                  # soft target update (target_model_update = 1e-2)
            for pt, p in zip(self.q_target.parameters(), self.q.parameters()):
                pt.mul_(1.0 - self.target_update).add_(p, alpha=self.target_update)
        return loss

    def fit(self, nb_steps, reset_every=64):
        """Vectorised dqn.fit(): nb_steps env.step() calls of all N envs; returns the last loss (tensor)."""
        obs = self.env.reset().clone()
        loss = None
        for k in range(nb_steps):
            if k and reset_every and k % reset_every == 0:
                obs = self.env.reset().clone()
            flat = self.act(obs)
            o, r, d, _ = self.env.step(self.processor.process_action(flat))
            self.remember(obs, flat, r, o, d)
            obs = o.clone()
            self.steps += 1
            if self.steps * self.n >= self.warmup and self.m_len >= self.batch_size:
                loss = self.learn()
        return loss
