"""Host-side logic of marl_llm_amd.rollout that needs no GPU: the epsilon coin accepts every numpy generator flavour
(np.random module, RandomState, Generator), the generic (non-fused) path pushes what it stepped."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")


class StubEnv:
    def __init__(self, E, N, D):
        self.E, self.N, self.D, self.t = E, N, D, 0
        self.actions = []

    def step_tensor(self, act):
        self.actions.append(act.clone())
        self.t += 1
        obs = torch.full((self.E, self.N, self.D), float(self.t))
        return obs, torch.ones(self.E, self.N), torch.zeros(self.E, self.N, dtype=torch.uint8), torch.zeros(self.E, self.N, 2)


class Pushes:
    def __init__(self):
        self.n = 0

    def push(self, obs, act, rew, next_obs, done, prior):
        assert obs.shape == next_obs.shape and act.shape[-1] == 2
        self.n += 1


@pytest.mark.parametrize("rng", [None, np.random.RandomState(0), np.random.default_rng(0)])
def test_rollout_epsilon_coin_accepts_any_numpy_generator(rng):
    from marl_llm_amd.rollout import rollout
    E, N, D = 2, 3, 8
    env = StubEnv(E, N, D)
    policy = lambda x: torch.zeros(x.shape[0], 2)
    rep = Pushes()
    obs, rews = rollout(env, policy, 6, torch.zeros(E, N, D), replay=rep, epsilon=1.0, host_rng=rng)   # coin < 1 always: uniform actions
    assert rep.n == 6 and rews.shape == (6,) and torch.equal(rews, torch.ones(6))
    assert all(a.abs().max() <= 1 and a.abs().sum() > 0 for a in env.actions)
    assert torch.equal(obs, torch.full((E, N, D), 6.0))
    env2 = StubEnv(E, N, D)
    obs, rews = rollout(env2, policy, 3, torch.zeros(E, N, D), epsilon=0.0, noise_scale=0.0, track_reward=False)
    assert rews is None and all(a.abs().sum() == 0 for a in env2.actions)
