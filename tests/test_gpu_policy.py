"""Fused policy MLP (csrc/policy_mlp.hip) vs the torch module it replaces.

Tolerances (stated here as the brief requires for floating-point kernels):
  * against the fp32 torch forward: |d action| <= 3e-2 -- weights and layer inputs are rounded to bf16 (2^-9 relative each),
    four layers of 180..192 terms with fp32 accumulation; the tanh output is in [-1, 1];
  * against a torch emulation of the kernel's contract (bf16-rounded weights and layer inputs, fp32 sums; the kernel also
    carries the biases of layers 2-4 in bf16, as torch.autocast does): <= 8e-3
    (only the summation order differs, which flips the bf16 rounding of near-tie hidden activations: one bf16 ulp, 2^-8
    relative, at a time; the maximum over 5e5 outputs was 4e-3);
  * integer data (everything exactly representable, sums < 2^24): the pre-activation path is exact, which pins the MFMA
    fragment layout and the permuted-k weight packing -- checked with an asymmetric weight matrix."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _emulate(module, x):
    import torch
    import torch.nn.functional as F
    bf = lambda t: t.to(torch.bfloat16).to(torch.float32)
    h = bf(x)
    for k, fc in enumerate((module.fc1, module.fc2, module.fc3, module.fc4)):
        h = F.linear(h, bf(fc.weight), fc.bias)
        if k < 3:
            h = bf(F.leaky_relu(h))
    return torch.tanh(h)


@pytest.mark.parametrize("rows", [1, 31, 32, 33, 1000, 131071, 131072 + 77, 4096 * 64])
def test_fused_policy_matches_torch(rows):
    import torch
    from marl_llm_amd.rollout import FusedPolicy, PolicyMLP
    torch.manual_seed(3)
    m = PolicyMLP(192, 2, 180).cuda()
    with torch.no_grad():                                   # weights large enough that the tanh is not just its linear part
        for fc in (m.fc1, m.fc2, m.fc3, m.fc4):
            fc.weight.mul_(2.0); fc.bias.uniform_(-0.3, 0.3)
    f = FusedPolicy(m)
    x = torch.randn(rows, 192, device="cuda") * 0.7
    with torch.no_grad():
        ref, emu = m(x), _emulate(m, x)
    got = f(x)
    assert got.shape == (rows, 2) and torch.isfinite(got).all()
    assert (got - emu).abs().max().item() <= 8e-3
    assert (got - ref).abs().max().item() <= 3e-2
    assert ref.abs().max().item() > 0.3                      # the comparison is not vacuous


def test_fragment_layout_with_exact_integer_data():
    """Small integers everywhere and positive pre-activations (leaky-ReLU = identity): every layer is exact in bf16 / fp32,
    so any error in the lane <-> (row, k) maps or in the permuted weight packing shows up as an O(1) difference.  The weight
    matrices are asymmetric (entry depends differently on row and column)."""
    import torch
    from marl_llm_amd.rollout import FusedPolicy, PolicyMLP
    m = PolicyMLP(192, 2, 180).cuda().double()
    g = torch.Generator().manual_seed(0)
    with torch.no_grad():
        for fc, scale in ((m.fc1, 1), (m.fc2, 1), (m.fc3, 1), (m.fc4, 1)):
            o, i = fc.weight.shape
            w = torch.zeros(o, i, dtype=torch.float64)
            rows_idx = torch.arange(o)
            for rep_ in range(3):                            # three ones per output row at asymmetric positions
                w[rows_idx, (7 * rows_idx + 13 * rep_ + rows_idx // 5) % i] += 1.0
            fc.weight.copy_(w); fc.bias.copy_(torch.randint(0, 3, (o,), generator=g).double())
    x = torch.randint(0, 4, (257, 192), generator=g).double().cuda()
    mf = PolicyMLP(192, 2, 180).cuda()
    mf.load_state_dict({k: v.float() for k, v in m.state_dict().items()})
    got = FusedPolicy(mf)(x.float().contiguous())
    with torch.no_grad():
        h = x
        for fc in (m.fc1, m.fc2, m.fc3):
            h = torch.nn.functional.linear(h, fc.weight, fc.bias)            # all values are small non-negative integers
        pre = torch.nn.functional.linear(h, m.fc4.weight, m.fc4.bias)
    assert pre.max().item() < 2 ** 8 and h.max().item() < 2 ** 8              # exactly representable in bf16
    assert torch.equal(got.double(), torch.tanh(pre.float()).double()) or (got.double() - torch.tanh(pre)).abs().max().item() < 1e-6


def test_narrow_observation_and_refresh():
    """in_dim = 188 (no self state, assembly.py:120) and a weight update through refresh()."""
    import torch
    from marl_llm_amd.rollout import FusedPolicy, PolicyMLP
    torch.manual_seed(5)
    m = PolicyMLP(188, 2, 180).cuda()
    f = FusedPolicy(m)
    x = torch.randn(777, 188, device="cuda")
    with torch.no_grad():
        assert (f(x) - _emulate(m, x)).abs().max().item() <= 8e-3
        m.fc4.weight.mul_(-3.0)
        f.refresh()
        assert (f(x) - _emulate(m, x)).abs().max().item() <= 8e-3


def test_rollout_with_fused_policy(shapes=None):
    import torch
    from marl_llm_amd.batched import SwarmBatch
    from marl_llm_amd.rollout import FusedPolicy, PolicyMLP, rollout
    from marl_llm_amd.shapes import r_avoid_for, synthetic_shape_set
    from marl_llm_amd.synth import synthetic_batch
    shapes = synthetic_shape_set()
    n_a, E = 64, 32
    sy = synthetic_batch(E, n_a, shapes, seed=2)
    sb = SwarmBatch(n_env=E, n_agents=n_a, n_cells_max=sy["cells"].shape[2], r_avoid=r_avoid_for(n_a, shapes), device="cuda:0")
    sb.set_cells(sy["cells"], sy["n_g"], sy["l_cell"]); sb.set_state(sy["p"], sy["dp"])
    obs = sb.observe()
    torch.manual_seed(0)
    f = FusedPolicy(PolicyMLP(192, 2, 180).cuda())
    obs2, rews = rollout(sb, f, 5, obs)
    assert obs2.shape == (E, n_a, 192) and torch.isfinite(obs2).all() and rews.shape == (5,)
    sb.close()


def test_bf16_observation_rows_give_identical_actions():
    """swarm_policy_forward_bf16 on bf16 rows == swarm_policy_forward on the same values held in float32 (the fp32 path
    rounds its input to bf16 first, so the MFMA operands are bit-identical)."""
    import torch
    from marl_llm_amd.rollout import FusedPolicy, PolicyMLP
    torch.manual_seed(11)
    f = FusedPolicy(PolicyMLP(192, 2, 180).cuda())
    xb = (torch.randn(5000, 192, device="cuda") * 0.8).to(torch.bfloat16).contiguous()
    assert torch.equal(f(xb), f(xb.float().contiguous()))
    with pytest.raises(RuntimeError):                     # 188-wide bf16 rows are not 16-byte aligned
        FusedPolicy(PolicyMLP(188, 2, 180).cuda())(torch.zeros(4, 188, device="cuda", dtype=torch.bfloat16))


def test_bf16_env_output_and_rollout():
    """obs_dtype=bfloat16: the env's obs / a_prior are the float32 outputs rounded to nearest-even bf16, bit for bit; the
    bf16 rows feed the fused policy directly in a device-resident rollout."""
    import torch
    from marl_llm_amd.batched import SwarmBatch
    from marl_llm_amd.rollout import DeviceReplay, FusedPolicy, PolicyMLP, rollout
    from marl_llm_amd.shapes import r_avoid_for, synthetic_shape_set
    from marl_llm_amd.synth import synthetic_batch
    shapes = synthetic_shape_set()
    n_a, E = 64, 24
    sy = synthetic_batch(E, n_a, shapes, seed=4, assembled_fraction=0.5)
    outs = {}
    for dt in (torch.float32, torch.bfloat16):
        sb = SwarmBatch(n_env=E, n_agents=n_a, n_cells_max=sy["cells"].shape[2], r_avoid=r_avoid_for(n_a, shapes),
                        obs_dtype=dt, device="cuda:0")
        sb.set_cells(sy["cells"], sy["n_g"], sy["l_cell"]); sb.set_state(sy["p"], sy["dp"])
        o0 = sb.observe().clone()
        act = torch.zeros((E, n_a, 2), device="cuda")
        for _ in range(3):
            o, r, d, pri = sb.step(act)
            act = pri.float()
        outs[dt] = (o0, o.clone(), pri.clone(), r.clone(), sb)
    f32, b16 = outs[torch.float32], outs[torch.bfloat16]
    assert b16[0].dtype == torch.bfloat16 and b16[2].dtype == torch.bfloat16
    assert torch.equal(b16[0], f32[0].to(torch.bfloat16))
    # the trajectories stay identical only while the fed-back prior is identical: compare the first step's outputs
    f32[4].set_state(sy["p"], sy["dp"]); f32[4].observe()
    b16[4].set_state(sy["p"], sy["dp"]); b16[4].observe()
    z = torch.zeros((E, n_a, 2), device="cuda")
    of, rf, _, pf = f32[4].step(z)
    ob, rb, _, pb = b16[4].step(z)
    assert torch.equal(ob, of.to(torch.bfloat16)) and torch.equal(pb, pf.to(torch.bfloat16)) and torch.equal(rb, rf)
    sb = b16[4]
    torch.manual_seed(1)
    pol = FusedPolicy(PolicyMLP(192, 2, 180).cuda())
    rep = DeviceReplay(4 * E * n_a, 192, 2, "cuda", obs_dtype=torch.bfloat16)
    obs2, rews = rollout(sb, pol, 4, ob, replay=rep, noise_scale=0.1)
    assert obs2.dtype == torch.bfloat16 and len(rep) == 4 * E * n_a and torch.isfinite(rews).all()
    for x in outs.values():
        x[4].close()


def test_in_kernel_exploration_noise_is_standard_normal():
    """swarm_policy_forward_explore adds noise_scale * N(0, 1) in the kernel's epilogue (agents.py:93-96) from a counter-based
    generator keyed by (seed, step, row): moments against the normal distribution (and against torch.randn's on the same
    sample size), reproducibility per key, independence between keys and between the two action components."""
    import torch
    from marl_llm_amd.rollout import FusedPolicy, PolicyMLP
    m = PolicyMLP(192, 2, 180).cuda()
    with torch.no_grad():
        for p in m.parameters():
            p.zero_()                                   # actor output tanh(0) = 0: the action IS the clamped noise
    f = FusedPolicy(m)
    rows, scale = 1 << 18, 0.2                          # |scale * z| < 1 up to 5 sigma: the clamp never bites
    x = torch.randn(rows, 192, device="cuda")
    a = f(x, noise_scale=scale, seed=7, step=3)
    b = f(x, noise_scale=scale, seed=7, step=3)
    c = f(x, noise_scale=scale, seed=7, step=4)
    d = f(x, noise_scale=scale, seed=8, step=3)
    assert torch.equal(a, b) and not torch.equal(a, c) and not torch.equal(a, d)
    assert torch.equal(f(x), torch.zeros_like(a))       # noise_scale = 0: the plain forward
    z = (a / scale).double()
    t = torch.randn(rows, 2, device="cuda", dtype=torch.float64)
    n = z.numel()

    def moments(v):
        v = v.flatten()
        mu = v.mean(); s = v.std()
        return mu.item(), s.item(), (((v - mu) / s) ** 3).mean().item(), (((v - mu) / s) ** 4).mean().item()
    for got in (moments(z), moments(z[:, 0]), moments(z[:, 1])):
        assert abs(got[0]) < 5 / n ** 0.5 * 1.5 and abs(got[1] - 1) < 5 / n ** 0.5 * 1.5
        assert abs(got[2]) < 0.03 and abs(got[3] - 3) < 0.06
    ref = moments(t)
    assert abs(ref[2]) < 0.03 and abs(ref[3] - 3) < 0.06                      # the same bars hold for torch.randn
    corr = lambda u, v: ((u - u.mean()) * (v - v.mean())).mean().item() / (u.std() * v.std()).item()
    assert abs(corr(z[:, 0], z[:, 1])) < 0.01                                  # the two components of a row
    assert abs(corr(z.flatten(), (c / scale).double().flatten())) < 0.01       # consecutive steps
    assert abs(corr(z[:-1, 0], z[1:, 0])) < 0.01                               # neighbouring rows
    # tails: P(|z| > 3) = 0.0027
    assert abs((z.abs() > 3).double().mean().item() - 0.0027) < 0.0005
    # clamp: a large scale saturates at +-1
    big = f(x, noise_scale=5.0, seed=1, step=1)
    assert big.abs().max().item() == 1.0


def test_fused_rollout_writes_the_ring_in_place(shapes=None):
    """rollout's fused path (FusedPolicy + ChainedReplay + SwarmBatch): the policy kernel writes the action into the ring
    slot, the env step writes next_obs / reward / done / prior there -- no push copies.  With the noise off it must hold
    exactly the transitions of the push-based path."""
    import torch
    from marl_llm_amd.batched import SwarmBatch
    from marl_llm_amd.rollout import ChainedReplay, FusedPolicy, PolicyMLP, rollout
    from marl_llm_amd.shapes import r_avoid_for, synthetic_shape_set
    shapes = synthetic_shape_set()
    E, N, K = 12, 32, 3
    n = E * N
    ng_max = max(np.asarray(g).shape[0] for g in shapes["grid_coords"])
    torch.manual_seed(0)
    pol = FusedPolicy(PolicyMLP(192, 2, 180).cuda())
    rings = []
    for fused in (True, False):
        sb = SwarmBatch(n_env=E, n_agents=N, n_cells_max=ng_max, r_avoid=r_avoid_for(N, shapes)); sb.set_shapes(shapes)
        obs = sb.reset(seed=5)
        ring = ChainedReplay(K, n, sb.obs_dim, 2, sb.device)
        if fused:
            obs, rews = rollout(sb, pol, 5, obs, replay=ring)
        else:
            class Push:                                   # same ring, but through push(): rollout takes the generic path
                def push(self, *a):
                    ring.push(*a)
            obs, rews = rollout(sb, pol, 5, obs, replay=Push())
        rings.append((ring, obs.clone(), rews.clone()))
        sb.close()
    (a, oa, ra), (b, ob, rb) = rings
    assert a.cur == b.cur and a.count == b.count == K and torch.equal(oa, ob) and torch.equal(ra, rb)
    for name in ("obs", "act", "rew", "done", "act_prior"):
        assert torch.equal(getattr(a, name), getattr(b, name)), name
    assert torch.equal(a.obs[a.cur], oa.reshape(n, -1))
    # with noise: actions differ from the plain forward but stay in [-1, 1]; the same (seed, step0) reproduces the rollout
    outs = []
    for rep in range(2):
        sb = SwarmBatch(n_env=E, n_agents=N, n_cells_max=ng_max, r_avoid=r_avoid_for(N, shapes)); sb.set_shapes(shapes)
        obs = sb.reset(seed=5)
        ring = ChainedReplay(K, n, sb.obs_dim, 2, sb.device)
        obs, _ = rollout(sb, pol, 4, obs, replay=ring, noise_scale=0.1, seed=11, step0=100, track_reward=False)
        outs.append((ring.act.clone(), obs.clone()))
        sb.close()
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert outs[0][0].abs().max() <= 1 and not torch.equal(outs[0][0], a.act)
