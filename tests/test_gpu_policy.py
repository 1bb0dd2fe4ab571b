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
