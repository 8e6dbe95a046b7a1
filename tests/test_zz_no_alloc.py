"""The entry point that allocates nothing (mad_gn_coef_static) against the two-call form it is a variant of
(mad_gn_partial + mad_gn_finalize with a caller-owned slab), through the C ABI.  Collected last on purpose:
its name sorts after every other test file."""
import pytest
import torch

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


def _case(dev, B, H, C1, C2, dt, with_ada, seed):
    from mad_amd.unet_spec import gn_groups
    g = torch.Generator().manual_seed(seed)
    x1 = (torch.randn(B, H, H, C1, generator=g) * 1.7 + 0.3).to(dt).to(dev)
    x2 = (torch.randn(B, H, H, C2, generator=g) * 0.6 - 0.2).to(dt).to(dev) if C2 else None
    C = C1 + C2
    gamma = (1 + 0.2 * torch.randn(C, generator=g)).to(dev)
    beta = (0.1 * torch.randn(C, generator=g)).to(dev)
    ada = (0.3 * torch.randn(B, 2 * C + 5, generator=g)).to(dev) if with_ada else None
    return x1, x2, gamma, beta, ada, (2 * C + 5 if with_ada else 0), gn_groups(C)


def _both(dev, *case_args):
    from mad_amd import ops
    x1, x2, gamma, beta, ada, stride, groups = _case(dev, *case_args)
    B, H = x1.shape[0], x1.shape[1]
    want = ops.gn_finalize(ops.gn_partial(x1, x2), H * H, gamma, beta, groups, 1e-5, ada=ada, ada_stride=stride)
    out = torch.full_like(want, float("nan"))
    torch.cuda.synchronize()
    before = torch.cuda.memory_stats()["allocation.all.allocated"]
    got = ops.gn_coef_static(x1, gamma, beta, groups, 1e-5, x2=x2, ada=ada, ada_stride=stride, out=out)
    assert torch.cuda.memory_stats()["allocation.all.allocated"] == before      # the wrapper allocated nothing
    torch.cuda.synchronize()
    assert got.data_ptr() == out.data_ptr()
    return got, want


def test_static_slab_matches_the_two_call_form_twice_in_a_row():
    """Two calls in a row, the second on a smaller input whose partial sums cover only the head of the slab
    the first call filled: both bit-identical to the caller-owned-slab form."""
    dev = torch.device("cuda:0")
    got, want = _both(dev, 4, 64, 128, 0, torch.bfloat16, True, 1)          # 4 x 64 splits x 128 channels
    assert torch.isfinite(want).all() and torch.equal(got, want)
    got, want = _both(dev, 2, 16, 32, 32, torch.bfloat16, False, 2)         # 2 x 4 splits x 64 channels, concat
    assert torch.isfinite(want).all() and torch.equal(got, want)
    got, want = _both(dev, 3, 8, 48, 0, torch.float32, True, 3)             # fp32, one split, 16-channel... groups
    assert torch.isfinite(want).all() and torch.equal(got, want)


def test_static_slab_rejects_what_does_not_fit():
    from mad_amd import _lib, ops
    dev = torch.device("cuda:0")
    limit = _lib.lib.mad_gn_coef_static_max_floats()
    B, H, C = 33, 64, 512                                                    # 33 x 64 splits x 512 x 2 > 2^21
    assert B * 64 * C * 2 > limit
    x = torch.zeros(B, H, H, C, dtype=torch.bfloat16, device=dev)
    with pytest.raises(RuntimeError, match="static slab"):
        ops.gn_coef_static(x, torch.ones(C, device=dev), torch.zeros(C, device=dev), 32, 1e-5)
    # the largest shape that fits still matches
    x = torch.randn(32, H, H, C, device=dev).to(torch.bfloat16)
    gamma, beta = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    want = ops.gn_finalize(ops.gn_partial(x), H * H, gamma, beta, 32, 1e-5)
    assert torch.equal(ops.gn_coef_static(x, gamma, beta, 32, 1e-5), want)
