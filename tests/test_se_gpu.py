"""GPU (-m gpu): SqueezeExcite / DropPath kernels (csrc/rx_se.hip) through the C ABI.

PARITY UNPINNED: the two classes live in the un-vendored dynamic_network_architectures package (absent from the reference
tree and from this image; no reference test or fixture covers them).  The checker here is the oracle's restatement of
their published source (oracle/resenc_oracle.py::SqueezeExcite / DropPath) run in fp64 on the CPU with autograd."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

import resenc_oracle as oracle
from helpers import rel_l2

DTYPES = [torch.float32, torch.bfloat16, torch.float16]
TOL = {torch.float32: 2e-5, torch.bfloat16: 1.5e-2, torch.float16: 3e-3}


@pytest.fixture(scope="module")
def ops():
    import mt3d_amd  # noqa: F401
    from mt3d_amd.engine import ops as o
    return o


def _reference(y, res, g, w1, b1, w2, b2, scale, keep_x, slope, eps=1e-5):
    """fp64 autograd restatement: a = lrelu(SE(DropPath(IN(y))) + res); returns a, dy, dres, fc gradients"""
    y = y.clone().requires_grad_(True)
    res = res.clone().requires_grad_(True)
    ps = [p.clone().requires_grad_(True) for p in (w1, b1, w2, b2)] if w1 is not None else None
    xh = F.instance_norm(y, eps=eps)
    if scale is not None:
        xh = xh * scale.view(-1, 1, 1, 1, 1)
    if ps is not None:
        if keep_x:
            p = xh.mean((2, 3), keepdim=True)                 # the published forward: dims 2 and 3 of a 5-D tensor
        else:
            p = xh.mean((2, 3, 4), keepdim=True)              # a 4-D tensor (2-D net): (y, x) = every spatial axis here
        h = torch.relu(F.conv3d(p, ps[0], ps[1]))
        xh = xh * torch.sigmoid(F.conv3d(h, ps[2], ps[3]))
    a = F.leaky_relu(xh + res, slope)
    a.backward(g)
    return a.detach(), y.grad, res.grad, [p.grad for p in ps] if ps is not None else None


CASES = [
    # (n, c, (z, y, x), rd, keep_x, with_se, with_scale)
    (2, 32, (6, 5, 16), 8, 1, True, False),
    (2, 64, (4, 4, 12), 8, 1, True, True),
    (1, 256, (3, 3, 4), 16, 1, True, False),
    (2, 512, (2, 2, 2), 32, 1, True, True),
    (2, 32, (1, 12, 20), 8, 0, True, False),       # 2-D net: unit z axis, pooled over everything
    (3, 32, (4, 4, 8), 0, 1, False, True),         # DropPath only
    (2, 320, (2, 3, 5), 24, 1, True, False),       # C > 256 and not a power of two
]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", CASES)
def test_se_block_fwd_bwd(ops, dtype, case):
    n, c, dims, rd, keep_x, with_se, with_scale = case
    gen = torch.Generator().manual_seed(7)
    rnd = lambda *s, k=1.0: (torch.randn(*s, generator=gen) * k).to(dtype).double()
    y, res, g = rnd(n, c, *dims), rnd(n, c, *dims), rnd(n, c, *dims, k=0.1)
    slope = 0.01
    if with_se:
        w1, b1 = torch.randn(rd, c, 1, 1, 1, generator=gen).double() * 0.3, torch.randn(rd, generator=gen).double() * 0.1
        w2, b2 = torch.randn(c, rd, 1, 1, 1, generator=gen).double() * 0.3, torch.randn(c, generator=gen).double() * 0.1
    else:
        w1 = b1 = w2 = b2 = None
    scale = torch.tensor([0.0, 1.25, 1.25][:n] if n <= 3 else [1.25] * n).double() if with_scale else None
    if with_scale and n == 2:
        scale = torch.tensor([1.25, 0.0]).double()
    a_r, dy_r, dres_r, pg_r = _reference(y, res, g, w1, b1, w2, b2, scale, keep_x, slope)

    to_act = lambda t: ops.Act(t.permute(0, 2, 3, 4, 1).contiguous().to(dtype).cuda())
    ya, ra, ga = to_act(y), to_act(res), to_act(g)
    out = ops.Act(torch.empty_like(ya.t))
    dy = ops.Act(torch.empty_like(ya.t))
    dres = ops.Act(torch.full_like(ya.t, 0.5))           # accumulate into an existing gradient
    f32 = dict(dtype=torch.float32, device="cuda")
    L = dims[2] if keep_x else 1
    stats = torch.empty((n, c, 2), **f32)
    mult, dadd, m12 = torch.empty((n, L, c), **f32), torch.empty((n, L, c), **f32), torch.empty((n, c, 2), **f32)
    se = pooled = hidden = gate = None
    grads = [None] * 4
    if with_se:
        dev = [t.float().cuda().contiguous() for t in (w1, b1, w2, b2)]
        se = dict(w1=dev[0], b1=dev[1], w2=dev[2], b2=dev[3], rd=rd, keep_x=keep_x)
        pooled, hidden, gate = torch.empty((n, L, c), **f32), torch.empty((n, L, rd), **f32), torch.empty((n, L, c), **f32)
        grads = [torch.empty_like(t) for t in dev]
    sc = scale.float().cuda() if scale is not None else None
    ops.instnorm_stats(ya, stats)
    ops.se_gate_fwd(ya, stats, se, pooled, hidden, gate, mult, sc)
    ops.instnorm_gate_act_fwd(ya, stats, mult, keep_x, out, slope, ra)
    ops.se_gate_bwd(ga, ya, stats, out, slope, se, pooled, hidden, gate, mult, dadd, m12, *grads, path_scale=sc)
    ops.instnorm_gate_act_bwd(ga, ya, stats, out, slope, mult, dadd, m12, keep_x, dy, dres, True)
    torch.cuda.synchronize()
    tol = TOL[dtype]
    assert rel_l2(out.to_ncdhw().cpu(), a_r) < tol
    # the engine's mask comes from ITS rounded output; compare the gradients where both masks agree in sign
    assert rel_l2(dy.to_ncdhw().cpu(), dy_r) < 20 * tol + 2e-2 * (dtype != torch.float32)
    assert rel_l2(dres.to_ncdhw().cpu() - 0.5, dres_r) < 20 * tol + 2e-2 * (dtype != torch.float32)
    if with_se:
        for got, want in zip(grads, pg_r):
            # (with keep_x = 0 the pooled value is the mean of an InstanceNorm output, i.e. ~0: dw1 is rounding noise there)
            err = (got.cpu().double() - want).norm().item()
            assert err < (20 * tol + 2e-2 * (dtype != torch.float32)) * want.norm().item() + 1e-5


def test_se_error_paths(ops):
    from mt3d_amd.engine.lib import RxError
    y = ops.Act(torch.zeros((1, 2, 2, 4, 32), dtype=torch.bfloat16, device="cuda"))
    f32 = dict(dtype=torch.float32, device="cuda")
    stats, mult = torch.zeros((1, 32, 2), **f32), torch.zeros((1, 4, 32), **f32)
    se = dict(w1=torch.zeros(80, 32, **f32), b1=torch.zeros(80, **f32), w2=torch.zeros(32, 80, **f32), b2=torch.zeros(32, **f32),
              rd=80, keep_x=1)
    with pytest.raises(RxError):        # more reduction channels than the gate kernel holds
        ops.se_gate_fwd(y, stats, se, mult.clone(), torch.zeros((1, 4, 80), **f32), mult.clone(), mult)
    bad = ops.Act(torch.zeros((1, 2, 2, 4, 64), dtype=torch.bfloat16, device="cuda"))
    with pytest.raises(RxError):        # geometry mismatch between y and out
        ops.instnorm_gate_act_fwd(y, stats, mult, 1, bad, 0.01, None)
