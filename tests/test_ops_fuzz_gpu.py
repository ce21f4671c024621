"""GPU (-m gpu): randomized shapes through the conv / transposed-conv / InstanceNorm entry points against torch CPU fp64 --
channel counts, extents (ragged against every tile size: 4x4x16 halo tiles, 128 / 256-row GEMM tiles, 16-voxel K steps), per-axis
kernels and strides, batch, bias, accumulation, strided input views.  The fixed cases of tests/test_ops_gpu.py pin the kernels the
comments name; these draws walk the dispatch table between them (RX_FUZZ_SEED picks other draws)."""
import os
import random

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from test_ops_gpu import TOL, last_kernel, out_dim, rel, rnd, to_act


@pytest.fixture(scope="module")
def ops():
    import mt3d_amd  # noqa: F401
    from mt3d_amd.engine import lib, ops as _ops
    lib.require_device()
    return _ops


def conv_draws(n=48):
    rng = random.Random(int(os.environ.get("RX_FUZZ_SEED", "31337")))
    out = []
    while len(out) < n:
        ci, co = rng.choice([32, 32, 64, 64, 96, 128, 256, 512]), rng.choice([32, 32, 64, 64, 96, 128, 256, 512])
        k = tuple(rng.choice([1, 3, 3]) for _ in range(3))
        s = tuple(rng.choice([1, 1, 2]) for _ in range(3))
        big = rng.random() < 0.4
        dims = tuple(rng.choice([4, 6, 8, 12, 16, 20, 24, 32, 36, 48, 64] if big else [1, 2, 3, 4, 5, 7, 8, 9, 16, 17]) for _ in range(3))
        nb = rng.choice([1, 2, 3])
        od = tuple(out_dim(d, kk, ss) for d, kk, ss in zip(dims, k, s))
        if min(od) < 1:
            continue
        macs = nb * od[0] * od[1] * od[2] * ci * co * k[0] * k[1] * k[2]
        if macs > 6e9 or nb * dims[0] * dims[1] * dims[2] * max(ci, co) > 6e7:
            continue
        out.append((ci, co, dims, k, s, nb, rng.random() < 0.5, rng.choice([0, 32])))
    return out


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32, torch.float16])
@pytest.mark.parametrize("i", range(48))
def test_random_conv3d(ops, dtype, i):
    ci, co, dims, k, s, n, with_bias, c0 = conv_draws()[i]
    if dtype != torch.bfloat16 and i % 3:          # every draw in bf16, a third of them in the other two types
        pytest.skip("subset")
    x = rnd((n, ci, *dims), dtype, 1 + i)
    w = rnd((co, ci, *k), dtype, 2 + i, scale=(ci * k[0] * k[1] * k[2]) ** -0.5)
    b = rnd((co,), torch.float32, 3 + i) if with_bias else None
    pad = [(kk - 1) // 2 for kk in k]
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    y_ref = F.conv3d(xr, wr, b, stride=s, padding=pad)
    odims = tuple(y_ref.shape[2:])
    gy = rnd(tuple(y_ref.shape), dtype, 4 + i)
    y_ref.backward(gy)
    xa = to_act(ops, x, dtype, ld=ci + c0, c0=c0)
    w_fwd, w_bwd = ops.pack_conv_weight(w.float().cuda(), dtype)
    ya = ops.Act.zeros(n, *odims, co, dtype)
    ops.conv3d_fwd(xa, w_fwd, b.float().cuda() if with_bias else None, ya, k, s)
    kf = last_kernel(ops)
    assert rel(ya.to_ncdhw(), y_ref.detach()) < TOL[dtype], ("fwd", kf, ci, co, dims, k, s, n)
    gya = to_act(ops, gy, dtype)
    base = rnd((n, ci, *dims), dtype, 5 + i, scale=0.5)
    dxa = to_act(ops, base, dtype)
    ops.conv3d_bwd_data(gya, w_bwd, dxa, k, s, accumulate=True)
    kd = last_kernel(ops)
    assert rel(dxa.to_ncdhw(), base + xr.grad) < 2 * TOL[dtype], ("dgrad+", kd, ci, co, dims, k, s, n)
    ops.conv3d_bwd_data(gya, w_bwd, dxa, k, s, accumulate=False)
    assert rel(dxa.to_ncdhw(), xr.grad) < TOL[dtype], ("dgrad", kd, ci, co, dims, k, s, n)
    dw = torch.full((co, ci, *k), float("nan"), dtype=torch.float32, device="cuda")
    ops.conv3d_bwd_weight(xa, gya, dw, k, s)
    kw = last_kernel(ops)
    assert rel(dw, wr.grad) < TOL[dtype], ("wgrad", kw, ci, co, dims, k, s, n)
    if with_bias:
        db = torch.full((co,), float("nan"), device="cuda")
        ops.channel_sum(gya, db)
        assert rel(db, gy.sum((0, 2, 3, 4))) < 1e-4


def convT_draws(n=16):
    rng = random.Random(int(os.environ.get("RX_FUZZ_SEED", "31337")) + 1)
    out = []
    while len(out) < n:
        ci, co = rng.choice([32, 64, 96, 128, 256, 512]), rng.choice([32, 64, 96, 128, 256])
        s = tuple(rng.choice([1, 2, 2]) for _ in range(3))
        dims = tuple(rng.choice([1, 2, 3, 4, 5, 8, 12, 16, 24]) for _ in range(3))
        nb = rng.choice([1, 2, 3])
        if nb * dims[0] * dims[1] * dims[2] * s[0] * s[1] * s[2] * ci * co > 3e9:
            continue
        out.append((ci, co, dims, s, nb, rng.random() < 0.5))
    return out


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("i", range(16))
def test_random_convT3d(ops, dtype, i):
    ci, co, dims, s, n, with_bias = convT_draws()[i]
    x = rnd((n, ci, *dims), dtype, 11 + i)
    w = rnd((ci, co, *s), dtype, 12 + i, scale=ci ** -0.5)
    b = rnd((co,), torch.float32, 13 + i) if with_bias else None
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    y_ref = F.conv_transpose3d(xr, wr, b, stride=s)
    gy = rnd(tuple(y_ref.shape), dtype, 14 + i)
    y_ref.backward(gy)
    od = tuple(y_ref.shape[2:])
    xa = to_act(ops, x, dtype)
    w_fwd, w_bwd = ops.pack_convT_weight(w.float().cuda(), dtype)
    ya = to_act(ops, torch.zeros((n, 2 * co, *od), dtype=torch.float64), dtype)        # written into the first half of a concat
    up = ops.Act(ya.t, 0, co)
    ops.convT3d_fwd(xa, w_fwd, b.float().cuda() if with_bias else None, up, s)
    assert rel(up.to_ncdhw(), y_ref.detach()) < TOL[dtype], ("fwd", ci, co, dims, s, n)
    assert (ya.t[..., co:] == 0).all()                                                  # the other half is untouched
    gya = to_act(ops, gy, dtype)
    dxa = ops.Act.zeros(n, *dims, ci, dtype)
    ops.convT3d_bwd_data(gya, w_bwd, dxa, s)
    assert rel(dxa.to_ncdhw(), xr.grad) < TOL[dtype], ("dgrad", ci, co, dims, s, n)
    dw = torch.full(tuple(w.shape), float("nan"), dtype=torch.float32, device="cuda")
    ops.convT3d_bwd_weight(xa, gya, dw, s)
    assert rel(dw, wr.grad) < TOL[dtype], ("wgrad", ci, co, dims, s, n)


def norm_draws(n=24):
    rng = random.Random(int(os.environ.get("RX_FUZZ_SEED", "31337")) + 2)
    return [(rng.choice([32, 64, 96, 128, 256, 512]), tuple(rng.choice([1, 2, 3, 4, 5, 8, 9, 16, 20, 33]) for _ in range(3)),
             rng.choice([1, 2, 3]), rng.random() < 0.5, rng.choice([0.01, 0.0, 1.0]), rng.random() < 0.5) for _ in range(n)]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("i", range(24))
def test_random_instnorm_act(ops, dtype, i):
    c, dims, n, with_res, slope, acc_res = norm_draws()[i]
    if dims[0] * dims[1] * dims[2] < 8:
        pytest.skip("a norm over 1-7 voxels: the output is (nearly) a sign pattern and its gradient a cancellation residue")
    y = (rnd((n, c, *dims), dtype, 21 + i) * 0.7 + 0.2).to(dtype).double()       # (representable in the storage type again)
    r = rnd((n, c, *dims), dtype, 22 + i)
    g = rnd((n, c, *dims), dtype, 23 + i, scale=0.3)
    yr, rr = y.clone().requires_grad_(True), r.clone().requires_grad_(True)
    z = F.instance_norm(yr, eps=1e-5)
    if with_res:
        z = z + rr
    ref = F.leaky_relu(z, slope) if slope != 1.0 else z
    ref.backward(g)
    ya, ra, ga = to_act(ops, y, dtype), to_act(ops, r, dtype), to_act(ops, g, dtype)
    out = ops.Act.empty(n, *dims, c, dtype)
    stats = torch.empty((n, c, 2), device="cuda")
    ops.instnorm_fwd(ya, stats, out, slope, ra if with_res else None)
    assert rel(out.to_ncdhw(), ref.detach()) < TOL[dtype], ("fwd", c, dims, n, with_res, slope)
    dy = ops.Act.empty(n, *dims, c, dtype)
    rbase = rnd((n, c, *dims), dtype, 24 + i, scale=0.2)
    dres = to_act(ops, rbase, dtype) if with_res else None
    # mask from the saved output for residual layers, from the sign of xhat otherwise (out = None)
    ops.instnorm_act_bwd(ga, ya, stats, out if (with_res and slope != 1.0) else None, dy, slope, dres, acc_res and with_res)
    assert rel(dy.to_ncdhw(), yr.grad) < 6 * TOL[dtype], ("bwd", c, dims, n, with_res, slope, rel(dy.to_ncdhw(), yr.grad))
    if with_res:
        want = rr.grad + (rbase if acc_res else 0)
        assert rel(dres.to_ncdhw(), want) < 3 * TOL[dtype], ("dres", c, dims, n)
