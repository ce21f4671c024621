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


# ---- fused entry points == their unfused sequences, at random extents around the sizes where the dispatch changes --------------
def fused_draws(n=20):
    rng = random.Random(int(os.environ.get("RX_FUZZ_SEED", "31337")) + 3)
    out = []
    for _ in range(n):
        c = rng.choice([32, 32, 64, 64, 128])
        dims = (rng.choice([4, 8, 12, 14, 16, 20, 30]), rng.choice([8, 16, 20, 32, 36]), rng.choice([16, 16, 32, 48, 64, 24, 8]))
        out.append((c, dims, rng.choice([1, 2, 3]), rng.choice([0.01, 0.0, 1.0]), rng.random() < 0.5))
    return out


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("i", range(20))
def test_random_fused_conv_norm_entry_points(ops, dtype, i):
    """rx_conv3d_fwd_stats == conv + stats (same y, same statistics to round-off); rx_conv3d_bwd_data_instats == the plain data
    gradient (same dx bit for bit) and, when it reports `fused`, the two InstanceNorm-backward means of the two-pass reduction;
    rx_instnorm_act_pool_fwd == apply + pool bit for bit -- whatever kernel the extent selects"""
    c, dims, n, slope, acc = fused_draws()[i]
    k, s = (3, 3, 3), (1, 1, 1)
    x = to_act(ops, rnd((n, c, *dims), dtype, seed=31 + i), dtype)
    w = rnd((c, c, 3, 3, 3), torch.float32, seed=32 + i, scale=(27 * c) ** -0.5).float().cuda()
    wf, wb = ops.pack_conv_weight(w, dtype)
    y1, y2 = ops.Act.empty(n, *dims, c, dtype), ops.Act.empty(n, *dims, c, dtype)
    s1, s2 = torch.empty((n, c, 2), device="cuda"), torch.empty((n, c, 2), device="cuda")
    ops.conv3d_fwd(x, wf, None, y1, k, s)
    ops.instnorm_stats(y1, s1)
    ops.conv3d_fwd_stats(x, wf, None, y2, k, s, s2)
    assert torch.equal(y1.t, y2.t), (c, dims, n, last_kernel(ops))
    assert torch.allclose(s1, s2, rtol=5e-5, atol=5e-6), ((s1 - s2).abs().max().item(), c, dims, n)
    # data gradient with the InstanceNorm-backward sums of the layer that produced x
    g = to_act(ops, rnd((n, c, *dims), dtype, seed=33 + i, scale=0.2), dtype)
    base = rnd((n, c, *dims), dtype, seed=34 + i, scale=0.1)
    dx1, dx2 = to_act(ops, base, dtype), to_act(ops, base, dtype)
    m12 = torch.full((n, c, 2), float("nan"), device="cuda")
    ops.conv3d_bwd_data(g, wb, dx1, k, s, acc)
    fused = ops.conv3d_bwd_data_instats(g, wb, dx2, k, s, acc, y1, s1, slope, m12)
    assert torch.equal(dx1.t, dx2.t), (c, dims, n, acc)
    d1, d2 = ops.Act.empty(n, *dims, c, dtype), ops.Act.empty(n, *dims, c, dtype)
    ops.instnorm_act_bwd(dx1, y1, s1, None, d1, slope)
    if fused:
        ops.instnorm_act_bwd_apply(dx2, y1, s1, None, d2, m12, slope)
        a_, b_ = d2.tensor().double(), d1.tensor().double()
        assert ((a_ - b_).norm() / b_.norm().clamp_min(1e-30)).item() < 5e-3, (c, dims, n, slope)
    # block epilogue + pool of the next skip path
    if all(d % 2 == 0 for d in dims):
        res = to_act(ops, rnd((n, c, *dims), dtype, seed=35 + i), dtype)
        pd = tuple(d // 2 for d in dims)
        o1, o2 = ops.Act.empty(n, *dims, c, dtype), ops.Act.empty(n, *dims, c, dtype)
        p1, p2 = ops.Act.empty(n, *pd, c, dtype), ops.Act.empty(n, *pd, c, dtype)
        ops.instnorm_act_fwd(y1, s1, o1, 0.01, res)
        ops.avgpool_fwd(o1, p1, (2, 2, 2))
        ops.instnorm_act_pool_fwd(y1, s1, o2, p2, (2, 2, 2), 0.01, res)
        assert torch.equal(o1.t, o2.t) and torch.equal(p1.t, p2.t), (c, dims, n)
