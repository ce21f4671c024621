"""GPU (-m gpu): every C-ABI entry point of librxunet.so against a plain torch CPU fp64 reference
of the torch primitive it replaces.  fp32 mode must agree to ~1e-5 (exact-fp32 MFMA / VALU);
bf16/f16 modes are checked against the same reference evaluated on inputs rounded to the storage
type (tolerance = a few ulps of the 8/11-bit mantissa on the output rounding, stated per test)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DTYPES = [torch.float32, torch.bfloat16, torch.float16]
TOL = {torch.float32: 2e-5, torch.bfloat16: 1.2e-2, torch.float16: 2e-3}  # rel-L2 per tensor


@pytest.fixture(scope="module")
def ops():
    import mt3d_amd  # noqa: F401
    from mt3d_amd.engine import lib, ops as _ops
    lib.require_device()
    return _ops


def rel(a, b):
    a, b = a.double().cpu().flatten(), b.double().cpu().flatten()
    return ((a - b).norm() / b.norm().clamp(min=1e-30)).item()


def rnd(shape, dtype, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    t = (torch.randn(shape, generator=g) * scale)
    return t.to(dtype).double()  # value representable in the storage dtype, held in fp64


def to_act(ops, x_ncdhw, dtype, ld=None, c0=0):
    """NCDHW fp64 (values already representable) -> device Act, optionally inside a wider buffer."""
    n, c, z, y, x = x_ncdhw.shape
    ld = c if ld is None else ld
    buf = torch.full((n, z, y, x, ld), 7.0, dtype=dtype, device="cuda")  # poison the unused channels
    buf[..., c0:c0 + c] = x_ncdhw.permute(0, 2, 3, 4, 1).to(dtype).cuda()
    return ops.Act(buf, c0, c)


def out_dim(i, k, s):
    return (i + 2 * ((k - 1) // 2) - k) // s + 1


CONV_CASES = [
    # (ci, co, (z,y,x), kernel, stride)
    (32, 32, (8, 8, 8), (3, 3, 3), (1, 1, 1)),
    (32, 64, (6, 10, 12), (3, 3, 3), (1, 1, 1)),      # ragged: voxel count not a tile multiple
    (64, 96, (8, 8, 8), (3, 3, 3), (2, 2, 2)),        # Co % 64 != 0 -> 32-wide tile
    (64, 64, (4, 16, 16), (3, 3, 3), (1, 2, 2)),      # anisotropic stride
    (32, 32, (5, 8, 8), (1, 3, 3), (1, 1, 1)),        # anisotropic kernel
    (64, 32, (8, 8, 8), (1, 1, 1), (1, 1, 1)),        # 1x1x1 projection
    (256, 256, (4, 4, 4), (3, 3, 3), (1, 1, 1)),      # deep layer: few voxels -> split-K path
    (512, 512, (4, 4, 4), (3, 3, 3), (1, 1, 1)),      # bottleneck: 256 panel pairs -> single-split direct wgrad epilogue
    (512, 512, (3, 8, 8), (3, 3, 3), (1, 1, 1)),      # same, several (ragged) tiles per workgroup
    (32, 32, (30, 36, 64), (3, 3, 3), (1, 1, 1)),     # >= 512 tiles of 4x4x16: persistent weight-stationary kernel (ragged z/y)
    (64, 64, (14, 32, 64), (3, 3, 3), (1, 1, 1)),     # >= 256 tiles, 64 channels: wave-specialised producer/consumer kernel
    (128, 64, (32, 32, 64), (3, 3, 3), (2, 2, 2)),    # stride-2 data gradient on the parity-class halo kernel (64 dY channels)
    (128, 64, (24, 40, 72), (3, 3, 3), (2, 2, 2)),    # same, ragged dY tiles
    (64, 128, (12, 20, 18), (1, 1, 1), (1, 1, 1)),    # 1x1x1 projection above 4096 voxels: the streaming pointwise kernel (fwd and bwd-data)
    # round 3: any kernel size 1..7 / stride 1..4 per axis (the reference passes a manual model_config's values straight to
    # Conv(k, stride, pad=(k-1)//2), build_network_from_config.py:85-148) -- tap tables of up to 343 entries
    (32, 32, (9, 10, 12), (5, 5, 5), (1, 1, 1)),      # 125 taps
    (32, 64, (7, 9, 11), (7, 7, 7), (1, 1, 1)),       # 343 taps: the table limit; weights beyond the 27-tap pack tile
    (32, 32, (6, 12, 13), (1, 5, 7), (1, 2, 1)),      # mixed sizes, one strided axis
    (64, 32, (12, 12, 12), (3, 3, 3), (3, 3, 3)),     # stride 3: 27 output parity classes in the data gradient (4 launches of <= 8)
    (32, 32, (16, 16, 16), (5, 5, 5), (4, 4, 4)),     # stride 4, 5-wide kernel: 64 classes
    (32, 32, (9, 9, 12), (7, 3, 5), (2, 1, 3)),       # everything different
    (32, 32, (8, 9, 10), (2, 4, 6), (1, 2, 2)),       # even kernel sizes: pad (k-1)//2 is asymmetric, the output shrinks
    (32, 32, (8, 8, 8), (1, 1, 1), (2, 2, 2)),        # kernel narrower than the stride: 7 of the 8 classes of dx have no tap (zeros)
    (64, 96, (14, 32, 32), (3, 3, 3), (1, 1, 1)),     # 128 ragged tiles x 3 (fwd) / 2 (dgrad) 32-channel blocks: the 32-channel-block DMA pipeline
]


# which kernel instantiation the library must dispatch (16-bit modes) for the cases whose comment names one: a dispatch change
# that silently moved them back onto the generic gather kernels would otherwise keep this file green.
# (fwd, bwd-data, bwd-weight) as reported by rx_last_conv_kernel(); None = not pinned.
EXPECT_KERNELS = {
    (256, 256, (4, 4, 4), (3, 3, 3), (1, 1, 1)): ("igemm_fat_kernel", "igemm_fat_kernel", "wgrad_halo_kernel"),
    (512, 512, (4, 4, 4), (3, 3, 3), (1, 1, 1)): ("igemm_fat_kernel", "igemm_fat_kernel", "wgrad_halo_kernel"),
    (512, 512, (3, 8, 8), (3, 3, 3), (1, 1, 1)): ("igemm_fat_kernel", "igemm_fat_kernel", "wgrad_halo_kernel"),
    (32, 32, (30, 36, 64), (3, 3, 3), (1, 1, 1)): ("conv_halo32p_kernel", "conv_halo32p_kernel", "wgrad_halo16ws_kernel"),
    (64, 64, (14, 32, 64), (3, 3, 3), (1, 1, 1)): ("conv_halo64ws_kernel", "conv_halo64ws_kernel", "wgrad_halo16ws_kernel"),
    (64, 96, (14, 32, 32), (3, 3, 3), (1, 1, 1)): ("conv_halo32ws_kernel", "conv_halo32ws_kernel", None),
    (128, 64, (32, 32, 64), (3, 3, 3), (2, 2, 2)): (None, "dgrad_s2p_kernel", None),
    (128, 64, (24, 40, 72), (3, 3, 3), (2, 2, 2)): (None, "dgrad_s2p_kernel", None),
    (64, 128, (12, 20, 18), (1, 1, 1), (1, 1, 1)): ("pointwise_kernel", "pointwise_kernel", None),
}


def last_kernel(ops):
    return ops.load().rx_last_conv_kernel().decode()


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv3d_fwd_bwd(ops, dtype, case):
    ci, co, dims, k, s = case
    seen = {}
    n = 2
    x = rnd((n, ci, *dims), dtype, 1)
    w = rnd((co, ci, *k), dtype, 2, scale=(ci * k[0] * k[1] * k[2]) ** -0.5)
    b = rnd((co,), torch.float32, 3)
    pad = [(kk - 1) // 2 for kk in k]
    xr = x.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    y_ref = F.conv3d(xr, wr, b, stride=s, padding=pad)
    odims = tuple(out_dim(d, kk, ss) for d, kk, ss in zip(dims, k, s))
    assert tuple(y_ref.shape[2:]) == odims
    gy = rnd(tuple(y_ref.shape), dtype, 4)
    y_ref.backward(gy)

    xa = to_act(ops, x, dtype, ld=ci + 32, c0=32)       # read through a channel slice (concat view)
    w_fwd, w_bwd = ops.pack_conv_weight(w.float().cuda(), dtype)
    ya = ops.Act.zeros(n, *odims, co, dtype)
    ops.conv3d_fwd(xa, w_fwd, b.float().cuda(), ya, k, s)
    seen["fwd"] = last_kernel(ops)
    assert rel(ya.to_ncdhw(), y_ref.detach()) < TOL[dtype]
    # no bias
    ops.conv3d_fwd(xa, w_fwd, None, ya, k, s)
    assert rel(ya.to_ncdhw(), (y_ref.detach() - b.view(1, -1, 1, 1, 1))) < TOL[dtype]

    gya = to_act(ops, gy, dtype)
    dxa = ops.Act.zeros(n, *dims, ci, dtype)
    ops.conv3d_bwd_data(gya, w_bwd, dxa, k, s, accumulate=False)
    seen["dgrad"] = last_kernel(ops)
    assert rel(dxa.to_ncdhw(), xr.grad) < TOL[dtype]
    ops.conv3d_bwd_data(gya, w_bwd, dxa, k, s, accumulate=True)      # dx += ...
    assert rel(dxa.to_ncdhw(), 2 * xr.grad) < 2 * TOL[dtype]

    dw = torch.empty((co, ci, *k), dtype=torch.float32, device="cuda")
    ops.conv3d_bwd_weight(xa, gya, dw, k, s)
    seen["wgrad"] = last_kernel(ops)
    assert rel(dw, wr.grad) < TOL[dtype]
    if dtype != torch.float32 and case in EXPECT_KERNELS:
        for which, want in zip(("fwd", "dgrad", "wgrad"), EXPECT_KERNELS[case]):
            assert want is None or seen[which] == want, (case, which, seen)


CONVT_CASES = [
    (64, 32, (4, 4, 4), (2, 2, 2)),
    (128, 64, (3, 5, 6), (2, 2, 2)),
    (64, 64, (4, 6, 6), (1, 2, 2)),
    (512, 256, (2, 2, 2), (2, 2, 2)),
    (64, 32, (12, 18, 20), (2, 2, 2)),         # >= 4096 voxels: the streaming pointwise kernel (x once, y once)
    (256, 128, (8, 9, 16), (2, 2, 2)),         # 8 taps x 256 channels: 132 KB of weights would leave one workgroup per CU -> gather kernel
    (128, 64, (10, 16, 24), (1, 2, 2)),        # anisotropic stride
    (64, 32, (16, 32, 40), (2, 2, 2)),         # >= 32768 coarse voxels: convT_wgrad_kernel (every operand byte once)
    (128, 64, (16, 32, 34), (2, 2, 2)),        # the same on the 4 x 2 block instantiation (80 KB of LDS panels)
    (64, 64, (20, 32, 32), (1, 2, 2)),         # four taps: waves 4..7 of the weight-gradient workgroup only stage
    (64, 32, (5, 6, 7), (3, 3, 3)),            # round 3: stride 3 (27 phases: four launches of <= 8) and 4
    (32, 32, (4, 5, 6), (1, 4, 3)),
    (64, 64, (8, 10, 12), (4, 4, 4)),          # 64 taps: pointwise weights would not fit -> gather kernel; naive pack (> 27 taps)
]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", CONVT_CASES)
def test_convT3d_fwd_bwd(ops, dtype, case):
    ci, co, dims, s = case
    n = 2
    x = rnd((n, ci, *dims), dtype, 5)
    w = rnd((ci, co, *s), dtype, 6, scale=ci ** -0.5)
    b = rnd((co,), torch.float32, 7)
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    y_ref = F.conv_transpose3d(xr, wr, b, stride=s)
    gy = rnd(tuple(y_ref.shape), dtype, 8)
    y_ref.backward(gy)
    odims = tuple(d * ss for d, ss in zip(dims, s))

    xa = to_act(ops, x, dtype)
    w_fwd, w_bwd = ops.pack_convT_weight(w.float().cuda(), dtype)
    # write straight into channels [0, co) of a 2*co-wide concat buffer (decoder.py:146-147)
    cat = ops.Act(torch.full((n, *odims, 2 * co), 3.0, dtype=dtype, device="cuda"))
    ops.convT3d_fwd(xa, w_fwd, b.float().cuda(), cat.slice(0, co), s)
    if dtype != torch.float32 and n * dims[0] * dims[1] * dims[2] >= 4096 and ci <= 128:
        assert last_kernel(ops) == "pointwise_kernel"
    assert rel(cat.slice(0, co).to_ncdhw(), y_ref.detach()) < TOL[dtype]
    assert torch.all(cat.slice(co, co).tensor() == 3.0)              # the skip half is untouched

    gya = to_act(ops, gy, dtype, ld=2 * co, c0=0)
    dxa = ops.Act.zeros(n, *dims, ci, dtype)
    ops.convT3d_bwd_data(gya, w_bwd, dxa, s)
    nv = n * dims[0] * dims[1] * dims[2]
    assert rel(dxa.to_ncdhw(), xr.grad) < TOL[dtype]
    ops.convT3d_bwd_data(gya, w_bwd, dxa, s, True)                    # accumulate (a second decoder's gradient)
    assert rel(dxa.to_ncdhw(), 2 * xr.grad) < 2 * TOL[dtype]
    dw = torch.empty((ci, co, *s), dtype=torch.float32, device="cuda")
    ops.convT3d_bwd_weight(xa, gya, dw, s)
    if dtype != torch.float32 and nv >= 32768 and (ci // 32, co // 32) in ((2, 1), (4, 2), (1, 1), (2, 2)):
        assert last_kernel(ops) == "convT_wgrad_kernel"
    assert rel(dw, wr.grad) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("c,dims", [(32, (8, 8, 8)), (64, (5, 6, 7)), (320, (4, 4, 4)), (512, (2, 2, 2)),
                                    (32, (20, 16, 16)), (16, (8, 8, 8)), (96, (16, 16, 16))])   # small + large paths
def test_instnorm_lrelu_residual(ops, dtype, c, dims):
    n = 2
    y = (rnd((n, c, *dims), dtype, 9) * 2 + 0.5).to(dtype).double()   # keep values representable
    r = rnd((n, c, *dims), dtype, 10)
    g = rnd((n, c, *dims), dtype, 11)
    ya, ra, ga = to_act(ops, y, dtype), to_act(ops, r, dtype), to_act(ops, g, dtype)
    stats = torch.empty((n, c, 2), dtype=torch.float32, device="cuda")
    ops.instnorm_stats(ya, stats)
    mean_ref = y.mean(dim=(2, 3, 4))
    var_ref = y.var(dim=(2, 3, 4), unbiased=False)
    assert rel(stats[..., 0], mean_ref) < 1e-5
    assert rel(stats[..., 1], (var_ref + 1e-5).rsqrt()) < 1e-5
    for with_res, slope in [(False, 0.01), (True, 0.01), (False, 1.0)]:
        yr = y.clone().requires_grad_(True)
        rr = r.clone().requires_grad_(True)
        pre = F.instance_norm(yr, eps=1e-5) + (rr if with_res else 0.0)
        ref = F.leaky_relu(pre, slope) if slope != 1.0 else pre
        oa = ops.Act.zeros(n, *dims, c, dtype)
        ops.instnorm_act_fwd(ya, stats, oa, slope, ra if with_res else None)
        assert rel(oa.to_ncdhw(), ref.detach()) < TOL[dtype]
        # fused stats+apply entry point (single launch on small tensors) must agree with the two-call path
        ob, stats2 = ops.Act.zeros(n, *dims, c, dtype), torch.zeros_like(stats)
        ops.instnorm_fwd(ya, stats2, ob, slope, ra if with_res else None)
        assert rel(stats2, stats) < 1e-5
        assert rel(ob.to_ncdhw(), oa.to_ncdhw()) < (1e-6 if dtype == torch.float32 else TOL[dtype])
        # backward: feed the mask from the reference output (rounded) so both sides agree on signs
        out_ref_act = to_act(ops, ref.detach().to(dtype).double(), dtype)
        ref.backward(g)
        dya = ops.Act.zeros(n, *dims, c, dtype)
        dra = ops.Act.zeros(n, *dims, c, dtype) if with_res else None
        ops.instnorm_act_bwd(ga, ya, stats, out_ref_act if slope != 1.0 else None, dya, slope, dra, False)
        assert rel(dya.to_ncdhw(), yr.grad) < 3 * TOL[dtype]
        if not with_res and slope != 1.0:
            # no residual: the mask is the sign of the normalised value, the saved output is not needed at all
            dyb = ops.Act.zeros(n, *dims, c, dtype)
            ops.instnorm_act_bwd(ga, ya, stats, None, dyb, slope, None, False)
            assert rel(dyb.to_ncdhw(), yr.grad) < 3 * TOL[dtype]
        if with_res:
            assert rel(dra.to_ncdhw(), rr.grad) < TOL[dtype]
            ops.instnorm_act_bwd(ga, ya, stats, out_ref_act, dya, slope, dra, True)   # accumulate
            assert rel(dra.to_ncdhw(), 2 * rr.grad) < 2 * TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("c,dims", [(32, (8, 8, 8)), (64, (5, 12, 16))])      # single-launch range / three-launch path
def test_channel_dropout_as_masked_statistics(ops, dtype, c, dims):
    """nn.Dropout3d between conv and InstanceNorm (simple_conv_blocks.py:57-66) == the norm computed with eps * (1-p)^2 plus
    rx_instnorm_stats_mask (rstd = 0 for the dropped planes): forward and backward against fp64 torch autograd of
    leaky_relu(instance_norm(y * mask / (1 - p)))."""
    n, p, eps, slope = 3, 0.3, 1e-5, 0.01
    yv = rnd((n, c, *dims), dtype, seed=1) + 0.2
    gv = rnd((n, c, *dims), dtype, seed=2, scale=0.1)
    keep = torch.bernoulli(torch.full((n, c), 1 - p), generator=torch.Generator().manual_seed(3))
    keep[0, 0], keep[1, 1] = 0.0, 1.0
    yr = yv.clone().requires_grad_(True)
    ref = F.leaky_relu(F.instance_norm(yr * (keep.double() / (1 - p)).view(n, c, 1, 1, 1), eps=eps), slope)
    ref.backward(gv)
    ya, ga = to_act(ops, yv, dtype), to_act(ops, gv, dtype)
    out, dy = ops.Act.empty(n, *dims, c, dtype), ops.Act.empty(n, *dims, c, dtype)
    stats = torch.empty((n, c, 2), device="cuda")
    ops.instnorm_stats(ya, stats, eps * (1 - p) ** 2)
    ops.instnorm_stats_mask(stats, keep.cuda())
    ops.instnorm_act_fwd(ya, stats, out, slope)
    ops.instnorm_act_bwd(ga, ya, stats, None, dy, slope)
    torch.cuda.synchronize()
    assert (stats[..., 1].cpu()[keep == 0] == 0).all() and (stats[..., 1].cpu()[keep == 1] > 0).all()
    o = out.to_ncdhw().double().cpu()
    assert (o[keep == 0] == 0).all()
    assert rel(o, ref.detach()) < TOL[dtype]
    d = dy.to_ncdhw().double().cpu()
    assert (d[keep == 0] == 0).all()
    assert rel(d, yr.grad) < 5 * TOL[dtype]          # (cancellation in the norm backward + the output rounding)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("s", [(2, 2, 2), (1, 2, 2), (3, 3, 3), (1, 4, 2)])
def test_avgpool(ops, dtype, s):
    n, c, dims = 2, 64, (4, 6, 8) if max(s) <= 2 else (12, 12, 24)
    x = rnd((n, c, *dims), dtype, 12)
    xr = x.clone().requires_grad_(True)
    ref = F.avg_pool3d(xr, s, s)
    g = rnd(tuple(ref.shape), dtype, 13)
    ref.backward(g)
    xa = to_act(ops, x, dtype)
    odims = tuple(d // ss for d, ss in zip(dims, s))
    ya = ops.Act.zeros(n, *odims, c, dtype)
    ops.avgpool_fwd(xa, ya, s)
    assert rel(ya.to_ncdhw(), ref.detach()) < TOL[dtype]
    dxa = ops.Act.zeros(n, *dims, c, dtype)
    ops.avgpool_bwd(to_act(ops, g, dtype), dxa, s)
    assert rel(dxa.to_ncdhw(), xr.grad) < TOL[dtype]
    ops.avgpool_bwd(to_act(ops, g, dtype), dxa, s, accumulate=True)
    assert rel(dxa.to_ncdhw(), 2 * xr.grad) < 2 * TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cin,k", [(1, (3, 3, 3)), (2, (3, 3, 3)), (1, (1, 3, 3)), (4, (3, 3, 3)), (5, (3, 3, 3)), (8, (3, 3, 3)),   # > 4: VALU kernels
                                   (6, (1, 3, 3)), (8, (3, 1, 3)), (5, (1, 1, 1)), (4, (1, 1, 1)), (8, (3, 3, 1)),    # few taps x > 4 channels (fuzz find)
                                   (12, (3, 3, 3)), (16, (3, 3, 3)), (16, (1, 3, 3))])                               # weights beyond 48 KB of LDS
def test_stem(ops, dtype, cin, k):
    n, co, dims = 2, 32, (6, 9, 10)
    x = rnd((n, cin, *dims), torch.float32, 14)
    w = rnd((co, cin, *k), torch.float32, 15, scale=0.3)
    b = rnd((co,), torch.float32, 16)
    wr = w.clone().requires_grad_(True)
    ref = F.conv3d(x, wr, b, padding=[(kk - 1) // 2 for kk in k])
    g = rnd(tuple(ref.shape), dtype, 17)
    ref.backward(g)
    xd = x.float().cuda().contiguous()
    oa = ops.Act.zeros(n, *dims, co, dtype)
    ops.stem_conv_fwd(xd, w.float().cuda(), b.float().cuda(), oa, k)
    assert rel(oa.to_ncdhw(), ref.detach()) < TOL[dtype]
    # conv + InstanceNorm statistics in one call (one pass on the MFMA kernel, the two calls elsewhere): the same output bits,
    # (mean, rstd) of the values as stored
    ob = ops.Act.zeros(n, *dims, co, dtype)
    st_f, st_s = torch.zeros((n, co, 2), device="cuda"), torch.zeros((n, co, 2), device="cuda")
    ops.stem_conv_fwd_stats(xd, w.float().cuda(), b.float().cuda(), ob, k, st_f)
    assert torch.equal(ob.tensor(), oa.tensor())
    ops.instnorm_stats(oa, st_s)
    yv = oa.to_ncdhw().double().cpu()
    mean, var = yv.mean(dim=(2, 3, 4)), yv.var(dim=(2, 3, 4), unbiased=False)
    for st in (st_f, st_s):
        assert rel(st[..., 0].cpu().double(), mean) < 1e-5 and rel(st[..., 1].cpu().double(), (var + 1e-5).rsqrt()) < 1e-5
    dw = torch.empty((co, cin, *k), dtype=torch.float32, device="cuda")
    ops.stem_conv_bwd_weight(xd, to_act(ops, g, dtype), dw, k)
    # 16-bit modes run the MFMA kernel, which rounds the image to the compute dtype (fp32 mode stays exact)
    assert rel(dw, wr.grad) < (2e-5 if dtype == torch.float32 else TOL[dtype])


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("k", [1, 3, 8, 9, 16, 17, 40, 64, 65, 117, 200])   # K <= 8: lean kernel; 9..16 / ..32 / ..64: wider accumulator arrays; > 64: chunks of 64 + a softmax pass; backward in chunks of 16
def test_head(ops, dtype, k):
    from mt3d_amd.engine import lib
    n, c, dims = 2, 32, (6, 7, 8)
    x = rnd((n, c, *dims), dtype, 18)
    w = rnd((k, c), torch.float32, 19, scale=0.3)
    b = rnd((k,), torch.float32, 20)
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = F.conv3d(xr, wr.view(k, c, 1, 1, 1), br)
    g = rnd(tuple(ref.shape), torch.float32, 21)
    ref.backward(g)
    xa = to_act(ops, x, dtype)
    out = torch.empty((n, k, *dims), dtype=torch.float32, device="cuda")
    wd, bd = w.float().cuda(), b.float().cuda()
    ops.head_fwd(xa, wd, bd, out)
    assert rel(out, ref.detach()) < 2e-5          # fp32 accumulate of exactly representable inputs
    ops.head_fwd(xa, wd, bd, out, lib.RX_ACT_SIGMOID)
    assert rel(out, torch.sigmoid(ref.detach())) < 2e-5
    ops.head_fwd(xa, wd, bd, out, lib.RX_ACT_SOFTMAX)
    assert rel(out, torch.softmax(ref.detach(), 1)) < 2e-5
    dxa = ops.Act.zeros(n, *dims, c, dtype)
    dw = torch.empty((k, c), dtype=torch.float32, device="cuda")
    db = torch.empty((k,), dtype=torch.float32, device="cuda")
    ops.head_bwd(g.float().cuda().contiguous(), xa, wd, dxa, dw, db)
    assert rel(dxa.to_ncdhw(), xr.grad) < TOL[dtype]
    assert rel(dw, wr.grad) < 2e-5
    assert rel(db, br.grad) < 2e-5


@pytest.mark.parametrize("dtype", DTYPES)
def test_channel_sum_and_pack(ops, dtype):
    n, c, dims = 2, 64, (5, 6, 7)
    x = rnd((n, c, *dims), dtype, 22)
    out = torch.empty((c,), dtype=torch.float32, device="cuda")
    ops.channel_sum(to_act(ops, x, dtype), out)
    assert rel(out, x.sum(dim=(0, 2, 3, 4))) < 1e-5
    w = rnd((96, 64, 3, 3, 3), dtype, 23)
    w_fwd, w_bwd = ops.pack_conv_weight(w.float().cuda(), dtype)
    ref = w.reshape(96, 64, 27)
    assert torch.equal(w_fwd.double().cpu(), ref.permute(2, 0, 1))
    assert torch.equal(w_bwd.double().cpu(), ref.permute(2, 1, 0))
    wt = rnd((64, 32, 2, 2, 2), dtype, 24)
    t_fwd, t_bwd = ops.pack_convT_weight(wt.float().cuda(), dtype)
    reft = wt.reshape(64, 32, 8)
    assert torch.equal(t_fwd.double().cpu(), reft.permute(2, 1, 0))
    assert torch.equal(t_bwd.double().cpu(), reft.permute(2, 0, 1))


def test_error_paths(ops):
    from mt3d_amd.engine.lib import RxError
    x = ops.Act.zeros(1, 4, 4, 4, 24, torch.bfloat16)      # 24 channels: not a multiple of 32
    y = ops.Act.zeros(1, 4, 4, 4, 32, torch.bfloat16)
    w = torch.zeros((27, 32, 24), dtype=torch.bfloat16, device="cuda")
    with pytest.raises(RxError):
        ops.conv3d_fwd(x, w, None, y, (3, 3, 3), (1, 1, 1))
    with pytest.raises(RxError):
        ops.avgpool_fwd(y, y, (2, 2, 2))                    # geometry mismatch


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize("case", [
    (32, 32, (30, 36, 64)),      # conv_halo32p: statistics from the conv epilogue (ragged z / y tiles)
    (64, 64, (14, 32, 64)),      # conv_halo64ws: same, 64-channel block
    (64, 128, (8, 16, 32)),      # two channel blocks share the tiles
    (32, 64, (6, 10, 12)),       # not a persistent-kernel layer: conv followed by the separate statistics pass
    (256, 256, (16, 16, 16)),    # deep layer on the one-tile-per-workgroup kernel: statistics from its epilogue too
    (128, 96, (12, 16, 32)),     # same kernel (Co % 64 != 0), ragged z tiles
])
def test_conv3d_fwd_stats(ops, dtype, case):
    """rx_conv3d_fwd_stats == rx_conv3d_fwd + rx_instnorm_stats: same y bit for bit, same (mean, rstd) to fp32 round-off"""
    ci, co, dims = case
    n = 3 if dims[0] == 30 else 2
    x = to_act(ops, rnd((n, ci, *dims), dtype, seed=3), dtype)
    w = rnd((co, ci, 3, 3, 3), torch.float32, seed=4, scale=0.1).float().cuda()
    b = rnd((co,), torch.float32, seed=5).float().cuda()
    wf, _ = ops.pack_conv_weight(w, dtype)
    y1, y2 = ops.Act.empty(n, *dims, co, dtype), ops.Act.empty(n, *dims, co, dtype)
    s1 = torch.empty((n, co, 2), device="cuda")
    s2 = torch.empty((n, co, 2), device="cuda")
    k, s = (3, 3, 3), (1, 1, 1)
    ops.conv3d_fwd(x, wf, b, y1, k, s)
    ops.instnorm_stats(y1, s1)
    ops.conv3d_fwd_stats(x, wf, b, y2, k, s, s2)
    torch.cuda.synchronize()
    assert torch.equal(y1.t, y2.t)
    assert torch.allclose(s1, s2, rtol=2e-5, atol=2e-6), (s1 - s2).abs().max()


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("stride", [(2, 2, 2), (1, 2, 2)])
def test_instnorm_act_pool_fwd_is_the_two_calls(ops, dtype, stride):
    n, c, dims = 2, 32, (8, 12, 20)
    y = to_act(ops, rnd((n, c, *dims), dtype, seed=1), dtype)
    res = to_act(ops, rnd((n, c, *dims), dtype, seed=2), dtype, ld=64, c0=32)        # residual inside a wider buffer
    stats = torch.empty((n, c, 2), device="cuda")
    ops.instnorm_stats(y, stats)
    pd = tuple(d // s for d, s in zip(dims, stride))
    o1, o2 = ops.Act.empty(n, *dims, c, dtype), ops.Act.empty(n, *dims, c, dtype)
    p1, p2 = ops.Act.empty(n, *pd, c, dtype), ops.Act.empty(n, *pd, c, dtype)
    ops.instnorm_act_fwd(y, stats, o1, 0.01, res)
    ops.avgpool_fwd(o1, p1, stride)
    ops.instnorm_act_pool_fwd(y, stats, o2, p2, stride, 0.01, res)
    torch.cuda.synchronize()
    assert torch.equal(o1.t, o2.t) and torch.equal(p1.t, p2.t)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_planar_concat_convs_equal_interleaved(ops, dtype):
    """rx_act.cs: the 64-channel concat as two dense 32-channel planes.  Forward (planar x), backward-data (planar dx, with and
    without accumulation) and backward-weight (planar x) must reproduce the interleaved layout bit for bit."""
    n, dims = 2, (16, 32, 64)                    # 4*8*4*2 = 256 tiles: the full-resolution kernels
    xin = rnd((n, 64, *dims), dtype, 1)
    inter = to_act(ops, xin, dtype)                                        # (n, z, y, x, 64)
    root = torch.stack([inter.t[..., :32].contiguous(), inter.t[..., 32:].contiguous()], 0)
    planar = ops.Act.planar(root)
    assert planar.c == 64 and torch.equal(planar.tensor(), inter.t) and planar.slice(32, 32).plane == 1
    w = rnd((32, 64, 3, 3, 3), torch.float32, 2, scale=0.05).float().cuda()
    wf, wb = ops.pack_conv_weight(w, dtype)
    k, s = (3, 3, 3), (1, 1, 1)
    y1, y2 = ops.Act.empty(n, *dims, 32, dtype), ops.Act.empty(n, *dims, 32, dtype)
    ops.conv3d_fwd(inter, wf, None, y1, k, s)
    ops.conv3d_fwd(planar, wf, None, y2, k, s)
    st1, st2 = torch.empty((n, 32, 2), device="cuda"), torch.empty((n, 32, 2), device="cuda")
    ops.conv3d_fwd_stats(planar, wf, None, y2, k, s, st2)
    ops.instnorm_stats(y1, st1)
    torch.cuda.synchronize()
    assert torch.equal(y1.t, y2.t) and torch.allclose(st1, st2, rtol=2e-5, atol=2e-6)
    gy = to_act(ops, rnd((n, 32, *dims), dtype, 3, scale=0.1), dtype)
    base = rnd((n, 64, *dims), dtype, 4, scale=0.1)
    for acc in (False, True):
        d1 = to_act(ops, base, dtype)
        d2 = ops.Act.planar(torch.stack([d1.t[..., :32].contiguous(), d1.t[..., 32:].contiguous()], 0))
        ops.conv3d_bwd_data(gy, wb, d1, k, s, acc)
        ops.conv3d_bwd_data(gy, wb, d2, k, s, acc)
        torch.cuda.synchronize()
        assert torch.equal(d1.t, d2.tensor()), acc
    dw1, dw2 = torch.empty_like(w), torch.empty_like(w)
    ops.conv3d_bwd_weight(inter, gy, dw1, k, s)
    ops.conv3d_bwd_weight(planar, gy, dw2, k, s)
    torch.cuda.synchronize()
    assert torch.equal(dw1, dw2)
    from mt3d_amd.engine.lib import RxError
    small = ops.Act.planar(torch.zeros((2, 1, 4, 4, 16, 32), dtype=dtype, device="cuda"))     # too small for the halo kernels
    with pytest.raises(RxError):
        ops.conv3d_fwd(small, wf, None, ops.Act.empty(1, 4, 4, 16, 32, dtype), k, s)
    with pytest.raises(RxError):
        ops.instnorm_stats(planar, torch.empty((n, 64, 2), device="cuda"))              # every other entry point refuses cs != 0


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("slope", [0.01, 1.0])
@pytest.mark.parametrize("accumulate", [False, True])
@pytest.mark.parametrize("c,dims", [(32, (30, 36, 64)),      # >= 512 tiles of 4x4x16, ragged z / y: conv_halo32p (sums on the consumer waves)
                                    (64, (14, 30, 64)),      # conv_halo64ws: sums on the producer waves out of the LDS-staged tile, ragged z / y
                                    (128, (14, 32, 64))])    # two channel blocks per tile
def test_conv3d_bwd_data_instats(ops, dtype, slope, accumulate, c, dims):
    """rx_conv3d_bwd_data_instats: same dx as rx_conv3d_bwd_data bit for bit, and (mean g', mean g'*xhat) of the InstanceNorm
    layer it completes equal to the two-pass reduction over the stored dx (fp64 on the host)"""
    n = 3
    dyv = to_act(ops, rnd((n, c, *dims), dtype, seed=1, scale=0.2), dtype)
    w = rnd((c, c, 3, 3, 3), torch.float32, seed=2, scale=0.1).float().cuda()
    _, wb = ops.pack_conv_weight(w, dtype)
    yv = to_act(ops, rnd((n, c, *dims), dtype, seed=3) + 0.3, dtype)
    base = rnd((n, c, *dims), dtype, seed=5, scale=0.1)
    dx1, dx2 = to_act(ops, base, dtype), to_act(ops, base, dtype)
    stats = torch.empty((n, c, 2), device="cuda")
    ops.instnorm_stats(yv, stats)
    m12 = torch.full((n, c, 2), float("nan"), device="cuda")
    k, s = (3, 3, 3), (1, 1, 1)
    ops.conv3d_bwd_data(dyv, wb, dx1, k, s, accumulate)
    fused = ops.conv3d_bwd_data_instats(dyv, wb, dx2, k, s, accumulate, yv, stats, slope, m12)
    torch.cuda.synchronize()
    assert fused, "these layers must take a persistent kernel that delivers the sums"
    assert torch.equal(dx1.t, dx2.t)
    g = dx1.to_ncdhw().double().cpu()
    y = yv.to_ncdhw().double().cpu()
    mean = stats[..., 0].double().cpu().view(n, c, 1, 1, 1)
    rstd = stats[..., 1].double().cpu().view(n, c, 1, 1, 1)
    xh = (y - mean) * rstd
    if slope != 1.0:
        g = torch.where(xh > 0, g, g * slope)
    want = torch.stack([g.mean((2, 3, 4)), (g * xh).mean((2, 3, 4))], -1)
    got = m12.double().cpu()
    scale = want.abs().max().item()
    assert torch.isfinite(got).all() and (got - want).abs().max().item() < 2e-4 * scale + 1e-7
    dy1, dy2 = ops.Act.empty(n, *dims, c, dtype), ops.Act.empty(n, *dims, c, dtype)
    ops.instnorm_act_bwd(dx1, yv, stats, None, dy1, slope)
    ops.instnorm_act_bwd_apply(dx2, yv, stats, None, dy2, m12, slope)
    torch.cuda.synchronize()
    a_, b_ = dy2.tensor().double().cpu(), dy1.tensor().double().cpu()
    assert ((a_ - b_).norm() / b_.norm()).item() < 3e-3


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("k", [1, 3])
@pytest.mark.parametrize("slope", [0.01, 1.0])
def test_instnorm_act_bwd_head_is_head_bwd_then_instnorm_bwd(ops, dtype, k, slope):
    """rx_instnorm_act_bwd_head rebuilds the head's data gradient g = dout x w inside both InstanceNorm passes: same dy, bit
    for bit, as rx_head_bwd(dx = g) followed by rx_instnorm_act_bwd(g, ...); and against an fp64 torch reference"""
    n, c, dims = 2, 32, (12, 20, 24)
    V = dims[0] * dims[1] * dims[2]
    y = to_act(ops, rnd((n, c, *dims), dtype, seed=21), dtype)
    out = to_act(ops, rnd((n, c, *dims), dtype, seed=22), dtype)
    dout = rnd((n, k, *dims), torch.float32, seed=23, scale=0.01).float().cuda().contiguous()
    w = rnd((k, c), torch.float32, seed=24).float().cuda().contiguous()
    stats = torch.empty((n, c, 2), device="cuda")
    ops.instnorm_stats(y, stats)
    g = ops.Act.empty(n, *dims, c, dtype)
    dw, db = torch.empty((k, c), device="cuda"), torch.empty((k,), device="cuda")
    dy1, dy2 = ops.Act.empty(n, *dims, c, dtype), ops.Act.empty(n, *dims, c, dtype)
    ops.head_bwd(dout, out, w, g, dw, db)
    ops.instnorm_act_bwd(g, y, stats, None, dy1, slope)
    dw2, db2 = torch.empty_like(dw), torch.empty_like(db)
    ops.head_bwd(dout, out, w, None, dw2, db2)
    ops.instnorm_act_bwd_head(dout, w, y, stats, dy2, slope)
    torch.cuda.synchronize()
    assert torch.equal(dy1.t, dy2.t)
    assert torch.equal(dw, dw2) and torch.equal(db, db2)
    # round 3: the head's own dw / db out of the same reduce pass, the activation recomputed from y (`out` here must then BE the
    # activation of y: rebuild it with the forward kernel) -- no head_bwd launch, dy unchanged bit for bit
    b0 = torch.zeros((k,), device="cuda")
    logits = torch.empty((n, k, *dims), device="cuda")
    act_out = ops.Act.empty(n, *dims, c, dtype)
    ops.instnorm_act_head_fwd(y, stats, act_out, w, b0, logits, 0, slope)
    logits2 = torch.empty_like(logits)
    ops.instnorm_act_head_fwd(y, stats, None, w, b0, logits2, 0, slope)            # nobody needs the activated output: not stored
    dw_ref, db_ref = torch.empty_like(dw), torch.empty_like(db)
    ops.head_bwd(dout, act_out, w, None, dw_ref, db_ref)
    dw3, db3, dy3 = torch.empty_like(dw), torch.empty_like(db), ops.Act.empty(n, *dims, c, dtype)
    ops.instnorm_act_bwd_head(dout, w, y, stats, dy3, slope, dw=dw3, db=db3)
    torch.cuda.synchronize()
    assert torch.equal(logits, logits2)
    assert torch.equal(dy3.t, dy2.t)
    assert rel(dw3, dw_ref) < 2e-5 and rel(db3, db_ref) < 2e-5, (rel(dw3, dw_ref), rel(db3, db_ref))
    # fp64 reference of the same arithmetic
    yd = y.t.double()
    mean, rstd = stats[..., 0].double().view(n, 1, 1, 1, c), stats[..., 1].double().view(n, 1, 1, 1, c)
    xh = (yd - mean) * rstd
    gd = torch.einsum("nkzyx,kc->nzyxc", dout.double(), w.double())
    if slope != 1.0:
        gd = torch.where(xh > 0, gd, gd * slope)
    m1 = gd.mean(dim=(1, 2, 3), keepdim=True)
    m2 = (gd * xh).mean(dim=(1, 2, 3), keepdim=True)
    ref = rstd * (gd - m1 - xh * m2)
    assert rel(dy2.t.float().cpu(), ref.float().cpu()) < (6e-3 if dtype == torch.bfloat16 else 8e-4)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("k,act", [(1, 0), (3, 0), (1, 1), (3, 2)])
def test_instnorm_act_head_fwd_is_the_two_calls(ops, dtype, k, act):
    """rx_instnorm_act_head_fwd == rx_instnorm_act_fwd + rx_head_fwd: activated output bit for bit, fp32 logits to round-off (also with
    the eval-mode sigmoid / softmax)"""
    from mt3d_amd.engine import lib as _l
    codes = {0: _l.RX_ACT_NONE, 1: _l.RX_ACT_SIGMOID, 2: _l.RX_ACT_SOFTMAX}
    n, c, dims = 2, 32, (12, 20, 24)
    y = to_act(ops, rnd((n, c, *dims), dtype, seed=31), dtype)
    w = rnd((k, c), torch.float32, seed=32, scale=0.3).float().cuda().contiguous()
    b = rnd((k,), torch.float32, seed=33).float().cuda().contiguous()
    stats = torch.empty((n, c, 2), device="cuda")
    ops.instnorm_stats(y, stats)
    o1, o2 = ops.Act.empty(n, *dims, c, dtype), ops.Act.empty(n, *dims, c, dtype)
    l1 = torch.empty((n, k, *dims), device="cuda")
    l2 = torch.empty((n, k, *dims), device="cuda")
    ops.instnorm_act_fwd(y, stats, o1, 0.01)
    ops.head_fwd(o1, w, b, l1, codes[act])
    ops.instnorm_act_head_fwd(y, stats, o2, w, b, l2, codes[act], 0.01)
    torch.cuda.synchronize()
    assert torch.equal(o1.t, o2.t)
    assert torch.allclose(l1, l2, rtol=2e-6, atol=2e-6), (l1 - l2).abs().max()     # same products, fp32 sums differ in the last bit


@pytest.mark.parametrize("dtype", DTYPES)
def test_pack_multi_matches_reference_layouts(ops, dtype):
    """rx_pack_multi (one table launch per 40 tensors): every packed copy equals the plain permutation of the PyTorch weight
    -- w_fwd [T][Co][Ci], w_bwd [T][Ci][Co] for Conv3d (Co,Ci,kz,ky,kx) and ConvTranspose3d (Ci,Co,kz,ky,kx) -- bit for bit
    (a cast to the compute type is the only arithmetic), and equals the single-tensor entry points.  45 tensors: two launches."""
    g = torch.Generator().manual_seed(3)
    shapes = [(0, 32, 32, (3, 3, 3)), (0, 64, 32, (3, 3, 3)), (0, 64, 32, (1, 1, 1)), (1, 64, 32, (2, 2, 2)), (1, 128, 64, (1, 2, 2)),
              (0, 96, 64, (1, 3, 3)), (0, 512, 256, (3, 3, 3)), (1, 512, 512, (2, 2, 2)), (0, 40, 24, (3, 3, 3))] * 5
    items, refs = [], []
    for kind, a, b, k in shapes:
        w = torch.randn((a, b, *k), generator=g).cuda()
        t = k[0] * k[1] * k[2]
        co, ci = (a, b) if kind == 0 else (b, a)
        wf = torch.full((t, co, ci), 7.0, dtype=dtype, device="cuda")
        wb = torch.full((t, ci, co), 7.0, dtype=dtype, device="cuda")
        items.append((w, kind, wf, wb))
        flat = w.reshape(a, b, t)
        if kind == 0:       # (Co, Ci, T)
            refs.append((flat.permute(2, 0, 1).to(dtype), flat.permute(2, 1, 0).to(dtype)))
        else:               # (Ci, Co, T)
            refs.append((flat.permute(2, 1, 0).to(dtype), flat.permute(2, 0, 1).to(dtype)))
    ops.pack_weights_multi(items, dtype)
    for (w, kind, wf, wb), (rf, rb) in zip(items, refs):
        assert torch.equal(wf, rf.contiguous()) and torch.equal(wb, rb.contiguous()), (kind, tuple(w.shape))
        one = ops.pack_conv_weight(w, dtype) if kind == 0 else ops.pack_convT_weight(w, dtype)
        assert torch.equal(one[0], wf) and torch.equal(one[1], wb)
    # forward-only plans pack no bwd copy
    w, kind, wf, wb = items[0]
    wf2 = torch.zeros_like(wf)
    ops.pack_weights_multi([(w, kind, wf2, None)], dtype)
    assert torch.equal(wf2, wf)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("pool", [None, (2, 2, 2), (1, 2, 2)])
def test_instnorm_act_bwd_res_masked_gradient_written_once(ops, dtype, pool):
    """rx_instnorm_act_bwd_res (round 3): the residual-block epilogue backward with g' = g * lrelu'(out) written once into the
    residual gradient, optionally adding the next stage's AvgPool gradient on the fly -- against (a) an fp64 restatement and
    (b) the unfused sequence rx_avgpool_bwd(accumulate) + rx_instnorm_act_bwd it replaces.  g sits in the second half of a
    wider buffer (the concat gradient of decoder.py:147, as in the plan)."""
    n, c, dims, slope = 2, 64, (8, 12, 16), 0.01
    y = to_act(ops, rnd((n, c, *dims), dtype, seed=41), dtype)
    out = to_act(ops, rnd((n, c, *dims), dtype, seed=42), dtype)
    g0 = rnd((n, c, *dims), dtype, seed=43, scale=0.1)
    stats = torch.empty((n, c, 2), device="cuda")
    ops.instnorm_stats(y, stats)
    pd = tuple(d // s for d, s in zip(dims, pool)) if pool else None
    pg = to_act(ops, rnd((n, c, *pd), dtype, seed=44, scale=0.1), dtype) if pool else None
    # unfused
    g1 = to_act(ops, g0, dtype, ld=2 * c, c0=c)
    if pool:
        ops.avgpool_bwd(pg, g1, pool, True)
    dy1, dr1 = ops.Act.empty(n, *dims, c, dtype), ops.Act.empty(n, *dims, c, dtype)
    ops.instnorm_act_bwd(g1, y, stats, out, dy1, slope, dr1, False)
    # fused
    g2 = to_act(ops, g0, dtype, ld=2 * c, c0=c)
    dy2, dr2 = ops.Act.empty(n, *dims, c, dtype), ops.Act.empty(n, *dims, c, dtype)
    ops.instnorm_act_bwd_res(g2, y, stats, out, dy2, dr2, slope, pool_dy=pg, pool_stride=pool or (1, 1, 1))
    torch.cuda.synchronize()
    tol = TOL[dtype]
    assert rel(dr2.t.float(), dr1.t.float()) < tol          # (the fused path adds the pool term before rounding)
    assert rel(dy2.t.float(), dy1.t.float()) < 3 * tol
    # fp64 restatement
    gd = to_act(ops, g0, dtype).t.double()
    if pool:
        up = pg.t.double()
        for ax, s in enumerate(pool):
            up = up.repeat_interleave(s, dim=1 + ax)
        gd = gd + up / (pool[0] * pool[1] * pool[2])
    gd = torch.where(out.t.double() > 0, gd, gd * slope)
    mean, rstd = stats[..., 0].double().view(n, 1, 1, 1, c), stats[..., 1].double().view(n, 1, 1, 1, c)
    xh = (y.t.double() - mean) * rstd
    ref = rstd * (gd - gd.mean(dim=(1, 2, 3), keepdim=True) - xh * (gd * xh).mean(dim=(1, 2, 3), keepdim=True))
    assert rel(dr2.t.float().cpu(), gd.float().cpu()) < tol
    assert rel(dy2.t.float().cpu(), ref.float().cpu()) < 3 * tol
    # in place (d_residual == g) is allowed: same results
    g3 = to_act(ops, g0, dtype)
    dy3 = ops.Act.empty(n, *dims, c, dtype)
    ops.instnorm_act_bwd_res(g3, y, stats, out, dy3, g3, slope, pool_dy=pg, pool_stride=pool or (1, 1, 1))
    torch.cuda.synchronize()
    assert torch.equal(g3.t, dr2.t) and torch.equal(dy3.t, dy2.t)
