"""GPU (-m gpu): the whole hot path -- NetworkFromConfig forward + backward on the HIP engine --
against (a) the golden vectors produced by the REAL reference (tests/golden) and (b) the CPU oracle
run live on the same seeded weights / inputs.

Tolerances (north-star): fp32 mode <= 1e-3 relative with a bit-exact decision map (sign of a 1-channel
logit / argmax over channels); we assert 2e-4 on logits.  bf16 / fp16 are throughput modes: the
reference's OWN bf16-autocast forward drifts 1.1e-2 rel-L2 from its fp32 forward (SURVEY headline 5),
so the bound asserted here is 4e-2 rel-L2 on logits and 1.5e-1 on parameter gradients."""
import pytest
import torch

pytestmark = pytest.mark.gpu

import resenc_oracle as oracle
from golden_cases import CASES
from helpers import forced_dropout, golden_dropout_masks, load_golden, rel_l2


@pytest.fixture(scope="module")
def NetworkFromConfig():
    import mt3d_amd  # noqa: F401
    from mt3d_amd.builders.build_network_from_config import NetworkFromConfig as N
    from mt3d_amd.engine import lib
    lib.require_device()
    return N


def build(NetworkFromConfig, case):
    c = CASES[case]
    mgr = oracle.make_mgr(c["patch"], c["tasks"], c["in_channels"], c["batch"], c["autoconfigure"], c["model_config"])
    torch.manual_seed(c["seed"])
    return NetworkFromConfig(mgr).cuda(), c, mgr


def decision_map(t):
    return (t > 0) if t.shape[1] == 1 else t.argmax(1)


@pytest.mark.parametrize("case", list(CASES))
def test_fp32_matches_reference_golden(NetworkFromConfig, case):
    g = load_golden(case)
    net, c, _ = build(NetworkFromConfig, case)
    x = torch.from_numpy(g["x"]).cuda()
    targets = {k[len("target."):]: torch.from_numpy(v).cuda() for k, v in g.items() if k.startswith("target.")}
    net.train()
    masks = golden_dropout_masks(g)                 # channel dropout: the planes the reference kept in this very step
    with forced_dropout(masks):
        out = net(x)
    if masks:
        plan = next(iter(net._plans.values()))
        assert len(plan._drops) == len(masks) and any((m == 0).any() for m in masks)
    assert list(out.keys()) == list(c["tasks"].keys())
    for k, v in out.items():
        ref = torch.from_numpy(g[f"logits.{k}"])
        assert v.dtype == torch.float32 and v.shape == ref.shape
        assert rel_l2(v.cpu(), ref) < 2e-4, (k, rel_l2(v.cpu(), ref))
        flips = (decision_map(v.cpu()) != decision_map(ref)).sum().item()
        assert flips == 0, f"{flips} decision flips in {k}"
    loss = oracle.train_loss(out, targets, c["tasks"])
    assert abs(loss.item() - float(g["loss"])) < 1e-4
    loss.backward()
    params = dict(net.named_parameters())
    for n, ck in zip(g["param_names"], g["grad_checksums"]):
        p = params[n]
        if ck[0] == 0.0:
            assert p.grad is None, f"{n}: unused deep-supervision head must stay grad-less"
            continue
        assert p.grad is not None, n
        l2 = p.grad.double().norm().item()
        # (biases in front of an InstanceNorm have an exactly-zero gradient: both sides hold ~1e-8 noise)
        assert abs(l2 - ck[2]) <= 1e-3 * ck[2] + 1e-6, (n, l2, ck[2])
        if f"grad.{n}" in g and ck[2] > 1e-6:
            assert rel_l2(p.grad.cpu(), g[f"grad.{n}"]) < 1e-3, n
    net.eval()
    with torch.no_grad():
        ev = net(x)
    for k, v in ev.items():
        assert rel_l2(v.cpu(), g[f"eval.{k}"]) < 2e-4, k


def cosine(a, b):
    a, b = a.double().flatten().cpu(), torch.as_tensor(b).double().flatten()
    return (a @ b / (a.norm() * b.norm()).clamp(min=1e-30)).item()


@pytest.mark.parametrize("dtype,ltol,cos_min", [(torch.bfloat16, 4e-2, 0.90), (torch.float16, 8e-3, 0.985)])
@pytest.mark.parametrize("case", ["auto16_2head", "auto_aniso_bias"])
def test_low_precision_modes_against_oracle(NetworkFromConfig, case, dtype, ltol, cos_min):
    """bf16 / fp16 throughput modes.  Logits: rel-L2 against the fp32 reference golden (the reference's own
    bf16 autocast drifts 1.1e-2).  Gradients: rounding the activations to 8 (11) mantissa bits flips ~1 %
    (~0.1 %) of the LeakyReLU masks per layer, so a tensor-wise rel-L2 bound is not meaningful; asserted
    instead: direction (cosine) and magnitude (l2 ratio) of every parameter gradient."""
    g = load_golden(case)
    net, c, mgr = build(NetworkFromConfig, case)
    x = torch.from_numpy(g["x"]).cuda()
    targets = {k[len("target."):]: torch.from_numpy(v).cuda() for k, v in g.items() if k.startswith("target.")}
    net.train()
    with torch.autocast("cuda", dtype=dtype):      # the reference's way of choosing the compute dtype
        out = net(x)
    for k, v in out.items():
        r = rel_l2(v.cpu(), g[f"logits.{k}"])
        assert r < ltol, (k, r)
    loss = oracle.train_loss(out, targets, c["tasks"])
    assert abs(loss.item() - float(g["loss"])) < 2e-2
    loss.backward()
    params = dict(net.named_parameters())
    for n, ck in zip(g["param_names"], g["grad_checksums"]):
        if ck[0] == 0.0:
            assert params[n].grad is None
            continue
        if ck[2] < 1e-5:
            continue
        l2 = params[n].grad.double().norm().item()
        assert 0.7 < l2 / ck[2] < 1.4, (n, l2, ck[2])
        if f"grad.{n}" in g:
            assert cosine(params[n].grad, g[f"grad.{n}"]) > cos_min, n


def test_fp32_live_oracle_32cube_every_cpu_clean_seed(NetworkFromConfig):
    """the data seed of the two-step test below used to be picked by looking at the ENGINE's error (oracle/seed_margin_gpu.py)
    -- selection on the system under test.  Here the seeds are chosen by the CPU-only criterion alone (oracle fp32 vs its own fp64
    evaluation < 3e-5 at two thread counts, oracle/scan_seeds.py procedure; of data seeds 1..30 and 107 exactly these six qualify, the
    other 25 flip a LeakyReLU mask on the CPU itself) and ALL of them run: logits 2e-4 and identical decisions on every seed; gradients
    within 1e-3 of the oracle on at least four of the six, and never beyond a single mask flip (1e-2).  Measured: 4.3e-6 .. 4.5e-6 on
    seeds 4, 16, 23, 107; 3.6e-3 / 1.5e-3 on seeds 1 / 2, where the engine's fp32 summation order flips one mask the CPU's does not."""
    tasks = {"sheet": {"channels": 1, "activation": "sigmoid", "loss_fn": "BCEDiceLoss",
                       "loss_kwargs": {"alpha": 0.5, "beta": 0.5}}}
    mgr = oracle.make_mgr((32, 32, 32), tasks, 1, 1, True, {})
    torch.manual_seed(5)
    ref = oracle.NetworkFromConfig(mgr).double()      # the fp64 evaluation: no mask ambiguity on the reference side
    torch.manual_seed(5)
    net = NetworkFromConfig(mgr).cuda()
    worst = {}
    for ds in (1, 2, 4, 16, 23, 107):
        x, targets = oracle.synthetic_batch(1, 1, (32, 32, 32), tasks, ds)
        ref.train(); net.train()
        o_r, o_n = ref(x.double()), net(x.cuda())
        assert rel_l2(o_n["sheet"].cpu(), o_r["sheet"].detach()) < 2e-4
        assert torch.equal(o_n["sheet"].cpu() > 0, o_r["sheet"] > 0)
        ref.zero_grad(); net.zero_grad()
        oracle.train_loss(o_r, {k: v.double() for k, v in targets.items()}, tasks).backward()
        oracle.train_loss(o_n, {k: v.cuda() for k, v in targets.items()}, tasks).backward()
        pr, pn = dict(ref.named_parameters()), dict(net.named_parameters())
        worst[ds] = max(rel_l2(pn[n].grad.cpu(), pr[n].grad) for n in pr if pr[n].grad is not None and pr[n].grad.norm() > 1e-6)
    print("32^3 live oracle, worst gradient rel-L2 per CPU-clean data seed:", {k: f"{v:.2e}" for k, v in worst.items()})
    assert sum(v < 1e-3 for v in worst.values()) >= 4, worst
    assert max(worst.values()) < 1e-2, worst


def test_fp32_live_oracle_32cube_two_steps(NetworkFromConfig):
    """a deeper net (32^3 -> 4 stages) against the oracle run live, plus: a second step after an
    in-place parameter update must see the new weights (packed copies are refreshed by version).  (Data seed 107 is one of the four of
    six CPU-clean seeds on which the engine flips no mask either -- the test above runs all six.)"""
    tasks = {"sheet": {"channels": 1, "activation": "sigmoid", "loss_fn": "BCEDiceLoss",
                       "loss_kwargs": {"alpha": 0.5, "beta": 0.5}}}
    mgr = oracle.make_mgr((32, 32, 32), tasks, 1, 1, True, {})
    torch.manual_seed(5)
    ref = oracle.NetworkFromConfig(mgr)
    torch.manual_seed(5)
    net = NetworkFromConfig(mgr).cuda()
    x, targets = oracle.synthetic_batch(1, 1, (32, 32, 32), tasks, 107)
    for step in range(2):
        ref.train(); net.train()
        o_r = ref(x)
        o_n = net(x.cuda())
        assert rel_l2(o_n["sheet"].cpu(), o_r["sheet"].detach()) < 2e-4
        assert torch.equal(o_n["sheet"].cpu() > 0, o_r["sheet"] > 0)
        l_r = oracle.train_loss(o_r, targets, tasks)
        l_n = oracle.train_loss(o_n, {k: v.cuda() for k, v in targets.items()}, tasks)
        ref.zero_grad(); net.zero_grad()
        l_r.backward(); l_n.backward()
        pr, pn = dict(ref.named_parameters()), dict(net.named_parameters())
        for n in pr:
            if pr[n].grad is None:
                assert pn[n].grad is None
                continue
            if pr[n].grad.norm() > 1e-6:
                # step 0: data seed 107 has LeakyReLU mask margin at the initial weights (the oracle's fp32 gradients sit 6e-6
                # from its own fp64 evaluation, oracle/scan_seeds.py procedure) -> the north-star 1e-3 bar.  Step 1 runs on
                # the UPDATED weights, where no seed of 30 keeps every mask clear of zero in a net this deep (the oracle's
                # own fp32-vs-fp64 distance is 2.5e-3..2.9e-2 there, tests/test_oracle_golden.py::
                # test_fp32_gradients_are_mask_discontinuous): its job is the logits check above (new weights are seen)
                tol = 1e-3 if step == 0 else 3e-2
                assert rel_l2(pn[n].grad.cpu(), pr[n].grad) < tol, (step, n, rel_l2(pn[n].grad.cpu(), pr[n].grad))
        with torch.no_grad():
            for n in pr:
                if pr[n].grad is not None:
                    pr[n].sub_(0.05 * pr[n].grad)
                    pn[n].sub_(0.05 * pr[n].grad.cuda())


def test_backward_after_overwritten_buffers_is_refused(NetworkFromConfig):
    net, c, _ = build(NetworkFromConfig, "manual_2in")
    x = torch.rand(1, 2, 16, 16, 16, device="cuda")
    out1 = net(x)
    net(x)                                    # second forward of the same shape reuses the buffers
    with pytest.raises(RuntimeError):
        out1["seg"].sum().backward()


def test_hip_graph_replay_is_bit_identical_to_eager(NetworkFromConfig):
    """opt-in HIP graphs (plan.use_graphs): after two eager passes the forward and backward launch lists are captured
    and replayed.  Same kernels, same order, deterministic reductions -> parameters after 6 SGD steps must be
    IDENTICAL to the eager run's; also covered: gradient accumulation across replays (the graph re-writes the storage
    of the gradients it returned, so a kept .grad is moved out of the way first)."""
    c = CASES["auto16_2head"]

    def run(use_graphs):
        net, _, _ = build(NetworkFromConfig, "auto16_2head")
        net.train()
        x, targets = oracle.synthetic_batch(c["batch"], c["in_channels"], c["patch"], c["tasks"], 7)
        x = x.cuda()
        targets = {k: v.cuda() for k, v in targets.items()}
        opt = torch.optim.SGD([p for p in net.parameters()], lr=0.05)
        losses = []
        for step in range(6):
            with torch.autocast("cuda", dtype=torch.bfloat16):
                out = net(x)
            for plan in net._plans.values():
                plan.use_graphs = use_graphs
            loss = oracle.train_loss(out, targets, c["tasks"])
            loss.backward()
            if step == 4:
                continue          # no optimizer step / zero_grad: step 5 ACCUMULATES into the kept gradients
            opt.step()
            opt.zero_grad(set_to_none=True)
            losses.append(loss.item())
        if use_graphs:
            states = [st for plan in net._plans.values() for st in plan._gstate.values()]
            assert states and all(st.get("graph") is not None for st in states), "graphs were never captured"
        return losses, {n: p.detach().clone() for n, p in net.named_parameters()}

    l_e, p_e = run(False)
    l_g, p_g = run(True)
    assert l_e == l_g
    for n in p_e:
        assert torch.equal(p_e[n], p_g[n]), n


def test_streamed_optimizer_step_is_bit_identical(NetworkFromConfig):
    """engine/streamed_step.py: the fused AdamW update runs on the side stream in forward order, chunk by chunk, each chunk
    followed by the re-pack of its parameters; the next forward waits per parameter.  Same per-tensor arithmetic ->
    parameters (and losses) after 6 steps must equal the plain `optimizer.step()` run bit for bit; an evaluation forward
    (a different plan of the same module) in between must see the updated weights."""
    from mt3d_amd.engine.streamed_step import StreamedOptimizerStep
    c = CASES["auto16_2head"]

    def run(streamed):
        net, _, _ = build(NetworkFromConfig, "auto16_2head")
        x, targets = oracle.synthetic_batch(c["batch"], c["in_channels"], c["patch"], c["tasks"], 7)
        x = x.cuda()
        targets = {k: v.cuda() for k, v in targets.items()}
        params = [p for p in net.parameters()]
        opt = torch.optim.AdamW(params, lr=1e-2, weight_decay=0.01, fused=True)
        stepper = StreamedOptimizerStep(opt, net, chunk_bytes=1 << 16) if streamed else None     # many chunks
        losses, evals = [], []
        for step in range(6):
            net.train()
            with torch.autocast("cuda", dtype=torch.bfloat16):
                out = net(x)
            loss = oracle.train_loss(out, targets, c["tasks"])
            loss.backward()
            torch.nn.utils.clip_grad_norm_(params, 3)
            if streamed:
                stepper.step()
            else:
                opt.step()
            opt.zero_grad(set_to_none=True)
            losses.append(loss.item())
            if step in (2, 4):
                net.eval()
                with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
                    evals.append({k: v.clone() for k, v in net(x).items()})
        if streamed:
            stepper.synchronize()
        torch.cuda.synchronize()
        return losses, evals, {n: p.detach().clone() for n, p in net.named_parameters()}

    l_a, e_a, p_a = run(False)
    l_b, e_b, p_b = run(True)
    assert l_a == l_b
    for a, b in zip(e_a, e_b):
        for k in a:
            assert torch.equal(a[k], b[k]), k
    for n in p_a:
        assert torch.equal(p_a[n], p_b[n]), n


@pytest.mark.parametrize("opt_kind", ["adamw_fused", "adamw_foreach", "sgd_fused"])
def test_training_plan_sees_optimizer_updates(NetworkFromConfig, opt_kind):
    """torch's FUSED optimizers update parameters without bumping Tensor._version: the packed weight copies of the
    training plan must still be refreshed after every step.  After 3 optimizer steps the training plan's forward must
    equal (bit for bit) the forward of a FRESH module loaded with the same state_dict."""
    net, c, _ = build(NetworkFromConfig, "auto16_2head")
    x, targets = oracle.synthetic_batch(c["batch"], c["in_channels"], c["patch"], c["tasks"], 7)
    x = x.cuda()
    targets = {k: v.cuda() for k, v in targets.items()}
    params = [p for p in net.parameters()]
    opt = {"adamw_fused": lambda: torch.optim.AdamW(params, lr=1e-2, fused=True),
           "adamw_foreach": lambda: torch.optim.AdamW(params, lr=1e-2, foreach=True),
           "sgd_fused": lambda: torch.optim.SGD(params, lr=0.05, momentum=0.9, fused=True)}[opt_kind]()

    def fwd(m):
        m.train()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            return m(x)
    first = None
    for _ in range(3):
        out = fwd(net)
        first = first if first is not None else {k: v.detach().clone() for k, v in out.items()}
        oracle.train_loss(out, targets, c["tasks"]).backward()
        opt.step()
        opt.zero_grad(set_to_none=True)
    out = fwd(net)
    fresh, _, _ = build(NetworkFromConfig, "auto16_2head")
    fresh.load_state_dict(net.state_dict())
    ref = fwd(fresh)
    for k in out:
        assert torch.equal(out[k], ref[k]), k
        assert not torch.equal(out[k], first[k]), "the weights never moved"


def test_fp32_training_trajectory_matches_oracle(NetworkFromConfig):
    """three optimizer steps (torch SGD with momentum on both sides, fused on the device) of the engine in fp32 mode
    against the CPU oracle network: the per-step losses must track each other -- an end-to-end check that every
    parameter update reaches the kernels (packed weight copies included) and that the gradients drive the same descent."""
    c = CASES["auto16_2head"]
    mgr = oracle.make_mgr(c["patch"], c["tasks"], c["in_channels"], c["batch"], c["autoconfigure"], c["model_config"])
    torch.manual_seed(c["seed"])
    ref = oracle.NetworkFromConfig(mgr)
    torch.manual_seed(c["seed"])
    net = NetworkFromConfig(mgr).cuda()
    x, targets = oracle.synthetic_batch(c["batch"], c["in_channels"], c["patch"], c["tasks"], c["data_seed"])
    xg = x.cuda()
    tg = {k: v.cuda() for k, v in targets.items()}
    o_ref = torch.optim.SGD([p for p in ref.parameters()], lr=0.05, momentum=0.9)
    o_net = torch.optim.SGD([p for p in net.parameters()], lr=0.05, momentum=0.9, fused=True)
    l_ref, l_net = [], []
    for _ in range(4):
        ref.train(); net.train()
        lr_ = oracle.train_loss(ref(x), targets, c["tasks"])
        ln_ = oracle.train_loss(net(xg), tg, c["tasks"])
        o_ref.zero_grad(); o_net.zero_grad(set_to_none=True)
        lr_.backward(); ln_.backward()
        o_ref.step(); o_net.step()
        l_ref.append(lr_.item()); l_net.append(ln_.item())
    assert l_ref[-1] < l_ref[0] - 1e-3, "the oracle itself did not descend"
    for a, b in zip(l_ref, l_net):
        assert abs(a - b) < 2e-3 * max(1.0, abs(a)), (l_ref, l_net)


def test_planar_concat_plan_matches_interleaved_plan(monkeypatch):
    """RX_PLANAR_MIN_VOXELS lowered so that a 32^3 net takes the planar full-resolution concat: logits and every gradient must
    equal the interleaved plan's (same kernels, same arithmetic, only the addressing of the concat halves differs)."""
    import mt3d_amd  # noqa: F401
    from mt3d_amd.builders.build_network_from_config import NetworkFromConfig
    # two decoders: the second one's concat gets the skip by a copy into ITS planar buffer
    tasks = {"sheet": {"channels": 1, "activation": "none", "loss_fn": "BCEDiceLoss", "loss_kwargs": {"alpha": 0.5, "beta": 0.5}},
             "normals": {"channels": 3, "activation": "none", "loss_fn": "MaskedCosineLoss"}}
    mgr = oracle.make_mgr((32, 32, 64), tasks, 1, 2, True, {})
    x, t = oracle.synthetic_batch(2, 1, (32, 32, 64), tasks, 5)
    res = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("RX_PLANAR_CAT", mode)
        monkeypatch.setenv("RX_PLANAR_MIN_VOXELS", "1")
        torch.manual_seed(7)
        net = NetworkFromConfig(mgr).cuda()
        net.compute_dtype = torch.bfloat16
        out = net(x.cuda())
        loss = oracle.train_loss(out, {k: v.cuda() for k, v in t.items()}, tasks)
        loss.backward()
        plan = next(iter(net._plans.values()))
        cats = [r.a["cat"].act for r in plan.dec_tapes[0] if r.kind == "convT"]
        assert any(c.is_planar_cat for c in cats) == (mode == "1")
        assert all(c.is_planar_cat == (mode == "1") for tape in plan.dec_tapes for c in [r.a["cat"].act for r in tape if r.kind == "convT"][-1:])
        res[mode] = (torch.cat([out["sheet"], out["normals"]], 1).detach().clone(),
                     {n: p.grad.detach().clone() for n, p in net.named_parameters() if p.grad is not None})
    assert torch.equal(res["0"][0], res["1"][0])
    for n in res["0"][1]:
        assert torch.equal(res["0"][1][n], res["1"][1][n]), n


def test_streamed_engine_adamw_is_bit_identical(NetworkFromConfig):
    """StreamedOptimizerStep over EngineAdamW (flat mode): clip coefficient + AdamW table kernel per chunk on the side stream, re-pack
    behind it, next forward waits per parameter -> same parameters and losses, bit for bit, as clip_and_step() on the main stream"""
    from mt3d_amd.engine.streamed_step import StreamedOptimizerStep
    from mt3d_amd.training.optim import EngineAdamW, clip_and_step
    c = CASES["auto16_2head"]

    def run(streamed):
        net, _, _ = build(NetworkFromConfig, "auto16_2head")
        x, targets = oracle.synthetic_batch(c["batch"], c["in_channels"], c["patch"], c["tasks"], 7)
        x = x.cuda()
        targets = {k: v.cuda() for k, v in targets.items()}
        params = [p for p in net.parameters()]
        opt = EngineAdamW(params, model=None, lr=1e-2, weight_decay=0.01)
        stepper = StreamedOptimizerStep(opt, net, chunk_bytes=1 << 16) if streamed else None
        losses, evals = [], []
        for step in range(5):
            net.train()
            with torch.autocast("cuda", dtype=torch.bfloat16):
                out = net(x)
            loss = oracle.train_loss(out, targets, c["tasks"])
            loss.backward()
            if streamed:
                opt.clip_grad_norm(0.5)             # small max_norm: the coefficient is active
                stepper.step()
            else:
                clip_and_step(opt, params, 0.5)
            opt.zero_grad(set_to_none=True)
            losses.append(loss.item())
            if step == 2:
                net.eval()
                with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
                    evals.append({k: v.clone() for k, v in net(x).items()})
        if streamed:
            stepper.synchronize()
        torch.cuda.synchronize()
        return losses, evals, {n: p.detach().clone() for n, p in net.named_parameters()}

    l_a, e_a, p_a = run(False)
    l_b, e_b, p_b = run(True)
    assert l_a == l_b
    for a, b in zip(e_a, e_b):
        for k in a:
            assert torch.equal(a[k], b[k]), k
    for n in p_a:
        assert torch.equal(p_a[n], p_b[n]), n


def test_torch_compile_wrapper_trains_and_keeps_reference_checkpoint_keys(NetworkFromConfig):
    """reference train.py:133 wraps the model in `torch.compile` unconditionally and saves `_orig_mod.`-prefixed keys
    (train.py:250).  The engine's forward is `torch.compiler.disable`d (nothing to trace: ctypes launches on raw pointers),
    so the wrapper must run the same kernels: logits and gradients bit-identical to the bare module, training step works."""
    net, c, _ = build(NetworkFromConfig, "manual_2in")
    g = load_golden("manual_2in")
    x = torch.from_numpy(g["x"]).cuda()
    targets = {k[len("target."):]: torch.from_numpy(v).cuda() for k, v in g.items() if k.startswith("target.")}
    net.train()
    out = net(x)
    oracle.train_loss(out, targets, c["tasks"]).backward()
    bare = {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}
    bare_logits = {k: v.detach().clone() for k, v in out.items()}
    net.zero_grad(set_to_none=True)

    cm = torch.compile(net)
    keys = list(cm.state_dict().keys())
    assert keys and all(k.startswith("_orig_mod.") for k in keys)
    assert [k[len("_orig_mod."):] for k in keys] == list(net.state_dict().keys())
    opt = torch.optim.AdamW(cm.parameters(), lr=1e-3, fused=True)
    with torch.autocast("cuda", enabled=False):
        out = cm(x)
    for k in out:
        assert torch.equal(out[k], bare_logits[k]), k
    loss = oracle.train_loss(out, targets, c["tasks"])
    loss.backward()
    for n, p in net.named_parameters():
        if n in bare:
            assert torch.equal(p.grad, bare[n]), n
    before = net.shared_encoder.stem.convs[0].conv.weight.detach().clone()
    opt.step()
    assert not torch.equal(before, net.shared_encoder.stem.convs[0].conv.weight)
    out2 = cm(x)                                        # the next forward sees the updated weights
    assert not torch.equal(out2[next(iter(out2))], bare_logits[next(iter(out2))])
    cm.eval()
    with torch.no_grad():
        ev = cm(x)
    assert all(v.shape == bare_logits[k].shape for k, v in ev.items())


@pytest.mark.parametrize("variant", ["two_heads_bf16", "se_droppath_fp32", "dropout_bf16", "padded_channels_bf16"])
def test_launch_program_replay_is_bit_identical_to_eager(NetworkFromConfig, monkeypatch, variant):
    """launch programs (default on, include/rxunet.h "launch programs"): after two eager passes every forward / backward
    launch list is recorded by the library while it executes and then replayed from C.  Same launches, same streams,
    deterministic reductions -> losses, evaluation outputs and parameters after 9 steps must be IDENTICAL to a run with
    RX_PROGRAMS=0 (one host call per launch).  Covered on purpose: a different input every step (the program reads a static
    copy), gradient accumulation across replays (a kept .grad is moved off the persistent gradient storage), an evaluation
    forward between training steps (its own plan and the without-re-pack variant), a task left out of the loss (falls
    back to the eager list), SqueezeExcite + DropPath (per-step random factors drawn outside the program)."""
    from golden_cases import _manual, TASKS_2HEAD
    if variant == "two_heads_bf16":
        c = CASES["auto16_2head"]
        patch, tasks, cin, batch, auto, mc, dtype = c["patch"], c["tasks"], c["in_channels"], c["batch"], True, {}, torch.bfloat16
    elif variant == "padded_channels_bf16":   # 24/48/80 features: shadow parameters refreshed before, gradients sliced after the program
        patch, tasks, cin, batch, auto, dtype = (16, 16, 16), TASKS_2HEAD, 1, 2, False, torch.bfloat16
        mc = _manual(features_per_stage=[24, 48, 80], squeeze_excitation=True, conv_bias=True)
    elif variant == "dropout_bf16":       # channel dropout: new masks every step (device generator), read by the program from a fixed buffer
        patch, tasks, cin, batch, auto, dtype = (32, 32, 32), TASKS_2HEAD, 1, 2, True, torch.bfloat16
        mc = {"dropout_op_kwargs": {"p": 0.2}}
    else:
        patch, tasks, cin, batch, auto, dtype = (16, 16, 16), TASKS_2HEAD, 1, 2, False, None
        mc = _manual(squeeze_excitation=True, stochastic_depth_p=0.3)

    def run(programs):
        monkeypatch.setenv("RX_PROGRAMS", "1" if programs else "0")
        mgr = oracle.make_mgr(patch, tasks, cin, batch, auto, mc)
        torch.manual_seed(4)
        net = NetworkFromConfig(mgr).cuda()
        opt = torch.optim.SGD([p for p in net.parameters()], lr=0.05)
        losses, evals = [], []
        for step in range(9):
            x, targets = oracle.synthetic_batch(batch, cin, patch, tasks, 100 + step)
            x, targets = x.cuda(), {k: v.cuda() for k, v in targets.items()}
            net.train()
            with torch.autocast("cuda", dtype=dtype, enabled=dtype is not None):
                out = net(x)
            if step == 7:                         # one task outside the loss: unused-head path
                loss = oracle.train_loss({"sheet": out["sheet"]}, {"sheet": targets["sheet"]}, {"sheet": tasks["sheet"]})
            else:
                loss = oracle.train_loss(out, targets, tasks)
            loss.backward()
            losses.append(loss.item())
            if step == 4:
                continue                          # no optimizer step / zero_grad: step 5 ACCUMULATES into the kept gradients
            opt.step()
            opt.zero_grad(set_to_none=True)
            if step in (2, 5, 6):
                net.eval()
                with torch.no_grad(), torch.autocast("cuda", dtype=dtype, enabled=dtype is not None):
                    evals.append({k: v.clone() for k, v in net(x).items()})
        plans = list(net._plans.values())
        recorded = [st for plan in plans for st in plan._pstate.values() if st.get("prog") is not None]
        if programs:
            assert len(recorded) >= 3 and all(len(st["prog"]) > 10 for st in recorded), "programs were never recorded"
        else:
            assert not recorded
        return losses, evals, {n: p.detach().clone() for n, p in net.named_parameters()}

    l_e, e_e, p_e = run(False)
    l_p, e_p, p_p = run(True)
    assert l_e == l_p
    for a, b in zip(e_e, e_p):
        for k in a:
            assert torch.equal(a[k], b[k]), k
    for n in p_e:
        assert torch.equal(p_e[n], p_p[n]), n


def test_channel_dropout_on_the_fused_paths(NetworkFromConfig):
    """dropout_op_kwargs p > 0 at a size where the statistics come out of conv epilogues, the head is fused into the layer below
    it and the InstanceNorm backward sums ride in data-gradient epilogues (bf16, 64^3): with FORCED masks the planes the masks
    drop are exactly zero after the norm, a bf16 step agrees with the fp32 step on the same masks, the weight gradient of a
    channel dropped in every sample is exactly zero, repeated steps are bit-identical; eval ignores the masks; free-running
    masks keep ~(1 - p) of the planes and change from step to step."""
    from mt3d_amd.engine import plan as plan_mod
    p = 0.3
    patch, tasks = (64, 64, 64), {"sheet": {"channels": 1, "activation": "none", "loss_fn": "BCEDiceLoss",
                                            "loss_kwargs": {"alpha": 0.5, "beta": 0.5}}}
    mgr = oracle.make_mgr(patch, tasks, 1, 2, True, {"dropout_op_kwargs": {"p": p}})
    torch.manual_seed(0)
    net = NetworkFromConfig(mgr).cuda()
    x, t = oracle.synthetic_batch(2, 1, patch, tasks, 3)
    x, t = x.cuda(), {k: v.cuda() for k, v in t.items()}
    gen = torch.Generator().manual_seed(5)
    masks = {}

    def draw(self, d):
        i = next(j for j, e in enumerate(self._drops) if e is d)
        if i not in masks:
            mk = torch.bernoulli(torch.full(tuple(d["keep"].shape), 1 - p), generator=gen)
            if i == 1:
                mk[:, 3] = 0            # one channel of the first block's conv1 dropped in BOTH samples
            masks[i] = mk
        d["keep"].copy_(masks[i])
    orig = plan_mod.Plan._draw_dropout
    plan_mod.Plan._draw_dropout = draw
    try:
        res = {}
        for dt in (torch.float32, torch.bfloat16):
            net.compute_dtype = dt
            runs = []
            for _ in range(3 if dt == torch.bfloat16 else 1):       # (bf16: passes 3.. are program replays)
                net.zero_grad(set_to_none=True)
                out = net(x)
                loss = oracle.train_loss(out, t, tasks)
                loss.backward()
                runs.append((out["sheet"].detach().clone(), {n: q.grad.detach().clone() for n, q in net.named_parameters() if q.grad is not None}))
            for o, g in runs[1:]:
                assert torch.equal(o, runs[0][0]) and all(torch.equal(g[n], runs[0][1][n]) for n in g)
            res[dt] = runs[0]
            plan = [pl for pl in net._plans.values() if pl.dtype == dt and pl.needs_grad][0]
            assert len(plan._drops) == len(masks)
            # the activations after a dropped plane's InstanceNorm are exactly zero
            inacts = [r for tape in [plan.enc_tape] + plan.dec_tapes for r in tape if r.kind == "inact" and r.a["drop"] is not None]
            for i, r in enumerate(inacts):
                if r.a.get("head_dw_fused"):                 # the layer under a task head does not store its activated output (round 3)
                    continue
                a = r.a["out"].act.tensor().float()          # (N, Z, Y, X, C)
                dropped = masks[i] == 0
                if dropped.any():
                    planes = a.abs().amax(dim=(1, 2, 3))      # (N, C)
                    assert (planes[dropped.cuda()] == 0).all(), i
                    assert (planes[~dropped.cuda()] > 0).all(), i
        lo, hi = res[torch.bfloat16], res[torch.float32]
        assert rel_l2(lo[0].cpu(), hi[0].cpu()) < 4e-2
        wname = "shared_encoder.stages.0.blocks.0.conv1.conv.weight"
        assert (hi[1][wname][3] == 0).all() and (lo[1][wname][3] == 0).all() and hi[1][wname].abs().sum() > 0
        for n in hi[1]:
            a_, b_ = lo[1][n].double().flatten(), hi[1][n].double().flatten()
            if b_.norm() > 1e-6 and not n.endswith(".conv.bias"):
                # direction only (bf16 flips ~1 % of the LeakyReLU masks per layer; the deep stages of a random-init net carry
                # gradients of norm 1e-4 that are mostly that noise: 0.79 seen at stage 3)
                assert (a_ @ b_ / (a_.norm() * b_.norm())).item() > (0.9 if ".stages.0." in n or ".stem." in n else 0.6), n
    finally:
        plan_mod.Plan._draw_dropout = orig
    # eval: identity (same output whatever the keep buffers hold)
    net.eval()
    with torch.no_grad():
        e1 = net(x)["sheet"].clone()
        for pl in net._plans.values():
            for d in pl._drops:
                d["keep"].zero_()
        e2 = net(x)["sheet"].clone()
    assert torch.equal(e1, e2) and e1.abs().sum() > 0
    # free-running masks: ~(1 - p) kept, different from step to step
    net.train()
    torch.manual_seed(9)
    fr = []
    for _ in range(2):
        net(x)
        plan = [pl for pl in net._plans.values() if pl.dtype == torch.bfloat16 and pl.needs_grad][0]
        fr.append(torch.cat([d["keep"].flatten() for d in plan._drops]).clone())
    assert not torch.equal(fr[0], fr[1]) and set(fr[0].unique().tolist()) <= {0.0, 1.0}
    assert abs(fr[0].mean().item() - (1 - p)) < 0.05


@pytest.mark.parametrize("case", ["auto16_2head", "odd_channels", "two_d", "wide_in_stem"])
def test_shared_encoder_is_callable_on_its_own(NetworkFromConfig, case):
    """VERDICT r2 "What's missing" #5: upstream `model.shared_encoder(x)` is an ordinary forward (encoder.py:148-158: feature
    extraction); here it runs the encoder part of the engine's plan and returns the per-stage outputs as NCDHW fp32 tensors.
    Checked against the oracle's encoder on the golden case's own input, in eval mode (fp32)."""
    g = load_golden(case)
    net, c, mgr = build(NetworkFromConfig, case)
    x = torch.from_numpy(g["x"]).cuda()
    torch.manual_seed(c["seed"])
    ref = oracle.NetworkFromConfig(mgr).eval()
    with torch.no_grad():
        want = ref.shared_encoder(torch.from_numpy(g["x"]))
    net.eval()
    got = net.shared_encoder(x)
    assert isinstance(got, list) and len(got) == len(want)
    for a, b in zip(got, want):
        assert a.dtype == torch.float32 and tuple(a.shape) == tuple(b.shape)
        assert rel_l2(a.cpu(), b) < 2e-4, rel_l2(a.cpu(), b)
    # the whole network still runs on the same plans afterwards, and a bare container has nothing to run on
    with torch.no_grad():
        ev = net(x)
    for k, v in ev.items():
        assert rel_l2(v.cpu(), g[f"eval.{k}"]) < 2e-4, k
    import copy
    bare = copy.deepcopy(net.shared_encoder)      # a copy of the container alone is not part of any network
    with pytest.raises(RuntimeError):
        bare(x)
    # ... and the decoders likewise (decoder.py:137-162): encoder outputs in, the task's raw logits out -- the same bits as the
    # whole network's logits, since the outputs handed over are exactly what the plan held
    logits = net.forward_logits(x)
    for name in net.task_decoders.keys():
        dl = net.task_decoders[name](got)
        assert dl.dtype == torch.float32 and torch.equal(dl, logits[name]), name
    with pytest.raises(RuntimeError):
        copy.deepcopy(net.task_decoders[name])(got)
    with pytest.raises(ValueError):
        net.task_decoders[name](got[:-1])
    # a deep copy of the NETWORK runs its containers on the copy's own weights
    twin = copy.deepcopy(net)
    with torch.no_grad():
        for q in twin.parameters():
            q.mul_(0.5)
    t_sk = twin.shared_encoder(x)
    assert not torch.equal(t_sk[-1], got[-1])
    assert torch.equal(twin.task_decoders[name](t_sk), twin.forward_logits(x)[name])
    assert torch.equal(net.shared_encoder(x)[-1], got[-1])
    # 16-bit plans (planar concat buffers at full resolution, fused head kernels): same contract
    net.compute_dtype = torch.bfloat16
    sk16 = net.shared_encoder(x)
    lg16 = net.forward_logits(x)
    for name in net.task_decoders.keys():
        assert torch.equal(net.task_decoders[name](sk16), lg16[name]), name
