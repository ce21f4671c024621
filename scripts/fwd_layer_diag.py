"""Which forward kernel is not reproducible when a second process shares the GPU?  Two processes, each: cfg2 network (128^3, batch 2,
bf16), the same batch every pass, weights re-packed every pass; after each forward EVERY saved tensor of the plan (conv outputs,
statistics, activated outputs, in tape order) is compared with the first pass; the first differing tensor names the kernel.
    python scripts/fwd_layer_diag.py [passes]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import torch  # noqa: E402


def work(tag, passes):
    import mt3d_amd  # noqa: F401
    from types import SimpleNamespace
    from mt3d_amd.builders.build_network_from_config import NetworkFromConfig
    from mt3d_amd.engine import lib
    tasks = {"sheet": {"channels": 1, "activation": "none", "weight": 1, "loss_fn": "BCEDiceLoss", "loss_kwargs": {"alpha": 0.5, "beta": 0.5}}}
    mgr = SimpleNamespace(tasks=tasks, train_patch_size=(128, 128, 128), train_batch_size=2, in_channels=1, vram_max=16.0, autoconfigure=True, model_config={}, verbose=False)
    torch.manual_seed(3)
    net = NetworkFromConfig(mgr).cuda().train()
    net.compute_dtype = torch.bfloat16
    gen = torch.Generator().manual_seed(99)
    x = torch.rand((2, 1, 128, 128, 128), generator=gen).cuda()
    gsc = (torch.randn((2, 1, 128, 128, 128), generator=gen) * 1e-3).cuda()
    BWD = os.environ.get("DIAG_BWD", "1") == "1"
    ref = None
    for i in range(passes):
        net._weights_epoch += 1                 # force the weight re-pack, as after an optimizer step
        out = net(x)
        if BWD:
            (out["sheet"] * gsc).sum().backward()
            for q in net.parameters():
                q.grad = None
        torch.cuda.synchronize()
        plan = next(iter(net._plans.values()))
        items = []
        for ti, tape in enumerate([plan.enc_tape] + plan.dec_tapes):
            for ri, rec in enumerate(tape):
                a = rec.a
                if rec.kind in ("conv", "stem", "convT"):
                    items.append((f"t{ti}.{ri}:{rec.kind}:y{tuple(a['y'].act.t.shape)}", a["y"].act.tensor()))
                elif rec.kind == "inact":
                    items.append((f"t{ti}.{ri}:stats", a["stats"]))
                    items.append((f"t{ti}.{ri}:inact:out{tuple(a['out'].act.t.shape)}", a["out"].act.tensor()))
                elif rec.kind == "pool":
                    items.append((f"t{ti}.{ri}:pool", a["y"].act.tensor()))
        items.append(("logits(clone taken by the autograd function)", out["sheet"]))
        items.append(("logits(engine buffer, read after the device sync)", plan.outputs["sheet"].clone()))
        if ref is None:
            ref = [(n, t.clone()) for n, t in items]
            print(f"[{tag}] pass {i}: reference ({len(items)} tensors)", flush=True)
            continue
        first = None
        nbad = 0
        for (n, t), (_, r) in zip(items, ref):
            if not torch.equal(t, r):
                nbad += 1
                if first is None:
                    d = (t.float() - r.float()).abs()
                    first = f"{n}: {int((d > 0).sum())} elements differ, max |d| {d.max().item():.3e}"
        print(f"[{tag}] pass {i}: {'all equal' if first is None else f'{nbad} tensors differ, FIRST ' + first}", flush=True)


if __name__ == "__main__":
    import torch.multiprocessing as mp
    passes = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    ctx = mp.get_context("spawn")
    ps = [ctx.Process(target=work, args=(f"proc{r}", passes)) for r in range(2)]
    for p in ps:
        p.start()
    for p in ps:
        p.join()
