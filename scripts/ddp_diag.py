"""Diagnostic for the full-size DDP mismatch: repeat forward+backward of the cfg2 network (128^3, batch 2, bf16) on ONE fixed batch and
report, per pass, whether logits and gradients are bit-identical to pass 0.
    python scripts/ddp_diag.py single            one process, no synchroniser
    python scripts/ddp_diag.py buckets           one process, GradSync attached (world 1: bucket views, no collective)
    python scripts/ddp_diag.py two               two processes on the GPU at once, no synchroniser, no process group
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import torch  # noqa: E402


def work(tag, with_sync, passes=8, alt=False):
    import mt3d_amd  # noqa: F401
    from types import SimpleNamespace
    from mt3d_amd.builders.build_network_from_config import NetworkFromConfig
    from mt3d_amd.engine.ddp import GradSync
    from mt3d_amd.training.losses.losses import BCEDiceLoss
    tasks = {"sheet": {"channels": 1, "activation": "none", "weight": 1, "loss_fn": "BCEDiceLoss", "loss_kwargs": {"alpha": 0.5, "beta": 0.5}}}
    mgr = SimpleNamespace(tasks=tasks, train_patch_size=(128, 128, 128), train_batch_size=2, in_channels=1, vram_max=16.0, autoconfigure=True, model_config={}, verbose=False)
    torch.manual_seed(3)
    net = NetworkFromConfig(mgr).cuda().train()
    gen = torch.Generator().manual_seed(99)
    xs = [torch.rand((2, 1, 128, 128, 128), generator=gen).cuda() for _ in range(2)]
    ts = [(torch.rand((2, 1, 128, 128, 128), generator=gen) > 0.8).float().cuda() for _ in range(2)]
    loss = BCEDiceLoss(alpha=0.5, beta=0.5)
    params = list(net.parameters())
    names = [n for n, _ in net.named_parameters()]
    sync = GradSync(bucket_bytes=128 << 20) if with_sync else None
    ref = {}
    for i in range(passes):
        d = (i % 2) if alt else 0
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out = net(xs[d])
            if sync is not None:
                for plan in net._plans.values():
                    plan.grad_sync = sync
            l = loss(out["sheet"], ts[d])
        l.backward()
        torch.cuda.synchronize()
        g = [None if p.grad is None else p.grad.clone() for p in params]
        for p in params:
            p.grad = None
        o = out["sheet"].detach().clone()
        if d not in ref:
            ref[d] = (o, g)
            print(f"[{tag}] pass {i} data {d}: reference", flush=True)
            continue
        same_o = torch.equal(o, ref[d][0])
        bad = [(names[j], ((a - b).norm() / b.norm().clamp(min=1e-30)).item()) for j, (a, b) in enumerate(zip(g, ref[d][1]))
               if a is not None and not torch.equal(a, b)]
        print(f"[{tag}] pass {i} data {d}: logits {'same' if same_o else 'DIFFER'}, {len(bad)} gradients differ {bad[:3]}", flush=True)


if __name__ == "__main__":
    mode = sys.argv[1]
    if mode == "single":
        work("single", False)
        work("single-alt", False, alt=True)
    elif mode == "buckets":
        work("buckets", True)
    elif mode == "two":
        import torch.multiprocessing as mp
        ctx = mp.get_context("spawn")
        ps = [ctx.Process(target=work, args=(f"proc{r}", False)) for r in range(2)]
        for p in ps:
            p.start()
        for p in ps:
            p.join()
