"""Bit-level A/B of the SE gate kernels between two builds of the library (RX_LIBRARY selects the build).

    python scripts/probes/se_bwd_ab.py dump /tmp/a.pt                       # with librxunet.so
    RX_LIBRARY=$PWD/.../librxunet_base.so python scripts/probes/se_bwd_ab.py dump /tmp/b.pt
    python scripts/probes/se_bwd_ab.py diff /tmp/a.pt /tmp/b.pt

Every output of rx_se_gate_fwd / rx_se_gate_bwd on seeded fp32 inputs, several (C, rd, lines) shapes."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))

CASES = [(2, 32, (6, 5, 16), 8, 1), (2, 64, (4, 4, 12), 4, 1), (1, 256, (3, 3, 4), 16, 1), (2, 512, (2, 2, 2), 32, 1),
         (3, 320, (2, 6, 40), 20, 1), (2, 128, (1, 12, 20), 8, 0), (3, 128, (14, 8, 64), 8, 1)]


def dump(path):
    import mt3d_amd  # noqa: F401
    from mt3d_amd.engine import ops
    res = {}
    f32 = dict(dtype=torch.float32, device="cuda")
    for i, (n, c, dims, rd, keep_x) in enumerate(CASES):
        gen = torch.Generator().manual_seed(11 + i)
        rnd = lambda *s, k=1.0: torch.randn(*s, generator=gen) * k
        to_act = lambda t: ops.Act(t.permute(0, 2, 3, 4, 1).contiguous().cuda())
        ya, ra, ga = to_act(rnd(n, c, *dims)), to_act(rnd(n, c, *dims)), to_act(rnd(n, c, *dims, k=0.1))
        out = ops.Act(torch.empty_like(ya.t))
        dev = [rnd(rd, c, k=0.3).cuda(), rnd(rd, k=0.1).cuda(), rnd(c, rd, k=0.3).cuda(), rnd(c, k=0.1).cuda()]
        se = dict(w1=dev[0], b1=dev[1], w2=dev[2], b2=dev[3], rd=rd, keep_x=keep_x)
        L = dims[2] if keep_x else 1
        stats = torch.empty((n, c, 2), **f32)
        mult, dadd, m12 = torch.empty((n, L, c), **f32), torch.empty((n, L, c), **f32), torch.empty((n, c, 2), **f32)
        pooled, hidden, gate = torch.empty((n, L, c), **f32), torch.empty((n, L, rd), **f32), torch.empty((n, L, c), **f32)
        grads = [torch.zeros_like(t) for t in dev]
        sc = torch.tensor([1.25, 0.5, 1.0][:n], **f32)
        ops.instnorm_stats(ya, stats)
        ops.se_gate_fwd(ya, stats, se, pooled, hidden, gate, mult, sc)
        ops.instnorm_gate_act_fwd(ya, stats, mult, keep_x, out, 0.01, ra)
        ops.se_gate_bwd(ga, ya, stats, out, 0.01, se, pooled, hidden, gate, mult, dadd, m12, *grads, path_scale=sc)
        torch.cuda.synchronize()
        for k, t in dict(pooled=pooled, hidden=hidden, gate=gate, mult=mult, dadd=dadd, m12=m12, dw1=grads[0], db1=grads[1], dw2=grads[2],
                         db2=grads[3]).items():
            res[f"{i}.{k}"] = t.cpu()
    torch.save(res, path)


def diff(a, b):
    a, b = torch.load(a, weights_only=True), torch.load(b, weights_only=True)
    for k in a:
        d = (a[k].double() - b[k].double()).abs().max().item()
        s = a[k].double().abs().max().item()
        print(f"{k:12s} max|a-b| = {d:.3e}   max|a| = {s:.3e}   {'IDENTICAL' if torch.equal(a[k], b[k]) else ''}")


if __name__ == "__main__":
    dump(sys.argv[2]) if sys.argv[1] == "dump" else diff(*sys.argv[2:4])
