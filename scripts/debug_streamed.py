import sys, os
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"), os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests")]
import torch
import mt3d_amd
from mt3d_amd.builders.build_network_from_config import NetworkFromConfig
from mt3d_amd.engine.streamed_step import StreamedOptimizerStep
import resenc_oracle as oracle
from golden_cases import CASES
c = CASES["auto16_2head"]
def run(streamed, nsteps, sync_each=False):
    mgr = oracle.make_mgr(c["patch"], c["tasks"], c["in_channels"], c["batch"], c["autoconfigure"], c["model_config"])
    torch.manual_seed(c["seed"]); net = NetworkFromConfig(mgr).cuda()
    x, targets = oracle.synthetic_batch(c["batch"], c["in_channels"], c["patch"], c["tasks"], 7)
    x = x.cuda(); targets = {k: v.cuda() for k, v in targets.items()}
    params = [p for p in net.parameters()]
    opt = torch.optim.AdamW(params, lr=1e-2, weight_decay=0.01, fused=True)
    st = StreamedOptimizerStep(opt, net, chunk_bytes=1 << 16) if streamed else None
    losses = []
    for step in range(nsteps):
        net.train()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out = net(x)
        loss = oracle.train_loss(out, targets, c["tasks"]); loss.backward(); losses.append(loss.item())
        torch.nn.utils.clip_grad_norm_(params, 3)
        (st.step() if streamed else opt.step())
        if sync_each: torch.cuda.synchronize()
        opt.zero_grad(set_to_none=True)
    torch.cuda.synchronize()
    print("streamed" if streamed else "plain", "sync" if sync_each else "", losses)
    return {n: p.detach().clone() for n, p in net.named_parameters()}, opt
a, oa = run(False, 4); b, ob = run(True, 4); c2, oc = run(True, 4, sync_each=True)
bad = [n for n in a if not torch.equal(a[n], b[n])]
bad2 = [n for n in a if not torch.equal(a[n], c2[n])]
print("after 2 steps: differing params streamed:", len(bad), "of", len(a), bad[:6])
print("with sync after each step:", len(bad2), bad2[:6])
for n in bad[:3]:
    print(n, (a[n] - b[n]).abs().max().item(), a[n].abs().max().item())

# ---- are the packs fresh after a streamed step?
from mt3d_amd.engine import ops
mgr = oracle.make_mgr(c["patch"], c["tasks"], c["in_channels"], c["batch"], c["autoconfigure"], c["model_config"])
torch.manual_seed(c["seed"]); net = NetworkFromConfig(mgr).cuda()
x, targets = oracle.synthetic_batch(c["batch"], c["in_channels"], c["patch"], c["tasks"], 7)
x = x.cuda(); targets = {k: v.cuda() for k, v in targets.items()}
params = [p for p in net.parameters()]
opt = torch.optim.AdamW(params, lr=1e-2, weight_decay=0.01, fused=True)
st = StreamedOptimizerStep(opt, net, chunk_bytes=1 << 16)
for step in range(2):
    net.train()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = net(x)
    loss = oracle.train_loss(out, targets, c["tasks"]); loss.backward()
    torch.nn.utils.clip_grad_norm_(params, 3)
    st.step(); torch.cuda.synchronize()
    opt.zero_grad(set_to_none=True)
plan = [p for p in net._plans.values() if p.needs_grad][0]
bad = 0
names = {id(p): n for n, p in net.named_parameters()}
for ent in plan.packs:
    p = ent["param"]
    w = p.detach()
    if ent["kind"] == "conv":
        wf, wb = ops.pack_conv_weight(w, plan.dtype)
    else:
        wf, wb = ops.pack_convT_weight(w, plan.dtype)
    torch.cuda.synchronize()
    okf = torch.equal(wf, ent["w_fwd"]); okb = ent["w_bwd"] is None or torch.equal(wb, ent["w_bwd"])
    fresh = ent["version"] == p._version
    if not (okf and okb):
        bad += 1
        if bad < 6: print("STALE PACK", names.get(id(p)), okf, okb, "marked fresh:", fresh, "grad:", p.grad is not None)
print("stale packs:", bad, "of", len(plan.packs))
