import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, os.path.join(R, "oracle"), os.path.join(R, "tests")]
import torch
import mt3d_amd
from mt3d_amd.builders.build_network_from_config import NetworkFromConfig
from mt3d_amd.engine.streamed_step import StreamedOptimizerStep
import resenc_oracle as oracle
from golden_cases import CASES
c = CASES["auto16_2head"]
def mk():
    mgr = oracle.make_mgr(c["patch"], c["tasks"], c["in_channels"], c["batch"], c["autoconfigure"], c["model_config"])
    torch.manual_seed(c["seed"]); return NetworkFromConfig(mgr).cuda()
x, targets = oracle.synthetic_batch(c["batch"], c["in_channels"], c["patch"], c["tasks"], 7)
x = x.cuda(); targets = {k: v.cuda() for k, v in targets.items()}
def fwd_loss(net):
    net.train()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = net(x)
    return oracle.train_loss(out, targets, c["tasks"])
for streamed in (False, True):
    net = mk(); params = [p for p in net.parameters()]
    opt = torch.optim.AdamW(params, lr=1e-2, weight_decay=0.01, fused=True)
    st = StreamedOptimizerStep(opt, net, chunk_bytes=1 << 16)
    for step in range(2):
        loss = fwd_loss(net); loss.backward()
        torch.nn.utils.clip_grad_norm_(params, 3)
        (st.step() if streamed else opt.step()); torch.cuda.synchronize()
        opt.zero_grad(set_to_none=True)
    with torch.no_grad():
        l_same = fwd_loss(net).item()           # no-grad plan (fresh packs from refresh_packs)
    l_train = fwd_loss(net).item()              # training plan
    net2 = mk(); net2.load_state_dict(net.state_dict())
    l_fresh = fwd_loss(net2).item()
    print("streamed" if streamed else "plain", "nograd-plan", l_same, "train-plan", l_train, "fresh net", l_fresh)

# ---- plain path: which packs are stale right after opt.step() + training forward?
from mt3d_amd.engine import ops
net = mk(); params = [p for p in net.parameters()]
opt = torch.optim.AdamW(params, lr=1e-2, weight_decay=0.01, fused=True)
names = {id(p): n for n, p in net.named_parameters()}
for step in range(2):
    loss = fwd_loss(net); loss.backward()
    torch.nn.utils.clip_grad_norm_(params, 3)
    v0 = {id(p): p._version for p in params}
    opt.step(); torch.cuda.synchronize()
    bumped = sum(1 for p in params if p._version != v0[id(p)])
    print("step", step, "params whose version moved:", bumped, "of", len(params))
    opt.zero_grad(set_to_none=True)
l = fwd_loss(net); torch.cuda.synchronize()
plan = [p for p in net._plans.values() if p.needs_grad][0]
bad = 0
for ent in plan.packs:
    p = ent["param"]
    wf, wb = (ops.pack_conv_weight if ent["kind"] == "conv" else ops.pack_convT_weight)(p.detach(), plan.dtype)
    torch.cuda.synchronize()
    if not torch.equal(wf, ent["w_fwd"]):
        bad += 1
        if bad < 8: print("STALE w_fwd:", names.get(id(p)), "entry version", ent["version"], "param version", p._version)
print("stale packs after the training forward:", bad, "of", len(plan.packs), "loss", l.item())
