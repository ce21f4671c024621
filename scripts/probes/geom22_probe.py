import sys, os
sys.path[:0] = ['/root/repo', '/root/repo/oracle', '/root/repo/tests']
import torch
import resenc_oracle as oracle
import test_fuzz_gpu as T
import mt3d_amd
from mt3d_amd.builders.build_network_from_config import NetworkFromConfig
i = 22
c = T.geometry_configs()[i]
print("default threads", torch.get_num_threads(), "affinity", len(os.sched_getaffinity(0)))
mgr = oracle.make_mgr(c["patch"], c["tasks"], c["cin"], c["batch"], False, c["mc"])
def engine():
    torch.manual_seed(100 + i)
    net = NetworkFromConfig(mgr).cuda(); net.compute_dtype = torch.float32
    x, t = oracle.synthetic_batch(c["batch"], c["cin"], c["patch"], c["tasks"], 7 + i)
    o = net(x.cuda()); tt = T.targets_for(c, o, 7 + i)
    oracle.train_loss(o, {k: v.cuda() for k, v in tt.items()}, c["tasks"]).backward()
    return {n: p.grad.detach().cpu().double() for n, p in net.named_parameters() if p.grad is not None}
def orc(dtype, thr):
    torch.set_num_threads(thr)
    torch.manual_seed(100 + i)
    ref = oracle.NetworkFromConfig(mgr).to(dtype)
    x, t = oracle.synthetic_batch(c["batch"], c["cin"], c["patch"], c["tasks"], 7 + i)
    o = ref(x.to(dtype)); tt = T.targets_for(c, o, 7 + i)
    oracle.train_loss(o, {k: v.to(dtype) for k, v in tt.items()}, c["tasks"]).backward()
    return {n: p.grad.detach().double() for n, p in ref.named_parameters() if p.grad is not None}
g1, g2 = engine(), engine()
print("engine deterministic:", all(torch.equal(g1[n], g2[n]) for n in g1))
g64 = orc(torch.float64, 16)
name = 'shared_encoder.stages.2.0.convs.0.conv.weight'
def cos(a, b): a, b = a.flatten(), b.flatten(); return (a @ b / (a.norm() * b.norm())).item()
print("engine vs fp64:", cos(g1[name], g64[name]), "worst over params", min(cos(g1[n], g64[n]) for n in g64 if g64[n].norm() > 1e-6))
for thr in (1, 4, 8, 16, 32):
    g32 = orc(torch.float32, thr)
    print(f"oracle fp32 @{thr} threads vs fp64: {cos(g32[name], g64[name]):.5f}   engine vs this oracle: {cos(g1[name], g32[name]):.5f}")
