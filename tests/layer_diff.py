"""TEST-SIDE DEBUGGING TOOL (imports the randomized test's draws).  First layer where the 16-bit engine leaves the fp32 engine: python tests/layer_diff.py <fuzz index> [bf16|fp16] [medium]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]      # (test_fuzz_gpu imports the oracle)
import torch
import mt3d_amd  # noqa
from mt3d_amd.builders.build_network_from_config import NetworkFromConfig
from types import SimpleNamespace
import test_fuzz_gpu as fz
i = int(sys.argv[1]); dt = {"bf16": torch.bfloat16, "fp16": torch.float16}[sys.argv[2] if len(sys.argv) > 2 else "bf16"]
medium = len(sys.argv) > 3 and sys.argv[3] == "medium"          # (RX_FUZZ_SEED selects the draw set, as in the test)
c = (fz.medium_configs() if medium else fz.configs())[i]
print(c)
mgr = SimpleNamespace(tasks=c["tasks"], train_patch_size=tuple(c["patch"]), train_batch_size=c["batch"], in_channels=c["cin"], vram_max=16.0,
                      autoconfigure=False, model_config=dict(c["mc"]), verbose=False)
torch.manual_seed((300 if medium else 100) + i)
net = NetworkFromConfig(mgr).cuda()
g = torch.Generator().manual_seed((40 if medium else 7) + i)
x = torch.rand((c["batch"], c["cin"], *c["patch"]), generator=g).cuda()
acts = {}
for d in (torch.float32, dt):
    net.compute_dtype = d
    out = net(x)
    plan = [p for p in net._plans.values() if p.dtype == d][0]
    rows = []
    for tape in [plan.enc_tape] + plan.dec_tapes:
        for r in tape:
            if r.kind in ("conv", "stem", "convT"):
                rows.append((r.kind + ":y", r.a["y"].act.tensor().float().clone(), plan.fwd and None))
            elif r.kind == "inact":
                rows.append(("inact:out", r.a["out"].act.tensor().float().clone(), r.a["stats"].clone()))
            elif r.kind == "pool":
                rows.append(("pool:y", r.a["y"].act.tensor().float().clone(), None))
    acts[d] = (rows, {k: v.clone() for k, v in out.items()})
ra, rb = acts[torch.float32][0], acts[dt][0]
for j, ((ka, ta, sa), (kb, tb, sb)) in enumerate(zip(ra, rb)):
    e = ((ta - tb).norm() / ta.norm().clamp_min(1e-30)).item()
    extra = ""
    if sa is not None:
        extra = f" | rstd max fp32 {sa[..., 1].max().item():.3e} 16-bit {sb[..., 1].max().item():.3e}"
    print(f"{j:3d} {ka:10s} shape {tuple(ta.shape)} rel diff {e:.3e} |fp32| {ta.norm().item():.3e}{extra}")
for k in acts[dt][1]:
    a, b = acts[torch.float32][1][k], acts[dt][1][k]
    print("logits", k, ((a - b).norm() / a.norm()).item())
