"""algorithmic FLOPs of a bench workload from its (meta-device) plan: 2*V_out*Co*Ci*taps per conv, x3 for a train step"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import torch, bench
import mt3d_amd  # noqa
from mt3d_amd.builders.build_network_from_config import NetworkFromConfig
from mt3d_amd.engine.plan import Plan
args = sys.argv[1:] or ["cfg2=17.5", "inkyaml=28.0", "dumbyaml=37.3", "cfg2se=19.2", "cfg3=16.0", "cfg1=3.3"]
for item in args:
    wl, _, ms = item.partition("=")
    ms = float(ms or 0)
    w = bench.WORKLOADS[wl]
    net = NetworkFromConfig(bench.make_mgr(w))
    B = w["batch"]
    plan = Plan(net.to("meta"), (B, w["in_channels"], *w["patch"]), torch.bfloat16, "meta", needs_grad=True)
    fl, agg = 0, {}
    for tape in [plan.enc_tape] + plan.dec_tapes:
        for rec in tape:
            if rec.kind not in ("conv", "convT", "stem"):
                continue
            a = rec.a
            y = a["y"].act
            vol = 1
            for d in y.dims:
                vol *= d
            ci = a["x"].act.c if "x" in a else w["in_channels"]
            taps = 1
            for t in (a["kernel"] if rec.kind != "convT" else (1, 1, 1)):
                taps *= t
            f = 2 * vol * y.c * ci * taps
            fl += f
            key = (tuple(y.dims[1:]), ci, y.c, taps)
            agg[key] = agg.get(key, [0, 0])
            agg[key][0] += f
            agg[key][1] += 1
    line = f"{wl}: batch {B}, fwd {fl / 1e12:.3f} TFLOP, train step {3 * fl / 1e12:.2f} TFLOP"
    if ms:
        line += f" -> {3 * fl / 1e12 / (ms * 1e-3):.0f} TFLOP/s at {ms} ms"
    print(line)
    if os.environ.get("DETAIL"):
        for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:16]:
            print(f"    out {k[0]} {k[1]}->{k[2]} taps {k[3]}: {v[1]} convs, {v[0] / 1e9:.0f} GFLOP fwd")
