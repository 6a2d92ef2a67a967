import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _lp_bootstrap  # noqa
import bench
from lightning_pose_amd import ops
dev = torch.device("cuda:0")
for scale in (60, 200, 1000):
    model = bench.build_model(dev, 17, 384)
    sd = model.state_dict()
    for k_ in sd:
        if k_.startswith("head") and k_.endswith("weight"):
            sd[k_] = sd[k_] * scale
    model.load_state_dict(sd)
    batch = bench.synth_batch(dev, 0, 384, 8, 16, 17)
    model.train()
    seen = {}
    orig = ops._decode_prune_auto.after
    def after(stats, n_up):
        seen["s"] = (stats[..., 1].float() / n_up).flatten().quantile(torch.tensor([0.1, 0.5, 0.9], device=stats.device)).tolist()
        seen["hm"] = None
        return orig(stats, n_up)
    ops._decode_prune_auto.after = after
    out = model.training_step(batch, 0)
    torch.cuda.synchronize()
    print("scale", scale, "n_eff / n_up quantiles 10/50/90 %:", seen["s"], flush=True)
    ops._decode_prune_auto.after = orig
