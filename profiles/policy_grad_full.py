"""The bf16-mixed POLICY's own gradients at a step fixture (reference arithmetic rounded where the product rounds, torch CPU, autograd rounding
the gradients at the same places): how far are the policy's stem / head gradients and logged scalars from the fp32 fixture's?  This is the
yardstick for the product's bf16-mixed deviations at BASELINE's real batch.      python profiles/policy_grad_full.py c2full > profiles/archive/r03_policy_grad_c2full.json"""
import json
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _lp_bootstrap  # noqa: E402,F401
from lightning_pose_amd.models.backbones._init import seeded_state_dict  # noqa: E402
from oracle import restated as O  # noqa: E402
from tests.golden.step_inputs import PCA_LOG_WEIGHT, RESIDUAL_GAIN, TEMPORAL, TORCH_SEED, make_step_inputs  # noqa: E402

torch.set_num_threads(int(os.environ.get("THREADS", "6")))
name = sys.argv[1] if len(sys.argv) > 1 else "c2"
g = np.load(os.path.join(ROOT, "tests", "golden", f"step_{name}.npz"))
inp = make_step_inputs(name, O.generate_heatmaps)
cfg, batch = inp["cfg"], inp["batch"]
K = cfg["K"]
torch.manual_seed(TORCH_SEED)
sd = seeded_state_dict(K, 2)
for k in list(sd):
    if k.endswith("bn3.weight"):
        sd[k] = torch.full_like(sd[k], RESIDUAL_GAIN)
for k in g.files:
    if k.startswith("head/"):
        sd["head." + k[len("head/"):]] = torch.from_numpy(g[k])
dev = torch.device(os.environ.get("DEVICE", "cpu"))   # (the full-batch fixtures need ~80 GB of fp32 autograd state: DEVICE=cuda:0 on the GPU box)
model = O.OracleTracker(K, 2, torch_seed=0)
model.load_state_dict(sd, strict=True)
model.train()
model.to(dev)


def _to(d):
    return {k: (_to(v) if isinstance(v, dict) else v.to(dev) if torch.is_tensor(v) else v) for k, v in d.items()}


batch = _to(batch)
if dev.type != "cpu":
    torch.set_default_device(dev)   # the oracle's index grids / constants are created where the data lives


class Policy:
    downsample_factor = model.downsample_factor

    def __call__(self, x):
        h = O.forward_bf16_policy(model, x.reshape(-1, 3, x.shape[-2], x.shape[-1]))
        return h.reshape(x.shape[0], -1, h.shape[-2], h.shape[-1])


unsup = {"temporal": dict(TEMPORAL),
         "pca_singleview": {"log_weight": PCA_LOG_WEIGHT, "mean": torch.from_numpy(g["pca_mean"]).to(dev),
                            "kept_eigenvectors": torch.from_numpy(g["pca_kept"]).to(dev), "epsilon": float(g["pca_eps"]),
                            "columns": inp["cols"]}}
loss, logs = O.training_step(Policy(), batch, unsup, 1.0)
loss.backward()
want = dict(zip([str(n) for n in g["log_names"]], g["log_values"]))
out = {"config": name, "scalars_rel_vs_fp32": {}}
for k, v in logs.items():
    if k in want and abs(float(want[k])) > 0 and "weight" not in k.replace("_weighted", ""):
        out["scalars_rel_vs_fp32"][k] = round(abs(float(v) / float(want[k]) - 1), 6)
grads = {n_: p_.grad.float().cpu() for n_, p_ in model.named_parameters() if p_.grad is not None}
for k in g.files:
    if k.startswith("grad/"):
        a, b = grads[k[len("grad/"):]].reshape(-1), torch.from_numpy(g[k]).reshape(-1)
        if float(b.norm()) < 1e-6:
            continue
        out[k] = {"cos": round(float(F.cosine_similarity(a, b, dim=0)), 5), "norm_ratio": round(float(a.norm() / b.norm()), 4)}
norms = dict(zip([str(n) for n in g["grad_names"]], g["grad_norms"]))
worst = max((abs(float(grads[n_].norm()) / (w + 1e-30) - 1.0), n_) for n_, w in norms.items() if w > 1e-6 and n_ in grads)
out["worst_grad_norm_ratio_minus_1"] = [round(worst[0], 4), worst[1]]
print(json.dumps(out, indent=1))
