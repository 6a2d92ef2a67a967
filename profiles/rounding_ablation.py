"""Which rounding points of the bf16-mixed policy cost what.

The reference's own arithmetic (oracle.restated.OracleTracker: torch fp32 on the CPU, pinned against the verbatim reference) is run on a
step fixture's frames four times - no rounding, trunk only, head only, both (= the policy the product implements) - and the quantities the
losses see are compared with the un-rounded run: keypoints (frame px), confidences, heat-map peak height, heat-map MSE, temporal loss.

    python profiles/rounding_ablation.py c2 [c1 c5 ...]  > profiles/archive/r03_rounding_ablation.json        (build container: CPU only)

Round 4, the per-stage table: `--stages` runs one variant per STAGE of the trunk (only the stem's / layer1's / ... /
layer4's rounding points on), one per KIND of rounding point across the trunk (weights / convolution outputs / inner activations / the
residual stream), and the policy with one group switched back to fp32 each ("policy-<group>"), so that a cheap dominant group - if there
is one - shows up as the variant whose removal brings the scalars inside 1e-2:

    python profiles/rounding_ablation.py --stages c2full c2  > profiles/r04_rounding_stages.json"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _lp_bootstrap  # noqa: E402,F401
from lightning_pose_amd.models.backbones._init import seeded_state_dict  # noqa: E402
from oracle import restated as O  # noqa: E402
from tests.golden.step_inputs import RESIDUAL_GAIN, TORCH_SEED, make_step_inputs  # noqa: E402

torch.set_num_threads(int(os.environ.get("THREADS", "8")))
out = {}
STAGES = "--stages" in sys.argv
_names = [a for a in sys.argv[1:] if not a.startswith("--")]
_ALL = ("stem", "layer1", "layer2", "layer3", "layer4")
_KINDS = ("trunk:w", "trunk:z", "trunk:a", "trunk:res")
if STAGES:
    VARIANTS = [("fp32", ())] + [(s_, (s_,)) for s_ in _ALL] + [(k_, (k_,)) for k_ in _KINDS] + [("head_only", ("head",)), ("trunk_only", ("trunk",))]
    VARIANTS += [(f"policy-{s_}", tuple(t for t in _ALL if t != s_) + ("head",)) for s_ in _ALL]
    VARIANTS += [(f"policy-{k_}", tuple(t for t in _KINDS if t != k_) + ("head",)) for k_ in _KINDS]
    VARIANTS += [("policy", ("trunk", "head"))]
else:
    VARIANTS = [("fp32", ()), ("trunk_only", ("trunk",)), ("head_only", ("head",)), ("policy", ("trunk", "head"))]
for name in (_names or ["c2"]):
    g = np.load(os.path.join(ROOT, "tests", "golden", f"step_{name}.npz"))
    inp = make_step_inputs(name, O.generate_heatmaps)
    cfg, batch = inp["cfg"], inp["batch"]
    K, V, HW = cfg["K"], cfg["V"], cfg["HW"]
    torch.manual_seed(TORCH_SEED)
    sd = seeded_state_dict(K * V if False else K, 2)
    for k in list(sd):
        if k.endswith("bn3.weight"):
            sd[k] = torch.full_like(sd[k], RESIDUAL_GAIN)
    for k in g.files:
        if k.startswith("head/"):
            sd["head." + k[len("head/"):]] = torch.from_numpy(g[k])
    semi = cfg["S"] > 0
    lab = batch["labeled"] if semi else batch
    sets = [("lab", lab, "images")] + ([("unl", batch["unlabeled"], "frames")] if semi else [])
    runs = {}
    for tag, rounding in VARIANTS:
        model = O.OracleTracker(K, 2, torch_seed=0)
        model.load_state_dict(sd, strict=True)
        model.train()
        res = {}
        with torch.no_grad():
            for st, bd, key in sets:
                x = bd[key].reshape(-1, 3, HW, HW)
                h = O.forward_bf16_policy(model, x, rounding=rounding)
                h = h.reshape(bd[key].shape[0], -1, h.shape[-2], h.shape[-1])
                kp, conf = O.soft_argmax(h, 2, 1000.0)
                if st == "unl":
                    kp = O.undo_affine(kp, bd["transforms"], bool(bd.get("is_multiview", False)))
                res[st] = dict(kp=O.model_to_frame(kp, HW, HW, bd["bbox"], V), conf=conf, peak=h.flatten(2).max(-1).values,
                               heat=h if st == "lab" else None)
            if "heatmaps" in lab:
                t = lab["heatmaps"]
                keep = t.flatten(2).sum(-1) > 0
                res["hm_mse"] = float((((res["lab"]["heat"] - t) ** 2)[keep]).mean() * t.shape[-1] * t.shape[-2])
                res["rmse"] = float(O.rmse_loss(lab["keypoints"], res["lab"]["kp"]))
            if semi:
                kpu = res["unl"]["kp"].reshape(res["unl"]["kp"].shape[0], -1, 2)
                res["temporal_raw"] = float((kpu[1:] - kpu[:-1]).norm(dim=-1).mean())
                from tests.golden.step_inputs import TEMPORAL  # the fixture's own loss settings (epsilon, threshold)
                res["temporal_loss"] = float(O.temporal_loss(res["unl"]["kp"], res["unl"]["conf"], TEMPORAL["epsilon"], TEMPORAL["prob_threshold"]))
        res["lab"]["heat"] = None
        runs[tag] = res
        print(name, tag, {k_: v_ for k_, v_ in res.items() if isinstance(v_, float)}, file=sys.stderr, flush=True)
    ref = runs["fp32"]
    rep = {}
    for tag in [t for t, _ in VARIANTS if t != "fp32"]:
        r = runs[tag]
        row = {}
        for st in ("lab", "unl"):
            if st not in r:
                continue
            ok = ref[st]["peak"] >= 0.03
            ok2 = ok.repeat_interleave(2, dim=1)
            err = (r[st]["kp"] - ref[st]["kp"]).abs()[ok2]
            row[f"{st}_kp_px_mean"], row[f"{st}_kp_px_max"] = round(float(err.mean()), 4), round(float(err.max()), 3)
            row[f"{st}_kp_px_p99"] = round(float(err.quantile(0.99)), 3)
            row[f"{st}_conf_abs_max"] = round(float((r[st]["conf"] - ref[st]["conf"]).abs()[ok].max()), 4)
            row[f"{st}_peak_rel_max"] = round(float(((r[st]["peak"] - ref[st]["peak"]).abs() / ref[st]["peak"])[ok].max()), 4)
        if "hm_mse" in r:
            row["heatmap_mse_rel"] = round(abs(r["hm_mse"] / ref["hm_mse"] - 1), 5)
        if "temporal_raw" in r:
            row["temporal_rel"] = round(abs(r["temporal_raw"] / ref["temporal_raw"] - 1), 5)
        if "temporal_loss" in r:
            row["temporal_loss_rel"] = round(abs(r["temporal_loss"] / ref["temporal_loss"] - 1), 5)
        if "rmse" in r:
            row["rmse_rel"] = round(abs(r["rmse"] / ref["rmse"] - 1), 5)
        rep[tag] = row
    out[name] = rep
    print(name, json.dumps(rep), file=sys.stderr, flush=True)
print(json.dumps(out, indent=1))
