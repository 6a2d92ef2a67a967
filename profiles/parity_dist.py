"""Distribution of the keypoint / peak-height deviations of one parity step vs its fixture (device):  python profiles/parity_dist.py c2full bf16-mixed ..."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _lp_bootstrap  # noqa: E402,F401
from tests import test_step_parity as T  # noqa: E402


class G(dict):
    def t(self, k):
        return torch.from_numpy(np.asarray(self[k]))


dev = torch.device("cuda:0")
args = sys.argv[1:]
for name, precision in zip(args[0::2], args[1::2]):
    g = G(np.load(os.path.join(ROOT, "tests", "golden", f"step_{name}.npz"), allow_pickle=False))
    model, out, seen, inp = T._run(name, dev, precision, g)
    rep = {"config": name, "precision": precision}
    want = dict(zip([str(n) for n in g["log_names"]], g["log_values"]))
    got = {k: float(v) for k, v in model.logged.items()}
    rep["scalars_rel"] = {k: round(abs(got[k] / float(v) - 1), 6) for k, v in want.items() if abs(float(v)) > 0 and "weight" not in k.replace("_weighted", "")}
    for meth, tag in (("get_loss_inputs_labeled", "lab"), ("get_loss_inputs_unlabeled", "unl")):
        if meth not in seen:
            continue
        d = seen[meth]
        peak = g.t(f"{tag}_heat_max")
        flat = d["heatmaps_pred"].reshape(peak.shape[0], peak.shape[1], -1)
        mypeak = flat.max(-1).values
        for thr in (0.03, 0.05, 0.07):
            ok = peak >= thr
            ok2 = ok.repeat_interleave(2, dim=1)
            err = (d["keypoints_pred"] - g.t(f"{tag}_keypoints_pred")).abs()[ok2]
            q = torch.tensor([0.5, 0.9, 0.99, 0.999])
            rel = ((mypeak - peak).abs() / peak)[ok]
            rep[f"{tag}_thr{thr}"] = {"n": int(ok.sum()), "kp_mean": round(float(err.mean()), 5), "kp_q50_90_99_999": [round(float(x), 4) for x in err.quantile(q)],
                                      "kp_max": round(float(err.max()), 3), "n_over_1.5px": int((err > 1.5).sum()), "n_over_0.003px": int((err > 3e-3).sum()),
                                      "peak_rel_q50_99": [round(float(x), 4) for x in rel.quantile(torch.tensor([0.5, 0.99]))], "peak_rel_max": round(float(rel.max()), 4),
                                      "argmax_agree": round(float((flat.argmax(-1)[ok] == g.t(f"{tag}_heat_argmax")[ok]).float().mean()), 4)}
            if f"bf16ref_{tag}_keypoints_pred" in g:
                pol = (g.t(f"bf16ref_{tag}_keypoints_pred") - g.t(f"{tag}_keypoints_pred")).abs()[ok2]
                rep[f"{tag}_thr{thr}"]["policy_kp_mean_max"] = [round(float(pol.mean()), 5), round(float(pol.max()), 3)]
    print(json.dumps(rep), flush=True)
