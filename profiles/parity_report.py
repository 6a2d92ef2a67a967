"""Measured end-to-end parity of one training step vs the verbatim reference's golden steps (tests/golden/step_*.npz), per config and
precision, WITHOUT asserting: the numbers DESIGN.md section 5 quotes.   python profiles/parity_report.py [cpu-emu] > profiles/rNN_parity.txt"""
import json
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _lp_bootstrap  # noqa: E402,F401
from tests.conftest import Golden  # noqa: E402
from tests import test_step_parity as P  # noqa: E402

dev = torch.device("cuda:0")
names = [a for a in sys.argv[1:] if a != "cpu-emu"] or ["s64", "c1", "c2", "c5", "c5v4", "c2full", "c4", "c4full"]
names = [n for n in names if os.path.exists(os.path.join(ROOT, "tests", "golden", f"step_{n}.npz"))]
if len(sys.argv) > 1 and sys.argv[1] == "cpu-emu":   # build container: the emulated kernels, the small case only
    from lightning_pose_amd import _lib, ops
    from tests.hipemu import emu
    _lib._lib = emu.emu_lib()
    ops.require_device = lambda *a: None
    ops.require_device_type = lambda d: None
    ops._stream = lambda: None
    dev, names = torch.device("cpu"), ["s64"]

for name in names:
    with np.load(os.path.join(ROOT, "tests", "golden", f"step_{name}.npz"), allow_pickle=False) as z:
        g = Golden({k: z[k] for k in z.files})
    for precision in ("fp32", "bf16-mixed"):
        model, out, seen, inp = P._run(name, dev, precision, g)
        rec = {"config": name, "precision": precision}
        want = dict(zip([str(n) for n in g["log_names"]], g["log_values"]))
        got = {k: float(v) for k, v in model.logged.items()}
        rec["scalars"] = {k: {"got": round(got[k], 7), "ref": round(float(v), 7), "rel": round(abs(got[k] - float(v)) / (abs(float(v)) + 1e-30), 6)}
                          for k, v in want.items() if "weight" not in k.replace("_weighted", "") and k != "total_unsupervised_importance"}
        for meth, tag in (("get_loss_inputs_labeled", "lab"), ("get_loss_inputs_unlabeled", "unl")):
            if meth not in seen:
                continue
            d = seen[meth]
            peak = g.t(f"{tag}_heat_max")
            for sel, ok in (("all", torch.ones_like(peak, dtype=torch.bool)), ("peaked", peak >= P.PEAK_MIN)):
                ok2 = ok.repeat_interleave(2, dim=1)
                err = (d["keypoints_pred"] - g.t(f"{tag}_keypoints_pred")).abs()[ok2]
                cerr = (d["confidences"] - g.t(f"{tag}_confidences")).abs()[ok]
                flat = d["heatmaps_pred"].reshape(peak.shape[0], peak.shape[1], -1)
                rec[f"{tag}_{sel}"] = {"maps": int(ok.sum()), "kp_err_px_max": round(float(err.max()), 5), "kp_err_px_mean": round(float(err.mean()), 5),
                                       "kp_err_px_p99": round(float(err.quantile(0.99)), 5), "conf_err_max": round(float(cerr.max()), 6),
                                       "conf_err_p99": round(float(cerr.quantile(0.99)), 6), "kp_frac_over_1px": round(float((err > 1.0).float().mean()), 5),
                                       "kp_frac_over_1p5px": round(float((err > 1.5).float().mean()), 5),
                                       "peak_rel_err_p99": round(float(((flat.max(-1).values - peak).abs() / peak)[ok].quantile(0.99)), 5),
                                       "peak_rel_err_max": round(float(((flat.max(-1).values - peak).abs() / peak)[ok].max()), 5),
                                       "argmax_equal": round(float((flat.argmax(-1) == g.t(f"{tag}_heat_argmax"))[ok].float().mean()), 4)}
        grads = {n_: p_.grad.detach().float().cpu() for n_, p_ in model.named_parameters() if p_.grad is not None}
        gg = {}
        for k in [k for k in g if k.startswith("grad/")]:
            a, b = grads[k[len("grad/"):]].reshape(-1), g.t(k).reshape(-1)
            if float(b.norm()) > 1e-6:
                gg[k[5:]] = {"cos": round(float(F.cosine_similarity(a, b, dim=0)), 6), "norm_ratio": round(float(a.norm() / b.norm()), 5)}
        rec["grads"] = gg
        if "grad_names" in g:
            norms = dict(zip([str(n) for n in g["grad_names"]], g["grad_norms"]))
            rec["grad_norm_worst_rel"] = round(max(abs(float(grads[n_].norm()) / (w + 1e-30) - 1.0) for n_, w in norms.items() if w > 1e-6), 5)
        print(json.dumps(rec), flush=True)
        del model
