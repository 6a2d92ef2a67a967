"""The optional "fp32 residual stream" policy (Engine.residual_fp32, LP_RESIDUAL_FP32=1) against the benchmarked bf16-mixed policy, on the device:
every logged loss scalar of the BASELINE-config step fixtures (relative error against the verbatim reference's golden step) and the labeled
keypoints' mean / 99th-percentile error, both policies side by side.      python profiles/residual_fp32_report.py [c1 c5 c5v4 c2 c2full]"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _lp_bootstrap  # noqa: E402,F401
import tests.conftest  # noqa: E402,F401
from tests import test_step_parity as P  # noqa: E402

names = [a for a in sys.argv[1:]] or ["c1", "c5", "c5v4", "c2", "c2full"]
out = {}
for name in names:
    with np.load(os.path.join(ROOT, "tests", "golden", f"step_{name}.npz"), allow_pickle=False) as z:
        g = tests.conftest.Golden({k: z[k] for k in z.files})
    want = dict(zip([str(n) for n in g["log_names"]], [float(v) for v in g["log_values"]]))
    row = {}
    for flag in ("0", "1"):
        os.environ["LP_RESIDUAL_FP32"] = flag
        model, _, seen, _ = P._run(name, torch.device("cuda:0"), "bf16-mixed", g)
        got = {k: float(v) for k, v in model.logged.items()}
        rel = {k: abs(got[k] - v) / (abs(v) + 1e-30) for k, v in want.items() if "weight" not in k.replace("_weighted", "") and k != "total_unsupervised_importance"}
        kp = seen["get_loss_inputs_labeled"]["keypoints_pred"].reshape(-1)
        ref = g.t("lab_keypoints_pred").reshape(-1) if "lab_keypoints_pred" in g else None
        rec = {"scalars_rel": {k: float(f"{v:.4g}") for k, v in rel.items()}, "misses_1e-2": sorted(k for k, v in rel.items() if v > 1e-2)}
        if ref is not None:
            err = (kp - ref).abs()
            err = err[torch.isfinite(err)]
            rec["lab_kp_px_mean"], rec["lab_kp_px_p99"] = round(float(err.mean()), 4), round(float(err.quantile(0.99)), 3)
        row["residual_fp32" if flag == "1" else "bf16-mixed"] = rec
        del model
        torch.cuda.empty_cache()
    out[name] = row
    print(name, json.dumps(row), flush=True)
os.environ.pop("LP_RESIDUAL_FP32", None)
