#!/usr/bin/env bash
# The FIRST call on an 8-GPU MI355X node (nothing in this repository has run on more than one GPU: DESIGN.md section 7).  One process per GPU,
# RCCL over xGMI, launched exactly as the driver launches bench.py.  Produces, under gpurun_out/r05_scale/:
#   n{1,2,4,8}.json            the bench line at 1 / 2 / 4 / 8 ranks (weak scaling: 64 labeled + 128 unlabeled frames per GPU)
#   n8_syncbn_gather.json      the same at 8 ranks with the one-shot SyncBatchNorm transport (all-gather + local add; --syncbn-gather)
#   n8_no_sync_bn.json         ... with per-rank BatchNorm statistics (what the 106 SyncBatchNorm messages cost)
#   summary.txt                frames/s, efficiency vs n1, and the asserted facts below
# Asserted per line: n_gpus == N, config.comm_per_step.backend == "nccl" (= RCCL on ROCm), sync_bn_messages == 106 with SyncBatchNorm on,
# grad_buckets == 2, buckets_sent_during_backward > 0.  Does NOT run several ranks on one device (round 4 did, functionally; no value here).
set -euo pipefail
cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r05_scale
mkdir -p "$OUT"
run() {  # N extra-args... -> $OUT/<tag>.json
  local n=$1 tag=$2; shift 2
  local port=$((29500 + RANDOM % 2000))
  if [ "$n" = 1 ]; then
    python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary "$@" | tail -1 > "$OUT/$tag.json"
  else
    python -m torch.distributed.run --nnodes=1 --nproc-per-node "$n" --master-addr 127.0.0.1 --master-port "$port" \
      bench.py --gpus "$n" --steps 20 --warmup 5 --no-cpu-baseline --no-secondary "$@" | tail -1 > "$OUT/$tag.json"
  fi
}
for n in 1 2 4 8; do run $n n$n; done
run 8 n8_syncbn_gather --syncbn-gather
run 8 n8_no_sync_bn --no-sync-bn
python - "$OUT" <<'PY'
import json, sys, os
out = sys.argv[1]
rows = {}
for tag in ("n1", "n2", "n4", "n8", "n8_syncbn_gather", "n8_no_sync_bn"):
    with open(os.path.join(out, tag + ".json")) as fh:
        rows[tag] = json.loads(fh.read())
base = rows["n1"]["value"]
lines = []
for tag, r in rows.items():
    n, comm = r["n_gpus"], r["config"]["comm_per_step"]
    assert n == int(tag[1]), (tag, n)
    if n > 1:
        assert comm["backend"] == "nccl", comm          # torch's "nccl" backend IS RCCL on ROCm
        assert comm["grad_buckets"] == 2 and comm["buckets_sent_during_backward"] > 0, comm
        assert comm["sync_bn_messages"] == (0 if tag.endswith("no_sync_bn") else 106), comm
    lines.append(f"{tag:18s} {n} GPU  {r['value']:9.1f} frames/s  {r['ms_per_step']:7.2f} ms/step  efficiency {r['value'] / (n * base):.3f}  "
                 f"sync_bn_messages {comm['sync_bn_messages']}  transport {comm.get('sync_bn_transport', '-')}")
open(os.path.join(out, "summary.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
