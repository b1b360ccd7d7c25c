#!/bin/bash
# First hardware run of the multi-GPU paths in ONE command (nothing below has ever executed on >= 2 GPUs).
#   tools/multigpu_preflight.sh [out.json]        (on a node with 2 .. 8 MI355X; default gpurun_out/multigpu_preflight.json)
# 1. tests/test_gpu_tile_shard.py::test_rccl_two_gpus_bit_identical  (ncclCommInitRank with 2 ranks, exact + allreduce slab exchange)
# 2. bench.py --gpus N for N in {2, 4, 8} (those the node has) x --shard {volumes, tiles, models}
# Output: one JSON {"gpus_visible", "rccl_test": {...}, "runs": [{"gpus", "shard", "rc", "seconds", "line": <bench JSON line>}, ...],
#                   "scaling_curve": [{"gpus", "value"}]}  -- the `volumes` lines are the scaling curve (weak scaling, whole-job volumes/s);
# `tiles` / `models` are strong scaling of ONE volume and carry `comm` = boa_comm_stats (calls, bytes) to hold against DESIGN section 6's table.
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${1:-$ROOT/gpurun_out/multigpu_preflight.json}
mkdir -p "$(dirname "$OUT")"
LOGD=$(dirname "$OUT")/multigpu_preflight_logs; mkdir -p "$LOGD"
export HSA_ENABLE_IPC_MODE_LEGACY=0 MASTER_ADDR=127.0.0.1
cd "$ROOT"
NGPU=$(python - <<'PY'
import torch
print(torch.cuda.device_count())
PY
)
echo "[preflight] $NGPU GPUs visible"
STEPS=${PREFLIGHT_STEPS:-3}; WARM=${PREFLIGHT_WARMUP:-1}
T0=$(date +%s)
timeout 1200 python -m pytest tests/test_gpu_tile_shard.py -q -m gpu -k "rccl" -x > "$LOGD/rccl_test.log" 2>&1
RCCL_RC=$?
echo "[preflight] RCCL tests rc=$RCCL_RC ($(( $(date +%s) - T0 )) s): $(tail -1 "$LOGD/rccl_test.log")"
RUNS=()
for N in 2 4 8; do
  [ "$N" -le "$NGPU" ] || continue
  for SH in volumes tiles models; do
    TAG="n${N}_${SH}"
    T1=$(date +%s)
    # the driver's own launch line (one rank per GPU over RCCL); extras that only make sense on one GPU are skipped by bench.py itself
    timeout 1800 python -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 --master-port $((29600 + N)) \
        bench.py --gpus "$N" --steps "$STEPS" --warmup "$WARM" --shard "$SH" > "$LOGD/$TAG.json" 2> "$LOGD/$TAG.log"
    RC=$?
    echo "[preflight] --gpus $N --shard $SH rc=$RC ($(( $(date +%s) - T1 )) s)"
    RUNS+=("$N:$SH:$RC:$(( $(date +%s) - T1 ))")
  done
done
python - "$OUT" "$LOGD" "$NGPU" "$RCCL_RC" "${RUNS[@]}" <<'PY'
import json, os, sys
out, logd, ngpu, rccl_rc, runs = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5:]
res = {"gpus_visible": ngpu,
       "rccl_test": {"rc": rccl_rc, "tail": open(os.path.join(logd, "rccl_test.log")).read().strip().splitlines()[-3:]},
       "runs": [], "scaling_curve": []}
for r in runs:
    n, sh, rc, sec = r.split(":")
    line = None
    try:
        txt = open(os.path.join(logd, f"n{n}_{sh}.json")).read().strip().splitlines()
        line = json.loads(txt[-1]) if txt else None
    except Exception as e:  # noqa: BLE001
        line = {"error": f"{type(e).__name__}: {e}"}
    keep = None
    if isinstance(line, dict) and "value" in line:
        keep = {k: line.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "ms_per_step", "scaling", "roofline", "hbm_roofline", "comm")}
        keep["ranks_seen"] = (line.get("config") or {}).get("ranks_seen")
        if sh == "volumes":
            res["scaling_curve"].append({"gpus": int(n), "value": line["value"], "ranks_seen": keep["ranks_seen"]})
    res["runs"].append({"gpus": int(n), "shard": sh, "rc": int(rc), "seconds": int(sec), "line": keep if keep else line})
res["how_to_read"] = ("scaling_curve = the `--shard volumes` lines (one volume per GPU, no data-path collective: the driver's SCALE curve); "
                      "`tiles` / `models` share ONE volume (strong scaling) -- compare `comm.bytes_sent_or_reduced` with DESIGN.md section 6")
json.dump(res, open(out, "w"), indent=1)
print(json.dumps({"wrote": out, "scaling_curve": res["scaling_curve"], "failed": [r for r in res["runs"] if r["rc"] != 0]}))
PY
