#!/bin/bash
# the N > 1 control flow of bench.py on a one-GPU box: 2 ranks share device 0, the exchange runs over gloo
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
MSPLAT_BENCH_ONE_DEVICE=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 20 --warmup 5 > gpurun_out/r2h_n2.json 2> gpurun_out/r2h_n2.err
tail -c 600 gpurun_out/r2h_n2.err
python - <<PY
import json
d = json.loads([l for l in open("gpurun_out/r2h_n2.json") if l.startswith("{")][-1])
print("n_gpus", d["n_gpus"], "rccl_ranks", d["rccl_ranks"], "fps", d["value"], "gather", d.get("gather"), "also", {k: (v["value"], v["config"]["key"], v.get("gather")) for k, v in d.get("also", {}).items()})
PY
