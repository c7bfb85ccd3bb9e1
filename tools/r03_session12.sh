#!/bin/bash
# round-3 GPU session 12: the torch.distributed.run launch path on a real GPU (world size 1: RCCL init, device
# tensors in the collectives, every --gather mode, every codec) -- what an N > 1 run executes, minus the peers
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/s12; rm -rf $O; mkdir -p $O
P=29540
run() { n=$1; shift; P=$((P+1)); timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $P bench.py --gpus 1 --steps 1 --warmup 1 --gib 1 --only --no-cpu "$@" > $O/$n.json 2> $O/$n.err; echo "$n rc=$? $(tail -c 300 $O/$n.json | tr '\n' ' ' | cut -c1-200)"; }
run lz4_none
run lz4_rccl --gather rccl
run lz4_d2h --gather d2h
run lz4_dec --mode decompress --gather rccl
run zstd --codec zstd --gather rccl
run brotli --codec brotli --gather d2h
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/s12/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d["value"], d["n_gpus"], d.get("gather"), d.get("gather_ms"), d.get("per_rank_ms"), d.get("roundtrip_verified"))
    except Exception as e:
        print(f, "ERR", e, open(f.replace(".json",".err")).read()[-800:])
PY
