#!/usr/bin/env bash
# Multi-GPU training launcher with the reference's argument order (tools/dist_train.sh:1-17 there):
#     bash tools/dist_train.sh CONFIG GPUS [train.py args ...]
# env: PORT (default 29500), MASTER_ADDR (default 127.0.0.1), NNODES / NODE_RANK (default 1 / 0).
# One process per GPU over RCCL.  Single node (the case the hot path covers) goes through tools/dist_launch.py, which starts the
# ranks as fresh interpreters without touching the GPU itself; NNODES > 1 hands over to torch.distributed.run.
set -euo pipefail
CONFIG=$1
GPUS=$2
NNODES=${NNODES:-1}
NODE_RANK=${NODE_RANK:-0}
PORT=${PORT:-29500}
MASTER_ADDR=${MASTER_ADDR:-"127.0.0.1"}
HERE="$(cd "$(dirname "$0")" && pwd)"
export PYTHONPATH="$HERE/..":${PYTHONPATH:-}
export HSA_ENABLE_IPC_MODE_LEGACY=${HSA_ENABLE_IPC_MODE_LEGACY:-0}
if [ "$NNODES" = "1" ]; then
    MASTER_ADDR=$MASTER_ADDR python "$HERE/dist_launch.py" --nproc "$GPUS" --port "$PORT" \
        "$HERE/train.py" "$CONFIG" --launcher pytorch "${@:3}"
else
    python -m torch.distributed.run --nnodes="$NNODES" --node_rank="$NODE_RANK" --master_addr="$MASTER_ADDR" \
        --nproc_per_node="$GPUS" --master_port="$PORT" "$HERE/train.py" "$CONFIG" --launcher pytorch "${@:3}"
fi
