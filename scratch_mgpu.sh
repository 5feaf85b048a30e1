#!/bin/bash
# two ranks on ONE GPU over gloo: exercises bench.py's multi-process path (init, arena broadcast, max-over-ranks timing)
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 WORLD_SIZE=2 XFR_DIST_BACKEND=gloo XFR_FORCE_DEVICE=0
RANK=1 LOCAL_RANK=1 python bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline --batch 16 > /tmp/r1.log 2>&1 &
RANK=0 LOCAL_RANK=0 python bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline --batch 16 2>&1 | tail -2 | cut -c1-400
wait
echo "rank1 tail:"; tail -2 /tmp/r1.log | cut -c1-300
