#!/bin/bash
# A/B of one environment setting against none, timed step on one box: bash tools/ab_env.sh "XFR_CFG_REMAP=4:15" [models...]
E=$1; shift; MODELS=${@:-resnet101 resnet50_128 lightcnn}
for rep in 1 2; do for m in $MODELS; do for f in "" "$E"; do
env $f python bench.py --model $m --no-cpu-baseline --no-secondary --no-sustained --no-unfused-ref --no-profile --steps 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$m [$f]', round(d['value'],1), round(d['ms_per_step'],3))"
done; done; done
