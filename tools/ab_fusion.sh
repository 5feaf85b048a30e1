#!/bin/bash
# A/B of one xfr_engine_set_epilogue_fusion level against the default on the same box: bash tools/ab_fusion.sh LEVEL [models...]
L=${1:-35}; shift; MODELS=${@:-resnet101 resnet50_128 lightcnn}
for rep in 1 2; do for m in $MODELS; do for f in 3 $L; do
python bench.py --model $m --fusion $f --no-cpu-baseline --no-secondary --no-sustained --no-unfused-ref --no-profile --steps 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$m fusion $f', round(d['value'],1), round(d['ms_per_step'],3))"
done; done; done
