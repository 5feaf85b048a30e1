# A-B of three forward slots against two (xfr_engine_set_pipeline bit 2) on one box: bash tools/ab_slots.sh
for rep in 1 2; do for m in resnet101 resnet50_128 lightcnn; do for f in "" "XFR_PIPE_SLOTS=3"; do
env $f python bench.py --model $m --no-cpu-baseline --no-secondary --no-sustained --no-unfused-ref --no-profile --steps 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$m [$f]', round(d['value'],1), round(d['ms_per_step'],3), d.get('outputs_ok'))"
done; done; done
