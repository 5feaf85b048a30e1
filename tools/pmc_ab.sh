#!/bin/bash
# per-launch-shape FETCH_SIZE / WRITE_SIZE of one model: bash tools/pmc_ab.sh MODEL [FUSION]
export TMPDIR=/tmp
M=${1:-resnet101}; F=${2:-3}
D=gpurun_out/pmc_grid; mkdir -p $D
for c in FETCH_SIZE WRITE_SIZE; do
PB="python bench.py --model $M --fusion $F --steps 2 --warmup 1 --no-cpu-baseline --no-profile --serial --no-sustained --no-secondary"
rocprofv3 --pmc $c --output-format csv -d $D/$c -o x -- $PB > /dev/null 2> $D/$c.err
python profiles/pmc_by_grid.py $D/$c 45 > $D/${c}_by_grid_$M.txt
rm -rf $D/$c
done
