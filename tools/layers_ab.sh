#!/bin/bash
# in-kernel per-layer GEMM tables (serial schedule) of two epilogue-fusion levels on one box: bash tools/layers_ab.sh MODEL LEVEL_A LEVEL_B
M=${1:-resnet101}; A=${2:-3}; B=${3:-35}
D=gpurun_out/layers_ab; mkdir -p $D
for f in $A $B $A $B; do
rm -f $D/ll.csv
python bench.py --model $M --fusion $f --steps 3 --warmup 2 --no-cpu-baseline --serial --no-sustained --no-profile --no-secondary --launch-log-csv $D/ll.csv > /dev/null 2>&1
python profiles/layer_table.py $D/ll.csv > $D/layers_${M}_$f.txt
head -12 $D/layers_${M}_$f.txt
done
