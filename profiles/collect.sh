#!/bin/bash
# Collect one round's rocprofv3 evidence for bench.py on the GPU box (run from the repo root, e.g. through gpurun):
#   bash profiles/collect.sh r2      -> gpurun_out/prof_r2/..., summaries copied to profiles/r2/
# Counters are collected in their own passes (--pmc never together with a trace domain other than the kernel trace).
set -u
R=${1:-r2}
D=gpurun_out/prof_$R
P=profiles/$R
export TMPDIR=/tmp
mkdir -p "$D" "$P"
ST="--kernel-trace --stats --output-format csv"
rocprofv3 $ST -d $D/serial -o $R -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --serial --no-unfused-ref --no-sustained > $P/bench_serial_under_rocprof.json 2> $D/serial.err
rocprofv3 $ST -d $D/pipelined -o $R -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-profile --no-sustained > $P/bench_pipelined_under_rocprof.json 2> $D/pipelined.err
cp $D/serial/${R}_kernel_stats.csv $P/kernel_stats_serial.csv
cp $D/pipelined/${R}_kernel_stats.csv $P/kernel_stats_pipelined.csv
rm -f $D/*/${R}_kernel_trace.csv
PB="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile --serial --no-sustained"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA --output-format csv -d $D/pmc_mfma -o $R -- $PB > /dev/null 2> $D/pmc_mfma.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $D/pmc_fetch -o $R -- $PB > /dev/null 2> $D/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $D/pmc_write -o $R -- $PB > /dev/null 2> $D/pmc_write.err
python profiles/pmc_summary.py $D/pmc_mfma > $P/pmc_mfma.txt
python profiles/pmc_summary.py $D/pmc_fetch > $P/pmc_FETCH_SIZE.txt
python profiles/pmc_summary.py $D/pmc_write > $P/pmc_WRITE_SIZE.txt
rm -f $D/pmc_*/${R}_counter_collection.csv
# per-layer GEMM table (HIP events inside the engine, serial schedule) and the plain bench line
rm -f $D/layers.csv; python bench.py --steps 3 --warmup 2 --no-cpu-baseline --serial --no-sustained --no-profile --profile-csv $D/layers.csv > /dev/null 2>&1
python profiles/layer_table.py $D/layers.csv > $P/gemm_layers_serial.txt
python bench.py > $P/bench_default.json 2> $D/bench_default.err
python bench.py --model resnet50_128 --no-cpu-baseline > $P/bench_resnet50_128.json 2> /dev/null
python bench.py --model lightcnn --no-cpu-baseline > $P/bench_lightcnn.json 2> /dev/null
mkdir -p gpurun_out/$P && cp -r $P/. gpurun_out/$P/
echo "collected: $(ls $P | tr '\n' ' ')"
