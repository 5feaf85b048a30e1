#!/bin/bash
# Collect one round's rocprofv3 evidence for bench.py on the GPU box (run from the repo root, e.g. through gpurun):
#   bash profiles/collect.sh r6      -> gpurun_out/prof_r6/..., summaries copied to profiles/r6/
# Counters are collected in their own passes (--pmc never together with a trace domain other than the kernel trace).
# Every BASELINE.json single-GPU configuration gets the same set: resnet101 (no suffix), resnet50_128 (_r50), lightcnn (_lcnn).
set -u
R=${1:-r6}
D=gpurun_out/prof_$R
P=profiles/$R
export TMPDIR=/tmp
mkdir -p "$D" "$P"
ST="--kernel-trace --stats --output-format csv"
# optional 2nd argument: the model specs to (re)collect, e.g. "resnet50_128:_r50" after a change that only touches that backbone's schedule; the
# model-independent files further down are then skipped
SPECS=${2:-"resnet101: resnet50_128:_r50 lightcnn:_lcnn"}
for spec in $SPECS; do
  M=${spec%%:*}; T=${spec##*:}
  B="python bench.py --model $M"
  # one stream: the schedule whose kernel durations a profiler can attribute
  rocprofv3 $ST -d $D/serial$T -o $R -- $B --steps 5 --warmup 2 --no-cpu-baseline --serial --no-unfused-ref --no-sustained --no-profile > $P/bench_serial_under_rocprof$T.json 2> $D/serial$T.err
  cp $D/serial$T/${R}_kernel_stats.csv $P/kernel_stats_serial$T.csv
  rm -f $D/serial$T/${R}_kernel_trace.csv
  PB="$B --steps 2 --warmup 1 --no-cpu-baseline --no-profile --serial --no-sustained"
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA --output-format csv -d $D/pmc_mfma$T -o $R -- $PB > /dev/null 2> $D/pmc_mfma$T.err
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $D/pmc_fetch$T -o $R -- $PB > /dev/null 2> $D/pmc_fetch$T.err
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $D/pmc_write$T -o $R -- $PB > /dev/null 2> $D/pmc_write$T.err
  python profiles/pmc_summary.py $D/pmc_mfma$T > $P/pmc_mfma$T.txt
  python profiles/pmc_summary.py $D/pmc_fetch$T > $P/pmc_FETCH_SIZE$T.txt
  python profiles/pmc_summary.py $D/pmc_write$T > $P/pmc_WRITE_SIZE$T.txt
  rm -f $D/pmc_*$T/${R}_counter_collection.csv
  # per-layer GEMM table (HIP events inside the engine, serial schedule)
  rm -f $D/layers$T.csv $D/launchlog$T.csv; $B --steps 3 --warmup 2 --no-cpu-baseline --serial --no-sustained --no-profile --profile-csv $D/layers$T.csv --launch-log-csv $D/launchlog$T.csv > /dev/null 2>&1
  python profiles/layer_table.py $D/layers$T.csv > $P/gemm_layers_serial$T.txt
  # the same launches timed by the kernels themselves (first GEMM of a step included correctly)
  python profiles/layer_table.py $D/launchlog$T.csv > $P/gemm_layers_inkernel$T.txt
  # the plain bench line (timed three-stream schedule, launch-log roofline, clock, CPU baseline) + the timeline it came from
  $B --no-secondary --timeline-json $P/gemm_timeline$T.json --launch-log-out $P/launch_log$T.csv > $P/bench_default$T.json 2> $D/bench_default$T.err
done
if [ -n "${2:-}" ]; then
  python bench.py > $P/bench_driver_line.json 2> $D/bench_driver_line.err
  mkdir -p gpurun_out/$P && cp -r $P/. gpurun_out/$P/
  echo "collected ($SPECS): $(ls $P | tr '\n' ' ')"
  exit 0
fi
# the timed schedule under the profiler (rocprofv3 serialises the queues: kernel mix only)
rocprofv3 $ST -d $D/pipelined -o $R -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-profile --no-sustained > $P/bench_pipelined_under_rocprof.json 2> $D/pipelined.err
cp $D/pipelined/${R}_kernel_stats.csv $P/kernel_stats_pipelined.csv
rm -f $D/pipelined/${R}_kernel_trace.csv
# clocks and power: rocm-smi sampled once a second (a) during the sustained loop of the bench, (b) during back-to-back GEMM launches
# on three streams (the chip at its power limit); the in-kernel clock measurements next to them
smi() { while true; do date +%s.%N; rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|power" ; sleep 1; done; }
smi > $P/smi_bench_sustained.txt & SMI=$!
python bench.py --no-cpu-baseline --no-profile --sustained-seconds 15 > /dev/null 2>&1
kill $SMI; wait $SMI 2>/dev/null
smi > $P/smi_gemm_saturated.txt & SMI=$!
python tools/conv_sweep.py --cfgs 3000007,3000004 --reps 30000 --only 0 --clock > $P/gemm_saturated.txt 2>&1
kill $SMI; wait $SMI 2>/dev/null
python tools/conv_sweep.py --cfgs 7,4 --reps 300 --only 0,1 --stamps > $P/gemm_stamps.txt 2>&1
python tools/clock_probe.py --steps 20 > $P/clock_probe_timed.json 2> /dev/null
python tools/clock_probe.py --steps 10 --serial > $P/clock_probe_serial.json 2> /dev/null
# where a workgroup's life goes while the step runs (sampled stamps per launch), do different layers' launches help each other,
# the step against the batch size, the RISE-scale forward sweep, the isolated kernels on every layer shape
python tools/phase_probe.py 2> /dev/null | grep -v amdgpu > $P/phase_probe_timed.txt
python tools/phase_probe.py --serial 2> /dev/null | grep -v amdgpu > $P/phase_probe_serial.txt
python tools/pair_probe.py 2> /dev/null | grep -v amdgpu > $P/pair_probe.txt
for b in 32 64 96; do python bench.py --batch $b --steps 20 --warmup 4 --no-cpu-baseline --no-sustained --no-secondary 2> /dev/null; done > $P/bench_batch_sweep.jsonl
python tools/embeddings_sweep.py --masks 6500 > $P/embeddings_sweep.json 2> /dev/null
python tools/embeddings_sweep.py --masks 6500 --no-split > $P/embeddings_sweep_nosplit.json 2> /dev/null
# round 5: the raw launch log of the timed schedule -> frac, recomputed without repo code; the lean schedule against the literal one on this box;
# BASELINE.json configs[4] (job mix) and its dominant method under the profiler
python profiles/frac_from_launch_log.py $P/launch_log.csv --alg-gflop 2768.4 > $P/frac_from_launch_log.txt 2>&1
python profiles/frac_from_launch_log.py $P/launch_log_r50.csv --alg-gflop 2961.4 >> $P/frac_from_launch_log.txt 2>&1
for rep in 1 2; do for m in resnet101 resnet50_128; do for f in "" "--no-lean"; do
  python bench.py --model $m $f --no-cpu-baseline --no-secondary --no-sustained --no-profile --steps 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$m [$f]', round(d['value'],1), 'maps/s', round(d['ms_per_step'],3), 'ms')"
done; done; done > $P/lean_ab.txt
# bf16x6 modes (xfr_engine_set_split_gemm; 3 = the default since round 6: forward convolutions and backward-data GEMMs) on this box, alternating; GEMM
# error of the kernels against float64
for rep in 1 2; do for m in resnet101 resnet50_128; do for f in "--split-gemm 0" "--split-gemm 1" ""; do
  python bench.py --model $m $f --no-split-leg --no-cpu-baseline --no-secondary --no-sustained --no-profile --steps 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$m [$f]', round(d['value'],1), 'maps/s', round(d['ms_per_step'],3), 'ms', 'outputs_ok', d['outputs_ok'], '1-cos(row 0)', 1-d['row0_cosine_vs_reference'])"
done; done; done > $P/split_gemm_ab.txt
python tools/conv_error_probe.py --extra 2> /dev/null | grep -v amdgpu > $P/conv_error_probe.txt
python tools/conv_sweep.py --cfgs 7,4,9 --reps 200 --only 0,1,3,4,10 2> /dev/null | grep -v amdgpu > $P/conv_sweep_bf16x6.txt
python tools/conv_sweep.py --cfgs 7,4,9 --reps 200 --only 0,1,3,4 --nb 32 2> /dev/null | grep -v amdgpu >> $P/conv_sweep_bf16x6.txt
python tools/conv_sweep.py --cfgs 7,4,9 --reps 200 --only 0,1,3,4 --nb 8 2> /dev/null | grep -v amdgpu >> $P/conv_sweep_bf16x6.txt
python tools/mean_ebp_probe.py 2> /dev/null | grep -v amdgpu > $P/mean_ebp_probe.txt
python bench.py --inpainting-game > $P/bench_inpainting_game.json 2> /dev/null
python tools/subtree_probe.py --log 2> /dev/null | grep -v amdgpu > $P/weighted_subtree_probe.txt
rocprofv3 $ST -d $D/subtree -o $R -- python tools/subtree_probe.py --reps 5 > /dev/null 2> $D/subtree.err
cp $D/subtree/${R}_kernel_stats.csv $P/kernel_stats_weighted_subtree.csv
rm -f $D/subtree/${R}_kernel_trace.csv
# the line the driver sees: the default command, secondary configurations and whole-host CPU figure included
python bench.py > $P/bench_driver_line.json 2> $D/bench_driver_line.err
python tools/conv_sweep.py --cfgs 4,5,7 --reps 100 --set r101 2> /dev/null | grep -v amdgpu > $P/conv_sweep.txt
python tools/conv_sweep.py --cfgs 4,7 --reps 100 --set lcnn --nb 128 2> /dev/null | grep -v amdgpu >> $P/conv_sweep.txt
mkdir -p gpurun_out/$P && cp -r $P/. gpurun_out/$P/
echo "collected: $(ls $P | tr '\n' ' ')"
