#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r6c; O=gpurun_out/r6c
export XFR_QUIET=1
# 1. load-phase variants (whole tiles)
for f in xfr_amd/csrc/variants/libxfr_amd_*.so; do
  XFR_STREAM_K=0 timeout 300 python tools/sp2_prof.py --lib $f 2>&1 | grep -v amdgpu.ids >> $O/prof.txt
done
# 2. precision with the sign phases
timeout 600 python tools/conv_error_probe.py --cfgs 9 > $O/err_v2_signs.txt 2>&1
# 3. the timed step: kernel v1 / v2, stream-K on / off, modes 1 / 3
B="python bench.py --steps 20 --warmup 5 --no-profile --no-sustained --no-split-leg --no-secondary --no-cpu-baseline"
for rep in 1 2; do
for cfg in "1 1 1" "2 1 1" "2 0 1" "1 1 3" "2 1 3" "2 0 3" "2 1 0"; do
  set -- $cfg
  echo "kernel v$1 stream_k $2 mode $3: $(XFR_SPLIT_KERNEL=$1 XFR_STREAM_K=$2 timeout 600 $B --split-gemm $3 2>/dev/null | python -c 'import sys,json; j=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print(j["value"], j["ms_per_step"], j.get("outputs_ok"), j.get("row0_check",{}))')" >> $O/bench_ab.txt
done
done
cat $O/prof.txt $O/bench_ab.txt; grep -v amdgpu $O/err_v2_signs.txt
