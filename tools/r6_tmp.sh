#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r6j; O=gpurun_out/r6j
export XFR_QUIET=1
timeout 600 python tools/conv_error_probe.py --extra --cfgs 7,9 2>&1 | grep -v amdgpu > $O/err_patch.txt
for nb in 64 32; do
timeout 600 python tools/conv_sweep.py --cfgs 7,9 --only 0,3,6,10 --nb $nb 2>&1 | grep -v amdgpu > $O/sweep_patch_nb$nb.txt
XFR_SPLIT_NO_PATCH=1 timeout 600 python tools/conv_sweep.py --cfgs 9 --only 0,3,6,10 --nb $nb 2>&1 | grep -v amdgpu > $O/sweep_nopatch_nb$nb.txt
done
B="python bench.py --steps 20 --warmup 5 --no-profile --no-sustained --no-split-leg --no-secondary --no-cpu-baseline"
for rep in 1 2 3; do
for model in resnet101 resnet50_128; do
  echo "$model patch: $(timeout 600 $B --model $model 2>/dev/null | python -c 'import sys,json; j=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print("%.1f %.3f" % (j["value"], j["ms_per_step"]), j.get("outputs_ok"), j.get("row0_cosine_vs_reference"))')" >> $O/patch_ab.txt
  echo "$model no patch: $(XFR_SPLIT_NO_PATCH=1 timeout 600 $B --model $model 2>/dev/null | python -c 'import sys,json; j=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print("%.1f %.3f" % (j["value"], j["ms_per_step"]), j.get("outputs_ok"), j.get("row0_cosine_vs_reference"))')" >> $O/patch_ab.txt
done
done
cat $O/err_patch.txt $O/sweep_patch_nb64.txt $O/sweep_nopatch_nb64.txt $O/sweep_patch_nb32.txt $O/sweep_nopatch_nb32.txt $O/patch_ab.txt
