#!/bin/bash
# round-6 first GPU pass: the v2 bf16x6 kernel against v1 -- precision, isolated rates, then the split tests
cd /root/repo; mkdir -p gpurun_out/r6a; O=gpurun_out/r6a
export XFR_QUIET=1
for v in 2 1; do
  XFR_SPLIT_KERNEL=$v timeout 600 python tools/conv_error_probe.py --extra > $O/err_v$v.txt 2>&1
done
XFR_SPLIT_KERNEL=2 XFR_STREAM_K=0 timeout 600 python tools/conv_error_probe.py --extra --cfgs 9 > $O/err_v2_whole.txt 2>&1
for nb in 64 32 8; do
  XFR_SPLIT_KERNEL=2 timeout 900 python tools/conv_sweep.py --cfgs 7,4,9 --only 0,1,2,3,4,10,11 --nb $nb > $O/sweep_v2_nb$nb.txt 2>&1
  XFR_SPLIT_KERNEL=2 XFR_STREAM_K=0 timeout 900 python tools/conv_sweep.py --cfgs 9 --only 0,1,2,3,4,10,11 --nb $nb > $O/sweep_v2_whole_nb$nb.txt 2>&1
  XFR_SPLIT_KERNEL=1 timeout 900 python tools/conv_sweep.py --cfgs 9 --only 0,1,2,3,4,10,11 --nb $nb > $O/sweep_v1_nb$nb.txt 2>&1
done
XFR_SPLIT_KERNEL=2 timeout 600 python tools/conv_sweep.py --cfgs 9 --only 0,1,3 --nb 64 --stamps > $O/stamps_v2.txt 2>&1
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "split or tail" > $O/pytest_split.txt 2>&1
tail -3 $O/pytest_split.txt
