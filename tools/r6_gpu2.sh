#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r6b; O=gpurun_out/r6b
export XFR_QUIET=1 XFR_STREAM_K=0
for f in xfr_amd/csrc/variants/libxfr_amd_*.so; do
  for shape in 256,14,14,256,3,1,1 1024,14,14,256,1,1,0; do
    timeout 300 python tools/sp2_prof.py --lib $f --shape $shape 2>&1 | grep -v amdgpu.ids >> $O/prof.txt
  done
done
cat $O/prof.txt
