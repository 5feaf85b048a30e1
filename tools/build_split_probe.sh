#!/bin/bash
# Build the bf16-split GEMM experiment (tools/split_gemm_probe.hip) for gfx950; the binary and the compiler's temporaries go to tools/bin/ (git-ignored).
set -e
cd "$(dirname "$0")"
mkdir -p bin
hipcc --offload-arch=gfx950 -O3 -std=c++17 split_gemm_probe.hip -o bin/split_gemm_probe -save-temps=obj
