#!/bin/bash
# Diagnostic: build kernel variants for A/B timing on the GPU box (tools/time_kernels.py).  usage: tools/ab_build.sh name "-DFLAG=.. ..." [name flags ...]
# The variants land in tools/ab/ (git-ignored, but they travel with gpurun).
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/ab
while [ $# -ge 2 ]; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -shared -mllvm -amdgpu-mfma-vgpr-form -Iinclude $2 multimodn_amd/csrc/mmn_kernels.hip -o tools/ab/lib_$1.so &
  shift 2
done
wait
ls -la tools/ab
