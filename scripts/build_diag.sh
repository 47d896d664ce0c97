#!/bin/bash
# builds the product library and the stamped diagnostic one (build/diag/libagbnp_hip_pstamps.so)
set -e
cd "$(dirname "$0")/.."
make -C openmm_agbnp_plugin_amd/csrc 2>&1 | grep -E "error|Error" && exit 1
mkdir -p build/diag
cd openmm_agbnp_plugin_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wall -Wno-unused-function -DAGBNP_TREE_BLOCK=192 -mllvm -amdgpu-kernarg-preload-count=16 -DAGBNP_PAIR_STAMPS -shared -o ../../build/diag/libagbnp_hip_pstamps.so tree_kernels.hip pair_kernels.hip adapter_kernels.hip engine.hip i4_tables.cpp 2>&1 | grep -E "error" && exit 1
echo built
