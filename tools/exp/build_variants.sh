#!/bin/bash
# builds libmas_hip variants with different -D settings into tools/exp/variants/ (A/B experiments; *.so is git-ignored but travels with gpurun)
# usage: build_variants.sh name1 "-DFLAG..." name2 "-DFLAG..." ...
cd "$(dirname "$0")/../.."
mkdir -p tools/exp/variants
while [ $# -gt 1 ]; do
  name=$1; flags=$2; shift 2
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -O3 -ffp-contract=off -fPIC -shared $flags multi_agent_solver_amd/csrc/mas_capi.hip multi_agent_solver_amd/csrc/mas_models_ilp.hip -o tools/exp/variants/libmas_hip_$name.so &
done
wait
ls -la tools/exp/variants/
