#!/bin/bash
# GPU box, round 5, fifth call: what the lattice's GB launch costs if every strip took the Coulomb-only walk (timing build,
# results wrong) next to the real thing; subtree shapes of 2clr / 1dwc / the lattice for the offline packing experiments.
mkdir -p gpurun_out
BENCH_ARGS="--system 1dwc_x4 --steps 60 --warmup 6 --cpu-evals 0 --secondary 0" bash scripts/abx.sh 1 "lattice_far0|-|AGBNP_HIP_GB_FAR=0" "lattice_far1|-|" "lattice_allfar|build/diag/lib_allfar.so|" 2>&1 | tee gpurun_out/r5e_abx_lattice.log
timeout -k 10 300 python scripts/shape_dump.py 2clr 1dwc 1dwc_x4 trpcage 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5e_shapes.log
