// Launcher of the OpenMM data-convention adapter on the input side (see adapter_kernels.hip; the output side is part of
// k_outputs / the energy workgroup, PairArgs::omm).
#pragma once
#include <hip/hip_runtime.h>

namespace agbnp {

hipError_t launch_adapt_positions(int n, const void* posq, int posq_is_double, const void* correction, const int* atom_index, double* xyz,
                                  int* ctx_slot, hipStream_t st);

hipError_t launch_order_maps(int n, const int* atom_index, const int* a2h, int* ctx_slot, int* hslot, hipStream_t st);

}  // namespace agbnp
