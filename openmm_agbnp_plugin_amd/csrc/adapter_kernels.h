// Launchers of the OpenMM data-convention adapters (see adapter_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace agbnp {

hipError_t launch_adapt_positions(int n, const void* posq, int posq_is_double, const void* correction, const int* atom_index, double* xyz,
                                  hipStream_t st);
hipError_t launch_adapt_outputs(int n, int padded, const int* atom_index, double* force_xyz, double* energy, unsigned long long* force_fixed,
                                void* energy_buffer, int energy_is_double, int energy_slot, hipStream_t st);

}  // namespace agbnp
