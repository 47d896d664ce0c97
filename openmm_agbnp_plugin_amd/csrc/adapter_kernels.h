// Launcher of the OpenMM data-convention adapter on the input side (see adapter_kernels.hip; the output side is part of
// k_outputs / the energy workgroup, PairArgs::omm).
#pragma once
#include <hip/hip_runtime.h>

namespace agbnp {

hipError_t launch_adapt_positions(int n, const void* posq, int posq_is_double, const void* correction, const int* atom_index, double* xyz,
                                  int* ctx_slot, hipStream_t st);

hipError_t launch_order_maps(int n, const int* atom_index, const int* a2h, int* ctx_slot, int* hslot, hipStream_t st);

// five-launch mode: the words beside the work-slot rows that say where the tree finds the position of every item's root --
// map = h2a (the caller's [3n] array: atom indices) or hslot (an OpenMM context's posq: slots of the context's order).  Launched
// when the entry point changes from one evaluation to the next, or the context has reordered its atoms.
hipError_t launch_row_atoms(int nslots, const int* rows, const int* map, int* row_atoms, hipStream_t st);

}  // namespace agbnp
