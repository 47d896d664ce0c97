// Data-convention adapters between an OpenMM GPU ComputeContext and the engine (GPU-platform side of the boundary).
//
// What the reference's GPU platform works on (platforms/opencl/src/OpenCLAGBNPKernels.cpp:541-556 and
// platforms/opencl/src/kernels/GVolReduceTree.cl:92-121):
//   posq          real4 per atom {x, y, z, q} in the CONTEXT's atom order (OpenMM reorders atoms for locality;
//                 atomIndex[slot] = the particle's index in the System / Force), float or double; in mixed
//                 precision a second float4 array holds the low-order parts (posqCorrection)
//   force buffer  64-bit fixed point, value * 2^32, three planes [x: 0..P) [y: P..2P) [z: 2P..3P) over the
//                 padded atom count P, context order, accumulated with integer atomics (GVolReduceTree.cl:117-119)
//   energy buffer per-thread slots of the context's energy accumulator, float or double (GVolReduceTree.cl:112)
// The engine computes in FP64 on [3N] xyz arrays in the Force's particle order, so the glue is:
//   k_adapt_positions   posq[slot] (+ correction) -> xyz[3 * atomIndex[slot]], and the inverse map particle -> slot
//   output side         fused into the engine's own last kernel: k_outputs adds a particle's force as fixed point at its
//                       slot and the energy workgroup adds into the context's accumulator (PairArgs::omm,
//                       pair_kernels.hip); an evaluation that overflowed adds nothing to the context either
#include <hip/hip_runtime.h>

#include "adapter_kernels.h"
#include "agbnp_common.h"

namespace agbnp {

template <class Real4>
__global__ __launch_bounds__(256) void k_adapt_positions(int n, const Real4* __restrict__ posq, const float4* __restrict__ correction,
                                                         const int* __restrict__ atom_index, double* __restrict__ xyz,
                                                         int* __restrict__ ctx_slot) {
  const int slot = blockIdx.x * blockDim.x + threadIdx.x;
  if (slot >= n) return;
  const Real4 p = posq[slot];
  double x = (double)p.x, y = (double)p.y, z = (double)p.z;
  if (correction) {  // mixed precision: position = posq + posqCorrection, both float
    const float4 c = correction[slot];
    x += (double)c.x;
    y += (double)c.y;
    z += (double)c.z;
  }
  const int i = atom_index ? atom_index[slot] : slot;
  xyz[3 * i] = x;
  xyz[3 * i + 1] = y;
  xyz[3 * i + 2] = z;
  ctx_slot[i] = slot;  // the way back, for the output side (k_outputs adds the forces at the context's slot itself)
}

// The maps that let k_prep read an OpenMM context's posq itself (OpenmmSource, pair_kernels.h): particle -> slot and heavy
// index -> slot.  Built when the context's atomIndex array is first seen and again after the engine has found it changed
// (k_prep checks every particle's entry against atomIndex in every evaluation: kStatOrderStale).
__global__ __launch_bounds__(256) void k_order_maps(int n, const int* __restrict__ atom_index, const int* __restrict__ a2h,
                                                   int* __restrict__ ctx_slot, int* __restrict__ hslot) {
  const int slot = blockIdx.x * blockDim.x + threadIdx.x;
  if (slot >= n) return;
  const int i = atom_index ? atom_index[slot] : slot;
  if (i < 0 || i >= n) return;  // (not a permutation: the check in k_prep will say so)
  ctx_slot[i] = slot;
  const int h = a2h[i];
  if (h >= 0) hslot[h] = slot;
}

hipError_t launch_order_maps(int n, const int* atom_index, const int* a2h, int* ctx_slot, int* hslot, hipStream_t st) {
  hipLaunchKernelGGL(k_order_maps, dim3((n + 255) / 256), dim3(256), 0, st, n, atom_index, a2h, ctx_slot, hslot);
  return hipGetLastError();
}

__global__ __launch_bounds__(256) void k_row_atoms(int nwords, const int* __restrict__ rows, const int* __restrict__ map, int* __restrict__ row_atoms) {
  const int w = blockIdx.x * blockDim.x + threadIdx.x;  // item k of work slot s: w = kMaxItems * s + k
  if (w >= nwords) return;
  const int item = rows[(size_t)kRowStride * (w / kMaxItems) + (w % kMaxItems)];
  row_atoms[w] = item >= 0 ? map[item & 0xffffff] : 0;
}

hipError_t launch_row_atoms(int nslots, const int* rows, const int* map, int* row_atoms, hipStream_t st) {
  const int nwords = nslots * kMaxItems;
  hipLaunchKernelGGL(k_row_atoms, dim3((nwords + 255) / 256), dim3(256), 0, st, nwords, rows, map, row_atoms);
  return hipGetLastError();
}

hipError_t launch_adapt_positions(int n, const void* posq, int posq_is_double, const void* correction, const int* atom_index, double* xyz,
                                  int* ctx_slot, hipStream_t st) {
  const dim3 grid((n + 255) / 256), block(256);
  if (posq_is_double)
    hipLaunchKernelGGL(k_adapt_positions<double4>, grid, block, 0, st, n, static_cast<const double4*>(posq),
                       static_cast<const float4*>(nullptr), atom_index, xyz, ctx_slot);
  else
    hipLaunchKernelGGL(k_adapt_positions<float4>, grid, block, 0, st, n, static_cast<const float4*>(posq),
                       static_cast<const float4*>(correction), atom_index, xyz, ctx_slot);
  return hipGetLastError();
}

}  // namespace agbnp
