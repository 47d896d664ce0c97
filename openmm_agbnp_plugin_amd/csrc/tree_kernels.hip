// __global__ entry points of the overlap-tree stage (see tree_kernels.h for the algorithm).
#include "tree_kernels.h"

namespace agbnp {

// Build + cavity passes of one heavy atom's subtree (reference steps A-D of
// platforms/reference/src/ReferenceAGBNPKernels.cpp:293-384, restated in oracle run_cavity()).
template <int NCAP, int ACAP, bool GLOBAL>
__global__ __launch_bounds__(64) void k_tree_cavity(TreeArgs A) {
  extern __shared__ __align__(16) char smem[];
  TreeStore<NCAP, ACAP> S;
  S.carve(GLOBAL ? (A.scratch + (size_t)blockIdx.x * A.scratch_stride) : smem);
  const int lane = threadIdx.x;

  for (int hi = blockIdx.x; hi < A.nh; hi += gridDim.x) {
    for (int la = lane; la < ACAP; la += 64) {
      S.at[6][la] = 0.0;
      S.at[7][la] = 0.0;
      S.at[8][la] = 0.0;
      S.at[9][la] = 0.0;
    }
    __syncthreads();
    int nnodes = 0, natoms = 0;
    const int rc = build_subtree<NCAP, ACAP>(S, A, lane, hi, &nnodes, &natoms);
    if (rc != kBuildOk) {
      if (lane == 0) {
        atomicAdd(&A.status[rc == kBuildNodeOverflow ? kStatNodeOverflow : kStatAtomOverflow], 1);
        A.hdr[hi].nnodes = 0;
        A.hdr[hi].natoms = 0;
      }
      __syncthreads();
      continue;
    }

    // ---- topology out (8 B/node) for the pseudo-volume pass
    int pool_off = 0, atom_off = 0;
    if (lane == 0) {
      pool_off = atomicAdd(&A.status[kStatPoolUsed], nnodes);
      atom_off = atomicAdd(&A.status[kStatAtomPoolUsed], natoms);
      atomicMax(&A.status[kStatMaxNodes], nnodes);
      atomicMax(&A.status[kStatMaxAtoms], natoms);
    }
    pool_off = __builtin_amdgcn_readfirstlane(pool_off);
    atom_off = __builtin_amdgcn_readfirstlane(atom_off);
    const bool pool_ok = (pool_off + nnodes <= A.pool_cap) && (atom_off + natoms <= A.atom_pool_cap);
    if (!pool_ok) {
      if (lane == 0) {
        atomicAdd(&A.status[kStatPoolOverflow], 1);
        A.hdr[hi].nnodes = 0;
        A.hdr[hi].natoms = 0;
      }
    } else {
      for (int n = lane; n < nnodes; n += 64)
        A.node_pool[pool_off + n] = make_ushort4(S.nla[n], S.npar[n], S.ncs[n], S.ncc[n]);
      for (int la = lane; la < natoms; la += 64) A.atom_pool[atom_off + la] = S.at_gidx[la];
      if (lane == 0) {
        SubtreeHeader h;
        h.nnodes = nnodes;
        h.natoms = natoms;
        h.pool_off = pool_off;
        h.atom_off = atom_off;
        h.lvl[0] = 0;
        for (int L = 1; L <= 9; L++) h.lvl[L] = S.lvl[L];
        A.hdr[hi] = h;
      }
    }

    // ---- pass 1: enlarged radii, nu = +gamma/roffset
    const double e1 = sweep_bottomup<NCAP, ACAP, true>(S, lane);
    // pass-1 self volumes are a diagnostic only; hand them out and reset the accumulator
    for (int la = lane; la < natoms; la += 64) {
      const double sv = S.at[9][la];
      if (A.sv_large != nullptr && sv != 0.0) glb_add(&A.sv_large[S.at_gidx[la]], sv);
      S.at[9][la] = 0.0;
      // switch the local atoms to vdW radii, nu = -gamma/roffset
      const int hj = S.at_gidx[la];
      S.at[3][la] = A.a_vdw[hj];
      S.at[4][la] = A.v_vdw[hj];
      S.at[5][la] = -S.at[5][la];
    }
    __syncthreads();

    // ---- rescan + pass 2
    rescan_topdown<NCAP, ACAP>(S, lane);
    const double e2 = sweep_bottomup<NCAP, ACAP, true>(S, lane);

    // ---- flush per-atom sums
    for (int la = lane; la < natoms; la += 64) {
      const int hj = S.at_gidx[la];
      glb_add(&A.gx[hj], S.at[6][la]);
      glb_add(&A.gy[hj], S.at[7][la]);
      glb_add(&A.gz[hj], S.at[8][la]);
      glb_add(&A.sv_vdw[hj], S.at[9][la]);
    }
    if (lane == 0) {
      A.epart[2 * hi] = e1;
      A.epart[2 * hi + 1] = e2;
      atomicAdd(&A.status[kStatTotalNodes], nnodes);
    }
    __syncthreads();
  }
}

// Pseudo-volume pass (reference steps K+L, ReferenceAGBNPKernels.cpp:718-747): the stored topology is
// reloaded, Gaussians are recomputed top-down with vdW radii and nu_i = (W_i+U_i)/V_i, then the
// gradient-only bottom-up sweep runs.  The reference does two sweeps (W then U); the sweep is linear
// in nu, so one sweep with the sum gives the same gradient.
template <int NCAP, int ACAP, bool GLOBAL>
__global__ __launch_bounds__(64) void k_tree_pseudo(TreeArgs A) {
  extern __shared__ __align__(16) char smem[];
  TreeStore<NCAP, ACAP> S;
  S.carve(GLOBAL ? (A.scratch + (size_t)blockIdx.x * A.scratch_stride) : smem);
  const int lane = threadIdx.x;
  for (int hi = blockIdx.x; hi < A.nh; hi += gridDim.x) {
    const SubtreeHeader* H = &A.hdr[hi];
    const int nnodes = H->nnodes, natoms = H->natoms;
    if (nnodes <= 1) continue;  // a lone atom has no position-dependent volume
    const int pool_off = H->pool_off, atom_off = H->atom_off;
    if (lane < 10) S.lvl[lane] = H->lvl[lane];
    for (int n = lane; n < nnodes; n += 64) {
      const ushort4 t = A.node_pool[pool_off + n];
      S.nla[n] = t.x;
      S.npar[n] = t.y;
      S.ncs[n] = t.z;
      S.ncc[n] = t.w;
    }
    for (int la = lane; la < natoms; la += 64) {
      const int hj = A.atom_pool[atom_off + la];
      S.at_gidx[la] = hj;
      S.at[0][la] = A.hx[hj];
      S.at[1][la] = A.hy[hj];
      S.at[2][la] = A.hz[hj];
      S.at[3][la] = A.a_vdw[hj];
      S.at[4][la] = A.v_vdw[hj];
      S.at[5][la] = A.gam[hj];
      S.at[6][la] = 0.0;
      S.at[7][la] = 0.0;
      S.at[8][la] = 0.0;
    }
    __syncthreads();
    rescan_topdown<NCAP, ACAP>(S, lane);
    sweep_bottomup<NCAP, ACAP, false>(S, lane);
    for (int la = lane; la < natoms; la += 64) {
      const int hj = S.at_gidx[la];
      glb_add(&A.gx[hj], S.at[6][la]);
      glb_add(&A.gy[hj], S.at[7][la]);
      glb_add(&A.gz[hj], S.at[8][la]);
    }
    __syncthreads();
  }
}

// ---- host-side launchers -------------------------------------------------------------------------
// variant: 0 = (512 nodes, 64 atoms) LDS, 1 = (1024,128) LDS, 2 = (2048,256) LDS, 3 = (32768,1024) global scratch
constexpr int kGlobalNodeCap = 32768;
constexpr int kGlobalAtomCap = 1024;

size_t tree_variant_lds_bytes(int variant) {
  switch (variant) {
    case 0: return TreeStore<512, 64>::kBytes;
    case 1: return TreeStore<1024, 128>::kBytes;
    case 2: return TreeStore<2048, 256>::kBytes;
    default: return 0;
  }
}
size_t tree_variant_scratch_bytes(int variant) {
  return variant == 3 ? ((TreeStore<kGlobalNodeCap, kGlobalAtomCap>::kBytes + 255) / 256) * 256 : 0;
}
int tree_variant_node_cap(int variant) {
  static const int caps[4] = {512, 1024, 2048, kGlobalNodeCap};
  return caps[variant];
}

template <class K>
static hipError_t launch_tree(K kernel, int grid, size_t lds, const TreeArgs& A, hipStream_t st) {
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(kernel, dim3(grid), dim3(64), lds, st, A);
  return hipGetLastError();
}

hipError_t launch_tree_cavity(int variant, int global_grid, const TreeArgs& A, hipStream_t st) {
  if (A.nh <= 0) return hipSuccess;
  switch (variant) {
    case 0: return launch_tree(k_tree_cavity<512, 64, false>, A.nh, TreeStore<512, 64>::kBytes, A, st);
    case 1: return launch_tree(k_tree_cavity<1024, 128, false>, A.nh, TreeStore<1024, 128>::kBytes, A, st);
    case 2: return launch_tree(k_tree_cavity<2048, 256, false>, A.nh, TreeStore<2048, 256>::kBytes, A, st);
    default: return launch_tree(k_tree_cavity<kGlobalNodeCap, kGlobalAtomCap, true>, global_grid < A.nh ? global_grid : A.nh, 0, A, st);
  }
}

hipError_t launch_tree_pseudo(int variant, int global_grid, const TreeArgs& A, hipStream_t st) {
  if (A.nh <= 0) return hipSuccess;
  switch (variant) {
    case 0: return launch_tree(k_tree_pseudo<512, 64, false>, A.nh, TreeStore<512, 64>::kBytes, A, st);
    case 1: return launch_tree(k_tree_pseudo<1024, 128, false>, A.nh, TreeStore<1024, 128>::kBytes, A, st);
    case 2: return launch_tree(k_tree_pseudo<2048, 256, false>, A.nh, TreeStore<2048, 256>::kBytes, A, st);
    default: return launch_tree(k_tree_pseudo<kGlobalNodeCap, kGlobalAtomCap, true>, global_grid < A.nh ? global_grid : A.nh, 0, A, st);
  }
}

}  // namespace agbnp
