// The per-atom half of the preparation of an evaluation, shared by k_prep (pair_kernels.hip) and -- in the five-launch mode,
// the default for version 1 -- by the trailing workgroups of the cavity launch (tree_kernels.hip): the caller's positions into the
// heavy-atom table and the pair stages' records, the pair stages' accumulators cleared, the evaluation counted in, the
// neighbour rows' staleness test.
#pragma once
#include <hip/hip_runtime.h>

#include "agbnp_common.h"
#include "pair_kernels.h"

namespace agbnp {

// rows of the heavy-atom table that prep_atoms clears for the next evaluation (tree_kernels.h, HeavyRow: kHvGx .. kHvSvVdw are
// consecutive; checked by a static_assert in tree_kernels.hip)
constexpr int kPrepHvSvLarge = 9, kPrepHvGx = 10;

// ---- where a position comes from: the caller's [3n] array, or an OpenMM context's posq (OpenmmSource) ------------------
struct Pos3 {
  double x, y, z;
};
__device__ __forceinline__ Pos3 slot_position(const PairArgs& P, int slot) {  // as k_adapt_positions reads it (adapter_kernels.hip)
  if (P.in.is_double) {
    const double4 p = static_cast<const double4*>(P.in.posq)[slot];
    return Pos3{p.x, p.y, p.z};
  }
  const float4 p = static_cast<const float4*>(P.in.posq)[slot];
  Pos3 r{(double)p.x, (double)p.y, (double)p.z};
  if (P.in.correction) {  // mixed precision: position = posq + posqCorrection, both float
    const float4 c = P.in.correction[slot];
    r.x += (double)c.x, r.y += (double)c.y, r.z += (double)c.z;
  }
  return r;
}
__device__ __forceinline__ Pos3 atom_position(const PairArgs& P, int a) {
  if (P.in.posq) return slot_position(P, P.omm.ctx_slot[a]);
  return Pos3{P.pos[3 * a], P.pos[3 * a + 1], P.pos[3 * a + 2]};
}
__device__ __forceinline__ Pos3 heavy_position(const PairArgs& P, int h) {
  if (P.in.posq) return slot_position(P, P.in.hslot[h]);
  return atom_position(P, P.h2a[h]);
}


// i: the atom (or slot, or status word) of this thread; first_block: the thread belongs to the first workgroup of the role.
// five == false (k_prep, in front of the tree launch): the status words and tree accumulators of THIS evaluation are cleared.
// five == true (trailing workgroups of the cavity launch, whose forest workgroups are adding to this evaluation's tree
// accumulators and overflow words at the same time): those of the NEXT evaluation are -- the other parity's table, shapes and
// status block (PairArgs::next_*), which nobody touches while this evaluation runs -- and the neighbour masks this evaluation's
// trees were built from are checked against where the heavy atoms are now (PairArgs::mask_ref): one of them further than half
// the masks' skin away voids the evaluation (kStatOrderStale, bit 1); the device lays the masks down anew in that evaluation's
// Born-rows launch, the host only repeats it.
__device__ __forceinline__ void prep_atoms(const PairArgs& P, int i, bool first_block, bool five) {
  // the status words of ONE evaluation start from zero; the sticky ones (overflow log since the last
  // agbnp_hip_finish) are left alone, and the evaluation takes its running number
  if (i < kStatEvalWords && first_block) (five ? P.next_estatus : P.estatus)[i] = 0;
  if (i == 0) P.status[kStatEvalSeq] += 1;
  if (i < P.nslots && !P.rows_on) {  // (the row form needs none of the tiles' records)
    // bounding box of every block of 64 slots in pair order (one wave each) for the tile culling of the
    // chain-rule stage; padding slots are neutral
    const int a = P.pslot[i];
    double sx = 0.0, sy = 0.0, sz = 0.0;
    int2 sm = make_int2(0, 0);
    if (a >= 0) {
      const Pos3 r = atom_position(P, a);
      sx = r.x, sy = r.y, sz = r.z, sm = P.ameta[a];
    }
    // the slot's record for the range-limited stages (a padding slot keeps a harmless position and says so)
    P.prec[i] = make_double4(sx, sy, sz, __hiloint2double(a >= 0 ? 0 : -1, sm.x | ((sm.y & 0x7fff) << 16)));
    P.ys[i] = 0.0;  // GB Y sums arrive through atomics
    double lo[3] = {sx, sy, sz}, hi[3] = {sx, sy, sz};
    for (int d = 0; d < 3; d++) {
      lo[d] = a >= 0 ? lo[d] : 1e30;
      hi[d] = a >= 0 ? hi[d] : -1e30;
    }
    for (int off = 32; off > 0; off >>= 1)
      for (int d = 0; d < 3; d++) {
        lo[d] = fmin(lo[d], __shfl_xor(lo[d], off, 64));
        hi[d] = fmax(hi[d], __shfl_xor(hi[d], off, 64));
      }
    if ((i & 63) == 0) {
      for (int d = 0; d < 3; d++) {
        P.pbox[6 * (i >> 6) + d] = lo[d];
        P.pbox[6 * (i >> 6) + 3 + d] = hi[d];
      }
    }
  }
  if (((P.fast && !P.gb_rows) || P.gb_far) && i < ((P.n + 63) & ~63)) {  // atom-order block boxes for the tile culling of the cut GB stage / the far-strip test
    double lo[3], hi[3];
    {
      const Pos3 r = atom_position(P, i < P.n ? i : 0);
      const double rr[3] = {r.x, r.y, r.z};
      for (int d = 0; d < 3; d++) {
        lo[d] = i < P.n ? rr[d] : 1e30;
        hi[d] = i < P.n ? lo[d] : -1e30;
      }
    }
    for (int off = 32; off > 0; off >>= 1)
      for (int d = 0; d < 3; d++) {
        lo[d] = fmin(lo[d], __shfl_xor(lo[d], off, 64));
        hi[d] = fmax(hi[d], __shfl_xor(hi[d], off, 64));
      }
    if ((i & 63) == 0)
      for (int d = 0; d < 3; d++) {
        P.abox[6 * (i >> 6) + d] = lo[d];
        P.abox[6 * (i >> 6) + 3 + d] = hi[d];
      }
  }
  if (P.rows_on) {
    // Row form of the range-limited stages: its neighbour rows were built with a skin; they stay exact as long as no atom
    // is further than half the skin from where it was then.  (NaN reference positions -- a fresh context -- fail the test.)
    bool moved = false;
    if (i < P.n) {
      const Pos3 r = atom_position(P, i);
      const double rx = r.x - P.nl_ref[3 * i], ry = r.y - P.nl_ref[3 * i + 1], rz = r.z - P.nl_ref[3 * i + 2];
      moved = !(fma(rz, rz, fma(ry, ry, rx * rx)) <= P.nl_move2);
    }
    if (__ballot(moved) != 0ull && (threadIdx.x & 63) == 0) atomicOr(&P.nl_flag[0], 1);
    if (i < 3) P.nl_nitems[2 * i + ((P.nl_flag[1] + 1) & 1)] = 0;  // the work-item buffers that the next rebuild fills
  }
  if (i >= P.n) return;
  if (P.zero_out) {  // (the evaluation's own outputs are added much later: the tree launch lies in between)
    P.zero_out[3 * i] = 0.0;
    P.zero_out[3 * i + 1] = 0.0;
    P.zero_out[3 * i + 2] = 0.0;
    if (i == 0) P.zero_out[3 * (size_t)P.n] = 0.0;
  }
  if (P.in.posq && P.in.atom_index && P.in.atom_index[P.omm.ctx_slot[i]] != i) atomicOr(&P.estatus[kStatOrderStale], 1);  // (the context has reordered its atoms)
  const Pos3 r_i = atom_position(P, i);
  const double x = r_i.x, y = r_i.y, z = r_i.z;
  const double inv_vol_i = P.rows_on ? P.inv_vol_a[i] : 0.0;  // (asked for with everything else: no load waits for the heavy index)
  const int screener_i = P.rows_on ? P.ameta[i].y : 0;
  P.aposq[i] = make_double4(x, y, z, P.charge[i]);
  if (P.rows_on) {
    P.bw[i] = 0.0;  // brw + bru arrives through the GB stage's atomics
    P.grec[i] = make_double4(0.0, 0.0, 0.0, 0.0);  // ... G through the Born rows'
  }
  P.gb_fx[i] = 0.0;  // GB sums arrive through atomics
  P.gb_fy[i] = 0.0;
  P.gb_fz[i] = 0.0;
  P.born_part[i] = 0.0;  // Born sums and chain-rule sums arrive through atomics too
  if (!P.rows_on) {      // (the row form assembles the chain-rule force from its G and H sums)
    P.db_fx[i] = 0.0;
    P.db_fy[i] = 0.0;
    P.db_fz[i] = 0.0;
  }
  P.db_wu[i] = 0.0;
  const int h = P.a2h[i];
  if (h >= 0) {
    P.hx[h] = x;
    P.hy[h] = y;
    P.hz[h] = z;
    if (!five) {
      P.gx[h] = 0.0;
      P.gy[h] = 0.0;
      P.gz[h] = 0.0;
      P.sv_vdw[h] = 0.0;
      P.sv_large[h] = 0.0;
      P.sizes[h] = make_int2(0, 0);  // subtree shapes are summed up by the tree workgroups (several may share a subtree)
    } else {
      // (this evaluation's tree workgroups are adding to its own rows right now: the NEXT evaluation's are cleared)
      double* nx = P.next_hv + h;
      nx[(size_t)kPrepHvGx * P.hstride] = 0.0;
      nx[(size_t)(kPrepHvGx + 1) * P.hstride] = 0.0;
      nx[(size_t)(kPrepHvGx + 2) * P.hstride] = 0.0;
      nx[(size_t)(kPrepHvGx + 3) * P.hstride] = 0.0;  // self volumes (vdW radii)
      nx[(size_t)kPrepHvSvLarge * P.hstride] = 0.0;
      P.next_sizes[h] = make_int2(0, 0);
      // the neighbour masks of this evaluation's trees are good while no heavy atom is further than half their skin from
      // where it was when they were laid down (NaN reference positions fail the test)
      const double mx = x - P.mask_ref[3 * h], my = y - P.mask_ref[3 * h + 1], mz = z - P.mask_ref[3 * h + 2];
      const double moved2 = fma(mz, mz, fma(my, my, mx * mx));
      if (!(moved2 <= P.mask_move2))
        atomicOr(&P.estatus[kStatOrderStale], 2);  // beyond half the skin: this evaluation's trees may have missed a neighbour
      else if (moved2 > 0.25 * P.mask_move2)
        P.estatus[kStatMaskAging] = 1;  // beyond a quarter: still exact, the masks are renewed before they stop being so
    }
    if (P.rows_on) {
      P.rec_h[h] = make_double4(x, y, z, inv_vol_i);
      P.hrec[h] = make_double4(0.0, 0.0, 0.0, 0.0);  // H arrives through the chain-rule rows' atomics
      P.hrow[h] = make_double4(x, y, z, __hiloint2double(0, i | (screener_i << 24)));
    }
  }
}

}  // namespace agbnp
