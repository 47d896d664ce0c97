// Pair stages of AGBNP1 on gfx950: inverse Born radii, GB pair energy/forces, Born-radius chain-rule
// forces, plus the per-atom glue between them.
//
// Per-atom glue (Born radius from the descreening sum, bru from Y, nu from W+U) has no kernels of its own: the
// consumer recomputes it from the finished sums in its prologue, which costs a few dozen flops per atom and
// saves three launches per evaluation.
//
// Reference semantics (platforms/reference/src/ReferenceAGBNPKernels.cpp, restated in oracle run_v1()):
//   :420-433  volume scaling factors s_i = selfvol_i / (4 pi R_i^3 / 3)
//   :435-454  beta_i = 1/R_i - (1/4pi) sum_{j heavy, j!=i, d<2nm} s_j Q(d; type_i, type_j);  B_i, f'_i (:41-55)
//   :464-504  GB pair energy, direct pair force, Y_i                         (ALL pairs, no cutoff)
//   :513-542  van der Waals energy, brw_i, bru_i
//   :555-586  chain-rule forces through the Born radii and the U_j / W_j sums
//
// Machine mapping: every pair stage walks 64x64 atom tiles, one workgroup of four waves per tile.  Lane l of a
// wave keeps atom i of block I and its sums in registers and meets the atoms of block J in cyclic order: their
// static records come from a doubled copy of the block in LDS, their running sums travel round the wave by DPP
// rotation, so every pair is evaluated once for both ends.  The sums of the four waves meet in LDS and leave
// as one set of FP64 HBM atomics per tile (float atomics execute at the memory side at a fixed chip-wide byte
// rate, so their bytes are kept small); the I4 spline tables sit in LDS.  The range-limited stages (Born sums,
// chain rule) walk the atoms heavy-first, skip H x H tiles and cull tiles by bounding boxes.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "agbnp_common.h"
#include "device_math.h"
#include "pair_kernels.h"
#include "prep_role.h"

namespace agbnp {

// Diagnostic build only (-DAGBNP_PAIR_STAMPS): wall-clock stamps (100 MHz) of every workgroup of the three pair
// kernels: [0] entry, [1] records in LDS, [2] walk done, [3] sums handed to the atomics, [4] HW_ID, [5] XCC_ID,
// [6] item, [7..11] finer stamps of the prologue.  Read back with agbnp_debug_pair_log (scripts/pair_timeline.py).  Never compiled into the product library.
#ifdef AGBNP_PAIR_STAMPS
constexpr int kPairLogSlots = 4096;
__device__ unsigned long long g_pair_log[3][kPairLogSlots][12];
#define PAIR_STAMP(kern, idx)                                                                                        \
  do {                                                                                                               \
    if (threadIdx.x == 0 && blockIdx.x < kPairLogSlots) g_pair_log[kern][blockIdx.x][idx] = wall_clock64();          \
  } while (0)
#define PAIR_STAMP_WAIT(kern, idx, what)                                                                            \
  do {                                                                                                               \
    asm volatile("s_waitcnt " what ::: "memory");                                                                    \
    PAIR_STAMP(kern, idx);                                                                                           \
  } while (0)
#define PAIR_STAMP_WHERE(kern, item)                                                                                 \
  do {                                                                                                               \
    if (threadIdx.x == 0 && blockIdx.x < kPairLogSlots) {                                                            \
      unsigned xcc__ = 0, hw__ = 0;                                                                                  \
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc__));                                          \
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw__));                                            \
      g_pair_log[kern][blockIdx.x][4] = hw__;                                                                        \
      g_pair_log[kern][blockIdx.x][5] = xcc__;                                                                       \
      g_pair_log[kern][blockIdx.x][6] = (unsigned)(item);                                                            \
    }                                                                                                                \
  } while (0)
#define PAIR_STAMP_HW(kern)                                                                                          \
  do {                                                                                                               \
    if (threadIdx.x == 0 && blockIdx.x < kPairLogSlots) {                                                            \
      unsigned xcc__ = 0, hw__ = 0;                                                                                  \
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc__));                                          \
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw__));                                            \
      g_pair_log[kern][blockIdx.x][4] = hw__;                                                                        \
      g_pair_log[kern][blockIdx.x][5] = xcc__;                                                                       \
    }                                                                                                                \
  } while (0)
#else
#define PAIR_STAMP(kern, idx)
#define PAIR_STAMP_WAIT(kern, idx, what)
#define PAIR_STAMP_WHERE(kern, item)
#endif
#ifdef AGBNP_PAIR_STAMPS
#define ROWS_LOG_COUNTS(kern, todo, nsteps)                                                                         \
  do {                                                                                                               \
    if (threadIdx.x == 0 && blockIdx.x < kPairLogSlots) g_pair_log[kern][blockIdx.x][6] = (unsigned)(todo), g_pair_log[kern][blockIdx.x][9] = (unsigned)(nsteps); \
  } while (0)
#endif
}  // namespace agbnp
#include "row_kernels.h"
namespace agbnp {

// ---- I4 spline (uniform nodes x_k = k*dr, k = 0..15; table entry = {y_k, z_k = y2_k*dr^2/6}) -------------------
// Natural cubic spline of the reference (AGBNPUtils.h:104-115 -> SplineFitter): on interval k, with t in [0,1),
//   S = (1-t) y_k + t y_k+1 + ((1-t)^3 - (1-t)) z_k + (t^3 - t) z_k+1
// expanded in powers of t (same polynomial, fewer operations than the a/b form):
//   S = y_k + t (dy - 2 z_k - z_k+1) + 3 z_k t^2 + (z_k+1 - z_k) t^3,     dy = y_k+1 - y_k
struct SplineCubic {
  double c0, c1, c2, c3, t;
};
__device__ __forceinline__ SplineCubic spline_cubic(const double2* __restrict__ tab, int base, double d) {
  // callers only ask inside the table (d < kI4MaxA), so the knot index needs no clamp and the position inside the
  // interval is the hardware fraction: one conversion instead of two (conversions issue at quarter rate)
  const double u = d * ((kI4Nodes - 1) / kI4MaxA);
  const int k = (int)u;
  const double2 lo = tab[base + k], hi = tab[base + k + 1];
  SplineCubic c;
  c.t = __builtin_amdgcn_fract(u);
  c.c0 = lo.x;
  c.c1 = (hi.x - lo.x) - fma(2.0, lo.y, hi.y);
  c.c2 = 3.0 * lo.y;
  c.c3 = hi.y - lo.y;
  return c;
}
__device__ __forceinline__ double spline_value(const double2* __restrict__ tab, int base, double d) {
  const SplineCubic c = spline_cubic(tab, base, d);
  return fma(fma(fma(c.c3, c.t, c.c2), c.t, c.c1), c.t, c.c0);
}
__device__ __forceinline__ void spline_value_deriv(const double2* __restrict__ tab, int base, double d, double& val, double& der) {
  const SplineCubic c = spline_cubic(tab, base, d);
  val = fma(fma(fma(c.c3, c.t, c.c2), c.t, c.c1), c.t, c.c0);
  der = fma(fma(3.0 * c.c3, c.t, 2.0 * c.c2), c.t, c.c1) * ((kI4Nodes - 1) / kI4MaxA);
}

// The I4 tables go from memory to LDS in batches of four independent loads per thread: every load of a batch is in
// flight before the first is waited for (a plain copy loop waits for each load before it issues the next: one memory
// round trip per 256 entries at the start of every tile).
struct LutBatch {
  double2 v[4];
};
__device__ __forceinline__ LutBatch lut_fetch(const double2* __restrict__ lut, int lut_entries, int base) {
  LutBatch b;
#pragma unroll
  for (int r = 0; r < 4; r++) b.v[r] = lut[min(base + (int)threadIdx.x + 256 * r, lut_entries - 1)];
  return b;
}
__device__ __forceinline__ void lut_store(double2* __restrict__ s_lut, const LutBatch& b, int lut_entries, int base) {
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int t = base + (int)threadIdx.x + 256 * r;
    if (t < lut_entries) s_lut[t] = b.v[r];
  }
}
__device__ __forceinline__ void lut_copy_rest(double2* __restrict__ s_lut, const double2* __restrict__ lut, int lut_entries) {
  for (int base = 1024; base < lut_entries; base += 1024) lut_store(s_lut, lut_fetch(lut, lut_entries, base), lut_entries, base);
}

// ---- geometry in, accumulators cleared -------------------------------------------------------------------
// Level-2 neighbour search, one workgroup per 64x64 tile of heavy atoms (I <= J), riding in the k_prep launch (it
// reads the caller's positions directly, so it does not depend on the rest of k_prep): lane i of every wave tests a
// quarter of block J against atom 64 I + i; the four 16-bit pieces meet in LDS and leave as 64 consecutive 64-bit
// masks nbmask[J][64 I + i] (see agbnp_common.h).  Same test as the tree workgroup's own: d^2 < rcut2, j younger.
__device__ void neighbor_tile(const PairArgs& P, int tile, bool write_ref = false) {  // write_ref: five-launch mode, see k_rows
  __shared__ double s_x[128], s_y[128], s_z[128];
  __shared__ unsigned short s_bits[4][64];
  const int t = threadIdx.x;
  // tile -> (I, J), I <= J, rows of the upper triangle laid end to end (no table: one dependent load less)
  const int nhb = (P.nh + 63) >> 6;
  int I = (int)(0.5 * ((double)(2 * nhb + 1) - sqrt((double)(2 * nhb + 1) * (double)(2 * nhb + 1) - 8.0 * (double)tile)));
  I = max(0, min(I, nhb - 1));
  while (I > 0 && I * nhb - I * (I - 1) / 2 > tile) I--;                   // (guards against rounding)
  while (I + 1 < nhb && (I + 1) * nhb - (I + 1) * I / 2 <= tile) I++;
  if (nhb == 0 || tile >= nhb * (nhb + 1) / 2) return;                      // (no such tile: nothing to search)
  const int J = I + (tile - (I * nhb - I * (I - 1) / 2));
  if (t < 128) {
    const int h = 64 * (t < 64 ? I : J) + (t & 63);
    const Pos3 r = heavy_position(P, h < P.nh ? h : P.nh - 1);
    s_x[t] = r.x;
    s_y[t] = r.y;
    s_z[t] = r.z;
  }
  __syncthreads();
  if (write_ref && I == J && t < 64 && 64 * I + t < P.nh) {  // the diagonal tile of a block also says where its atoms are now
    P.mask_ref[3 * (64 * I + t)] = s_x[t];
    P.mask_ref[3 * (64 * I + t) + 1] = s_y[t];
    P.mask_ref[3 * (64 * I + t) + 2] = s_z[t];
  }
  const int li = t & 63, jq = t >> 6, hi = 64 * I + li;
  const double xi = s_x[li], yi = s_y[li], zi = s_z[li];
  const int j0 = 16 * jq;  // each of the four waves takes a quarter of block J
  unsigned hits = 0;
#pragma unroll
  for (int jj = 0; jj < 16; jj++) {
    const int lj = j0 + jj, hj = 64 * J + lj;
    const double dx = s_x[64 + lj] - xi, dy = s_y[64 + lj] - yi, dz = s_z[64 + lj] - zi;
    if (hj < P.nh && hj > hi && dx * dx + dy * dy + dz * dz < P.mask_rcut2) hits |= 1u << jj;
  }
  s_bits[jq][li] = (unsigned short)hits;
  __syncthreads();
  if (jq == 0) {
    const unsigned long long m = (unsigned long long)s_bits[0][li] | ((unsigned long long)s_bits[1][li] << 16) |
                                 ((unsigned long long)s_bits[2][li] << 32) | ((unsigned long long)s_bits[3][li] << 48);
    P.nbmask[(size_t)J * ((size_t)nhb * 64) + hi] = hi < P.nh ? m : 0ull;
  }
}

__global__ __launch_bounds__(256) void k_prep(PairArgs P, int prep_blocks) {
#ifdef AGBNP_TIMING_PREP  // timing experiment only (results wrong): 1 = an empty launch of the same grid, 2 = no neighbour-mask
                          // tiles, 3 = the neighbour-mask tiles alone; the raw event interval of k_prep says what each part costs
  if (AGBNP_TIMING_PREP == 1) return;
  if (AGBNP_TIMING_PREP == 2 && (int)blockIdx.x >= prep_blocks) return;
  if (AGBNP_TIMING_PREP == 3 && (int)blockIdx.x < prep_blocks) return;
#endif
  if ((int)blockIdx.x >= prep_blocks) return neighbor_tile(P, blockIdx.x - prep_blocks);
  prep_atoms(P, blockIdx.x * blockDim.x + threadIdx.x, blockIdx.x == 0, false);
}

// five-launch mode: the neighbour masks alone (a launch of its own, only when the masks in hand have gone stale), and where
// the heavy atoms are while they are laid down
__global__ __launch_bounds__(256) void k_masks(PairArgs P, int ref_blocks) {
  if ((int)blockIdx.x >= ref_blocks) return neighbor_tile(P, blockIdx.x - ref_blocks);
  const int h = blockIdx.x * blockDim.x + threadIdx.x;
  if (h >= P.nh) return;
  const Pos3 r = heavy_position(P, h);
  P.mask_ref[3 * h] = r.x;
  P.mask_ref[3 * h + 1] = r.y;
  P.mask_ref[3 * h + 2] = r.z;
}

// ---- GB pairs, symmetric 64x64 tiles (all pairs, no cutoff) ------------------------------------------------
// A workgroup of four waves owns one tile (I <= J).  In every wave lane l keeps atom i = 64 I + l and its sums in
// registers and meets a quarter of block J in cyclic order: the static record of a j atom (position, charge, B,
// 1/B) is read from a doubled copy of the block in LDS (the step number is an immediate offset, no address
// arithmetic), the four running sums of the j atom travel round the wave by DPP wave rotation, so every (i, j)
// pair of the tile meets exactly once and both ends are updated from one evaluation of the pair terms (half the
// FP64 work of the row form, no vector memory in the loop).  A diagonal tile visits cyclic distances 1..32
// (distance 32 only from the lower half of the lanes).  The sums of the four waves meet in LDS and leave as one
// set of FP64 HBM atomics per tile.

__device__ __forceinline__ double rot1(double v) {  // lane l <- lane l+1 (mod 64); bound_ctrl: no "old" operand to set up
  const unsigned long long u = (unsigned long long)__double_as_longlong(v);
  const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)u, 0x134, 0xf, 0xf, true);
  const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(u >> 32), 0x134, 0xf, 0xf, true);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}


#ifndef AGBNP_PACK_ROUNDS_RULE
#define AGBNP_PACK_ROUNDS_RULE 1  // (0: diagnostic build without the rounds rule of packing_role)
#endif
constexpr size_t kRoleScratchBytes = 4352;  // LDS the two roles borrow from their host kernel's dynamic area
// ---- two single-workgroup roles, off the critical path:
//   energy:      fixed-order sum of every energy partial, ADDED to the caller's scalar; needs the GB stage's partials:
//                first workgroup of the chain-rule launch
//   bookkeeping: tree statistics, subtree order and forest packing of the NEXT evaluation; needs the tree's shapes
//                only: first workgroup of the GB launch
// (version 0 has no pair stages: both ride in the output launch)
__device__ __forceinline__ double block_sum_256(double v, double* red4) {
  v = wave_sum(v);
  const int t = threadIdx.x;
  if ((t & 63) == 0) red4[t >> 6] = v;
  __syncthreads();
  const double r = (red4[0] + red4[1]) + (red4[2] + red4[3]);  // fixed order -> reproducible
  __syncthreads();
  return r;
}

// DEVPAR: the instantiation of the five-launch mode's device-side parity (PairArgs::five == 2; the host launches it instead)
template <bool DEVPAR = false>
__device__ void energy_role(const PairArgs& P, int version, double* __restrict__ energy_out, double* __restrict__ components,
                            char* scratch) {  // scratch: kRoleScratchBytes of LDS
  const int t = threadIdx.x;
  // (five-launch mode: P is NOT rebased for this role -- the evaluation counter is asked for with the role's first loads and
  // used where the status block is, at its end; a rebase at the kernel's head put a cold round trip in front of the sums and
  // 0.3 us on the chain-rule launch)
  const int epoch_now = DEVPAR ? P.epoch[0] : -1;  // (otherwise the host has named the set: the offset below is 0)
    double* red4 = reinterpret_cast<double*>(scratch);
      // strided partial sums with 8 independent loads in flight per thread (a dependent load per trip would
      // cost one HBM/L2 latency each); the per-thread order is fixed, so the result is reproducible
      auto strided_sum = [&](const double* __restrict__ a, int count, int stride, int offset) {
        double acc = 0.0;
        for (int base = 0; base < count; base += 256 * 8) {
          double v[8];
  #pragma unroll
          for (int b = 0; b < 8; b++) {
            const int k = base + b * 256 + t;
            v[b] = k < count ? a[(size_t)k * stride + offset] : 0.0;
          }
  #pragma unroll
          for (int b = 0; b < 8; b++) acc += v[b];
        }
        return acc;
      };
      const int nslots = P.cur_nforests[0];  // cavity energies are per work slot
    const double ecav1 = strided_sum(P.epart, nslots, 2, 0), ecav2 = strided_sum(P.epart, nslots, 2, 1);
      double eatom = 0, egb = 0;
      if (version == 1) {
        eatom = strided_sum(P.e_atom, P.n, 1, 0);
        egb = strided_sum(P.egb_part, P.egb_parts, 1, 0);
      }
      const double o0 = block_sum_256(ecav1, red4), o1 = block_sum_256(ecav2, red4);
      const double o2 = block_sum_256(eatom, red4), o3 = block_sum_256(egb, red4);
      if (t == 0) {
        components[0] = o0;
        components[1] = o1;
        components[2] = o2;
        components[3] = o3;
        // An evaluation whose tree stage overflowed (capacity or forest packing: the words are final once
        // k_tree_cavity has ended) is incomplete: NOTHING of it reaches the caller's buffers -- here the energy,
        // in k_outputs the forces -- and it is entered in the sticky log that agbnp_hip_finish reports, so that
        // queued or graph-replayed evaluations cannot lose an overflow to the next evaluation's k_prep.  This
        // role runs exactly once per evaluation, after the tree stage.
        const int* es = P.estatus + (DEVPAR ? 16 * ((epoch_now + 1) & 1) : 0);  // (behind the GB launch: the counter has moved on)
        const int node = es[kStatNodeOverflow], atom = es[kStatAtomOverflow], pack = es[kStatPackOverflow];
        const int rowo = es[kStatRowOverflow];  // (final before the chain-rule launch: every row is built in the Born launch)
        const int order = es[kStatOrderStale];
        P.status[kStatStickyHealed] += es[kStatSpareForests];  // (forests healed inside k_tree_cavity: nothing withheld, a diagnostic)
        if ((node | atom | pack | rowo | order) == 0) {
          const double e = o0 + o1 + o2 + o3;
          if (P.omm.force_fixed == nullptr)
            energy_out[0] += e;
          else if (P.omm.energy_buffer && P.omm.energy_is_double)  // an OpenMM context's accumulator (GVolReduceTree.cl:112)
            static_cast<double*>(P.omm.energy_buffer)[P.omm.energy_slot] += e;
          else if (P.omm.energy_buffer)
            static_cast<float*>(P.omm.energy_buffer)[P.omm.energy_slot] += (float)e;
        } else {
          const int seq = P.status[kStatEvalSeq] - 1;  // k_prep counted this evaluation in
          P.status[kStatBadCount] += 1;
          P.status[kStatStickyNode] |= node;
          P.status[kStatStickyAtom] |= atom;
          P.status[kStatStickyPack] |= pack;
          P.status[kStatStickyRow] |= rowo;
          P.status[kStatStickyOrder] |= order;
          P.status[kStatStickySplit] = max(P.status[kStatStickySplit], es[kStatSplitWanted]);
          const int fo = es[kStatForestOverflow];
          P.status[kStatStickyForest] |= ((fo & 0xffff) ? 1 : 0) | ((fo >> 16) ? 2 : 0);
          if (seq >= 0 && seq < kStatBadBits) P.status[kStatBadBitmap + (seq >> 5)] |= 1 << (seq & 31);
        }
        if (P.host_status) {  // the host's window on the log (agbnp_hip_poll): the withheld count first, then the running number
          P.host_status[1] = P.status[kStatBadCount];
          __threadfence_system();
          P.host_status[0] = P.status[kStatEvalSeq];
        }
      }
}

__device__ __forceinline__ bool evaluation_overflowed(const int* __restrict__ status) {
  return (status[kStatNodeOverflow] | status[kStatAtomOverflow] | status[kStatPackOverflow] | status[kStatRowOverflow] | status[kStatOrderStale]) != 0;
}

// Bookkeeping for the NEXT evaluation (geometry changes little between MD steps, so this step's subtree shapes
// predict the next step's work): tree statistics, the subtrees sorted by weight (largest first), and their packing
// into forests = work slots of the tree kernels.
//   weight w = max(nodes / Tn, local atoms / Ta, 1/8) in units of 1/1024, with Tn, Ta = 90 % of the store's capacity.
//   A subtree with w > 1/2 is a forest of its own.  The others (sorted, descending) are dealt over Fs forests in
//   serpentine order (0, 1, .., Fs-1, Fs-1, .., 0, 0, 1, ..), which balances the forests to within one item.  Fs is the
//   smallest count that keeps the mean fill at or below 85 % and the roots per forest at or below 8 -- raised, if need
//   be, so that the total number of forests just fills a whole number of rounds of resident workgroups: the
//   kernels are bound by latency per workgroup, so F workgroups of n nodes cost about ceil(F / resident) * (a + b n),
//   and a round that is only partly filled costs as much as a full one.
// A forest that overflows anyway (kStatPackOverflow) makes the host repeat the evaluation on the one-subtree-per-slot
// packing written here, and every such event lowers the capacities assumed here by 15 % for good (pack_state);
// after six of them packing stays off.
// The bookkeeping is serial in nature (two sorts and a packing) and sits on ONE workgroup that shares its CU with the
// host kernel's tiles: every dependent LDS round trip costs it 0.2-0.4 us there, ~22 us in all on 1dwc, and a host kernel
// cannot end before its role does.  So it comes in two halves that each hide underneath a launch with time to spare:
//   packing_role  (first workgroup of the GB launch): subtree shapes -> work items -> forests (sorted by weight), the
//                 forests' predicted times;
//   dealing_role  (second workgroup of the chain-rule launch): forests ranked by predicted time -> work slots, every
//                 forest's items written into its slot's row.
// The shapes are fetched once, sixteen loads per thread in flight together, and packed into LDS; the passes are rolled
// loops over LDS without divisions (at most four parts, at most eight places per forest).
// kRounds: with the rounds rule (below).  The GB rows kernel of the fast mode hosts the role without it: its own walk sits at
// the register limit and spilled with the longer role inlined beside it.
template <bool kRounds>
__device__ void packing_role(const PairArgs& P, char* scratch, int scratch_bytes) {
  const int t = threadIdx.x;
  constexpr int kBins = 512, kBatch = 16;
  constexpr unsigned kUnit = 1024;
  unsigned long long* comb = reinterpret_cast<unsigned long long*>(scratch);  // [kBins] count << 32 | weight
  unsigned long long* part = comb + kBins;                                       // [4]
  int* imax = reinterpret_cast<int*>(part + 4);                                  // [24]
  static_assert(sizeof(unsigned long long) * (kBins + 4) + sizeof(int) * 24 <= kRoleScratchBytes, "role scratch");
  // the rest of the lent LDS: [nh] subtree shapes (nodes << 9 | local atoms) if they leave room for as many times, then
  // the predicted times of the forests
  const int lent = (scratch_bytes - (int)kRoleScratchBytes) / (int)sizeof(int);
  const bool shapes_in_lds = 2 * P.nh <= lent;
  int* lds_shape = reinterpret_cast<int*>(scratch + kRoleScratchBytes);
  int* lds_time = lds_shape + (shapes_in_lds ? P.nh : 0);
  const int lds_forests = lent - (shapes_in_lds ? P.nh : 0);
  if (shapes_in_lds) {
#pragma unroll 1
    for (int base = 0; base < P.nh; base += 256 * kBatch) {
      int2 sz[kBatch];
#pragma unroll
      for (int b = 0; b < kBatch; b++) sz[b] = P.sizes[min(base + b * 256 + t, P.nh - 1)];  // (clamped, unconditional: all in flight at once)
#pragma unroll
      for (int b = 0; b < kBatch; b++)
        if (base + b * 256 + t < P.nh) lds_shape[base + b * 256 + t] = (sz[b].x << 9) | sz[b].y;
    }
  }
  auto shape = [&](int h) {
    if (!shapes_in_lds) return P.sizes[h];
    const int v = lds_shape[h];
    return make_int2(v >> 9, v & 511);
  };
  for (int k = t; k < kBins; k += 256) comb[k] = 0ull;
  // every control word is asked for here, together, underneath the shapes (one cold round trip for all of them)
  const int st_node = P.estatus[kStatNodeOverflow], st_atom = P.estatus[kStatAtomOverflow], st_pack = P.estatus[kStatPackOverflow];
  const int st_forest = P.estatus[kStatForestOverflow], st_spare = P.estatus[kStatSpareForests];
  const int ps_level = P.pack_state[0], age = P.pack_state[1], ps_clean = P.pack_state[2];
  const int tot_planned = P.pack_state[4], max_planned = P.pack_state[5];
  const int ps_heat = P.pack_state[7], ps_need = P.pack_state[8];
  const bool overflow = (st_node | st_atom | st_pack) != 0;
  // level: how often the assumed capacity has been tightened by 15 %.  Round 6: a forest that outgrows its store is HEALED
  // inside k_tree_cavity (built again in smaller sets; kStatSpareForests counts them) and costs that one evaluation a few tens
  // of microseconds -- a tightened level can cost EVERY evaluation a whole round of forests (2clr: one round at level 0, two at
  // level 1: 139 -> 189 us).  So a healed forest only makes this role plan anew, from this evaluation's shapes, and adds to a
  // leaky counter (`heat`: + 16 per evaluation with healed forests, - 1 per evaluation); the level goes up when the counter says
  // that heals keep coming at more than about one evaluation in eight (128), or when a forest could NOT be healed (st_forest: no
  // spare slot left).  The level has a MEMORY: the clean EVALUATIONS asked for before a step is given back (`need`, at least
  // kPackRelax = 64; evaluations, not plans: ADVICE r05 -- plans also come from drift and after every fallback, so four of them
  // could be four evaluations) double every time the level has to go up again (cap 1024) and halve after four times that many
  // clean evaluations in a row -- round 5 gave a step back after four plans whatever had happened before, and a packing that
  // mispredicted was tried again, identically, 64 evaluations later.
  constexpr int kPackRelax = 64;
  const int need = max(ps_need, kPackRelax);
  int heat = max(ps_heat - 1, 0) + (st_spare != 0 ? 16 : 0);
  const bool hot = heat >= 128;
  if (hot) heat = 0;
  const bool tighten = st_forest != 0 || hot;
  const bool relax = st_pack == 0 && st_spare == 0 && !tighten && ps_level > 0 && ps_clean >= need;
  const int level = min(6, ps_level + (tighten ? 1 : 0) - (relax ? 1 : 0));
  const int need_next = tighten ? min(2 * need, 1024) : (ps_clean >= 4 * need ? max(need / 2, kPackRelax) : need);
  const bool pack = P.pack_enabled && !overflow && level < 6;
  float share = 0.9f;
  for (int k = 0; k < level; k++) share *= 0.85f;
  const float inv_tn = (float)kUnit / (share * (float)P.tree_node_cap), inv_ta = (float)kUnit / (share * (float)P.tree_atom_cap);
  // A big subtree can be shared by several work items (each expands a residue class of its level-2 nodes).  That
  // pays when the device holds every workgroup at once with room to spare (few subtrees: the kernel lasts as long
  // as its slowest workgroup, and idle slots are free); with more subtrees than resident workgroups the extra items
  // crowd the forests of the others and the kernel gets slower (measured on 1dwc: 57 -> 59 us), so they stay whole.
  const bool roomy = 2 * P.nh <= P.tree_slots;
  const int max_parts = !pack ? 1 : (roomy ? min(4, max(1, P.tree_slot_cap / max(P.nh, 1))) : min(P.split_big, 4));
  const int split_nodes = roomy ? 48 : (int)((float)P.split_permille * 0.001f * share * (float)P.tree_node_cap);
  auto inv_parts = [&](int parts) { return parts == 1 ? 1.0f : parts == 2 ? 0.5f : parts == 3 ? (1.0f / 3.0f) : 0.25f; };
  // Two reasons to share a subtree among several work items.  Speed: min(max_parts, 1 + nodes / split_nodes) (see above).
  // Fit (whatever the packing mode, also on the fallback after an overflow): an item holds the root, every level-2 node
  // and its share of the deeper ones; the parts go up, to at most four, until that prediction stays under 85 % of the
  // store -- so that a system with a few subtrees beyond the smallest store (2clr: 479 nodes) stays on it
  // instead of moving every forest to the next larger one (four workgroups per CU instead of five).
  // (the heaviest item of a shared subtree is taken to hold 1.35 x its even share of the deeper nodes; the bound tightens with
  // the level like everything else here.  An item that outgrows the store all the same is dealt with where it happens, round 6:
  // the tree launch builds a whole or two-way-shared subtree again as the parts of a four-way share, at once, and this role then
  // plans from the complete shapes of that evaluation; a three-way share records its subtree as "more than four stores' worth"
  // (kStatSplitWanted; the evaluation is void) and is handed to four items here; a four-way share that does not fit asks for
  // the next capacity variant)
  const float fit_nodes = 0.85f * (share * (1.0f / 0.9f)) * (float)P.tree_node_cap;
  auto parts_of = [&](int2 sz) {
    int p = min(max_parts, 1 + (sz.x >= split_nodes ? 1 : 0) + (sz.x >= 2 * split_nodes ? 1 : 0) + (sz.x >= 3 * split_nodes ? 1 : 0));
    if (P.split_fit) {  // the smallest p <= 4 with (1 + l2) + 1.35 deep / p <= fit_nodes, straight-line
      const int l2 = max(sz.y - 1, 0);
      const float need = 1.35f * (float)max(sz.x - 1 - l2, 0), room = fit_nodes - (float)(1 + l2);
      p = max(p, need <= room ? 1 : need <= 2.0f * room ? 2 : need <= 3.0f * room ? 3 : 4);
    }
    return p;
  };
  auto weight = [&](int2 sz, int parts) -> unsigned {  // of one work item of the subtree, 128..2047
    const int l2 = max(sz.y - 1, 0);
    const float nodes = (float)(1 + l2) + (float)(max(sz.x - 1 - l2, 0) + parts - 1) * inv_parts(parts);
    const float w = fmaxf(fmaxf(nodes * inv_tn, (float)sz.y * inv_ta), (float)(kUnit / 8));
    return (unsigned)fminf(w, 2047.0f);
  };
  __syncthreads();
  // Five-launch mode: this role is the one place where the device's evaluation counter advances (every thread of the
  // workgroup has taken its parity from it by now -- rebase_for_parity at the kernel's head; no other workgroup of this
  // launch reads it): the launches behind this one see the new value and count back by one, the next evaluation's first
  // launches see it as it is.
  if (P.five && t == 0) {
    atomicAdd(P.epoch, 1);
    atomicAdd(P.epoch_tree, 1);
  }
  // the tree statistics of THIS evaluation: plain reads, no histogram yet (three evaluations in four, or fifteen in sixteen,
  // need nothing else of this role)
  int tot = 0, mx = 0, ma = 0;
#pragma unroll 1
  for (int h0 = t; h0 < P.nh; h0 += 256 * 4) {  // (four LDS reads in flight)
    int2 sz[4];
#pragma unroll
    for (int c = 0; c < 4; c++) sz[c] = shape(min(h0 + 256 * c, P.nh - 1));
#pragma unroll
    for (int c = 0; c < 4; c++) {
      if (h0 + 256 * c >= P.nh) continue;
      tot += sz[c].x;
      mx = max(mx, sz[c].x);
      ma = max(ma, sz[c].y);
    }
  }
  for (int off = 32; off > 0; off >>= 1) {
    tot += __shfl_xor(tot, off, 64);
    mx = max(mx, __shfl_xor(mx, off, 64));
    ma = max(ma, __shfl_xor(ma, off, 64));
  }
  if ((t & 63) == 0) {
    imax[t >> 6] = mx;
    imax[4 + (t >> 6)] = ma;
    imax[8 + (t >> 6)] = tot;
  }
  __syncthreads();
  const int tot_now = (imax[8] + imax[9]) + (imax[10] + imax[11]);
  const int max_now = max(max(imax[0], imax[1]), max(imax[2], imax[3]));
  // The tree statistics are taken at every evaluation.  The PACKING is planned anew when the one in use is not a plan at
  // all (a fresh context, or the fallback after an overflow: its age says so), when this evaluation overflowed, when the
  // trees have DRIFTED from the shapes the packing was planned for (total nodes by more than 1.5 %, or the largest subtree
  // grown by more than 6 %), and otherwise at every replan_every-th evaluation: geometries change little between MD steps,
  // a misprediction is caught by the overflow protocol whatever the packing's age, and everything below this point is
  // latency (a plan ends ~5 us after the GB launch's own work items on 1dwc) that small systems and version 0 (whose
  // k_outputs launch lasts as long as this role) cannot hide at all.
  const bool drifted = abs(tot_now - tot_planned) * 64 > tot_planned || max_now * 16 > max_planned * 17;
  const bool plan = overflow || age + 1 >= P.replan_every || drifted || st_spare != 0;  // (a healed forest: the shapes have left the plan behind)
  PAIR_STAMP(1, 7);
  if (t == 0) {
    P.estatus[kStatTotalNodes] = tot_now;
    P.estatus[kStatMaxNodes] = max_now;
    P.estatus[kStatMaxAtoms] = max(max(imax[4], imax[5]), max(imax[6], imax[7]));
    P.pack_state[0] = level;
    // (after an overflow the fallback written below is no plan: the next clean evaluation plans anew)
    if (P.pack_enabled != 3) P.pack_state[1] = plan ? (overflow ? P.replan_every : 0) : age + 1;
    P.pack_state[2] = (tighten || relax || st_spare != 0 || overflow) ? 0 : min(ps_clean + 1, 1 << 20);  // clean evaluations in a row
    P.pack_state[7] = heat;
    P.pack_state[8] = need_next;
    if (plan) P.pack_state[3] += 1;  // (plans so far: a diagnostic)
    if (plan && !overflow) P.pack_state[4] = tot_now, P.pack_state[5] = max_now;  // (the shapes this plan is made for)
    if (!plan) {
      P.estatus[kStatForests] = P.nforests[0];       // (the packing stays)
      P.forest_time[P.tree_slot_cap] = 2;           // tells dealing_role that there is nothing to deal
    }
  }
  if (!plan) return;
  // (count, weight) histogram of the work items over the weight bins
#pragma unroll 1
  for (int h0 = t; h0 < P.nh; h0 += 256 * 4) {
    int2 sz[4];
#pragma unroll
    for (int c = 0; c < 4; c++) sz[c] = shape(min(h0 + 256 * c, P.nh - 1));
#pragma unroll
    for (int c = 0; c < 4; c++) {
      if (h0 + 256 * c >= P.nh) continue;
      const int parts = parts_of(sz[c]);
      const unsigned w = weight(sz[c], parts);
      atomicAdd(&comb[kBins - 1 - (w >> 2)], ((unsigned long long)parts << 32) | (w * (unsigned)parts));
    }
  }
  __syncthreads();
  // exclusive scan of the (count, weight) histogram (bins are in descending weight order): thread t owns bins 2t, 2t+1
  const unsigned long long h0 = comb[2 * t], h1 = comb[2 * t + 1];
  unsigned long long incl = h0 + h1;
  for (int off = 1; off < 64; off <<= 1) {
    const unsigned long long v = __shfl_up(incl, off, 64);
    if ((t & 63) >= off) incl += v;
  }
  __syncthreads();
  if ((t & 63) == 63) part[t >> 6] = incl;
  __syncthreads();
  unsigned long long before = 0ull;
  for (int w = 0; w < (t >> 6); w++) before += part[w];
  const unsigned long long excl = before + incl - (h0 + h1);
  comb[2 * t] = excl;
  comb[2 * t + 1] = excl + h0;
  // class boundaries in the sorted order (whole bins): A = weight > 3/4 (alone), B = (1/2, 3/4] (takes one partner
  // of weight <= 1/4 from the light end of the order, as long as there are any), the rest is dealt over Fs forests
  auto at_bin = [&](int bin, int slot) {
    if (2 * t == bin || 2 * t + 1 == bin) {
      const unsigned long long e = (2 * t == bin) ? excl : excl + h0;
      imax[slot] = (int)(e >> 32);
      imax[slot + 1] = (int)(unsigned)(e & 0xffffffffull);
    }
  };
  at_bin(kBins - 1 - 192, 12);  // items / weight above 3/4
  at_bin(kBins - 1 - 128, 14);  // above 1/2
  at_bin(kBins - 1 - 64, 16);   // above 1/4
  if (t == 255) {
    imax[18] = (int)((excl + h0 + h1) >> 32);                       // work items
    imax[19] = (int)(unsigned)((excl + h0 + h1) & 0xffffffffull);  // total weight
  }
  __syncthreads();
  const int nitems = imax[18];
  const bool forests = pack && P.pack_enabled != 2;  // (2: every work item alone, big subtrees still shared)
  const int na = forests ? imax[12] : nitems;      // forests of one heavy item
  const int nab = forests ? imax[14] : nitems;     // ... plus the forests led by a class-B item
  const int nb = nab - na;
  const int npair = forests ? min(nb, nitems - imax[16]) : 0;  // class-B items that get a light partner
  const int nc = nitems - nab - npair;          // items dealt over the remaining forests
  const int round = max(1, (int)(((long long)P.tree_slots * P.round_permille) / 1000));  // a round of resident workgroups, a few % spare
  int fs = 0, rounds_classes = (nab + round - 1) / round;
  if (nc > 0) {
    const unsigned wc = (unsigned)imax[19] - (unsigned)imax[15];               // (the partners' weight is left in: safe side)
    const int fmin = max((int)((wc + 869u) / 870u), (nc + 7) / 8);            // mean fill <= 85 %, at most 8 roots
    rounds_classes = (nab + fmin + round - 1) / round;                          // whole rounds that hold fmin
    fs = min(nc, max(fmin, rounds_classes * round - nab));
  }
  const int nf = nab + fs;
  const int full = fs > 0 ? nc / fs : 0, rem = fs > 0 ? nc % fs : 0;  // full serpentine rounds, items of the last one
  auto small_start = [&](int f) {  // first position (among the dealt items) of forest f of the last class
    return f * full + ((full & 1) ? max(0, f - (fs - rem)) : min(f, rem));
  };
  auto forest_first = [&](int f) {  // first work item of forest f
    if (f < na) return f;
    if (f < nab) {
      const int j = f - na;
      return na + 2 * min(j, npair) + max(0, j - npair);
    }
    return f < nf ? nab + npair + small_start(f - nab) : nitems;
  };
  auto class_place = [&](int pos, int* place) {  // position in descending weight order -> forest, place (the classes' rule)
    if (pos < nab) {
      *place = 0;  // a forest of its own, or the leader of a class-B forest
      return pos;
    }
    if (pos >= nitems - npair) {
      *place = 1;  // the lightest item joins the heaviest class-B item
      return na + (nitems - 1 - pos);
    }
    const int k = pos - nab;
    int r = 0;  // k / fs: the serpentine round, at most kMaxItems - 1
#pragma unroll
    for (int q = 1; q < kMaxItems; q++) r += k >= q * fs ? 1 : 0;
    const int idx = k - r * fs;
    *place = r;
    return nab + ((r & 1) ? fs - 1 - idx : idx);
  };
  // predicted time of a work item (fit of a workgroup timeline: 0.09 us per node, 0.34 per local atom, 1.6 per root;
  // units of 0.01 us)
  auto item_time = [&](int2 sz, int parts) {
    const int l2 = max(sz.y - 1, 0);
    return 9 * (1 + l2 + (int)((float)max(sz.x - 1 - l2, 0) * inv_parts(parts))) + 34 * sz.y + 160;
  };
  PAIR_STAMP(1, 8);
  if (P.pack_enabled == 3) return;  // (diagnostics: the packing is frozen from the host)
  // ---- The classes above waste room on mid-size systems: 2clr's 3358 work items weigh 1120 stores' worth (most of them
  // bound by the 64 local atoms of a store, not by its nodes), yet 445 class-A/B forests with one light partner each and
  // the 85 % mean fill of the rest make 1477 forests -- two rounds of 1280, a third full.  When the total weight says a
  // round can be saved, a second rule is tried: THE ROUNDS RULE.  The F heaviest items lead a forest each (F = the rounds
  // the weight needs, filled); then, pass by pass, every forest that still has room for the HEAVIEST item left (a bound
  // that decouples the forests' decisions from each other) is open, and the open forests take the next items of the
  // sorted order, the heaviest open forest the lightest item of the batch.  A pass is a count, a block scan and one
  // look-up per open forest; at most seven passes.  If an item is left over (no forest open) the classes' packing is
  // used after all.  Needs the sorted order as an array: the items' weights by position in LDS, their identity and
  // predicted time in global scratch (pack_items), the forests' running sums in LDS.
  const int rounds_weight = (int)(((unsigned)imax[19] + (unsigned)(kUnit * round) - 1u) / (unsigned)(kUnit * round));
  const int F = min(nitems, rounds_weight * round);
  const int region = F;  // ints: running sums (weight | items << 16), later the predicted times (if the rule fails and the
                         // classes' forests outnumber it, they are handed over unranked: a rare path of a rare path)
  unsigned short* ws = reinterpret_cast<unsigned short*>(lds_time + region);  // [nitems] weight by sorted position
  unsigned short* where = ws + ((nitems + 1) & ~1);                            // [nitems] forest << 3 | place by sorted position
  const bool try_rounds = forests && nc > 0 && rounds_weight < rounds_classes && F <= 8191 && P.pack_items != nullptr &&
                          region + (nitems + 1) / 2 * 2 <= lds_forests && kRounds && AGBNP_PACK_ROUNDS_RULE;
  if (!try_rounds) {
#pragma unroll 1
    for (int f = t; f <= nf; f += 256) P.forest_start[f] = forest_first(f);
    // (the forests are ranked by predicted time when the times fit the lent LDS: always, short of ~3000 forests)
    const bool rank_by_time = nf <= lds_forests;
    if (rank_by_time)
      for (int f = t; f < nf; f += 256) lds_time[f] = 0;
    __syncthreads();
    if (t == 0) {
      P.nforests[0] = nf;
      P.estatus[kStatForests] = nf;
      P.forest_time[P.tree_slot_cap] = rank_by_time ? 1 : 0;  // (word behind the times: are they there)
    }
    PAIR_STAMP(1, 9);
    // sorted order -> place inside the forests.  Four subtrees per thread and trip, their LDS round trips (shape, then the
    // ranked add) in flight together: the role's time is its chain of dependent LDS latencies.
    constexpr int kChains = 4;
#pragma unroll 1
    for (int h0 = t; h0 < P.nh; h0 += 256 * kChains) {
      int2 sz[kChains];
      int parts[kChains];
      unsigned long long v[kChains];
#pragma unroll
      for (int c = 0; c < kChains; c++) sz[c] = shape(min(h0 + 256 * c, P.nh - 1));
#pragma unroll
      for (int c = 0; c < kChains; c++) {
        parts[c] = parts_of(sz[c]);
        const unsigned w = weight(sz[c], parts[c]);
        v[c] = 0ull;
        if (h0 + 256 * c < P.nh) v[c] = atomicAdd(&comb[kBins - 1 - (w >> 2)], ((unsigned long long)parts[c] << 32) | (w * (unsigned)parts[c]));
      }
#pragma unroll
      for (int c = 0; c < kChains; c++) {
        const int h = h0 + 256 * c;
        if (h >= P.nh) continue;
        const int tm = item_time(sz[c], parts[c]);
#pragma unroll 1
        for (int part = 0; part < parts[c]; part++) {
          int place;
          const int forest = class_place((int)(v[c] >> 32) + part, &place);  // position in descending weight order -> forest
          P.order[kMaxItems * forest + place] = h | (part << 24) | ((parts[c] - 1) << 26);
          if (rank_by_time) atomicAdd(&lds_time[forest], tm);
        }
      }
    }
    PAIR_STAMP(1, 10);
    __syncthreads();
    if (rank_by_time)
      for (int f = t; f < nf; f += 256) P.forest_time[f] = lds_time[f];
    return;
  }
  // ---- the rounds rule.  (i) the sorted order, materialised
#pragma unroll 1
  for (int h = t; h < P.nh; h += 256) {
    const int2 sz = shape(h);
    const int parts = parts_of(sz);
    const unsigned w = weight(sz, parts);
    const int first = (int)(atomicAdd(&comb[kBins - 1 - (w >> 2)], ((unsigned long long)parts << 32) | (w * (unsigned)parts)) >> 32);
    const int tm = item_time(sz, parts);
    for (int part = 0; part < parts; part++) {
      ws[first + part] = (unsigned short)w;
      P.pack_items[first + part] = make_int2(h | (part << 24) | ((parts - 1) << 26), tm);
    }
  }
  __syncthreads();  // (also makes the global stores of this workgroup visible to its own later loads)
  // (ii) the passes.  Thread t owns forests [t per, (t + 1) per): ranks follow the forests' order
  const int per = (F + 255) / 256, f_lo = min(F, t * per), f_hi = min(F, f_lo + per);
  for (int f = f_lo; f < f_hi; f++) lds_time[f] = (int)ws[f] | (1 << 16);
  int taken = F;  // items placed so far = the next position
  bool stuck = false;
#pragma unroll 1
  for (int pass = 1; pass < kMaxItems && taken < nitems && !stuck; pass++) {
    const int wtop = ws[taken];
    int open = 0;
    for (int f = f_lo; f < f_hi; f++) open += (lds_time[f] & 0xffff) + wtop <= (int)kUnit ? 1 : 0;
    int incl = open;  // block scan of the open counts (wave scan, then the four wave totals)
    for (int off = 1; off < 64; off <<= 1) {
      const int v = __shfl_up(incl, off, 64);
      if ((t & 63) >= off) incl += v;
    }
    __syncthreads();  // (the previous pass is done with imax[20..23])
    if ((t & 63) == 63) imax[20 + (t >> 6)] = incl;
    __syncthreads();
    int before = 0, total = 0;
    for (int w = 0; w < 4; w++) {
      before += w < (t >> 6) ? imax[20 + w] : 0;
      total += imax[20 + w];
    }
    const int take = min(total, nitems - taken);
    stuck = total == 0;
    int rank = before + incl - open;
    for (int f = f_lo; f < f_hi; f++) {
      const int sum = lds_time[f];
      if ((sum & 0xffff) + wtop <= (int)kUnit) {
        if (rank < take) {
          const int pos = taken + (take - 1 - rank);  // the heaviest open forest takes the lightest item of the batch
          where[pos] = (unsigned short)((f << 3) | (sum >> 16));
          lds_time[f] = sum + (int)ws[pos] + (1 << 16);
        }
        rank++;
      }
    }
    taken += take;
  }
  __syncthreads();
  const bool placed = taken >= nitems;  // (the same on every thread)
  const int nf_used = placed ? F : nf;
  if (placed) {  // forest_start from the forests' item counts
    int mine = 0;
    for (int f = f_lo; f < f_hi; f++) mine += lds_time[f] >> 16;
    int incl = mine;
    for (int off = 1; off < 64; off <<= 1) {
      const int v = __shfl_up(incl, off, 64);
      if ((t & 63) >= off) incl += v;
    }
    if ((t & 63) == 63) imax[20 + (t >> 6)] = incl;
    __syncthreads();
    int run = incl - mine;
    for (int w = 0; w < (t >> 6); w++) run += imax[20 + w];
    for (int f = f_lo; f < f_hi; f++) {
      P.forest_start[f] = run;
      run += lds_time[f] >> 16;
    }
    if (t == 255) P.forest_start[F] = nitems;
    __syncthreads();
  } else {
#pragma unroll 1
    for (int f = t; f <= nf; f += 256) P.forest_start[f] = forest_first(f);
  }
  const bool rank_by_time = nf_used <= region;
  if (rank_by_time)
    for (int f = t; f < nf_used; f += 256) lds_time[f] = 0;
  __syncthreads();
  if (t == 0) {
    P.nforests[0] = nf_used;
    P.estatus[kStatForests] = nf_used;
    P.forest_time[P.tree_slot_cap] = rank_by_time ? 1 : 0;
  }
  PAIR_STAMP(1, 9);
  // (iii) every position hands its item to its forest
  {
    constexpr int kBatch = 4;
#pragma unroll 1
    for (int base = 0; base < nitems; base += 256 * kBatch) {
      int2 it[kBatch];
#pragma unroll
      for (int b = 0; b < kBatch; b++) it[b] = P.pack_items[min(base + b * 256 + t, nitems - 1)];
#pragma unroll
      for (int b = 0; b < kBatch; b++) {
        const int pos = base + b * 256 + t;
        if (pos >= nitems) continue;
        int forest, place;
        if (placed) {
          const int code = where[pos];
          forest = pos < F ? pos : code >> 3;
          place = pos < F ? 0 : code & 7;
        } else {
          forest = class_place(pos, &place);
        }
        P.order[kMaxItems * forest + place] = it[b].x;
        if (rank_by_time) atomicAdd(&lds_time[forest], it[b].y);
      }
    }
  }
  PAIR_STAMP(1, 10);
  __syncthreads();
  if (rank_by_time)
    for (int f = t; f < nf_used; f += 256) P.forest_time[f] = lds_time[f];
}

// Second half.  Work slot s runs on CU s mod ncus (observed: the dispatcher deals workgroups round-robin over the CUs),
// and a CU's workgroups slow each other down, so the forests are ranked by predicted time (descending, histogram sort)
// and dealt over the CUs in serpentine order: row 0 left to right, row 1 right to left, ...  A forest's items go into
// the row of its work slot (fixed stride: a tree workgroup fetches its items and their number in ONE round trip, with
// no slot -> forest indirection in front of it).
__device__ void dealing_role(const PairArgs& P, char* scratch, int scratch_bytes) {
  const int t = threadIdx.x;
  constexpr int kBins = 512;
  unsigned long long* comb = reinterpret_cast<unsigned long long*>(scratch);  // [kBins]
  unsigned long long* part = comb + kBins;                                       // [4]
  int* lds_time = reinterpret_cast<int*>(scratch + kRoleScratchBytes);          // [nf]
  const int lent = (scratch_bytes - (int)kRoleScratchBytes) / (int)sizeof(int);
  if (P.pack_enabled == 3) return;  // (diagnostics: the packing is frozen from the host)
  const int nf = min(P.nforests[0], P.tree_slot_cap);
  const int told = P.forest_time[P.tree_slot_cap];  // packing_role's word: 0 = forests without times, 1 = with times, 2 = no new packing
  if (told == 2) return;
  const bool ranked = told == 1 && nf <= lent;
  auto hand_over = [&](int f, int slot) {
    const int2 se = make_int2(P.forest_start[f], P.forest_start[f + 1]);
    const int4* src = reinterpret_cast<const int4*>(P.order + (size_t)kMaxItems * f);
    int4 lo = src[0], hi = src[1];
    const int m = se.y - se.x;  // -1 behind the forest's own items (the working copy may hold older ones)
    lo.x = 0 < m ? lo.x : -1, lo.y = 1 < m ? lo.y : -1, lo.z = 2 < m ? lo.z : -1, lo.w = 3 < m ? lo.w : -1;
    hi.x = 4 < m ? hi.x : -1, hi.y = 5 < m ? hi.y : -1, hi.z = 6 < m ? hi.z : -1, hi.w = 7 < m ? hi.w : -1;
    int4* dst = reinterpret_cast<int4*>(P.rows + (size_t)kRowStride * slot);
    dst[0] = lo, dst[1] = hi;
    dst[2] = make_int4(m, 0, 0, 0);
    if (P.row_atoms) {  // five-launch mode: the atoms of the items' roots (the tree reads the caller's positions itself)
      const int it[kMaxItems] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
      int4 a0, a1;
      int at[kMaxItems];
      // (an OpenMM context's posq: the root's SLOT in the context's order -- the next evaluation is taken to come through the same
      // entry point; the host rewrites the words when it does not, engine.hip sync_row_atoms)
      const int* __restrict__ where = P.in.posq ? P.in.hslot : P.h2a;
#pragma unroll
      for (int k = 0; k < kMaxItems; k++) at[k] = it[k] >= 0 ? where[it[k] & 0xffffff] : 0;
      a0 = make_int4(at[0], at[1], at[2], at[3]), a1 = make_int4(at[4], at[5], at[6], at[7]);
      int4* da = reinterpret_cast<int4*>(P.row_atoms + (size_t)kMaxItems * slot);
      da[0] = a0, da[1] = a1;
    }
  };
  static_assert(kMaxItems == 8 && kRowStride >= kMaxItems + 4, "two 16-byte words of items, then the word with their number");
  if (!ranked) {  // as they come
#pragma unroll 1
    for (int f = t; f < nf; f += 256) hand_over(f, f);
    return;
  }
  for (int k = t; k < kBins; k += 256) comb[k] = 0ull;
  {  // the times: eight loads per thread in flight together
    constexpr int kBatch = 8;
#pragma unroll 1
    for (int base = 0; base < nf; base += 256 * kBatch) {
      int tm[kBatch];
#pragma unroll
      for (int b = 0; b < kBatch; b++) tm[b] = P.forest_time[min(base + b * 256 + t, nf - 1)];
#pragma unroll
      for (int b = 0; b < kBatch; b++)
        if (base + b * 256 + t < nf) lds_time[base + b * 256 + t] = tm[b];
    }
  }
  __syncthreads();
  auto time_bin = [&](int tm) { return kBins - 1 - min(kBins - 1, tm >> 4); };
#pragma unroll 1
  for (int f = t; f < nf; f += 256) atomicAdd(&comb[time_bin(lds_time[f])], 1ull);
  __syncthreads();
  {
    const unsigned long long c0 = comb[2 * t], c1 = comb[2 * t + 1];
    unsigned long long inc2 = c0 + c1;
    for (int off = 1; off < 64; off <<= 1) {
      const unsigned long long v = __shfl_up(inc2, off, 64);
      if ((t & 63) >= off) inc2 += v;
    }
    __syncthreads();
    if ((t & 63) == 63) part[t >> 6] = inc2;
    __syncthreads();
    unsigned long long before2 = 0ull;
    for (int w = 0; w < (t >> 6); w++) before2 += part[w];
    const unsigned long long ex2 = before2 + inc2 - (c0 + c1);
    comb[2 * t] = ex2;
    comb[2 * t + 1] = ex2 + c0;
  }
  __syncthreads();
  const int ncu = max(P.ncus, 1), last_row = nf / ncu, last_width = nf - last_row * ncu;
#pragma unroll 1
  for (int f = t; f < nf; f += 256) {
    const int r = (int)atomicAdd(&comb[time_bin(lds_time[f])], 1ull);  // rank by descending predicted time
    const int q = r / ncu, p = r - q * ncu;
    const int width = q < last_row ? ncu : last_width;
    hand_over(f, q * ncu + ((q & 1) ? width - 1 - p : p));
  }
}

// Epilogue shared by the two tile kernels: the four waves of a tile hold partial sums for the same 64 i atoms
// (lane = atom) and, rotated, for the same 64 j atoms.  They meet in LDS and leave as ONE set of FP64 HBM
// atomics per tile (8 rows of 64), added in a fixed order within the tile.  Float atomics execute at the memory
// side at a fixed chip-wide byte rate, so the bytes they carry are what has to be kept small.
struct TileSums {
  double red[4][8][64];
};
__device__ __forceinline__ void tile_sums_store(TileSums& T, int wave, int lane, int jslot, const double (&vi)[4], const double (&vj)[4]) {
  for (int q = 0; q < 4; q++) {
    T.red[wave][q][lane] = vi[q];
    T.red[wave][4 + q][jslot] = vj[q];
  }
}
__device__ __forceinline__ double tile_sums_fold(const TileSums& T, int row, int lane) {
  return (T.red[0][row][lane] + T.red[1][row][lane]) + (T.red[2][row][lane] + T.red[3][row][lane]);
}

// ---- GB strips: 128 x 64 --------------------------------------------------------------------------------------------
// Away from the diagonal a workgroup takes a STRIP: blocks I0 and I0 + 1 against block J.  A lane keeps TWO i atoms
// (one of each block) and meets the 64 j atoms once: the j record (three LDS reads) and the wave rotation of the j sums
// (eight DPP moves) -- a fifth of the instructions of a pair step -- are paid once for two pairs, and the i-side
// prologue / epilogue and the J atomics once for two tiles.  (The kernel runs at its instruction-issue bound: fewer
// instructions per pair is the only lever.)
struct StripSums {
  double red[4][12][64];  // per wave: rows 0-3 block I0 {fx, fy, fz, Y}, 4-7 block I0 + 1, 8-11 block J
};
constexpr int kGbStripFlag = 1 << 24;  // work item = I0 | J << 12 | flag

// Squared gap between the bounding box of block J and the nearer of the boxes of blocks I0, I0 + 1 (atom-order boxes of k_prep)
__device__ __forceinline__ double strip_gap2(const PairArgs& P, int I0, int J) {
  double gmin = 1e300;
  for (int b = 0; b < 2; b++) {
    double gap2 = 0.0;
    for (int d = 0; d < 3; d++) {
      const double g = fmax(0.0, fmax(P.abox[6 * J + d] - P.abox[6 * (I0 + b) + 3 + d], P.abox[6 * (I0 + b) + d] - P.abox[6 * J + 3 + d]));
      gap2 += g * g;
    }
    gmin = fmin(gmin, gap2);
  }
  return gmin;
}
// max over the wave of the HIGH words of positive doubles (monotonic in the value): 32-bit DPP steps; lane 63 holds it
__device__ __forceinline__ int wave_max_hi_to_lane63(int m) {
  m = max(m, __builtin_amdgcn_update_dpp(m, m, 0x111, 0xf, 0xf, false));  // row_shr:1 (a lane without a source keeps its own)
  m = max(m, __builtin_amdgcn_update_dpp(m, m, 0x112, 0xf, 0xf, false));  // row_shr:2
  m = max(m, __builtin_amdgcn_update_dpp(m, m, 0x114, 0xf, 0xf, false));  // row_shr:4
  m = max(m, __builtin_amdgcn_update_dpp(m, m, 0x118, 0xf, 0xf, false));  // row_shr:8 -> lane 15 of every row
  m = max(m, __builtin_amdgcn_update_dpp(m, m, 0x142, 0xa, 0xf, false));  // row_bcast:15 -> rows 1, 3
  m = max(m, __builtin_amdgcn_update_dpp(m, m, 0x143, 0xc, 0xf, false));  // row_bcast:31 -> rows 2, 3
  return m;
}

// kFar (reference mode, systems large enough to have such strips: the host picks the instantiation): a strip whose blocks
// are so far apart that exp(-d^2 / (4 B_i B_j)) < 2^-60 for every one of its pairs -- box gap^2 > 4 * 60 ln2 * Bmax_J *
// Bmax_I, the Born radii of its 192 atoms are in the prologue's registers -- walks a Coulomb-only loop: the pair term
// 1/sqrt(d^2 + B_i B_j eta) IS 1/sqrt(d^2) to FP64 rounding there (B_i B_j eta < 2^-67 d^2), the direct force's factor
// 1 - eta/4 IS 1, and the Y term (< 2^-60 of a pair's Coulomb energy) is dropped: no exp2, no Born-radius products, no Y
// sums -- about half the instructions of a pair step (ReferenceAGBNPKernels.cpp:477-499 evaluates the full formula for
// every pair; the 16 608-atom lattice has 71 % of its tiles out there, 1dwc none).
constexpr double kGbFarFactor = 4.0 * 60.0 * 0.69314718055994530942;
template <bool kCut, bool kFar>
__device__ __forceinline__ void gb_strip(int n, int I0, int J, const double4* __restrict__ aposq, const double* __restrict__ born_part,
                                         const double* __restrict__ inv_rvdw, double* __restrict__ gb_rows, double* __restrict__ egb_out,
                                         const PairArgs& P, char* s_area, double* s_e, int* s_bmax) {
  double2* const s_xy = reinterpret_cast<double2*>(s_area);  // block J twice over
  double2* const s_zq = s_xy + 128;
  double2* const s_bb = s_zq + 128;
  double2* const s_i = s_bb + 128;  // blocks I0, I0 + 1: [block][{x, y}, {z, q}, {B, -log2(e)/(4 B)}][64]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (kCut) {  // fast mode: both tiles of the strip beyond the cutoff
    if (strip_gap2(P, I0, J) >= P.gb_cut2) {
      if (threadIdx.x == 0) egb_out[0] = 0.0;
      return;
    }
  }
  const double far_gap2 = kFar ? strip_gap2(P, I0, J) : 0.0;  // (scalar loads, underneath the records)
  PAIR_STAMP_WHERE(1, I0 | (J << 12) | kGbStripFlag);
  // the Y sums leave by pair-order slot (the chain-rule stage reads them so): the wave that will add them asks for
  // the slots of its three blocks now
  // (unconditional loads: a choice here would have to wait for them)
  const int ysa = P.a2s[min(64 * I0 + lane, n - 1)], ysc = P.a2s[min(64 * I0 + 64 + lane, n - 1)], ysj = P.a2s[min(64 * J + lane, n - 1)];
  double beta_a = 0.0, beta_c = 0.0, beta_j = 0.0;  // row form of the chain rule: wave 3 (idle here) turns its Y totals into bw shares
  if (P.rows_on && wave == 3) {
    const int ka = min(64 * I0 + lane, n - 1), kc = min(64 * I0 + 64 + lane, n - 1), kj = min(64 * J + lane, n - 1);
    const double ra = inv_rvdw[ka], rc = inv_rvdw[kc], rj = inv_rvdw[kj], pa = born_part[ka], pc = born_part[kc], pj = born_part[kj];
    beta_a = bw_beta(born_radius(ra, pa));
    beta_c = bw_beta(born_radius(rc, pc));
    beta_j = bw_beta(born_radius(rj, pj));
  }
  if (wave < 3) {  // wave 0 prepares block J, waves 1 and 2 the two i blocks
    const int a = 64 * (wave == 0 ? J : I0 + wave - 1) + lane;
    const bool va = a < n;
    const int ac = va ? a : n - 1;
    const double4 pa = aposq[ac];
    const BornRadius bra = born_radius(inv_rvdw[ac], born_part[ac]);
    const double qa = va ? pa.w : 0.0;  // zero charge switches a padded atom off
    if (wave == 0) {
      s_xy[lane] = s_xy[lane + 64] = make_double2(pa.x, pa.y);
      s_zq[lane] = s_zq[lane + 64] = make_double2(pa.z, qa);
      s_bb[lane] = s_bb[lane + 64] = make_double2(bra.br, bra.inv_br);
    } else {
      double2* r = s_i + (wave - 1) * 192;
      r[lane] = make_double2(pa.x, pa.y);
      r[64 + lane] = make_double2(pa.z, qa);
      r[128 + lane] = make_double2(bra.br, (-0.25 * 1.4426950408889634074) * bra.inv_br);
    }
    if (kFar) {  // an upper bound of the block's largest Born radius: (max high word + 1) << 32
      const int m = wave_max_hi_to_lane63(__double2hiint(bra.br));
      if (lane == 63) s_bmax[wave] = m + 1;
    }
  }
  __syncthreads();
  PAIR_STAMP(1, 1);
  __builtin_amdgcn_s_setprio(0);  // (the walk: see k_gb_tiles)
  const int start = 16 * wave;  // the four waves take a quarter of the cyclic distances each
  const double2 axy = s_i[lane], azq = s_i[64 + lane], abc = s_i[128 + lane];
  const double2 cxy = s_i[192 + lane], czq = s_i[256 + lane], cbc = s_i[320 + lane];
  const int base = (lane + start) & 63;
  const double2* __restrict__ jxy = s_xy + base;
  const double2* __restrict__ jzq = s_zq + base;
  const double2* __restrict__ jbb = s_bb + base;
  double fxa = 0, fya = 0, fza = 0, ya = 0, fxc = 0, fyc = 0, fzc = 0, yc = 0, fxj = 0, fyj = 0, fzj = 0, yj = 0, e = 0;
  bool far = false;
  if (kFar) {
    const double bj = __hiloint2double(s_bmax[0], 0), bi = __hiloint2double(max(s_bmax[1], s_bmax[2]), 0);
    far = __builtin_amdgcn_readfirstlane(far_gap2 > kGbFarFactor * bj * bi ? 1 : 0) != 0;  // (the same for every lane of the workgroup)
#ifdef AGBNP_TIMING_ALL_FAR  // timing experiment only (results wrong): what the launch costs if EVERY strip took the short walk
    far = true;
#endif
  }
  if (kFar && far) {
#pragma unroll 4
    for (int k = 0; k < 16; k++) {  // Coulomb only (see above)
      const double2 xy = jxy[k], zq = jzq[k];
      const double dxa = xy.x - axy.x, dya = xy.y - axy.y, dza = zq.x - azq.x;
      const double d2a = fma(dza, dza, fma(dya, dya, dxa * dxa));
      const double fa = rsqrt_pos(d2a);
      const double s1a = (azq.y * zq.y) * fa;
      const double mwa = s1a * (fa * fa);
      const double dxc = xy.x - cxy.x, dyc = xy.y - cxy.y, dzc = zq.x - czq.x;
      const double d2c = fma(dzc, dzc, fma(dyc, dyc, dxc * dxc));
      const double fc = rsqrt_pos(d2c);
      const double s1c = (czq.y * zq.y) * fc;
      const double mwc = s1c * (fc * fc);
      e += s1a + s1c;
      fxa = fma(dxa, mwa, fxa);
      fya = fma(dya, mwa, fya);
      fza = fma(dza, mwa, fza);
      fxc = fma(dxc, mwc, fxc);
      fyc = fma(dyc, mwc, fyc);
      fzc = fma(dzc, mwc, fzc);
      fxj = rot1(fma(-dxa, mwa, fma(-dxc, mwc, fxj)));
      fyj = rot1(fma(-dya, mwa, fma(-dyc, mwc, fyj)));
      fzj = rot1(fma(-dza, mwa, fma(-dzc, mwc, fzj)));
    }
  } else
#pragma unroll 4
  for (int k = 0; k < 16; k++) {
    const double2 xy = jxy[k], zq = jzq[k], bj = jbb[k];
    // pair (atom of block I0, j)
    const double dxa = xy.x - axy.x, dya = xy.y - axy.y, dza = zq.x - azq.x;
    const double d2a = fma(dza, dza, fma(dya, dya, dxa * dxa));
    const double bba = abc.x * bj.x;
    const double eta = exp2_nonpositive(d2a * (abc.y * bj.y));
    const double fa = rsqrt_pos(fma(bba, eta, d2a));
    double qqa = azq.y * zq.y;
    if (kCut) qqa = d2a < P.gb_cut2 ? qqa : 0.0;
    const double s1a = qqa * fa;
    const double s3a = s1a * (fa * fa);
    const double mwa = fma(-0.25, eta, 1.0) * s3a;
    const double yta = fma(0.25, d2a, bba) * (eta * s3a);
    // pair (atom of block I0 + 1, j)
    const double dxc = xy.x - cxy.x, dyc = xy.y - cxy.y, dzc = zq.x - czq.x;
    const double d2c = fma(dzc, dzc, fma(dyc, dyc, dxc * dxc));
    const double bbc = cbc.x * bj.x;
    const double etc = exp2_nonpositive(d2c * (cbc.y * bj.y));
    const double fc = rsqrt_pos(fma(bbc, etc, d2c));
    double qqc = czq.y * zq.y;
    if (kCut) qqc = d2c < P.gb_cut2 ? qqc : 0.0;
    const double s1c = qqc * fc;
    const double s3c = s1c * (fc * fc);
    const double mwc = fma(-0.25, etc, 1.0) * s3c;
    const double ytc = fma(0.25, d2c, bbc) * (etc * s3c);
    e += s1a + s1c;
    fxa = fma(dxa, mwa, fxa);
    fya = fma(dya, mwa, fya);
    fza = fma(dza, mwa, fza);
    ya += yta;
    fxc = fma(dxc, mwc, fxc);
    fyc = fma(dyc, mwc, fyc);
    fzc = fma(dzc, mwc, fzc);
    yc += ytc;
    fxj = rot1(fma(-dxa, mwa, fma(-dxc, mwc, fxj)));
    fyj = rot1(fma(-dya, mwa, fma(-dyc, mwc, fyj)));
    fzj = rot1(fma(-dza, mwa, fma(-dzc, mwc, fzj)));
    yj = rot1(yj + (yta + ytc));
  }
  const double kf = -2.0 * kDielFactor;
  __builtin_amdgcn_s_setprio(3);
  __syncthreads();  // every wave is done with the records
  PAIR_STAMP(1, 2);
  StripSums& S = *reinterpret_cast<StripSums*>(s_area);
  const int jslot = (lane + start + 16) & 63;  // whose sums the lane holds after the rotations
  S.red[wave][0][lane] = kf * fxa;
  S.red[wave][1][lane] = kf * fya;
  S.red[wave][2][lane] = kf * fza;
  S.red[wave][3][lane] = ya;
  S.red[wave][4][lane] = kf * fxc;
  S.red[wave][5][lane] = kf * fyc;
  S.red[wave][6][lane] = kf * fzc;
  S.red[wave][7][lane] = yc;
  S.red[wave][8][jslot] = kf * fxj;
  S.red[wave][9][jslot] = kf * fyj;
  S.red[wave][10][jslot] = kf * fzj;
  S.red[wave][11][jslot] = yj;
  e = wave_sum(e);
  if (lane == 0) s_e[wave] = e;
  __syncthreads();
  // thread (wave q, lane l) adds quantity q of atom l of the three blocks: rows gb_fx, gb_fy, gb_fz by atom, Y by slot
  double* __restrict__ row = gb_rows + (size_t)wave * n;
  auto fold = [&](int r) { return (S.red[0][r][lane] + S.red[1][r][lane]) + (S.red[2][r][lane] + S.red[3][r][lane]); };
  const int ia = 64 * I0 + lane, ic = ia + 64, j = 64 * J + lane;
  const bool det = P.det != 0;  // deterministic mode: a tile's totals are rounded to the sums' quantum (device_math.h)
  const double qs = wave == 3 ? kQSum : kQGrad;
  if (kFar && far && wave == 3) {
    // (a far strip has no Y sums to add)
  } else if (P.rows_on && wave == 3) {  // (the row form never runs in the deterministic mode)
    if (ia < n) hbm_add(&P.bw[ia], beta_a * fold(3));
    if (ic < n) hbm_add(&P.bw[ic], beta_c * fold(7));
    if (j < n) hbm_add(&P.bw[j], beta_j * fold(11));
  } else {
    if (ia < n) hbm_add(wave == 3 ? &P.ys[ysa] : &row[ia], quantize(fold(wave), qs, det));
    if (ic < n) hbm_add(wave == 3 ? &P.ys[ysc] : &row[ic], quantize(fold(4 + wave), qs, det));
    if (j < n) hbm_add(wave == 3 ? &P.ys[ysj] : &row[j], quantize(fold(8 + wave), qs, det));
  }
  if (threadIdx.x == 0) egb_out[0] = 2.0 * kDielFactor * ((s_e[0] + s_e[1]) + (s_e[2] + s_e[3]));
  PAIR_STAMP(1, 3);
}

// ---- GB strips in packed single precision (fast mode + AGBNP_HIP_MODE_SINGLE) -----------------------------------------
// The reference's GPU platform computes its pair terms in single precision; this is the strip of above with the two
// i atoms of a lane as the two halves of a packed FP32 operand (v_pk_fma_f32 & co: two pairs per instruction), the
// hardware's exp2 / rsqrt, positions taken relative to the first atom of block J before they are rounded to FP32, the
// j sums travelling by ONE DPP move each.  Born radii and everything outside the pair loop stay FP64; a strip's
// totals leave in FP64 exactly like the FP64 strip's.  ~50 instructions per two pairs against 130.
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float rot1f(float v) {  // lane l <- lane l+1 (mod 64)
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x134, 0xf, 0xf, true));
}
template <bool kCut>
__device__ __forceinline__ void gb_strip_f32(int n, int I0, int J, const double4* __restrict__ aposq, const double* __restrict__ born_part,
                                             const double* __restrict__ inv_rvdw, double* __restrict__ gb_rows, double* __restrict__ egb_out,
                                             const PairArgs& P, char* s_area, double* s_e) {
  float4* const s_jr = reinterpret_cast<float4*>(s_area);         // block J twice over: {x, y, z, q}
  float2* const s_jb = reinterpret_cast<float2*>(s_jr + 128);     // ... {B, 1/B}
  float4* const s_ir = reinterpret_cast<float4*>(s_jb + 128);     // blocks I0, I0 + 1: [block][64] {x, y, z, q}
  float2* const s_ib = reinterpret_cast<float2*>(s_ir + 128);     // ... {B, -log2(e)/(4 B)}
  static_assert(sizeof(StripSums) >= 128 * 16 + 128 * 8 + 128 * 16 + 128 * 8, "single-precision records fit the area of the sums");
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (kCut) {  // both tiles of the strip beyond the cutoff
    double gmin = 1e300;
    for (int b = 0; b < 2; b++) {
      double gap2 = 0.0;
      for (int d = 0; d < 3; d++) {
        const double g = fmax(0.0, fmax(P.abox[6 * J + d] - P.abox[6 * (I0 + b) + 3 + d], P.abox[6 * (I0 + b) + d] - P.abox[6 * J + 3 + d]));
        gap2 += g * g;
      }
      gmin = fmin(gmin, gap2);
    }
    if (gmin >= P.gb_cut2) {
      if (threadIdx.x == 0) egb_out[0] = 0.0;
      return;
    }
  }
  const int ysa = P.a2s[min(64 * I0 + lane, n - 1)], ysc = P.a2s[min(64 * I0 + 64 + lane, n - 1)], ysj = P.a2s[min(64 * J + lane, n - 1)];
  const double4 origin = aposq[min(64 * J, n - 1)];  // (uniform address: a scalar load)
  if (wave < 3) {  // wave 0 prepares block J, waves 1 and 2 the two i blocks
    const int a = 64 * (wave == 0 ? J : I0 + wave - 1) + lane;
    const bool va = a < n;
    const int ac = va ? a : n - 1;
    const double4 pa = aposq[ac];
    const BornRadius bra = born_radius(inv_rvdw[ac], born_part[ac]);
    const float4 rec = make_float4((float)(pa.x - origin.x), (float)(pa.y - origin.y), (float)(pa.z - origin.z), va ? (float)pa.w : 0.0f);
    if (wave == 0) {
      s_jr[lane] = s_jr[lane + 64] = rec;
      s_jb[lane] = s_jb[lane + 64] = make_float2((float)bra.br, (float)bra.inv_br);
    } else {
      s_ir[(wave - 1) * 64 + lane] = rec;
      s_ib[(wave - 1) * 64 + lane] = make_float2((float)bra.br, (float)((-0.25 * 1.4426950408889634074) * bra.inv_br));
    }
  }
  __syncthreads();
  const int start = 16 * wave;  // the four waves take a quarter of the cyclic distances each
  const float4 ra = s_ir[lane], rc = s_ir[64 + lane];
  const float2 ba = s_ib[lane], bc = s_ib[64 + lane];
  const v2f xi = {ra.x, rc.x}, yi = {ra.y, rc.y}, zi = {ra.z, rc.z}, qi = {ra.w, rc.w}, bi = {ba.x, bc.x}, ci = {ba.y, bc.y};
  const int base = (lane + start) & 63;
  const float4* __restrict__ jr = s_jr + base;
  const float2* __restrict__ jb = s_jb + base;
  const float cut2 = (float)P.gb_cut2;
  v2f fx = {0.f, 0.f}, fy = {0.f, 0.f}, fz = {0.f, 0.f}, ys = {0.f, 0.f}, e2 = {0.f, 0.f};
  float fxj = 0.f, fyj = 0.f, fzj = 0.f, yj = 0.f;
#pragma unroll 4
  for (int k = 0; k < 16; k++) {
    const float4 rj = jr[k];
    const float2 bj = jb[k];
    const v2f dx = rj.x - xi, dy = rj.y - yi, dz = rj.z - zi;
    const v2f d2 = dz * dz + (dy * dy + dx * dx);
    const v2f bb = bi * bj.x;
    const v2f arg = d2 * (ci * bj.y);
    const v2f et = {__builtin_amdgcn_exp2f(arg.x), __builtin_amdgcn_exp2f(arg.y)};  // exp(-d^2 / (4 B_i B_j))
    const v2f den = bb * et + d2;
    const v2f f = {__builtin_amdgcn_rsqf(den.x), __builtin_amdgcn_rsqf(den.y)};
    v2f qq = qi * rj.w;
    if (kCut) qq = {d2.x < cut2 ? qq.x : 0.f, d2.y < cut2 ? qq.y : 0.f};
    const v2f s1 = qq * f;
    const v2f s3 = s1 * (f * f);
    const v2f mw = (1.0f - 0.25f * et) * s3;
    const v2f yt = (0.25f * d2 + bb) * (et * s3);
    e2 += s1;
    fx += dx * mw;
    fy += dy * mw;
    fz += dz * mw;
    ys += yt;
    const v2f gx = dx * mw, gy = dy * mw, gz = dz * mw;
    fxj = rot1f(fxj - (gx.x + gx.y));
    fyj = rot1f(fyj - (gy.x + gy.y));
    fzj = rot1f(fzj - (gz.x + gz.y));
    yj = rot1f(yj + (yt.x + yt.y));
  }
  const double kf = -2.0 * kDielFactor;
  __syncthreads();  // every wave is done with the records
  StripSums& S = *reinterpret_cast<StripSums*>(s_area);
  const int jslot = (lane + start + 16) & 63;  // whose sums the lane holds after the rotations
  S.red[wave][0][lane] = kf * (double)fx.x;
  S.red[wave][1][lane] = kf * (double)fy.x;
  S.red[wave][2][lane] = kf * (double)fz.x;
  S.red[wave][3][lane] = (double)ys.x;
  S.red[wave][4][lane] = kf * (double)fx.y;
  S.red[wave][5][lane] = kf * (double)fy.y;
  S.red[wave][6][lane] = kf * (double)fz.y;
  S.red[wave][7][lane] = (double)ys.y;
  S.red[wave][8][jslot] = kf * (double)fxj;
  S.red[wave][9][jslot] = kf * (double)fyj;
  S.red[wave][10][jslot] = kf * (double)fzj;
  S.red[wave][11][jslot] = (double)yj;
  const double e = wave_sum((double)e2.x + (double)e2.y);
  if (lane == 0) s_e[wave] = e;
  __syncthreads();
  double* __restrict__ row = gb_rows + (size_t)wave * n;
  auto fold = [&](int r) { return (S.red[0][r][lane] + S.red[1][r][lane]) + (S.red[2][r][lane] + S.red[3][r][lane]); };
  const int ia = 64 * I0 + lane, ic = ia + 64, j = 64 * J + lane;
  const bool det = P.det != 0;
  const double qs = wave == 3 ? kQSum : kQGrad;
  if (ia < n) hbm_add(wave == 3 ? &P.ys[ysa] : &row[ia], quantize(fold(wave), qs, det));
  if (ic < n) hbm_add(wave == 3 ? &P.ys[ysc] : &row[ic], quantize(fold(4 + wave), qs, det));
  if (j < n) hbm_add(wave == 3 ? &P.ys[ysj] : &row[j], quantize(fold(8 + wave), qs, det));
  if (threadIdx.x == 0) egb_out[0] = 2.0 * kDielFactor * ((s_e[0] + s_e[1]) + (s_e[2] + s_e[3]));
}

// kMasks (round 6): the instantiations of the five-launch mode where the Born stage is not the FP64 row launch that carries the
// masks' renewal (tile kernels: deterministic mode, AGBNP_HIP_ROWS=0): the workgroups behind the last work item are the tiles that
// lay the level-2 neighbour masks down anew when the cavity launch's trailing workgroups asked for it (k_rows, MASKS)
template <bool kCut, bool kSingle, bool kFar, bool kMasks = false>
__global__ __launch_bounds__(256) void k_gb_tiles(int n, const int* __restrict__ items, const double4* __restrict__ aposq,
                                                  const double* __restrict__ born_part, const double* __restrict__ inv_rvdw,
                                                  const double* __restrict__ alpha, double* __restrict__ born,
                                                  double* __restrict__ born_fp, double* __restrict__ brw,
                                                  double* __restrict__ e_atom, double* __restrict__ gb_rows,
                                                  double* __restrict__ egb_part, PairArgs P) {
  if (kMasks && (int)blockIdx.x > P.gb_items_count) {
    if (((P.estatus[kStatOrderStale] & 2) | P.estatus[kStatMaskAging]) == 0) return;
    return neighbor_tile(P, (int)blockIdx.x - 1 - P.gb_items_count, true);
  }
  // one LDS area, two lives: the atom records during the walk, the sums of the four waves after it
  __shared__ __align__(16) char s_area[sizeof(StripSums)];
  static_assert(sizeof(StripSums) >= sizeof(TileSums) && sizeof(TileSums) >= kRoleScratchBytes, "the packing workgroup borrows the tile area");
  static_assert(sizeof(StripSums) >= sizeof(double2) * (3 * 128 + 6 * 64), "strip records fit the area of the sums");
  // workgroup 0 does the bookkeeping of the next evaluation (it needs the tree's shapes only): mostly serial work that
  // hides underneath this launch, the longest of the pair stages
  __shared__ double s_e[4];
  __shared__ int s_bmax[4];  // (kFar: the blocks' largest Born radii)
  if (blockIdx.x == 0) {
    PAIR_STAMP(1, 0);
    rebase_for_parity(P, 0);
    packing_role<true>(P, s_area, (int)sizeof(StripSums));
    PAIR_STAMP(1, 3);
    return;
  }
  static_assert(sizeof(TileSums) >= sizeof(double2) * (3 * 128 + 3 * 64), "records fit the area of the sums");
  double2* const s_xy = reinterpret_cast<double2*>(s_area);  // block J twice over: entry m and m + 64 are atom 64 J + m
  double2* const s_zq = s_xy + 128;
  double2* const s_bb = s_zq + 128;
  double2* const s_ixy = s_bb + 128;                          // block I: {x, y}, {z, q}, {B, -log2(e)/(4 B)}
  double2* const s_izq = s_ixy + 64;
  double2* const s_ibc = s_izq + 64;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // A workgroup's prologue and epilogue are a few instructions between memory round trips, its walk a few thousand
  // instructions without one: the former run at high priority, so that a workgroup that arrives beside three that are
  // walking gets its loads out at once instead of when the SIMD has nothing else to do (when it is too late to hide them).
  __builtin_amdgcn_s_setprio(3);
  const int item = items[blockIdx.x - 1];
  const int I = item & 0xfff, J = (item >> 12) & 0xfff;
  PAIR_STAMP(1, 0);
  if (item & kGbStripFlag) {
    if (kSingle) return gb_strip_f32<kCut>(n, I, J, aposq, born_part, inv_rvdw, gb_rows, egb_part + (blockIdx.x - 1), P, s_area, s_e);
    return gb_strip<kCut, kFar>(n, I, J, aposq, born_part, inv_rvdw, gb_rows, egb_part + (blockIdx.x - 1), P, s_area, s_e, s_bmax);
  }
  const bool diag = I == J;
  if (kCut && !diag) {  // fast mode: a tile whose two blocks are further apart than the cutoff has no pair to meet
    double gap2 = 0.0;
    for (int d = 0; d < 3; d++) {
      const double g = fmax(0.0, fmax(P.abox[6 * J + d] - P.abox[6 * I + 3 + d], P.abox[6 * I + d] - P.abox[6 * J + 3 + d]));
      gap2 += g * g;
    }
    if (gap2 >= P.gb_cut2) {
      if (threadIdx.x == 0) egb_part[blockIdx.x - 1] = 0.0;
      return;
    }
  }
  PAIR_STAMP_WHERE(1, item);
  const int ysi = P.a2s[min(64 * I + lane, n - 1)], ysj = P.a2s[min(64 * J + lane, n - 1)];  // (the Y sums leave by pair-order slot, see gb_strip)
  double beta_i = 0.0, beta_j = 0.0;  // row form of the chain rule: see gb_strip
  if (P.rows_on && wave == 3) {
    const int ki = min(64 * I + lane, n - 1), kj = min(64 * J + lane, n - 1);
    const double ri = inv_rvdw[ki], rj = inv_rvdw[kj], pi = born_part[ki], pj = born_part[kj];
    beta_i = bw_beta(born_radius(ri, pi));
    beta_j = bw_beta(born_radius(rj, pj));
  }
  // Born radii from the finished descreening sums (every tile recomputes them for its 128 atoms: a few dozen
  // flops per atom against 4096 pair evaluations, and one kernel launch less per evaluation):
  // wave 0 prepares block J, wave 1 block I
  if (wave < 2) {
    const int a = 64 * (wave == 0 ? J : I) + lane;
    const bool va = a < n;
    const int ac = va ? a : n - 1;
    const double4 pa = aposq[ac];
    const BornRadius bra = born_radius(inv_rvdw[ac], born_part[ac]);
    const double qa = va ? pa.w : 0.0;  // zero charge switches a padded atom off
    if (wave == 0) {
      s_xy[lane] = s_xy[lane + 64] = make_double2(pa.x, pa.y);
      s_zq[lane] = s_zq[lane + 64] = make_double2(pa.z, qa);
      s_bb[lane] = s_bb[lane + 64] = make_double2(bra.br, bra.inv_br);
    } else {
      s_ixy[lane] = make_double2(pa.x, pa.y);
      s_izq[lane] = make_double2(pa.z, qa);
      s_ibc[lane] = make_double2(bra.br, (-0.25 * 1.4426950408889634074) * bra.inv_br);
      if (diag && va) {
        // the diagonal tile of a block publishes the per-atom results exactly once:
        // B_i, f'_i, vdW energy + GB self energy, brw_i (ReferenceAGBNPKernels.cpp:477,513-533)
        const double bh = bra.br + kHBRadius, bh3 = bh * bh * bh, al = alpha[a];
        const double brw_a = -(1. / (4. * kPi)) * 3. * al * bra.br * bra.br * bra.fp / (bh3 * bh);
        born[a] = bra.br;
        born_fp[a] = bra.fp;
        e_atom[a] = al / bh3 + kDielFactor * pa.w * pa.w * bra.inv_br;
        brw[a] = brw_a;
        P.srec[ysi] = make_double4(bra.br, bra.fp, brw_a, pa.w);  // the chain-rule stage's copy, by slot (a = 64 I + lane here)
        if (P.rows_on) hbm_add(&P.bw[a], bw_alpha(bra, brw_a, pa.w));
      }
    }
  }
  __syncthreads();
  PAIR_STAMP(1, 1);
  __builtin_amdgcn_s_setprio(0);
  // the four waves take a quarter of the cyclic distances each (diagonal tile: distances 1..32, 8 per wave)
  const int nsteps = diag ? 8 : 16;
  const int start = (diag ? 1 : 0) + nsteps * wave;  // cyclic offset of the first j met by lane l
  const double2 ixy = s_ixy[lane], izq = s_izq[lane], ibc = s_ibc[lane];
  const double xi = ixy.x, yi_ = ixy.y, zi = izq.x, qi = izq.y, bi = ibc.x, ci = ibc.y;
  // diagonal tile, cyclic distance 32 (the last step of the last wave): the pair (l, l+32) would otherwise be
  // met from both ends
  const int masked_step = diag ? 32 - start : -1;
  const double qlast = lane >= 32 ? 0.0 : qi;
  const int base = (lane + start) & 63;
  const double2* __restrict__ jxy = s_xy + base;
  const double2* __restrict__ jzq = s_zq + base;
  const double2* __restrict__ jbb = s_bb + base;
  // sums in units that leave the constant factors to the epilogue:
  //   e = sum qq f,  F_i = -2k sum D qq (1 - et/4) f^3,  Y = sum qq (B_i B_j + d^2/4) et f^3   (qq = q_i q_j)
  double fxi = 0, fyi = 0, fzi = 0, yi = 0, fxj = 0, fyj = 0, fzj = 0, yj_acc = 0, e = 0;
#pragma unroll 8
  for (int k = 0; k < nsteps; k++) {
    const double2 xy = jxy[k], zq = jzq[k], bj = jbb[k];
    const double dx = xy.x - xi, dy = xy.y - yi_, dz = zq.x - zi;
    const double d2 = fma(dz, dz, fma(dy, dy, dx * dx));
    const double bb = bi * bj.x;
    const double et = exp2_nonpositive(d2 * (ci * bj.y));  // exp(-d^2 / (4 B_i B_j))
    const double fgb = rsqrt_pos(fma(bb, et, d2));
    double qq = (k == masked_step ? qlast : qi) * zq.y;
    if (kCut) qq = d2 < P.gb_cut2 ? qq : 0.0;  // (a pair beyond the cutoff contributes to nothing: every term carries qq)
    const double s1 = qq * fgb;
    e += s1;
    const double s3 = s1 * (fgb * fgb);
    const double mw = fma(-0.25, et, 1.0) * s3;
    fxi = fma(dx, mw, fxi);
    fyi = fma(dy, mw, fyi);
    fzi = fma(dz, mw, fzi);
    fxj = fma(-dx, mw, fxj);
    fyj = fma(-dy, mw, fyj);
    fzj = fma(-dz, mw, fzj);
    const double yt = fma(0.25, d2, bb) * (et * s3);
    yi += yt;
    yj_acc += yt;
    fxj = rot1(fxj);
    fyj = rot1(fyj);
    fzj = rot1(fzj);
    yj_acc = rot1(yj_acc);
  }
  const double kf = -2.0 * kDielFactor;
  __builtin_amdgcn_s_setprio(3);
  __syncthreads();  // every wave is done with the records
  PAIR_STAMP(1, 2);
  TileSums& s_sums = *reinterpret_cast<TileSums*>(s_area);
  {
    const double vi4[4] = {kf * fxi, kf * fyi, kf * fzi, yi}, vj4[4] = {kf * fxj, kf * fyj, kf * fzj, yj_acc};
    tile_sums_store(s_sums, wave, lane, (lane + start + nsteps) & 63, vi4, vj4);  // jslot: whose sums the lane holds now
  }
  e = wave_sum(e);
  if (lane == 0) s_e[wave] = e;
  __syncthreads();
  // thread (wave q, lane l) adds quantity q of atom l of block I and of block J: rows gb_fx, gb_fy, gb_fz by atom, Y by slot
  double* __restrict__ row = gb_rows + (size_t)wave * n;
  const int i = 64 * I + lane, j = 64 * J + lane;
  const bool det = P.det != 0;
  const double qs = wave == 3 ? kQSum : kQGrad;
  if (P.rows_on && wave == 3) {
    if (i < n) hbm_add(&P.bw[i], beta_i * tile_sums_fold(s_sums, 3, lane));
    if (j < n) hbm_add(&P.bw[j], beta_j * tile_sums_fold(s_sums, 7, lane));
  } else {
    if (i < n) hbm_add(wave == 3 ? &P.ys[ysi] : &row[i], quantize(tile_sums_fold(s_sums, wave, lane), qs, det));
    if (j < n) hbm_add(wave == 3 ? &P.ys[ysj] : &row[j], quantize(tile_sums_fold(s_sums, 4 + wave, lane), qs, det));
  }
  if (threadIdx.x == 0) egb_part[blockIdx.x - 1] = 2.0 * kDielFactor * ((s_e[0] + s_e[1]) + (s_e[2] + s_e[3]));
  PAIR_STAMP(1, 3);
}

// ---- descreening sums of the inverse Born radii, 64x64 tiles in "pair order" with range culling -------------
// Reference loop (ReferenceAGBNPKernels.cpp:435-449): sum_i over all atoms, heavy j != i, d < 2 nm:
//   born_part_i += s_j Q(d; type_i, type_j)       s_j = selfvol_j / (4 pi R_j^3 / 3)   (:420-433)
// Only heavy atoms descreen, so the atoms are walked in pair order (pslot): all heavy atoms first, then all
// hydrogens, each group padded to whole blocks of 64 (slot h of a heavy block IS heavy atom h).  Two kinds of tiles:
//   heavy x heavy  symmetric tiles (I <= J), every unordered pair met once, both directions (two look-ups)
//   heavy x H      full tiles, one direction (the heavy atom descreens the hydrogen, one look-up)
// One workgroup = one tile, four waves of a quarter of the cyclic distances each; block I in registers, the
// static record of block J from a doubled LDS copy, the running sum of the j atom travels by DPP rotation.  A
// tile whose two bounding boxes are more than the table's 2 nm reach apart exits at once.
template <bool kBoth>
__device__ __forceinline__ void born_walk(double& sum_i, double& sum_j, const double2* __restrict__ s_lut,
                                          const double2* __restrict__ jxy, const double2* __restrict__ jzs,
                                          const double* __restrict__ jty, double xi, double yi, double zi, double si, int row,
                                          int tsr, int nsteps, int masked_step, bool vi, bool lower, int ntj, double range2) {
#pragma unroll 4
  for (int k = 0; k < nsteps; k++) {
    const double2 xy = jxy[k], zs = jzs[k];
    const double ty = jty[k];
    const double dx = xy.x - xi, dy = xy.y - yi, dz = zs.x - zi;
    const double d2 = fma(dz, dz, fma(dy, dy, dx * dx));
    const int tj = __double2loint(ty);  // screened type | screener type << 16
    if (d2 < range2 && vi && __double2hiint(ty) >= 0 && (k != masked_step || lower)) {
      const double d = d2 * rsqrt_pos(d2);
      sum_j = fma(si, spline_value(s_lut, ((tj & 0xffff) * ntj + tsr) * kLutStride, d), sum_j);   // i descreens j
      if (kBoth) sum_i = fma(zs.y, spline_value(s_lut, (row + (tj >> 16)) * kLutStride, d), sum_i);  // j descreens i
    }
    sum_j = rot1(sum_j);
  }
}

__global__ __launch_bounds__(256) void k_born_tiles(int nh, int nhb, int ntj, int lut_entries, const int* __restrict__ items,
                                                   const int* __restrict__ pslot, const double* __restrict__ pbox,
                                                   const double4* __restrict__ prec, const double* __restrict__ sv_vdw,
                                                   const double* __restrict__ inv_vol_h, const double2* __restrict__ lut,
                                                   double* __restrict__ born_part, double range2, int det, int cull_first) {
  extern __shared__ double2 s_lut[];
  __shared__ double2 s_xy[128], s_zs[128];  // block J twice over: {x, y}, {z, s}
  __shared__ double s_ty[128];               // low word: types, high word: >= 0 for a real atom
  __shared__ double2 s_ixy[64], s_izs[64];   // block I, on its way from memory to the lanes' registers
  __shared__ double s_ity[64];
  __shared__ double s_red[4][2][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  PAIR_STAMP(0, 0);
  const int item = items[blockIdx.x];
  PAIR_STAMP_WAIT(0, 7, "lgkmcnt(0)");  // kernel arguments and the item are here
  const int I = item & 0xfff, J = (item >> 12) & 0xfff;
  const bool diag = I == J;
  const bool both = J < nhb;  // heavy x heavy
  // Everything the tile reads from memory is asked for here, before anything is waited for, so the workgroup's start
  // is ONE round trip deep (records by slot: no slot -> atom indirection; a slot of a heavy block is the heavy atom
  // itself, so its self volume comes straight from the tree's row).
  auto out_of_range = [&]() {  // workgroup-uniform range test on the two bounding boxes
    if (diag) return false;
    double gap2 = 0.0;
    for (int d = 0; d < 3; d++) {
      const double g = fmax(0.0, fmax(pbox[6 * J + d] - pbox[6 * I + 3 + d], pbox[6 * I + d] - pbox[6 * J + 3 + d]));
      gap2 += g * g;
    }
    return gap2 >= range2;
  };
  // (a large system culls most of its tiles: there the test comes first and a culled tile costs two scalar loads; a
  // small one culls next to none: there the test waits until the tile's loads are on their way)
  if (cull_first && out_of_range()) return;
  // Every slot of the tile is fetched ONCE (the start of a launch is a burst of every workgroup's loads at the same
  // time: it is their volume that the first microseconds wait for): lanes 0..15 of wave w take slots 16w.. of block I,
  // lanes 16..31 those of block J (the upper half of the wave repeats the same addresses), and block I reaches the
  // lanes' registers through LDS.
  const int islot = 64 * I + lane, jslot = 64 * J + lane;
  const int fpart = (lane >> 4) & 1, fidx = 16 * wave + (lane & 15);
  const int fslot = 64 * (fpart ? J : I) + fidx, fh = min(fslot, nh - 1);  // (clamped: the loads are unconditional, the choice comes after)
  const double4 fr = prec[fslot];
  const double fsv = sv_vdw[fh], fiv = inv_vol_h[fh];
  const int a_out = pslot[wave == 0 ? islot : jslot];  // wave 0 adds the sums of block I, wave 1 those of block J: by atom
  const LutBatch lut0 = lut_fetch(lut, lut_entries, 0);
  PAIR_STAMP_WAIT(0, 8, "vmcnt(0)");  // everything has arrived
  const double fs = fslot < nh ? fsv * fiv : 0.0;
  lut_store(s_lut, lut0, lut_entries, 0);
  lut_copy_rest(s_lut, lut, lut_entries);
  PAIR_STAMP_WAIT(0, 9, "vmcnt(0) lgkmcnt(0)");  // the tables are in LDS (this wave's part)
  if (!cull_first && out_of_range()) return;
  PAIR_STAMP_WHERE(0, item);
  if (lane < 32) {
    if (fpart) {
      s_xy[fidx] = s_xy[fidx + 64] = make_double2(fr.x, fr.y);
      s_zs[fidx] = s_zs[fidx + 64] = make_double2(fr.z, fs);
      s_ty[fidx] = s_ty[fidx + 64] = fr.w;
    } else {
      s_ixy[fidx] = make_double2(fr.x, fr.y);
      s_izs[fidx] = make_double2(fr.z, fs);
      s_ity[fidx] = fr.w;
    }
  }
  __syncthreads();
  PAIR_STAMP(0, 1);
  const double2 ixy = s_ixy[lane], izs = s_izs[lane];
  const double ity = s_ity[lane];
  const int nsteps = diag ? 8 : 16;
  const int start = (diag ? 1 : 0) + nsteps * wave;  // cyclic offset of the first j met by lane l
  const bool vi = __double2hiint(ity) >= 0;
  const double xi = ixy.x, yi = ixy.y, zi = izs.x, si = izs.y;
  const int2 mi = make_int2(__double2loint(ity) & 0xffff, __double2loint(ity) >> 16);  // {screened type, screener type}: block I is always a heavy block
  const int base = (lane + start) & 63;
  double sum_i = 0.0, sum_j = 0.0;
  // diagonal tile, cyclic distance 32 (the last step of the last wave): one end only
  if (both)
    born_walk<true>(sum_i, sum_j, s_lut, s_xy + base, s_zs + base, s_ty + base, xi, yi, zi, si, mi.x * ntj, mi.y, nsteps,
                    diag ? 32 - start : -1, vi, lane < 32, ntj, range2);
  else
    born_walk<false>(sum_i, sum_j, s_lut, s_xy + base, s_zs + base, s_ty + base, xi, yi, zi, si, 0, mi.y, nsteps, -1, vi, true, ntj, range2);
  s_red[wave][0][lane] = sum_i;
  s_red[wave][1][(lane + start + nsteps) & 63] = sum_j;  // whose sum the lane holds after the rotations
  __syncthreads();
  PAIR_STAMP(0, 2);
  if (wave < 2) {  // wave 0 adds the sums of block I, wave 1 those of block J
    if (wave == 0 && !both) return;
    const int a = a_out;
    if (a >= 0)
      hbm_add(&born_part[a], quantize((s_red[0][wave][lane] + s_red[1][wave][lane]) + (s_red[2][wave][lane] + s_red[3][wave][lane]), kQBorn, det != 0));
    PAIR_STAMP(0, 3);
  }
}

// ---- Born-radius chain rule, 64x64 tiles in "pair order" with range culling ------------------------------
// Reference loop (ReferenceAGBNPKernels.cpp:555-586) over ordered (i, heavy j != i, d < 2 nm):
//   W_j += brw_i Q,  U_j += bru_i Q,  F_i += D (brw_i + bru_i) s_j Q'/d,  F_j -= same      (D = r_j - r_i)
// Only heavy atoms descreen, so the atoms are walked in pair order (pslot): all heavy atoms first, then all
// hydrogens, each group padded to whole blocks of 64.  That leaves two kinds of tiles and no per-lane type tests:
//   heavy x heavy  symmetric tiles (I <= J), every unordered pair met once, both directions (two look-ups)
//   heavy x H      full tiles, one direction (the heavy atom descreens the hydrogen, one look-up)
//   H x H          nothing to do, never scheduled
// which halves the table look-ups (the LDS pipe is what bounds this kernel).  Machinery as in k_gb_tiles:
// block I in registers, the static record of block J from a doubled LDS copy, the sums of the j atom travel by
// DPP rotation.  A work item whose two bounding boxes are more than the table's 2 nm reach apart exits at once.
struct DbornLane {
  double x, y, z, bw, s;   // the lane's own atom i
  int row, tsr;            // screened type * ntj, screener type
  double fxi, fyi, fzi, wui, fxj, fyj, fzj, wuj;
};

template <bool kBoth>
__device__ __forceinline__ void dborn_walk(DbornLane& L, const double2* __restrict__ s_lut, const double2* __restrict__ jxy,
                                           const double2* __restrict__ jzw, const double2* __restrict__ jsm, int nsteps,
                                           int masked_step, bool vi, bool lower, int ntj, double range2) {
#pragma unroll 4
  for (int k = 0; k < nsteps; k++) {
    const double2 xy = jxy[k], zw = jzw[k], sm = jsm[k];
    const double dx = xy.x - L.x, dy = xy.y - L.y, dz = zw.x - L.z;
    const double d2 = fma(dz, dz, fma(dy, dy, dx * dx));
    const int tj = __double2loint(sm.y);  // screened type | screener type << 16
    if (d2 < range2 && vi && __double2hiint(sm.y) >= 0 && (k != masked_step || lower)) {
      const double rinv = rsqrt_pos(d2);
      const double d = d2 * rinv;
      double q2, dq2;  // i descreens j
      spline_value_deriv(s_lut, ((tj & 0xffff) * ntj + L.tsr) * kLutStride, d, q2, dq2);
      L.wui = fma(zw.y, q2, L.wui);
      double t = zw.y * L.s * dq2;
      if (kBoth) {  // j descreens i
        double q1, dq1;
        spline_value_deriv(s_lut, (L.row + (tj >> 16)) * kLutStride, d, q1, dq1);
        L.wuj = fma(L.bw, q1, L.wuj);
        t = fma(L.bw * sm.x, dq1, t);
      }
      t *= rinv;
      L.fxi = fma(dx, t, L.fxi);
      L.fyi = fma(dy, t, L.fyi);
      L.fzi = fma(dz, t, L.fzi);
      L.fxj = fma(-dx, t, L.fxj);
      L.fyj = fma(-dy, t, L.fyj);
      L.fzj = fma(-dz, t, L.fzj);
    }
    L.fxj = rot1(L.fxj);
    L.fyj = rot1(L.fyj);
    L.fzj = rot1(L.fzj);
    if (kBoth) L.wuj = rot1(L.wuj);
  }
}

__global__ __launch_bounds__(256) void k_dborn_tiles(int n, int nhb, int ntj, int lut_entries, const int* __restrict__ items,
                                                    const int* __restrict__ pslot, const double* __restrict__ pbox,
                                                    const double4* __restrict__ prec, const double4* __restrict__ srec,
                                                    const double* __restrict__ ys, const double* __restrict__ sv_vdw,
                                                    const double* __restrict__ inv_vol_h, int nh, const double2* __restrict__ lut,
                                                    double* __restrict__ db_rows, PairArgs P, double* __restrict__ energy_out,
                                                    double* __restrict__ components, int role_bytes) {
  // the first workgroup carries the energy sum (see above)
  extern __shared__ double2 s_lut[];
  if (blockIdx.x == 0) return energy_role(P, 1, energy_out, components, reinterpret_cast<char*>(s_lut));
  if (blockIdx.x == 1) return dealing_role(P, reinterpret_cast<char*>(s_lut), role_bytes);  // second half of the bookkeeping
  // block J twice over (entry m and m + 64 are slot 64 J + m): {x, y}, {z, bw}, {s, types|validity}
  __shared__ double2 s_rec[3][128];
  // one workgroup = one tile; its four waves take a quarter of the cyclic distances each and share the j records
  // and the spline tables
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  PAIR_STAMP(2, 0);
  const int item = items[blockIdx.x - 2];
  const int I = item & 0xfff, J = (item >> 12) & 0xfff;
  const bool diag = I == J;
  const bool both = J < nhb;  // heavy x heavy
  // Everything the tile reads from memory is asked for here, before anything is waited for (see k_born_tiles): the
  // slot's geometry and types (k_prep), {B, f', brw, q} and the finished Y sum (GB stage), the self volume (tree).
  struct SlotData {
    double4 r, g;
    double y, sv, iv;
  };
  auto fetch = [&](int slot) {  // (unconditional loads, clamped: nothing is waited for in here)
    SlotData d;
    const int h = min(slot, nh - 1);
    d.r = prec[slot];
    d.g = srec[slot];
    d.y = ys[slot];
    d.sv = sv_vdw[h];
    d.iv = inv_vol_h[h];
    return d;
  };
  // {bw, s} of an atom: bw = brw + bru with bru = -(1/4pi) k (q^2 + Y B) f' (ReferenceAGBNPKernels.cpp:534-542),
  // formed here from the finished GB sums instead of a per-atom kernel in between
  auto weights = [&](const SlotData& d, bool slot_is_heavy) {
    const double bru = -(1. / (4. * kPi)) * kDielFactor * (d.g.w * d.g.w + d.y * d.g.x) * d.g.y;
    return make_double2(d.g.z + bru, slot_is_heavy ? d.sv * d.iv : 0.0);  // slot h of a heavy block is heavy atom h
  };
  auto out_of_range = [&]() {  // workgroup-uniform range test on the two bounding boxes
    if (diag) return false;
    double gap2 = 0.0;
    for (int d = 0; d < 3; d++) {
      const double g = fmax(0.0, fmax(pbox[6 * J + d] - pbox[6 * I + 3 + d], pbox[6 * I + d] - pbox[6 * J + 3 + d]));
      gap2 += g * g;
    }
    return gap2 >= P.range2;
  };
  if (P.cull_first && out_of_range()) return;  // (see k_born_tiles)
  // every slot of the tile is fetched once, block I reaches the lanes' registers through LDS (see k_born_tiles)
  const int islot = 64 * I + lane, jslot = 64 * J + lane;
  const int fpart = (lane >> 4) & 1, fidx = 16 * wave + (lane & 15);
  const int fslot = 64 * (fpart ? J : I) + fidx;
  const SlotData fd = fetch(fslot);
  const int ai = pslot[islot], aj = pslot[jslot];  // the force rows are by atom
  const LutBatch lut0 = lut_fetch(lut, lut_entries, 0);
  lut_store(s_lut, lut0, lut_entries, 0);
  lut_copy_rest(s_lut, lut, lut_entries);
  if (!P.cull_first && out_of_range()) return;
  PAIR_STAMP_WHERE(2, item);
  double2* const s_irec = s_lut + lut_entries;  // [3][64] block I, behind the tables (the launch sizes the area for both)
  if (lane < 32) {
    const bool vf = __double2hiint(fd.r.w) >= 0;
    const double2 wf = vf ? weights(fd, fslot < nh) : make_double2(0.0, 0.0);
    // .w of the record: low word screened type | screener type << 16 (only read in heavy x heavy tiles); high word >= 0 for a real atom
    const double2 r0 = make_double2(fd.r.x, fd.r.y), r1 = make_double2(fd.r.z, wf.x), r2 = make_double2(wf.y, fd.r.w);
    if (fpart) {
      s_rec[0][fidx] = s_rec[0][fidx + 64] = r0;
      s_rec[1][fidx] = s_rec[1][fidx + 64] = r1;
      s_rec[2][fidx] = s_rec[2][fidx + 64] = r2;
    } else {
      s_irec[fidx] = r0;
      s_irec[64 + fidx] = r1;
      s_irec[128 + fidx] = r2;
    }
  }
  const int nsteps = diag ? 8 : 16;
  const int start = (diag ? 1 : 0) + nsteps * wave;  // cyclic offset of the first j met by lane l
  __syncthreads();
  PAIR_STAMP(2, 1);
  const double2 i0 = s_irec[lane], i1 = s_irec[64 + lane], i2 = s_irec[128 + lane];
  const bool vi = __double2hiint(i2.y) >= 0;
  const int2 mi = make_int2(__double2loint(i2.y) & 0xffff, __double2loint(i2.y) >> 16);  // {screened type, screener type}: block I is always a heavy block
  DbornLane L;
  L.x = i0.x, L.y = i0.y, L.z = i1.x, L.bw = i1.y, L.s = i2.x;
  L.row = mi.x * ntj, L.tsr = mi.y;
  L.fxi = L.fyi = L.fzi = L.wui = L.fxj = L.fyj = L.fzj = L.wuj = 0.0;
  const int base = (lane + start) & 63;
  const double2* __restrict__ jxy = s_rec[0] + base;
  const double2* __restrict__ jzw = s_rec[1] + base;
  const double2* __restrict__ jsm = s_rec[2] + base;
  // diagonal tile, cyclic distance 32 (the last step of the last wave): one end only
  if (both)
    dborn_walk<true>(L, s_lut, jxy, jzw, jsm, nsteps, diag ? 32 - start : -1, vi, lane < 32, ntj, P.range2);
  else
    dborn_walk<false>(L, s_lut, jxy, jzw, jsm, nsteps, -1, vi, true, ntj, P.range2);
  __syncthreads();  // every wave is done with the spline tables: their LDS now carries the sums of the four waves
  PAIR_STAMP(2, 2);
  TileSums& s_sums = *reinterpret_cast<TileSums*>(s_lut);
  {
    const double vi4[4] = {L.fxi, L.fyi, L.fzi, L.wui}, vj4[4] = {L.fxj, L.fyj, L.fzj, L.wuj};
    tile_sums_store(s_sums, wave, lane, (lane + start + nsteps) & 63, vi4, vj4);  // jslot: whose sums the lane holds now
  }
  __syncthreads();
  // thread (wave q, lane l) adds quantity q of slot l of block I and of block J: rows db_fx, db_fy, db_fz by atom,
  // row db_wu by heavy index (= the slot of a heavy block: only heavy atoms collect W+U, and the tree reads it so)
  double* __restrict__ row = db_rows + (size_t)wave * n;
  const bool det = P.det != 0;
  const double qs = wave == 3 ? kQSum : kQGrad;
  if (vi) hbm_add(&row[wave == 3 ? 64 * I + lane : ai], quantize(tile_sums_fold(s_sums, wave, lane), qs, det));
  if (aj >= 0 && (both || wave < 3)) hbm_add(&row[wave == 3 ? 64 * J + lane : aj], quantize(tile_sums_fold(s_sums, 4 + wave, lane), qs, det));
  PAIR_STAMP(2, 3);
}

// ---- row form of the pair stages: the launches (device code in row_kernels.h) -----------------------------------------
// (launch bounds: six waves per SIMD = three workgroups per CU, 80 vector registers; the GB rows, whose pair terms and
// bookkeeping role need more, four)
// SINGLE: the Born / chain-rule rows with their pair terms in single precision (row_kernels.h; fast mode + AGBNP_HIP_MODE_SINGLE)
// MASKS (five-launch mode, Born rows only; `role_bytes` then carries the number of the first such workgroup): behind the row
// and list-building workgroups the grid holds one workgroup per tile of the level-2 neighbour masks.  They leave on two scalar
// loads unless the cavity launch of this evaluation found a heavy atom a quarter of the masks' skin (or more: then the
// evaluation is void) from where it was when the masks were laid down; then they lay them down anew from this evaluation's
// positions, for the next one: the masks heal on the device, whatever is queued behind.
// DEVPAR: the evaluation's set of accumulators is named by the device's own count (PairArgs::five == 2, contexts that have
// been captured into a graph); a launch of its own instantiation, so that eager launches carry no trace of it.
template <int KIND, bool SINGLE = false, bool MASKS = false, bool DEVPAR = false>
__global__ __launch_bounds__(64 * row_waves(KIND), KIND == kGbRows ? 4 : 6) void k_rows(PairArgs P, double* __restrict__ energy_out, double* __restrict__ components, int role_bytes) {
  extern __shared__ double2 s_dyn[];

  int blk = blockIdx.x;
  if (MASKS && KIND == kBornRows && blk >= role_bytes) {
    if (threadIdx.x >= 256) return;
    if (DEVPAR) rebase_for_parity(P, 0);
    if (((P.estatus[kStatOrderStale] & 2) | P.estatus[kStatMaskAging]) == 0) return;
    return neighbor_tile(P, blk - role_bytes, true);
  }
  // (five-launch mode: the Born rows read this evaluation's self volumes and write its status block -- rows_workgroup asks for
  // the device's evaluation counter with its first loads and moves the two pointers when it first needs them: a rebase HERE
  // would put one more cold scalar round trip in front of every workgroup's prologue, +0.6 us on the launch)
  if (KIND == kChainRows) {  // the chain-rule launch carries the two single-workgroup roles (see k_dborn_tiles): four waves each
    if (blk < 2 && threadIdx.x >= 256) return;
    if (blk == 0) return energy_role<DEVPAR>(P, 1, energy_out, components, reinterpret_cast<char*>(s_dyn));
    if (blk == 1) return dealing_role(P, reinterpret_cast<char*>(s_dyn), role_bytes);
    blk -= 2;
  }
  if (KIND == kGbRows) {  // the GB launch carries the first half of the bookkeeping (see k_gb_tiles)
    if (blk == 0) {
      static_assert(kGbRowWaves >= 4, "the bookkeeping role is written for 256 lanes");
      if (threadIdx.x >= 256) return;
      PAIR_STAMP(1, 0);
      rebase_for_parity(P, 0);
      packing_role<false>(P, reinterpret_cast<char*>(s_dyn), role_bytes);
      PAIR_STAMP(1, 3);
      return;
    }
    blk -= 1;
  }
  __shared__ int s_busy;
  rows_workgroup<KIND, row_waves(KIND), SINGLE, MASKS && DEVPAR>(P, blk, s_dyn, &s_busy);
}

// ---- outputs: one launch, three concurrent roles ---------------------------------------------------------
//   blocks [0, nfb)  forces: F = -grad(tree) + sum of the pair partial rows, ADDED to the caller's buffer
//   block  nfb       energy: fixed-order sum of every energy partial, ADDED to the caller's scalar
//   block  nfb+1     bookkeeping for the NEXT evaluation: tree statistics and the largest-first subtree order

__global__ __launch_bounds__(256) void k_outputs(PairArgs P, int version, double* __restrict__ force_out,
                                                 double* __restrict__ energy_out, double* __restrict__ components, int role_bytes,
                                                 int mask_from) {
  // version 0 has no pair stages to carry the two single-workgroup roles: they are the first two workgroups here
  extern __shared__ char s_role[];  // role_bytes when version != 1
  int blk = blockIdx.x;
  if (mask_from >= 0 && blk >= mask_from) {
    // version 0 in the five-launch mode (round 6): the tiles that lay the level-2 neighbour masks down anew when this evaluation's
    // trailing workgroups found a heavy atom a quarter of the masks' skin from where it was (or beyond half: the evaluation is
    // void) -- what the tail of the Born-rows launch does for version 1 (k_rows, MASKS)
    if (((P.estatus[kStatOrderStale] & 2) | P.estatus[kStatMaskAging]) == 0) return;
    return neighbor_tile(P, blk - mask_from, true);
  }
  rebase_for_parity(P, 1);  // (five-launch mode with an output launch of its own: behind the GB launch)
  if (version != 1) {
    if (blk == 0) return energy_role(P, version, energy_out, components, s_role);
    if (blk == 1) {  // both halves of the bookkeeping, one after the other
      packing_role<true>(P, s_role, role_bytes);
      __threadfence();
      __syncthreads();  // (the second half reads back what the first wrote)
      return dealing_role(P, s_role, role_bytes);
    }
    blk -= 2;
  }
  const int t = threadIdx.x;
  const int i = blk * 256 + t;
  if (P.rows_on && i == 0) {  // this evaluation's neighbour lists are up to date (k_prep of the next one tests again)
    rows_close_evaluation(P.nl_flag, P.nl_nitems, P.row_target, P.gb_rows != 0);
  }
  if (i >= P.n) return;
  double fx = 0, fy = 0, fz = 0;
  const int h = P.a2h[i];
  const int ctx_slot = P.omm.force_fixed ? P.omm.ctx_slot[i] : 0;  // (asked for with the rest, used at the end)
  if (h >= 0) {  // cavity + pseudo-volume gradients -> force
    fx = -P.gx[h];
    fy = -P.gy[h];
    fz = -P.gz[h];
  }
  if (version == 1 && P.rows_on) {
    // chain-rule force of the row form: bw_i G_i + s_i H_i (the second term for heavy atoms only, see k_rows)
    const double bwi = P.bw[i];
    const double4 g = P.grec[i];
    double sh = 0.0;
    double4 hh = make_double4(0.0, 0.0, 0.0, 0.0);
    if (h >= 0) sh = P.sv_vdw[h] * P.inv_vol_h[h], hh = P.hrec[h];
    fx += P.gb_fx[i] + fma(bwi, g.x, sh * hh.x);
    fy += P.gb_fy[i] + fma(bwi, g.y, sh * hh.y);
    fz += P.gb_fz[i] + fma(bwi, g.z, sh * hh.z);
  } else if (version == 1) {
    fx += P.gb_fx[i] + P.db_fx[i];
    fy += P.gb_fy[i] + P.db_fy[i];
    fz += P.gb_fz[i] + P.db_fz[i];
  }
  if (evaluation_overflowed(P.estatus)) return;  // incomplete evaluation: withheld (see energy_role), the caller repeats it
  if (P.omm.force_fixed) {
    // an OpenMM context's force buffer: 64-bit fixed point, value * 2^32 rounded to nearest, three planes over the
    // padded atom count in the context's atom order, integer atomics (GVolReduceTree.cl:117-119)
    auto to_fixed = [](double f) { return (unsigned long long)(long long)rint(f * 4294967296.0); };
    atomicAdd(&P.omm.force_fixed[ctx_slot], to_fixed(fx));
    atomicAdd(&P.omm.force_fixed[ctx_slot + P.omm.padded], to_fixed(fy));
    atomicAdd(&P.omm.force_fixed[ctx_slot + 2 * P.omm.padded], to_fixed(fz));
    return;
  }
  force_out[3 * i] += fx;
  force_out[3 * i + 1] += fy;
  force_out[3 * i + 2] += fz;
}

// ---- launchers -----------------------------------------------------------------------------------------
#define AGBNP_CHECK_LAUNCH()             \
  do {                                   \
    hipError_t e__ = hipGetLastError();  \
    if (e__ != hipSuccess) return e__;   \
  } while (0)

#define AGBNP_MARK(id)                               \
  do {                                               \
    if (tl) {                                        \
      hipError_t m__ = tl->mark(id, st);             \
      if (m__ != hipSuccess) return m__;             \
    }                                                \
  } while (0)

hipError_t launch_masks(const PairArgs& P, hipStream_t st, Timeline* tl) {
  AGBNP_MARK(kKPrep);  // (booked as k_prep: it takes that launch's place in the evaluations that need it)
  const int ref_blocks = (std::max(P.nh, 1) + 255) / 256;
  hipLaunchKernelGGL(k_masks, dim3(ref_blocks + P.nb_tiles), dim3(256), 0, st, P, ref_blocks);
  return hipGetLastError();
}

hipError_t launch_prep(const PairArgs& P, hipStream_t st, Timeline* tl) {
  AGBNP_MARK(kKPrep);
  const int n = std::max(std::max(P.n, P.nslots), (int)kStatEvalWords);
  const int prep_blocks = (n + 255) / 256;
  hipLaunchKernelGGL(k_prep, dim3(prep_blocks + P.nb_tiles), dim3(256), 0, st, P, prep_blocks);
  return hipGetLastError();
}

hipError_t launch_pair_stages(const PairArgs& P, double* energy_out, double* components, hipStream_t st, Timeline* tl) {
  const size_t lds = (size_t)P.lut_entries * sizeof(double2);
  if (lds > 32 * 1024) {  // beyond the default workgroup allowance (k_dborn_tiles adds 22 KB of static tile records and sums)
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_born_tiles), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_dborn_tiles), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lds + 3 * 64 * sizeof(double2)));
    if (e != hipSuccess) return e;
  }
  if (P.rows_on) {  // row form of the two range-limited stages (and, in fast mode, of the GB stage)
    auto gb = P.fast ? k_gb_tiles<true, false, false> : (P.gb_far ? k_gb_tiles<false, false, true> : k_gb_tiles<false, false, false>);
    const int born_groups = (P.n + kRowGroup - 1) / kRowGroup, chain_groups = (P.nh + kRowGroup - 1) / kRowGroup;
    auto walk_blocks = [](int lists, int cap) { return (lists + kRowWaves - 1) / kRowWaves * ((cap + kRowSlice - 1) / kRowSlice); };
    const int born_blocks = walk_blocks(born_groups * kBornParts, P.nlh_cap), chain_blocks = walk_blocks(chain_groups * kChainParts, P.nla_cap);
    const int gb_blocks = (born_groups * kGbParts + kGbRowWaves - 1) / kGbRowWaves * ((P.nlg_cap + kRowSlice - 1) / kRowSlice);
    // the lists of the later launches are built in the Born launch
    const int build_blocks = (chain_groups * kChainParts + (P.gb_rows ? born_groups * kGbParts : 0) + kRowWaves - 1) / kRowWaves;
    const size_t table_lds = (size_t)2 * P.nti * P.ntj * kRowIntervals * sizeof(double2);
    const size_t born_lds = table_lds, chain_lds = std::max(table_lds, sizeof(TileSums));  // (>= what the two roles borrow)
    AGBNP_MARK(kKBornRows);
    if (P.single && P.five)  // (five-launch mode, the single-precision rows of the fast mode: + the conditional mask tiles; host-named set only)
      hipLaunchKernelGGL((k_rows<kBornRows, true, true>), dim3(born_blocks + build_blocks + P.nb_tiles), dim3(64 * kRowWaves), born_lds, st, P, (double*)nullptr,
                         (double*)nullptr, born_blocks + build_blocks);
    else if (P.single)
      hipLaunchKernelGGL((k_rows<kBornRows, true>), dim3(born_blocks + build_blocks), dim3(64 * kRowWaves), born_lds, st, P, (double*)nullptr, (double*)nullptr, 0);
    else if (P.five == 2)  // (five-launch mode, device-side parity: + the conditional mask tiles)
      hipLaunchKernelGGL((k_rows<kBornRows, false, true, true>), dim3(born_blocks + build_blocks + P.nb_tiles), dim3(64 * kRowWaves), born_lds, st, P,
                         (double*)nullptr, (double*)nullptr, born_blocks + build_blocks);
    else if (P.five)  // (five-launch mode: + the conditional mask tiles)
      hipLaunchKernelGGL((k_rows<kBornRows, false, true>), dim3(born_blocks + build_blocks + P.nb_tiles), dim3(64 * kRowWaves), born_lds, st, P, (double*)nullptr,
                         (double*)nullptr, born_blocks + build_blocks);
    else
      hipLaunchKernelGGL(k_rows<kBornRows>, dim3(born_blocks + build_blocks), dim3(64 * kRowWaves), born_lds, st, P, (double*)nullptr, (double*)nullptr, 0);
    AGBNP_CHECK_LAUNCH();
    if (P.gb_rows) {
      AGBNP_MARK(kKGbRows);
      hipLaunchKernelGGL(k_rows<kGbRows>, dim3(1 + gb_blocks), dim3(64 * kGbRowWaves), sizeof(StripSums), st, P, (double*)nullptr, (double*)nullptr, (int)sizeof(StripSums));
    } else {
      AGBNP_MARK(kKGbTiles);
      hipLaunchKernelGGL(gb, dim3(P.gb_items_count + 1), dim3(256), 0, st, P.n, P.gb_items, (const double4*)P.aposq,
                         (const double*)P.born_part, P.inv_rvdw, P.alpha, P.born, P.born_fp, P.brw, P.e_atom, P.gb_fx, P.egb_part, P);
    }
    AGBNP_CHECK_LAUNCH();
    AGBNP_MARK(kKDbornRows);
    if (P.single)
      hipLaunchKernelGGL((k_rows<kChainRows, true>), dim3(2 + chain_blocks), dim3(64 * kRowWaves), chain_lds, st, P, energy_out, components, (int)chain_lds);
    else if (P.five == 2)
      hipLaunchKernelGGL((k_rows<kChainRows, false, false, true>), dim3(2 + chain_blocks), dim3(64 * kRowWaves), chain_lds, st, P, energy_out, components, (int)chain_lds);
    else
      hipLaunchKernelGGL(k_rows<kChainRows>, dim3(2 + chain_blocks), dim3(64 * kRowWaves), chain_lds, st, P, energy_out, components, (int)chain_lds);
    AGBNP_CHECK_LAUNCH();
    return hipSuccess;
  }
  AGBNP_MARK(kKBornTiles);
  if (P.db_items_count > 0)
    hipLaunchKernelGGL(k_born_tiles, dim3(P.db_items_count), dim3(256), lds, st, P.nh, P.nhb, P.ntj, P.lut_entries, P.db_items, P.pslot,
                       (const double*)P.pbox, (const double4*)P.prec, (const double*)P.sv_vdw, P.inv_vol_h, P.lut, P.born_part, P.range2, P.det, P.cull_first);
  AGBNP_CHECK_LAUNCH();
  AGBNP_MARK(kKGbTiles);
  auto gb = P.fast ? (P.single ? k_gb_tiles<true, true, false> : k_gb_tiles<true, false, false>)
                   : (P.gb_far ? k_gb_tiles<false, false, true> : k_gb_tiles<false, false, false>);
  if (P.five)  // (five-launch mode on the tile kernels: the masks' renewal rides at the tail of this launch)
    gb = P.fast ? (P.single ? k_gb_tiles<true, true, false, true> : k_gb_tiles<true, false, false, true>)
                : (P.gb_far ? k_gb_tiles<false, false, true, true> : k_gb_tiles<false, false, false, true>);
  hipLaunchKernelGGL(gb, dim3(P.gb_items_count + 1 + (P.five ? P.nb_tiles : 0)), dim3(256), 0, st, P.n, P.gb_items, (const double4*)P.aposq,
                     (const double*)P.born_part, P.inv_rvdw, P.alpha, P.born, P.born_fp, P.brw, P.e_atom, P.gb_fx, P.egb_part, P);
  AGBNP_CHECK_LAUNCH();
  AGBNP_MARK(kKDbornTiles);
  // (+ 2: the energy workgroup and the dealing workgroup; with no heavy atom there is no tile but the roles still run)
  const size_t db_lds = std::max(lds + 3 * 64 * sizeof(double2), sizeof(TileSums));  // tables + block I's records; later the sums
  hipLaunchKernelGGL(k_dborn_tiles, dim3(P.db_items_count + 2), dim3(256), db_lds, st, P.n, P.nhb, P.ntj, P.lut_entries, P.db_items, P.pslot,
                     (const double*)P.pbox, (const double4*)P.prec, (const double4*)P.srec, (const double*)P.ys, (const double*)P.sv_vdw,
                     P.inv_vol_h, P.nh, P.lut, P.db_fx, P, energy_out, components, (int)db_lds);
  AGBNP_CHECK_LAUNCH();
  return hipSuccess;
}

hipError_t launch_outputs(const PairArgs& P, int version, double* force_out, double* energy_out, double* components, hipStream_t st,
                          Timeline* tl, bool mask_tiles) {
  AGBNP_MARK(kKOutputs);
  // (version 0: the roles' LDS, with room for the packed shapes and as many forest times of up to 6 k subtrees -- or, where that
  // is more, for the rounds rule of the packing: shapes, a round of running sums, the sorted order of ~1.25 items per subtree)
  const int nh1 = std::max(P.nh, 1);
  const int classes_ints = std::min(2 * nh1 + 64, 12288), rounds_ints = std::min(nh1 + P.tree_slots + nh1 + nh1 / 4 + 64, 14000);
  const int role_bytes = version == 1 ? 0 : (int)kRoleScratchBytes + 4 * std::max(classes_ints, rounds_ints);
  const int blocks = (P.n + 255) / 256 + (version == 1 ? 0 : 2);
  hipLaunchKernelGGL(k_outputs, dim3(blocks + (mask_tiles ? P.nb_tiles : 0)), dim3(256), role_bytes, st, P, version, force_out, energy_out, components, role_bytes,
                     mask_tiles ? blocks : -1);
  AGBNP_CHECK_LAUNCH();
  AGBNP_MARK(-1);
  return hipSuccess;
}

}  // namespace agbnp

#ifdef AGBNP_PAIR_STAMPS
extern "C" void agbnp_debug_pair_log(unsigned long long* out) {
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(agbnp::g_pair_log), sizeof(agbnp::g_pair_log));
}
#endif
