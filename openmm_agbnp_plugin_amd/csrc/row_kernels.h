// Row form of the pair stages (device code of k_rows in pair_kernels.hip; rows_close_evaluation is shared with the output
// side of k_tree_pseudo in tree_kernels.hip).  See DESIGN.md s.4e.
#pragma once
#include <hip/hip_runtime.h>

#include "agbnp_common.h"
#include "device_math.h"
#include "pair_kernels.h"

#ifndef PAIR_STAMP  // (diagnostic stamps exist in pair_kernels.hip's -DAGBNP_PAIR_STAMPS build only)
#define PAIR_STAMP(kern, idx)
#define PAIR_STAMP_WAIT(kern, idx, what)
#define PAIR_STAMP_WHERE(kern, item)
#endif
#ifndef PAIR_STAMP_HW
#define PAIR_STAMP_HW(kern)
#endif
#ifndef ROWS_LOG_COUNTS
#define ROWS_LOG_COUNTS(kern, todo, nsteps)
#endif

namespace agbnp {

__device__ __forceinline__ void hbm_add(double* p, double v) {  // global_atomic_add_f64
#ifdef AGBNP_TIMING_NO_ATOMICS  // timing experiment only: results are wrong
  if (v == 1.2345e300) *p = v;
  return;
#endif
  __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ double wave_sum(double v) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// ---- per-atom Born-radius algebra, recomputed by its consumers instead of a kernel of its own ---------------
// beta_i = 1/R_i - (1/4pi) sum_j s_j Q  ->  B_i = 1/f(beta_i), f' (ReferenceAGBNPKernels.cpp:41-55,450-454)
struct BornRadius {
  double br, inv_br, fp;
};
__device__ __forceinline__ BornRadius born_radius(double inv_rvdw, double qsum) {
  const double beta = inv_rvdw - (1. / (4. * kPi)) * qsum;
  const double amin = 1. / kI4MaxA, a2 = 1. / (kI4MaxA * kI4MaxA);
  BornRadius r;
  if (beta < 0.0) {
    r.inv_br = amin;
    r.fp = 0.0;
  } else {
    r.inv_br = sqrt(a2 + beta * beta);
    r.fp = beta / r.inv_br;
  }
  r.br = 1. / r.inv_br;
  return r;
}
// bw_i = brw_i + bru_i with bru_i = -(1/4pi) k (q_i^2 + Y_i B_i) f'_i (ReferenceAGBNPKernels.cpp:524-542) is linear in the GB
// stage's Y sum: bw_i = alpha_i + beta_i Y_i.  With the row form of the chain rule the GB tiles add alpha_i (once, the
// diagonal tile) and beta_i * (their share of Y_i) straight into bw_i, so that a chain-rule row gathers ONE word per neighbour.
__device__ __forceinline__ double bw_beta(const BornRadius& r) { return -(1. / (4. * kPi)) * kDielFactor * r.br * r.fp; }
__device__ __forceinline__ double bw_alpha(const BornRadius& r, double brw, double q) { return brw - (1. / (4. * kPi)) * kDielFactor * (q * q) * r.fp; }

// ---- row form of the pair stages ---------------------------------------------------------------------------------------
// The tile kernels above meet every pair of two 64-atom blocks; on a protein of a few thousand atoms a third of those pairs
// lie inside the tables' 2 nm reach, and a wave pays for all of them.  The row form meets (almost) only pairs in reach and
// needs no atomics and no second look-up per pair:
//
//   Born row of atom a (every atom):      beta-sum_a = sum_b s_b Q_{t(a) t'(b)}(d)          b heavy, b != a, d < 2 nm
//                                          G_a       = sum_b (r_b - r_a) s_b Q'_{t(a) t'(b)}(d) / d
//   chain-rule row of heavy atom a:       (W+U)_a    = sum_b bw_b Q_{t(b) t'(a)}(d)           b of any kind, b != a, d < 2 nm
//                                          H_a       = sum_b (r_b - r_a) bw_b Q'_{t(b) t'(a)}(d) / d
//   chain-rule force:                      F_a      += bw_a G_a + s_a H_a                     (H_a = 0 for a hydrogen)
//
// which is the reference's loop (ReferenceAGBNPKernels.cpp:435-449,555-586) with its two force updates sorted by the atom
// they land on: force[i] += w with w = dist bw_i s_j Q'/d sums up to bw_i G_i, force[j] -= w to s_j H_j.  G does not depend
// on bw, so it rides in the Born rows, where the same table entry is being looked up anyway: each stage looks a pair up
// once, as the symmetric tiles do.
//
// A GROUP = kRowGroup consecutive row atoms (bonded neighbours, a fraction of a nm apart) shares one neighbour list: a lane
// gathers ONE neighbour record per step (list entry -> {x, y, z, .} and weight from memory: the vector-memory pipe is what
// a one-row-per-wave form is bound by) and meets it with the group's row atoms, which are wave-uniform and live in scalar
// registers.  The list of a group is cut into parts, one wave each (the parts take the 64-candidate chunks of the
// candidate order in turn); the sums of a wave meet in a transposing butterfly, those of the parts in LDS, and leave
// through plain stores.  Spline entries in 32-byte power form come from the slices of the group's row types in LDS.
//
// Neighbour lists: entries (index | type << 24) of every candidate within reach + skin of ANY row atom of the group, in
// the order of a static candidate list sorted by type -- the lanes of a step then mostly read consecutive entries of one
// table row: distinct LDS banks.  The lists are rebuilt, on the device and by the waves that own them, in the evaluation
// whose k_prep found an atom further than skin / 2 from where it was at the last build (the lists of both kinds in the
// Born launch, so both see the same positions and an overflowing list is known before the energy is added up);
// d < 2 nm is still tested per pair, so the sums hold exactly the reference's pairs.
constexpr int kRowIntervals = kI4Nodes - 1;

// a value every lane holds alike, moved to scalar registers (the row atoms of a group: VALU operands, no vector registers)
__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ double uniform(double v) {
  return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}
template <int kCtrl>
__device__ __forceinline__ double dpp_move(double v) {  // lane <- the lane that the DPP control names (bound_ctrl: no "old" operand)
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), kCtrl, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), kCtrl, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lane_xor1(double v) { return dpp_move<0xB1>(v); }                   // quad_perm [1, 0, 3, 2]
__device__ __forceinline__ double lane_xor2(double v) { return dpp_move<0x4E>(v); }                   // quad_perm [2, 3, 0, 1]
__device__ __forceinline__ double lane_xor4(double v) { return dpp_move<0x1B>(dpp_move<0x141>(v)); }  // row_half_mirror, then quad_perm [3, 2, 1, 0]
__device__ __forceinline__ double lane_xor8(double v) { return dpp_move<0x128>(v); }                  // row_ror:8

enum RowKind { kBornRows = 0, kChainRows = 1, kGbRows = 2 };
// Which atoms own rows, which are candidates:   Born rows: every atom <- heavy atoms;  chain-rule rows: heavy atoms <- every
// atom;  GB rows (fast mode only: the reference's GB meets ALL pairs and stays on the tiles): every atom <- every atom
// within the cutoff.
struct RowAtoms {  // wave-uniform (scalar registers): the row atoms of a group
  double x[kRowGroup], y[kRowGroup], z[kRowGroup];
  int self[kRowGroup];   // index of the row atom in the candidates' numbering (never met: -1 for a row that is no candidate)
  int rows;              // valid rows (the last group may be short)
};

template <int KIND>
__device__ __forceinline__ RowAtoms row_atoms(const PairArgs& P, int group) {
  RowAtoms A;
  const int nrows = KIND == kChainRows ? P.nh : P.n;
  A.rows = min(kRowGroup, nrows - kRowGroup * group);
  const double4* __restrict__ pos = KIND == kChainRows ? static_cast<const double4*>(P.hrow) : static_cast<const double4*>(P.aposq);
  double4 pr[kRowGroup];
  int self[kRowGroup];
#pragma unroll
  for (int r = 0; r < kRowGroup; r++) {  // (all in flight together)
    const int row = min(kRowGroup * group + r, nrows - 1);
    pr[r] = pos[row];
    self[r] = KIND == kBornRows ? P.a2h[row] : row;  // (Born rows: -1 for a hydrogen, no candidate has that index)
  }
#pragma unroll
  for (int r = 0; r < kRowGroup; r++) {
    const bool there = r < A.rows;
    A.x[r] = uniform(there ? pr[r].x : 1e30);  // a row that does not exist is out of everybody's reach
    A.y[r] = uniform(pr[r].y);
    A.z[r] = uniform(pr[r].z);
    A.self[r] = uniform(!there ? -1 : KIND == kChainRows ? (__double2loint(pr[r].w) & 0xffffff) : self[r]);
  }
  return A;
}

// builds one part of a group's list: the candidates perm[64 c + lane] of chunks c = part, part + parts, ... (index |
// type << 24, ~0u = padding), their records rec[index] = {x, y, z, .}; returns the number of entries (may exceed stride)
__device__ __forceinline__ int row_build(const RowAtoms& A, const unsigned* __restrict__ perm, int np, int part, int parts,
                                         const double4* __restrict__ rec, double build2, unsigned* __restrict__ list, int stride, int lane) {
  int cnt = 0;
  for (int base = 64 * part; base < np; base += 64 * parts) {
    const unsigned e = perm[base + lane];
    const double4 r = rec[e != ~0u ? (int)(e & 0xffffffu) : 0];
    double dmin = 1e300;
#pragma unroll
    for (int q = 0; q < kRowGroup; q++) {
      const double dx = r.x - A.x[q], dy = r.y - A.y[q], dz = r.z - A.z[q];
      dmin = fmin(dmin, fma(dz, dz, fma(dy, dy, dx * dx)));
    }
    const bool ok = e != ~0u && dmin < build2;
    const unsigned long long m = __ballot(ok);
    const int at = cnt + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
    if (ok && at < stride) list[at] = e;
    cnt += __popcll(m);
  }
  return cnt;
}

// Work items: a list is walked in SLICES of kRowSlice entries (four steps), one wave each, so that no wave works longer
// than four steps whatever the length of its list (the launch lasts as long as its slowest wave); item = slice * lists +
// list, eight consecutive items per workgroup (a workgroup's items are the same slice of eight neighbouring lists: it is
// empty as a whole, and leaves at once, or not at all).  The sums of a wave leave as one set of FP64 atomics.
// The slice length in use lives on the device (nl_flag[2]) and only grows.  A launch whose working workgroups are a few
// more than two per CU is as long as the workgroups on the CUs that hold three (1dwc, 596 workgroups on 256 CUs: lifetimes
// up to 9.5 us on the CUs with two, up to 12.9 us on those with three); longer slices for a fifth fewer workgroups take
// that tail away.  The evaluation that has rebuilt the lists looks at the work items it laid down and, if any row launch
// has between one and two times `row_target` workgroups, asks for one more rebuild with slices of 64 entries more.
__device__ __forceinline__ int row_slice_length(const PairArgs& P) { return min(max(P.nl_flag[2], kRowSlice), kRowSliceMax); }
// the last launch of an evaluation (one lane): a rebuild is counted, the lists are good from here on
__device__ __forceinline__ void rows_close_evaluation(int* nl_flag, const int* nl_nitems, int row_target, bool gb_rows) {
  if (!nl_flag[0]) return;
  nl_flag[1] += 1;  // (builds so far; its parity names the work-item buffers in use)
  nl_flag[0] = 0;
  if (row_target <= 0) return;
  const int buf = nl_flag[1] & 1;
  int items = max(nl_nitems[2 * 0 + buf], nl_nitems[2 * 1 + buf]);  // (Born rows, chain-rule rows)
  (void)gb_rows;  // (the GB rows' workgroups are small and their launch holds fewer waves than the device: left alone)
  const int wgs = (items + kRowWaves - 1) / kRowWaves, rs = nl_flag[2];
  if (wgs > row_target && wgs <= 2 * row_target && rs < kRowSliceMax) {
    nl_flag[2] = max(rs, kRowSlice) + 64;
    nl_flag[0] = 1;
  }
}

struct RowLists {  // the lists of one kind
  unsigned* list;
  int* count;
  int stride, cap, groups, parts;  // cap: entries the launches walk (<= stride)
};
template <int KIND>
__device__ __forceinline__ RowLists row_lists(const PairArgs& P) {
  RowLists L;
  L.list = KIND == kBornRows ? P.nlh : KIND == kChainRows ? P.nla : P.nlg;
  L.count = KIND == kBornRows ? P.nlh_count : KIND == kChainRows ? P.nla_count : P.nlg_count;
  L.stride = KIND == kBornRows ? P.nlh_stride : KIND == kChainRows ? P.nla_stride : P.nlg_stride;
  L.cap = KIND == kBornRows ? P.nlh_cap : KIND == kChainRows ? P.nla_cap : P.nlg_cap;
  L.groups = ((KIND == kChainRows ? P.nh : P.n) + kRowGroup - 1) / kRowGroup;
  L.parts = KIND == kBornRows ? kBornParts : KIND == kChainRows ? kChainParts : kGbParts;
  return L;
}
// the work items of a freshly built list go into the buffer that the NEXT evaluations use (see PairArgs::nl_items); one lane
template <int KIND>
__device__ __forceinline__ void append_items(const PairArgs& P, int sub, int entries, bool at_least_one) {
  const int rs = row_slice_length(P);
  const int nsl = max((entries + rs - 1) / rs, at_least_one ? 1 : 0);
  if (nsl == 0) return;
  const int buf = (P.nl_flag[1] + 1) & 1;
  const int base = atomicAdd(&P.nl_nitems[2 * KIND + buf], nsl);
  unsigned* items = P.nl_items + (size_t)(2 * KIND + buf) * P.nl_items_cap;
  for (int sl = 0; sl < nsl && base + sl < P.nl_items_cap; sl++) items[base + sl] = (unsigned)sub | ((unsigned)sl << 24);
}

// one wave builds list `sub` of kind KIND (the lists of the later launches are built in the Born launch: both see the same
// positions, and an overflowing list is known before the energy is added up)
template <int KIND>
__device__ __forceinline__ void build_list(const PairArgs& P, int sub, int lane, int stale, int* estatus) {  // estatus: the evaluation's status block
  const RowLists L = row_lists<KIND>(P);
  if (sub >= L.groups * L.parts) return;
  if (!stale) {  // the list stands; is all of it walked?  (the walk may have been set up for a shorter reach, or narrowed)
    if (lane == 0 && L.count[sub] > L.cap) estatus[kStatRowOverflow] = 1;
    return;
  }
  const int g = sub / L.parts, part = sub - g * L.parts;
  const RowAtoms A = row_atoms<KIND>(P, g);
  const int cnt = row_build(A, P.aperm, P.aperm_n, part, L.parts, static_cast<const double4*>(P.aposq), KIND == kGbRows ? P.nlg_build2 : P.nl_build2,
                            L.list + (size_t)sub * L.stride, L.stride, lane);
  if (lane == 0) {
    // the length is stored UNtruncated (it may exceed the stride) and clamped where it bounds a walk: every later evaluation
    // queued before the host reacts sees count > cap above and is withheld too, not just the one that rebuilt the list
    L.count[sub] = cnt;
    if (cnt > L.cap) estatus[kStatRowOverflow] = 1;
    // (GB rows: the first slice of a group's first list also publishes the per-atom results, neighbours or not)
    append_items<KIND>(P, sub, min(cnt, L.cap), KIND == kGbRows && part == 0);
  }
}

// The work of one workgroup of WAVES waves of a row launch of kind KIND (blk: its number among the launch's row
// workgroups).
// SINGLE (Born and chain-rule rows under AGBNP_HIP_MODE_FAST | AGBNP_HIP_MODE_SINGLE): the pair terms in single precision, as
// the reference's GPU platform computes them (AGBNPBornRadii.cl:181-430 is all-float): the table in LDS as one 16-byte
// float4 {c0, c1, c2, c3} per entry (half the copy, one look-up per pair instead of two), distances from positions relative
// to the group's first row atom, hardware rsqrt, the sums of a slice in FP32; everything that leaves a wave stays FP64.
// PARITY (five-launch mode, Born rows): the self volumes and the status block are those of the evaluation's set (pair_kernels.h,
// rebase_for_parity); the counter that names it is loaded with the first loads and used where the two pointers are.
template <int KIND, int WAVES, bool SINGLE = false, bool PARITY = false>
__device__ __forceinline__ void rows_workgroup(const PairArgs& P, int blk, double2* s_dyn, int* s_busy_word) {
  static_assert(!SINGLE || KIND != kGbRows, "the GB rows choose their precision at run time (P.single)");
  constexpr int R = kRowGroup;
  static_assert(R == 4, "the butterfly below folds 16 sums");
  int& s_busy = *s_busy_word;  // (a word of the caller's LDS)
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const RowLists L = row_lists<KIND>(P);
  const int NP = L.parts, stride = L.stride;
  const int nlists = L.groups * NP, lists_pad = (nlists + WAVES - 1) / WAVES * WAVES;
  const int stale = P.nl_flag[0];
  const int epoch_now = PARITY ? P.epoch[0] : 0;  // (device-side parity, pair_kernels.h: in flight with the flags; else the host has moved the pointers)
  const int rs = row_slice_length(P);
  const int walk_blocks = lists_pad / WAVES * ((L.cap + kRowSlice - 1) / kRowSlice);  // (as the host lays the grid out)
  auto flag_row_overflow = [&]() { P.estatus[(PARITY ? 16 * (epoch_now & 1) : 0) + kStatRowOverflow] = 1; };
  if (KIND == kBornRows && blk >= walk_blocks) {
    // The lists of the later launches are built here, in the Born launch, by workgroups that only look at the lists'
    // lengths in an evaluation whose lists are still good.
    const int sub = (blk - walk_blocks) * WAVES + wave;
    const int chain_lists = row_lists<kChainRows>(P).groups * kChainParts;
    if (sub < chain_lists)
      build_list<kChainRows>(P, sub, lane, stale, P.estatus + (PARITY ? 16 * (epoch_now & 1) : 0));
    else if (P.gb_rows)
      build_list<kGbRows>(P, sub - chain_lists, lane, stale, P.estatus + (PARITY ? 16 * (epoch_now & 1) : 0));
    return;
  }
  PAIR_STAMP((KIND == kChainRows ? 2 : KIND == kGbRows ? 1 : 0), 0);
  const int item = blk * WAVES + wave;
  // Work items.  Normally from the compact list laid down when the lists were built: the launch's first workgroups are the
  // ones with work, the rest leave here.  The Born rows of a rebuild evaluation build their lists themselves, slice by
  // slice, and are laid out by slice number for that one evaluation.
  const bool compact = !(KIND == kBornRows && stale);
  int slice, li;
  bool active;
  if (compact) {
    const int buf = (P.nl_flag[1] + (stale ? 1 : 0)) & 1;  // (a rebuild evaluation's later launches already walk the new lists)
    const int nitems = min(P.nl_nitems[2 * KIND + buf], P.nl_items_cap);
    if (blk * WAVES >= nitems) {
      if (KIND == kGbRows && lane == 0 && item < P.egb_parts) P.egb_part[item] = 0.0;  // (every partial is summed up)
      return;
    }
    PAIR_STAMP_HW((KIND == kChainRows ? 2 : KIND == kGbRows ? 1 : 0));
    PAIR_STAMP((KIND == kChainRows ? 2 : KIND == kGbRows ? 1 : 0), 10);
    active = item < nitems;
    const unsigned it = (P.nl_items + (size_t)(2 * KIND + buf) * P.nl_items_cap)[active ? item : 0];
    PAIR_STAMP_WAIT((KIND == kChainRows ? 2 : KIND == kGbRows ? 1 : 0), 11, "vmcnt(0)");
    li = (int)(it & 0xffffffu);
    slice = (int)(it >> 24);
  } else {
    slice = item / lists_pad, li = item - slice * lists_pad;  // (a workgroup's items share the slice)
    active = li < nlists;
  }
  const int sub = active ? li : 0;  // (group, part)
  const int group = sub / NP, part = sub - group * NP;
  const unsigned* list = L.list + (size_t)sub * stride;  // (not restrict: a build rewrites it)
  const int first = rs * slice;
  // Everything that does not depend on anything is asked for at once: the length of the list, the first two steps of the
  // slice (the lists start out zeroed: an entry beyond the length is a valid index), the row atoms and their types, the table.
  const int listed_raw = L.count[sub];  // (untruncated: beyond the stride for a list that did not fit)
  const int listed = min(listed_raw, stride);
  unsigned e1 = list[min(first + lane, stride - 1)], e2 = list[min(first + 64 + lane, stride - 1)];
  const unsigned types = KIND == kGbRows ? 0u : (KIND == kChainRows ? P.cslice : P.bslice)[group];  // one byte per row
  const RowAtoms A = row_atoms<KIND>(P, group);
  double row_rv[R], row_bp[R], row_q[R];  // GB rows: 1/R_vdw, descreening sum and charge of the row atoms
  if (KIND == kGbRows) {
#pragma unroll
    for (int r = 0; r < R; r++) {
      const int a = min(R * group + r, P.n - 1);
      row_rv[r] = P.inv_rvdw[a], row_bp[r] = P.born_part[a], row_q[r] = static_cast<const double4*>(P.aposq)[a].w;
    }
  }
  const int ne = (KIND == kChainRows ? P.nti : P.ntj) * kRowIntervals;  // entries of a slice of the table (one row type)
  const int tab = P.nti * P.ntj * kRowIntervals;                        // ... of the table: {c0, c1} of every entry, then {c2, c3}
  double2* const s_tab = s_dyn;
  constexpr int kWg = 64 * WAVES;
  const double2* __restrict__ gtab = KIND == kChainRows ? P.pwt_a : P.pw_a;  // (pw_b / pwt_b follow pw_a / pwt_a in memory)
  const int tx = threadIdx.x;
  // a workgroup whose eight slices are all beyond the ends of their lists has nothing to do (not known while the lists
  // are being rebuilt); it leaves before it asks for the table: nearly half of the workgroups of a launch are such
  // (GB rows: the first slice of a group's first list also publishes the per-atom results, neighbours or not)
  const bool mine = active && ((KIND == kBornRows && stale) || first < listed || (KIND == kGbRows && slice == 0 && part == 0));
  if (KIND == kGbRows && lane == 0 && !mine && item < P.egb_parts) P.egb_part[item] = 0.0;  // (every partial is summed up)
  if (!compact) {  // (laid out by slice number: a workgroup may be empty as a whole)
    if (threadIdx.x == 0) s_busy = 0;
    __syncthreads();
    if (mine && lane == 0) s_busy = 1;
    __syncthreads();
    if (!s_busy) return;
  }
  float4* const s_tab4 = reinterpret_cast<float4*>(s_dyn);  // SINGLE: {c0, c1, c2, c3} per entry
  double2 tv0, tv1, tv2, tv3;
  if (KIND != kGbRows && !SINGLE) tv0 = gtab[min(tx, 2 * tab - 1)], tv1 = gtab[min(kWg + tx, 2 * tab - 1)], tv2 = gtab[min(2 * kWg + tx, 2 * tab - 1)];  // (1dwc: 1440 entries)
  if (KIND != kGbRows && SINGLE)  // (1dwc: 720 entries, both halves of an entry per thread)
    tv0 = gtab[min(tx, tab - 1)], tv1 = gtab[tab + min(tx, tab - 1)], tv2 = gtab[min(kWg + tx, tab - 1)], tv3 = gtab[tab + min(kWg + tx, tab - 1)];
  PAIR_STAMP_WAIT((KIND == kChainRows ? 2 : KIND == kGbRows ? 1 : 0), 7, "vmcnt(0) lgkmcnt(0)");
  if (KIND != kGbRows && !SINGLE) {
    if (tx < 2 * tab) s_tab[tx] = tv0;
    if (kWg + tx < 2 * tab) s_tab[kWg + tx] = tv1;
    if (2 * kWg + tx < 2 * tab) s_tab[2 * kWg + tx] = tv2;
    for (int base = 3 * kWg; base < 2 * tab; base += kWg)  // (larger tables)
      if (base + tx < 2 * tab) s_tab[base + tx] = gtab[base + tx];
  }
  if (KIND != kGbRows && SINGLE) {
    if (tx < tab) s_tab4[tx] = make_float4((float)tv0.x, (float)tv0.y, (float)tv1.x, (float)tv1.y);
    if (kWg + tx < tab) s_tab4[kWg + tx] = make_float4((float)tv2.x, (float)tv2.y, (float)tv3.x, (float)tv3.y);
    for (int base = 2 * kWg; base < tab; base += kWg)  // (larger tables)
      if (base + tx < tab) {
        const double2 a = gtab[base + tx], b = gtab[tab + base + tx];
        s_tab4[base + tx] = make_float4((float)a.x, (float)a.y, (float)b.x, (float)b.y);
      }
  }
  int slice_at[R];  // first entry of the row's slice of the table
#pragma unroll
  for (int r = 0; r < R; r++) slice_at[r] = uniform((int)((types >> (8 * r)) & 0xffu) * ne);
  int count = listed;
  if (KIND == kBornRows && stale && active) {
    // every slice of a list rebuilds the list for itself (the same entries at the same places: the copies agree), so that
    // no wave waits for another workgroup's
    count = row_build(A, P.hperm, P.hperm_n, part, NP, static_cast<const double4*>(P.rec_h), P.nl_build2, P.nlh + (size_t)sub * stride, stride, lane);
    if (lane == 0 && slice == 0) {
      P.nlh_count[sub] = count;  // (untruncated, see build_list)
      if (count > L.cap) flag_row_overflow();
      append_items<kBornRows>(P, sub, min(count, L.cap), false);
    }
    if (slice == 0 && part == 0 && lane < A.rows) {  // where the atoms were when the lists were built
      const int a = kRowGroup * group + lane;
      const double4 pa = static_cast<const double4*>(P.aposq)[a];
      P.nl_ref[3 * a] = pa.x, P.nl_ref[3 * a + 1] = pa.y, P.nl_ref[3 * a + 2] = pa.z;
    }
    count = min(count, stride);
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");  // (the wave's own stores, read back by other lanes)
    e1 = list[min(first + lane, stride - 1)], e2 = list[min(first + 64 + lane, stride - 1)];
  }
  if (KIND == kBornRows && active && slice == 0 && lane == 0 && !stale && listed_raw > L.cap) flag_row_overflow();  // (not all of it is walked)
  const int todo = active ? max(0, min(count - first, rs)) : 0;  // entries of this slice
  const int nsteps = (todo + 63) >> 6;
  double acc[4 * R];
#pragma unroll
  for (int q = 0; q < 4 * R; q++) acc[q] = 0.0;
  constexpr double kPerNode = (kI4Nodes - 1) / kI4MaxA;
  double esum = 0.0;    // GB rows: sum of q_a q_b f over the wave's pairs
  double beta_r[R];     // GB rows: beta of the row atoms (see bw_beta)
  if (KIND != kGbRows) {
    const double4* __restrict__ rec = KIND == kChainRows ? static_cast<const double4*>(P.aposq) : static_cast<const double4*>(P.rec_h);
    const double* __restrict__ wsrc = KIND == kChainRows ? static_cast<const double*>(P.bw)
                                                          : static_cast<const double*>(P.sv_vdw) + (PARITY ? (size_t)(epoch_now & 1) * P.table_doubles : 0);
    // two steps ahead: the list entry; one step ahead: the neighbour's record and weight.  (Every load is unconditional, its
    // index clamped into the list's stride: a load under a condition makes the compiler wait for everything in flight.)
    double4 r1 = rec[e1 & 0xffffffu];
    double w1 = wsrc[e1 & 0xffffffu];
    __syncthreads();  // the table is in LDS
    PAIR_STAMP((KIND == kChainRows ? 2 : 0), 8);
    PAIR_STAMP_WAIT((KIND == kChainRows ? 2 : 0), 1, "vmcnt(0)");  // the first records are here
    ROWS_LOG_COUNTS((KIND == kChainRows ? 2 : 0), todo, nsteps);
    if (SINGLE) {
      float ox[R], oy[R], oz[R], accf[4 * R];
#pragma unroll
      for (int r = 0; r < R; r++) ox[r] = (float)(A.x[r] - A.x[0]), oy[r] = (float)(A.y[r] - A.y[0]), oz[r] = (float)(A.z[r] - A.z[0]);
#pragma unroll
      for (int q = 0; q < 4 * R; q++) accf[q] = 0.0f;
      const float per_node = (float)kPerNode, range2f = (float)P.range2;
      for (int k = 0; k < nsteps; k++) {
        const unsigned e = e1;
        const double4 rb = r1;
        const double wb = w1;
        e1 = e2;
        r1 = rec[e1 & 0xffffffu];
        w1 = wsrc[e1 & 0xffffffu];
        e2 = list[min(first + 64 * (k + 2) + lane, stride - 1)];
        const int b = (int)(e & 0xffffffu);
        const int tent = (int)(e >> 24) * kRowIntervals;
        const float w = (float)((KIND == kChainRows ? wb : wb * rb.w) * kPerNode);
        const float cut2 = 64 * k + lane < todo ? range2f : -1.0f;  // (a lane beyond the slice meets nobody)
        const float nx = (float)(rb.x - A.x[0]), ny = (float)(rb.y - A.y[0]), nz = (float)(rb.z - A.z[0]);
#pragma unroll
        for (int r = 0; r < R; r++) {
          const float dx = nx - ox[r], dy = ny - oy[r], dz = nz - oz[r];
          const float d2 = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
          if (d2 < cut2 && b != A.self[r]) {
            const float rinv = __builtin_amdgcn_rsqf(d2);
            const float u = fminf((d2 * rinv) * per_node, (float)kRowIntervals - 1e-3f);  // (rounding must not step past the last interval)
            const int ent = slice_at[r] + tent + (int)u;
            const float t = u - (float)(int)u;
            const float4 c = s_tab4[ent];
            const float b2 = fmaf(c.w, t, c.z), b1 = fmaf(b2, t, c.y);
            const float val = fmaf(b1, t, c.x), der = fmaf(fmaf(c.w, t, b2), t, b1);
            accf[4 * r] = fmaf(w, val, accf[4 * r]);
            const float g = w * der * rinv;
            accf[4 * r + 1] = fmaf(dx, g, accf[4 * r + 1]);
            accf[4 * r + 2] = fmaf(dy, g, accf[4 * r + 2]);
            accf[4 * r + 3] = fmaf(dz, g, accf[4 * r + 3]);
          }
        }
      }
#pragma unroll
      for (int q = 0; q < 4 * R; q++) acc[q] = (double)accf[q];
    } else
    for (int k = 0; k < nsteps; k++) {
      const unsigned e = e1;
      const double4 rb = r1;
      const double wb = w1;
      e1 = e2;
      r1 = rec[e1 & 0xffffffu];
      w1 = wsrc[e1 & 0xffffffu];
      e2 = list[min(first + 64 * (k + 2) + lane, stride - 1)];
      const int b = (int)(e & 0xffffffu);
      const int tent = (int)(e >> 24) * kRowIntervals;
      const double w = (KIND == kChainRows ? wb : wb * rb.w) * kPerNode;  // bw_b, or s_b = selfvol_b / V_b (times the table's d(t)/d(d))
      const double range2 = 64 * k + lane < todo ? P.range2 : -1.0;  // (a lane beyond the slice meets nobody)
#pragma unroll
      for (int r = 0; r < R; r++) {
        const double dx = rb.x - A.x[r], dy = rb.y - A.y[r], dz = rb.z - A.z[r];
        const double d2 = fma(dz, dz, fma(dy, dy, dx * dx));
        if (d2 < range2 && b != A.self[r]) {
          const double rinv = rsqrt_pos(d2);
          const double u = (d2 * rinv) * kPerNode;
          const int ent = slice_at[r] + tent + (int)u;
          const double t = __builtin_amdgcn_fract(u);
          const double2 ca = s_tab[ent], cb = s_tab[ent + tab];
          // value and derivative of c0 + c1 t + c2 t^2 + c3 t^3 in five operations
          const double b2 = fma(cb.y, t, cb.x), b1 = fma(b2, t, ca.y);
          const double val = fma(b1, t, ca.x), der = fma(fma(cb.y, t, b2), t, b1);
          acc[4 * r] = fma(w, val, acc[4 * r]);
          const double g = w * der * rinv;
          acc[4 * r + 1] = fma(dx, g, acc[4 * r + 1]);
          acc[4 * r + 2] = fma(dy, g, acc[4 * r + 2]);
          acc[4 * r + 3] = fma(dz, g, acc[4 * r + 3]);
        }
      }
    }
  } else {
    // GB rows (fast mode): pair terms as in k_gb_tiles, every ordered pair from its row's side -- F_a = -2k sum_b D q_a q_b
    // (1 - et/4) f^3, Y_a = sum_b q_a q_b (B_a B_b + d^2/4) et f^3, and the pair energy 2k sum_{a<b} = k sum_a sum_{b != a}.
    // A neighbour's Born radius is formed from its finished descreening sum on the fly (one per lane and step).
    const double4* __restrict__ rec = static_cast<const double4*>(P.aposq);
    double qa[R], ba[R], ca_[R];  // charge, B, -log2(e) / (4 B) of the row atoms
    BornRadius bra[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
      bra[r] = born_radius(row_rv[r], row_bp[r]);
      qa[r] = uniform(r < A.rows ? row_q[r] : 0.0);
      ba[r] = uniform(bra[r].br);
      ca_[r] = uniform((-0.25 * 1.4426950408889634074) * bra[r].inv_br);
      beta_r[r] = bw_beta(bra[r]);
    }
    if (active && slice == 0 && part == 0 && lane < A.rows) {
      // the first slice of a group's first list publishes the per-atom results once (as the diagonal tile does in
      // k_gb_tiles): B_i, f'_i, vdW energy + GB self energy, brw_i, and alpha_i of bw_i (ReferenceAGBNPKernels.cpp:477,513-533)
      const int a = R * group + lane;
      BornRadius br = bra[0];
      double qv = row_q[0];
#pragma unroll
      for (int r = 1; r < R; r++) {
        if (lane == r) br = bra[r], qv = row_q[r];
      }
      const double al = P.alpha[a];
      const double bh = br.br + kHBRadius, bh3 = bh * bh * bh;
      const double brw_a = -(1. / (4. * kPi)) * 3. * al * br.br * br.br * br.fp / (bh3 * bh);
      P.born[a] = br.br;
      P.born_fp[a] = br.fp;
      P.e_atom[a] = al / bh3 + kDielFactor * qv * qv * br.inv_br;
      P.brw[a] = brw_a;
      hbm_add(&P.bw[a], bw_alpha(br, brw_a, qv));
    }
    double4 r1 = rec[e1 & 0xffffffu];
    double p1 = P.born_part[e1 & 0xffffffu], v1 = P.inv_rvdw[e1 & 0xffffffu];
    PAIR_STAMP(1, 8);
    PAIR_STAMP_WAIT(1, 1, "vmcnt(0)");  // the first records are here
    ROWS_LOG_COUNTS(1, todo, nsteps);
    if (P.single) {
      // AGBNP_HIP_MODE_SINGLE: the pair terms in single precision, the default precision of the reference's GPU platform
      // (hardware exp2 / rsqrt, a third of the instructions).  Positions are taken relative to the group's first row atom
      // before they are rounded (1e-7 of a nanometre, not of the box); the sums of a slice run in single precision and
      // leave in FP64 like the others; Born radii stay FP64.
      float ox[R], oy[R], oz[R], qf[R], bf[R], cf[R], accf[4 * R];
#pragma unroll
      for (int r = 0; r < R; r++) {
        ox[r] = (float)(A.x[r] - A.x[0]), oy[r] = (float)(A.y[r] - A.y[0]), oz[r] = (float)(A.z[r] - A.z[0]);
        qf[r] = (float)qa[r], bf[r] = (float)ba[r], cf[r] = (float)ca_[r];
      }
#pragma unroll
      for (int q = 0; q < 4 * R; q++) accf[q] = 0.0f;
      float esumf = 0.0f;
      for (int k = 0; k < nsteps; k++) {
        const unsigned e = e1;
        const double4 rb = r1;
        const BornRadius bb_ = born_radius(v1, p1);
        e1 = e2;
        r1 = rec[e1 & 0xffffffu];
        p1 = P.born_part[e1 & 0xffffffu], v1 = P.inv_rvdw[e1 & 0xffffffu];
        e2 = list[min(first + 64 * (k + 2) + lane, stride - 1)];
        const int b = (int)(e & 0xffffffu);
        const float cut2 = 64 * k + lane < todo ? (float)P.gb_cut2 : -1.0f;  // (a lane beyond the slice meets nobody)
        const float nx = (float)(rb.x - A.x[0]), ny = (float)(rb.y - A.y[0]), nz = (float)(rb.z - A.z[0]);
        const float qb = (float)rb.w, bj = (float)bb_.br, ibj = (float)bb_.inv_br;
#pragma unroll
        for (int r = 0; r < R; r++) {
          const float dx = nx - ox[r], dy = ny - oy[r], dz = nz - oz[r];
          const float d2 = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
          if (d2 < cut2 && b != A.self[r]) {
            const float bb = bf[r] * bj;
            const float et = __builtin_amdgcn_exp2f(d2 * (cf[r] * ibj));  // exp(-d^2 / (4 B_a B_b))
            const float fgb = __builtin_amdgcn_rsqf(fmaf(bb, et, d2));
            const float s1 = (qf[r] * qb) * fgb;
            esumf += s1;
            const float s3 = s1 * (fgb * fgb);
            const float mw = fmaf(-0.25f, et, 1.0f) * s3;
            accf[4 * r] = fmaf(fmaf(0.25f, d2, bb), et * s3, accf[4 * r]);  // Y
            accf[4 * r + 1] = fmaf(dx, mw, accf[4 * r + 1]);
            accf[4 * r + 2] = fmaf(dy, mw, accf[4 * r + 2]);
            accf[4 * r + 3] = fmaf(dz, mw, accf[4 * r + 3]);
          }
        }
      }
#pragma unroll
      for (int q = 0; q < 4 * R; q++) acc[q] = (double)accf[q];
      esum = (double)esumf;
    } else
    for (int k = 0; k < nsteps; k++) {
      const unsigned e = e1;
      const double4 rb = r1;
      const BornRadius bb_ = born_radius(v1, p1);
      e1 = e2;
      r1 = rec[e1 & 0xffffffu];
      p1 = P.born_part[e1 & 0xffffffu], v1 = P.inv_rvdw[e1 & 0xffffffu];
      e2 = list[min(first + 64 * (k + 2) + lane, stride - 1)];
      const int b = (int)(e & 0xffffffu);
      const double cut2 = 64 * k + lane < todo ? P.gb_cut2 : -1.0;  // (a lane beyond the slice meets nobody)
#pragma unroll
      for (int r = 0; r < R; r++) {
        const double dx = rb.x - A.x[r], dy = rb.y - A.y[r], dz = rb.z - A.z[r];
        const double d2 = fma(dz, dz, fma(dy, dy, dx * dx));
        if (d2 < cut2 && b != A.self[r]) {
          const double bb = ba[r] * bb_.br;
          const double et = exp2_nonpositive(d2 * (ca_[r] * bb_.inv_br));  // exp(-d^2 / (4 B_a B_b))
          const double fgb = rsqrt_pos(fma(bb, et, d2));
          const double s1 = (qa[r] * rb.w) * fgb;
          esum += s1;
          const double s3 = s1 * (fgb * fgb);
          const double mw = fma(-0.25, et, 1.0) * s3;
          acc[4 * r] = fma(fma(0.25, d2, bb), et * s3, acc[4 * r]);  // Y
          acc[4 * r + 1] = fma(dx, mw, acc[4 * r + 1]);
          acc[4 * r + 2] = fma(dy, mw, acc[4 * r + 2]);
          acc[4 * r + 3] = fma(dz, mw, acc[4 * r + 3]);
        }
      }
    }
    esum = wave_sum(esum);
    if (lane == 0 && item < P.egb_parts) P.egb_part[item] = kDielFactor * esum;
  }
  PAIR_STAMP((KIND == kChainRows ? 2 : KIND == kGbRows ? 1 : 0), 2);
  if (nsteps == 0) return;
  // 16 sums per lane -> one per lane: four transposing butterfly stages (a lane keeps the half of the sums that its bit of
  // the stage selects and adds its partner's copy of them; DPP moves), then the four 16-lane rows of the wave are added up
  const bool u0 = lane & 1, u1 = lane & 2, u2 = lane & 4, u3 = lane & 8;
#pragma unroll
  for (int q = 0; q < 8; q++) acc[q] = (u0 ? acc[q + 8] : acc[q]) + lane_xor1(u0 ? acc[q] : acc[q + 8]);
#pragma unroll
  for (int q = 0; q < 4; q++) acc[q] = (u1 ? acc[q + 4] : acc[q]) + lane_xor2(u1 ? acc[q] : acc[q + 4]);
#pragma unroll
  for (int q = 0; q < 2; q++) acc[q] = (u2 ? acc[q + 2] : acc[q]) + lane_xor4(u2 ? acc[q] : acc[q + 2]);
  double total = (u3 ? acc[1] : acc[0]) + lane_xor8(u3 ? acc[0] : acc[1]);
  total += __shfl_xor(total, 16, 64);
  total += __shfl_xor(total, 32, 64);
  // bit s of the lane chose the halves of stage s: the lane's sum is number bit-reversed(lane & 15) = 4 * row + quantity
  const int q = ((lane & 1) << 3) | ((lane & 2) << 1) | ((lane & 4) >> 1) | ((lane & 8) >> 3);
  const int r = q >> 2, c = q & 3;
  if (lane < 16 && r < A.rows) {
    const int row = R * group + r;  // atom (Born and GB rows) or heavy index (chain-rule rows)
    if (KIND == kGbRows) {
      double beta = 0.0;
#pragma unroll
      for (int rr = 0; rr < R; rr++) beta = rr == r ? beta_r[rr] : beta;
      if (c == 0)
        hbm_add(&P.bw[row], beta * total);  // bw_a = alpha_a + beta_a Y_a
      else
        hbm_add(P.gb_fx + (size_t)(c - 1) * P.n + row, (-2.0 * kDielFactor) * total);
    } else if (c == 0) {
      // (the unit of the value sums goes back in: w carried the table's d(t)/d(d) for the derivatives)
      hbm_add(KIND == kChainRows ? &P.db_wu[row] : &P.born_part[row], total * (1.0 / kPerNode));
    } else {
      hbm_add(reinterpret_cast<double*>((KIND == kChainRows ? P.hrec : P.grec) + row) + (c - 1), total);
    }
  }
  PAIR_STAMP((KIND == kChainRows ? 2 : KIND == kGbRows ? 1 : 0), 3);
}

}  // namespace agbnp
