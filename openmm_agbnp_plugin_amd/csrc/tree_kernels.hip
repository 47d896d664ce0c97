// __global__ entry points of the overlap-tree stage (see tree_kernels.h for the algorithm).
#include "tree_kernels.h"

#include "pair_kernels.h"
#include "prep_role.h"
#include "row_kernels.h"

namespace agbnp {

// Diagnostic build only (-DAGBNP_STAMPS): shader-clock cycles per phase of k_tree_cavity, summed over
// workgroups (lane 0).  Never compiled into the product library.
#ifdef AGBNP_STAMPS
__device__ unsigned long long g_stamps[16];
__device__ unsigned long long g_stamps_max[16];  // slowest workgroup per phase; [15] = slowest workgroup in total
__device__ unsigned long long g_stamps_slowest[24];  // the slowest workgroup's own phases [0..15] + slot, roots, nodes, atoms
// per work slot: roots, nodes, local atoms, start / end on the 100 MHz wall clock, shader cycles, XCC id, CU id
constexpr int kWgLogSlots = 8192;
__device__ unsigned long long g_wg_log[kWgLogSlots][24];
// per-workgroup sums live in the store (S.stamps); flushed once at the end (no contention inside phases)
#define STAMP_BEGIN()                                   \
  if (tid < 16) S.stamps[tid] = 0;                      \
  __syncthreads();                                      \
  const unsigned long long t_wall0__ = wall_clock64();  \
  const unsigned long long t_cyc0__ = __builtin_readcyclecounter(); \
  unsigned long long t_prev__ = __builtin_readcyclecounter()
#define STAMP(i)                                                   \
  do {                                                             \
    lds_barrier();                                                 \
    if (tid == 0) {                                                \
      const unsigned long long t_now__ = __builtin_readcyclecounter(); \
      S.stamps[i] += t_now__ - t_prev__;                           \
      t_prev__ = t_now__;                                          \
    }                                                              \
  } while (0)
#define STAMP_FLUSH()                                              \
  do {                                                             \
    __syncthreads();                                               \
    if (tid == 0 && slot < kWgLogSlots) {                          \
      unsigned xcc__ = 0, hw__ = 0;                                \
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc__)); \
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw__));   \
      g_wg_log[slot][0] = m; g_wg_log[slot][1] = nnodes; g_wg_log[slot][2] = natoms; \
      g_wg_log[slot][3] = t_wall0__; g_wg_log[slot][4] = wall_clock64();            \
      g_wg_log[slot][5] = __builtin_readcyclecounter() - t_cyc0__;                  \
      g_wg_log[slot][6] = xcc__; g_wg_log[slot][7] = hw__;                          \
      for (int i__ = 0; i__ < 16; i__++) g_wg_log[slot][8 + i__] = S.stamps[i__];  \
    }                                                              \
    if (tid < 16) atomicAdd(&g_stamps[tid], S.stamps[tid]);        \
    if (tid < 16) atomicMax(&g_stamps_max[tid], S.stamps[tid]);    \
    if (tid == 0) {                                                \
      const unsigned long long tot__ = S.stamps[0] + S.stamps[1] + S.stamps[2] + S.stamps[3] + S.stamps[4] + S.stamps[5] + S.stamps[6]; \
      if (tot__ > atomicMax(&g_stamps_max[15], tot__)) {          \
        for (int i__ = 0; i__ < 16; i__++) g_stamps_slowest[i__] = S.stamps[i__]; \
        g_stamps_slowest[16] = slot; g_stamps_slowest[17] = m; g_stamps_slowest[18] = nnodes; g_stamps_slowest[19] = natoms; \
      }                                                            \
    }                                                              \
    __syncthreads();                                               \
  } while (0)
#else
#define STAMP_BEGIN()
#define STAMP(i)
#define STAMP_FLUSH()
#endif
// -DAGBNP_STAMPS_PSEUDO moves the stamps from k_tree_cavity to k_tree_pseudo
#if defined(AGBNP_STAMPS_PSEUDO)
#define CSTAMP_BEGIN()
#define CSTAMP(i)
#define CSTAMP_FLUSH()
#define PSTAMP_BEGIN() STAMP_BEGIN()
#define PSTAMP(i) STAMP(i)
#define PSTAMP_FLUSH() STAMP_FLUSH()
#else
#define CSTAMP_BEGIN() STAMP_BEGIN()
#define CSTAMP(i) STAMP(i)
#define CSTAMP_FLUSH() STAMP_FLUSH()
#define PSTAMP_BEGIN()
#define PSTAMP(i)
#define PSTAMP_FLUSH()
#endif

// Build + cavity passes of one forest = the subtrees of up to kMaxRoots heavy atoms (reference steps A-D of
// platforms/reference/src/ReferenceAGBNPKernels.cpp:293-384, restated in oracle run_cavity()).  Work slot s builds the
// work items of its row rows[kRowStride * s ..]; the packing comes from the previous evaluation's subtree sizes
// (packing_role / dealing_role in pair_kernels.hip).
//
// Scheduling: the launch holds as many workgroups as the device keeps resident (or fewer, if there are fewer forests);
// a workgroup starts on forest blockIdx.x and then takes forests from a device-wide queue until it is empty: one
// returning atomic per forest, issued in the MIDDLE of the forest before it -- after that forest's last global load
// has been consumed (the memory counter is in order: a load issued behind the atomic would have to wait for it, and
// when every workgroup asks at once the queue word serves them one after the other) and long before its result is
// needed.  Forests are ordered largest first, so this is longest-processing-time-first list scheduling over the CUs.  It matters because a
// CU's time is proportional to the nodes it has to build (a workgroup timeline of 1dwc with one forest per
// workgroup: CUs that happened to receive 1000 nodes finished at 68 us, CUs with 500 nodes at 35 us, and the kernel
// lasts as long as its unluckiest CU), and the hardware dispatcher knows nothing about forest sizes.
__device__ __forceinline__ int next_forest(int tid, int* lds_word, int ticket, int grid) {
  // ticket: what this workgroup's atomicAdd on the queue returned (lane 0); every lane gets the same next slot
  // grid: forest workgroups of the launch (each started on the forest of its own number)
  if (tid == 0) *lds_word = grid + ticket;
  lds_barrier();
  const int slot = *lds_word;
  lds_barrier();
  return slot;
}

// Waves per SIMD the compiler is asked to leave room for = workgroups per CU that the variant's LDS footprint allows
// (one wave of a workgroup per SIMD): 5 for the 432-node store (<= 96 VGPRs), 4 for 512 nodes (<= 128), fewer beyond.
constexpr int tree_waves_per_simd(int ncap, int bs) {
  if (bs == 192) return ncap <= 512 ? 4 : 2;  // (three-wave workgroups: five per CU are 15 waves, at most four on a SIMD)
  return bs >= 256 ? (ncap <= 432 ? 5 : (ncap <= 512 ? 4 : 2)) : (ncap <= 192 ? 3 : 2);  // (one-wave workgroups: ~10 per CU for the small store)
}

#ifdef AGBNP_STAMPS  // (diagnostic build: the stamps take 136 bytes of the store's last LDS granule)
constexpr int kPendCap = 4;
#else
constexpr int kPendCap = 16;  // work items that may wait while a forest that outgrew its store is built again in smaller sets
#endif
// SV1: the launch also collects the self volumes of pass 1 (enlarged radii; a diagnostic: agbnp_hip_set_diagnostics)
// FIVE: the five-launch mode's instantiation (k_tree_cavity_five below): positions straight from the caller's array.
// POSQ: FIVE with the positions in an OpenMM context's posq (TreeArgs::posq; row_atoms holds slots)
template <int NCAP, int ACAP, int BS, bool GLOBAL, bool SV1, bool FIVE, bool POSQ = false>
__device__ __forceinline__ void cavity_forests(const TreeArgs& A, const int tree_blocks) {  // tree_blocks: forest workgroups of the launch
  extern __shared__ __align__(16) char smem[];
  TreeStore<NCAP, ACAP> S;
  S.carve(GLOBAL ? (A.scratch + (size_t)blockIdx.x * A.scratch_stride) : smem);
  __shared__ int s_next;  // hand-off word of the work queue (in LDS for every variant)
  // healing (below): work items of the slot's row that wait for another set, their number, and the work slot the set in hand
  // writes its topology to -- in LDS, not in registers: the build keeps ~190 scalar values live as it is
  __shared__ int s_pend[kPendCap + 2];
  int& s_npend = s_pend[kPendCap];
  int& s_out = s_pend[kPendCap + 1];
  const int tid = threadIdx.x;
  // the row of the workgroup's own work slot is requested before anything is waited for (blockIdx.x < slot_cap: the
  // row exists whether or not the slot is in use)
  int first_item = -1, first_atom = 0;
  if ((tid & 63) < kMaxRoots) first_item = A.rows[(size_t)kRowStride * blockIdx.x + (tid & 63)];
  if (FIVE && (tid & 63) < kMaxRoots) first_atom = A.row_atoms[(size_t)kMaxItems * blockIdx.x + (tid & 63)];
  const int nforests = min(A.nforests()[0], A.slot_cap);  // (never above the slots the per-slot arrays hold)
  if (blockIdx.x == 0 && tid == 0) A.cur_nforests()[0] = nforests;
  if ((int)blockIdx.x >= nforests && tid == 0) {  // an idle work slot: k_tree_pseudo's workgroup of the same slot finds nothing to replay
    A.hdr[blockIdx.x].nnodes = 0;
    A.hdr[blockIdx.x].natoms = 0;
  }

  const bool queued = nforests > tree_blocks;  // otherwise every forest has a workgroup of its own
  const int lane = tid & 63;
  for (int slot = blockIdx.x; slot < nforests;) {
    constexpr int kNotAsked = 0x3fffffff;  // (a forest that fails before it asks ends the workgroup's run: the evaluation is void anyway)
    int ticket = kNotAsked;  // lane 0: the queue is asked for the forest after this one ONCE per slot, however many sets it takes
    // the slot's row of work items (written in slot order by the bookkeeping of the previous evaluation: no
    // slot -> forest indirection in front of it): lanes 0..7 of every wave fetch one item each
    // (-1 = no item.  Counting the items with a ballot instead of reading the row's count word out of lane 8 makes
    // k_tree_cavity 1.6 us faster on 1dwc, A/B on one box -- same spill counts, different register allocation.)
    int my_item = first_item, my_atom = first_atom;
    if (slot != (int)blockIdx.x) {  // (a forest from the queue)
      // (the lane number through an opaque move: otherwise the compiler forms the per-lane row address at the kernel's head, keeps
      // it for this path alone and -- at the register limit -- parks it in scratch: 8 bytes of stores per lane and launch)
      int lane_q = lane;
      asm volatile("" : "+v"(lane_q));
      my_item = -1;
      if (lane_q < kMaxRoots) my_item = A.rows[(size_t)kRowStride * slot + lane_q];
      if (FIVE && lane_q < kMaxRoots) my_atom = A.row_atoms[(size_t)kMaxItems * slot + lane_q];
    }
    // HEALING (round 6).  A forest that outgrows its store -- the packing was planned from an earlier geometry's shapes -- used
    // to void the whole evaluation (kStatPackOverflow: withheld, repeated by the host on one subtree per slot, the packing
    // tightened for everybody).  A failed build has touched nothing outside this workgroup's store, so the workgroup simply
    // builds the row's items again in smaller SETS: the first half now, the items left over (s_pend) afterwards, each later set
    // into a SPARE work slot (numbered from max(forests, forest workgroups of the launch) on: no workgroup of either tree launch
    // has such a slot as its own; k_tree_pseudo reaches them through its queue).  Likewise a LONE item that outgrows the store
    // while its subtree is whole or shared two ways is built again as the parts of a four-way share that cover it (part p of 2 =
    // parts p and 3 - p of 4: level2_owner).  The evaluation stays complete; kStatSpareForests tells the bookkeeping to plan anew.
    if (tid == 0) s_npend = 0, s_out = slot;  // (visible behind the barrier in front of the build)
    for (;;) {        // the sets of the slot's row: ONE, unless a forest outgrew its store
    const int m = __popcll(__ballot(my_item >= 0));  // 1..kMaxRoots
    int items[kMaxRoots];
#pragma unroll
    for (int q = 0; q < kMaxRoots; q++) items[q] = __builtin_amdgcn_readlane(my_item, q);
    for (int la = tid; la < ACAP; la += BS) {
      S.at[6][la] = 0.0;
      S.at[7][la] = 0.0;
      S.at[8][la] = 0.0;
      S.at[9][la] = 0.0;
    }
    tree_barrier<NCAP>();
    CSTAMP_BEGIN();
    int nnodes = 0, natoms = 0;
    int rc = build_forest<NCAP, ACAP, BS, FIVE, POSQ>(S, A, tid, my_item, items, m, &nnodes, &natoms, my_atom);
    CSTAMP(0);
    double e_sum = 0.0;
    int npairs = 0;
    constexpr bool want_sv1 = SV1;
    const bool det = A.det != 0;
    // the vdW parameters of pass 2 are requested now and arrive underneath pass 1 (natoms <= ACAP <= BS for the
    // LDS variants: one atom per lane)
    const int hj_mine = (rc == kBuildOk && tid < natoms) ? S.at_gidx[tid] : work_item_root(items[0]);
    const double a_vdw_mine = A.hvat(kHvAVdw, hj_mine), v_vdw_mine = A.hvat(kHvVVdw, hj_mine);
    // ---- pass 1: enlarged radii, nu = +gamma/roffset (reference steps A-B, ReferenceAGBNPKernels.cpp:293-339).
    // The node slots still hold the Gaussians of the build, so only the atom paths and the membership list are
    // laid down before the gather.  Its gradient stays in the local accumulators and leaves with that of pass 2.
    if (rc == kBuildOk && !volume_pass<NCAP, ACAP, BS, true, true>(S, tid, m, nnodes, natoms, want_sv1, &e_sum, &npairs, det))
      rc = kBuildNodeOverflow;  // the membership list does not fit: same protocol as a node overflow
    if (rc != kBuildOk) {
      const int parts0 = work_item_parts(items[0]);
      // ---- healed here if it can be (see above): a forest of several items in two halves, a lone splittable item as the
      // parts of a four-way share
      const bool refine = m == 1 && rc == kBuildNodeOverflow && (A.split_fit & 1) != 0 && parts0 <= 2;
      const int more = (A.split_fit & 2) == 0 ? 0 : (m > 1 ? m - m / 2 : (refine ? 4 / parts0 - 1 : 0));  // items this set hands to s_pend
      const int npend = s_npend;
      if (more > 0 && npend + more <= kPendCap) {
        tree_barrier<NCAP>();  // (everybody has read the count)
        if (m > 1) {
          // (the lane's item again from the store's root words, its root's atom again through h2a: nothing is kept alive
          // across the build for this path)
          const int keep = m / 2;
          int mine = -1;
          if (lane < m) {
            const int pp = S.rt[kRtPart + lane];
            mine = S.rt[kRtHeavy + lane] | ((pp & 0xff) << 24) | (((pp >> 8) - 1) << 26);
          }
          if (tid < 64 && lane >= keep && lane < m) s_pend[npend + lane - keep] = mine;
          my_item = lane < keep ? mine : -1;
          if (FIVE) my_atom = my_item >= 0 ? (POSQ ? A.hslot : A.out.h2a)[work_item_root(my_item)] : 0;
        } else {
          const int root = work_item_root(items[0]), p = work_item_part(items[0]);
          auto part_of_four = [&](int q) { return root | (q << 24) | (3 << 26); };
          if (tid == 0) {
            if (parts0 == 1) {
              s_pend[npend] = part_of_four(1), s_pend[npend + 1] = part_of_four(2), s_pend[npend + 2] = part_of_four(3);
            } else {
              s_pend[npend] = part_of_four(3 - p);
            }
          }
          my_item = lane == 0 ? part_of_four(parts0 == 1 ? 0 : p) : -1;
          if (FIVE) my_atom = lane == 0 ? (POSQ ? A.hslot : A.out.h2a)[root] : 0;
        }
        if (tid == 0) s_npend = npend + more;
        continue;  // the same slot again, with the smaller set (the barrier in front of the build orders the list)
      }
      if (tid == 0) {
        // a forest that does not fit is a packing misprediction (repeat unpacked), and so is a lone work item whose nodes
        // do not fit while its subtree can still be shared among more items (each expands a residue class of the level-2
        // branches; every item holds all level-2 atoms, so an ATOM overflow is not helped by sharing); a lone item of a
        // subtree that is shared four ways already needs the next capacity variant
        // (round 6: what gets here is what could not be healed above: a three-way share, the waiting list full)
        const bool splittable = m == 1 && rc == kBuildNodeOverflow && parts0 < 4 && (A.split_fit & 1) != 0;
        atomicAdd(&A.status[(m > 1 || splittable) ? kStatPackOverflow : (rc == kBuildNodeOverflow ? kStatNodeOverflow : kStatAtomOverflow)], 1);
        if (m > 1) atomicAdd(&A.status[kStatForestOverflow], rc == kBuildNodeOverflow ? 1 : 0x10000);  // (what tightens the packing's assumed capacity)
        if (splittable) {
          atomicMax(&A.status[kStatSplitWanted], parts0);
          // The device reacts by itself (evaluations may be queued behind this one long before a host sees the log): the
          // subtree's shape is recorded as "more than four stores' worth of nodes", so that this evaluation's bookkeeping
          // -- whose fallback packing shares to FIT -- hands the subtree to four work items in the very next evaluation
          // (its level-2 count is left at what the other items of the subtree report, or 0: the conservative side)
          atomicAdd(reinterpret_cast<int*>(&A.sizes[work_item_root(items[0])]), 4 * NCAP);
        }
        A.hdr[s_out].nnodes = 0;
        A.hdr[s_out].natoms = 0;
      }
      tree_barrier<NCAP>();
      break;  // on to the next forest (the evaluation is void: what waits is dropped)
    }
    CSTAMP(1);
    // Take delivery of the prefetched vdW parameters HERE, while nothing else is in flight: the memory counter is in
    // order, so a wait placed after the topology stores below would also wait for every one of them.
    asm volatile("" ::"v"(a_vdw_mine), "v"(v_vdw_mine));
    if (queued && tid == 0 && ticket == kNotAsked) ticket = atomicAdd(&A.status[kStatCavityQueue], 1);  // the forest AFTER this one
    const int out = s_out;  // (== slot unless this is a later set of a healed row)
    if (tid == 0) {
      // level-1 nodes: volume V_i, coefficient +1 (gaussvol.cpp:138-141); once per subtree (its part 0)
      double e1 = e_sum;
      for (int q = 0; q < m; q++) e1 += (S.rt[kRtPart + q] & 0xff) == 0 ? quantize(S.at[5][q] * S.at[4][q], kQEnergy, det) : 0.0;
      // (cavity energies are summed per work slot of the packing: a later set of a healed row adds to its slot's word)
      if (out == slot)
        A.epart[2 * slot] = e1;
      else
        A.epart[2 * slot] += e1;
    }
    if (want_sv1) {  // diagnostics: enlarged-radius self volumes
      for (int la = tid; la < natoms; la += BS) {
        glb_add(&A.hvat(kHvSvLarge, S.at_gidx[la]),
                (la < m && (S.rt[kRtPart + la] & 0xff) == 0) ? S.at[9][la] + quantize(S.at[4][la], kQVol, det) : S.at[9][la]);
        S.at[9][la] = 0.0;
      }
    }
    // ---- topology out for the pseudo-volume pass: the atom paths (8 B/node), the membership list and the local
    // atom list are all a replay needs; fixed stride per work slot, no allocation traffic
    {
      const size_t pool_off = (size_t)out * NCAP, atom_off = (size_t)out * ACAP;
      const unsigned long long* path = reinterpret_cast<const unsigned long long*>(S.nd[6]);
      for (int n = m + tid; n < nnodes; n += BS) A.node_pool[pool_off + n] = path[n];
      for (int la = tid; la < natoms; la += BS) A.atom_pool[atom_off + la] = S.at_gidx[la];
      if (TreeStore<NCAP, ACAP>::kPairGather) {
        const size_t pair_off = (size_t)out * TreeStore<NCAP, ACAP>::PCAP;
        for (int k = tid; k < npairs; k += BS) A.pair_pool[pair_off + k] = S.pairs[k];
      }
      if (tid == 0) {
        SubtreeHeader h;
        h.nnodes = nnodes;
        h.natoms = natoms;
        h.nroots = m;
        h.npairs = npairs;
        for (int q = 0; q < kMaxRoots; q++) h.partners[q] = q < m ? S.rt[kRtCount + q] : 0;
        A.hdr[out] = h;
      }
      // per-subtree shape for the next evaluation's packing and the statistics (zeroed by k_prep; the work items
      // of a shared subtree add their own nodes, part 0 the root and the partner count)
      if (tid < m) {
        const bool first = (S.rt[kRtPart + tid] & 0xff) == 0;
        int* sz = reinterpret_cast<int*>(&A.sizes[S.rt[kRtHeavy + tid]]);  // (no global load behind the stores above)
        atomicAdd(&sz[0], S.rt[kRtNodes + tid] + (first ? 1 : 0));
        if (first) sz[1] = 1 + S.rt[kRtCount + tid];
      }
    }
    CSTAMP(2);
    // switch the local atoms to vdW radii, nu = -gamma/roffset, for pass 2, whose self volumes the Born stage needs
    if (ACAP <= BS) {
      if (tid < natoms) {
        S.at[3][tid] = a_vdw_mine;
        S.at[4][tid] = v_vdw_mine;
        S.at[5][tid] = -S.at[5][tid];
      }
    } else {
      for (int la = tid; la < natoms; la += BS) {
        const int hj = S.at_gidx[la];
        S.at[3][la] = A.hvat(kHvAVdw, hj);
        S.at[4][la] = A.hvat(kHvVVdw, hj);
        S.at[5][la] = -S.at[5][la];
      }
    }
    tree_barrier<NCAP>();  // the topology stores above keep draining underneath pass 2
    CSTAMP(3);

    // ---- pass 2: vdW radii, nu = -gamma/roffset
    volume_pass<NCAP, ACAP, BS, true>(S, tid, m, nnodes, natoms, true, &e_sum, &npairs, det);
    CSTAMP(4);
    root_gradients_from_invariance<NCAP, ACAP, BS>(S, tid, m);
    CSTAMP(5);

    // ---- flush per-atom sums (a root's self volume: its own sphere + every node of its tree)
    // (one row per quantity: an atom's four adds go to four memory channels.  One 32-byte record per atom -- a quarter
    // of the 64-byte atomic requests -- measured 1.3 us SLOWER: same-line adds queue at the memory side.)
    for (int la = tid; la < natoms; la += BS) {
      const int hj = S.at_gidx[la];
      glb_add(&A.hvat(kHvGx, hj), S.at[6][la]);
      glb_add(&A.hvat(kHvGy, hj), S.at[7][la]);
      glb_add(&A.hvat(kHvGz, hj), S.at[8][la]);
      glb_add(&A.hvat(kHvSvVdw, hj),
              (la < m && (S.rt[kRtPart + la] & 0xff) == 0) ? S.at[9][la] + quantize(S.at[4][la], kQVol, det) : S.at[9][la]);
    }
    if (tid == 0) {
      double e2 = e_sum;
      for (int q = 0; q < m; q++) e2 += (S.rt[kRtPart + q] & 0xff) == 0 ? quantize(S.at[5][q] * S.at[4][q], kQEnergy, det) : 0.0;
      if (out == slot)
        A.epart[2 * slot + 1] = e2;
      else
        A.epart[2 * slot + 1] += e2;
    }
    tree_barrier<NCAP>();
    CSTAMP(6);
    CSTAMP_FLUSH();
    const int npend = s_npend;
    if (npend == 0) break;
    // ---- the next set of a healed row: the items that wait (at most kMaxRoots of them), into a spare work slot
    {
      if (tid == 0) s_next = atomicAdd(&A.status[kStatSpareForests], 1);
      lds_barrier();
      const int spare = max(nforests, tree_blocks) + s_next;
      lds_barrier();
      if (spare >= A.slot_cap) {  // no spare slot left (never seen: the pools hold four slots per subtree): the evaluation is void
        if (tid == 0) {
          atomicAdd(&A.status[kStatPackOverflow], 1);
          atomicAdd(&A.status[kStatForestOverflow], 1);
        }
        break;
      }
      const int take = min(npend, kMaxRoots);
      my_item = lane < take ? s_pend[npend - take + lane] : -1;
      if (FIVE) my_atom = my_item >= 0 ? (POSQ ? A.hslot : A.out.h2a)[work_item_root(my_item)] : 0;
      lds_barrier();  // (the list is read: the next set may add to it)
      if (tid == 0) s_npend = npend - take, s_out = spare;
    }
    }
    if (!queued) break;
    slot = __builtin_amdgcn_readfirstlane(next_forest(tid, &s_next, ticket, tree_blocks));  // (wave-uniform by construction: keeps everything derived from it in scalar registers)
  }
}

template <int NCAP, int ACAP, int BS, bool GLOBAL, bool SV1>
__global__ __launch_bounds__(BS, tree_waves_per_simd(NCAP, BS)) void k_tree_cavity(TreeArgs A) {
  cavity_forests<NCAP, ACAP, BS, GLOBAL, SV1, false>(A, (int)gridDim.x);
}

// Five-launch mode (the default for version 1; engine.hip): there is no k_prep launch.  The forest workgroups read the caller's positions
// themselves; the workgroups BEHIND them in the grid -- dispatched when the first forests have left, done long before the last
// ones are -- do k_prep's per-atom work for the launches that follow (prep_role.h) and clear the other parity's tree
// accumulators, subtree shapes and status words for the NEXT evaluation.
static_assert(kPrepHvGx == kHvGx && kPrepHvGx + 3 == kHvSvVdw && kPrepHvSvLarge == kHvSvLarge, "prep_role.h addresses the table's rows by number");
// DEVPAR: the evaluation's set is named by the device's own count (contexts that have been captured into a graph: pair_kernels.h)
// POSQ: the instantiation of agbnp_hip_execute_openmm (round 6): the forest workgroups read the context's posq (double4, float4,
// float4 + correction) at the context's slots; the trailing workgroups read it through the same maps (prep_role.h) and check
// every particle's entry against atomIndex (kStatOrderStale: the context has reordered its atoms, the evaluation is void)
template <int NCAP, int ACAP, int BS, bool DEVPAR = false, bool POSQ = false>
__global__ __launch_bounds__(BS, tree_waves_per_simd(NCAP, BS)) void k_tree_cavity_five(TreeArgs A, PairArgs P, int tree_blocks) {
  if ((int)blockIdx.x >= tree_blocks) {
    const int b = (int)blockIdx.x - tree_blocks;
    if (DEVPAR) rebase_for_parity(P, 0);
    return prep_atoms(P, b * BS + (int)threadIdx.x, b == 0, true);
  }
  if (DEVPAR) rebase_tree_for_parity(A, 0);
  cavity_forests<NCAP, ACAP, BS, false, false, true, POSQ>(A, tree_blocks);
}

// ---- the forces leave with the last tree launch (TreeOutputs) ------------------------------------------------------
__device__ __forceinline__ bool evaluation_void(const int* __restrict__ status) {  // an overflowed evaluation adds nothing
  return (status[kStatNodeOverflow] | status[kStatAtomOverflow] | status[kStatPackOverflow] | status[kStatRowOverflow] | status[kStatOrderStale]) != 0;
}
__device__ __forceinline__ void add_force(const TreeOutputs& O, int atom, double fx, double fy, double fz) {
  if (O.force_fixed) {
    // an OpenMM context's force buffer: 64-bit fixed point, value * 2^32 rounded to nearest, three planes over the padded
    // atom count in the context's atom order, integer atomics (GVolReduceTree.cl:117-119)
    auto to_fixed = [](double f) { return (unsigned long long)(long long)rint(f * 4294967296.0); };
    const int s = O.ctx_slot[atom];
    atomicAdd(&O.force_fixed[s], to_fixed(fx));
    atomicAdd(&O.force_fixed[s + O.padded], to_fixed(fy));
    atomicAdd(&O.force_fixed[s + 2 * O.padded], to_fixed(fz));
  } else {
    glb_add(&O.force[3 * atom], fx);
    glb_add(&O.force[3 * atom + 1], fy);
    glb_add(&O.force[3 * atom + 2], fz);
  }
}
// everything that was complete before the pseudo-volume launch began: cavity gradients (rows of the heavy-atom table), GB
// direct force, chain-rule force; one atom per lane.  (Atomic adds: the forest workgroups add to the same words.)
__device__ __forceinline__ void outputs_role(const TreeArgs& A, int blk, int bs) {
  const TreeOutputs& O = A.out;
  const int i = blk * bs + (int)threadIdx.x;
  if (O.rows_on && i == 0) {  // (as k_outputs does)
    rows_close_evaluation(O.nl_flag, O.nl_nitems, O.row_target, O.gb_rows != 0);
  }
  if (i >= O.n) return;
  const int h = O.a2h[i];
  double fx = 0.0, fy = 0.0, fz = 0.0;
  if (h >= 0) fx = -A.hvat(kHvGx, h), fy = -A.hvat(kHvGy, h), fz = -A.hvat(kHvGz, h);
  const size_t n = (size_t)O.n;
  if (O.rows_on) {  // bw_i G_i + s_i H_i (pair_kernels.hip, k_rows)
    const double bwi = O.bw[i];
    const double4 g = O.grec[i];
    double sh = 0.0;
    double4 hh = make_double4(0.0, 0.0, 0.0, 0.0);
    if (h >= 0) sh = A.hvat(kHvSvVdw, h) * A.hvat(kHvInvVol, h), hh = O.hrec[h];
    fx += O.gb_f[i] + fma(bwi, g.x, sh * hh.x);
    fy += O.gb_f[n + i] + fma(bwi, g.y, sh * hh.y);
    fz += O.gb_f[2 * n + i] + fma(bwi, g.z, sh * hh.z);
  } else {
    fx += O.gb_f[i] + O.db_f[i];
    fy += O.gb_f[n + i] + O.db_f[n + i];
    fz += O.gb_f[2 * n + i] + O.db_f[2 * n + i];
  }
  if (evaluation_void(A.status)) return;
  add_force(O, i, fx, fy, fz);
}

// Replay of a stored forest with vdW radii: reference steps K+L (ReferenceAGBNPKernels.cpp:718-747),
// nu_i = (W_i+U_i)/V_i formed on the fly from the chain-rule sums, gradient only.  The reference does two passes
// (W then U); the pass is linear in nu, so one pass with the sum gives the same gradient.
// PIPE: the instantiation whose replay of QUEUED forests is pipelined (round 4); it keeps the next forest's registers live
// across the volume pass and spills one at the 128-register bound.  Launches whose forests fit one round (TreeOutputs::enabled:
// the engine's own rule for fusing the forces) take the lean one; either handles any number of forests.
template <int NCAP, int ACAP, int BS, bool GLOBAL, bool PIPE = true, bool DEVPAR = false>
__global__ __launch_bounds__(BS, tree_waves_per_simd(NCAP, BS)) void k_tree_pseudo(TreeArgs A) {
  if (DEVPAR) rebase_tree_for_parity(A, 1);  // (five-launch mode, device-side parity: the set of this evaluation; behind the GB launch, see pair_kernels.h)
  extern __shared__ __align__(16) char smem[];
  TreeStore<NCAP, ACAP> S;
  __shared__ int s_next;
  const int tid = threadIdx.x;
  // forest workgroups of the launch; the output workgroups come FIRST in the grid (they only need what earlier launches
  // left, and would otherwise wait for a forest workgroup to leave before they get a slot)
  const int grid = A.out.enabled ? A.out.forest_blocks : (int)gridDim.x;
  const int out_blocks = (int)gridDim.x - grid;
  if ((int)blockIdx.x < out_blocks) return outputs_role(A, blockIdx.x, BS);
  const int block = (int)blockIdx.x - out_blocks;  // number of the forest workgroup
  S.carve_replay(GLOBAL ? (A.scratch + (size_t)block * A.scratch_stride) : smem);
  const bool write_forces = A.out.enabled != 0;
  // forests to replay: those of the packing -- and, when k_tree_cavity healed a forest that had outgrown its store (it built the
  // row's items in several sets), the sets it put into spare work slots, numbered from max(forests, forest workgroups) on: no
  // workgroup's own slot, so the queue is what reaches them (a one-round launch turns into a queued one for that evaluation)
  const int nspare = A.status[kStatSpareForests];
  const int nplanned = A.cur_nforests()[0];  // (consumed when the first forest's loads are on their way)
  const int nforests = nspare > 0 ? max(nplanned, grid) + nspare : nplanned;
  // The workgroup's own work slot needs no test: k_tree_cavity leaves "nothing to replay" in the header of an idle slot,
  // so the forest's topology is requested straight away.
  constexpr bool kPipelined = PIPE && !GLOBAL && NCAP <= 8 * BS && ACAP <= BS && TreeStore<NCAP, ACAP>::kPairGather;
  if constexpr (kPipelined) {
    // Round 4: the replay of QUEUED forests is pipelined (systems with more forests than resident workgroups: the 16.6 k-atom
    // lattice replays ~4 forests per workgroup).  A forest's two dependent trips to memory -- stored topology, then the
    // per-atom parameters of its local atoms -- used to start when the forest before it had flushed its sums; now the
    // topology of the NEXT forest is requested as soon as the current one's has left the registers for LDS (after_pairs,
    // in the middle of the volume pass), and its per-atom parameters behind the current forest's root gradients, in front
    // of the flush.  Lattice: k_tree_pseudo 51.5 -> 48.4-51.2 us (A/B on one box, profiles/r04): the trips were mostly hidden
    // by the other four workgroups of the CU already.  One forest per workgroup (1dwc): the same work as before.
    constexpr int KP = (NCAP + BS - 1) / BS;
    constexpr int kPairWordsP = TreeStore<NCAP, ACAP>::PCAP / 8;  // 16-byte words of the pair list
    constexpr int kPairRegsP = (kPairWordsP + BS - 1) / BS;         // ... per lane
    unsigned long long pw[KP];
    uint4 pair_word[kPairRegsP];
    int hj_pre = 0;
    int h_nnodes = 0, h_natoms = 0, h_m = 0, h_npairs = 0, h_partners[kMaxRoots];  // the forest's header (wave-uniform)
    auto request_topology = [&](int slot_) {  // one round trip: header, paths, membership list, local atoms (capacity-strided
                                              // slots: reading past the forest's own entries is harmless, the values are masked)
      const SubtreeHeader* H = &A.hdr[slot_];  // written by k_tree_cavity's workgroup of the same slot
      const size_t pool_off = (size_t)slot_ * NCAP, atom_off = (size_t)slot_ * ACAP;
#pragma unroll
      for (int k = 0; k < kPairRegsP; k++) {
        pair_word[k] = make_uint4(0, 0, 0, 0);
        if (tid + k * BS < kPairWordsP)
          pair_word[k] = reinterpret_cast<const uint4*>(A.pair_pool + (size_t)slot_ * TreeStore<NCAP, ACAP>::PCAP)[tid + k * BS];
      }
#pragma unroll
      for (int k = 0; k < KP; k++) pw[k] = tid + k * BS < NCAP ? A.node_pool[pool_off + tid + k * BS] : 0ull;
      hj_pre = tid < ACAP ? A.atom_pool[atom_off + tid] : 0;
      h_nnodes = H->nnodes, h_natoms = H->natoms, h_m = H->nroots, h_npairs = H->npairs;
#pragma unroll
      for (int q = 0; q < kMaxRoots; q++) h_partners[q] = H->partners[q];
    };
    double px = 0.0, py = 0.0, pz = 0.0, pa = 0.0, pv = 0.0, pnu = 0.0;  // per-atom parameters of the forest about to be replayed
    int pidx = 0;
    auto request_parameters = [&]() {  // second round trip (needs the local atom list); one atom per lane
      if (tid < h_natoms) {
        const int hj = hj_pre;
        px = A.hvat(kHvX, hj), py = A.hvat(kHvY, hj), pz = A.hvat(kHvZ, hj);
        pa = A.hvat(kHvAVdw, hj), pv = A.hvat(kHvVVdw, hj);
        pnu = A.db_wu[hj] * A.hvat(kHvInvVol, hj);
        pidx = write_forces ? A.out.h2a[hj] : 0;
      }
    };
    const bool queued = nforests > grid;
    const bool det = A.det != 0;
    // whether the evaluation is void (its forces are then withheld) is final since the launches before this one: asked for
    // NOW, with the first forest's topology, not in front of the flush, where the answer is one more cold round trip that
    // every workgroup waits for with its sums ready (round 4)
    const bool is_void = write_forces && evaluation_void(A.status);
    request_topology(block);
    bool have_parameters = false;
    for (int slot = block;;) {  // (`slot`: the forest in hand; the diagnostic stamps log by it)
      PSTAMP_BEGIN();
      const int nnodes = h_nnodes, natoms = h_natoms, m = h_m;
      int npairs = h_npairs;
      if (nnodes <= m) {  // not built (capacity overflow: the host repeats the evaluation), lone atoms only, or an idle slot
        if (!queued) break;
        int ticket = 0;
        if (tid == 0) ticket = atomicAdd(&A.status[kStatPseudoQueue], 1);
        const int slot_ = __builtin_amdgcn_readfirstlane(next_forest(tid, &s_next, ticket, grid));
        if (slot_ >= nforests) break;
        request_topology(slot_);
        have_parameters = false;
        slot = slot_;
        continue;
      }
      if (tid == 0) {
        int run = m;
        for (int q = 0; q < m; q++) {
          S.rt[kRtCount + q] = h_partners[q];
          S.rt[kRtBase + q] = run;
          run += h_partners[q];
        }
      }
      {
        unsigned long long* path = reinterpret_cast<unsigned long long*>(S.nd[6]);
#pragma unroll
        for (int k = 0; k < KP; k++)
          if (tid + k * BS >= m && tid + k * BS < nnodes) path[tid + k * BS] = pw[k];
      }
      if (!have_parameters) request_parameters();
      if (tid < natoms) {
        S.at_gidx[tid] = hj_pre;
        S.at[0][tid] = px;
        S.at[1][tid] = py;
        S.at[2][tid] = pz;
        S.at[3][tid] = pa;
        S.at[4][tid] = pv;
        S.at[5][tid] = pnu;
        S.at[6][tid] = 0.0;
        S.at[7][tid] = 0.0;
        S.at[8][tid] = 0.0;
        // (the self-volume accumulators are idle in this pass: the slot carries the atom's index for the flush)
        S.at[9][tid] = write_forces ? __hiloint2double(0, pidx) : 0.0;
      }
      tree_barrier<NCAP>();
      PSTAMP(0);
      // every global load of this forest has been consumed: ask for the number of the next one (see k_tree_cavity) ...
      int ticket = 0x3fffffff;
      if (queued && tid == 0) ticket = atomicAdd(&A.status[kStatPseudoQueue], 1);
      int next = 0x7fffffff;
      bool next_there = false;
      // ... and, when this forest's topology has left the registers, for the next forest's (the ticket has had the node step
      // to come back)
      auto after_pairs = [&]() {
        if (!queued) return;
        next = __builtin_amdgcn_readfirstlane(next_forest(tid, &s_next, ticket, grid));
        next_there = next < nforests;
        if (next_there) request_topology(next);
      };
      double e_sum = 0.0;
      volume_pass<NCAP, ACAP, BS, false, false>(S, tid, m, nnodes, natoms, false, &e_sum, &npairs, det, pair_word, after_pairs);
      PSTAMP(1);
      root_gradients_from_invariance<NCAP, ACAP, BS>(S, tid, m);
      // the next forest's per-atom parameters: its local atom list is back by now; requested IN FRONT of the flush below
      have_parameters = false;
      if (next_there && h_nnodes > h_m) {
        request_parameters();
        have_parameters = true;
      }
      if (write_forces) {  // force = -gradient, straight into the caller's buffer (nothing of an overflowed evaluation)
        if (!is_void) {
          if (A.out.force_fixed) {
            for (int la = tid; la < natoms; la += BS) add_force(A.out, __double2loint(S.at[9][la]), -S.at[6][la], -S.at[7][la], -S.at[8][la]);
          } else {
            // the caller's [n][3] array: the three components of an atom are one lane each, next to each other, so that they
            // leave in ONE atomic request per atom.  (A lane per atom and one instruction per component sends three requests
            // to the same line one behind the other, and same-line adds queue at the memory side: +1.9 us on 1dwc.)
            for (int k = tid; k < 3 * natoms; k += BS) {
              const int la = k / 3, c = k - 3 * la;
              const double* row = c == 0 ? S.at[6] : (c == 1 ? S.at[7] : S.at[8]);
              glb_add(&A.out.force[3 * (size_t)__double2loint(S.at[9][la]) + c], -row[la]);
            }
          }
        }
      } else {
        for (int la = tid; la < natoms; la += BS) {
          const int hj = S.at_gidx[la];
          glb_add(&A.hvat(kHvGx, hj), S.at[6][la]);
          glb_add(&A.hvat(kHvGy, hj), S.at[7][la]);
          glb_add(&A.hvat(kHvGz, hj), S.at[8][la]);
        }
      }
      tree_barrier<NCAP>();
      PSTAMP(2);
      PSTAMP_FLUSH();
      if (!next_there) break;  // (one forest per workgroup, or the queue is empty)
      slot = next;
    }
    return;
  }
  for (int slot = block;;) {
    int ticket = 0x3fffffff;  // (a forest that asks for no successor ends the workgroup's run)
    do {
    PSTAMP_BEGIN();
    // One round trip to the stored topology: the paths, the membership list and the local atom list are requested
    // together with the header (capacity-strided slots: reading past the forest's own entries is harmless, the
    // values are masked below), a second one to the per-atom parameters.
    const SubtreeHeader* H = &A.hdr[slot];  // written by k_tree_cavity's workgroup of the same slot
    const size_t pool_off = (size_t)slot * NCAP, atom_off = (size_t)slot * ACAP;
    constexpr bool kPrefetch = !GLOBAL && NCAP <= 8 * BS && ACAP <= BS;  // topology requested together with the header
    unsigned long long pw[kPrefetch ? (NCAP + BS - 1) / BS : 1];
    int hj_pre = 0;
    constexpr bool kPairs = TreeStore<NCAP, ACAP>::kPairGather;
    constexpr int kPairWords = kPairs ? TreeStore<NCAP, ACAP>::PCAP / 8 : 1;  // 16-byte words of the pair list
    constexpr int kPairRegs = (kPairWords + BS - 1) / BS;                       // ... per lane
    uint4 pair_word[kPairRegs];
#pragma unroll
    for (int k = 0; k < kPairRegs; k++) {
      pair_word[k] = make_uint4(0, 0, 0, 0);
      if (kPairs && tid + k * BS < kPairWords)
        pair_word[k] = reinterpret_cast<const uint4*>(A.pair_pool + (size_t)slot * TreeStore<NCAP, ACAP>::PCAP)[tid + k * BS];
    }
    if (kPrefetch) {
#pragma unroll
      for (int k = 0; k < (NCAP + BS - 1) / BS; k++) pw[k] = tid + k * BS < NCAP ? A.node_pool[pool_off + tid + k * BS] : 0ull;
      hj_pre = tid < ACAP ? A.atom_pool[atom_off + tid] : 0;
    }
    const int nnodes = H->nnodes, natoms = H->natoms, m = H->nroots;
    int npairs = H->npairs;
    const bool queued = nforests > grid;
    if (nnodes <= m) {  // not built (capacity overflow: the host repeats the evaluation), lone atoms only, or an idle slot
      if (queued && tid == 0) ticket = atomicAdd(&A.status[kStatPseudoQueue], 1);
      break;
    }
    if (tid == 0) {
      int run = m;
      for (int q = 0; q < m; q++) {
        const int c = H->partners[q];
        S.rt[kRtCount + q] = c;
        S.rt[kRtBase + q] = run;
        run += c;
      }
    }
    {
      unsigned long long* path = reinterpret_cast<unsigned long long*>(S.nd[6]);
      if (!kPrefetch) {
        for (int n = m + tid; n < nnodes; n += BS) path[n] = A.node_pool[pool_off + n];
      } else {
#pragma unroll
        for (int k = 0; k < (NCAP + BS - 1) / BS; k++)
          if (tid + k * BS >= m && tid + k * BS < nnodes) path[tid + k * BS] = pw[k];
      }
    }
    for (int la = tid; la < natoms; la += BS) {
      const int hj = kPrefetch ? hj_pre : A.atom_pool[atom_off + la];
      S.at_gidx[la] = hj;
      S.at[0][la] = A.hvat(kHvX, hj);
      S.at[1][la] = A.hvat(kHvY, hj);
      S.at[2][la] = A.hvat(kHvZ, hj);
      S.at[3][la] = A.hvat(kHvAVdw, hj);
      S.at[4][la] = A.hvat(kHvVVdw, hj);
      S.at[5][la] = A.db_wu[hj] * A.hvat(kHvInvVol, hj);
      S.at[6][la] = 0.0;
      S.at[7][la] = 0.0;
      S.at[8][la] = 0.0;
      // (the self-volume accumulators are idle in this pass: the slot carries the atom's index for the flush)
      S.at[9][la] = write_forces ? __hiloint2double(0, A.out.h2a[hj]) : 0.0;
    }
    tree_barrier<NCAP>();
    PSTAMP(0);
    // every global load of this forest has been consumed: ask for the next one (see k_tree_cavity)
    if (queued && tid == 0) ticket = atomicAdd(&A.status[kStatPseudoQueue], 1);
    double e_sum = 0.0;
    volume_pass<NCAP, ACAP, BS, false>(S, tid, m, nnodes, natoms, false, &e_sum, &npairs, A.det != 0, kPairs ? pair_word : nullptr);
    PSTAMP(1);
    root_gradients_from_invariance<NCAP, ACAP, BS>(S, tid, m);
    if (write_forces) {  // force = -gradient, straight into the caller's buffer (nothing of an overflowed evaluation)
      if (!evaluation_void(A.status)) {
        if (A.out.force_fixed) {
          for (int la = tid; la < natoms; la += BS) add_force(A.out, __double2loint(S.at[9][la]), -S.at[6][la], -S.at[7][la], -S.at[8][la]);
        } else {
          // the caller's [n][3] array: the three components of an atom are one lane each, next to each other, so that they
          // leave in ONE atomic request per atom.  (A lane per atom and one instruction per component sends three requests
          // to the same line one behind the other, and same-line adds queue at the memory side: +1.9 us on 1dwc.)
          for (int k = tid; k < 3 * natoms; k += BS) {
            const int la = k / 3, c = k - 3 * la;
            const double* row = c == 0 ? S.at[6] : (c == 1 ? S.at[7] : S.at[8]);
            glb_add(&A.out.force[3 * (size_t)__double2loint(S.at[9][la]) + c], -row[la]);
          }
        }
      }
    } else {
      for (int la = tid; la < natoms; la += BS) {
        const int hj = S.at_gidx[la];
        glb_add(&A.hvat(kHvGx, hj), S.at[6][la]);
        glb_add(&A.hvat(kHvGy, hj), S.at[7][la]);
        glb_add(&A.hvat(kHvGz, hj), S.at[8][la]);
      }
    }
    tree_barrier<NCAP>();
    PSTAMP(2);
    PSTAMP_FLUSH();
    } while (false);
    if (nforests <= grid) break;  // every forest has a workgroup of its own
    slot = __builtin_amdgcn_readfirstlane(next_forest(tid, &s_next, ticket, grid));  // (wave-uniform by construction: keeps everything derived from it in scalar registers)
    if (slot >= nforests) break;
  }
}

#ifdef AGBNP_STAMPS
extern "C" void agbnp_debug_stamps(unsigned long long* out, int reset) {
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 16);
  if (reset) {
    unsigned long long z[16] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), z, sizeof(z));
  }
}
extern "C" void agbnp_debug_wg_log(unsigned long long* out, int slots) {
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wg_log), sizeof(unsigned long long) * 24 * (size_t)(slots < kWgLogSlots ? slots : kWgLogSlots));
}
extern "C" void agbnp_debug_stamps_slowest(unsigned long long* out) {
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps_slowest), sizeof(unsigned long long) * 24);
}
extern "C" void agbnp_debug_stamps_max(unsigned long long* out, int reset) {
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps_max), sizeof(unsigned long long) * 16);
  if (reset) {
    unsigned long long z[16] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_stamps_max), z, sizeof(z));
  }
}
#endif

// ---- host-side launchers -------------------------------------------------------------------------
// variant: 0 = (432 nodes, 64 local atoms) in LDS, five workgroups per CU; 1 = (512, 64), four; 2 = (1024, 128), two;
// 3 = (2048, 256), one; 4 = (32768, 256) in a per-workgroup HBM slab
constexpr int kGlobalNodeCap = 32768;
constexpr int kGlobalAtomCap = 256;  // one byte per atom in the path words
constexpr int kBS = AGBNP_TREE_BLOCK;  // lanes per subtree (compile-time knob, default kTreeBlock)
static_assert(kBS <= kTreeBlock && kBS % 64 == 0, "tree block size");
// LDS is handed out in granules of 1280 bytes on gfx950: five workgroups per CU need <= 25 granules each
static_assert((TreeStore<432, 64>::kBytes + 16 + sizeof(int) * kPendCap + 1279) / 1280 * 5 <= 128, "five build workgroups per CU");
static_assert((TreeStore<512, 64>::kBytes + 16 + sizeof(int) * kPendCap + 1279) / 1280 * 4 <= 128, "four build workgroups per CU");
// Experimental build (-DAGBNP_TREE_BLOCK=64): one-wave workgroups, one subtree (or part of one) each, in a small store
#if AGBNP_TREE_BLOCK == 64
#define AGBNP_SMALL_STORE 192, 48
constexpr int kSmallNodes = 192, kSmallAtoms = 48;
#else
#define AGBNP_SMALL_STORE 432, 64
constexpr int kSmallNodes = 432, kSmallAtoms = 64;
#endif
size_t tree_variant_lds_bytes(int variant);
int tree_variant_node_cap(int variant);
// workgroups of the build kernel that a CU holds: LDS granules of 1280 B (128 per CU), 32 waves, the register budget
int tree_variant_wgs_per_cu(int variant) {
  const size_t bytes = tree_variant_lds_bytes(variant) + 16 + sizeof(int) * kPendCap;  // (+ the kernels' static LDS)
  if (tree_variant_lds_bytes(variant) == 0) return 1;
  const int by_lds = (int)(128 / ((bytes + 1279) / 1280));
  const int waves = kBS / 64;
  const int by_regs = 4 * tree_waves_per_simd(tree_variant_node_cap(variant), kBS) / waves;
  return std::max(1, std::min(std::min(by_lds, by_regs), 32 / waves));
}

size_t tree_variant_lds_bytes(int variant) {
  switch (variant) {
    case 0: return TreeStore<AGBNP_SMALL_STORE>::kBytes;
    case 1: return TreeStore<512, 64>::kBytes;
    case 2: return TreeStore<1024, 128>::kBytes;
    case 3: return TreeStore<2048, 256>::kBytes;
    default: return 0;
  }
}
size_t tree_variant_scratch_bytes(int variant) {
  return variant == 4 ? ((TreeStore<kGlobalNodeCap, kGlobalAtomCap>::kBytes + 255) / 256) * 256 : 0;
}
int tree_variant_node_cap(int variant) {
  static const int caps[5] = {kSmallNodes, 512, 1024, 2048, kGlobalNodeCap};
  return caps[variant];
}

int tree_variant_atom_cap(int variant) {
  static const int caps[5] = {kSmallAtoms, 64, 128, 256, kGlobalAtomCap};
  return caps[variant];
}

template <class K>
static hipError_t launch_tree(K kernel, int grid, size_t lds, const TreeArgs& A, hipStream_t st) {
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(kernel, dim3(grid), dim3(kBS), lds, st, A);
  return hipGetLastError();
}

// slots: workgroups to launch = min(work slots that may be planned, workgroups the device keeps resident)
hipError_t launch_tree_cavity(int variant, int global_grid, int slots, const TreeArgs& A, hipStream_t st) {
  if (A.nh <= 0) return hipSuccess;
#ifdef AGBNP_STAMPS  // diagnostic build only: time the largest subtrees alone (results are incomplete)
  if (const char* env = getenv("AGBNP_DIAG_TREE_GRID"))
    return launch_tree(k_tree_cavity<AGBNP_SMALL_STORE, kBS, false, false>, std::min(A.nh, atoi(env)), TreeStore<AGBNP_SMALL_STORE>::kBytes, A, st);
#endif
  const bool sv1 = A.want_sv_large != 0;  // (the diagnostic self volumes of pass 1: a kernel of their own)
#define AGBNP_CAVITY(...) (sv1 ? launch_tree(k_tree_cavity<__VA_ARGS__, true>, grid, lds, A, st) : launch_tree(k_tree_cavity<__VA_ARGS__, false>, grid, lds, A, st))
  int grid = slots;
  size_t lds = 0;
  switch (variant) {
    case 0: lds = TreeStore<AGBNP_SMALL_STORE>::kBytes; return AGBNP_CAVITY(AGBNP_SMALL_STORE, kBS, false);
    case 1: lds = TreeStore<512, 64>::kBytes; return AGBNP_CAVITY(512, 64, kBS, false);
    case 2: lds = TreeStore<1024, 128>::kBytes; return AGBNP_CAVITY(1024, 128, kBS, false);
    case 3: lds = TreeStore<2048, 256>::kBytes; return AGBNP_CAVITY(2048, 256, kBS, false);
    default: grid = global_grid < A.nh ? global_grid : A.nh; return AGBNP_CAVITY(kGlobalNodeCap, kGlobalAtomCap, kBS, true);
  }
#undef AGBNP_CAVITY
}

template <class K>
static hipError_t launch_five(K kernel, dim3 grid, size_t lds, const TreeArgs& A, const PairArgs& P, int slots, hipStream_t st) {
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(kernel, grid, dim3(kBS), lds, st, A, P, slots);
  return hipGetLastError();
}

// (round 6: every LDS variant -- rounds 5 left the mode for good at the first store beyond (512, 64))
hipError_t launch_tree_cavity_five(int variant, int slots, const TreeArgs& A, const PairArgs& P, hipStream_t st) {
  if (A.nh <= 0 || variant > 3) return hipErrorInvalidValue;  // (the engine leaves the mode before it gets here)
  const int work = std::max(std::max(P.n, P.nslots), (int)kStatEvalWords);
  const int prep_blocks = (work + kBS - 1) / kBS;
  const bool dev = A.five == 2, posq = A.posq != nullptr;
  const dim3 grid(slots + prep_blocks);
#define AGBNP_FIVE(NC, AC)                                                                                              \
  do {                                                                                                                 \
    const size_t lds = TreeStore<NC, AC>::kBytes;                                                                      \
    if (dev && posq) return launch_five(k_tree_cavity_five<NC, AC, kBS, true, true>, grid, lds, A, P, slots, st);      \
    if (dev) return launch_five(k_tree_cavity_five<NC, AC, kBS, true, false>, grid, lds, A, P, slots, st);             \
    if (posq) return launch_five(k_tree_cavity_five<NC, AC, kBS, false, true>, grid, lds, A, P, slots, st);            \
    return launch_five(k_tree_cavity_five<NC, AC, kBS, false, false>, grid, lds, A, P, slots, st);                     \
  } while (0)
  switch (variant) {
    case 0: AGBNP_FIVE(kSmallNodes, kSmallAtoms);
    case 1: AGBNP_FIVE(512, 64);
    case 2: AGBNP_FIVE(1024, 128);
    default: AGBNP_FIVE(2048, 256);
  }
#undef AGBNP_FIVE
}

hipError_t launch_tree_pseudo(int variant, int global_grid, int slots, const TreeArgs& A0, hipStream_t st) {
  if (A0.nh <= 0) return hipSuccess;
  // the forces leave with this launch (TreeOutputs): its forest workgroups are followed by one lane per atom
  TreeArgs A = A0;
  A.out.forest_blocks = variant <= 3 ? slots : (global_grid < A.nh ? global_grid : A.nh);
  const int grid = A.out.forest_blocks + (A.out.enabled ? (A.out.n + kBS - 1) / kBS : 0);
  switch (variant) {
    case 0:
      if (A.five == 2 && A.out.enabled) return launch_tree(k_tree_pseudo<AGBNP_SMALL_STORE, kBS, false, false, true>, grid, TreeStore<AGBNP_SMALL_STORE>::kReplayBytes, A, st);
      if (A.five == 2) return launch_tree(k_tree_pseudo<AGBNP_SMALL_STORE, kBS, false, true, true>, grid, TreeStore<AGBNP_SMALL_STORE>::kReplayBytes, A, st);
      if (A.out.enabled) return launch_tree(k_tree_pseudo<AGBNP_SMALL_STORE, kBS, false, false>, grid, TreeStore<AGBNP_SMALL_STORE>::kReplayBytes, A, st);
      return launch_tree(k_tree_pseudo<AGBNP_SMALL_STORE, kBS, false>, grid, TreeStore<AGBNP_SMALL_STORE>::kReplayBytes, A, st);
    case 1:
      if (A.five == 2 && A.out.enabled) return launch_tree(k_tree_pseudo<512, 64, kBS, false, false, true>, grid, TreeStore<512, 64>::kReplayBytes, A, st);
      if (A.five == 2) return launch_tree(k_tree_pseudo<512, 64, kBS, false, true, true>, grid, TreeStore<512, 64>::kReplayBytes, A, st);
      if (A.out.enabled) return launch_tree(k_tree_pseudo<512, 64, kBS, false, false>, grid, TreeStore<512, 64>::kReplayBytes, A, st);
      return launch_tree(k_tree_pseudo<512, 64, kBS, false>, grid, TreeStore<512, 64>::kReplayBytes, A, st);
    case 2:
      if (A.five == 2) return launch_tree(k_tree_pseudo<1024, 128, kBS, false, true, true>, grid, TreeStore<1024, 128>::kReplayBytes, A, st);
      return launch_tree(k_tree_pseudo<1024, 128, kBS, false>, grid, TreeStore<1024, 128>::kReplayBytes, A, st);
    case 3:
      if (A.five == 2) return launch_tree(k_tree_pseudo<2048, 256, kBS, false, true, true>, grid, TreeStore<2048, 256>::kReplayBytes, A, st);
      return launch_tree(k_tree_pseudo<2048, 256, kBS, false>, grid, TreeStore<2048, 256>::kReplayBytes, A, st);
    default: return launch_tree(k_tree_pseudo<kGlobalNodeCap, kGlobalAtomCap, kBS, true>, grid, 0, A, st);
  }
}

}  // namespace agbnp
