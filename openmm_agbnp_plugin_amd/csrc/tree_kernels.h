// GaussVol overlap-tree kernels for gfx950 (MI355X).
//
// One workgroup (BS = 192 lanes: five of them on a CU are at most four waves per SIMD, so the build keeps its ~124
// vector registers without spilling) owns a FOREST -- the complete overlap subtrees rooted at up to eight heavy atoms
// (or one residue class of the level-2 branches of a big subtree) -- and keeps it in LDS for its whole life: build
// (large radii) -> volume pass 1 on the build's Gaussians -> topology out -> vdW radii -> volume pass 2, all inside
// one launch.  Only per-atom sums (gradients, self volumes), one energy pair per forest and the stored topology
// (8-byte atom path per node + 2-byte (atom, node) memberships) leave the CU.
//
// What is computed (reference restated in oracle/agbnp_oracle.cpp):
//   build      : gaussvol/gaussvol.cpp:197-250 (child scan over YOUNGER siblings), :154-192 (children sorted
//                by switched volume, descending), :376-397 (tree growth), MAX_ORDER=8 stop (:211)
//   merge      : gaussvol/gaussvol.cpp:60-93 (ogauss_alpha), :18-41 (pol_switchfunc)
//   bottom-up  : gaussvol/gaussvol.cpp:400-487 (compute_volume_underslot2_r), only the quantities the
//                Reference kernel consumes: self volumes, energy, energy gradient
//   rescan     : gaussvol/gaussvol.cpp:254-327 (volumes) and :330-372 (gammas only)
//
// Design differences from the reference (same numbers, different machine):
//   * breadth-first node order inside a forest (levels contiguous over all its trees) so that the expansion is
//     lane-parallel over a level; the reference is depth-first recursive.
//   * level-synchronous expansion: every (node, younger sibling) pair of a level is one task = one lane;
//     tasks -> switched volumes in LDS -> per-node child counts (prefix sum) -> per-task rank inside its
//     node's kept set -> children written directly at their sorted position.
//   * no bottom-up recursion: the reference's (psi, F, P) recursion (gaussvol.cpp:400-487) is the chain rule for
//     E = sum_n c_n gamma_n s(G_n) G_n, where G_n is the UNswitched overlap of the atoms of node n.  G_n is a
//     product of Gaussians, so dG_n/dr_m = -2 a_m (r_m - c_n) G_n for every atom m of the node, and
//       dE/dr_m   = sum_{n contains m} c_n gamma_n sfp_n * (-2 a_m)(r_m - c_n) G_n
//       selfvol_m = sum_{n contains m} c_n s(G_n) G_n                 (c_n = +-1/level)
//   * no top-down rescan: every node carries its atom list (path word), so a pass evaluates every node directly from
//     its atoms (telescoped chain of pairwise merges, see volume_pass) -- no level ordering, no barriers between
//     levels -- then turns (G_n, gamma_n, level) into the two scalars coef_n, w_n and gathers them over a list of
//     (atom, node) memberships sorted by atom (equal pieces per lane, one LDS FP64 add per run of an atom).
//     A root's gradient follows from translation invariance (the gradients of a tree sum to 0).
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "agbnp_common.h"
#include "device_math.h"

namespace agbnp {

// Rows of the heavy-atom table: ONE device allocation [kHvRows][hstride] of doubles, heavy index.  The tree kernels
// address it as base + (row * hstride + h): one 64-bit base in scalar registers instead of fifteen pointers (the
// build kernel keeps ~100 scalar values live as it is; every pointer less is two fewer spilled).
enum HeavyRow {
  kHvX = 0, kHvY, kHvZ,        // positions (written by k_prep)
  kHvALarge, kHvVLarge,        // Gaussian exponent / volume, enlarged radii
  kHvAVdw, kHvVVdw,            // ... vdW radii
  kHvGam,                      // gamma / roffset (pass 1 uses +gam, pass 2 -gam)
  kHvInvVol,                   // 1 / (4 pi R^3 / 3), vdW radius
  kHvSvLarge,                  // self volumes with the enlarged radii (diagnostic)
  kHvGx, kHvGy, kHvGz,         // tree sums: gradient dE/dr ...
  kHvSvVdw,                    // ... and self volumes (vdW radii); the four rows are consecutive (row kHvGx + c)
  kHvRows
};

// Where the forces of an AGBNP1 evaluation leave (version 1): the pseudo-volume launch is the last one that contributes to
// them, so it adds them to the caller's buffer itself -- its forest workgroups their own gradients, a few workgroups at
// the end of its grid everything that was complete before it started -- and the evaluation needs no output launch.
struct TreeOutputs {
  int enabled;                 // 0: k_tree_pseudo adds into the heavy-atom table's gradient rows and k_outputs follows
  int n;                       // particles
  int forest_blocks;           // workgroups of the launch that replay forests (the output workgroups follow them)
  const int *a2h, *h2a;        // [n] atom -> heavy index or -1, [nh] back
  double* force;               // [3n] the caller's FP64 forces (added to) ...
  unsigned long long* force_fixed;  // ... or an OpenMM context's 2^32 fixed-point planes [3 * padded] at ctx_slot[atom]
  int padded;
  const int* ctx_slot;
  const double *gb_f, *db_f;   // [3][n] GB direct force, chain-rule force (tile form), rows x | y | z
  int rows_on;                 // chain-rule force of the row form instead: bw_i G_i + s_i H_i
  const double* bw;            // [n]
  const double4 *grec, *hrec;  // [n], [nh]
  const int* nl_nitems;        // row form: work items laid down by a rebuild, and what the slice length is tuned against (rows_close_evaluation)
  int row_target, gb_rows;
  int* nl_flag;                // row form: [0] is cleared (this evaluation's neighbour lists are up to date)
};

struct PairArgs;  // (pair_kernels.h)

struct TreeArgs {
  TreeOutputs out;
  int nh;                      // heavy atoms
  unsigned hstride;            // row stride of the heavy-atom table
  double* hv;                  // [kHvRows][hstride]
  int want_sv_large;           // collect the enlarged-radius self volumes too (diagnostic: extra HBM atomics)
  int det;                     // deterministic mode: order-dependent sums only take quantized terms (device_math.h)
  int split_fit;               // bit 0: a lone work item that outgrows the store asks for its subtree to be shared (kStatSplitWanted);
                               // bit 1: a forest that outgrows its store is healed inside the launch (cavity_forests, tree_kernels.hip)
  const int* rows;             // [kRowStride * slots] work items: item k of work slot s at kRowStride * s + k, their number at + kMaxItems
  const int* packing;          // [slot_cap + 1] forest_start (bookkeeping's own),
                               // [slot_cap + 1] work slots in use (rewritten for the NEXT evaluation while this one's
                               // pair stages run), [slot_cap + 2] the copy k_tree_cavity takes for THIS evaluation
  int slot_cap;
  const unsigned long long* nbmask;  // [nhb][nhb * 64] level-2 neighbour masks from the k_prep launch (agbnp_common.h)
  int nhb;                     // blocks of 64 heavy atoms
  // pass 3 (pseudo-volume): nu_i = (W_i + U_i) / V_i (ReferenceAGBNPKernels.cpp:718-722,738-742), formed on the fly
  const double* db_wu;         // [nh] W+U per heavy atom
  double rcut2;                // conservative squared cutoff of the 2-body overlap search
  double* epart;               // [2 * slots] cavity energies E1, E2 per work slot
  SubtreeHeader* hdr;          // [slots] by work slot
  int2* sizes;                 // [nh] {nodes, local atoms} of every subtree, compact copy for the bookkeeping block
  unsigned long long* node_pool;  // [slots][NCAP] atom path of every node (all a replay needs), fixed stride per slot
  int* atom_pool;              // [slots][ACAP] local atom -> heavy index
  unsigned short* pair_pool;   // [slots][4 NCAP] (atom, node) membership pairs sorted by atom (variants up to 512 nodes)
  int* status;                 // [kStatTotalWords]
  char* scratch;               // GLOBAL variant: per-workgroup slab
  size_t scratch_stride;
  // five-launch mode (engine.hip): the tree reads the caller's positions itself -- no k_prep has put them into the table
  const double* pos;           // [3n] the caller's positions
  const int* row_atoms;        // [kMaxItems * slots] atom of the root of item k of work slot s (beside `rows`)
  // ... or an OpenMM context's posq (agbnp_hip_execute_openmm in the five-launch mode; the POSQ instantiations,
  // OpenCLAGBNPKernels.cpp:541-556 for the conventions): positions by the context's SLOT -- row_atoms then holds the slot of
  // every item's root (engine.hip keeps it so), a candidate's slot comes through hslot
  const void* posq;            // [padded] double4, or float4 ...
  const float4* posq_corr;     // ... + float4 correction (mixed precision), or null
  int posq_double;
  const int* hslot;            // [nh] heavy index -> context slot
  // ... and works on the set of {heavy-atom table, subtree shapes, status block} that the parity of the device's evaluation
  // counter names (PairArgs::epoch, pair_kernels.h); hv / sizes / status are those of set 0
  int five;
  const int* epoch;
  size_t table_doubles, sizes_stride;

  __device__ __forceinline__ double& hvat(int row, int h) const { return hv[(unsigned)row * hstride + (unsigned)h]; }
  __device__ __forceinline__ const int* forest_start() const { return packing; }
  __device__ __forceinline__ const int* nforests() const { return packing + slot_cap + 1; }
  __device__ __forceinline__ int* cur_nforests() const { return const_cast<int*>(packing) + slot_cap + 2; }
};

// (see rebase_for_parity, pair_kernels.h)
__device__ __forceinline__ void rebase_tree_for_parity(TreeArgs& A, int after_role) {
  if (A.five != 2) return;
  const int par = (A.epoch[0] + after_role) & 1;
  A.hv += (size_t)par * A.table_doubles;
  A.sizes += (size_t)par * A.sizes_stride;
  A.status += 16 * par;
}

constexpr int kTreeBlock = 256;  // lanes per subtree workgroup (upper bound of the BS template parameter)
// A workgroup builds a FOREST: up to kMaxRoots subtrees (of different heavy atoms) side by side in one store, level
// by level.  Every phase of the expansion and of the volume passes is bound by latency, not by work, so a forest of a
// few hundred nodes costs little more than one subtree of a hundred.
constexpr int kMaxRoots = kMaxItems;
constexpr int kRootWords = 6 * kMaxRoots + 8;
enum RootWord {
  kRtHeavy = 0,              // heavy index of the root
  kRtCount = kMaxRoots,      // level-2 partners
  kRtBase = 2 * kMaxRoots,   // first partner slot
  kRtNodes = 3 * kMaxRoots,  // nodes below the root that this workgroup owns
  kRtPart = 4 * kMaxRoots,   // part | parts << 8: a big subtree is shared by `parts` work items; item `part` expands the
                             // level-2 nodes whose rank is congruent to it (branches under different level-2 nodes are
                             // independent) and treats the other level-2 nodes as siblings only
  kRtOff = 5 * kMaxRoots,    // first (root, block) pair of the root in the forest's list of neighbour-mask words
  kRtNum = 6 * kMaxRoots
};
// work item = heavy index | part << 24 | (parts - 1) << 26
__host__ __device__ inline int work_item_root(int e) { return e & 0xffffff; }
__host__ __device__ inline int work_item_part(int e) { return (e >> 24) & 3; }
__host__ __device__ inline int work_item_parts(int e) { return ((e >> 26) & 3) + 1; }
// Which work item of a subtree shared `parts` ways expands the level-2 node of sorted rank r: the parts in serpentine order
// (0, 1, .., p-1, p-1, .., 1, 0, 0, 1, ..).  A level-2 node pairs with its YOUNGER siblings only, so the branch under rank r
// shrinks quickly with r; plain residue classes (r mod p) give item 0 the largest branch of every group of p, the serpentine
// evens that out (what matters when a subtree is shared so that its items FIT the store, round 4).
__host__ __device__ inline int level2_owner(int r, int parts) {
  const int m2 = r % (2 * parts);
  return m2 < parts ? m2 : 2 * parts - 1 - m2;
}


// ---- LDS / scratch carve-out -----------------------------------------------------------------
template <int NCAP, int ACAP>
struct TreeStore {
  static constexpr int TCAP = 8 * ACAP < NCAP ? 8 * ACAP : NCAP;  // tasks per expansion batch (tmap bytes live in cand_vol, volumes in nd[6])
  static_assert(TCAP <= NCAP && TCAP <= 8 * ACAP, "task volumes are staged in the spare node slot, the byte map in cand_vol");
  static constexpr int kMaskWords = (TCAP + 63) / 64;
  // Node rows.  Stores for up to 64 local atoms keep SIX: the node's gamma sum is not stored (a volume pass adds it up
  // along the node's atoms) and the gather weight w_n takes the place of the unswitched volume; the larger variants keep
  // a seventh row because their gather folds across waves through the volume row.
  static constexpr int kRows = ACAP <= 64 ? 6 : 7;
  static_assert(2 * ACAP <= NCAP, "level-2 staging slots must not collide with the level-2 nodes");
  double* nd[7];   // node slots: 0-2 centre, 3 exponent, 4 unswitched volume, 6 scratch (task volumes, then atom paths);
                   // [5] exists with seven rows only.  After the node step of a volume pass: 3 coef_n, wrow w_n
  double* wrow;    // gather weights w_n: nd[4] (six rows) or nd[5] (seven)
  double* at[10];  // local atoms: 0-2 centre, 3 exponent, 4 volume, 5 gamma, 6-8 gradient acc, 9 self-volume acc
  double* cand_vol;  // [ACAP] level-2 candidate volumes; reused as the task->node byte map of a batch
  double* misc;      // [8]: per-wave partial sums of a volume pass
  int* cand_idx;     // [ACAP]
  int* at_gidx;      // [ACAP]
  int* lvl;          // [12]
  int* ctl;          // [12] workgroup control words (counters, scan partials)
  int* rt;           // [kRootWords] the roots of the forest in this store: heavy index, partner count, first partner
                     // slot, node count (kMaxRoots each), then the number of roots
  unsigned short *nla, *npar, *ncs, *ncc;  // [NCAP]
  unsigned short *tstart, *cbase;          // [kTreeBlock + 2] per-batch task start / child base
  unsigned long long* kmask;               // [kMaskWords] per 64 tasks of a batch: which ones survive the switch
  // (atom, node) membership pairs sorted by atom, one 16-bit word each (atom << 9 | node): the gather of a volume
  // pass walks this list instead of testing every node against every atom.  Built once per subtree after the
  // build, when the four 16-bit topology arrays are dead: the list lives in their place.
#ifdef AGBNP_NO_PAIR_GATHER  // timing experiment: the atomics-free (atom, slice) gather for every variant
  static constexpr bool kPairGather = false;
#else
  static constexpr bool kPairGather = NCAP <= 512 && ACAP <= 128;
#endif
  static constexpr int PCAP = 4 * NCAP;
  unsigned short* pairs;  // [PCAP]
  int* pcnt;              // [ACAP] per-atom counts / fill cursors while the list is built (overlays cand_vol)
#ifdef AGBNP_STAMPS
  unsigned long long* stamps;  // [16] diagnostic build only
  static constexpr size_t kStampBytes = 16 * sizeof(unsigned long long) + 8;
#else
  static constexpr size_t kStampBytes = 0;
#endif

  static constexpr size_t kBytes = sizeof(double) * (kRows * (size_t)NCAP + 10 * (size_t)ACAP + ACAP + 8) +
                                   sizeof(int) * (2 * (size_t)ACAP + 24 + kRootWords) +
                                   sizeof(unsigned short) * (4 * (size_t)NCAP + 2 * (kTreeBlock + 2)) + 8 +
                                   sizeof(unsigned long long) * kMaskWords + kStampBytes;

  // A replay (volume pass from stored atom paths) needs the node rows, the local atom table and the per-wave partial
  // sums, but none of the build's bookkeeping arrays: 3/4 of the full footprint.
  static constexpr int kReplayRows = kRows;
  static constexpr size_t kReplayBytes = sizeof(double) * (kReplayRows * (size_t)NCAP + 10 * (size_t)ACAP + 8) +
                                         sizeof(int) * ((size_t)ACAP + kRootWords) + kStampBytes;
  static_assert(!kPairGather || sizeof(unsigned short) * PCAP <= sizeof(double) * NCAP, "pair list fits the path row");
  __device__ __forceinline__ void carve_rows(double* d) {
    for (int k = 0; k < 5; k++) nd[k] = d + (size_t)k * NCAP;
    nd[5] = kRows == 7 ? d + 5 * (size_t)NCAP : nullptr;
    nd[6] = d + (size_t)(kRows - 1) * NCAP;
    wrow = kRows == 7 ? nd[5] : nd[4];
  }
  __device__ __forceinline__ void carve_replay(char* base) {
    double* d = reinterpret_cast<double*>(base);
    carve_rows(d);
    d += kReplayRows * (size_t)NCAP;
    for (int k = 0; k < 10; k++) at[k] = d + (size_t)k * ACAP;
    d += 10 * (size_t)ACAP;
    misc = d;
    d += 8;
#ifdef AGBNP_STAMPS
    stamps = reinterpret_cast<unsigned long long*>(d);
    d += 17;
#endif
    at_gidx = reinterpret_cast<int*>(d);
    rt = at_gidx + ACAP;
    pairs = reinterpret_cast<unsigned short*>(nd[6]);  // a replay drops the pair list over the atom paths once the node step is done
    pcnt = nullptr;
    cand_vol = nullptr;
    cand_idx = nullptr;
    lvl = ctl = nullptr;
    nla = npar = ncs = ncc = tstart = cbase = nullptr;
    kmask = nullptr;
  }

  __device__ __forceinline__ void carve(char* base) {
    double* d = reinterpret_cast<double*>(base);
    carve_rows(d);
    d += kRows * (size_t)NCAP;
    for (int k = 0; k < 10; k++) at[k] = d + (size_t)k * ACAP;
    d += 10 * (size_t)ACAP;
    cand_vol = d;
    d += ACAP;
    misc = d;
    d += 8;
    int* ip = reinterpret_cast<int*>(d);
    cand_idx = ip;
    ip += ACAP;
    at_gidx = ip;
    ip += ACAP;
    lvl = ip;
    ip += 12;
    ctl = ip;
    ip += 12;
    rt = ip;
    ip += kRootWords;
    unsigned short* sp = reinterpret_cast<unsigned short*>(ip);
    pairs = sp;
    pcnt = reinterpret_cast<int*>(cand_vol);
    nla = sp;
    npar = sp + NCAP;
    ncs = sp + 2 * (size_t)NCAP;
    ncc = sp + 3 * (size_t)NCAP;
    tstart = sp + 4 * (size_t)NCAP;
    cbase = tstart + (kTreeBlock + 2);
    kmask = reinterpret_cast<unsigned long long*>((reinterpret_cast<uintptr_t>(cbase + (kTreeBlock + 2)) + 7) & ~(uintptr_t)7);
#ifdef AGBNP_STAMPS
    stamps = kmask + kMaskWords;
#endif
  }
};

// ---- Gaussian merge ----------------------------------------------------------------------------
// 1/x to ~1 ulp: hardware seed + two Newton steps (an IEEE divide costs ~4x as many instructions)
__device__ __forceinline__ double fast_rcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
  r = fma(fma(-x, r, 1.0), r, r);
  return r;
}

__device__ __forceinline__ double dev_switch(double gvol, double& sp) {
  // gaussvol.cpp:18-41 with volmina/volminb = VOLMINA/VOLMINB
  const double va = kVolMinA, vb = kVolMinB;
  const double w = 1.0 / (vb - va);
  double u = (gvol - va) * w;
  double u2 = u * u;
  double u3 = u * u2;
  double s = u3 * (10.0 - 15.0 * u + 6.0 * u2);
  double d = w * 30.0 * u2 * (1.0 - 2.0 * u + u2);
  if (gvol > vb) {
    s = 1.0;
    d = 0.0;
  } else if (gvol < va) {
    s = 0.0;
    d = 0.0;
  }
  sp = d;
  return s;
}

// Overlap of two Gaussians (gaussvol/gaussvol.cpp:60-93): returns the switched volume s(V) V (the keep test of the
// expansion) and hands back the unswitched V, which is the node record's volume -- the rest of the record is only
// the weighted centre, see dev_merge_known
__device__ __forceinline__ double dev_merge_volume2(double x1, double y1, double z1, double a1, double v1, double x2, double y2,
                                                    double z2, double a2, double v2, double& gvol_out) {
  const double dx = x2 - x1, dy = y2 - y1, dz = z2 - z1;
  const double d2 = dx * dx + dy * dy + dz * dz;
  const double df = a1 * a2 * fast_rcp(a1 + a2);
  const double q = df * (1.0 / kPi);
  const double gvol = v1 * (v2 * pow_three_halves(q) * exp_nonpositive(-df * d2));
  gvol_out = gvol;
  double sp;
  return dev_switch(gvol, sp) * gvol;
}

// node record of an overlap whose unswitched volume is already known: exponent sum and weighted centre, no exp / sqrt
__device__ __forceinline__ void dev_merge_known(double x1, double y1, double z1, double a1, double x2, double y2, double z2,
                                                double a2, double gvol, double& x, double& y, double& z, double& a) {
  const double a12 = a1 + a2;
  const double deltai = fast_rcp(a12);
  x = (x1 * a1 + x2 * a2) * deltai;
  y = (y1 * a1 + y2 * a2) * deltai;
  z = (z1 * a1 + z2 * a2) * deltai;
  a = a12;
  (void)gvol;
}

// ---- wave / workgroup helpers -------------------------------------------------------------------
// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for every global store and atomic the
// wave has in flight (vmcnt(0)); after the topology has been written out or the per-atom sums flushed nothing in
// the workgroup depends on those, and under load they take microseconds to retire.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
// The barrier of the tree kernels: their working set is in LDS for every variant but the HBM-scratch one.
template <int NCAP>
__device__ __forceinline__ void tree_barrier() {
  if (NCAP > 2048)
    __syncthreads();
  else
    lds_barrier();
}

__device__ __forceinline__ int lane_prefix(unsigned long long mask) {
  return __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
}

__device__ __forceinline__ void lds_add(double* p, double v) {
#ifdef AGBNP_TIMING_PLAIN_LDS_ADD  // timing experiment only: racy, results are wrong
  *p += v;
  return;
#endif
  // LDS FP64 add; with -munsafe-fp-atomics this is ds_add_f64
  __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void glb_add(double* p, double v) {
#ifdef AGBNP_TIMING_NO_ATOMICS  // timing experiment only: results are wrong
  if (v == 1.2345e300) *p = v;
  return;
#endif
  __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// wave64 inclusive scan (DPP: 4 shifts inside each row of 16 lanes, then two row broadcasts)
__device__ __forceinline__ int wave_inclusive_scan(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);  // row_shr:1
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);  // row_shr:2
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);  // row_shr:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);  // row_shr:8
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1,3
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2,3
  return v;
}

// sum of one double per lane over the wave, returned on every lane (DPP adds inside rows, two row broadcasts,
// then a broadcast of lane 63): ~3x cheaper than six ds_bpermute exchanges
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_f64(double v) {
  const unsigned long long u = (unsigned long long)__double_as_longlong(v);
  const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)u, CTRL, ROW_MASK, 0xf, false);
  const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(u >> 32), CTRL, ROW_MASK, 0xf, false);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ double wave_sum_f64(double v) {
  v += dpp_f64<0x111, 0xf>(v);  // row_shr:1   (lanes without a source add 0.0)
  v += dpp_f64<0x112, 0xf>(v);  // row_shr:2
  v += dpp_f64<0x114, 0xf>(v);  // row_shr:4
  v += dpp_f64<0x118, 0xf>(v);  // row_shr:8   -> lane 15 of every row holds the row sum
  v += dpp_f64<0x142, 0xa>(v);  // row_bcast:15 -> rows 1,3
  v += dpp_f64<0x143, 0xc>(v);  // row_bcast:31 -> rows 2,3; lane 63 holds the wave sum
  const unsigned long long u = (unsigned long long)__double_as_longlong(v);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, 63);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), 63);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

#ifdef AGBNP_STAMPS
#define AGBNP_BUILD_STAMP(i)                                         \
  do {                                                               \
    if (tid == 0) {                                                  \
      const unsigned long long t_now__ = __builtin_readcyclecounter(); \
      S.stamps[i] += t_now__ - tb_prev__;                            \
      tb_prev__ = t_now__;                                           \
    }                                                                \
  } while (0)
#define AGBNP_BUILD_STAMP_BEGIN() unsigned long long tb_prev__ = __builtin_readcyclecounter()
#else
#define AGBNP_BUILD_STAMP(i)
#define AGBNP_BUILD_STAMP_BEGIN()
#endif

// bits of the tasks [ts, te) (te - ts <= 63) that survived, bit b <-> task ts + b
__device__ __forceinline__ unsigned long long kept_bits(const unsigned long long* kmask, int ts, int te) {
  const int len = te - ts;
  if (len <= 0) return 0ull;
  const int c0 = ts >> 6, sh = ts & 63;
  unsigned long long m = kmask[c0] >> sh;
  if (sh + len > 64) m |= kmask[c0 + 1] << (64 - sh);  // sh > 0 here
  return m & ((1ull << len) - 1ull);
}

enum BuildResult { kBuildOk = 0, kBuildNodeOverflow = 1, kBuildAtomOverflow = 2 };

// ---- EXPERIMENT (compile-time, -DAGBNP_WAVE_TAIL=<nodes>, 0 = off = the product): a wave-local tail of the
// expansion.  1dwc's levels 5-7 hold 25 / 5 / 0.1 nodes per forest, yet each still pays three workgroup-barrier-delimited
// phases (~6.8 k cycles per level whatever its width: profiles/r05/stamps_tree_five_vs_six_launches.txt).  When a level (from
// level 3 on: the ownership of level-2 nodes is the workgroup path's business) has at most AGBNP_WAVE_TAIL nodes, wave 0
// finishes the remaining levels alone -- the same three phases per batch, the same arithmetic and the same node order, with
// `s_waitcnt lgkmcnt(0)` where the workgroup path has `s_barrier` (a wave's LDS traffic is in order) -- while the other
// waves wait at the barrier behind the build.  Returns false if the store overflowed.
#ifndef AGBNP_WAVE_TAIL
#define AGBNP_WAVE_TAIL 0
#endif
__device__ __forceinline__ void wave_sync() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}
template <int NCAP, int ACAP>
__device__ __forceinline__ bool expand_tail_wave(const TreeStore<NCAP, ACAP>& S, int lane, int L, int level_begin, int& tail_io) {
  constexpr int TCAP = TreeStore<NCAP, ACAP>::TCAP;
  unsigned char* tmap = reinterpret_cast<unsigned char*>(S.cand_vol);
  double* tvol = S.nd[6];
  int tail = tail_io;
  for (; L < kMaxOrder; L++) {
    const int lb = level_begin, le = tail;
    level_begin = le;
    if (lb >= le) break;
    for (int nb = lb; nb < le;) {
      // phase 0 (as the workgroup path's narrow form)
      const int k = nb + lane;
      const bool has = k < le;
      const int cnt = has ? (int)S.ncs[k] - k - 1 : 0;
      const int incl = wave_inclusive_scan(cnt);
      const bool inb = has && (incl <= TCAP);
      const int nin = __popcll(__ballot(inb));
      const int T = __builtin_amdgcn_readlane(incl, nin - 1);
      if (inb) {
        const int excl = incl - cnt;
        S.tstart[lane] = (unsigned short)excl;
        for (int i = 0; i < cnt; i++) tmap[excl + i] = (unsigned char)lane;
      }
      if (lane == 0) S.tstart[nin] = (unsigned short)T;
      wave_sync();
      // phase 1
      int keep_j = 0, keep_ts = 0, keep_te = 0, keep_la = 0;
      double keep_v = 0.0;
      for (int t = lane; t < T; t += 64) {
        const int j = tmap[t];
        const int kk = nb + j;
        const int tsj = S.tstart[j], tej = S.tstart[j + 1];
        const int s = kk + 1 + (t - tsj);
        const int la = S.nla[s];
        double gv;
        const double v = dev_merge_volume2(S.nd[0][kk], S.nd[1][kk], S.nd[2][kk], S.nd[3][kk], S.nd[4][kk], S.at[0][la],
                                           S.at[1][la], S.at[2][la], S.at[3][la], S.at[4][la], gv);
        const bool kept = v > kMinGvol;
        tvol[t] = kept ? gv : 0.0;
        if (t == lane) keep_j = j, keep_ts = tsj, keep_te = tej, keep_la = la, keep_v = kept ? gv : 0.0;
        const unsigned long long km = __ballot(kept);
        if (lane == 0) S.kmask[t >> 6] = km;
      }
      wave_sync();
      // phases 2 + 3
      const int nwords = (T + 63) >> 6;
      int created = 0;
      for (int w = 0; w < nwords; w++) created += __popcll(S.kmask[w]);
      if (tail + created > NCAP) return false;
      auto kept_before = [&](int ts) {
        const int wi = ts >> 6;
        int c = __popcll(S.kmask[wi] & ((1ull << (ts & 63)) - 1ull));
        for (int w = 0; w < wi; w++) c += __popcll(S.kmask[w]);
        return c;
      };
      auto create = [&](int t, double v, int j, int ts, int te, int la_known) {
        const int kk = nb + j;
        if (v > 0.0 || t == ts) {
          int rank = 0, c = 0;
          for (int s0 = ts; s0 < te; s0 += 63) {
            const int s1 = (ACAP <= 64 || s0 + 63 >= te) ? te : s0 + 63;
            unsigned long long m = kept_bits(S.kmask, s0, s1);
            c += __popcll(m);
            if (v > 0.0)
              for (; m; m &= m - 1) {
                const int u = s0 + __builtin_ctzll(m);
                const double vu = tvol[u];
                rank += (vu > v || (vu == v && u < t)) ? 1 : 0;
              }
            if (ACAP <= 64) break;
          }
          const int cb = tail + kept_before(ts);
          if (t == ts && c > 0) {
            S.ncs[kk] = (unsigned short)cb;
            S.ncc[kk] = (unsigned short)c;
          }
          if (v > 0.0) {
            const int slot = cb + rank;
            const int la = la_known >= 0 ? la_known : (int)S.nla[kk + 1 + (t - ts)];
            double mx, my, mz, ma;
            dev_merge_known(S.nd[0][kk], S.nd[1][kk], S.nd[2][kk], S.nd[3][kk], S.at[0][la], S.at[1][la], S.at[2][la], S.at[3][la], v,
                            mx, my, mz, ma);
            S.nd[0][slot] = mx;
            S.nd[1][slot] = my;
            S.nd[2][slot] = mz;
            S.nd[3][slot] = ma;
            S.nd[4][slot] = v;
            S.nla[slot] = (unsigned short)la;
            S.npar[slot] = (unsigned short)kk;
            S.ncs[slot] = (unsigned short)(cb + c);
            S.ncc[slot] = 0;
          }
        }
      };
      if (lane < T) create(lane, keep_v, keep_j, keep_ts, keep_te, keep_la);
      for (int t = lane + 64; t < T; t += 64) {
        const int j = tmap[t];
        create(t, tvol[t], j, S.tstart[j], S.tstart[j + 1], -1);
      }
      wave_sync();
      tail += created;
      nb += nin;
    }
  }
  tail_io = tail;
  return true;
}

// ---- build the forest of the heavy atoms roots[0..m) (large radii) -----------------------------------------
// Node and local-atom numbering: 0..m-1 are the roots (level 1), then the level-2 nodes of root 0, of root 1, ...
// (level-2 node k <-> local atom k), then level 3 of all trees, and so on: levels are contiguous over the whole
// forest, sibling lists never mix trees.  returns BuildResult (workgroup-uniform); on success *nnodes_out /
// *natoms_out are set.
// position `idx` of the caller's [3n] array, or of an OpenMM context's posq (as slot_position reads it, prep_role.h)
template <bool POSQ>
__device__ __forceinline__ void tree_position(const TreeArgs& A, int idx, double& x, double& y, double& z) {
  if (!POSQ) {
    const double* __restrict__ pr = A.pos + 3 * (size_t)idx;
    x = pr[0], y = pr[1], z = pr[2];
  } else if (A.posq_double) {
    const double4 p = static_cast<const double4*>(A.posq)[idx];
    x = p.x, y = p.y, z = p.z;
  } else {
    const float4 p = static_cast<const float4*>(A.posq)[idx];
    x = (double)p.x, y = (double)p.y, z = (double)p.z;
    if (A.posq_corr) {  // mixed precision: position = posq + posqCorrection, both float
      const float4 c = A.posq_corr[idx];
      x += (double)c.x, y += (double)c.y, z += (double)c.z;
    }
  }
}

template <int NCAP, int ACAP, int BS, bool FIVE = false, bool POSQ = false>
__device__ int build_forest(const TreeStore<NCAP, ACAP>& S, const TreeArgs& A, int tid, int my_item, const int (&items)[kMaxRoots], int m,
                            int* nnodes_out, int* natoms_out, int my_atom = 0) {
  // FIVE (five-launch mode): positions come from the caller's array -- a root's through the atom index that arrived with its
  // work item (my_atom), a candidate's through h2a (one more dependent load, on a table every workgroup of the CU shares)
  // my_item: work item number tid of the forest (lanes tid < m); items[]: all of them, wave-uniform (scalar registers)
  constexpr int TCAP = TreeStore<NCAP, ACAP>::TCAP;
  AGBNP_BUILD_STAMP_BEGIN();
  // The neighbour-mask words of the roots are requested NOW, together with the roots' own parameters: both only need
  // the roots' heavy indices, which arrived with the work items.  (root, block) pairs of root q: blocks
  // (heavy_q >> 6) .. nhb-1, laid end to end; lane tid takes pair tid (further trips, if any, follow the barrier).
  int off[kMaxRoots + 1];
  off[0] = 0;
#pragma unroll
  for (int q = 0; q < kMaxRoots; q++) off[q + 1] = off[q] + (q < m ? A.nhb - (work_item_root(items[q]) >> 6) : 0);
  const int npairs = off[kMaxRoots];
  const size_t mask_stride = (size_t)A.nhb * 64;
  auto pair_of = [&](int c, int& q, int& J, int& hq) {  // pair c -> root, block, the root's heavy index
    // one fused pass over the (non-decreasing) offsets.  (Looking off[q] up after the count makes the compiler index
    // the array dynamically, i.e. put it in scratch.)
    q = 0;
    int o = 0;
    hq = work_item_root(items[0]);
#pragma unroll
    for (int k = 1; k < kMaxRoots; k++) {
      const bool past = k < m && c >= off[k];
      q += past ? 1 : 0;
      o = past ? off[k] : o;
      hq = past ? work_item_root(items[k]) : hq;
    }
    J = (hq >> 6) + (c - o);
  };
  int q_first, J_first, hq_first;
  pair_of(tid, q_first, J_first, hq_first);
  const unsigned long long bits_first = tid < npairs ? A.nbmask[(size_t)J_first * mask_stride + hq_first] : 0ull;
  if (tid < m) {
    const int item = my_item;
    const int hi = work_item_root(item);
    double rx, ry, rz;
    if (FIVE) {
      tree_position<POSQ>(A, my_atom, rx, ry, rz);  // (POSQ: my_atom is the root's slot in the context's order)
    } else {
      rx = A.hvat(kHvX, hi), ry = A.hvat(kHvY, hi), rz = A.hvat(kHvZ, hi);
    }
    const double ra = A.hvat(kHvALarge, hi), rv = A.hvat(kHvVLarge, hi), rg = A.hvat(kHvGam, hi);
    S.at[0][tid] = rx;
    S.at[1][tid] = ry;
    S.at[2][tid] = rz;
    S.at[3][tid] = ra;
    S.at[4][tid] = rv;
    S.at[5][tid] = rg;
    S.at_gidx[tid] = hi;
    S.nd[0][tid] = rx;
    S.nd[1][tid] = ry;
    S.nd[2][tid] = rz;
    S.nd[3][tid] = ra;
    S.nd[4][tid] = rv;
    S.nla[tid] = (unsigned short)tid;
    S.npar[tid] = 0xFFFF;
    S.rt[kRtHeavy + tid] = hi;
    S.rt[kRtCount + tid] = 0;
    S.rt[kRtNodes + tid] = 0;
    S.rt[kRtPart + tid] = work_item_part(item) | (work_item_parts(item) << 8);
  }
  if (tid == 0) {
    S.ctl[0] = 0;  // level-2 candidate counter of the whole forest
    S.ctl[5] = 0;  // near-candidate counter (stage 1 of the search)
    S.ctl[6] = 0;  // set when the near list overflowed
    S.rt[kRtNum] = m;
  }
  tree_barrier<NCAP>();

  // ---- level 2: for every root, all heavy atoms with a larger index whose overlap with the root survives the
  // switch.  The tile workgroups of the k_prep launch have left, per heavy atom and block of 64 heavy atoms, a 64-bit
  // mask of the younger atoms inside the conservative cutoff (agbnp_common.h).  Stage 1 (here): the (root, block)
  // pairs of the forest's roots are laid end to end, one pair per lane and trip: fetch the word (one round trip for
  // the whole forest in all but huge systems), reserve as many near slots as it has bits with one LDS add, expand
  // the bits.  Stage 2 (below) takes the exact test (a Gaussian merge: exp, sqrt, reciprocal -- two hundred
  // instructions that a wave executes in full as soon as ONE of its lanes has work) densely, one near candidate per
  // lane; a hit takes a slot with an LDS counter and parks its atom record in the (still unused) upper node slots so
  // that ranking never goes back to HBM.
  constexpr int kNearCap = NCAP - ACAP;  // staging slots NCAP-1-p, p < kNearCap: clear of the roots and level-2 nodes (< ACAP)
  auto accept = [&](int q, int hj, double sv, double gvol, double xj, double yj, double zj, double aj, double vj, double gj) {
    const int p = atomicAdd(&S.ctl[0], 1);
    atomicAdd(&S.rt[kRtCount + q], 1);
    if (p < ACAP - m) {
      S.cand_vol[p] = sv;
      S.cand_idx[p] = hj | (q << 24);
      const int st = NCAP - 1 - p;  // staging slot of the accepted candidates (level-2 nodes land below ACAP)
      S.nd[0][st] = xj;
      S.nd[1][st] = yj;
      S.nd[2][st] = zj;
      S.nd[3][st] = aj;
      S.nd[4][st] = vj;
      S.at[9][p] = gj;  // (the self-volume accumulators are idle during the build; handed back as zeros below)
      S.nd[6][st] = gvol;  // unswitched overlap with the root: the level-2 node's volume
    }
  };
  for (int base = 0; base < npairs; base += BS) {
    const int c = base + tid;
    int q = q_first, J = J_first, hq;
    unsigned long long bits = bits_first;  // first trip: already in flight since before the barrier
    if (base > 0) {
      pair_of(c, q, J, hq);
      bits = c < npairs ? A.nbmask[(size_t)J * mask_stride + hq] : 0ull;
    }
    if (bits) {
      int p = atomicAdd(&S.ctl[5], (int)__popcll(bits));
      for (; bits; bits &= bits - 1, p++) {
        if (p < kNearCap)
          S.nd[6][NCAP - 1 - p] = __hiloint2double(0, (64 * J + __builtin_ctzll(bits)) | (q << 24));
        else
          S.ctl[6] = 1;  // more near candidates than staging slots (dense synthetic systems): next capacity variant
      }
    }
  }
  tree_barrier<NCAP>();
  if (S.ctl[6]) return kBuildAtomOverflow;  // more near candidates than staging slots: next capacity variant
  {
    // stage 2: exact test, one near candidate per lane: its record comes from the heavy-atom table (six gathers,
    // one round trip per trip of BS candidates).  An accepted candidate moves to the staging slot of its running
    // number, which is never above the near slots already read (numbers are handed out as trips complete).
    const int nnear = S.ctl[5];
    for (int base = 0; base < nnear; base += BS) {
      const bool mine = base + tid < nnear;
      const int st = NCAP - 1 - (mine ? base + tid : 0);
      const int packed = __double2loint(S.nd[6][st]);
      const int q = mine ? packed >> 24 : 0;
      const int hjn = mine ? packed & 0xffffff : 0;
      double xj, yj, zj;
      const double aj = A.hvat(kHvALarge, hjn), vj = A.hvat(kHvVLarge, hjn), gj = A.hvat(kHvGam, hjn);
      if (FIVE) {
        tree_position<POSQ>(A, POSQ ? A.hslot[hjn] : A.out.h2a[hjn], xj, yj, zj);
      } else {
        xj = A.hvat(kHvX, hjn), yj = A.hvat(kHvY, hjn), zj = A.hvat(kHvZ, hjn);
      }
      double sv = 0.0, gvol = 0.0;
      if (mine) sv = dev_merge_volume2(S.at[0][q], S.at[1][q], S.at[2][q], S.at[3][q], S.at[4][q], xj, yj, zj, aj, vj, gvol);
      tree_barrier<NCAP>();  // the near records of this trip are in registers: accepted ones may take staging slots
      if (mine && sv > kMinGvol) accept(q, packed & 0xffffff, sv, gvol, xj, yj, zj, aj, vj, gj);
      tree_barrier<NCAP>();
    }
  }
  tree_barrier<NCAP>();
  AGBNP_BUILD_STAMP(8);
  const int ncand = S.ctl[0];
  if (m + ncand > ACAP) return kBuildAtomOverflow;
  if (m + ncand > NCAP) return kBuildNodeOverflow;
  // first partner slot of every root (all lanes compute it: m <= kMaxRoots LDS reads)
  int cntq[kMaxRoots], baseq[kMaxRoots];
  {
    int run = m;
#pragma unroll
    for (int q = 0; q < kMaxRoots; q++) {
      cntq[q] = q < m ? S.rt[kRtCount + q] : 0;
      baseq[q] = run;
      run += cntq[q];
    }
  }
  if (tid < m) {
    int bq = 0, cq = 0;
#pragma unroll
    for (int q = 0; q < kMaxRoots; q++) {
      bq = (q == tid) ? baseq[q] : bq;
      cq = (q == tid) ? cntq[q] : cq;
    }
    S.rt[kRtBase + tid] = bq;
    S.ncs[tid] = (unsigned short)bq;  // first child
    S.ncc[tid] = (unsigned short)cq;
  }
  // rank inside the root's candidates by switched volume (descending; exact ties by atom index) and create local
  // atoms + level-2 nodes.  The rank is a count over all candidates: a chain of LDS round trips, four reads each.  With
  // up to 64 candidates (the usual forest) every wave counts a share of the list for candidate `lane` and the shares meet
  // in LDS (the task-start words are idle until the expansion): a third of the trips for one barrier more.
  constexpr int kWaves = BS / 64;
  const bool split_rank = kWaves > 1 && ncand <= 64;
  if (split_rank) {
    const int c = tid & 63, w = tid >> 6;
    const int share = (ncand + kWaves - 1) / kWaves, k0 = w * share, k1 = min(ncand, k0 + share);
    int part = 0;
    if (c < ncand) {
      const double my = S.cand_vol[c];
      const int packed = S.cand_idx[c];
      const int q = packed >> 24, hj = packed & 0xffffff;
      auto before = [&](double vk, int ik) {
        return ((ik >> 24) == q && (vk > my || (vk == my && (ik & 0xffffff) < hj))) ? 1 : 0;
      };
      int k = k0;
      for (; k + 4 <= k1; k += 4) {
        const double v0 = S.cand_vol[k], v1 = S.cand_vol[k + 1], v2 = S.cand_vol[k + 2], v3 = S.cand_vol[k + 3];
        const int i0 = S.cand_idx[k], i1 = S.cand_idx[k + 1], i2 = S.cand_idx[k + 2], i3 = S.cand_idx[k + 3];
        part += (before(v0, i0) + before(v1, i1)) + (before(v2, i2) + before(v3, i3));
      }
      for (; k < k1; k++) part += before(S.cand_vol[k], S.cand_idx[k]);
    }
    S.tstart[w * 64 + c] = (unsigned short)part;
    tree_barrier<NCAP>();
  }
  for (int c = tid; c < ncand; c += BS) {
    const double my = S.cand_vol[c];
    const int packed = S.cand_idx[c];
    const int q = packed >> 24, hj = packed & 0xffffff;
    // four independent LDS reads per trip: the loop is bound by LDS latency, not by the compares
    int rank = 0;
    int k = 0;
    auto before = [&](double vk, int ik) {  // same root, larger volume; exact tie (incl. itself: never counted): atom index
      return ((ik >> 24) == q && (vk > my || (vk == my && (ik & 0xffffff) < hj))) ? 1 : 0;
    };
    if (split_rank) {
      for (int w = 0; w < kWaves; w++) rank += S.tstart[w * 64 + c];
      k = ncand;
    }
    for (; k + 4 <= ncand; k += 4) {
      const double v0 = S.cand_vol[k], v1 = S.cand_vol[k + 1], v2 = S.cand_vol[k + 2], v3 = S.cand_vol[k + 3];
      const int i0 = S.cand_idx[k], i1 = S.cand_idx[k + 1], i2 = S.cand_idx[k + 2], i3 = S.cand_idx[k + 3];
      rank += (before(v0, i0) + before(v1, i1)) + (before(v2, i2) + before(v3, i3));
    }
    for (; k < ncand; k++) rank += before(S.cand_vol[k], S.cand_idx[k]);
    int bq = 0, cq = 0;
#pragma unroll
    for (int r = 0; r < kMaxRoots; r++) {
      bq = (r == q) ? baseq[r] : bq;
      cq = (r == q) ? cntq[r] : cq;
    }
    const int slot = bq + rank;  // level-2 node k <-> local atom k
    const int st = NCAP - 1 - c;
    const double x2 = S.nd[0][st], y2 = S.nd[1][st], z2 = S.nd[2][st], a2 = S.nd[3][st], v2 = S.nd[4][st], g2 = S.at[9][c];
    S.at[9][c] = 0.0;
    S.at[0][slot] = x2;
    S.at[1][slot] = y2;
    S.at[2][slot] = z2;
    S.at[3][slot] = a2;
    S.at[4][slot] = v2;
    S.at[5][slot] = g2;
    S.at_gidx[slot] = hj;
    const double gv2 = S.nd[6][st];  // unswitched overlap with the root, from the exact test of the search
    double cmx, cmy, cmz, cma;
    dev_merge_known(S.at[0][q], S.at[1][q], S.at[2][q], S.at[3][q], x2, y2, z2, a2, gv2, cmx, cmy, cmz, cma);
    S.nd[0][slot] = cmx;
    S.nd[1][slot] = cmy;
    S.nd[2][slot] = cmz;
    S.nd[3][slot] = cma;
    S.nd[4][slot] = gv2;
    S.nla[slot] = (unsigned short)slot;
    S.npar[slot] = (unsigned short)q;
    S.ncs[slot] = (unsigned short)(bq + cq);  // until the node is expanded: end of its sibling list
    S.ncc[slot] = 0;
  }
  tree_barrier<NCAP>();

  AGBNP_BUILD_STAMP(9);
  // ---- levels 3..8: level-synchronous expansion in batches of <= BS nodes / <= TCAP tasks
  unsigned char* tmap = reinterpret_cast<unsigned char*>(S.cand_vol);  // task -> node (index inside the batch)
  double* tvol = S.nd[6];                                                // task -> switched volume (0 = rejected)
  int tail = m + ncand;
  int L = 2;
  // (round 4: three dependent LDS round trips less per level -- the level bounds live in registers, every wave works the
  // batch's task counts out for itself instead of reading wave 0's result back after the barrier, and a forest without a
  // shared subtree does not look the ownership of its level-2 nodes up)
  bool any_shared = false;
#pragma unroll
  for (int q = 0; q < kMaxRoots; q++) any_shared = any_shared || (q < m && work_item_parts(items[q]) > 1);
  int level_begin = m;  // first node of level L (level 2 follows the roots)
#if AGBNP_WAVE_TAIL > 0
  int tail_from = -1;
#endif
  for (; L < kMaxOrder; L++) {
    const int lb = level_begin, le = tail;  // nodes of level L; their children go to level L+1 starting at `tail`
    level_begin = le;
    if (lb >= le) break;
#if AGBNP_WAVE_TAIL > 0
    if (L >= 3 && le - lb <= AGBNP_WAVE_TAIL && NCAP <= 2048) {  // the experiment above: wave 0 alone from here on
      tail_from = lb;
      break;
    }
#endif
    // a level-2 node is expanded by the work item that owns its rank (see kRtPart); deeper nodes by whoever created them
    auto owned_level2 = [&](int k) {
      if (L != 2 || !any_shared) return true;
      const int q = S.npar[k], pp = S.rt[kRtPart + q];
      return level2_owner(k - S.rt[kRtBase + q], pp >> 8) == (pp & 0xff);
    };
    for (int nb = lb; nb < le;) {
      // phase 0: one node per lane -> number of younger siblings = tasks, their prefix sum, the task -> node byte
      // map.  A node's ncs still holds the end of its sibling list at this point.  Up to 64 nodes are handled by
      // wave 0 alone; wider levels (forests, big subtrees) take all waves and one more barrier for up to BS nodes.
      const bool wide = BS > 64 && le - nb > 64;
      int nin, T;
      if (!wide) {
        // (every wave: the counts are one LDS read and a DPP scan; only wave 0 writes the map)
        const int ln = tid & 63;
        const int k = nb + ln;
        const bool has = k < le;
        const int cnt = (has && owned_level2(k)) ? (int)S.ncs[k] - k - 1 : 0;
        const int incl = wave_inclusive_scan(cnt);
        const bool inb = has && (incl <= TCAP);  // prefix property: the batch is lanes 0..nin-1
        nin = __popcll(__ballot(inb));
        T = __builtin_amdgcn_readlane(incl, nin - 1);  // nin >= 1: a single node has < ACAP <= TCAP tasks
        if (tid < 64) {
          if (inb) {
            const int excl = incl - cnt;
            S.tstart[tid] = (unsigned short)excl;
            for (int i = 0; i < cnt; i++) tmap[excl + i] = (unsigned char)tid;
          }
          if (tid == 0) S.tstart[nin] = (unsigned short)T;
        }
      } else {
        const int wv = tid >> 6, ln = tid & 63;
        const int k = nb + tid;
        const bool has = k < le;
        const int cnt = (has && owned_level2(k)) ? (int)S.ncs[k] - k - 1 : 0;
        const int local = wave_inclusive_scan(cnt);
        if (ln == 63) S.ctl[4 + wv] = local;
        tree_barrier<NCAP>();
        int woff = 0;
        for (int w = 0; w < wv; w++) woff += S.ctl[4 + w];
        const int incl = local + woff;
        const bool inb = has && (incl <= TCAP);  // prefix property over the whole workgroup
        const unsigned long long bm = __ballot(inb);
        const int nw = __popcll(bm);
        const int tw = nw > 0 ? __builtin_amdgcn_readlane(incl, nw > 0 ? nw - 1 : 0) : 0;
        if (inb) {
          const int excl = incl - cnt;
          S.tstart[tid] = (unsigned short)excl;
          for (int i = 0; i < cnt; i++) tmap[excl + i] = (unsigned char)tid;
        }
        if (ln == 0) {
          S.ctl[8 + wv] = nw;
          S.rt[kRtNum + 1 + wv] = tw;
        }
        tree_barrier<NCAP>();
        if (tid == 0) {
          int n = 0, T = 0;
          for (int w = 0; w < BS / 64; w++) {
            n += S.ctl[8 + w];
            T = S.ctl[8 + w] > 0 ? S.rt[kRtNum + 1 + w] : T;  // the last wave that holds batch nodes knows the total
          }
          S.tstart[n] = (unsigned short)T;
          S.ctl[1] = n;
          S.ctl[2] = T;
        }
      }
      tree_barrier<NCAP>();
      if (wide) {
        nin = S.ctl[1];
        T = S.ctl[2];
      }
      AGBNP_BUILD_STAMP(10);

      // phase 1: one task per lane -> switched volume of (node, sibling's atom).  A lane meets the same tasks again in phase
      // 3; what it has looked up for its FIRST one (node, the node's task range, the sibling's atom, the volume) stays in
      // registers across the barrier: three dependent LDS round trips less at the head of phase 3 for all but the widest
      // batches (round 4).
      int keep_j = 0, keep_ts = 0, keep_te = 0, keep_la = 0;
      double keep_v = 0.0;
      for (int t = tid; t < T; t += BS) {
        const int j = tmap[t];
        const int kk = nb + j;
        const int tsj = S.tstart[j], tej = S.tstart[j + 1];
        const int s = kk + 1 + (t - tsj);
        const int la = S.nla[s];
        double gv;
        const double v = dev_merge_volume2(S.nd[0][kk], S.nd[1][kk], S.nd[2][kk], S.nd[3][kk], S.nd[4][kk], S.at[0][la],
                                           S.at[1][la], S.at[2][la], S.at[3][la], S.at[4][la], gv);
        const bool kept = v > kMinGvol;
        // the UNSWITCHED volume is kept: it is the child's node volume, and the switched volume s(V) V that the
        // reference sorts by is strictly increasing in V wherever a child survives, so the order is the same
        tvol[t] = kept ? gv : 0.0;
        if (t == tid) keep_j = j, keep_ts = tsj, keep_te = tej, keep_la = la, keep_v = kept ? gv : 0.0;
        const unsigned long long km = __ballot(kept);  // tasks t0..t0+63 of this wave trip: t0 = t - lane
        if ((tid & 63) == 0) S.kmask[t >> 6] = km;
      }
      tree_barrier<NCAP>();
      AGBNP_BUILD_STAMP(11);

      // phases 2 + 3 in one: no scan over the nodes.  The survivor masks of phase 1 say everything: the children of
      // node j start at  tail + (number of survivors among the tasks before the node's first task)  -- a prefix
      // popcount over at most kMaskWords words, the same for every lane of a node -- and a kept task's rank inside its
      // node comes from the node's own piece of the mask.  One barrier-delimited phase less per batch.
      const int nwords = (T + 63) >> 6;  // words written by phase 1 of THIS batch (later ones are stale)
      int created = 0;
      for (int w = 0; w < nwords; w++) created += __popcll(S.kmask[w]);  // (same address on every lane: LDS broadcasts)
      if (tail + created > NCAP) return kBuildNodeOverflow;
      auto kept_before = [&](int ts) {  // survivors among the tasks [0, ts)
        const int wi = ts >> 6;
        int c = __popcll(S.kmask[wi] & ((1ull << (ts & 63)) - 1ull));
        for (int w = 0; w < wi; w++) c += __popcll(S.kmask[w]);
        return c;
      };
      AGBNP_BUILD_STAMP(12);
      auto create = [&](int t, double v, int j, int ts, int te, int la_known) {
        const int kk = nb + j;
        if (v > 0.0 || t == ts) {
          // the node's survivors: count (all pieces), and this task's rank among them (descending volume, index on ties)
          int rank = 0, c = 0;
          for (int s0 = ts; s0 < te; s0 += 63) {  // 63-bit pieces of the survivor mask (one piece when ACAP <= 64)
            const int s1 = (ACAP <= 64 || s0 + 63 >= te) ? te : s0 + 63;
            unsigned long long m = kept_bits(S.kmask, s0, s1);
            c += __popcll(m);
            if (v > 0.0)
              for (; m; m &= m - 1) {  // kept siblings only
                const int u = s0 + __builtin_ctzll(m);
                const double vu = tvol[u];
                rank += (vu > v || (vu == v && u < t)) ? 1 : 0;
              }
            if (ACAP <= 64) break;
          }
          const int cb = tail + kept_before(ts);
          if (t == ts && c > 0) {  // the node's first task also records the node's children
            S.ncs[kk] = (unsigned short)cb;  // from here on: first child
            S.ncc[kk] = (unsigned short)c;
          }
          if (v > 0.0) {
            const int slot = cb + rank;
            const int la = la_known >= 0 ? la_known : (int)S.nla[kk + 1 + (t - ts)];
            double mx, my, mz, ma;
            dev_merge_known(S.nd[0][kk], S.nd[1][kk], S.nd[2][kk], S.nd[3][kk], S.at[0][la], S.at[1][la], S.at[2][la], S.at[3][la], v,
                            mx, my, mz, ma);
            S.nd[0][slot] = mx;
            S.nd[1][slot] = my;
            S.nd[2][slot] = mz;
            S.nd[3][slot] = ma;
            S.nd[4][slot] = v;
            S.nla[slot] = (unsigned short)la;
            S.npar[slot] = (unsigned short)kk;
            S.ncs[slot] = (unsigned short)(cb + c);  // end of this child's sibling list
            S.ncc[slot] = 0;
          }
        }
      };
      if (tid < T) create(tid, keep_v, keep_j, keep_ts, keep_te, keep_la);
      for (int t = tid + BS; t < T; t += BS) {
        const int j = tmap[t];
        create(t, tvol[t], j, S.tstart[j], S.tstart[j + 1], -1);
      }
      tree_barrier<NCAP>();
      AGBNP_BUILD_STAMP(13);
      tail += created;
      nb += nin;
    }
  }
#if AGBNP_WAVE_TAIL > 0
  if (tail_from >= 0) {  // (every wave gets here behind the barrier that ended the level before)
    if (tid < 64) {
      const bool fits = expand_tail_wave<NCAP, ACAP>(S, tid, L, tail_from, tail);
      if (tid == 0) S.ctl[1] = fits ? tail : -1;
    }
    tree_barrier<NCAP>();
    tail = S.ctl[1];
    if (tail < 0) return kBuildNodeOverflow;
  }
#endif
  tree_barrier<NCAP>();
  *nnodes_out = tail;
  *natoms_out = m + ncand;
  return kBuildOk;
}

// ---- (3) of a volume pass: gather over the (atom, node) membership list, sorted by atom -------------------------------
template <int NCAP, int ACAP, int BS>
__device__ __forceinline__ void pair_gather(const TreeStore<NCAP, ACAP>& S, int tid, int total, bool with_selfvol, bool det) {
  // (3) gather over the pair list.  Every lane takes an equal, contiguous piece of the list (sorted by atom) and sums
  // the terms of every run of the same atom in registers: work proportional to the memberships (about three per
  // node), independent of how many local atoms the subtree has, balanced whatever the shape of the tree.
  // Combining the runs.  An LDS FP64 atomic holds the CU's LDS pipe for ~160 cycles per wave instruction, whatever
  // the number of active lanes, and a lane-by-lane flush issued about eight of them per wave and pass (racy plain
  // adds in their place: k_tree_cavity -4.4 us, k_tree_pseudo -2.0 us on 1dwc).  So only what really is shared goes
  // through an atomic (this form: -0.6 / -0.35 us; the scan below costs most of what the atomics did):
  //   * a run that begins and ends inside a lane's piece is that lane's alone: plain read-add-write;
  //   * the lane's FIRST run, if it continues the previous lane's last run, is handed to that lane (one DPP shift);
  //   * the lanes' LAST runs are combined inside the wave by a segmented DPP scan keyed by the atom (pieces are
  //     contiguous and the list is sorted, so the lanes of a run are neighbours); the last lane of a segment holds the
  //     run's sum and stores it -- plainly if the run lies inside the wave, with the atomic only if it touches the
  //     wave's first or last lane (it may go on in the neighbouring wave), as does lane 0's first run.
  // That leaves at most one atomic instruction per quantity, wave and pass.
  const int chunk = (total + BS - 1) / BS;
  int k = tid * chunk;
  const int kend = k + chunk < total ? k + chunk : total;
  const int lane = tid & 63;
  int cur = -1, head_atom = -1;
  bool in_first = true;
  double xa = 0.0, ya = 0.0, za = 0.0, ea = 0.0, gx = 0.0, gy = 0.0, gz = 0.0, sv = 0.0;
  double hx = 0.0, hy = 0.0, hz = 0.0, hs = 0.0;  // the piece's first run, if it ends inside the piece
  auto plain_add = [&](int atom, double vx, double vy, double vz, double vs) {  // nobody else touches `atom` in this pass
    S.at[6][atom] += vx;
    S.at[7][atom] += vy;
    S.at[8][atom] += vz;
    if (with_selfvol) S.at[9][atom] += vs;
  };
  auto atomic_add = [&](int atom, double vx, double vy, double vz, double vs) {
#ifdef AGBNP_TIMING_GATHER_PLAIN  // timing experiment only: racy across waves
    return plain_add(atom, vx, vy, vz, vs);
#endif
    lds_add(&S.at[6][atom], vx);
    lds_add(&S.at[7][atom], vy);
    lds_add(&S.at[8][atom], vz);
    if (with_selfvol) lds_add(&S.at[9][atom], vs);
  };
  for (; k < kend; k++) {
    const int pr = S.pairs[k];
    const int a = pr >> 9, n = pr & 511;
    const double cf = S.nd[3][n], cx = S.nd[0][n], cy = S.nd[1][n], cz = S.nd[2][n], wn = S.wrow[n];
    if (a != cur) {
      if (cur >= 0) {  // a run has ended inside the piece
        if (in_first) {
          head_atom = cur, hx = gx, hy = gy, hz = gz, hs = sv;
        } else {
          plain_add(cur, gx, gy, gz, sv);
        }
        in_first = false;
      }
      cur = a;
      xa = S.at[0][a], ya = S.at[1][a], za = S.at[2][a], ea = S.at[3][a];
      gx = gy = gz = sv = 0.0;
    }
    const double am = cf * ea;
    if (det) {  // the list's order inside a run comes from atomic cursors: exact sums of quantized terms
      gx += quantize(am * (xa - cx), kQGrad, true);
      gy += quantize(am * (ya - cy), kQGrad, true);
      gz += quantize(am * (za - cz), kQGrad, true);
    } else {
      gx = fma(am, xa - cx, gx);
      gy = fma(am, ya - cy, gy);
      gz = fma(am, za - cz, gz);
    }
    sv += wn;
  }
  // keys: the atom of the lane's last run; lanes without a piece get keys of their own (no two equal)
  const int key = cur >= 0 ? cur : -2 - lane;
  {
    // first runs: continuation of the previous lane's last run -> that lane takes it; else it is a complete run
    const int prev_key = __builtin_amdgcn_update_dpp(-1, key, 0x138, 0xf, 0xf, false);       // wave_shr:1 (lane 0 keeps -1)
    const bool head_cont = head_atom >= 0 && lane > 0 && prev_key == head_atom;
    if (head_atom >= 0 && !head_cont) {
      if (lane == 0)
        atomic_add(head_atom, hx, hy, hz, hs);  // (may continue the previous wave's last run)
      else
        plain_add(head_atom, hx, hy, hz, hs);
    }
    // lane l receives lane l+1's continuing first run (wave_shl:1; lane 63 receives nothing)
    const double sx = head_cont ? hx : 0.0, sy = head_cont ? hy : 0.0, sz = head_cont ? hz : 0.0, ss = head_cont ? hs : 0.0;
    gx += dpp_f64<0x130, 0xf>(sx);
    gy += dpp_f64<0x130, 0xf>(sy);
    gz += dpp_f64<0x130, 0xf>(sz);
    if (with_selfvol) sv += dpp_f64<0x130, 0xf>(ss);
  }
  // segmented inclusive scan over the lanes (4 shifts inside each row of 16 lanes, then two row broadcasts)
  auto seg_step = [&](auto ctrl_tag, auto mask_tag) {
    constexpr int CTRL = decltype(ctrl_tag)::value, MASK = decltype(mask_tag)::value;
    const int src_key = __builtin_amdgcn_update_dpp(-1, key, CTRL, MASK, 0xf, false);  // lanes without a source keep -1
    const bool same = src_key == key;
    const double ax = dpp_f64<CTRL, MASK>(gx), ay = dpp_f64<CTRL, MASK>(gy), az = dpp_f64<CTRL, MASK>(gz);
    gx += same ? ax : 0.0;
    gy += same ? ay : 0.0;
    gz += same ? az : 0.0;
    if (with_selfvol) {
      const double as = dpp_f64<CTRL, MASK>(sv);
      sv += same ? as : 0.0;
    }
  };
  seg_step(std::integral_constant<int, 0x111>{}, std::integral_constant<int, 0xf>{});  // row_shr:1
  seg_step(std::integral_constant<int, 0x112>{}, std::integral_constant<int, 0xf>{});  // row_shr:2
  seg_step(std::integral_constant<int, 0x114>{}, std::integral_constant<int, 0xf>{});  // row_shr:4
  seg_step(std::integral_constant<int, 0x118>{}, std::integral_constant<int, 0xf>{});  // row_shr:8
  seg_step(std::integral_constant<int, 0x142>{}, std::integral_constant<int, 0xa>{});  // row_bcast:15 -> rows 1, 3
  seg_step(std::integral_constant<int, 0x143>{}, std::integral_constant<int, 0xc>{});  // row_bcast:31 -> rows 2, 3
  {
    const int next_key = __builtin_amdgcn_update_dpp(-1, key, 0x130, 0xf, 0xf, false);  // wave_shl:1 (lane 63 keeps -1)
    const bool seg_end = cur >= 0 && (lane == 63 || next_key != key);
    const int key_first = __builtin_amdgcn_readfirstlane(key), key_last = __builtin_amdgcn_readlane(key, 63);
    if (seg_end) {
      if (key == key_first || key == key_last)
        atomic_add(cur, gx, gy, gz, sv);  // the run may go on in the neighbouring wave
      else
        plain_add(cur, gx, gy, gz, sv);
    }
  }
}

// ---- one volume pass over a built subtree for the radii / gammas currently in the local atom table ------------
// Adds the gradient (at[6..8]) and, if asked, the self volumes (at[9]) of every non-root local atom;
// returns sum_n c_n gamma_n V_n in *e_sum and sum_n c_n V_n in *w_sum over the nodes below the root (valid on
// every lane when WITH_ENERGY).  nd[6] (task volumes during the build) carries the atom paths: byte k of a
// node's path is the local atom added at level k+2, so a node knows its whole atom list without its parent.
//
// Step (1+2) is node-parallel and needs no level ordering:
//   FRESH_BUILD  the node slots still hold the Gaussians and gamma sums the build left there (same radii); a
//                node walks its ancestor chain once to lay down its path.
//   otherwise    a node recomputes its Gaussian directly from its atoms.  The reference's chain of pairwise
//                merges (gaussvol.cpp:60-93 applied along the path) telescopes:
//                  A_k = a_1+..+a_k,  c_k = (A_{k-1} c_{k-1} + a_k r_k)/A_k,  df_k = A_{k-1} a_k / A_k
//                  G   = (prod v_i) * ((prod a_i) / (A_K pi^(K-1)))^(3/2) * exp(-sum_k df_k |c_{k-1} - r_k|^2)
//                (same atom order as the reference's path; one sqrt and one exp per node instead of one per level),
//                so the six level barriers of a top-down rescan disappear and a replay needs nothing but the paths.
__device__ __forceinline__ double pi_power(int k) {  // pi^k, k = 1..7
  // selects, not a table: a table lives in memory, and a vector load issued here would have to wait (vmcnt is
  // in order) for every store and atomic of the topology write-out still in flight
  const double p2 = kPi * kPi, p4 = p2 * p2;
  double r = (k & 1) ? kPi : 1.0;
  r = (k & 2) ? r * p2 : r;
  r = (k & 4) ? r * p4 : r;
  return r;
}

// *npairs: length of the (atom, node) pair list; written by a FRESH_BUILD pass (which builds the list), read by the
// others.  Returns false (workgroup-uniform) if the list does not fit: the caller reports a capacity overflow.
struct NoHook {
  __device__ __forceinline__ void operator()() const {}
};
// after_pairs: called by every thread once the node step is done and the replayed pair list (pair_word) has been moved
// from registers into LDS -- the point from which a replay's registers are free for the NEXT forest's loads (k_tree_pseudo)
template <int NCAP, int ACAP, int BS, bool WITH_ENERGY, bool FRESH_BUILD = false, class AfterPairs = NoHook>
__device__ bool volume_pass(const TreeStore<NCAP, ACAP>& S, int tid, int m, int nnodes, int natoms, bool with_selfvol,
                            double* e_sum, int* npairs, bool det, const uint4* pair_word = nullptr, AfterPairs after_pairs = AfterPairs()) {
  static_assert(ACAP <= 256, "atom path stores one byte per level");
  static_assert(BS % 64 == 0 && BS >= 64, "whole waves");
  constexpr bool kPairs = TreeStore<NCAP, ACAP>::kPairGather;
  constexpr int PCAP = TreeStore<NCAP, ACAP>::PCAP;
  unsigned long long* path = reinterpret_cast<unsigned long long*>(S.nd[6]);
  AGBNP_BUILD_STAMP_BEGIN();
  if (FRESH_BUILD && kPairs) {
    for (int la = tid; la < ACAP; la += BS) S.pcnt[la] = 0;
    tree_barrier<NCAP>();
  }
  // (1+2) node-parallel: centre slots <- c_n, exponent slot <- coef_n = -2 c_n gamma_n sfp_n G_n,
  //       gamma slot <- w_n = c_n s(G_n) G_n   (c_n = +-1/level)
  // A node's path word: bytes 0..6 = the local atoms added at levels 2..8, byte 7 = its root (local atom < m).
  constexpr unsigned long long kPartners = 0x00ffffffffffffffull;
  double e_part = 0.0;
  for (int nbase = m; nbase < nnodes; nbase += BS) {  // whole waves stay together: per-root sums are folded wave by wave
    const int n = nbase + tid;
    int level = 0, rootq = -1;  // rootq < 0: no node on this lane, or an inert one
    double g = 0.0, gam = 0.0, w = 0.0;
    do {
      if (n >= nnodes) break;
    if (FRESH_BUILD) {
      {
        // a level-2 node that another work item owns is a sibling only: empty path = inert in every pass
        const int par = S.npar[n];
        if (par < m) {
          const int pp = S.rt[kRtPart + par];
          if (level2_owner(n - S.rt[kRtBase + par], pp >> 8) != (pp & 0xff)) {
            path[n] = (unsigned long long)par << 56;
            S.nd[3][n] = 0.0;
            S.wrow[n] = 0.0;
            break;
          }
        }
      }
      unsigned long long pw = 0ull;
      level = 1;
      int p = n;
      for (; p >= m; p = S.npar[p]) {  // leaf to root: the deepest atom is met first
        const int la = S.nla[p];
        pw = (pw << 8) | (unsigned long long)la;
        if (kPairs) atomicAdd(&S.pcnt[la], 1);
        gam += S.at[5][la];  // gamma_1..i (gaussvol.cpp:224: the root's gamma plus that of every atom added)
        level++;
      }
      rootq = p;
      gam += S.at[5][rootq];
      path[n] = pw | ((unsigned long long)rootq << 56);
      g = S.nd[4][n];
    } else {
      const unsigned long long pwr = path[n];
      if ((pwr & kPartners) == 0ull) {  // inert (a level-2 node owned by another work item)
        S.nd[3][n] = 0.0;
        S.wrow[n] = 0.0;
        break;
      }
      rootq = (int)(pwr >> 56);
      double A = S.at[3][rootq], cx = S.at[0][rootq], cy = S.at[1][rootq], cz = S.at[2][rootq];
      double pv = S.at[4][rootq], pa = A, E = 0.0;
      gam = S.at[5][rootq];
      level = 1;
      for (unsigned long long pw = pwr & kPartners; pw; pw >>= 8) {
        const int la = (int)(pw & 0xffull);
        const double xk = S.at[0][la], yk = S.at[1][la], zk = S.at[2][la], ak = S.at[3][la];
        const double dx = xk - cx, dy = yk - cy, dz = zk - cz;
        const double inv = fast_rcp(A + ak);
        const double wk = ak * inv;  // a_k / A_k
        E = fma(A * wk, fma(dz, dz, fma(dy, dy, dx * dx)), E);
        cx = fma(dx, wk, cx);  // (A c + a_k r_k) / (A + a_k)
        cy = fma(dy, wk, cy);
        cz = fma(dz, wk, cz);
        A += ak;
        pa *= ak;
        pv *= S.at[4][la];
        gam += S.at[5][la];
        level++;
      }
      const double q = pa * fast_rcp(A * pi_power(level - 1));
      g = pv * pow_three_halves(q) * exp_nonpositive(-E);
      S.nd[0][n] = cx;
      S.nd[1][n] = cy;
      S.nd[2][n] = cz;
    }
    const double cp = ((level & 1) ? 1.0 : -1.0) / (double)level;
    double sp;
    const double sw = dev_switch(g, sp);
    w = quantize(cp * sw * g, kQVol, det);  // (the self-volume sums are order-dependent: LDS atomics, list order)
    S.nd[3][n] = -2.0 * cp * gam * (sp * g + sw) * g;
    S.wrow[n] = w;
    e_part += quantize(gam * (cp * sw * g), kQEnergy, det);  // (which nodes a lane sums depends on the forest's composition)
    } while (false);
    // The root is in every node of its tree: its self volume is the tree's sum of w (and its node count the count).
    // The lanes of a wave mostly share one root (levels are contiguous runs per tree): fold per root inside the wave
    // and add once -- same-address LDS atomics from all lanes serialize.
    if (with_selfvol || FRESH_BUILD) {
      for (unsigned long long todo = __ballot(rootq >= 0); todo;) {
        const int q = __builtin_amdgcn_readlane(rootq, __builtin_ctzll(todo));
        const unsigned long long mine = __ballot(rootq == q);
        if (with_selfvol) {
          const double sq = wave_sum_f64(rootq == q ? w : 0.0);
          if ((tid & 63) == 0) lds_add(&S.at[9][q], sq);
        }
        if (FRESH_BUILD && (tid & 63) == 0) atomicAdd(&S.rt[kRtNodes + q], (int)__popcll(mine));
        todo &= ~mine;
      }
    }
  }
  if (WITH_ENERGY) {
    e_part = wave_sum_f64(e_part);
    if ((tid & 63) == 0) S.misc[tid >> 6] = e_part;
  }
  tree_barrier<NCAP>();
  if (WITH_ENERGY) {
    double es = 0.0;
    for (int w = 0; w < BS / 64; w++) es += S.misc[w];  // fixed order
    *e_sum = es;
  }
  AGBNP_BUILD_STAMP(7);
  if (kPairs) {
    if (FRESH_BUILD) {
      // counting sort of the (atom, node) memberships by atom: counts -> exclusive offsets (wave 0, two atoms per
      // lane) -> every node drops its own entries at its atoms' cursors
      static_assert(!kPairs || ACAP <= 128, "two atoms per lane");
      if (tid < 64) {
        const int a0 = 2 * tid, a1 = 2 * tid + 1;
        const int c0 = a0 < ACAP ? S.pcnt[a0] : 0, c1 = a1 < ACAP ? S.pcnt[a1] : 0;
        const int incl = wave_inclusive_scan(c0 + c1);
        if (a0 < ACAP) S.pcnt[a0] = incl - c0 - c1;
        if (a1 < ACAP) S.pcnt[a1] = incl - c1;
        if (tid == 63) S.ctl[4] = incl;
      }
      tree_barrier<NCAP>();
      const int total = S.ctl[4];
      *npairs = total;
      if (total > PCAP) return false;
      for (int n = m + tid; n < nnodes; n += BS)
        for (unsigned long long pw = path[n] & kPartners; pw; pw >>= 8) {
          const int la = (int)(pw & 0xffull);
          S.pairs[atomicAdd(&S.pcnt[la], 1)] = (unsigned short)((la << 9) | n);
        }
      tree_barrier<NCAP>();
    }
    if (pair_word) {  // replay: the pair list arrives in registers and takes the place of the atom paths
#pragma unroll
      for (int k = 0; k < (PCAP / 8 + BS - 1) / BS; k++)
        if (tid + k * BS < PCAP / 8) reinterpret_cast<uint4*>(S.pairs)[tid + k * BS] = pair_word[k];
      tree_barrier<NCAP>();
      after_pairs();
    }
    pair_gather<NCAP, ACAP, BS>(S, tid, *npairs, with_selfvol, det);
    tree_barrier<NCAP>();
    AGBNP_BUILD_STAMP(14);
    return true;
  }
  // (3) atom-owned gather (variants too large for 16-bit pair words).  The BS lanes form (atom, slice) pairs:
  // A = 16/32/64 atoms per round (the smallest
  {
    const int A = natoms <= 16 ? 16 : (natoms <= 32 ? 32 : 64);
    const int nslices = BS / A;
    const int slice = tid / A, al = tid - slice * A;
    for (int abase = 0; abase < natoms; abase += A) {
      const int a = abase + al;
      const bool live = a >= m && a < natoms;  // the root atoms are done by translation invariance
      const int aa_idx = live ? a : 0;
      const double xa = S.at[0][aa_idx], ya = S.at[1][aa_idx], za = S.at[2][aa_idx], ea = S.at[3][aa_idx];
      const unsigned long long pat = 0x0101010101010101ull * (unsigned long long)aa_idx;
      double gx = 0.0, gy = 0.0, gz = 0.0, sv = 0.0;
      // branch-free body: the six LDS reads of a node are independent of the membership test, so a trip costs
      // one LDS round trip instead of two (path -> test -> record)
      for (int n = m + slice; n < nnodes; n += nslices) {
        const unsigned long long x = path[n] ^ pat;
        const double cf = S.nd[3][n], cx = S.nd[0][n], cy = S.nd[1][n], cz = S.nd[2][n], wn = S.wrow[n];
        const bool member = live && (((x - 0x0101010101010101ull) & ~x & 0x8080808080808080ull) != 0ull);
        const double am = member ? cf * ea : 0.0;
        gx += quantize(am * (xa - cx), kQGrad, det);  // (node numbering = summation order depends on the forest's composition)
        gy += quantize(am * (ya - cy), kQGrad, det);
        gz += quantize(am * (za - cz), kQGrad, det);
        sv += member ? wn : 0.0;
      }
      // fold the slices that live in this wave (lanes A, 2A, ... apart)
      for (int off = A; off < 64; off <<= 1) {
        gx += __shfl_xor(gx, off, 64);
        gy += __shfl_xor(gy, off, 64);
        gz += __shfl_xor(gz, off, 64);
        sv += __shfl_xor(sv, off, 64);
      }
      // Exchange area for the cross-wave fold.  With ACAP <= 64 there is a single round, so the node records
      // are dead after the barrier and their first rows are reused; larger variants (several rounds possible)
      // have NCAP >= 4*BS and use the unswitched-volume row, which nothing reads after step (2).
      constexpr bool kSingleRound = ACAP <= 64;
      static_assert(kSingleRound || NCAP >= 4 * BS, "exchange area");
      double* ex0 = kSingleRound ? S.nd[0] : S.nd[4];
      double* ex1 = kSingleRound ? S.nd[1] : S.nd[4] + BS;
      double* ex2 = kSingleRound ? S.nd[2] : S.nd[4] + 2 * BS;
      double* ex3 = kSingleRound ? S.nd[3] : S.nd[4] + 3 * BS;
      tree_barrier<NCAP>();  // every wave is done reading the node records
      const int wave = tid >> 6, lane = tid & 63;
      if (lane < A) {
        ex0[wave * 64 + lane] = gx;
        ex1[wave * 64 + lane] = gy;
        ex2[wave * 64 + lane] = gz;
        ex3[wave * 64 + lane] = sv;
      }
      tree_barrier<NCAP>();
      if (tid < A && live) {  // wave 0 lanes own the atoms; fixed summation order
        double tx = 0.0, ty = 0.0, tz = 0.0, tv = 0.0;
        for (int w = 0; w < BS / 64; w++) {
          tx += ex0[w * 64 + tid];
          ty += ex1[w * 64 + tid];
          tz += ex2[w * 64 + tid];
          tv += ex3[w * 64 + tid];
        }
        S.at[6][a] += tx;
        S.at[7][a] += ty;
        S.at[8][a] += tz;
        if (with_selfvol) S.at[9][a] += tv;
      }
    }
  }
  tree_barrier<NCAP>();
  AGBNP_BUILD_STAMP(14);
  return true;
}

// ---- after the passes: a root's gradient = -(sum of its partners' gradients) (translation invariance of its tree)
template <int NCAP, int ACAP, int BS>
__device__ void root_gradients_from_invariance(const TreeStore<NCAP, ACAP>& S, int tid, int m) {
  const int wave = tid >> 6, lane = tid & 63;
  for (int q = wave; q < m; q += BS / 64) {  // one wave per root
    const int b = S.rt[kRtBase + q], e = b + S.rt[kRtCount + q];
    double sx = 0.0, sy = 0.0, sz = 0.0;
    for (int la = b + lane; la < e; la += 64) {
      sx += S.at[6][la];
      sy += S.at[7][la];
      sz += S.at[8][la];
    }
    sx = wave_sum_f64(sx);
    sy = wave_sum_f64(sy);
    sz = wave_sum_f64(sz);
    if (lane == 0) {
      S.at[6][q] = -sx;
      S.at[7][q] = -sy;
      S.at[8][q] = -sz;
    }
  }
  tree_barrier<NCAP>();
}

}  // namespace agbnp
