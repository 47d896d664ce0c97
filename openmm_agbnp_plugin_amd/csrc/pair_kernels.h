// Argument block and launchers of the pair stages (see pair_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <vector>

#include "agbnp_common.h"

namespace agbnp {

// Where an evaluation's results go when the caller is an OpenMM GPU context (agbnp_hip_execute_openmm): 2^32 fixed-point
// force planes in the context's atom order and one slot of its energy accumulator.  force_fixed == nullptr: the plain
// FP64 buffers of agbnp_hip_execute_device.
struct OpenmmTargets {
  unsigned long long* force_fixed = nullptr;  // [3 * padded] planes x | y | z
  int padded = 0;
  const int* ctx_slot = nullptr;              // [n] particle -> slot of the context's atom order (written by k_adapt_positions)
  void* energy_buffer = nullptr;
  int energy_is_double = 1, energy_slot = 0;
};

// Where an evaluation's positions come from when the caller is an OpenMM GPU context and the engine's particle -> slot maps
// are in place (agbnp_hip_execute_openmm without an adapter launch): k_prep reads posq itself.  posq == nullptr: PairArgs::pos.
struct OpenmmSource {
  const void* posq = nullptr;          // real4 per SLOT of the context's atom order, double4 or float4
  const float4* correction = nullptr;  // mixed precision: position = posq + correction
  int is_double = 0;
  const int* atom_index = nullptr;     // [padded] slot -> particle (the context's array): what the maps are checked against
  const int* hslot = nullptr;          // [nh] heavy index -> slot (OpenmmTargets::ctx_slot is the particle -> slot map)
};

struct PairArgs {
  OpenmmTargets omm;
  OpenmmSource in;
  int n, nh;
  // ---- per-evaluation input
  const double* pos;  // [3n] caller's positions (nm), atom order
  double* zero_out;   // agbnp_hip_execute_host: its staging buffer for forces [3n] and energy [1], cleared by k_prep (one stream
                      // operation less than a clearing launch of its own); nullptr otherwise
  // ---- static per-atom / per-heavy-atom parameters
  const int* a2h;      // [n] atom -> heavy index or -1
  const int* h2a;      // [nh]
  const double* charge;    // [n]
  const double* alpha;     // [n]
  const double* inv_rvdw;  // [n] 1/R_i
  const double* inv_vol_h;  // [nh] 1/(4 pi R^3/3), vdW radius
  const double* inv_vol_a;  // [n] the same by atom (0 for hydrogens)
  const double* gam_cav;   // [nh] gamma/roffset
  const double *a_large, *v_large;  // [nh] Gaussian exponent / volume with the enlarged radii
  double rcut2;            // conservative squared cutoff of the 2-body overlap search
  int nb_tiles;            // tiles of 64x64 heavy atoms (I <= J) of the level-2 neighbour search (k_prep launch)
  unsigned long long* nbmask;  // [nhb][nhb * 64] neighbour masks, see agbnp_common.h
  const int2* ameta;       // [n] {screened type, screener type or -1}
  const double2* lut;      // [nti*ntj*16] {y, y2*dr^2/6}
  int nti, ntj, lut_entries;
  // ---- geometry (SoA for the tree, packed records for the pair loops)
  double *hx, *hy, *hz;    // [nh]
  double4* aposq;          // [n] {x,y,z,q}
  const int* pslot;        // [nslots] pair order: heavy atoms, padding (-1) to a block of 64, hydrogens, padding
  int nslots, nhb;         // slots (multiple of 64), heavy blocks
  const int* a2s;          // [n] slot of an atom in pair order (inverse of pslot)
  double4* prec;           // [nslots] pair-order record written by k_prep: {x, y, z, types | validity} (low word: screened
                           // type | screener type << 16, high word >= 0 for a real atom): what a Born / chain-rule tile
                           // needs of a slot comes in ONE load, with no slot -> atom indirection in front of it
  double4* srec;           // [nslots] {B, f', brw, q} by slot, published by the GB stage's diagonal tiles for the chain rule
  double* ys;              // [nslots] the GB stage's Y sums by slot (atomic sums)
  double* pbox;            // [nslots/64][6] bounding box {min xyz, max xyz} of every 64-slot block
  double* abox;            // [ceil(n/64)][6] the same for the 64-atom blocks in ATOM order (GB tiles: fast mode, and the far-strip test of large systems)
  // Pair range.  Reference semantics: the descreening stages reach as far as the tables (2 nm), GB has no limit.
  // Fast mode (the semantics of the reference's OpenCL platform, AGBNPBornRadii.cl:268,430, AGBNPGBEnergy.cl:145,186):
  // every pair stage only meets pairs with r^2 < cutoff^2.
  double range2;           // squared reach of the Born / chain-rule stages: min(2 nm, cutoff)^2 in fast mode, else 4
  double gb_cut2;          // squared GB cutoff (fast mode) -- the GB kernel is compiled twice, this is read by the cut one
  int gb_far;              // reference mode: 1 = the GB launch is the instantiation that tests every strip for "so far apart that the
                           // pair terms are pure Coulomb to FP64 rounding" (systems large enough to have such strips)
  int cull_first;          // range-limited stages: 1 = a tile tests its bounding boxes before it asks for its data (large systems)
  int fast;                // 1 = fast mode
  int single;              // 1 = fast mode with the GB pair terms in single precision (GB rows; packed FP32 strips in the tile form)
  int det;                 // 1 = deterministic mode (device_math.h)
  // ---- tree accumulators / outputs
  double *gx, *gy, *gz;    // [nh] tree sums: gradient
  double *sv_vdw, *sv_large;  // [nh] self volumes (enlarged radii: diagnostic)
  double* epart;           // [2nh]
  int2* sizes;             // [nh] {nodes, local atoms} per subtree, summed up by the tree kernel
  int* forest_start;       // [nh+1] packing of the NEXT evaluation: slot s = order[forest_start[s] .. forest_start[s+1])
  int* nforests;           // [1] work slots of the NEXT evaluation
  const int* cur_nforests; // [1] work slots of THIS evaluation (energy partials are per slot)
  int* pack_state;         // [9] persistent: [0] the level: how often the capacity the packing assumes has been tightened (relaxes again
                           // after clean evaluations in a row: word [2] counts them, word [8] says how many are asked for -- the
                           // level's memory); [1] evaluations since the packing in use was planned (huge = it is no plan: one
                           // work item per slot); [3] packings planned so far (a diagnostic); [4], [5] total nodes / largest
                           // subtree of the evaluation the packing was planned from (drift trigger); [6] the tree launches' copy
                           // of the evaluation counter; [7] `heat`: leaky count of evaluations with healed forests
  int replan_every;        // a healthy packing is planned anew every so many evaluations (tuning knob, default 16), or when the trees have drifted
  int2* pack_items;        // [slots] packing_role's scratch: the work items in descending weight order {item, predicted time} (rounds rule)
  int* order;              // [kMaxItems * slots] the work items by FOREST (packing_role -> dealing_role): item k of forest f at kMaxItems * f + k
  int* forest_time;        // [slots + 1] predicted time of every forest (packing_role -> dealing_role), then: are they there
  int* rows;               // [kRowStride * slots] the work items of the NEXT evaluation in WORK-SLOT order (what the tree kernel
                           // reads: item k of slot s at kRowStride * s + k, their number at kRowStride * s + kMaxItems).  Slot s
                           // runs on CU s mod (number of CUs), so the bookkeeping ranks the forests by predicted time and deals
                           // them over the CUs in serpentine order
  int ncus;                // CUs of the device
  int tree_node_cap, tree_atom_cap, pack_enabled;  // capacity of the current tree variant; packing switch
  int tree_slots;          // tree workgroups resident on the device at once (a 'round')
  int split_big, split_permille;  // tuning knobs: parts and node threshold (share of the capacity) for sharing on a full device
  int split_fit;           // 1: subtrees whose items would not fit the store are shared among up to four items (AGBNP_HIP_SPLIT_FIT=0: off)
  int round_permille;      // share of the resident workgroups that the packing fills (tuning knob, default 1000: every resident slot)
  int tree_slot_cap;       // work slots the tree kernels are launched with (>= subtrees; bounds the sharing of subtrees)
  // ---- five-launch mode (the default for version 1; AGBNP_HIP_FIVE_LAUNCHES=0 switches it off; engine.hip): no k_prep launch.  The tree accumulators, the
  //      subtree shapes and the per-evaluation status words exist TWICE and alternate with the evaluation's parity; the
  //      trailing workgroups of the cavity launch (prep_role.h) clear the other set for the next evaluation
  int five;                // 0: six launches.  1: that mode, the HOST names the evaluation's set (eager launches: the pointers of this
                           // block are those of the set, the kernels take them as they are).  2: the DEVICE does (below)
  // Evaluation k of a context works on set k & 1.  The device counts the evaluations too: `epoch`, advanced once per
  // evaluation by the bookkeeping role (the first workgroup of the GB launch) in either form of the mode, so that host and
  // device always agree.  five == 2 (every evaluation of a context from its first stream capture on: a replayed graph freezes
  // its kernel arguments): the pointers of this block are those of set 0 and every kernel moves them itself
  // (rebase_for_parity) -- the launches in front of the GB launch by epoch & 1, the launches behind it by (epoch + 1) & 1; the
  // GB launch's own workgroups other than the role touch nothing that exists twice.  It costs each launch one more cold scalar
  // load (+0.5-1 us per evaluation of 1dwc, A/B on one box), which is why eager contexts let the host do it.
  int* epoch;              // (the pair launches' copy; the tree launches read epoch_tree: engine.hip, apply_parity)
  int* epoch_tree;
  size_t table_doubles;    // doubles from one heavy-atom table to the other
  size_t sizes_stride;     // shapes from one array to the other
  double* next_hv;         // (after rebase_for_parity) heavy-atom table of the other parity
  unsigned hstride;        // row stride of a table
  int2* next_sizes;
  int* next_estatus;
  double mask_rcut2;       // squared reach of the neighbour masks: the conservative cutoff of the level-2 search (+ the skin, in that mode)
  double* mask_ref;        // [3 nh] where the heavy atoms were when the neighbour masks were laid down
  double mask_move2;       // (skin of the masks / 2)^2
  int* row_atoms;          // [kMaxItems * slots] atom index of the root of every work item of a slot's row (the tree reads the
                           // caller's positions itself in that mode); written with the rows
  int* status;             // [kStatTotalWords]: the STICKY words (from kStatEvalSeq on) are addressed through this one,
  int* estatus;            // the words of ONE evaluation ([0, kStatEvalWords)) through this one: the same array -- or, in the
                           // five-launch mode, the block of the evaluation's parity (engine.hip)
  volatile int* host_status;  // pinned host memory, mapped: [0] evaluations completed since the last finish, [1] of them withheld --
                              // what agbnp_hip_poll reads without touching the device (written by the energy role)
  // ---- pair-stage intermediates
  double* born_part;       // [n] sum_j s_j Q (atomic sums of the tiles)
  double *born, *born_fp, *brw, *e_atom;  // [n]
  double *gb_fx, *gb_fy, *gb_fz;          // [n] GB direct force (atomic sums of the symmetric tiles); Y goes to ys
  const int* gb_items;     // [gb_items_count] tiles of k_gb_tiles: I | J<<12 (64-atom blocks, atom order, I <= J)
  int gb_items_count;
  const int* db_items;     // [db_items_count] tiles of k_born_tiles / k_dborn_tiles, same encoding over blocks of pair-order slots
  int db_items_count;
  double *db_fx, *db_fy, *db_fz, *db_wu;  // [n] chain-rule force by atom, W+U by heavy index (atomic sums)
  double* egb_part;        // [egb_parts]
  int egb_parts;
  // ---- row form of the range-limited stages (k_rows, pair_kernels.hip): one wave gathers over the neighbour row of one
  //      atom -- no pair is met that is out of reach, no sum leaves through an atomic.  Reference mode only; the tile
  //      kernels stay for the fast and deterministic modes, for more radius types than the per-wave table slices hold,
  //      and as the fallback when a neighbour row outgrows its stride.
  int rows_on;             // 1: Born sums and chain rule in row form
  double nl_build2;        // squared list radius: (reach + skin)^2
  double nl_move2;         // (skin / 2)^2: an atom further than this from where it was when the rows were built makes them stale
  int row_target;          // workgroups of a row launch that the device takes two per CU of (0: the slice length stays); see rows_close_evaluation
  int* nl_flag;            // [2]: entries per slice of a list (one wave walks a slice), tuned on the device; [0] != 0: the rows are stale for THIS evaluation (k_prep sets, the output side clears and counts the build; 1 on a fresh context)
                           // [1] how often the rows have been built so far (diagnostic)
  double* nl_ref;          // [3n] positions at the last build (NaN on a fresh context)
  const unsigned* hperm;   // [hperm_n] heavy index | screener type << 24, sorted by (type, index), padded with ~0u to whole chunks of 64
  const unsigned* aperm;   // [aperm_n] atom | screened type << 24, sorted by (type, index), same padding
  int hperm_n, aperm_n;
  unsigned* nlh;           // [groups of 4 atoms x kBornParts][nlh_stride] heavy neighbours of the group, entries as in hperm
  int* nlh_count;          // [groups x kBornParts]
  int nlh_stride;
  unsigned* nla;           // [groups of 4 heavy atoms x kChainParts][nla_stride] neighbours of any kind, entries as in aperm
  int* nla_count;          // [groups x kChainParts]
  int nla_stride;
  int gb_rows;             // 1: the GB stage runs in row form too (fast mode: only pairs inside the cutoff are met)
  unsigned* nlg;           // [groups of 4 atoms x kGbParts][nlg_stride] neighbours of any kind within the GB cutoff + skin
  int* nlg_count;          // [groups x kGbParts]
  int nlg_stride;
  double nlg_build2;       // squared radius of those lists
  // Work items of the row launches: (list | slice << 24) of every slice that exists, appended when the lists are built, so
  // that the workgroups with work are the FIRST of a launch and the surplus ones leave on one scalar load (a grid laid out
  // by slice number is half empty workgroups, and those ahead of a working one delay it by microseconds).  Two buffers per
  // kind: buffer (builds & 1) is the one in use, the other one is filled by the next rebuild.
  unsigned* nl_items;      // [3 kinds][2][nl_items_cap]
  int* nl_nitems;          // [3 kinds][2]
  int nl_items_cap;
  int nlh_cap, nla_cap, nlg_cap;  // entries of a list that the launches walk (multiples of 256, <= the strides): what the reach
                                  // of the current mode can fill at protein density; a list that outgrows it withholds the
                                  // evaluation and the host widens the walk
  double4* rec_h;          // [nh] {x, y, z, 1 / V_vdw} by heavy index (k_prep): what a Born row gathers of a neighbour (+ its self volume)
  double4* hrow;           // [nh] {x, y, z, atom | screener type << 24} by heavy index (k_prep): a chain-rule row's own record
  double* bw;              // [n] brw + bru by atom: the GB stage adds alpha_i (diagonal tile) + beta_i * (Y of the tile) with atomics
  double4* grec;           // [n] {G_x, G_y, G_z, -} of the Born rows: G_i = sum_j (r_j - r_i) s_j Q'_ij / d (atomic sums)
  double4* hrec;           // [nh] {H_x, H_y, H_z, -} of the chain-rule rows, by heavy index (atomic sums)
  const unsigned *bslice, *cslice;  // [groups] the table slices (= types) of a group's four row atoms, one byte each (Born / chain-rule groups)
  const double2 *pw_a, *pw_b;    // power-form spline coefficients {c0, c1}, {c2, c3} by [screened][screener][15 intervals]
  const double2 *pwt_a, *pwt_b;  // the same by [screener][screened][15]
};
// Five-launch mode: point the parity-dependent members of a kernel's own copy of the argument block at the evaluation's set.
// after_role: 0 in the launches in front of the GB launch and in its bookkeeping role, 1 behind it.
__device__ __forceinline__ void rebase_for_parity(PairArgs& P, int after_role) {
  if (P.five != 2) return;
  const int par = (P.epoch[0] + after_role) & 1;
  const size_t toff = (size_t)par * P.table_doubles, noff = (size_t)(1 - par) * P.table_doubles;
  P.next_hv = P.hx + noff;  // (hx is row 0 of set 0's table)
  P.next_sizes = P.sizes + (size_t)(1 - par) * P.sizes_stride;
  P.next_estatus = P.estatus + 16 * (1 - par);
  P.hx += toff, P.hy += toff, P.hz += toff;
  P.gx += toff, P.gy += toff, P.gz += toff;
  P.sv_vdw += toff, P.sv_large += toff;
  P.sizes += (size_t)par * P.sizes_stride;
  P.estatus += 16 * par;
}

constexpr int kRowGroup = 4;    // row atoms that share a neighbour list (pair_kernels.hip, k_rows)
constexpr int kRowSlice = 256, kRowWaves = 8;  // entries of the shortest slice of a list (what the launch grids are laid out for); waves per workgroup
// The GB rows keep no table in LDS, so their workgroups can be small: the launch (fast mode, 1dwc: 2840 one-wave items) is
// bound by the vector-memory pipe of the fullest CU (three gathers per step), and four-wave workgroups spread the waves more
// evenly over the CUs than eight-wave ones (8 or 12 waves on a CU instead of 8 or 16).
#ifndef AGBNP_GB_ROW_WAVES
#define AGBNP_GB_ROW_WAVES 4
#endif
constexpr int kGbRowWaves = AGBNP_GB_ROW_WAVES;
constexpr int row_waves(int kind) { return kind == 2 ? kGbRowWaves : kRowWaves; }  // (kind: RowKind)
constexpr int kRowSliceMax = 512;
constexpr int kChainParts = 4;  // waves (list parts) per group of chain-rule rows
constexpr int kBornParts = 2;   // ... per group of Born rows
constexpr int kGbParts = 2;     // ... per group of GB rows (fast mode)

// Optional per-kernel timing: an event is recorded on the evaluation's stream in front of every kernel
// (and one after the last); durations are read back after the stream has been synchronised.
struct Timeline {
  bool enabled = false;
  std::vector<hipEvent_t> events;
  std::vector<int> ids;  // kernel id that FOLLOWS events[k]; -1 = end of an evaluation
  size_t used = 0;
  hipError_t mark(int kernel_id, hipStream_t st) {
    if (!enabled) return hipSuccess;
    if (used == events.size()) {
      hipEvent_t e;
      hipError_t rc = hipEventCreate(&e);
      if (rc != hipSuccess) return rc;
      events.push_back(e);
      ids.push_back(-1);
    }
    ids[used] = kernel_id;
    return hipEventRecord(events[used++], st);
  }
};

hipError_t launch_prep(const PairArgs& P, hipStream_t st, Timeline* tl);
hipError_t launch_masks(const PairArgs& P, hipStream_t st, Timeline* tl);  // five-launch mode: the neighbour masks alone (with their skin) + their reference positions
hipError_t launch_pair_stages(const PairArgs& P, double* energy_out, double* components, hipStream_t st, Timeline* tl);
// mask_tiles: version 0 in the five-launch mode -- the renewal of the level-2 neighbour masks rides at the tail of this launch
hipError_t launch_outputs(const PairArgs& P, int version, double* force_out, double* energy_out, double* components, hipStream_t st,
                          Timeline* tl, bool mask_tiles = false);

}  // namespace agbnp
