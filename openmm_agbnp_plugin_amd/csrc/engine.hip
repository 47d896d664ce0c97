// Host engine + C ABI (include/agbnp_hip.h) of the gfx950 AGBNP force path.
//
// Mirrors the life cycle of the reference's platform kernel
// (platforms/reference/src/ReferenceAGBNPKernels.cpp): initialize() :58-137 -> agbnp_hip_create,
// execute() :139-149 -> agbnp_hip_execute_{host,device}, copyParametersToContext() :1796-1815 ->
// agbnp_hip_update_parameters.  All device work of one evaluation is enqueued on one stream:
//
//   k_prep -> k_tree_cavity -> [k_born_tiles -> k_gb_tiles -> k_dborn_tiles -> k_tree_pseudo] -> k_outputs
//
// (bracketed part only for version 1).  There is no CPU fallback: without a HIP device every entry
// point that computes fails with AGBNP_HIP_ERR_DEVICE.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/agbnp_hip.h"
#include "adapter_kernels.h"
#include "agbnp_common.h"
#include "i4_tables.h"
#include "pair_kernels.h"
#include "tree_kernels.h"

namespace agbnp {
size_t tree_variant_lds_bytes(int variant);
size_t tree_variant_scratch_bytes(int variant);
int tree_variant_node_cap(int variant);
int tree_variant_atom_cap(int variant);
int tree_variant_wgs_per_cu(int variant);
hipError_t launch_tree_cavity(int variant, int global_grid, int slots, const TreeArgs& A, hipStream_t st);
hipError_t launch_tree_pseudo(int variant, int global_grid, int slots, const TreeArgs& A, hipStream_t st);
hipError_t launch_tree_cavity_five(int variant, int slots, const TreeArgs& A, const PairArgs& P, hipStream_t st);
}  // namespace agbnp

using namespace agbnp;

namespace {

thread_local std::string g_create_error;  // message of the last failed agbnp_hip_create / host_tables on THIS thread

template <class T>
struct DevBuf {
  T* p = nullptr;
  size_t count = 0;
  hipError_t alloc(size_t n) {
    release();
    count = n;
    if (n == 0) return hipSuccess;
    return hipMalloc(reinterpret_cast<void**>(&p), n * sizeof(T));
  }
  // Same size as before: the data is replaced IN PLACE and the device address stays what it was -- kernel
  // arguments frozen into a captured HIP graph keep pointing at live memory across agbnp_hip_update_parameters.
  hipError_t upload(const std::vector<T>& v) {
    if (p == nullptr || count != v.size()) {
      hipError_t e = alloc(v.size());
      if (e != hipSuccess) return e;
    }
    if (v.empty()) return hipSuccess;
    return hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    count = 0;
  }
  ~DevBuf() { release(); }
};

constexpr int kGlobalVariant = 4;  // tree_kernels.hip: 0 (432, 64), 1 (512, 64), 2 (1024, 128), 3 (2048, 256) in LDS, 4 in HBM scratch
constexpr int kGlobalGrid = 256;  // persistent workgroups of the global-scratch variant

}  // namespace

struct agbnp_hip_context {
  int n = 0, nh = 0, version = 1, method = 0, device = 0;
  double cutoff = 1.0;
  std::string err;
  // host copies of the parameters (reference: ReferenceAGBNPKernels.h:60-91)
  std::vector<double> r_vdw, gamma, alpha, charge;
  std::vector<int> ish, a2h, h2a;
  I4TableSet lut;
  int variant = 0;
  hipStream_t stream = nullptr;
  // second stream + fork/join events: the enlarged-radius cavity pass runs underneath the pair kernels

  // static device data
  DevBuf<int> d_a2h, d_h2a, d_status, d_order, d_ftime, d_rows, d_forest, d_pack_items, d_gb_items, d_db_items, d_pslot, d_ctx_slot;
  int cus = 256;
  DevBuf<unsigned long long> d_nbmask;
  DevBuf<double> d_charge, d_alpha, d_inv_rvdw, d_inv_vol_a;
  DevBuf<double> d_heavy;  // [kHvRows][hstride]: every per-heavy-atom double array of the tree and pair stages (tree_kernels.h)
                           // (five-launch mode: TWO such tables, see below)
  size_t hstride = 64;
  // ---- five-launch mode (the default for version 1 since round 5; AGBNP_HIP_FIVE_LAUNCHES=0 keeps the k_prep launch; the LDS
  //      stores (variants 0-3), the FP64 row form of the pair stages; the caller's FP64 [3n] positions or -- round 6 -- an OpenMM context's
  //      posq; inside stream captures the device names the evaluation's set): no k_prep launch.  The trailing workgroups of the
  //      cavity launch do k_prep's per-atom work; what the tree launch needs clean BEFORE it starts -- its accumulators, the
  //      subtree shapes, the per-evaluation status words -- exists twice and alternates with the evaluation's parity (the
  //      trailing workgroups clear the other set); the tree reads the caller's positions itself; the level-2 neighbour masks
  //      carry a skin and are laid down anew ON THE DEVICE, by tiles at the tail of the Born-rows launch, when a heavy atom has
  //      used a quarter of it (beyond half the evaluation is void: a jump of more than 0.04 nm costs one withheld evaluation,
  //      include/agbnp_hip.h); a launch of their own (k_masks) lays them down for a fresh context and after an OpenMM context
  //      has reordered its atoms
  bool five = false;           // asked for
  bool five_active = false;    // ... and in effect (switched off for good by the HBM-resident store of variant 4, pair stages other than the FP64 row form, the diagnostic pass-1 self volumes)
  int parity = 0;              // of the evaluation whose results the device holds (read back at every harvest)
  int five_evals = 0;          // evaluations enqueued in the mode so far: evaluation k works on set k & 1
  bool five_device = false;    // the device names the set (from the context's first stream capture on: see PairArgs::five)
  bool masks_valid = false;
  double mask_skin = 0.08;     // nm; AGBNP_HIP_MASK_SKIN (0.06 and 0.08 cost the cavity launch the same; 0.04 renews the masks at every other evaluation of the headline's jitter)
  DevBuf<int> d_estatus, d_row_atoms, d_epoch;  // d_epoch[0]: evaluations the device has been through in that mode (its parity names the set)
  DevBuf<double> d_mask_ref;
  double* htable(int p) const { return d_heavy.p + (size_t)p * kHvRows * hstride; }
  double* hrow(int r) const { return htable(five_active ? parity : 0) + (size_t)r * hstride; }
  DevBuf<int2> d_ameta;
  DevBuf<double2> d_lut;
  // per-evaluation device data
  DevBuf<double> d_pbox, d_abox, d_epart;
  int mode = 0;  // AGBNP_HIP_MODE_* bits
  DevBuf<double4> d_aposq, d_prec, d_srec;
  DevBuf<double> d_ys;
  DevBuf<int> d_a2s;
  // row form of the range-limited stages (pair_kernels.hip, k_rows)
  DevBuf<unsigned> d_hperm, d_aperm, d_nlh, d_nla, d_nlg, d_bslice, d_cslice;
  DevBuf<int> d_nlh_count, d_nla_count, d_nlg_count, d_nl_flag, d_nl_nitems;
  DevBuf<unsigned> d_nl_items;
  int nl_items_cap = 0;
  DevBuf<double> d_egb_rows;   // per-wave energy partials of the GB rows
  int rows_policy = -1;        // AGBNP_HIP_ROWS: 0 = the tile kernels everywhere; unset or 1 = the row form wherever it can run (reference
                               // and fast mode; the deterministic and single-precision modes keep the tiles)
  int nlg_stride = 0;
  double row_fill = 1.5;       // AGBNP_HIP_ROW_FILL: the density bound behind the walked part of a list, in protein-interior densities
  int row_boost = 1;           // widens the part of a list that the row launches walk (doubles when a list has outgrown it)
  DevBuf<double> d_nl_ref, d_bw;
  DevBuf<double4> d_rec_h, d_hrow, d_grec, d_hrec;
  DevBuf<double2> d_pw;        // four arrays of nti * ntj * 15 entries: {c0, c1} / {c2, c3} by [screened][screener], the same by [screener][screened]
  int forests_hint = 0;        // forests of the last evaluation the host has read the status of (0: none yet); reset by a fallback packing
  bool fused_outputs = true;   // version 1: the pseudo-volume launch adds the forces itself (AGBNP_HIP_OUTPUT_LAUNCH=1: k_outputs does)
  bool rows_capable = false;   // the buffers above exist
  bool rows_disabled = false;  // a neighbour row outgrew its stride once: the tile kernels from then on
  double skin = 0.1;           // nm; AGBNP_HIP_SKIN
  double row_move = -1.0;      // nm; AGBNP_HIP_ROW_MOVE, read ONCE when the context is created (< 0: half the skin)
  int nlh_stride = 0, nla_stride = 0;  // entries reserved per list part
  DevBuf<int2> d_sizes;
  DevBuf<double> d_born_part, d_born, d_born_fp, d_brw, d_e_atom, d_gbf, d_dbf, d_egb_part, d_components;
  DevBuf<SubtreeHeader> d_hdr;
  DevBuf<unsigned long long> d_node_pool;
  DevBuf<unsigned short> d_pair_pool;
  DevBuf<int> d_atom_pool;
  DevBuf<char> d_scratch;
  // host-API staging
  DevBuf<double> d_pos_in, d_force_tmp, d_energy_tmp;
  std::vector<double> h_force_tmp;

  PairArgs P{};
  TreeArgs T{};
  Timeline timeline;
  double kernel_ms[kKernelCount] = {0};
  long kernel_launches[kKernelCount] = {0};
  int last_status[kStatTotalWords] = {0};
  std::vector<int> withheld;   // evaluations (numbered from the previous finish) that the last finish found withheld
  int withheld_count = 0;
  unsigned generation = 1;     // bumped whenever kernel arguments a captured graph has frozen go stale
  int fallback_parts = 1;      // the packing an overflowed evaluation is repeated on: every subtree shared among this many work items, each
                               // alone in its slot (1, or 4 once a lone item has outgrown the store; never lowered)
  bool heal = true;            // AGBNP_HIP_HEAL=0: a forest that outgrows its store voids the evaluation (rounds 2-5) instead of being built again in smaller sets
  bool split_fit = true;       // AGBNP_HIP_SPLIT_FIT=0: a lone subtree that outgrows the store moves the system to the next variant at once
  int tree_slots[5] = {1280, 1024, 512, 256, 256};  // resident tree workgroups per variant (CUs x workgroups per CU by LDS)
  int slot_cap = 1024;  // work slots of the tree kernels: 4 x subtrees + resident workgroups of the smallest variant
  double last_components[4] = {0, 0, 0, 0};
  std::vector<int> carried;     // withheld evaluations of execute_device harvested by an execute_host call in between (see there)
  int carried_count = 0, carried_seq = 0;
  bool unfinished = false;      // evaluations enqueued by execute_device / execute_openmm since the last finish
  int enqueued = 0;             // ... how many: what agbnp_hip_wait_verdict waits for (the device numbers them the same way)
  int last_pack[4] = {0, 0, 0, 0};  // {level, age, clean replans in a row, plans so far} of the forest packing as of the last harvest
  int* h_status = nullptr;      // pinned, mapped: {evaluations completed, withheld} since the last finish (agbnp_hip_poll)
  // pinned staging of the host-facing paths: what harvest() reads of the device arrives by asynchronous copies in front of
  // ONE stream synchronisation (four blocking reads before); execute_host's positions, forces and energy travel through
  // h_xfer ([3n] in, [3n + 1] out) instead of pageable memory
  struct HostReport {
    int status[kStatTotalWords];
    double components[4];
    int rows[4], pack[4];
    int five[36];  // five-launch mode: the two blocks of per-evaluation status words, then the device's evaluation counter
  };
  HostReport* h_report = nullptr;
  double* h_xfer = nullptr;
  // agbnp_hip_execute_host's short cut: an evaluation that the pinned status words call complete skips the reads of the
  // device (they are diagnostics) and leaves the log running; the reads are caught up with when somebody asks for a
  // diagnostic, when an evaluation is enqueued through a device-resident entry point, and every 1024 evaluations
  // agbnp_hip_execute_openmm without an adapter launch: k_prep reads the context's posq through the particle -> slot and heavy
  // index -> slot maps (d_ctx_slot, d_hslot), built for the atomIndex array at order_ptr and checked on the device in every
  // evaluation; a context that has reordered its atoms voids ONE evaluation (kStatOrderStale), the maps are rebuilt, the
  // caller repeats (AGBNP_HIP_ADAPTER_LAUNCH=1: the adapter launch of rounds 1-2 instead)
  DevBuf<int> d_hslot;
  bool order_valid = false;
  int row_atoms_kind = 0;       // five-launch mode: what d_row_atoms holds for the packing in use -- 0 atom indices (the caller's
                                // [3n] positions), 1 slots of an OpenMM context's order (posq), -1 stale (the context reordered)
  const int* order_ptr = nullptr;
  bool adapter_launch = false;
  int lazy_evals = 0;           // evaluations of execute_host since the log was last read and cleared: the FIRST entries of the
                                // running log (a device-resident entry point that follows counts on from there; nothing is
                                // synchronised for the hand-over, so it is safe inside a graph capture)
  int last_device_seq = 0;      // harvest(): evaluations of the device-resident entry points that the log just read held
  std::vector<void*> user_streams;  // streams the caller has enqueued on since the last finish (drained before parameters change)
  int last_rows[3] = {0, 0, 0};  // {stale flag, builds so far, entries per slice} of the row-form neighbour rows, as of the last harvest
  int row_slice = 0;           // AGBNP_HIP_ROW_SLICE: entries per slice, fixed (0: tuned on the device, see rows_close_evaluation)
  bool have_results = false;
  bool diagnostics = false;

  int fail(int code, const std::string& msg) {
    err = msg;
    return code;
  }
  int hipfail(hipError_t e, const char* what) {
    err = std::string(what) + ": " + hipGetErrorString(e);
    return AGBNP_HIP_ERR_DEVICE;
  }
};

namespace {

#define HIP_TRY(ctx, call)                                  \
  do {                                                      \
    hipError_t e__ = (call);                                \
    if (e__ != hipSuccess) return (ctx)->hipfail(e__, #call); \
  } while (0)

// Conservative squared cutoff of the 2-body overlap search: beyond it no pair of heavy atoms can have
// an unswitched overlap volume above VOLMINA (gaussvol.cpp:60-93 solved for d^2), so the pruned pairs
// would have been rejected by the volume test anyway.
double overlap_search_cutoff2(const std::vector<double>& a_large, const std::vector<double>& v_large) {
  std::vector<std::pair<double, double>> kinds;
  for (size_t i = 0; i < a_large.size(); i++) {
    std::pair<double, double> k(a_large[i], v_large[i]);
    if (std::find(kinds.begin(), kinds.end(), k) == kinds.end()) kinds.push_back(k);
  }
  double best = 0.0;
  for (auto& k1 : kinds)
    for (auto& k2 : kinds) {
      const double df = k1.first * k2.first / (k1.first + k2.first);
      const double pref = k1.second * k2.second * pow(df / kPi, 1.5);
      if (pref > kVolMinA) best = std::max(best, log(pref / kVolMinA) / df);
    }
  return best * (1.0 + 1e-6) + 1e-9;
}

// changed_only: agbnp_hip_update_parameters -- gamma, alpha and charge are all that may change there (radii and the
// hydrogen flags are refused before), so three arrays travel instead of nine
int upload_parameters(agbnp_hip_context* c, bool changed_only = false) {
  const int n = c->n, nh = c->nh;
  if (changed_only && c->d_heavy.p != nullptr) {
    if (c->h_xfer) {  // through the pinned staging of the host-facing paths: three copies in front of one wait
      double* q = c->h_xfer, *a = q + n, *g = a + n;
      for (int i = 0; i < n; i++) q[i] = c->charge[i], a[i] = c->alpha[i];
      for (int h = 0; h < nh; h++) g[h] = c->gamma[c->h2a[h]] / kRadiusIncrement;
      HIP_TRY(c, hipMemcpyAsync(c->d_charge.p, q, sizeof(double) * n, hipMemcpyHostToDevice, c->stream));
      HIP_TRY(c, hipMemcpyAsync(c->d_alpha.p, a, sizeof(double) * n, hipMemcpyHostToDevice, c->stream));
      for (int t = 0; t < (c->five ? 2 : 1) && nh > 0; t++)
        HIP_TRY(c, hipMemcpyAsync(c->htable(t) + (size_t)kHvGam * c->hstride, g, sizeof(double) * nh, hipMemcpyHostToDevice, c->stream));
      HIP_TRY(c, hipStreamSynchronize(c->stream));
      return AGBNP_HIP_OK;
    }
    std::vector<double> gam_cav(nh);
    for (int h = 0; h < nh; h++) gam_cav[h] = c->gamma[c->h2a[h]] / kRadiusIncrement;
    HIP_TRY(c, c->d_charge.upload(c->charge));
    HIP_TRY(c, c->d_alpha.upload(c->alpha));
    for (int t = 0; t < (c->five ? 2 : 1) && nh > 0; t++)
      HIP_TRY(c, hipMemcpy(c->htable(t) + (size_t)kHvGam * c->hstride, gam_cav.data(), sizeof(double) * nh, hipMemcpyHostToDevice));
    return AGBNP_HIP_OK;
  }
  std::vector<double> inv_rvdw(n), inv_vol_h(nh), gam_cav(nh), a_large(nh), v_large(nh), a_vdw(nh), v_vdw(nh);
  const double roffset = kRadiusIncrement;  // versions 0 and 1 (ReferenceAGBNPKernels.cpp:67-70)
  for (int i = 0; i < n; i++) inv_rvdw[i] = 1. / c->r_vdw[i];
  for (int h = 0; h < nh; h++) {
    const int i = c->h2a[h];
    const double rv = c->r_vdw[i];
    const double rl = rv + roffset;
    a_large[h] = kKFC / (rl * rl);
    v_large[h] = 4. * M_PI * pow(rl, 3) / 3.;
    a_vdw[h] = kKFC / (rv * rv);
    v_vdw[h] = 4. * M_PI * pow(rv, 3) / 3.;
    inv_vol_h[h] = 1.0 / v_vdw[h];
    gam_cav[h] = c->gamma[i] / roffset;
  }
  HIP_TRY(c, c->d_charge.upload(c->charge));
  HIP_TRY(c, c->d_alpha.upload(c->alpha));
  HIP_TRY(c, c->d_inv_rvdw.upload(inv_rvdw));
  {  // the heavy table's 1/V row once more, by ATOM (0 for hydrogens): k_prep fills the rows' records without waiting for a
     // heavy index first
    std::vector<double> inv_vol_a(n, 0.0);
    for (int h = 0; h < nh; h++) inv_vol_a[c->h2a[h]] = inv_vol_h[h];
    HIP_TRY(c, c->d_inv_vol_a.upload(inv_vol_a));
  }
  if (c->d_heavy.p == nullptr) {
    c->hstride = ((size_t)std::max(nh, 1) + 63) / 64 * 64;
    const size_t tables = c->five ? 2 : 1;
    HIP_TRY(c, c->d_heavy.alloc(tables * kHvRows * c->hstride));
    HIP_TRY(c, hipMemset(c->d_heavy.p, 0, sizeof(double) * tables * kHvRows * c->hstride));
  }
  auto put = [&](int row, const std::vector<double>& v) {  // in place: the addresses stay valid for captured graphs
    hipError_t e = hipSuccess;
    for (int t = 0; t < (c->five ? 2 : 1) && !v.empty() && e == hipSuccess; t++)
      e = hipMemcpy(c->htable(t) + (size_t)row * c->hstride, v.data(), sizeof(double) * v.size(), hipMemcpyHostToDevice);
    return e;
  };
  HIP_TRY(c, put(kHvInvVol, inv_vol_h));
  HIP_TRY(c, put(kHvGam, gam_cav));
  HIP_TRY(c, put(kHvALarge, a_large));
  HIP_TRY(c, put(kHvVLarge, v_large));
  HIP_TRY(c, put(kHvAVdw, a_vdw));
  HIP_TRY(c, put(kHvVVdw, v_vdw));
  c->T.rcut2 = overlap_search_cutoff2(a_large, v_large);
  return AGBNP_HIP_OK;
}

int ensure_scratch(agbnp_hip_context* c) {
  // topology store: fixed stride per subtree, sized for the current variant
  const size_t nslots = (size_t)c->slot_cap;
  const size_t need_nodes = nslots * (size_t)tree_variant_node_cap(c->variant);
  const size_t need_atoms = nslots * (size_t)tree_variant_atom_cap(c->variant);
  if (c->d_node_pool.count < need_nodes) {
    HIP_TRY(c, c->d_node_pool.alloc(need_nodes));
    c->T.node_pool = c->d_node_pool.p;
    c->generation++;
  }
  const size_t need_pairs = c->variant <= 1 ? 4 * need_nodes : 0;  // membership pairs of the variants up to 512 nodes
  if (c->d_pair_pool.count < need_pairs) {
    HIP_TRY(c, c->d_pair_pool.alloc(need_pairs));
    c->T.pair_pool = c->d_pair_pool.p;
    c->generation++;
  }
  if (c->d_atom_pool.count < need_atoms) {
    HIP_TRY(c, c->d_atom_pool.alloc(need_atoms));
    c->T.atom_pool = c->d_atom_pool.p;
    c->generation++;
  }
  if (c->variant != kGlobalVariant) return AGBNP_HIP_OK;
  const size_t stride = tree_variant_scratch_bytes(kGlobalVariant);
  const size_t need = stride * (size_t)std::min(kGlobalGrid, std::max(c->nh, 1));
  if (c->d_scratch.count < need) {
    HIP_TRY(c, c->d_scratch.alloc(need));
    c->generation++;
  }
  c->T.scratch = c->d_scratch.p;
  c->T.scratch_stride = stride;
  return AGBNP_HIP_OK;
}

// The members of the argument blocks that name one of the two sets of {heavy-atom table, subtree shapes, per-evaluation status
// words}.  Five-launch mode, eager (five == 1): the set of the evaluation about to be enqueued, five_evals & 1 -- the host
// counts.  From a context's first stream capture on (five == 2): set 0, and every kernel moves them to the set the DEVICE's
// count names (PairArgs::epoch, rebase_for_parity in pair_kernels.h).  The device counts in either form, so the two agree.
void apply_parity(agbnp_hip_context* c) {
  PairArgs& P = c->P;
  TreeArgs& T = c->T;
  const bool five = c->five_active;
  const int p = (five && !c->five_device) ? (c->five_evals & 1) : 0;
  const size_t nhp = (size_t)std::max(c->nh, 1);
  auto row = [&](int r) { return c->htable(p) + (size_t)r * c->hstride; };
  P.inv_vol_h = row(kHvInvVol);  // (static rows: the same in both tables)
  P.gam_cav = row(kHvGam);
  P.a_large = row(kHvALarge);
  P.v_large = row(kHvVLarge);
  P.hx = row(kHvX);
  P.hy = row(kHvY);
  P.hz = row(kHvZ);
  P.gx = row(kHvGx);
  P.gy = row(kHvGy);
  P.gz = row(kHvGz);
  P.sv_vdw = row(kHvSvVdw);
  P.sv_large = row(kHvSvLarge);
  T.hv = c->htable(p);
  P.sizes = c->d_sizes.p + (size_t)p * nhp;
  T.sizes = P.sizes;
  P.estatus = five ? c->d_estatus.p + 16 * p : c->d_status.p;
  T.status = P.estatus;  // (the tree kernels only touch words of their own evaluation)
  P.five = T.five = five ? (c->five_device ? 2 : 1) : 0;
  // the counter exists twice, each copy beside what its readers read first (a cold scalar load of its own costs a launch
  // 0.1-0.3 us): the pair launches' in the neighbour rows' flag line, the tree launches' behind the forest counts; the
  // bookkeeping role advances both
  P.epoch = c->d_nl_flag.p ? c->d_nl_flag.p + 3 : c->d_epoch.p;
  P.epoch_tree = c->d_forest.p + (size_t)c->slot_cap + 9;
  T.epoch = P.epoch_tree;
  P.table_doubles = T.table_doubles = (size_t)kHvRows * c->hstride;
  P.sizes_stride = T.sizes_stride = nhp;
  P.hstride = (unsigned)c->hstride;
  // (five == 1: the host names the set the trailing workgroups clear, too; five == 2: rebase_for_parity does)
  P.next_hv = (five && !c->five_device) ? c->htable(1 - p) : nullptr;
  P.next_sizes = (five && !c->five_device) ? c->d_sizes.p + (size_t)(1 - p) * nhp : nullptr;
  P.next_estatus = (five && !c->five_device) ? c->d_estatus.p + 16 * (1 - p) : nullptr;
  P.mask_ref = c->d_mask_ref.p;
  P.mask_move2 = 0.25 * c->mask_skin * c->mask_skin;
  P.row_atoms = five ? c->d_row_atoms.p : nullptr;
  T.row_atoms = P.row_atoms;
  // the masks of that mode reach a skin further than the exact test of the level-2 search does
  const double reach = sqrt(c->T.rcut2) + (five ? c->mask_skin : 0.0);
  P.mask_rcut2 = five ? reach * reach : c->T.rcut2;
}

void wire_args(agbnp_hip_context* c) {
  PairArgs& P = c->P;
  P.n = c->n;
  P.nh = c->nh;
  P.zero_out = nullptr;
  P.a2h = c->d_a2h.p;
  P.h2a = c->d_h2a.p;
  P.charge = c->d_charge.p;
  P.alpha = c->d_alpha.p;
  P.inv_rvdw = c->d_inv_rvdw.p;
  P.inv_vol_h = c->hrow(kHvInvVol);
  P.inv_vol_a = c->d_inv_vol_a.p;
  P.gam_cav = c->hrow(kHvGam);
  P.ameta = c->d_ameta.p;
  P.lut = c->d_lut.p;
  P.nti = c->lut.nscreened;
  P.ntj = c->lut.nscreener;
  P.lut_entries = c->lut.nscreened * c->lut.nscreener * kLutStride;
  P.hx = c->hrow(kHvX);
  P.hy = c->hrow(kHvY);
  P.hz = c->hrow(kHvZ);
  P.aposq = c->d_aposq.p;
  P.pbox = c->d_pbox.p;
  P.a2s = c->d_a2s.p;
  P.prec = c->d_prec.p;
  P.srec = c->d_srec.p;
  P.ys = c->d_ys.p;
  P.abox = c->d_abox.p;
  // The OpenCL platform only defines USE_CUTOFF for a method other than NoCutoff (OpenCLAGBNPKernels.cpp:487,1149-1150): with
  // NoCutoff the fast mode truncates nothing and IS the reference mode (the cutoff distance "will have no effect",
  // AGBNPForce.h); CutoffPeriodic is refused by agbnp_hip_set_mode (no box vectors cross this boundary).
  const bool cut = (c->mode & AGBNP_HIP_MODE_FAST) && c->method != 0;
  P.fast = cut ? 1 : 0;
  P.single = cut && (c->mode & AGBNP_HIP_MODE_SINGLE) ? 1 : 0;
  P.det = (c->mode & AGBNP_HIP_MODE_DETERMINISTIC) ? 1 : 0;
  P.range2 = P.fast ? std::min(kI4MaxA * kI4MaxA, c->cutoff * c->cutoff) : kI4MaxA * kI4MaxA;
  P.gb_cut2 = P.fast ? c->cutoff * c->cutoff : 1e300;
  // far strips (pair_kernels.hip, gb_strip): only systems with more than 8192 atoms can have blocks some 4 nm apart in
  // numbers that pay for the test (1dwc, 4152 atoms: none; 2clr, 5983: 3-5 %); AGBNP_HIP_GB_FAR = 0 / 1 forces it (tests)
  P.gb_far = !P.fast && (getenv("AGBNP_HIP_GB_FAR") ? atoi(getenv("AGBNP_HIP_GB_FAR")) != 0 : c->n > 8192) ? 1 : 0;
  P.pslot = c->d_pslot.p;
  P.nslots = (int)c->d_pslot.count;
  P.nhb = (c->nh + 63) / 64;
  P.cull_first = P.nslots / 64 > 96 ? 1 : 0;  // beyond ~6000 atoms most tiles are further apart than the tables reach
  P.db_items = c->d_db_items.p;
  P.db_items_count = (int)c->d_db_items.count;
  P.gx = c->hrow(kHvGx);
  P.gy = c->hrow(kHvGy);
  P.gz = c->hrow(kHvGz);
  P.sv_vdw = c->hrow(kHvSvVdw);
  P.sv_large = c->hrow(kHvSvLarge);
  P.epart = c->d_epart.p;
  P.status = c->d_status.p;
  {
    void* dev = nullptr;
    P.host_status = (c->h_status && hipHostGetDevicePointer(&dev, c->h_status, 0) == hipSuccess) ? static_cast<volatile int*>(dev) : nullptr;
  }
  P.born_part = c->d_born_part.p;
  P.born = c->d_born.p;
  P.born_fp = c->d_born_fp.p;
  P.brw = c->d_brw.p;
  P.e_atom = c->d_e_atom.p;
  const size_t row = (size_t)c->n;
  P.gb_fx = c->d_gbf.p;
  P.gb_fy = c->d_gbf.p + c->n;
  P.gb_fz = c->d_gbf.p + 2 * (size_t)c->n;
  P.gb_items = c->d_gb_items.p;
  P.gb_items_count = (int)c->d_gb_items.count;
  P.db_fx = c->d_dbf.p;
  P.db_fy = c->d_dbf.p + row;
  P.db_fz = c->d_dbf.p + 2 * row;
  P.db_wu = c->d_dbf.p + 3 * row;
  P.egb_part = c->d_egb_part.p;
  {
    // Row form (reference mode only: the fast mode cuts every stage at the cutoff and the deterministic mode fixes the
    // order of its sums through the tiles' quantized totals)
    const bool wanted = c->rows_policy != 0;  // (AGBNP_HIP_ROWS=0: the tile kernels everywhere)
    // (the single-precision option of the fast mode lives in the GB stage: in the GB rows where they can run, else in the
    // packed-FP32 strips of the tile form)
    const bool gb_rows_possible = P.fast && c->d_nlg.p != nullptr && getenv("AGBNP_HIP_NO_GB_ROWS") == nullptr;
    P.rows_on = c->rows_capable && !c->rows_disabled && c->version == 1 && !P.det && (!P.single || gb_rows_possible) && wanted ? 1 : 0;
    P.gb_rows = P.rows_on && gb_rows_possible ? 1 : 0;
    const double reach = sqrt(P.range2) + c->skin, gb_reach = c->cutoff + c->skin;  // (fast mode: the range-limited stages stop at the cutoff too)
    P.nl_build2 = reach * reach;
    P.nlg_build2 = gb_reach * gb_reach;
    P.nlg = c->d_nlg.p;
    P.nlg_count = c->d_nlg_count.p;
    P.nlg_stride = c->nlg_stride;
    P.nl_items = c->d_nl_items.p;
    P.nl_nitems = c->d_nl_nitems.p;
    P.nl_items_cap = c->nl_items_cap;
    {
      // what the launches walk of a list: the atoms that 1.5 x the density of a protein interior (105 atoms, 52 heavy ones
      // per nm^3) puts within reach + skin of a group of four bonded atoms (0.3 nm across), per part, in slices of 256 --
      // times row_boost after a list has outgrown it
      auto cap = [&](double radius, double density, int parts, int stride) {
        const double r = radius + 0.3, most = c->row_fill * density * (4.0 / 3.0) * M_PI * r * r * r / parts * c->row_boost;
        return std::max(256, (int)std::min(most + 255.0, 1e9) / 256 * 256);
      };
      P.nlh_cap = std::min(cap(reach, 52.0, kBornParts, c->nlh_stride), std::max(c->nlh_stride, 1));
      P.nla_cap = std::min(cap(reach, 105.0, kChainParts, c->nla_stride), std::max(c->nla_stride, 1));
      P.nlg_cap = std::min(cap(gb_reach, 105.0, kGbParts, c->nlg_stride), std::max(c->nlg_stride, 1));
    }
    if (P.gb_rows) {  // the GB rows leave one energy partial per wave
      P.egb_part = c->d_egb_rows.p;
      P.egb_parts = (int)c->d_egb_rows.count;
    } else {
      P.egb_part = c->d_egb_part.p;
      P.egb_parts = (int)c->d_egb_part.count;
    }
    // an atom further than this from where it was at the last build makes the lists stale: half the skin -- or
    // AGBNP_HIP_ROW_MOVE (nm; measurement only: 0 rebuilds the lists at every new geometry at the default skin, which is how
    // bench.py prices a rebuild evaluation)
    const double move = c->row_move >= 0.0 ? std::min(c->row_move, 0.5 * c->skin) : 0.5 * c->skin;
    P.nl_move2 = move * move;
    P.nl_flag = c->d_nl_flag.p;
    P.row_target = c->row_slice > 0 ? 0 : 2 * c->cus;
    P.nl_ref = c->d_nl_ref.p;
    P.hperm = c->d_hperm.p;
    P.aperm = c->d_aperm.p;
    P.hperm_n = (int)c->d_hperm.count;
    P.aperm_n = (int)c->d_aperm.count;
    P.nlh = c->d_nlh.p;
    P.nla = c->d_nla.p;
    P.nlh_count = c->d_nlh_count.p;
    P.nla_count = c->d_nla_count.p;
    P.nlh_stride = c->nlh_stride;
    P.nla_stride = c->nla_stride;
    P.bslice = c->d_bslice.p;
    P.cslice = c->d_cslice.p;
    P.rec_h = c->d_rec_h.p;
    P.hrow = c->d_hrow.p;
    P.bw = c->d_bw.p;
    P.grec = c->d_grec.p;
    P.hrec = c->d_hrec.p;
    const size_t tab = (size_t)c->lut.nscreened * c->lut.nscreener * (kI4Nodes - 1);
    P.pw_a = c->d_pw.p;
    P.pw_b = c->d_pw.p ? c->d_pw.p + tab : nullptr;
    P.pwt_a = c->d_pw.p ? c->d_pw.p + 2 * tab : nullptr;
    P.pwt_b = c->d_pw.p ? c->d_pw.p + 3 * tab : nullptr;
  }

  TreeArgs& T = c->T;
  T.nh = c->nh;
  T.hv = c->d_heavy.p;
  T.hstride = (unsigned)c->hstride;
  T.db_wu = c->d_dbf.p + 3 * (size_t)c->n;
  T.want_sv_large = c->diagnostics ? 1 : 0;  // pass-1 self volumes cost extra HBM atomics: opt-in
  T.det = (c->mode & AGBNP_HIP_MODE_DETERMINISTIC) ? 1 : 0;
  T.epart = c->d_epart.p;
  T.hdr = c->d_hdr.p;
  T.node_pool = c->d_node_pool.p;
  T.pair_pool = c->d_pair_pool.p;
  T.atom_pool = c->d_atom_pool.p;
  P.sizes = c->d_sizes.p;
  T.sizes = c->d_sizes.p;
  P.order = c->d_order.p;
  P.pack_items = reinterpret_cast<int2*>(c->d_pack_items.p);
  P.forest_time = c->d_ftime.p;
  P.rows = c->d_rows.p;
  T.rows = c->d_rows.p;
  P.a_large = c->hrow(kHvALarge);
  P.v_large = c->hrow(kHvVLarge);
  P.rcut2 = c->T.rcut2;
  const int nhb_c = (c->nh + 63) / 64;
  P.nb_tiles = nhb_c * (nhb_c + 1) / 2;
  P.nbmask = c->d_nbmask.p;
  T.nbmask = c->d_nbmask.p;
  T.nhb = nhb_c;
  {
    const size_t nhp1 = (size_t)c->slot_cap;
    P.forest_start = c->d_forest.p;
    P.nforests = c->d_forest.p + nhp1 + 1;
    P.cur_nforests = c->d_forest.p + nhp1 + 2;
    P.pack_state = c->d_forest.p + nhp1 + 3;
    P.ncus = c->cus;
    P.tree_slot_cap = c->slot_cap;
    P.tree_node_cap = tree_variant_node_cap(c->variant);
    P.tree_atom_cap = tree_variant_atom_cap(c->variant);
    const bool no_pack = getenv("AGBNP_HIP_NO_PACK") != nullptr;  // tuning knob: one subtree per work slot
    P.pack_enabled = no_pack ? 0 : (getenv("AGBNP_HIP_ITEMS_ALONE") ? 2 : 1);
    const int round_permille = getenv("AGBNP_HIP_ROUND_PERMILLE") ? atoi(getenv("AGBNP_HIP_ROUND_PERMILLE")) : 1000;
    P.round_permille = std::max(100, round_permille);
    P.replan_every = std::max(1, getenv("AGBNP_HIP_REPLAN_EVERY") ? atoi(getenv("AGBNP_HIP_REPLAN_EVERY")) : 16);
    const int split_big = getenv("AGBNP_HIP_SPLIT_BIG") ? atoi(getenv("AGBNP_HIP_SPLIT_BIG")) : 3;
    const int split_permille = getenv("AGBNP_HIP_SPLIT_PERMILLE") ? atoi(getenv("AGBNP_HIP_SPLIT_PERMILLE")) : 550;
    // a full device has slot_cap = 2 x subtrees work slots: more parts per subtree than that could plan more work items
    // than forest_start / order / the topology pools hold
    P.split_big = std::min(std::min(4, c->slot_cap / std::max(c->nh, 1)), std::max(1, split_big));
    P.split_big = std::max(1, P.split_big);
    P.split_permille = std::max(50, split_permille);
    c->split_fit = !(getenv("AGBNP_HIP_SPLIT_FIT") && atoi(getenv("AGBNP_HIP_SPLIT_FIT")) == 0) && c->slot_cap >= 4 * std::max(c->nh, 1);
    P.split_fit = c->split_fit ? 1 : 0;
    // (the tree launches' word: bit 0 the above, bit 1 = forests that outgrow their store are healed inside the launch -- round 6;
    // AGBNP_HIP_HEAL=0: they void the evaluation as in rounds 2-5, for the tests of the withheld-evaluation protocol and A/B runs)
    c->heal = !(getenv("AGBNP_HIP_HEAL") && atoi(getenv("AGBNP_HIP_HEAL")) == 0);
    T.split_fit = P.split_fit | (c->heal ? 2 : 0);
    T.packing = c->d_forest.p;
    T.slot_cap = c->slot_cap;
  }
  T.scratch = c->d_scratch.p;
  T.scratch_stride = tree_variant_scratch_bytes(kGlobalVariant);
  apply_parity(c);
}

int upload_identity_packing(agbnp_hip_context* c);

int harvest(agbnp_hip_context* c, int* repeat, hipStream_t st);
// reads of the device that agbnp_hip_execute_host's short cut has put off (its stream is idle: every call ends with a wait)
int catch_up(agbnp_hip_context* c) {
  if (c->lazy_evals == 0) return AGBNP_HIP_OK;
  int none = 0;
  return harvest(c, &none, c->stream);  // (every one of them was complete: nothing to repeat)
}

void note_stream(agbnp_hip_context* c, void* stream) {  // a caller's stream with work of this context on it
  c->unfinished = true;
  if (stream && std::find(c->user_streams.begin(), c->user_streams.end(), stream) == c->user_streams.end()) c->user_streams.push_back(stream);
}

// Buffers of the row form of the range-limited stages (k_rows): candidate orders sorted by type, neighbour rows at a fixed
// stride, the power-form spline coefficients.  Systems it does not take (version 0, more radius types than the per-wave
// table slices hold, more particles than the row buffers are sized for) simply keep the tile kernels.
int allocate_rows(agbnp_hip_context* c) {
  const int n = c->n, nh = c->nh;
  constexpr int kRowCap = 3072;      // entries per row (part): no protein holds that many heavy atoms within 2.1 nm of one point
  constexpr int kMaxTypes = 255;     // a row's type is one byte of its group's slice word
  constexpr size_t kMaxTableBytes = 40 * 1024;  // the power-form table lives in LDS whole (1dwc: 8 x 6 types, 23 KB)
  constexpr int kMaxParticles = 65536;
  const char* want = getenv("AGBNP_HIP_ROWS");
  c->rows_policy = want == nullptr ? -1 : (atoi(want) != 0 ? 1 : 0);
  if (c->version != 1 || nh == 0 || n > kMaxParticles || c->rows_policy == 0) return AGBNP_HIP_OK;
  if (c->lut.nscreened > kMaxTypes || c->lut.nscreener > kMaxTypes) return AGBNP_HIP_OK;
  if ((size_t)c->lut.nscreened * c->lut.nscreener * (kI4Nodes - 1) * 2 * sizeof(double2) > kMaxTableBytes) return AGBNP_HIP_OK;
  if (getenv("AGBNP_HIP_SKIN")) c->skin = std::min(1.0, std::max(0.0, atof(getenv("AGBNP_HIP_SKIN"))));
  if (getenv("AGBNP_HIP_ROW_MOVE")) c->row_move = std::max(atof(getenv("AGBNP_HIP_ROW_MOVE")), 0.0);
  if (getenv("AGBNP_HIP_ROW_SLICE")) c->row_slice = atoi(getenv("AGBNP_HIP_ROW_SLICE"));
  if (getenv("AGBNP_HIP_ROW_FILL")) c->row_fill = std::max(0.01, atof(getenv("AGBNP_HIP_ROW_FILL")));  // (tests: force the walk to widen)
  auto sorted_by_type = [&](int count, auto type_of) {
    std::vector<unsigned> v;
    for (int k = 0; k < count; k++) v.push_back((unsigned)k | ((unsigned)type_of(k) << 24));
    std::stable_sort(v.begin(), v.end(), [](unsigned a, unsigned b) { return (a >> 24) < (b >> 24); });
    while (v.size() % 64 != 0) v.push_back(~0u);
    return v;
  };
  const std::vector<unsigned> hperm = sorted_by_type(nh, [&](int h) { return c->lut.type_screener[c->h2a[h]]; });
  const std::vector<unsigned> aperm = sorted_by_type(n, [&](int a) { return c->lut.type_screened[a]; });
  HIP_TRY(c, c->d_hperm.upload(hperm));
  HIP_TRY(c, c->d_aperm.upload(aperm));
  // a list part takes every kBornParts-th (kChainParts-th) chunk of 64 candidates: it can hold all of them, up to the cap
  auto part_stride = [&](size_t candidates, int parts) { return std::max(128, std::min(64 * (int)((candidates / 64 + parts - 1) / parts), kRowCap)); };
  static_assert(kRowCap % 256 == 0, "a list is walked in slices of 256 entries");
  c->nlh_stride = part_stride(hperm.size(), kBornParts);
  c->nla_stride = part_stride(aperm.size(), kChainParts);
  // GB rows (fast mode; the cutoff is the force's and fixed for the life of the context): a list holds the atoms within
  // cutoff + skin of a group of four bonded atoms -- at most what twice the density of a protein interior (~105 atoms per
  // nm^3) puts into that sphere, whatever the size of the system
  {
    const double r = c->cutoff + c->skin + 0.3;
    const double most = 2.0 * 105.0 * (4.0 / 3.0) * M_PI * r * r * r / kGbParts;
    c->nlg_stride = c->method != 0 && c->cutoff > 0.0 && c->cutoff < 3.0
                        ? std::max(256, std::min(part_stride(aperm.size(), kGbParts), (int)((most + 255) / 256) * 256)) : 0;
  }
  if (getenv("AGBNP_HIP_ROW_STRIDE")) {  // (tests: force an overflow)
    c->nlh_stride = c->nla_stride = std::max(128, atoi(getenv("AGBNP_HIP_ROW_STRIDE")));
    if (c->nlg_stride) c->nlg_stride = c->nlh_stride;
  }
  const size_t born_lists = (size_t)((n + kRowGroup - 1) / kRowGroup) * kBornParts, chain_lists = (size_t)((nh + kRowGroup - 1) / kRowGroup) * kChainParts;
  HIP_TRY(c, c->d_nlh.alloc(born_lists * c->nlh_stride));
  HIP_TRY(c, c->d_nla.alloc(chain_lists * c->nla_stride));
  HIP_TRY(c, hipMemset(c->d_nlh.p, 0, sizeof(unsigned) * born_lists * c->nlh_stride));  // (entries beyond a list's length are read: valid indices)
  HIP_TRY(c, hipMemset(c->d_nla.p, 0, sizeof(unsigned) * chain_lists * c->nla_stride));
  {
    std::vector<unsigned> bs((n + kRowGroup - 1) / kRowGroup, 0u), cs((nh + kRowGroup - 1) / kRowGroup, 0u);
    for (int a = 0; a < n; a++) bs[a / kRowGroup] |= (unsigned)c->lut.type_screened[a] << (8 * (a % kRowGroup));
    for (int h = 0; h < nh; h++) cs[h / kRowGroup] |= (unsigned)c->lut.type_screener[c->h2a[h]] << (8 * (h % kRowGroup));
    HIP_TRY(c, c->d_bslice.upload(bs));
    HIP_TRY(c, c->d_cslice.upload(cs));
  }
  if (c->nlg_stride > 0) {
    const size_t gb_lists = (size_t)((n + kRowGroup - 1) / kRowGroup) * kGbParts;
    HIP_TRY(c, c->d_nlg.alloc(gb_lists * c->nlg_stride));
    HIP_TRY(c, hipMemset(c->d_nlg.p, 0, sizeof(unsigned) * gb_lists * c->nlg_stride));
    HIP_TRY(c, c->d_nlg_count.alloc(gb_lists));
    HIP_TRY(c, hipMemset(c->d_nlg_count.p, 0, sizeof(int) * gb_lists));
    const size_t waves = (gb_lists + 7) / 8 * 8 * (size_t)((c->nlg_stride + 255) / 256);  // one energy partial per wave of the GB rows
    HIP_TRY(c, c->d_egb_rows.alloc(waves));
    HIP_TRY(c, hipMemset(c->d_egb_rows.p, 0, sizeof(double) * waves));
  }
  {
    // work items: at most every slice of every list of the largest kind
    const size_t gb_lists = c->nlg_stride > 0 ? (size_t)((n + kRowGroup - 1) / kRowGroup) * kGbParts : 0;
    const size_t most = std::max(std::max(born_lists * ((c->nlh_stride + 255) / 256), chain_lists * ((c->nla_stride + 255) / 256)),
                                 gb_lists * ((c->nlg_stride + 255) / 256));
    c->nl_items_cap = (int)std::min<size_t>(most + 8, 1u << 30);
    HIP_TRY(c, c->d_nl_items.alloc((size_t)6 * c->nl_items_cap));
    HIP_TRY(c, hipMemset(c->d_nl_items.p, 0, sizeof(unsigned) * 6 * (size_t)c->nl_items_cap));
    HIP_TRY(c, c->d_nl_nitems.alloc(6));
    HIP_TRY(c, hipMemset(c->d_nl_nitems.p, 0, sizeof(int) * 6));
  }
  HIP_TRY(c, c->d_nlh_count.alloc(born_lists));
  HIP_TRY(c, c->d_nla_count.alloc(chain_lists));
  HIP_TRY(c, hipMemset(c->d_nlh_count.p, 0, sizeof(int) * born_lists));
  HIP_TRY(c, hipMemset(c->d_nla_count.p, 0, sizeof(int) * chain_lists));
  // {stale: the first evaluation builds the rows; builds so far; entries per slice}
  const std::vector<int> flag = {1, 0, c->row_slice > 0 ? std::min(std::max(c->row_slice, kRowSlice), kRowSliceMax) / 64 * 64 : kRowSlice, 0};
  HIP_TRY(c, c->d_nl_flag.upload(flag));
  HIP_TRY(c, c->d_nl_ref.alloc(3 * (size_t)n));
  HIP_TRY(c, hipMemset(c->d_nl_ref.p, 0xff, sizeof(double) * 3 * (size_t)n));  // NaN: every atom has "moved"
  HIP_TRY(c, c->d_bw.alloc(n));
  HIP_TRY(c, c->d_rec_h.alloc(nh));
  HIP_TRY(c, c->d_hrow.alloc(nh));
  HIP_TRY(c, c->d_grec.alloc(n));
  HIP_TRY(c, c->d_hrec.alloc(nh));
  HIP_TRY(c, hipMemset(c->d_hrec.p, 0, sizeof(double4) * nh));
  HIP_TRY(c, hipMemset(c->d_bw.p, 0, sizeof(double) * n));
  HIP_TRY(c, hipMemset(c->d_grec.p, 0, sizeof(double4) * n));
  // Power form of the natural cubic spline on interval k (t in [0, 1)): S = c0 + c1 t + c2 t^2 + c3 t^3 with the same
  // operations the tile kernels use on the knots {y, z = y2 dr^2 / 6} (spline_cubic in pair_kernels.hip)
  const int nti = c->lut.nscreened, ntj = c->lut.nscreener, ni = kI4Nodes - 1;
  const size_t tab = (size_t)nti * ntj * ni;
  const double dr = kI4MaxA / (kI4Nodes - 1);
  std::vector<double2> pw(4 * tab);
  for (int ti = 0; ti < nti; ti++)
    for (int tj = 0; tj < ntj; tj++)
      for (int k = 0; k < ni; k++) {
        const size_t o = ((size_t)ti * ntj + tj) * kI4Nodes + k;
        const double y0 = c->lut.y[o], y1 = c->lut.y[o + 1], z0 = c->lut.y2[o] * dr * dr / 6.0, z1 = c->lut.y2[o + 1] * dr * dr / 6.0;
        const double2 ca = make_double2(y0, (y1 - y0) - std::fma(2.0, z0, z1)), cb = make_double2(3.0 * z0, z1 - z0);
        const size_t by_screened = ((size_t)ti * ntj + tj) * ni + k, by_screener = ((size_t)tj * nti + ti) * ni + k;
        pw[by_screened] = ca;
        pw[tab + by_screened] = cb;
        pw[2 * tab + by_screener] = ca;
        pw[3 * tab + by_screener] = cb;
      }
  HIP_TRY(c, c->d_pw.upload(pw));
  c->rows_capable = true;
  return AGBNP_HIP_OK;
}

int allocate_work(agbnp_hip_context* c) {
  const int n = c->n, nh = c->nh;
  const size_t nhp = std::max(nh, 1);
  const int nblk = (n + 63) / 64;
  {
    // work items of the symmetric GB tile kernel: one workgroup per tile, off-diagonal tiles first
    if (nblk > 4095) return c->fail(AGBNP_HIP_ERR_CAPACITY, "more than 262080 particles are not supported by the tile index encoding");
    // away from the diagonal: strips of two i blocks (2p, 2p + 1) against one j block (flag bit 24, see gb_strip);
    // around it: single 64 x 64 tiles
    constexpr int kStrip = 1 << 24;
    std::vector<int> items;
    items.reserve((size_t)nblk * nblk / 2 + 4);
    const bool strips = getenv("AGBNP_HIP_NO_GB_STRIPS") == nullptr;
    if (strips) {
      for (int p2 = 0; 2 * p2 + 1 < nblk; p2++)
        for (int J = 2 * p2 + 2; J < nblk; J++) items.push_back((2 * p2) | (J << 12) | kStrip);
      for (int p2 = 0; 2 * p2 + 1 < nblk; p2++) items.push_back((2 * p2) | ((2 * p2 + 1) << 12));
    } else {
      for (int I = 0; I < nblk; I++)
        for (int J = I + 1; J < nblk; J++) items.push_back(I | (J << 12));
    }
    for (int I = 0; I < nblk; I++) items.push_back(I | (I << 12));
    HIP_TRY(c, c->d_gb_items.upload(items));
    c->P.egb_parts = (int)items.size();
  }
  {
    // pair order of the chain-rule stage: heavy atoms, padding, hydrogens, padding (blocks of 64 slots), and its
    // work items: symmetric heavy x heavy tiles first (two look-ups per pair), then the heavy x H tiles
    const int nhb = (nh + 63) / 64, nlb = (n - nh + 63) / 64;
    std::vector<int> pslot((size_t)(nhb + nlb) * 64, -1);
    for (int h = 0; h < nh; h++) pslot[h] = c->h2a[h];
    int k = nhb * 64;
    for (int i = 0; i < n; i++)
      if (c->a2h[i] < 0) pslot[k++] = i;
    if (pslot.empty()) pslot.assign(64, -1);
    HIP_TRY(c, c->d_pslot.upload(pslot));
    std::vector<int> a2s((size_t)std::max(n, 1), 0);
    for (size_t sl = 0; sl < pslot.size(); sl++)
      if (pslot[sl] >= 0) a2s[pslot[sl]] = (int)sl;
    HIP_TRY(c, c->d_a2s.upload(a2s));
    HIP_TRY(c, c->d_prec.alloc(pslot.size()));
    HIP_TRY(c, c->d_srec.alloc(pslot.size()));
    HIP_TRY(c, c->d_ys.alloc(pslot.size()));
    HIP_TRY(c, hipMemset(c->d_prec.p, 0, sizeof(double4) * pslot.size()));
    HIP_TRY(c, hipMemset(c->d_srec.p, 0, sizeof(double4) * pslot.size()));
    HIP_TRY(c, hipMemset(c->d_ys.p, 0, sizeof(double) * pslot.size()));
    // Work items, heaviest first: diagonal and heavy x heavy tiles (two look-ups per pair), then heavy x H.  The
    // launch is one round (every workgroup resident at once) and workgroup b starts on CU b mod (number of CUs), so
    // the sorted tiles are dealt over the CUs in serpentine order: every CU gets the same mix of heavy and light ones.
    std::vector<int> sorted;
    for (int I = 0; I < nhb; I++) sorted.push_back(I | (I << 12));
    for (int I = 0; I < nhb; I++)
      for (int J = I + 1; J < nhb; J++) sorted.push_back(I | (J << 12));
    for (int I = 0; I < nhb; I++)
      for (int J = nhb; J < nhb + nlb; J++) sorted.push_back(I | (J << 12));
    std::vector<int> items(sorted.size());
    {
      const size_t width = (size_t)std::max(c->cus, 1);
      for (size_t p = 0; p < sorted.size(); p++) {
        const size_t row = p / width, col = p % width;
        const size_t row_len = std::min(width, sorted.size() - row * width);
        items[row * width + ((row & 1) ? row_len - 1 - col : col)] = sorted[p];
      }
    }
    if (items.empty()) items.push_back(0);
    HIP_TRY(c, c->d_db_items.upload(items));
    if (nh == 0) c->d_db_items.count = 0;
  }

  {
    int rc = allocate_rows(c);
    if (rc != AGBNP_HIP_OK) return rc;
  }
  HIP_TRY(c, c->d_status.alloc(kStatTotalWords));
  HIP_TRY(c, hipMemset(c->d_status.p, 0, sizeof(int) * kStatTotalWords));
  if (hipHostMalloc(reinterpret_cast<void**>(&c->h_status), 4 * sizeof(int), hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess)  // (fine-grained: the host sees the device's writes mid-stream)
    c->h_status[0] = c->h_status[1] = c->h_status[2] = c->h_status[3] = 0;
  else
    c->h_status = nullptr;  // (agbnp_hip_poll then reports "unknown")
  {
    // level-2 neighbour search: tiles of 64x64 heavy atoms (I <= J), one 64-bit mask per (atom, block)
    const int nhb = (nh + 63) / 64;
    if (nhb > 4095) return c->fail(AGBNP_HIP_ERR_CAPACITY, "more than 262080 heavy atoms are not supported by the tile index encoding");
    const size_t words = (size_t)nhb * nhb * 64;
    HIP_TRY(c, c->d_nbmask.alloc(std::max<size_t>(words, 64)));
    HIP_TRY(c, hipMemset(c->d_nbmask.p, 0, sizeof(unsigned long long) * std::max<size_t>(words, 64)));
  }
  // up to four work items per subtree (shared subtrees), plus a launch's worth of slots: the spare slots that k_tree_cavity heals an
  // overgrown forest into are numbered from max(forests, forest workgroups of the launch) on, and at most 4 nh sets exist in all
  c->slot_cap = 4 * std::max(nh, 1) + c->tree_slots[0];
  const size_t nslots = (size_t)c->slot_cap;
  HIP_TRY(c, c->d_epart.alloc(2 * nslots));
  HIP_TRY(c, hipMemset(c->d_epart.p, 0, sizeof(double) * 2 * nslots));
  HIP_TRY(c, c->d_aposq.alloc(n));
  HIP_TRY(c, c->d_pbox.alloc(6 * c->d_pslot.count / 64));
  HIP_TRY(c, c->d_abox.alloc(6 * (size_t)nblk));
  HIP_TRY(c, c->d_sizes.alloc((c->five ? 2 : 1) * nhp));
  HIP_TRY(c, hipMemset(c->d_sizes.p, 0, sizeof(int2) * (c->five ? 2 : 1) * nhp));
  if (c->five) {
    HIP_TRY(c, c->d_estatus.alloc(2 * 16));
    HIP_TRY(c, hipMemset(c->d_estatus.p, 0, sizeof(int) * 2 * 16));  // (fast mode + single keep their own Born rows: no mask tiles there, see five_active)
    static_assert(kStatEvalWords <= 16, "a parity's block of per-evaluation status words");
    HIP_TRY(c, c->d_epoch.upload(std::vector<int>(4, 0)));
    HIP_TRY(c, c->d_mask_ref.upload(std::vector<double>(3 * nhp, std::nan(""))));
    HIP_TRY(c, c->d_row_atoms.alloc((size_t)kMaxItems * nslots));
    HIP_TRY(c, hipMemset(c->d_row_atoms.p, 0, sizeof(int) * kMaxItems * nslots));
  }
  HIP_TRY(c, c->d_born_part.alloc((size_t)n));
  HIP_TRY(c, c->d_born.alloc(n));
  HIP_TRY(c, c->d_born_fp.alloc(n));
  HIP_TRY(c, c->d_brw.alloc(n));
  HIP_TRY(c, c->d_e_atom.alloc(n));
  HIP_TRY(c, c->d_gbf.alloc(3 * (size_t)n));
  HIP_TRY(c, c->d_dbf.alloc(4 * (size_t)n));
  HIP_TRY(c, c->d_egb_part.alloc(c->P.egb_parts));
  HIP_TRY(c, c->d_components.alloc(4));
  {
    int rc = upload_identity_packing(c);  // the first evaluation: nothing is known about the tree yet
    if (rc != AGBNP_HIP_OK) return rc;
  }
  HIP_TRY(c, c->d_hdr.alloc(nslots));
  HIP_TRY(c, hipMemset(c->d_hdr.p, 0, sizeof(SubtreeHeader) * nslots));
  HIP_TRY(c, c->d_pos_in.alloc(3 * (size_t)n));
  HIP_TRY(c, c->d_ctx_slot.alloc(std::max(n, 1)));
  HIP_TRY(c, hipMemset(c->d_ctx_slot.p, 0, sizeof(int) * std::max(n, 1)));
  HIP_TRY(c, c->d_hslot.alloc(std::max(c->nh, 1)));
  HIP_TRY(c, hipMemset(c->d_hslot.p, 0, sizeof(int) * std::max(c->nh, 1)));
  c->adapter_launch = getenv("AGBNP_HIP_ADAPTER_LAUNCH") != nullptr && atoi(getenv("AGBNP_HIP_ADAPTER_LAUNCH")) != 0;
  HIP_TRY(c, c->d_force_tmp.alloc(3 * (size_t)n + 1));  // (+ the energy of agbnp_hip_execute_host)
  HIP_TRY(c, c->d_energy_tmp.alloc(1));
  HIP_TRY(c, hipMemset(c->d_force_tmp.p, 0, sizeof(double) * (3 * (size_t)n + 1)));
  HIP_TRY(c, hipMemset(c->d_energy_tmp.p, 0, sizeof(double)));
  c->h_force_tmp.resize(3 * (size_t)n + 1);
  const bool pinned = getenv("AGBNP_HIP_NO_PINNED_STAGING") == nullptr;  // (tests: the pageable fall-back of the host-facing paths)
  if (!pinned || hipHostMalloc(reinterpret_cast<void**>(&c->h_report), sizeof(agbnp_hip_context::HostReport), hipHostMallocDefault) != hipSuccess) c->h_report = nullptr;
  if (!pinned || hipHostMalloc(reinterpret_cast<void**>(&c->h_xfer), sizeof(double) * (6 * (size_t)n + 8), hipHostMallocDefault) != hipSuccess) c->h_xfer = nullptr;
  (void)hipGetLastError();  // (without pinned memory the host-facing paths fall back to pageable transfers)
  return AGBNP_HIP_OK;
}

int enqueue(agbnp_hip_context* c, const double* d_pos, double* d_force, double* d_energy, hipStream_t st) {
  int rc = ensure_scratch(c);
  if (rc != AGBNP_HIP_OK) return rc;
  c->enqueued++;
  c->P.pos = d_pos;
  c->P.tree_node_cap = tree_variant_node_cap(c->variant);  // the packing of the next evaluation is sized for the variant in use
  c->P.tree_atom_cap = tree_variant_atom_cap(c->variant);
  c->P.tree_slots = c->tree_slots[c->variant];
  Timeline* tl = c->timeline.enabled ? &c->timeline : nullptr;
  if (c->five_active) {
    // the mode ends for good where it cannot hold: the 32 768-node store in HBM (variant 4), pair stages other than the FP64 row form (the renewal of the neighbour masks rides in the Born rows' launch), the
    // diagnostic pass-1 self volumes (a kernel instantiation of the six-launch path only).  (A stream capture is fine: the
    // evaluation's parity lives on the device.)
    // Version 0 (round 6: TWO launches -- the cavity launch with its trailing workgroups, the output launch, whose tail carries the
    // masks' renewal) takes the host-named set only: its bookkeeping role, which advances the device's evaluation counter, runs
    // INSIDE the output launch beside the workgroups that would have to read the counter, so a stream capture ends the mode there.
    // (Round 6 too: pair stages other than the FP64 row form -- the tile kernels of the deterministic mode and of AGBNP_HIP_ROWS=0, the
    // single-precision rows of the fast mode -- stay in the mode with the host-named set: the masks' renewal rides at the tail of
    // the GB tile launch / of the single-precision Born rows; a stream capture ends the mode for them as well.)
    const bool host_set_only = c->version == 0 || !c->P.rows_on || c->P.single;
    bool capturing = false;
    if (host_set_only && !c->five_device) {
      hipStreamCaptureStatus cap0 = hipStreamCaptureStatusNone;
      capturing = hipStreamIsCapturing(st, &cap0) == hipSuccess && cap0 != hipStreamCaptureStatusNone;
    }
    if (c->variant > 3 || c->nh <= 0 || c->diagnostics || capturing || (host_set_only && c->five_device)) {
      c->five_active = false;
      c->parity = 0;
      apply_parity(c);
      c->generation++;
    }
  }
  // workgroups of the tree launches: what the device keeps resident for this variant (they take forests from a queue);
  // fewer if there cannot be that many forests
  const int tree_grid = std::max(1, std::min(c->slot_cap, c->tree_slots[c->variant]));
  if (c->five_active) {
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (!c->five_device && hipStreamIsCapturing(st, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone) {
      // a replayed graph freezes its kernel arguments: from here on the kernels take the evaluation's set from the device's own
      // count (which has kept step with the host's so far), for good -- the host cannot count replays
      c->five_device = true;
      c->generation++;
    }
    apply_parity(c);
    c->five_evals++;
    c->T.pos = d_pos;
    c->T.out.h2a = c->d_h2a.p;  // (the forest workgroups find a candidate's atom through it)
    // ... or an OpenMM context's posq (agbnp_hip_execute_openmm, round 6): the POSQ instantiation of the cavity launch reads it at
    // the context's slots; the words beside the rows must then hold SLOTS -- the dealing role writes them for the entry point of
    // the evaluation it runs in, so they are rewritten here when this evaluation comes through the other one, or when the
    // context's order has changed since (one small launch, never in a steady run)
    const int want_kind = c->P.in.posq ? 1 : 0;
    c->T.posq = c->P.in.posq;
    c->T.posq_corr = c->P.in.correction;
    c->T.posq_double = c->P.in.is_double;
    c->T.hslot = c->P.in.hslot;
    if (c->row_atoms_kind != want_kind) {
      HIP_TRY(c, launch_row_atoms(c->slot_cap, c->d_rows.p, want_kind ? c->d_hslot.p : c->d_h2a.p, c->d_row_atoms.p, st));
      c->row_atoms_kind = want_kind;
    }
    if (!c->masks_valid) {  // a fresh context, or an OpenMM context that has reordered its atoms (harvest): lay them down anew
      HIP_TRY(c, launch_masks(c->P, st, tl));
      c->masks_valid = true;
    }
    if (tl) HIP_TRY(c, tl->mark(kKTreeCavity, st));
    HIP_TRY(c, launch_tree_cavity_five(c->variant, tree_grid, c->T, c->P, st));
  } else {
    HIP_TRY(c, launch_prep(c->P, st, tl));
    if (tl) HIP_TRY(c, tl->mark(kKTreeCavity, st));
    HIP_TRY(c, launch_tree_cavity(c->variant, kGlobalGrid, tree_grid, c->T, st));
  }
  if (c->version == 1) {
    HIP_TRY(c, launch_pair_stages(c->P, d_energy, c->d_components.p, st, tl));
    if (tl) HIP_TRY(c, tl->mark(kKTreePseudo, st));
    // the forces leave with the pseudo-volume launch itself (TreeOutputs, tree_kernels.h): no output launch
    TreeOutputs& O = c->T.out;
    // The forces leave with the pseudo-volume launch when that launch is one round of workgroups (1dwc: -1 us, A/B on one
    // box).  With more forests than resident workgroups a workgroup replays several forests in a row, and the next
    // forest's loads queue behind the force atomics of the one before (three adds on one line of the caller's [n][3]
    // buffer retire more slowly than the heavy-atom table's separate rows: lattice of 16.6 k atoms 49 -> 99 us): there the
    // output launch stays.  (No heavy atom: no tree launch to carry them.)
    // (Round 4: the replay of queued forests is pipelined -- the next forest's data are asked for in front of the flush -- and
    // the forces were tried in the launch again: lattice k_tree_pseudo 51 -> 109 us once more.  On gfx9 loads, stores and
    // atomics share one in-order counter, and a wait that crosses the loop's back edge is a wait for everything, the
    // flush's atomics included.  AGBNP_HIP_FUSE_QUEUED=1 keeps the experiment reachable.)
    static const bool fuse_queued = getenv("AGBNP_HIP_FUSE_QUEUED") && atoi(getenv("AGBNP_HIP_FUSE_QUEUED")) != 0;
    // (Round 5: a system with more subtrees than that whose forests nevertheless fit ONE round -- 2clr under the rounds rule of
    // the packing: 3084 subtrees in 1280 forests -- is told by the forest count of the last evaluation the host has seen;
    // a stale hint costs time, never correctness.)
    const bool one_round = c->forests_hint > 0 && c->forests_hint <= c->tree_slots[c->variant];
    const bool fused = c->fused_outputs && c->nh > 0 && (c->nh <= 2 * c->tree_slots[c->variant] || one_round || (fuse_queued && c->variant <= 1));
    O.enabled = fused ? 1 : 0;
    O.n = c->n;
    O.a2h = c->d_a2h.p;
    O.h2a = c->d_h2a.p;
    O.force = c->P.omm.force_fixed ? nullptr : d_force;
    O.force_fixed = c->P.omm.force_fixed;
    O.padded = c->P.omm.padded;
    O.ctx_slot = c->P.omm.ctx_slot;
    O.gb_f = c->P.gb_fx;
    O.db_f = c->P.db_fx;
    O.rows_on = c->P.rows_on;
    O.bw = c->P.bw;
    O.grec = c->P.grec;
    O.hrec = c->P.hrec;
    O.nl_flag = c->P.nl_flag;
    O.nl_nitems = c->P.nl_nitems;
    O.row_target = c->P.row_target;
    O.gb_rows = c->P.gb_rows;
    HIP_TRY(c, launch_tree_pseudo(c->variant, kGlobalGrid, tree_grid, c->T, st));
    if (fused) {
      if (tl) HIP_TRY(c, tl->mark(-1, st));
      return AGBNP_HIP_OK;
    }
  }
  HIP_TRY(c, launch_outputs(c->P, c->version, d_force, d_energy, c->d_components.p, st, tl, c->five_active && c->version == 0));
  return AGBNP_HIP_OK;
}

// one work item per work slot (what a context starts with, and what an overflowed evaluation is repeated on): every subtree
// whole, or -- once a lone subtree has outgrown the store (fallback_parts) -- shared among two or four items
int upload_identity_packing(agbnp_hip_context* c) {
  const size_t parts = (size_t)std::max(1, std::min(c->fallback_parts, std::max(1, c->slot_cap / std::max(c->nh, 1))));
  const size_t nhp = (size_t)std::max(c->nh, 1) * parts, nslots = (size_t)c->slot_cap;
  std::vector<int> ident((size_t)kRowStride * nslots, -1);  // slot s: its one work item, -1 = no item, and the number 1
  for (size_t k = 0; k < nslots; k++) {
    const size_t item = std::min(k, nhp - 1);
    ident[(size_t)kRowStride * k] = (int)(item / parts) | (int)((item % parts) << 24) | (int)((parts - 1) << 26);  // (work_item_root / _part / _parts)
    ident[(size_t)kRowStride * k + kMaxItems] = 1;
  }
  HIP_TRY(c, c->d_rows.upload(ident));
  if (c->five) {  // (five-launch mode: the atom of every item's root, beside the rows)
    std::vector<int> atoms((size_t)kMaxItems * nslots, 0);
    for (size_t k = 0; k < nslots && c->nh > 0; k++) atoms[(size_t)kMaxItems * k] = c->h2a[std::min(k, nhp - 1) / parts];
    HIP_TRY(c, c->d_row_atoms.upload(atoms));
    c->row_atoms_kind = 0;
  }
  HIP_TRY(c, c->d_order.upload(std::vector<int>((size_t)kMaxItems * nslots + 8, 0)));  // (the bookkeeping's working copies)
  HIP_TRY(c, c->d_ftime.upload(std::vector<int>(nslots + 1, 0)));
  if (c->d_pack_items.p == nullptr) HIP_TRY(c, c->d_pack_items.upload(std::vector<int>(2 * nslots + 2, 0)));
  // layout: [0, slots] forest_start, [slots+1] number of forests, [slots+2] the count the running evaluation took,
  // [slots+3] how often a packed forest has overflowed so far (kept), [slots+4] the age of the packing in evaluations
  // (huge: this one is no plan, the next evaluation's bookkeeping plans at once)
  std::vector<int> forest(nslots + 3);  // (+ three persistent words behind it, see below)
  const int nitems = c->nh > 0 ? (int)nhp : 0;
  for (size_t k = 0; k <= nslots; k++) forest[k] = (int)std::min(k, (size_t)nitems);
  forest[nslots + 1] = nitems;
  forest[nslots + 2] = nitems;
  const int no_plan = 1 << 20;
  if (c->d_forest.p == nullptr) {
    forest.push_back(0);
    forest.push_back(no_plan);
    forest.push_back(0);  // [slots+5] clean evaluations in a row since the assumed capacity was last tightened or relaxed
    forest.push_back(0);  // [slots+6] packings planned so far (diagnostic: bench.py counts the plans inside a timed region)
    forest.push_back(0);  // [slots+7] total nodes and
    forest.push_back(0);  // [slots+8] largest subtree of the evaluation the packing in use was planned from (drift trigger)
    forest.push_back(0);  // [slots+9] five-launch mode: the tree launches' copy of the device's evaluation counter (beside the
                          // forest count they read first: the same cache line, no cold round trip of its own)
    forest.push_back(0);  // [slots+10] the packing's `heat`: a leaky count of evaluations with healed forests (packing_role)
    forest.push_back(0);  // [slots+11] ... and `need`: clean evaluations in a row before a tightened level is given back (its memory)
    return c->d_forest.upload(forest) == hipSuccess ? AGBNP_HIP_OK : c->fail(AGBNP_HIP_ERR_DEVICE, "upload of the forest packing failed");
  }
  HIP_TRY(c, hipMemcpy(c->d_forest.p, forest.data(), sizeof(int) * forest.size(), hipMemcpyHostToDevice));
  HIP_TRY(c, hipMemcpy(c->d_forest.p + nslots + 4, &no_plan, sizeof(int), hipMemcpyHostToDevice));
  return AGBNP_HIP_OK;
}

// after the stream is idle: read status + components; react to overflow.  *repeat = number of evaluations since the
// previous harvest whose forces and energy were withheld on the device (the caller must run those again).
int harvest(agbnp_hip_context* c, int* repeat, hipStream_t st) {
  *repeat = 0;
  // what is read of the device: asked for behind everything on the stream, then ONE wait
  if (c->h_report) {
    agbnp_hip_context::HostReport* r = c->h_report;
    HIP_TRY(c, hipMemcpyAsync(r->status, c->d_status.p, sizeof(int) * kStatTotalWords, hipMemcpyDeviceToHost, st));
    if (c->five_active)  // (the words of ONE evaluation live in its parity's block: both blocks come along, the counter says which)
      HIP_TRY(c, hipMemcpyAsync(r->five, c->d_estatus.p, sizeof(int) * 32, hipMemcpyDeviceToHost, st));
    if (c->five_active) HIP_TRY(c, hipMemcpyAsync(r->five + 32, c->P.epoch_tree, sizeof(int), hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipMemcpyAsync(r->components, c->d_components.p, sizeof(double) * 4, hipMemcpyDeviceToHost, st));
    if (c->rows_capable) HIP_TRY(c, hipMemcpyAsync(r->rows, c->d_nl_flag.p, sizeof(int) * 3, hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipMemcpyAsync(r->pack, c->d_forest.p + c->slot_cap + 3, sizeof(int) * 4, hipMemcpyDeviceToHost, st));
    // ... and a new log starts behind the reads, in front of the same wait (an empty log is cleared to what it is)
    HIP_TRY(c, hipMemsetAsync(c->d_status.p + kStatEvalSeq, 0, sizeof(int) * (kStatTotalWords - kStatEvalSeq), st));
  }
  HIP_TRY(c, hipStreamSynchronize(st));
  // per-kernel durations of everything enqueued since the last harvest
  Timeline& tl = c->timeline;
  for (size_t k = 0; k + 1 < tl.used; k++) {
    const int id = tl.ids[k];
    if (id < 0) continue;
    float ms = 0.f;
    HIP_TRY(c, hipEventElapsedTime(&ms, tl.events[k], tl.events[k + 1]));
    c->kernel_ms[id] += ms;
    c->kernel_launches[id]++;
  }
  tl.used = 0;
  if (c->h_report) {
    std::memcpy(c->last_status, c->h_report->status, sizeof(int) * kStatTotalWords);
    if (c->five_active) {
      // (device count: the last evaluation took epoch & 1 BEFORE its bookkeeping role advanced the counter; the host's count is one ahead as well)
      c->parity = c->five_device ? (c->h_report->five[32] + 1) & 1 : (c->five_evals + 1) & 1;
      std::memcpy(c->last_status, c->h_report->five + 16 * c->parity, sizeof(int) * kStatEvalWords);
    }
    std::memcpy(c->last_components, c->h_report->components, sizeof(double) * 4);
    if (c->rows_capable) std::memcpy(c->last_rows, c->h_report->rows, sizeof(int) * 3);
    std::memcpy(c->last_pack, c->h_report->pack, sizeof(int) * 4);
  } else {
    HIP_TRY(c, hipMemcpy(c->last_status, c->d_status.p, sizeof(int) * kStatTotalWords, hipMemcpyDeviceToHost));
    if (c->five_active) {
      int five[33];
      HIP_TRY(c, hipMemcpy(five, c->d_estatus.p, sizeof(int) * 32, hipMemcpyDeviceToHost));
      HIP_TRY(c, hipMemcpy(five + 32, c->P.epoch_tree, sizeof(int), hipMemcpyDeviceToHost));
      c->parity = c->five_device ? (five[32] + 1) & 1 : (c->five_evals + 1) & 1;  // (either count is one ahead of the last evaluation)
      std::memcpy(c->last_status, five + 16 * c->parity, sizeof(int) * kStatEvalWords);
    }
    HIP_TRY(c, hipMemcpy(c->last_components, c->d_components.p, sizeof(double) * 4, hipMemcpyDeviceToHost));
    if (c->rows_capable) HIP_TRY(c, hipMemcpy(c->last_rows, c->d_nl_flag.p, sizeof(int) * 3, hipMemcpyDeviceToHost));
    HIP_TRY(c, hipMemcpy(c->last_pack, c->d_forest.p + c->slot_cap + 3, sizeof(int) * 4, hipMemcpyDeviceToHost));
  }
  const int* s = c->last_status;
  c->withheld.clear();
  c->withheld_count = s[kStatBadCount];
  if (s[kStatEvalSeq] > 0) c->forests_hint = s[kStatForests];  // (what the NEXT evaluation runs on, written by the last one's bookkeeping)
  // the last evaluation's own words say whether the diagnostics on the device are those of a complete evaluation
  c->have_results = s[kStatEvalSeq] > 0 ? !(s[kStatNodeOverflow] | s[kStatAtomOverflow] | s[kStatPackOverflow] | s[kStatOrderStale] | s[kStatRowOverflow]) : c->have_results;
  if (!c->h_report && (s[kStatEvalSeq] != 0 || s[kStatBadCount] != 0))  // start a new log
    HIP_TRY(c, hipMemset(c->d_status.p + kStatEvalSeq, 0, sizeof(int) * (kStatTotalWords - kStatEvalSeq)));
  if (c->h_status) c->h_status[0] = c->h_status[1] = 0;  // (the stream is idle: nothing writes it now)
  const int host_first = c->lazy_evals;  // the log's first entries are execute_host's own (every one of them complete)
  c->last_device_seq = std::max(0, s[kStatEvalSeq] - host_first);
  c->enqueued = 0;  // (the device's running number starts over with the log)
  c->lazy_evals = 0;
  if (c->withheld_count == 0) return AGBNP_HIP_OK;
  for (int k = host_first; k < kStatBadBits && k < s[kStatEvalSeq]; k++)
    if (s[kStatBadBitmap + (k >> 5)] & (1 << (k & 31))) c->withheld.push_back(k - host_first);
  *repeat = c->withheld_count;
  // A repeat must not overflow for the same reason again: it runs one work item per work slot (the packing of the
  // evaluation that follows a withheld one is otherwise planned from whatever evaluation ran last), and if a lone item
  // outgrew the store while its subtree was shared by fewer than four, every subtree is shared four ways from here on
  // (straight to four: one repeat instead of a ladder of them; the fallback only ever runs repeats) ...
  // It stays at four when the capacity variant is raised below (ADVICE r04 asked for a reset; measured: a geometry that has to
  // climb several variants -- test_graph_replay_..., 5527-node subtree -- then pays TWO repeats per variant, whole subtrees
  // first and four-way next, instead of one; the fallback only ever runs repeats, the planned packings decide the parts from
  // the measured shapes whatever this says)
  const bool raise = s[kStatStickyNode] || s[kStatStickyAtom];
  if (s[kStatStickySplit] > 0) c->fallback_parts = 4;
  int rc = upload_identity_packing(c);
  if (rc != AGBNP_HIP_OK) return rc;
  c->forests_hint = 0;  // (one work item per slot again: the old rule decides)
  if (s[kStatStickyOrder] & 1) {
    c->order_valid = false;  // the context has reordered its atoms: the next agbnp_hip_execute_openmm rebuilds the maps
    // (five-launch mode: the device has renewed the neighbour masks in that evaluation -- from positions read through the STALE
    // maps, i.e. other atoms' positions: laid down anew by a launch of their own in front of the next evaluation)
    c->masks_valid = false;
  }
  // (bit 1, five-launch mode: a heavy atom had left the neighbour masks' skin.  The device has laid the masks down anew in
  // that very evaluation's Born launch: nothing for the host to do but repeat what was withheld)
  if (s[kStatStickyRow] && !c->rows_disabled) {
    // a neighbour list of the row-form pair stages outgrew what the launches walk of it: they walk twice as much from
    // here on -- or, if that already was the whole stride, the tile kernels take over (other launches either way: a
    // captured graph of this context is stale)
    const bool whole = c->P.nlh_cap >= c->nlh_stride && c->P.nla_cap >= c->nla_stride && (!c->P.gb_rows || c->P.nlg_cap >= c->nlg_stride);
    if (whole) {
      c->rows_disabled = true;
    } else {
      c->row_boost *= 2;
      const int stale = 1;  // (the work items were laid down for the narrower walk: rebuilt with the lists)
      HIP_TRY(c, hipMemcpy(c->d_nl_flag.p, &stale, sizeof(int), hipMemcpyHostToDevice));
    }
    wire_args(c);
    c->generation++;
  }
  if (raise) {
    // ... and, if a single subtree outgrew the store, on the next larger capacity variant
    if (c->variant >= kGlobalVariant)
      return c->fail(AGBNP_HIP_ERR_CAPACITY, "overlap subtree exceeds the largest supported capacity (32768 nodes / 255 partners per heavy atom)");
    c->variant++;
    c->generation++;  // other kernels, other scratch: a captured graph of this context is stale
  }
  // (a packed forest that outgrew its store: the device has already tightened its packing thresholds)
  return AGBNP_HIP_OK;
}

}  // namespace

extern "C" {

#ifndef AGBNP_SRC_HASH
#define AGBNP_SRC_HASH "unknown"  // (a build outside csrc/Makefile: diagnostic libraries)
#endif
const char* agbnp_hip_build_id(void) { return AGBNP_SRC_HASH; }

int agbnp_hip_device_count(void) {
  int k = 0;
  if (hipGetDeviceCount(&k) != hipSuccess) return 0;
  return k;
}

const char* agbnp_hip_last_error(const agbnp_hip_context* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int agbnp_hip_create(agbnp_hip_context** out, int n, const double* radius, const double* gamma, const double* vdw_alpha,
                     const double* charge, const int* ishydrogen, int version, int nonbonded_method, double cutoff, int device) {
  auto bail = [&](int code, const std::string& msg, agbnp_hip_context* c) {
    g_create_error = msg;
    delete c;
    if (out) *out = nullptr;
    return code;
  };
  if (!out || n <= 0 || !radius || !gamma || !vdw_alpha || !charge || !ishydrogen)
    return bail(AGBNP_HIP_ERR_INVALID_ARGUMENT, "agbnp_hip_create: null pointer or non-positive particle count", nullptr);
  if (version < 0 || version > 2) return bail(AGBNP_HIP_ERR_INVALID_ARGUMENT, "AGBNPForce::setVersion(): illegal version number", nullptr);
  if (version == 2)
    return bail(AGBNP_HIP_ERR_INVALID_ARGUMENT, "agbnp_hip_create: AGBNP version 2 is outside this engine's scope (versions 0 and 1 only)", nullptr);
  if (nonbonded_method < 0 || nonbonded_method > 2)
    return bail(AGBNP_HIP_ERR_INVALID_ARGUMENT, "agbnp_hip_create: illegal nonbonded method", nullptr);

  agbnp_hip_context* c = new agbnp_hip_context();
  c->n = n;
  c->version = version;
  c->method = nonbonded_method;
  c->cutoff = cutoff;
  c->device = device;
  c->fused_outputs = getenv("AGBNP_HIP_OUTPUT_LAUNCH") == nullptr;
  c->five = (version == 0 || version == 1) && !(getenv("AGBNP_HIP_FIVE_LAUNCHES") && atoi(getenv("AGBNP_HIP_FIVE_LAUNCHES")) == 0);  // (default since round 5, version 0 since round 6; 0: with the k_prep launch)
  c->five_active = c->five;
  if (getenv("AGBNP_HIP_MASK_SKIN")) c->mask_skin = std::min(0.5, std::max(0.0, atof(getenv("AGBNP_HIP_MASK_SKIN"))));
  c->r_vdw.assign(radius, radius + n);
  c->gamma.resize(n);
  c->alpha.assign(vdw_alpha, vdw_alpha + n);
  c->charge.assign(charge, charge + n);
  c->ish.resize(n);
  c->a2h.assign(n, -1);
  // parameter checks of ReferenceAGBNPKernels.cpp:96-117
  double common_gamma = -1;
  for (int i = 0; i < n; i++) {
    const bool h = ishydrogen[i] != 0;
    c->ish[i] = h ? 1 : 0;
    c->gamma[i] = h ? 0.0 : gamma[i];
    if (!(radius[i] > 0.0)) return bail(AGBNP_HIP_ERR_INVALID_ARGUMENT, "agbnp_hip_create: particle radius must be positive", c);
    if (common_gamma < 0 && !h) {
      common_gamma = gamma[i];
    } else if (!h && pow(common_gamma - gamma[i], 2) > FLT_MIN) {
      return bail(AGBNP_HIP_ERR_PARAMETERS, "initialize(): AGBNP does not support multiple gamma values.", c);
    }
    if (!h) {
      c->a2h[i] = (int)c->h2a.size();
      c->h2a.push_back(i);
    }
  }
  c->nh = (int)c->h2a.size();
  c->lut.build(c->r_vdw, c->ish);

  int ndev = 0;
  hipError_t he = hipGetDeviceCount(&ndev);
  if (he != hipSuccess || ndev <= 0)
    return bail(AGBNP_HIP_ERR_DEVICE, "agbnp_hip_create: no HIP device available (this engine has no CPU fallback)", c);
  if (device < 0 || device >= ndev) return bail(AGBNP_HIP_ERR_INVALID_ARGUMENT, "agbnp_hip_create: device index out of range", c);
#define CREATE_TRY(call)                                                                      \
  do {                                                                                        \
    hipError_t e__ = (call);                                                                  \
    if (e__ != hipSuccess) return bail(AGBNP_HIP_ERR_DEVICE, std::string(#call) + ": " + hipGetErrorString(e__), c); \
  } while (0)
  CREATE_TRY(hipSetDevice(device));
  {
    // resident tree workgroups per capacity variant: what one "round" of the forest packing is
    int cus = 256;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cus <= 0) cus = 256;
    c->cus = cus;
    for (int v = 0; v < kGlobalVariant; v++) c->tree_slots[v] = tree_variant_wgs_per_cu(v) * cus;
    c->tree_slots[kGlobalVariant] = kGlobalGrid;
  }
  CREATE_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));

  CREATE_TRY(c->d_a2h.upload(c->a2h));
  std::vector<int> h2a_pad = c->h2a;
  if (h2a_pad.empty()) h2a_pad.push_back(0);
  CREATE_TRY(c->d_h2a.upload(h2a_pad));
  std::vector<int2> ameta(n);
  for (int i = 0; i < n; i++) ameta[i] = make_int2(c->lut.type_screened[i], c->lut.type_screener[i]);
  CREATE_TRY(c->d_ameta.upload(ameta));
  const double dr = kI4MaxA / (kI4Nodes - 1);
  std::vector<double2> lut(std::max<size_t>(1, c->lut.y.size() / kI4Nodes * kLutStride), make_double2(0.0, 0.0));
  for (size_t k = 0; k < c->lut.y.size(); k++)  // rows padded to kLutStride entries (LDS bank spreading)
    lut[k / kI4Nodes * kLutStride + k % kI4Nodes] = make_double2(c->lut.y[k], c->lut.y2[k] * dr * dr / 6.0);
  CREATE_TRY(c->d_lut.upload(lut));
  if (lut.size() * sizeof(double2) > 128 * 1024)  // + 24 KB of tile records in k_dborn_tiles
    return bail(AGBNP_HIP_ERR_CAPACITY, "agbnp_hip_create: too many distinct radius pairs for the LDS-resident I4 tables", c);

  int rc = upload_parameters(c);
  if (rc != AGBNP_HIP_OK) return bail(rc, c->err, c);
  rc = allocate_work(c);
  if (rc != AGBNP_HIP_OK) return bail(rc, c->err, c);
  wire_args(c);
  *out = c;
  return AGBNP_HIP_OK;
}

int agbnp_hip_update_parameters(agbnp_hip_context* c, int n, const double* radius, const double* gamma, const double* vdw_alpha,
                                const double* charge, const int* ishydrogen) {
  if (!c) return AGBNP_HIP_ERR_INVALID_ARGUMENT;
  if (!radius || !gamma || !vdw_alpha || !charge || !ishydrogen) return c->fail(AGBNP_HIP_ERR_INVALID_ARGUMENT, "null pointer");
  if (n != c->n) return c->fail(AGBNP_HIP_ERR_PARAMETERS, "updateParametersInContext: The number of AGBNP particles has changed");
  for (int i = 0; i < n; i++) {
    if ((c->r_vdw[i] - radius[i]) * (c->r_vdw[i] - radius[i]) > 1.e-6)
      return c->fail(AGBNP_HIP_ERR_PARAMETERS, "updateParametersInContext: AGBNP plugin does not support changing atomic radii.");
    if (ishydrogen[i] && c->ish[i] == 0)
      return c->fail(AGBNP_HIP_ERR_PARAMETERS, "updateParametersInContext: AGBNP plugin does not support changing heavy/hydrogen atoms.");
  }
  for (int i = 0; i < n; i++) {
    c->gamma[i] = ishydrogen[i] ? 0.0 : gamma[i];
    c->alpha[i] = vdw_alpha[i];
    c->charge[i] = charge[i];
  }
  HIP_TRY(c, hipSetDevice(c->device));
  // the parameter arrays are rewritten in place (their addresses stay valid for captured graphs), so nothing of this
  // context may be in flight: the context's own stream is drained here, and a caller who enqueues on streams of its own
  // (agbnp_hip_execute_device / _openmm with a stream argument) calls agbnp_hip_finish on them first -- as it must anyway
  // to learn about withheld evaluations.  (Not hipDeviceSynchronize: that would stall every other context of the device.)
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  for (void* st : c->user_streams) HIP_TRY(c, hipStreamSynchronize((hipStream_t)st));
  return upload_parameters(c, true);  // (the arrays are rewritten in place: no kernel argument changes)
}

int agbnp_hip_execute_device(agbnp_hip_context* c, const double* d_pos, double* d_force, double* d_energy, void* stream) {
  if (!c) return AGBNP_HIP_ERR_INVALID_ARGUMENT;
  if (!d_pos || !d_force || !d_energy) return c->fail(AGBNP_HIP_ERR_INVALID_ARGUMENT, "agbnp_hip_execute_device: null pointer");
  HIP_TRY(c, hipSetDevice(c->device));
  hipStream_t st = stream ? (hipStream_t)stream : c->stream;
  note_stream(c, stream);
  return enqueue(c, d_pos, d_force, d_energy, st);
}

int agbnp_hip_execute_openmm(agbnp_hip_context* c, const void* d_posq, int posq_is_double, const void* d_posq_correction,
                             const int* d_atom_index, int padded_num_atoms, long long* d_force_buffer, void* d_energy_buffer,
                             int energy_is_double, int energy_slot, void* stream) {
  if (!c) return AGBNP_HIP_ERR_INVALID_ARGUMENT;
  if (!d_posq || !d_force_buffer) return c->fail(AGBNP_HIP_ERR_INVALID_ARGUMENT, "agbnp_hip_execute_openmm: null pointer");
  if (padded_num_atoms < c->n) return c->fail(AGBNP_HIP_ERR_INVALID_ARGUMENT, "agbnp_hip_execute_openmm: padded atom count below the particle count");
  if (posq_is_double && d_posq_correction)
    return c->fail(AGBNP_HIP_ERR_INVALID_ARGUMENT, "agbnp_hip_execute_openmm: a position correction only exists beside float positions");
  if (energy_slot < 0) return c->fail(AGBNP_HIP_ERR_INVALID_ARGUMENT, "agbnp_hip_execute_openmm: negative energy slot");
  HIP_TRY(c, hipSetDevice(c->device));
  hipStream_t st = stream ? (hipStream_t)stream : c->stream;
  note_stream(c, stream);
  // Input side.  Normally k_prep reads the context's posq itself, through the engine's maps of the context's atom order
  // (OpenmmSource): they are built when an atomIndex array is first seen (or seen again after the device has found it
  // changed), one small launch that a steady run never repeats.  AGBNP_HIP_ADAPTER_LAUNCH=1: an adapter launch per
  // evaluation instead (posq -> xyz in particle order, and the particle -> slot map for the output side).
  const bool fused = !c->adapter_launch;
  if (fused) {
    if (!c->order_valid || c->order_ptr != d_atom_index) {
      HIP_TRY(c, launch_order_maps(c->n, d_atom_index, c->d_a2h.p, c->d_ctx_slot.p, c->d_hslot.p, st));
      if (c->row_atoms_kind == 1) c->row_atoms_kind = -1;  // (slots of the old order: enqueue rewrites them)
      c->order_valid = true;
      c->order_ptr = d_atom_index;
    }
    c->P.in.posq = d_posq;
    c->P.in.correction = static_cast<const float4*>(d_posq_correction);
    c->P.in.is_double = posq_is_double;
    c->P.in.atom_index = d_atom_index;
    c->P.in.hslot = c->d_hslot.p;
  } else {
    HIP_TRY(c, launch_adapt_positions(c->n, d_posq, posq_is_double, d_posq_correction, d_atom_index, c->d_pos_in.p, c->d_ctx_slot.p, st));
  }
  // the output side is the engine's own last kernel: forces as fixed point at the context's slots, energy into its
  // accumulator (the kernel arguments are captured by value at launch, so the targets are set for this evaluation only)
  c->P.omm.force_fixed = reinterpret_cast<unsigned long long*>(d_force_buffer);
  c->P.omm.padded = padded_num_atoms;
  c->P.omm.ctx_slot = c->d_ctx_slot.p;
  c->P.omm.energy_buffer = d_energy_buffer;
  c->P.omm.energy_is_double = energy_is_double;
  c->P.omm.energy_slot = energy_slot;
  const int rc = enqueue(c, c->d_pos_in.p, c->d_force_tmp.p, c->d_energy_tmp.p, st);
  c->P.in = OpenmmSource();
  c->P.omm = OpenmmTargets();
  return rc;
}

int agbnp_hip_atom_order_changed(agbnp_hip_context* c) {
  if (!c) return AGBNP_HIP_ERR_INVALID_ARGUMENT;
  c->order_valid = false;
  return AGBNP_HIP_OK;
}

int agbnp_hip_finish(agbnp_hip_context* c, void* stream, int* must_repeat) {
  if (!c) return AGBNP_HIP_ERR_INVALID_ARGUMENT;
  int dummy = 0;
  if (!must_repeat) must_repeat = &dummy;
  HIP_TRY(c, hipSetDevice(c->device));
  hipStream_t st = stream ? (hipStream_t)stream : c->stream;
  c->user_streams.erase(std::remove(c->user_streams.begin(), c->user_streams.end(), stream), c->user_streams.end());
  int rc = harvest(c, must_repeat, st);  // (waits for the stream)
  if (rc != AGBNP_HIP_OK) return rc;
  if (c->carried_count > 0) {
    // evaluations of execute_device that a call of agbnp_hip_execute_host in between had to harvest: they were enqueued
    // BEFORE everything this finish has just read, so they come first and the later indices move up
    for (int& k : c->withheld) k += c->carried_seq;
    c->withheld.insert(c->withheld.begin(), c->carried.begin(), c->carried.end());
    c->withheld_count += c->carried_count;
    *must_repeat = c->withheld_count;
  }
  c->carried.clear();
  c->carried_count = c->carried_seq = 0;
  c->unfinished = !c->user_streams.empty();
  return AGBNP_HIP_OK;
}

int agbnp_hip_execute_host(agbnp_hip_context* c, const double* pos, double* forces, double* energy) {
  if (!c) return AGBNP_HIP_ERR_INVALID_ARGUMENT;
  if (!pos || !forces || !energy) return c->fail(AGBNP_HIP_ERR_INVALID_ARGUMENT, "agbnp_hip_execute_host: null pointer");
  HIP_TRY(c, hipSetDevice(c->device));
  const size_t bytes = sizeof(double) * 3 * (size_t)c->n;
  if (c->unfinished) {
    // The sticky overflow log is about to be read and cleared by this call's own harvest.  Evaluations that the caller
    // has enqueued with agbnp_hip_execute_device and not yet finished must not lose their entries: they are harvested
    // now and carried over to the caller's next agbnp_hip_finish.
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    for (void* st : c->user_streams) HIP_TRY(c, hipStreamSynchronize((hipStream_t)st));
    int pending = 0;
    int rc = harvest(c, &pending, c->stream);
    if (rc != AGBNP_HIP_OK) return rc;
    const int seq = c->last_device_seq;
    for (int k : c->withheld) c->carried.push_back(k + c->carried_seq);
    c->carried_count += pending;
    c->carried_seq += seq;
    c->unfinished = false;
  }
  const size_t n3 = 3 * (size_t)c->n;
  double* const d_energy = c->d_force_tmp.p + n3;  // forces and energy leave in one buffer: one clear, one copy back
  double* const h_in = c->h_xfer ? c->h_xfer : nullptr;
  double* const h_out = c->h_xfer ? c->h_xfer + n3 : c->h_force_tmp.data();
  if (h_in) std::memcpy(h_in, pos, bytes);
  for (int attempt = 0; attempt < 10; attempt++) {
    HIP_TRY(c, hipMemcpyAsync(c->d_pos_in.p, h_in ? h_in : pos, bytes, hipMemcpyHostToDevice, c->stream));
    c->P.zero_out = c->d_force_tmp.p;  // (cleared by k_prep: the kernel arguments are captured by value at launch)
    int rc = enqueue(c, c->d_pos_in.p, c->d_force_tmp.p, d_energy, c->stream);
    c->P.zero_out = nullptr;
    if (rc != AGBNP_HIP_OK) return rc;
    HIP_TRY(c, hipMemcpyAsync(h_out, c->d_force_tmp.p, bytes + sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (c->h_status && c->h_report && !c->timeline.enabled && c->enqueued < 1024) {
      // the short cut: the device's own word on this evaluation (pinned memory, written when its tree stage had ended).
      // The wait for the stream is a spin on its status: a blocking wait comes back 10-20 us late.
      for (hipError_t q = hipStreamQuery(c->stream); q != hipSuccess; q = hipStreamQuery(c->stream))
        if (q != hipErrorNotReady) {
          HIP_TRY(c, q);
          break;
        }
      const volatile int* h = c->h_status;
      const int judged = h[0];
      std::atomic_thread_fence(std::memory_order_acquire);
      if (judged == c->enqueued && h[1] == 0) {
        c->lazy_evals++;
        for (size_t k = 0; k < n3; k++) forces[k] += h_out[k];
        *energy = h_out[n3];
        return AGBNP_HIP_OK;
      }
    }
    int repeat = 0;
    rc = harvest(c, &repeat, c->stream);  // (its own reads follow on the stream; one wait for everything)
    if (rc != AGBNP_HIP_OK) return rc;
    if (repeat) continue;
    for (size_t k = 0; k < n3; k++) forces[k] += h_out[k];
    *energy = h_out[n3];
    return AGBNP_HIP_OK;
  }
  return c->fail(AGBNP_HIP_ERR_CAPACITY, "agbnp_hip_execute_host: capacity negotiation did not converge");
}

int agbnp_hip_get_scalar(agbnp_hip_context* c, int which, double* value) {
  if (!c || !value) return AGBNP_HIP_ERR_INVALID_ARGUMENT;
  if (c->lazy_evals) {
    HIP_TRY(c, hipSetDevice(c->device));
    const int rc_ = catch_up(c);
    if (rc_ != AGBNP_HIP_OK) return rc_;
  }
  if (which == 15) {  // why the last agbnp_hip_finish withheld evaluations (valid whether or not an evaluation has completed)
    const int* s = c->last_status;
    *value = (s[kStatStickyNode] ? 1 : 0) | (s[kStatStickyAtom] ? 2 : 0) | (s[kStatStickyPack] ? 4 : 0) | (s[kStatStickyRow] ? 8 : 0) |
             (s[kStatStickyOrder] ? 16 : 0) | (s[kStatStickyForest] << 5) | (s[kStatStickySplit] << 8);  // (32 / 64: a forest's nodes / local atoms)
    return AGBNP_HIP_OK;
  }
  if (which == 17) {  // forests healed inside the tree launch over the evaluations the last agbnp_hip_finish covered (none withheld for them)
    *value = c->last_status[kStatStickyHealed];
    return AGBNP_HIP_OK;
  }
  if (!c->have_results) return c->fail(AGBNP_HIP_ERR_INVALID_ARGUMENT, "no completed evaluation yet");
  switch (which) {
    case 0: *value = c->last_components[0]; break;
    case 1: *value = c->last_components[1]; break;
    case 2: *value = c->last_components[2]; break;
    case 3: *value = c->last_components[3]; break;
    case 4: *value = c->last_status[kStatMaxNodes]; break;
    case 5: *value = c->last_status[kStatTotalNodes]; break;
    case 6: *value = c->variant; break;
    case 7: *value = c->last_status[kStatMaxAtoms]; break;
    case 8: *value = c->last_status[kStatForests]; break;
    case 9: *value = c->P.rows_on; break;        // 1: the range-limited pair stages run in row form
    case 10: *value = c->last_rows[1]; break;    // builds of the neighbour rows so far
    case 13: *value = c->last_rows[2]; break;    // entries per slice of a neighbour row (one wave walks a slice)
    case 11: *value = c->last_pack[0]; break;    // forest packing: how far the assumed store capacity is tightened (0 = not)
    case 12: *value = c->last_pack[1]; break;    // ... evaluations since the packing in use was planned
    case 14: *value = c->last_pack[3]; break;    // ... packings planned so far
    case 16: *value = c->version == 1 ? (c->five_active ? 5 : 6) : (c->five_active ? 2 : 3); break;  // launches of an evaluation as the context runs now (no k_prep launch in the five-launch mode; see agbnp_hip.h)
    default: return c->fail(AGBNP_HIP_ERR_INVALID_ARGUMENT, "unknown scalar id");
  }
  return AGBNP_HIP_OK;
}

int agbnp_hip_get_vector(agbnp_hip_context* c, int which, double* out) {
  if (!c || !out) return AGBNP_HIP_ERR_INVALID_ARGUMENT;
  if (c->lazy_evals) {
    HIP_TRY(c, hipSetDevice(c->device));
    const int rc_ = catch_up(c);
    if (rc_ != AGBNP_HIP_OK) return rc_;
  }
  if (!c->have_results) return c->fail(AGBNP_HIP_ERR_INVALID_ARGUMENT, "no completed evaluation yet");
  HIP_TRY(c, hipSetDevice(c->device));
  const int n = c->n, nh = c->nh;
  std::vector<double> tmp(std::max(n, std::max(nh, 1)));
  auto heavy_to_atoms = [&](const double* dsrc, size_t stride, size_t word, double scale_by_inv_vol) -> int {
    std::vector<double> raw((size_t)std::max(nh, 1) * stride);
    HIP_TRY(c, hipMemcpy(raw.data(), dsrc, sizeof(double) * raw.size(), hipMemcpyDeviceToHost));
    for (int i = 0; i < n; i++) out[i] = 0.0;
    for (int h = 0; h < nh; h++) {
      double v = raw[(size_t)h * stride + word];
      if (scale_by_inv_vol != 0.0) v /= (4. * M_PI * pow(c->r_vdw[c->h2a[h]], 3) / 3.);
      out[c->h2a[h]] = v;
    }
    return AGBNP_HIP_OK;
  };
  switch (which) {
    case 0: return heavy_to_atoms(c->hrow(kHvSvVdw), 1, 0, 0.0);
    case 1:
      if (c->version != 1) return c->fail(AGBNP_HIP_ERR_INVALID_ARGUMENT, "Born radii exist for version 1 only");
      HIP_TRY(c, hipMemcpy(out, c->d_born.p, sizeof(double) * n, hipMemcpyDeviceToHost));
      return AGBNP_HIP_OK;
    case 2: return heavy_to_atoms(c->hrow(kHvSvVdw), 1, 0, 1.0);
    case 3:
      if (!c->diagnostics) return c->fail(AGBNP_HIP_ERR_INVALID_ARGUMENT, "enlarged-radius self volumes need agbnp_hip_set_diagnostics(ctx, 1) before the evaluation");
      return heavy_to_atoms(c->hrow(kHvSvLarge), 1, 0, 0.0);
    case 4:
    case 5: {  // overlap-tree shape: nodes / local atoms of the subtree rooted at every heavy atom (0 for hydrogens)
      std::vector<int2> sz(std::max(nh, 1));
      HIP_TRY(c, hipMemcpy(sz.data(), c->d_sizes.p + (size_t)(c->five_active ? c->parity : 0) * std::max(nh, 1), sizeof(int2) * std::max(nh, 1), hipMemcpyDeviceToHost));
      for (int i = 0; i < n; i++) out[i] = 0.0;
      for (int h = 0; h < nh; h++) out[c->h2a[h]] = which == 4 ? sz[h].x : sz[h].y;
      return AGBNP_HIP_OK;
    }
    default: return c->fail(AGBNP_HIP_ERR_INVALID_ARGUMENT, "unknown vector id");
  }
}

int agbnp_hip_get_table_sizes(agbnp_hip_context* c, int* nscreened, int* nscreener) {
  if (!c || !nscreened || !nscreener) return AGBNP_HIP_ERR_INVALID_ARGUMENT;
  *nscreened = c->lut.nscreened;
  *nscreener = c->lut.nscreener;
  return AGBNP_HIP_OK;
}

int agbnp_hip_get_tables(agbnp_hip_context* c, double* y, double* y2, int* type_screened, int* type_screener) {
  if (!c || !y || !y2 || !type_screened || !type_screener) return AGBNP_HIP_ERR_INVALID_ARGUMENT;
  memcpy(y, c->lut.y.data(), sizeof(double) * c->lut.y.size());
  memcpy(y2, c->lut.y2.data(), sizeof(double) * c->lut.y2.size());
  memcpy(type_screened, c->lut.type_screened.data(), sizeof(int) * c->n);
  memcpy(type_screener, c->lut.type_screener.data(), sizeof(int) * c->n);
  return AGBNP_HIP_OK;
}

int agbnp_hip_set_mode(agbnp_hip_context* c, int mode) {
  if (!c) return AGBNP_HIP_ERR_INVALID_ARGUMENT;
  if (mode & ~(AGBNP_HIP_MODE_FAST | AGBNP_HIP_MODE_DETERMINISTIC | AGBNP_HIP_MODE_SINGLE))
    return c->fail(AGBNP_HIP_ERR_INVALID_ARGUMENT, "agbnp_hip_set_mode: unknown mode bits");
  if ((mode & AGBNP_HIP_MODE_SINGLE) && !(mode & AGBNP_HIP_MODE_FAST))
    return c->fail(AGBNP_HIP_ERR_INVALID_ARGUMENT, "agbnp_hip_set_mode: single precision is an option of the fast mode (the Reference semantics are FP64)");
  if ((mode & AGBNP_HIP_MODE_FAST) && c->method == 2)
    return c->fail(AGBNP_HIP_ERR_INVALID_ARGUMENT, "agbnp_hip_set_mode: the fast mode does not take CutoffPeriodic (no periodic box crosses this boundary)");
  if ((mode & AGBNP_HIP_MODE_FAST) && c->method != 0 && !(c->cutoff > 0.0))
    return c->fail(AGBNP_HIP_ERR_INVALID_ARGUMENT, "agbnp_hip_set_mode: the fast mode needs a positive cutoff distance");
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipStreamSynchronize(c->stream));  // (this context's own work; other contexts of the device are not stalled)
  if (mode != c->mode) c->generation++;  // other kernel arguments: a captured graph is stale
  c->mode = mode;
  wire_args(c);
  if (c->rows_capable) {  // the neighbour lists were built for the reach of the mode that is being left
    const int stale = 1;
    HIP_TRY(c, hipMemcpy(c->d_nl_flag.p, &stale, sizeof(int), hipMemcpyHostToDevice));
  }
  return AGBNP_HIP_OK;
}

int agbnp_hip_get_mode(const agbnp_hip_context* c) { return c ? c->mode : -1; }

int agbnp_hip_set_diagnostics(agbnp_hip_context* c, int enabled) {
  if (!c) return AGBNP_HIP_ERR_INVALID_ARGUMENT;
  c->diagnostics = enabled != 0;
  wire_args(c);
  return AGBNP_HIP_OK;
}

int agbnp_hip_set_profiling(agbnp_hip_context* c, int enabled) {
  if (!c) return AGBNP_HIP_ERR_INVALID_ARGUMENT;
  c->timeline.enabled = enabled != 0;
  c->timeline.used = 0;
  for (int k = 0; k < kKernelCount; k++) {
    c->kernel_ms[k] = 0.0;
    c->kernel_launches[k] = 0;
  }
  return AGBNP_HIP_OK;
}

int agbnp_hip_num_kernels(void) { return kKernelCount; }

const char* agbnp_hip_kernel_name(int index) {
  static const char* names[kKernelCount] = {"k_prep",        "k_tree_cavity", "k_born_tiles", "k_gb_tiles", "k_dborn_tiles",
                                            "k_tree_pseudo", "k_outputs",     "k_born_rows",  "k_dborn_rows",  "k_gb_rows"};
  return (index >= 0 && index < kKernelCount) ? names[index] : "";
}

int agbnp_hip_get_kernel_times(agbnp_hip_context* c, double* total_ms, long* launches) {
  if (!c || !total_ms || !launches) return AGBNP_HIP_ERR_INVALID_ARGUMENT;
  for (int k = 0; k < kKernelCount; k++) {
    total_ms[k] = c->kernel_ms[k];
    launches[k] = c->kernel_launches[k];
  }
  return AGBNP_HIP_OK;
}

int agbnp_hip_host_tables(int n, const double* radius, const int* ishydrogen, int* nscreened, int* nscreener, double* y,
                          double* y2, int table_capacity, int* type_screened, int* type_screener) {
  if (n <= 0 || !radius || !ishydrogen || !nscreened || !nscreener || !y || !y2 || !type_screened || !type_screener) {
    g_create_error = "agbnp_hip_host_tables: null pointer or non-positive particle count";
    return AGBNP_HIP_ERR_INVALID_ARGUMENT;
  }
  I4TableSet t;
  t.build(std::vector<double>(radius, radius + n), std::vector<int>(ishydrogen, ishydrogen + n));
  *nscreened = t.nscreened;
  *nscreener = t.nscreener;
  if ((int)t.y.size() > table_capacity) {
    g_create_error = "agbnp_hip_host_tables: table_capacity too small";
    return AGBNP_HIP_ERR_INVALID_ARGUMENT;
  }
  memcpy(y, t.y.data(), sizeof(double) * t.y.size());
  memcpy(y2, t.y2.data(), sizeof(double) * t.y2.size());
  memcpy(type_screened, t.type_screened.data(), sizeof(int) * n);
  memcpy(type_screener, t.type_screener.data(), sizeof(int) * n);
  return AGBNP_HIP_OK;
}

int agbnp_hip_withheld_evaluations(const agbnp_hip_context* c, int* indices, int capacity) {
  if (!c) return -1;
  for (int k = 0; indices && k < capacity && k < (int)c->withheld.size(); k++) indices[k] = c->withheld[k];
  return c->withheld_count;
}

unsigned agbnp_hip_generation(const agbnp_hip_context* c) { return c ? c->generation : 0u; }

int agbnp_hip_poll(const agbnp_hip_context* c, int* evaluations_completed, int* withheld) {
  if (!c) return AGBNP_HIP_ERR_INVALID_ARGUMENT;
  if (!c->h_status) return AGBNP_HIP_ERR_DEVICE;
  const volatile int* h = c->h_status;
  const int done = h[0] - c->lazy_evals;  // (written after the withheld count, behind a system-scope fence; execute_host's own are not the caller's)
  std::atomic_thread_fence(std::memory_order_acquire);
  if (evaluations_completed) *evaluations_completed = done;
  if (withheld) *withheld = h[1];
  return AGBNP_HIP_OK;
}

int agbnp_hip_wait_verdict(const agbnp_hip_context* c, int evaluations, double timeout_seconds, int* evaluations_completed, int* withheld) {
  if (!c) return AGBNP_HIP_ERR_INVALID_ARGUMENT;
  if (!c->h_status) return AGBNP_HIP_ERR_DEVICE;
  const int target = evaluations > 0 ? evaluations + c->lazy_evals : c->enqueued;
  const volatile int* h = c->h_status;
  const auto t0 = std::chrono::steady_clock::now();
  int done = h[0];
  for (unsigned spins = 0; done < target; done = h[0]) {
    __builtin_ia32_pause();
    if ((++spins & 1023u) == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_seconds) break;
  }
  std::atomic_thread_fence(std::memory_order_acquire);  // (the withheld count is read AFTER the running number that vouches for it)
  if (evaluations_completed) *evaluations_completed = done - c->lazy_evals;
  if (withheld) *withheld = h[1];  // (written before the running number, behind a system-scope fence)
  return done >= target ? AGBNP_HIP_OK : AGBNP_HIP_ERR_TIMEOUT;
}

// ---- diagnostic entry points (not part of include/agbnp_hip.h; used by scripts/ only) ---------------------------------
// the forest packing as the device holds it, in WORK-SLOT order: the items of slot s at forest_start[s] .. forest_start[s+1]),
// and the per-subtree shapes
int agbnp_debug_get_packing(agbnp_hip_context* c, int* order, int order_cap, int* forest_start, int start_cap, int* nforests, int* sizes) {
  if (!c) return AGBNP_HIP_ERR_INVALID_ARGUMENT;
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipDeviceSynchronize());
  int nf = 0;
  HIP_TRY(c, hipMemcpy(&nf, c->d_forest.p + c->slot_cap + 1, sizeof(int), hipMemcpyDeviceToHost));
  nf = std::min(nf, c->slot_cap);
  *nforests = nf;
  std::vector<int> rows(c->d_rows.count);
  HIP_TRY(c, hipMemcpy(rows.data(), c->d_rows.p, sizeof(int) * rows.size(), hipMemcpyDeviceToHost));
  int run = 0;
  for (int s = 0; s < nf && s + 1 < start_cap; s++) {
    forest_start[s] = run;
    for (int k = 0; k < rows[(size_t)kRowStride * s + kMaxItems] && run < order_cap; k++) order[run++] = rows[(size_t)kRowStride * s + k];
    forest_start[s + 1] = run;
  }
  if (sizes) HIP_TRY(c, hipMemcpy(sizes, c->d_sizes.p + (size_t)(c->five_active ? c->parity : 0) * std::max(c->nh, 1), sizeof(int2) * std::max(c->nh, 1), hipMemcpyDeviceToHost));
  return AGBNP_HIP_OK;
}
// replaces the packing (same form) and (freeze != 0) stops the bookkeeping from planning new ones
int agbnp_debug_set_packing(agbnp_hip_context* c, const int* order, int norder, const int* forest_start, int nforests, int freeze) {
  if (!c) return AGBNP_HIP_ERR_INVALID_ARGUMENT;
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipDeviceSynchronize());
  if (order) {
    if (nforests > c->slot_cap) return c->fail(AGBNP_HIP_ERR_INVALID_ARGUMENT, "agbnp_debug_set_packing: more forests than work slots");
    std::vector<int> rows(c->d_rows.count, -1);
    for (int s = 0; s < nforests; s++) {
      const int count = std::min(forest_start[s + 1] - forest_start[s], (int)kMaxItems);
      rows[(size_t)kRowStride * s + kMaxItems] = count;
      for (int k = 0; k < count && forest_start[s] + k < norder; k++) rows[(size_t)kRowStride * s + k] = order[forest_start[s] + k];
    }
    HIP_TRY(c, hipMemcpy(c->d_rows.p, rows.data(), sizeof(int) * rows.size(), hipMemcpyHostToDevice));
    if (c->five) {  // (five-launch mode: the roots' atoms beside the rows)
      std::vector<int> atoms((size_t)kMaxItems * c->slot_cap, 0);
      for (int s = 0; s < nforests; s++)
        for (int k = 0; k < kMaxItems; k++) {
          const int item = rows[(size_t)kRowStride * s + k];
          if (item >= 0 && work_item_root(item) < c->nh) atoms[(size_t)kMaxItems * s + k] = c->h2a[work_item_root(item)];
        }
      HIP_TRY(c, hipMemcpy(c->d_row_atoms.p, atoms.data(), sizeof(int) * atoms.size(), hipMemcpyHostToDevice));
      c->row_atoms_kind = 0;
    }
    HIP_TRY(c, hipMemcpy(c->d_forest.p + c->slot_cap + 1, &nforests, sizeof(int), hipMemcpyHostToDevice));
  }
  c->P.pack_enabled = freeze ? 3 : c->P.pack_enabled;  // 3: the bookkeeping keeps its statistics but writes no packing
  return AGBNP_HIP_OK;
}

// row form: bw_i = brw_i + bru_i as the GB stage left it [n], W+U by heavy index [nh] (what the chain-rule stage left)
int agbnp_debug_get_rows(agbnp_hip_context* c, double* bw, double* wu) {
  if (!c || !c->rows_capable) return AGBNP_HIP_ERR_INVALID_ARGUMENT;
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipDeviceSynchronize());
  HIP_TRY(c, hipMemcpy(bw, c->d_bw.p, sizeof(double) * c->n, hipMemcpyDeviceToHost));
  HIP_TRY(c, hipMemcpy(wu, c->d_dbf.p + 3 * (size_t)c->n, sizeof(double) * c->nh, hipMemcpyDeviceToHost));
  return AGBNP_HIP_OK;
}

int agbnp_hip_num_particles(const agbnp_hip_context* c) { return c ? c->n : -1; }
int agbnp_hip_version(const agbnp_hip_context* c) { return c ? c->version : -1; }

void agbnp_hip_destroy(agbnp_hip_context* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->stream) {
    (void)hipStreamSynchronize(c->stream);
    (void)hipStreamDestroy(c->stream);
  }
  for (hipEvent_t e : c->timeline.events) (void)hipEventDestroy(e);
  if (c->h_status) (void)hipHostFree(c->h_status);
  if (c->h_report) (void)hipHostFree(c->h_report);
  if (c->h_xfer) (void)hipHostFree(c->h_xfer);
  delete c;
}

}  // extern "C"
