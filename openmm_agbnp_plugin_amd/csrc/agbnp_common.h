// Shared definitions for the gfx950 AGBNP engine (device + host).
//
// Model constants follow the reference's macros bit-for-bit: float literals promoted to double
// (gaussvol/gaussvol.h:46-63, openmmapi/include/AGBNPForce.h:14-33, openmmapi/include/AGBNPUtils.h:124-126).
#pragma once
#include <cfloat>
#include <cmath>
#include <cstdint>

namespace agbnp {

constexpr double kKFC = (2.2269859253f);            // gaussvol.h:46
constexpr double kMinGvol = FLT_MIN;                // gaussvol.h:52
constexpr int kMaxOrder = 8;                        // gaussvol.h:55
constexpr double kVolMinA = (0.01f * (0.001f));     // gaussvol.h:62  (float product)
constexpr double kVolMinB = (0.1f * (0.001f));      // gaussvol.h:63
constexpr double kRadiusIncrement = (0.5f * (0.1f));  // AGBNPForce.h:25
constexpr double kHBRadius = (1.4 * (0.1f));        // AGBNPForce.h:33
constexpr double kSolventRadius = (1.0 * (0.1f));   // AGBNPForce.h:30
constexpr double kI4MaxA = 2.0;                     // AGBNPUtils.h:124
constexpr int kI4Nodes = 16;                        // AGBNPUtils.h:126
// Row stride of the device copy of the tables, in 16-byte entries: one more than the knots, so that row r starts 4 r LDS
// banks on from row 0 and lanes that look up the same knot of different type pairs do not meet in one bank
constexpr int kLutStride = kI4Nodes + 1;
constexpr long kRadiusPrecision = 10000;            // AGBNPUtils.h:155
constexpr double kPi = 3.14159265358979323846;

// GB prefactor (ReferenceAGBNPKernels.cpp:465-468)
constexpr double kToKjMol = 4.184 * 332.0 / 10.0;
constexpr double kDielFactor = kToKjMol * (-0.5) * (1. / 1.0 - 1. / 80.0);

// Header of the stored overlap-tree topology of one work slot = one forest of up to 8 subtrees (written by the build
// kernel, consumed by the pseudo-volume pass, which walks the same slots: one load tells it everything).
struct SubtreeHeader {
  int nnodes;       // nodes of the forest including the level-1 roots (0: not built)
  int natoms;       // local atoms (roots first, then the level-2 partners of root 0, of root 1, ...)
  int nroots;       // subtrees in the forest
  int npairs;       // (atom, node) membership pairs stored for the replay
  int partners[8];  // level-2 partners of every root
};

// Level-2 neighbour masks: for every heavy atom i and every block J of 64 heavy atoms (J >= the block of i), one
// 64-bit word whose bit b says "heavy atom 64 J + b is YOUNGER than i (larger index) and inside the conservative
// overlap cutoff".  Written by the tile workgroups of the k_prep launch (one per 64x64 tile, I <= J, each position
// read once per tile), laid out [J][i] so that a tile stores 64 consecutive words; the tree workgroup of root i reads
// its row of <= nhb words in one round trip and expands the set bits into the near-candidate list.  (Every tree
// workgroup sweeping all younger positions itself moved 50 MB through the L2 per evaluation of 1dwc -- the sweep
// was bound by that, not by its arithmetic.)

constexpr int kMaxItems = 8;  // work items (subtrees or parts of one) per work slot = forest
constexpr int kRowStride = 16;  // a work slot's row: kMaxItems items, their number, padding -- 64 bytes, ONE load instruction

// status/overflow word indices (device int array of kStatTotalWords).
// Words [0, kStatEvalWords) belong to ONE evaluation: k_prep clears them.  The words from kStatEvalSeq on are STICKY:
// they survive from one evaluation to the next, so that a caller who queues many evaluations (or replays a graph)
// before agbnp_hip_finish still learns about every one that overflowed; only agbnp_hip_finish resets them.
enum StatusWord {
  kStatNodeOverflow = 0,   // a subtree needed more than NCAP nodes
  kStatAtomOverflow = 1,   // a node had more than ACAP children / level-2 partners
  kStatPseudoQueue = 2,    // work queue of k_tree_pseudo: forests handed out beyond the first one of every workgroup
  kStatMaxNodes = 3,       // max nodes of any subtree (diagnostic)
  kStatMaxAtoms = 4,       // max local atoms of any subtree (diagnostic)
  kStatCavityQueue = 5,    // work queue of k_tree_cavity (same)
  kStatRowOverflow = 6,    // a neighbour row of the row-form pair stages outgrew its stride (the host falls back to the tiles)
  kStatTotalNodes = 7,     // total nodes (all subtrees)
  kStatPackOverflow = 8,   // a forest of several subtrees did not fit: the packing mispredicted (repeat unpacked)
  kStatForests = 9,        // forests of the evaluation (diagnostic)
  kStatSplitWanted = 10,   // a work item that was ALONE in its store outgrew it while its subtree was shared by fewer than four
                           // items: the largest such part count (0: none).  Counted as a packing overflow: the repeat shares
                           // the subtree among more items instead of moving the whole system to a larger store
  kStatOrderStale = 11,    // agbnp_hip_execute_openmm: the context's atom order is not the one the engine's particle -> slot map
                           // was built for (OpenMM has reordered its atoms): evaluation void, the map is rebuilt, the host repeats
  kStatForestOverflow = 12,  // the part of kStatPackOverflow that IS a misprediction of the packing: forests of SEVERAL work items
                           // that outgrew their store (low half: by nodes, high half: by local atoms) (a lone splittable item is counted in kStatPackOverflow only: the
                           // capacity the packing assumes is not tightened for it)
  kStatMaskAging = 13,     // five-launch mode: a heavy atom is more than a quarter of the neighbour masks' skin from where it was when
                           // they were laid down (the evaluation is good; the masks are laid down anew in its Born launch)
  kStatSpareForests = 14,  // forests that outgrew their store inside k_tree_cavity and were HEALED there: the workgroup built the forest's
                           // work items again in smaller sets, the sets behind the first into spare work slots (numbered from
                           // max(forests, forest workgroups of the launch) on; k_tree_pseudo replays them through its queue).  The
                           // evaluation is complete; the packing's bookkeeping reads the word as "plan anew"
  kStatEvalWords = 15,     // ---- everything below is sticky
  kStatEvalSeq = 15,       // evaluations enqueued since the last agbnp_hip_finish
  kStatBadCount = 16,      // ... of which this many overflowed: their forces and energy were WITHHELD from the caller
  kStatStickyNode = 17,    // OR of the per-evaluation overflow words over those evaluations
  kStatStickyAtom = 18,
  kStatStickyPack = 19,
  kStatStickyRow = 20,
  kStatStickyOrder = 21,   // (bit 1: five-launch mode, the neighbour masks had gone stale)
  kStatStickySplit = 22,   // MAX of kStatSplitWanted over those evaluations
  kStatStickyForest = 23,  // bit 0: a forest of several items outgrew its NODES, bit 1: its local ATOMS (diagnostic: scalar 15)
  kStatStickyHealed = 24,  // forests healed inside k_tree_cavity over those evaluations (diagnostic: scalar 17; nothing was withheld for them)
  kStatWords = 25,
  kStatBadBitmap = 25,     // bit k of the bitmap: evaluation k since the last finish was withheld (k < kStatBadBits)
  kStatBadBits = 2048,
  kStatTotalWords = kStatBadBitmap + kStatBadBits / 32
};

// kernel ids of one evaluation, in launch order (bench/profiling support)
enum KernelId {
  kKPrep = 0, kKTreeCavity, kKBornTiles, kKGbTiles, kKDbornTiles, kKTreePseudo, kKOutputs,
  kKBornRows, kKDbornRows,  // row form of the two range-limited stages (take the place of the two tile kernels)
  kKGbRows,                 // row form of the GB stage (fast mode only)
  kKernelCount
};

}  // namespace agbnp
