/* ============================================================================================
 * agbnp_hip.h -- C ABI of the MI355X (gfx950) AGBNP / GaussVol force engine (libagbnp_hip.so).
 *
 * This is the drop-in boundary for ONE hot path of Gallicchio-Lab/openmm_agbnp_plugin: the work
 * behind  AGBNPPlugin::CalcAGBNPForceKernel  (reference: openmmapi/include/AGBNPKernels.h:19-47)
 * for AGBNPForce version 0 (GaussVol/GVolSA) and version 1 (AGBNP1).  Each entry point names the
 * reference interface it replaces.  Plain pointers and sizes only; no C++/torch/OpenMM types.
 *
 * Units: nm, kJ/mol, kJ/mol/nm, elementary charge -- the units of AGBNPForce::addParticle
 * (reference: openmmapi/include/AGBNPForce.h:66-77).
 *
 * Every function returns AGBNP_HIP_OK (0) or an error code; the message is available from
 * agbnp_hip_last_error().  No exception crosses this boundary.
 * ============================================================================================ */
#ifndef AGBNP_HIP_H_
#define AGBNP_HIP_H_

#ifdef __cplusplus
extern "C" {
#endif

typedef struct agbnp_hip_context agbnp_hip_context;

enum agbnp_hip_status {
  AGBNP_HIP_OK = 0,
  AGBNP_HIP_ERR_INVALID_ARGUMENT = 1, /* bad sizes / null pointers / illegal version                */
  AGBNP_HIP_ERR_PARAMETERS = 2,       /* the reference would throw OpenMMException for these params  */
  AGBNP_HIP_ERR_DEVICE = 3,           /* a HIP runtime call failed or no gfx950 device is available  */
  AGBNP_HIP_ERR_CAPACITY = 4,         /* an overlap subtree exceeded the largest supported capacity  */
  AGBNP_HIP_ERR_TIMEOUT = 5           /* agbnp_hip_wait_verdict: the device has not got that far yet */
};

/* nonbonded methods, values of AGBNPForce::NonbondedMethod (openmmapi/include/AGBNPForce.h:44-59) */
enum agbnp_hip_nonbonded_method { AGBNP_HIP_NoCutoff = 0, AGBNP_HIP_CutoffNonPeriodic = 1, AGBNP_HIP_CutoffPeriodic = 2 };

/* Replaces ReferenceCalcAGBNPForceKernel::initialize(system, force)
 * (platforms/reference/src/ReferenceAGBNPKernels.cpp:58-137): takes the per-particle parameters of
 * AGBNPForce::getParticleParameters (radius, gamma, vdw_alpha, charge, ishydrogen), the version
 * (0 = GVolSA, 1 = AGBNP1; 2 = AGBNP2 is out of scope -> AGBNP_HIP_ERR_INVALID_ARGUMENT), the nonbonded method
 * and cutoff (accepted and stored; like the Reference platform the engine evaluates all pairs --
 * see DESIGN.md), builds the I4 tables and uploads everything to device `device`.
 * Fails with AGBNP_HIP_ERR_PARAMETERS and the reference's message if heavy-atom gammas differ
 * (ReferenceAGBNPKernels.cpp:109-116). */
int agbnp_hip_create(agbnp_hip_context** out, int num_particles, const double* radius, const double* gamma,
                     const double* vdw_alpha, const double* charge, const int* ishydrogen, int version,
                     int nonbonded_method, double cutoff_distance, int device);

/* Replaces ReferenceCalcAGBNPForceKernel::copyParametersToContext (ReferenceAGBNPKernels.cpp:1796-1815):
 * gamma, alpha and charge may change; a changed particle count, radius (squared difference > 1e-6) or
 * heavy->hydrogen flip fails with AGBNP_HIP_ERR_PARAMETERS and the reference's message.
 * Drains this context's own stream and every caller stream it has enqueued on since the last agbnp_hip_finish() -- not the
 * device: other contexts keep running -- and rewrites the device copies in place, at unchanged addresses (a captured HIP
 * graph of this context stays valid). */
int agbnp_hip_update_parameters(agbnp_hip_context* ctx, int num_particles, const double* radius, const double* gamma,
                                const double* vdw_alpha, const double* charge, const int* ishydrogen);

/* Replaces ReferenceCalcAGBNPForceKernel::execute (ReferenceAGBNPKernels.cpp:139-149) with the CPU
 * platform's data conventions: positions[3N] in, forces ACCUMULATED into forces[3N] (+=), energy
 * RETURNED in *energy.  Host buffers; synchronous.  Subtree-capacity overflow is handled inside
 * (the evaluation is repeated with the next larger kernel variant). */
int agbnp_hip_execute_host(agbnp_hip_context* ctx, const double* positions, double* forces, double* energy);

/* Device-resident variant for GPU platforms (the data convention of the reference's OpenCL platform,
 * platforms/opencl/src/OpenCLAGBNPKernels.cpp:541-556: forces and energy are ADDED to device
 * buffers, nothing is returned).  d_positions[3N], d_forces[3N], d_energy[1] are FP64 device
 * pointers on the context's device; `stream` is a hipStream_t (NULL = the context's own stream).
 * Asynchronous: FIVE kernel launches for version 1 (since round 5; six where the five-launch mode does not apply -- see
 * below --, one more on systems whose forests do not fit one round of resident tree workgroups), TWO for version 0 (three), no
 * host synchronisation, no allocation once the context has run on its current capacity variant -- any number of
 * evaluations may be queued, or captured into a HIP graph and replayed, before agbnp_hip_finish().
 *
 * Five-launch mode (FP64 row form of the pair stages for version 1, every LDS-resident capacity variant, both device-resident
 * entry points; AGBNP_HIP_FIVE_LAUNCHES=0 turns it off).  There is no preparation launch: the tree launch reads d_positions itself and finds every heavy atom's level-2
 * neighbours through masks that were laid down with a skin (0.08 nm, AGBNP_HIP_MASK_SKIN) at an earlier evaluation; the
 * device renews the masks by itself when a heavy atom has used up a QUARTER of the skin (that evaluation is still exact).
 * What a caller has to know: an evaluation whose positions differ from those of the evaluation before it by more than HALF
 * the skin (0.04 nm) for some heavy atom -- a jump: a minimiser's long step, a Monte-Carlo move, unrelated geometries one
 * after the other -- may have been built from masks that no longer cover it.  It is then WITHHELD exactly like an
 * evaluation that overflowed (nothing added, logged, scalar 15 reports kind 16), the device has already renewed the masks
 * at its positions, and the repeat is right: agbnp_hip_execute_host() repeats by itself, the other entry points' callers
 * through the protocol they already have.  MD steps are two orders of magnitude below the threshold.  Captured graphs
 * keep the mode: from a context's first stream capture on, the device itself names the set of accumulators an evaluation
 * works on, so a replayed evaluation alternates like eager ones do (agbnp_hip_generation() changes once, at that capture).
 * (Run one eager evaluation before the first capture, as for the capacity variant: a context that has never evaluated lays
 * its first masks down with a launch of its own, and a graph that captured that launch repeats it at every replay.)
 * The mode ends for good, silently, where it cannot hold: the diagnostic self volumes, the 32 768-node store in HBM (capacity
 * variant 4), and a stream capture of a context whose pair stages are not the FP64 row form (version 0, the deterministic mode,
 * AGBNP_HIP_ROWS=0, the fast mode's single-precision rows: those keep the mode in eager launches only).  Scalar 16 says which
 * path runs (5 or 6 launches; version 0: 2 or 3).  Since round 6 agbnp_hip_execute_openmm() runs in the mode too (the tree
 * launch reads the context's posq at the context's slots).
 *
 * Forests that outgrow their store are healed inside the tree launch (round 6): the packing of several subtrees into one
 * LDS store is planned from an earlier evaluation's shapes; a forest that does not fit any more used to void the evaluation
 * (withheld, see below) -- now its workgroup builds it again in smaller sets, the evaluation is complete, scalar 17 counts
 * the sets.  What can still be withheld: a subtree that needs the next capacity variant, a neighbour row beyond its walk, a
 * jump (above), a reordered OpenMM context.
 *
 * Overflow contract.  The overlap-tree stage works in fixed-capacity LDS stores; an evaluation whose trees
 * outgrow them (or whose forest packing mispredicted) is INCOMPLETE.  Such an evaluation adds NOTHING to
 * d_forces / d_energy -- the outputs are gated on the device -- and is entered in a log on the device that only
 * agbnp_hip_finish() reads and clears.  So after any number of queued evaluations the caller's buffers hold
 * exactly the sum of the complete ones, and finish() says which ones are missing (the analogue of the reference
 * OpenCL platform's PanicButton protocol, OpenCLAGBNPKernels.cpp:3599-3634: forces invalidated, step retried). */
int agbnp_hip_execute_device(agbnp_hip_context* ctx, const double* d_positions, double* d_forces, double* d_energy,
                             void* stream);

/* The same evaluation in the data conventions of an OpenMM GPU ComputeContext -- what the reference's OpenCL platform
 * kernel reads and writes (platforms/opencl/src/OpenCLAGBNPKernels.cpp:541-556, kernels/GVolReduceTree.cl:92-121):
 *   d_posq            real4 {x, y, z, q} per atom in the CONTEXT's atom order, float4 (posq_is_double = 0) or double4
 *   d_posq_correction float4 low-order parts in mixed precision (position = posq + correction), else NULL
 *   d_atom_index      [N] context slot -> particle index of the System/Force (ComputeContext::getAtomIndexArray);
 *                     NULL = identity
 *   d_force_buffer    the context's 64-bit fixed-point force buffer (value * 2^32), three planes of
 *                     padded_num_atoms words [x | y | z] in context order; forces are ADDED with integer atomics
 *   d_energy_buffer   the context's energy accumulator, double (energy_is_double) or float; the energy is ADDED to
 *                     element energy_slot; NULL = no energy output
 * The same launches as agbnp_hip_execute_device: the engine's first kernel reads posq itself and its last writes the
 * context's buffers.  For that the engine keeps maps of the context's atom order (particle -> slot), built by one small launch
 * when a d_atom_index array is first seen.  OpenMM reorders its atoms now and then (same array, new contents): the first
 * kernel checks the maps against d_atom_index in every evaluation, so the first evaluation after a reorder is WITHHELD
 * like one that overflowed -- nothing of it reaches the context's buffers, agbnp_hip_finish() / agbnp_hip_wait_verdict()
 * report it -- the maps are rebuilt in the next call, and the repeat is right.  Same asynchronous contract, same
 * agbnp_hip_finish(). */
int agbnp_hip_execute_openmm(agbnp_hip_context* ctx, const void* d_posq, int posq_is_double, const void* d_posq_correction,
                             const int* d_atom_index, int padded_num_atoms, long long* d_force_buffer, void* d_energy_buffer,
                             int energy_is_double, int energy_slot, void* stream);

/* Tells the engine that the contents of the d_atom_index array it last saw have changed (OpenMM has reordered its atoms):
 * the next agbnp_hip_execute_openmm() rebuilds its maps first, and no evaluation is lost to the check.  Optional -- without
 * it the first evaluation after a reorder is withheld and repeated, see above. */
int agbnp_hip_atom_order_changed(agbnp_hip_context* ctx);

/* Waits for `stream`, then reads and clears the overflow log.  *must_repeat = the number of evaluations enqueued
 * since the previous agbnp_hip_finish() whose outputs were withheld (0: every one is complete and in the caller's
 * buffers).  If it is not 0 the context has already prepared the repeat (one subtree per workgroup, and the next
 * larger capacity variant if a single subtree did not fit): the caller runs those evaluations again -- their
 * positions in enqueue order come from agbnp_hip_withheld_evaluations() -- and calls finish() again.
 * Returns AGBNP_HIP_ERR_CAPACITY if a subtree exceeds the largest variant. */
int agbnp_hip_finish(agbnp_hip_context* ctx, void* stream, int* must_repeat);

/* Which evaluations the LAST agbnp_hip_finish() found withheld: writes up to `capacity` indices (0 = the first
 * evaluation enqueued through a device-resident entry point after the finish before it; the log's bitmap names the first
 * 2048 evaluations since that finish -- fewer by the agbnp_hip_execute_host calls in between, whose evaluations take the first
 * entries -- later ones are only counted)
 * and returns their total number (-1: null context). */
int agbnp_hip_withheld_evaluations(const agbnp_hip_context* ctx, int* indices, int capacity);

/* Non-blocking look at the overflow log (no device call, no synchronisation): how many of the evaluations enqueued since
 * the last agbnp_hip_finish() have COMPLETED on the device and how many of those were withheld, read from pinned host
 * memory that the device writes at the end of every evaluation.  A caller that must not stall its stream every step (an MD
 * loop) polls this after each enqueue and calls agbnp_hip_finish() only when *withheld is non-zero (or now and then: the
 * log names the first 2048 evaluations since the last finish, see agbnp_hip_withheld_evaluations).  What it learns is at least one evaluation old.  The reference's GPU platform does a
 * blocking read every step instead (OpenCLAGBNPKernels.cpp:3599-3634).  AGBNP_HIP_ERR_DEVICE: no pinned memory. */
int agbnp_hip_poll(const agbnp_hip_context* ctx, int* evaluations_completed, int* withheld);

/* The strict per-step check without draining the stream.  Blocks the calling HOST thread (no device call, nothing is
 * synchronised) until the device has delivered its verdict on `evaluations` evaluations since the last agbnp_hip_finish()
 * (0 or less: on every evaluation enqueued through this library's entry points since then; a caller that replays captured
 * graphs passes its own count) or `timeout_seconds` have passed, by watching the pinned status words of agbnp_hip_poll().
 * An evaluation's verdict -- complete, or withheld because a tree outgrew its store -- is final when its tree stage has
 * ended, about three quarters into the evaluation, and is written there; the forces follow on the stream, gated on the
 * device by the same words.  So *withheld == 0 on return means every one of those evaluations WILL add its forces and
 * energy, in stream order, and the caller may go on enqueuing work behind them; *withheld != 0 means what it means
 * after agbnp_hip_finish(): call it, then repeat.  This is the reference GPU platform's protocol (a blocking read of the
 * overflow flag in every step, OpenCLAGBNPKernels.cpp:3599-3634) with the same guarantee and without its pipeline drain:
 * the host waits for a word, not for the stream.  AGBNP_HIP_ERR_TIMEOUT: not there yet (the outputs are valid as far as
 * they go); AGBNP_HIP_ERR_DEVICE: no pinned memory. */
int agbnp_hip_wait_verdict(const agbnp_hip_context* ctx, int evaluations, double timeout_seconds, int* evaluations_completed,
                           int* withheld);

/* Changes whenever kernel arguments that a captured HIP graph of agbnp_hip_execute_device has frozen go stale:
 * after a finish() that raised the capacity variant or grew the scratch pools.  A caller that replays a graph
 * compares the value at capture time with the current one after every finish() and re-captures on a difference.
 * agbnp_hip_update_parameters() does NOT change it: parameters are rewritten in place at unchanged addresses. */
unsigned agbnp_hip_generation(const agbnp_hip_context* ctx);

/* Evaluation mode (default 0 = the Reference platform's semantics: the descreening sums reach as far as the tables,
 * 2 nm, GB meets ALL pairs, nonbonded method and cutoff are inert -- parity target of this engine).
 * AGBNP_HIP_MODE_FAST = the semantics of the reference's GPU (OpenCL) platform: every pair stage of AGBNP1 -- Born-radius
 * descreening sums, GB pair energy / direct forces / Y sums, chain-rule W/U sums and forces -- only meets pairs with
 * r^2 < cutoff_distance^2 (platforms/opencl/src/kernels/AGBNPBornRadii.cl:268,430, AGBNPGBEnergy.cl:145,186).  As on
 * that platform the cutoff only exists for a nonbonded method other than NoCutoff (USE_CUTOFF, OpenCLAGBNPKernels.cpp:487,
 * 1149-1150): with NoCutoff the fast mode truncates nothing and computes exactly what the Reference mode does.
 * CutoffPeriodic is rejected (AGBNP_HIP_ERR_INVALID_ARGUMENT): no periodic box crosses this boundary.  Tiles beyond the
 * cutoff are culled.
 * FP64 throughout; version 0 has no pair stage and is unaffected.  For comparisons with the OpenCL plugin; results
 * differ from the Reference platform by the truncated pairs.  Drains this context's own stream (not the device); bumps
 * agbnp_hip_generation().
 *
 * AGBNP_HIP_MODE_DETERMINISTIC (may be combined with either): bit-identical results from run to run.  By default sums
 * that many workgroups contribute to are FP64 atomics whose order is not fixed, so results differ by ~1e-16 relative
 * between runs.  In this mode every term that enters an order-dependent sum is first rounded to a fixed quantum
 * (2^-34 kJ/mol/nm for forces, 2^-52 nm^3 for self volumes, 2^-44 / 2^-40 for the pair-stage sums, 2^-36 kJ/mol for
 * energies): sums of such terms are exact in FP64, hence independent of their order.  Results stay within 1e-8 of the
 * default mode's (far inside the 1e-4 parity bar).
 *
 * AGBNP_HIP_MODE_SINGLE (only together with AGBNP_HIP_MODE_FAST): the pair stages compute their pair terms in single
 * precision (hardware exp2 / rsqrt; positions relative to a local origin before they are rounded), as the reference's
 * OpenCL platform does in its default precision (AGBNPBornRadii.cl:181-430, AGBNPGBEnergy.cl are all-float).  In the row
 * form -- where the fast mode normally runs -- that is all three of them: descreening sums, GB, chain rule (the spline table
 * as FP32 in LDS); in the tile form (no neighbour lists: NoCutoff, AGBNP_HIP_ROWS=0) the GB strips only.  The sums of a
 * slice or tile run in FP32, everything that leaves a wave, the Born-radius algebra, the trees and the energies stay
 * FP64.  Forces differ from the FP64 fast mode by ~5e-4 kJ/mol/nm, the energy by ~3e-3 kJ/mol on a 4000-atom protein.
 * Rejected without AGBNP_HIP_MODE_FAST: the Reference semantics are FP64. */
enum agbnp_hip_mode { AGBNP_HIP_MODE_REFERENCE = 0, AGBNP_HIP_MODE_FAST = 1, AGBNP_HIP_MODE_DETERMINISTIC = 2, AGBNP_HIP_MODE_SINGLE = 4 };
int agbnp_hip_set_mode(agbnp_hip_context* ctx, int mode);
int agbnp_hip_get_mode(const agbnp_hip_context* ctx);

/* Diagnostics of the LAST completed evaluation (test support; mirrors the quantities the reference
 * prints at verbose_level > 0, ReferenceAGBNPKernels.cpp:333-352,459-462,519).
 * scalars: 0 E_vol1  1 E_vol2  2 E_atom (vdW + GB self)  3 E_GB pair  4 max subtree nodes
 *          5 total tree nodes  6 kernel variant  7 max local atoms  8 work slots (forests) planned for the next evaluation
 *          9 1 if the range-limited pair stages run in row form (neighbour rows with a skin, rebuilt on the device when an
 *            atom has moved more than half the skin; Reference mode, version 1)  10 builds of those rows so far
 *          11 forest packing: how far the assumed store capacity is tightened (0 = not at all; one step of 15 % when healed
 *             forests keep coming or one could not be healed, given back after clean plans in a row -- more of them every time)  12 evaluations since the packing was planned
 *          13 entries per slice of a neighbour row (one wave of a row launch walks one slice; tuned on the device)
 *          15 why the last agbnp_hip_finish() withheld evaluations: 1 a subtree outgrew the store's nodes, 2 its local atoms,
 *             4 a forest packing mispredicted, 8 a neighbour row outgrew its walk, 16 the context reordered its atoms,
 *             32 / 64 a forest of several work items outgrew its nodes / its local atoms (the two kinds of 4);
 *             bits 8.. the part count of a lone work item that asked for its subtree to be shared further
 *          16 kernel launches of an evaluation as the context runs now: version 1: 5 (five-launch mode) or 6; version 0: 2 or 3
 *          17 forests that outgrew their store and were healed inside the tree launch (built again in smaller sets: the
 *             evaluation is complete, nothing is withheld for them) over the evaluations the last agbnp_hip_finish() covered
 *          14 forest packings planned so far (a packing in use is planned anew every AGBNP_HIP_REPLAN_EVERY-th evaluation,
 *             default 16, or when the trees have drifted from the shapes it was planned for)
 * vectors (length N, atom order): 0 self volume (vdW radii)  1 Born radius  2 volume scaling factor
 *          3 self volume (enlarged radii)
 *          4 / 5 nodes / local atoms of the overlap subtree rooted at the atom (tree shape, capacity planning) */
int agbnp_hip_set_diagnostics(agbnp_hip_context* ctx, int enabled); /* vector 3 is only collected when enabled */
int agbnp_hip_get_scalar(agbnp_hip_context* ctx, int which, double* value);
int agbnp_hip_get_vector(agbnp_hip_context* ctx, int which, double* out);

/* I4 lookup tables as uploaded (test support): sizes, then y/y2 [nscreened*nscreener*16] and per-atom types. */
int agbnp_hip_get_table_sizes(agbnp_hip_context* ctx, int* nscreened, int* nscreener);
int agbnp_hip_get_tables(agbnp_hip_context* ctx, double* y, double* y2, int* type_screened, int* type_screener);

/* Host-only table builder (no device needed; test support for the host logic): the radius typing and
 * the 16-node spline tables of AGBNPI42DLookupTable (openmmapi/src/AGBNPUtils.cpp:134-214).  y/y2 must hold
 * table_capacity doubles; fails with AGBNP_HIP_ERR_INVALID_ARGUMENT if nscreened*nscreener*16 exceeds it. */
int agbnp_hip_host_tables(int num_particles, const double* radius, const int* ishydrogen, int* nscreened, int* nscreener,
                          double* y, double* y2, int table_capacity, int* type_screened, int* type_screener);

/* Per-kernel timing (bench support).  When enabled, a hipEvent is recorded on the evaluation's stream in front
 * of every kernel; agbnp_hip_finish()/execute_host() turn them into accumulated milliseconds per kernel.
 * Enabling or disabling resets the accumulators.  total_ms / launches must hold agbnp_hip_num_kernels() items. */
int agbnp_hip_set_profiling(agbnp_hip_context* ctx, int enabled);
int agbnp_hip_num_kernels(void);
const char* agbnp_hip_kernel_name(int index);
int agbnp_hip_get_kernel_times(agbnp_hip_context* ctx, double* total_ms, long* launches);

int agbnp_hip_num_particles(const agbnp_hip_context* ctx);
int agbnp_hip_version(const agbnp_hip_context* ctx);

/* Message of the last error on this context; with ctx == NULL, of the last failed agbnp_hip_create() on the
 * calling thread (thread-local). */
const char* agbnp_hip_last_error(const agbnp_hip_context* ctx);

void agbnp_hip_destroy(agbnp_hip_context* ctx);

/* Number of HIP devices visible to this process (0 if none / runtime unavailable). */
int agbnp_hip_device_count(void);

/* What this library was built from: the first 16 hex digits of the SHA-256 of the engine's sources (csrc/Makefile), "unknown"
 * for a build made another way.  No counterpart in the reference; measurement support: scripts/profile_round.sh stores it
 * beside every rocprofv3 summary under profiles/, bench.py compares it with the library it is running (profile_head). */
const char* agbnp_hip_build_id(void);

#ifdef __cplusplus
}
#endif
#endif /* AGBNP_HIP_H_ */
