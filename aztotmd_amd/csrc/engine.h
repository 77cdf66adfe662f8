// Step driver + device management of one rank: the MI355X-native counterpart of the reference's
// init_cudaMD / md_to_host / free_device_md (cuInit.cu:756-1374) and of the loop body of
// main.cu:281-410.  One HIP stream, no host synchronisation inside a step (the reference blocks on
// cudaThreadSynchronize 11 times per step, main.cu:290-383), optional hipGraph replay.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "device_md.h"
#include "model.h"
#include "msg_layout.h"

namespace aztot {

struct Counts;
struct PairLists;

struct KernelTimer
{
    std::string name;
    double ms = 0.0;
    long long calls = 0;
};

}  // namespace aztot
#include "exchange.h"
namespace aztot {

// Measurement switches, read once from the environment variable AZTOT_DEBUG (a bit mask) when an engine is created.  Every bit but DBG_STAGE_ONLY keeps the
// results unchanged (up to summation order): they select between code paths for A/B timing and let the tests reach paths a normal run rarely takes.
enum DebugBit : unsigned
{
    DBG_BUILD_PHASE_MASK = 3,            // NOT result-preserving (phase timing of k_build_lists): 1 staging only, 2 + candidates and filter, 3 + compaction
    DBG_ALWAYS_CLEANUP = 4,              // the clean-up launch behind every k_pair_list whatever the system size (small systems on one GPU run without it: Engine::choose_optimism)
    DBG_KICK_EVERY_STEP = 128,           // k_integrate2 launched every step
    DBG_LARGE_KICK_PATH = 256,           // the deferred half-kick of large systems whatever the size
    DBG_GENERIC_PAIR = 512,              // the generic (switch-based) pair body instead of a specialised mode
    DBG_KEEP_VDW_CUT_TEST = 1024,        // Lennard-Jones family: keep the per-pair cut-off test although it always passes (family 6, Engine::construct)
    DBG_STAGE_ONLY = 2048,               // NOT result-preserving: the staging pair kernel stages its tile and stops (phase timing)
    DBG_SLAB_GRAPH = 4096,               // hipGraph replay of a loopback slab rank
    DBG_FIXED_INTERVAL = 8192,           // lazy re-sort at the fixed interval options.sort_every whatever the atoms' speed (exercises the wider-stencil fallback)
    DBG_OVERLAP_HALO = 16384,            // slab ranks: coordinate exchange of plain steps on a second stream beside the interior cells' pair forces (measured slower)
    DBG_NO_LISTS = 32768,                // no pair lists: the steps between two rebuilds stage every cell
    DBG_SHORT_LISTS = 65536,             // pair lists capped at 14 iterations (part of a liquid's cells then goes through the clean-up launch)
    DBG_NO_FUSE_NEXT = 131072,           // next-step fusion off ...
    DBG_FUSE_NEXT = 262144,              // ... or on, whatever the system size
    DBG_CHECK_EVERY_ATOM = 524288,       // plain steps check every atom against its reference position (no displacement bound)
    DBG_LIST_STATS = 2097152,            // list statistics on stderr (with AZTOT_VERBOSE)
    DBG_KICK_POST_SPLIT = 4194304,       // second half-kick and radiative thermostat as two launches
    DBG_GRAPH_ALWAYS = 8388608,          // hipGraph replay also above 500 000 atoms
    DBG_PLAIN_ONE_ATOM = 16777216,       // plain steps integrate one atom per thread
    DBG_NO_BOUNDARY_RADI = 33554432,     // thermostat runs close a step and open the next in two launches
    DBG_ONE_WAVE_PER_CELL = 67108864,    // one wave per cell in the staging kernel whatever the system
    DBG_ENERGIES_EVERY_STEP = 134217728, // = options.energies_every_step
    DBG_SETTLE_EVERY_CALL = 268435456    // every aztot_step call ends with the look / statistics / synchronisation (one GPU defers them to the next look or read)
};

// ---------------------------------------------------------------------------------------------------------------------------------------------------
// The host state machine in one table.  "Forget" = the three assignments `sinceSort_ = 1 << 30; listsValid_ = false;
// carryAgreed_ = false` that make the next step rebuild the cells.  API calls that touch state first complete a deferred call end (Engine::settle).
//
//  member            meaning                                              set by                                      cleared / reset by                         invalidated by (API)
//  ----------------  ---------------------------------------------------  ------------------------------------------  -----------------------------------------  ---------------------------------------
//  unsettled_        the last aztot_step returned before its end           step_body (every call)                      settle (at once when settle_now(), else     every read / write entry point settles
//                    (deferred kick, statistics, look) had happened                                                    by the next reader or the next due look)    first; mark_failed drops it
//  failed_           first error of a step / settle: no more stepping      mark_failed                                 never (the handle is dead for stepping)    -
//  hostStep_         steps launched so far (== DevStats::step when the     launch_step_kernels (+1), run_steps (graph  replay_from_snapshot (snapshot's),          aztot_set_clock; mark_failed re-reads it
//                    stream has drained)                                   replay: + cycle)                            set_clock                                  from the device
//  lazyK_            sort interval in force (1: every step)                adapt_sort_interval (from the longest step  -> 1: set_state with new v / f, a slab      aztot_set_state (v, f); aztot_forces keeps it
//                                                                          seen), constructor (debug: fixed)           violation (replay), never above lazyCap_
//  lazyMeasured_     lazyK_ comes from a measurement                       adapt_sort_interval                         set_state (v, f), slab violation            aztot_set_state (v, f)
//  lazyWindow_       steps between two looks: 8, 16 ... 256 (64 on slabs)  step_body / settle (doubling)               -> 8: set_state (v, f), slab violation      aztot_set_state (v, f)
//  sinceLook_        steps since the last look (runs on across calls)      step_body (+ n)                             every look (-> 0); set_state (v, f)         aztot_set_state (v, f)
//  lazyMargin_       K steps of the longest step may use slack / margin    adapt_sort_interval (x 4/3 per violation)   never shrinks                               -
//  sinceSort_        plain steps since the last rebuild (1 << 30: none)    sort_and_forces (0 at a rebuild, +1 plain)  Forget                                      aztot_forces, aztot_set_state, aztot_set_clock,
//                                                                                                                                                                  replay, list re-allocation, run_steps without carryOn
//  listsValid_       the lists on the device are those of the arrays as    launch_pair (build), prepare_next_call      Forget; sort_and_forces until launch_pair   as sinceSort_
//                    they stand (a call may open with plain steps)                                                     has recorded the new ones; free_lists
//  listsOn_          lists exist and plain steps walk them                 allocate_lists                              free_lists; adapt_sort_interval when most   -
//                                                                                                                      unlisted cells can never be listed
//  candLds_/iterLds_ LDS sizes of k_pair_list / k_build_lists              allocate_lists (= capacities)               adapt_sort_interval, prepare_next_call      - (baked into graphs: destroy_graphs)
//                                                                                                                      (tightened to the largest cell recorded)
//  unlistedState_    cells without a list in the lists in force: 0 ?,      launch_pair (slab: read back behind the     launch_pair at every build (-> 0), replay    -
//                    1 none, 2 some                                        build), prepare_next_call
//  unlistedAtLook_   the last look found cells recorded without a list     adapt_sort_interval                         the next look that saw a recording          -
//  carryAgreed_      slab ranks: all agreed the next call may open with    prepare_next_call (all-reduce)              Forget                                      as sinceSort_
//                    plain steps
//  optimistic_       plain steps run without the clean-up launch           choose_optimism (start of a settled call,   choose_optimism; replay_from_snapshot       - (P_.optimistic is baked into graphs)
//                    (P_.optimistic)                                       after every look)
//  safeLooks_        looks still to spend with the clean-up launch         constructor (1), replay (4 << rollbacks_)   every look (- 1)                            -
//  snap_.valid       a verified snapshot of the dynamic state exists       take_snapshot (start of a settled call,     choose_optimism (not optimistic any more),  aztot_forces, aztot_set_state, aztot_set_clock,
//                                                                          every clean look; kept when younger than    settle (no roll-back mode)                  mark_failed
//                                                                          kSnapshotKeepSteps)
//  stepsSinceSnap_   steps launched since the snapshot                     step_body (+ n), replay (= n)               take_snapshot (-> 0)                        -
//  kickOwed_         the last step's second half-kick has not been         step_body (= lazyKick_ at the end of a      finish_steps (k_integrate2); start of the   -
//                    applied and settle must apply it                      call), replay                               next step_body (the device flag
//                                                                                                                      DevStats::pendingKick makes the next
//                                                                                                                      integrate kernel pay it)
//  preIntegrated_    the step being launched was opened by the previous    launch_pair (next-step fusion),             sort_and_forces (consumes it), run_steps,   -
//                    step's pair kernel / boundary kernel                  launch_step_kernels (k_boundary_radi)       graph capture, replay
//  fuseNext_/fuseNow_/pairClosedStep_/overlapHalo_/candMode_/stepsLeftInRun_   per-launch scratch: set and consumed inside launch_step_kernels / launch_pair
//  graphs_, graphCycle_  captured cycles, valid for one buffer state and   graph_for_state                             destroy_graphs: any change of lazyK_,        aztot_set_clock, aztot_set_state (U / radius
//                    one set of launch parameters                                                                       optimistic_, candLds_ / iterLds_,          first touched), replay
//                                                                                                                      listWaves_, listsOn_, thermoTouched_
//  haloInfoPending_  slab: the boundary ranges of the last sort are still  sort_and_forces                             take_halo_info, prepare_next_call, replay   -
//                    on their way to pinned memory
//  lastStepEquil_    the last step left its kinetic energy in DevStats     launch_step_kernels                         launch_step_kernels                         -
// ---------------------------------------------------------------------------------------------------------------------------------------------------
class Engine
{
public:
    Engine(const Model& model, const aztot_options& opt, int rank, int nranks, Exchanger* xch);
    ~Engine();
    Engine(const Engine&) = delete;

    void step(int nsteps);
    void forces(bool withBonded = true);    // sort + pair forces (+ bonds / angles) on the current positions
    void get_stats(aztot_stats& out);
    void species_crossings(int64_t* out, int cap);
    void md_to_host(aztot_state& out);
    void set_state(const aztot_state& in);
    void get_clock(aztot_clock& out);
    void set_clock(const aztot_clock& in);
    int cell_table(int32_t dims[3], int32_t* cellStart, int capCells, int32_t* atomId, int capAtoms);
    int kernel_times(std::vector<KernelTimer>& out);
    void reset_kernel_times();
    void set_profile(bool on) { settle(); sync(); profile_ = on; }
    int n_atoms_global() const { return model_.nAt; }
    int comm_ranks() const { return xch_ ? xch_->comm_ranks() : 0; }
    void sync_all();                        // everything queued or deferred by earlier calls has happened when this returns (aztot_sync)

private:
    void step_body(int nsteps);
    bool settle_now() const;
    void settle();
    void mark_failed(const char* what) noexcept;
    void regrow_lists(int candCap, int iterCap);
    bool unsettled_ = false;                // the last aztot_step call returned with its end (deferred kick, statistics, look) still to come: Engine::settle
    std::string failed_;                    // first error of a step / settle: the handle no longer steps, reads still work
    void construct();
    void release();
    void destroy_graphs();
    void check_launch(const char* where);
    void choose_cells();
    void allocate();
    void upload_initial();
    void upload_bonded();
    void upload_ewald();
    void launch_ewald();
    void sort_and_forces(int stepMode, bool withBonded = true);   // 0: bin + sort + forces (aztot_forces), 1: a step that re-sorts, 2: a plain step of the lazy re-sort
    void launch_step_kernels();
    bool adapt_sort_interval();
    void look_sync();               // what a look reads (Counts, the list builder's report) copied into pinned memory behind the queued work, then ONE stream synchronisation
    void* hLook_ = nullptr;         // pinned: Counts followed by 16 ints of PairLists::noList
    void run_steps(int nsteps);
    void launch_pair();
    int pair_variant() const;
    void exchange_halo();
    void collect_and_finalize(unsigned slotMask);
    void finish_steps();
    void check_overflow();
    void sync();
    template <typename F> void timed(const char* name, F&& launch);
    AtomArrays& cur() { return buf_[cur_]; }
    AtomArrays& oth() { return buf_[cur_ ^ 1]; }

    Model model_;
    aztot_options opt_;
    unsigned debug_ = 0;            // AZTOT_DEBUG (DebugBit)
    StepParams P_{};
    SpecTable S_{};
    int rank_, nranks_;
    Exchanger* xch_;
    std::unique_ptr<Exchanger> ownedXch_;

    hipStream_t stream_ = nullptr;
    hipStream_t commStream_ = nullptr;      // slab ranks, plain steps: the coordinate exchange runs here while the interior cells' pair forces run on stream_
    hipEvent_t evIntegrated_ = nullptr, evHalo_ = nullptr;
    bool overlapHalo_ = false;              // the pair launch in flight is split: interior cells now, boundary cells once evHalo_ has fired
    int capacity_ = 0;          // atoms that fit in the per-atom arrays (owned + ghosts + slack)
    int nCellAlloc_ = 0;
    int maxBlocks_ = 0;
    AtomArrays buf_[2]{};
    int cur_ = 0;
    std::vector<void*> allocs_;
    DevPot* dPots_ = nullptr;
    int32_t *dCellOf_ = nullptr, *dSlotOf_ = nullptr, *dCellCount_ = nullptr, *dCellStart_ = nullptr;
    int32_t* dChunkTot_ = nullptr;
    int32_t *dTmpId_ = nullptr, *dTmpSrc_ = nullptr, *dTmpCell_ = nullptr, *dCellOfSorted_ = nullptr;
    double* dPartials_ = nullptr;
    DevStats* dStats_ = nullptr;
    Counts* dCounts_ = nullptr;
    double *dPhotons_ = nullptr, *dUvx_ = nullptr, *dUvy_ = nullptr, *dUvz_ = nullptr;
    double* dStage_ = nullptr;      // sub-totals of the two-level collect
    double* dEkGlobal_ = nullptr;   // kinetic energy over all ranks (equilibration scaling only)
    char* dMsg_[4] = {nullptr, nullptr, nullptr, nullptr};   // sendLeft, sendRight, fromLeft, fromRight
    MsgLayout lay_{};
    int pairBlocks_ = 0, pairBlocksUsed_ = 0, splitBlocks_ = 0;
    int blocksEver_ = 0;            // partial-sum entries any launch has booked into so far (what k_collect has to read of the accumulating rows)
    // lazy re-sort (one GPU): the cells are rebuilt only every lazyK_ steps; in between the atoms keep their slots, coordinates stay unwrapped and
    // every step checks that no atom has moved farther than lazySlack_ from where it was sorted (RefPos) - see Engine::step
    bool lazyOn_ = false;
    int lazyK_ = 1;                 // current sort interval (1: every step); adapted after every aztot_step call from the largest step seen
    int lazyCap_ = 32;
    long long lazyViolations_ = 0;
    int lazyWindow_ = 8;               // steps between two looks at the speeds: 8, 16, ... 256
    int sinceLook_ = 0;                // steps since the last look (runs on across calls)
    double lazyMargin_ = 1.15;         // one GPU: K steps of the longest step seen may use slack / lazyMargin_ (1.3 until round 4: 20 000 steps of C4T and 10 000 of C3T run
                                       // without a violation at 1.05 too); widened by every violation (a system that heats up)
    bool lazyMeasured_ = false;     // the interval has been sized from a measurement at least once
    double lastLookLen_ = 0.0;      // the longest step of the window before the last look (0: none to compare with): growing speeds widen the margin
    int sinceSort_ = 1 << 30;       // plain steps since the last sort
    long long rebuilds_ = 0;        // steps that rebuilt the cell list so far
    double lazySlack_ = 0.0;
    RefPos ref_{};
    // lists of the lazy re-sort (pair_tile.hip.h / pair_list.hip.h), recorded by the step that rebuilds the cells and walked by the plain steps: per cell the
    // candidates its tile held, and per atom of the cell its partners among them, dealt evenly to the lanes
    uint32_t* dCandList_ = nullptr;
    int32_t* dListMeta_ = nullptr;
    uint16_t* dPairList_ = nullptr;
    int candLds_ = 0, iterLds_ = 0; // what the LDS tiles of k_pair_list / k_build_lists are sized for (<= the capacities; from the largest cell recorded)
    float4* dRel_ = nullptr;        // [capacity + 64] position relative to the own cell's centre (f32) + cell z index, written by the sort for the list builder
    int candCap_ = 0, iterCap_ = 0; // capacities of the lists per cell (PairLists); grown when too many cells turn out not to fit
    int listGrowths_ = 0;
    int listWaves_ = 1;             // waves per cell in k_pair_list (PairLists::waves)
    size_t listLdsMax_ = 0;         // dynamic LDS this device grants a workgroup
    void allocate_lists(int candCap, int iterCap);
    void free_lists();
    PairLists pair_lists() const;
    double skinTarget_ = 0.0;       // the skin the cells were sized for (0: none)
    int32_t* dNoList_ = nullptr;    // [2]: cells recorded without / with a list since the host last looked
    bool listsOn_ = false;          // plain steps run k_pair_list (switched off when too many cells turn out to keep no list)
    int candMode_ = 0;              // for the pair launch in flight: 0 none, 1 record, 2 plain step of the lazy re-sort
    int halo_[5] = {0, 0, 0, 0, 0};  // slab ranks: ownedBegin, end of the left boundary layers, start of the right ones, ownedEnd, nTotal (after the last sort)
    // read back after every sort without stalling the stream: copied into pinned host memory behind the sort kernels, waited for only when the next plain
    // step needs the numbers (by then the list building and the pair kernel of the sort step are queued behind the copy)
    int32_t* hHalo_ = nullptr;       // pinned: {cellStart of the two boundary layers, ownedBegin, ownedEnd, nTotal}
    int32_t* dHaloInfo_ = nullptr;   // the same on the device, written by k_rank_gather
    hipEvent_t evHaloInfo_ = nullptr;
    bool haloInfoPending_ = false;
    void take_halo_info();
    void post_count_exchange();
    void adopt_halo_info(const int32_t* h);
    int graphCycle_ = 0;            // steps held by the captured graphs
    BondedTables bonded_{};         // all-null when the model has no bonds / angles
    bool hasBonded_ = false;
    bool thermoTouched_ = false;    // the caller set U / radius on a run whose model does not use them: keep them attached to their atoms
    bool lazyKick_ = false;         // large plain-NVE runs: integrate2 is folded into the next step's integrate1 (flushed by finish_steps)
    bool kickOwed_ = false;
    // Equilibration rescaling (control.txt: nequil / eqfreq) acts on a handful of steps only - step numbers the host can tell (hostStep_ mirrors DevStats::step) -,
    // so only those steps take the path that sums the kinetic energy on the device, decides a factor and scales; every other step of a thermostat run takes the
    // short path of a run without equilibration (case study 2: 6 launches per step -> 2)
    long long hostStep_ = 0;        // steps launched so far (== DevStats::step once the stream has drained)
    bool lastStepEquil_ = false;    // the last step launched left its kinetic energy in DevStats::local (k_reduce_kin), not in the partial sums
    bool equil_phase() const { return P_.nEq > 0 && hostStep_ < P_.nEq; }
    bool fuseEpilogue_ = false;     // small plain-NVE runs: the tile kernel's epilogue applies integrate2
    bool fuseNow_ = false;          // the pair launch in flight also does integrate2's job (plain NVE steps, tile kernel)
    bool ekinFromPair_ = false;     // where the last step left its kinetic-energy partials
    EwaldTables ew_{};              // reciprocal-space Ewald sum ('elec pme')
    bool hasEwald_ = false;

    // statistics window for the wall-momentum pressure (main.cpp:143-163)
    double lastMom_[6] = {0, 0, 0, 0, 0, 0};
    double pressure_ = 0.0;
    long long lastPresStep_ = 0;

    // profiling
    bool profile_ = false;
    std::vector<KernelTimer> timers_;
    std::map<std::string, int> timerIndex_;
    struct PendingEvent { int idx; hipEvent_t a, b; };
    std::vector<PendingEvent> pending_;
    std::vector<hipEvent_t> eventPool_;
    void drain_events();

    // hipGraph replay of a cycle of steps.  Kernel arguments are baked in at capture time, so a graph is valid for the buffer state it was captured in
    // (which AtomArrays is current, which coordinate arrays each of them holds: the sort ping-pongs the buffers, the fused next-step epilogue swaps
    // coordinate arrays); it also remembers the state it leaves behind
    struct BufState { int cur; double* xyz[2][3]; double* alt[2][3]; };
    // ... and what the host notes down while it launches a step for whoever reduces the statistics afterwards (finish_steps): a replayed cycle does not pass
    // through that host code, so the slot carries the notes of its last step (found by the call-pattern fuzz of round 4: aztot_forces - whose staging launch
    // books into more partial-sum slots than the list kernel - followed by a call served entirely by replays summed stale slots into the energies)
    struct LaunchNotes { int pairBlocksUsed = 0, blocksEver = 0; bool ekinFromPair = false, lastStepEquil = false; };
    struct GraphSlot { BufState before, after; LaunchNotes notes; hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr; };
    LaunchNotes launch_notes() const { LaunchNotes n; n.pairBlocksUsed = pairBlocksUsed_; n.blocksEver = blocksEver_; n.ekinFromPair = ekinFromPair_; n.lastStepEquil = lastStepEquil_; return n; }
    void adopt_launch_notes(const LaunchNotes& n) { pairBlocksUsed_ = n.pairBlocksUsed; blocksEver_ = std::max(blocksEver_, n.blocksEver); ekinFromPair_ = n.ekinFromPair; lastStepEquil_ = n.lastStepEquil; }
    std::vector<GraphSlot> graphs_;
    BufState buf_state() const;
    GraphSlot* graph_for_state(int cycle);
    int graph_cycle() const;
    bool can_graph() const;
    void set_buf_state(const BufState& b);
    bool capturing_ = false;
    // next-step fusion (pair_tile.hip.h NextStep): on plain NVE steps of a lazy run on one GPU the pair kernel's epilogue also does the next step's
    // k_integrate1_bin<2>; the new positions go to a second set of coordinate arrays, swapped in after the launch
    double* altXyz_[2][3] = {{nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}};
    bool fuseNextOk_ = false;       // this engine may fuse (decided once)
    bool fuseNext_ = false;         // the pair kernel of the step being launched fuses
    bool fuseNextTstat_ = false;    // ... in a run with the radiative thermostat: the fused epilogue also applies it (k_pair_list<..., TSTAT>; decided once)
    bool pairClosedStep_ = false;   // the pair launch of the step in flight has closed the step (second half-kick + thermostat): no k_integrate2_post / k_boundary_radi
    bool preIntegrated_ = false;    // the step being launched was opened by the previous step's pair kernel: no k_integrate1_bin
    int stepsLeftInRun_ = 0;        // steps that follow the one being launched before the host looks / the cycle ends
    // a sort interval may run on from one aztot_step call into the next (one GPU, pair lists): the lists recorded at the last rebuild are those of the
    // arrays as they stand; set_state / aztot_forces end that
    SplitArgs split_{};             // few cells, wide stencils: several waves per cell in the staging kernel (pair_tile.hip.h)
    bool listsValid_ = false;
    // slab ranks: how many cells the last list build left without a list, copied to pinned memory behind the build; once the copy has landed and says
    // "none", the clean-up launches of the interval are skipped (a slab rank cannot widen its stencil, so the clean-up launch has nothing else to do there)
    int32_t* hUnlisted_ = nullptr;
    hipEvent_t evUnlisted_ = nullptr;
    int unlistedState_ = 0;         // 0 unknown (copy in flight or never made), 1 known to be zero, 2 known to be non-zero
    bool carryAgreed_ = false;      // slab ranks: every rank can carry the interval on (decided together at the end of the previous call)
    // Slab ranks cannot repair a skin violation on the spot (they hold hw ghost layers: no wider stencil to fall back on), and the host learns of one only
    // at its next look.  So every look that finds none leaves a snapshot of the dynamic state (per-atom arrays, DevStats, Counts, partial sums - what an
    // exact restart needs, device to device), and a look that finds one goes back to it and runs the steps since again with the cells rebuilt every
    // step, which is exact whatever the speeds (all ranks together: the verdict is all-reduced).  A violation costs time, never exactness.
    struct Snapshot
    {
        AtomArrays A{};
        void *stats = nullptr, *counts = nullptr;
        double* partials = nullptr;
        BufState buf{};
        long long hostStep = 0;
        bool valid = false;
    } snap_;
    long long stepsSinceSnap_ = 0;
    // One GPU, small systems (a step is a handful of microsecond kernels: the clean-up launch behind every k_pair_list is a quarter of a 40 000-atom step and
    // nearly always has nothing to do): the same snapshots let the plain steps run WITHOUT it.  A look that finds a violation or a cell that kept no list
    // goes back and runs the window again with the launch in place (exact as before), and the run stays on the safe side for a few looks.
    bool optimistic_ = false;
    int safeLooks_ = 1;             // looks still to be spent with the clean-up launch in place (an engine's first; 8, 16, 32 ... after a window had to be run again)
    int rollbacks_ = 0;
    bool unlistedAtLook_ = false;   // the last look found cells recorded without a list
    void choose_optimism();
    bool rollback_on() const { return lazyOn_ && (nranks_ > 1 || optimistic_); }
    void take_snapshot();
    void replay_from_snapshot();
    void prepare_next_call();
};

void check_hip(hipError_t e, const char* what);

}  // namespace aztot
