// Engine: device management + step driver (see engine.h).
#include "engine.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <tuple>

#include "exchange.h"
#include "bonded.hip.h"
#include "ewald.hip.h"
#include "kernels.hip.h"
#include "pair_tile.hip.h"
#include "pair_list.hip.h"
#include "slab.hip.h"

namespace aztot {

void check_hip(hipError_t e, const char* what)
{
    if (e != hipSuccess)
        throw std::runtime_error(std::string("HIP error in ") + what + ": " + hipGetErrorString(e));
}
#define HIP_CHECK(x) check_hip((x), #x)

namespace {
inline int div_up(int a, int b) { return (a + b - 1) / b; }
constexpr int kFuseKickMaxAtoms = 262144;
constexpr int kLazyCapMax = 128;         // longest sort interval (steps)
constexpr int kListCandMax = 1920;       // the LDS tile of k_pair_list holds candCap + 1 records of 32 B next to a 1 KiB table: below 64 KiB
constexpr int kListIterMax = 248;
constexpr long long kSnapshotKeepSteps = 32;     // a verified snapshot younger than this is kept at the end of a call (Engine::settle)
}  // namespace

template <typename F>
void Engine::timed(const char* name, F&& launch)
{
    if (!profile_ || capturing_) { launch(); return; }
    int idx;
    auto it = timerIndex_.find(name);
    if (it == timerIndex_.end())
    {
        idx = (int)timers_.size();
        timers_.push_back(KernelTimer{name, 0.0, 0});
        timerIndex_[name] = idx;
    }
    else idx = it->second;
    auto get = [&]() {
        if (!eventPool_.empty()) { hipEvent_t e = eventPool_.back(); eventPool_.pop_back(); return e; }
        hipEvent_t e; HIP_CHECK(hipEventCreate(&e)); return e;
    };
    hipEvent_t a = get(), b = get();
    HIP_CHECK(hipEventRecord(a, stream_));
    launch();
    HIP_CHECK(hipEventRecord(b, stream_));
    pending_.push_back({idx, a, b});
    if (pending_.size() > 8192) { HIP_CHECK(hipStreamSynchronize(stream_)); drain_events(); }
}

void Engine::drain_events()
{
    for (auto& p : pending_)
    {
        float ms = 0.f;
        HIP_CHECK(hipEventElapsedTime(&ms, p.a, p.b));
        timers_[p.idx].ms += ms;
        timers_[p.idx].calls += 1;
        eventPool_.push_back(p.a); eventPool_.push_back(p.b);
    }
    pending_.clear();
}

void Engine::sync()
{
    HIP_CHECK(hipStreamSynchronize(stream_));
    if (profile_) drain_events();
}

int Engine::kernel_times(std::vector<KernelTimer>& out) { settle(); sync(); out = timers_; return (int)out.size(); }
void Engine::reset_kernel_times() { settle(); sync(); for (auto& t : timers_) { t.ms = 0; t.calls = 0; } }

// ---------------------------------------------------------------------------------------------------
// cell grid: split_cells (cuCellList.cu:9-34, div_type 1: edge >= requested size).  No pair tables are
// built (the reference's are O(nCell^2), cuCellList.cu:516-531): the stencil is walked on the fly with
// half-width ceil(rMax / edge) per axis (cells may be smaller than the cut-off).
// ---------------------------------------------------------------------------------------------------
void Engine::choose_cells()
{
    const Model& m = model_;
    double size = opt_.cell_size > 0 ? opt_.cell_size : ((m.use_clist && m.desired_cell_size > 0) ? m.desired_cell_size : m.rMax);
    if (!(size > 0)) size = std::max(m.L[0], std::max(m.L[1], m.L[2]));
    // Verlet skin (options.skin): the lazy re-sort keeps the cells for as long as no atom has moved farther than skin / 2, which needs a stencil that
    // reaches rc + skin.  When the user asks for cells of about the cut-off (`cell_list 8.5` next to rc 8.5: the reference's split_cells then makes
    // them L / floor(L / 8.5), cuCellList.cu:9-34 - whatever that overhangs the cut-off used to be all the slack there was: 0.05 A on the 361.305 A box,
    // nothing at all on a box of 42 x 8.5 A) the cells are sized rc + skin instead; "edge >= the requested size" still holds.  Deliberately finer or
    // coarser grids are left alone (their stencil reach is what it is).
    skinTarget_ = 0.0;
    const bool lazyWanted = opt_.sort_every != 1 && m.rMax > 0 && m.E[0] == 0.0 && m.E[1] == 0.0 && m.E[2] == 0.0 && opt_.skin >= 0.0;
    if (lazyWanted)
    {
        const double skin = opt_.skin > 0.0 ? opt_.skin : std::min(0.5, std::max(0.15, 0.036 * m.rMax));
        skinTarget_ = skin;
        if (opt_.cell_size <= 0 && size >= 0.95 * m.rMax && size < m.rMax + skin)
        {
            bool ok = true;
            for (int k = 0; k < 3; k++)
            {   // (never at the price of the lazy schedule itself: it wants 5 cells per axis on one GPU, or of a slab split that the finer grid allows)
                const int nFine = (int)std::floor(m.L[k] / size), nSkin = (int)std::floor(m.L[k] / (m.rMax + skin));
                if (nSkin < 5 && nFine >= 5) ok = false;
                if (k == 0 && nranks_ > 1 && nSkin / nranks_ < 2) ok = false;
            }
            if (ok) size = m.rMax + skin;
        }
    }
    for (int k = 0; k < 3; k++)
    {
        int n = (int)std::floor(m.L[k] / size);
        if (n < 1) n = 1;
        if (n > 1024) n = 1024;
        if (k == 0 && nranks_ > 1 && n % nranks_ != 0)
        {   // slab decomposition: the step is as slow as the rank with the most layers (41 layers on 8 ranks: seven of 5 and one of 6 - 85 % efficiency at
            // best), so a few per cent longer cells along x are the better deal when they make the layers divide evenly
            const int even = n - n % nranks_;
            if (even >= 2 * nranks_ && (double)n / even <= 1.08) n = even;
        }
        P_.nc[k] = n;
        P_.csz[k] = m.L[k] / n;
        P_.icsz[k] = n / m.L[k];
        int hw = (m.rMax > 0) ? (int)std::ceil(m.rMax / P_.csz[k] - 1e-12) : 0;
        if (hw < 0) hw = 0;
        P_.hw[k] = hw;
        P_.nOff[k] = std::min(2 * hw + 1, n);
    }
    if (nranks_ > 1)
    {
        // slab decomposition along x: contiguous, balanced runs of cell layers
        const int n = P_.nc[0], hw = std::max(P_.hw[0], 1);
        P_.hw[0] = hw;
        const int lo = (int)((long long)n * rank_ / nranks_), hi = (int)((long long)n * (rank_ + 1) / nranks_);
        if (hi - lo < 2 * hw)
            throw std::runtime_error("slab decomposition: every rank needs at least 2*ceil(rc/cell) cell layers along x");
        if ((hi - lo) + 2 * hw > n)
            throw std::runtime_error("slab decomposition: ghost layers would overlap the rank's own layers (box too small for this rank count)");
        P_.cx0 = lo - hw;
        P_.ncxLocal = (hi - lo) + 2 * hw;
        P_.xlo = lo * P_.csz[0];
        P_.xhi = hi * P_.csz[0];
        P_.nOff[0] = 2 * hw + 1;
    }
    else
    {
        P_.cx0 = 0; P_.ncxLocal = P_.nc[0]; P_.xlo = 0.0; P_.xhi = m.L[0];
    }
    P_.nCellLocal = P_.ncxLocal * P_.nc[1] * P_.nc[2];
}

Engine::Engine(const Model& model, const aztot_options& opt, int rank, int nranks, Exchanger* xch)
    : model_(model), opt_(opt), rank_(rank), nranks_(nranks), xch_(xch)
{
    try { construct(); }
    catch (...) { release(); throw; }      // ~Engine does not run for a constructor that throws: free the stream and every allocation here
}

void Engine::construct()
{
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        throw std::runtime_error("no HIP device available: the azTotMD hot path has no CPU fallback");
    if (opt_.device < 0 || opt_.device >= ndev) throw std::runtime_error("HIP device ordinal out of range");
    HIP_CHECK(hipSetDevice(opt_.device));
    HIP_CHECK(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
    if (nranks_ > 1)
    {
        HIP_CHECK(hipStreamCreateWithFlags(&commStream_, hipStreamNonBlocking));
        HIP_CHECK(hipEventCreateWithFlags(&evIntegrated_, hipEventDisableTiming));
        HIP_CHECK(hipEventCreateWithFlags(&evHalo_, hipEventDisableTiming));
    }
    profile_ = opt_.profile != 0;
    if (const char* dbg = std::getenv("AZTOT_DEBUG")) debug_ = (unsigned)std::strtoul(dbg, nullptr, 0);
    if (opt_.energies_every_step) debug_ |= DBG_ENERGIES_EVERY_STEP;
    // (the radiative thermostat kicks atoms at random - a photon's momentum E / (m c) with the reference's c -, so the longest step of the next window is
    //  not bounded by the trend of the last ones: case study 2 ran into violations at 1.15)
    if (model_.tstat_type == AZTOT_TSTAT_RADI) lazyMargin_ = 1.3;
    if (const char* e = std::getenv("AZTOT_MARGIN")) lazyMargin_ = std::max(1.0, std::atof(e));      // (experiments: K steps of the longest step seen may use slack / margin)
    if (nranks_ > 1 && !xch_ && !opt_.loopback_ranks) throw std::runtime_error("slab decomposition needs an exchanger");

    const Model& m = model_;
    P_.nSpec = m.nSpec();
    P_.nAtGlobal = m.nAt;
    for (int k = 0; k < 3; k++) { P_.L[k] = m.L[k]; P_.invL[k] = 1.0 / m.L[k]; P_.half[k] = m.L[k] * 0.5; P_.E[k] = m.E[k]; }
    P_.dt = m.tSt;
    P_.r2Max = m.r2Max;
    P_.elec_type = m.elec_type; P_.alpha = m.alpha; P_.el_scale = m.el_scale; P_.el_scale2 = m.el_scale2; P_.daipi2 = m.daipi2;
    P_.rReal = m.rReal; P_.fcoul = units::Fcoul_scale; P_.sqrtpi = std::sqrt(units::pi);
    P_.tstat = m.tstat_type; P_.nEq = m.nEq; P_.freqEq = m.freqEq; P_.tKin = m.tKin; P_.revDegFree = m.revDegFree; P_.rkB = 1.0 / units::kB;
    if (m.tstat_type == AZTOT_TSTAT_NOSE) { P_.rQmass = 0.5 / m.tKin / m.tau / m.tau; P_.qMassTau2 = 2 * m.tKin; }
    P_.revLight = 3.33567e-5;   // cuTemp.cu:225 (SURVEY C-19: 100x the physical 1/c, kept for behavioural parity)
    P_.radFrac = 0.9;           // cuTemp.cu:639
    P_.radThr = 1e-4;           // cuTemp.cu:747
    P_.numPi = 3.14159;         // cuTemp.cu:228
    P_.seed = opt_.seed;
    P_.rank = rank_; P_.nranks = nranks_;
    P_.pad0 = debug_;
    P_.use_radii = 0;
    for (const auto& p : m.pairpots) if (p.type && p.use_radii) P_.use_radii = 1;
    P_.single_lj = (m.nSpec() == 1 && m.pairpots[0].type == AZTOT_VDW_LJ && m.elec_type == AZTOT_ELEC_NONE) ? 1 : 0;
    P_.ljDropR2 = 1e300;
    if (P_.single_lj)
    {   // fer_lj (vdw.cpp:16-26): f = p2 s^3 (2 s^3 - 1) / r^2 with s = p1 / r^2 (p1 = sigma^2, p2 = 24 eps).  For s >= 1: |f| <= 3 p2 s^7 / p1,
        // so |f| <= 1e5 whenever s^7 <= 1e5 p1 / (3 p2); for s < 1: |f| < p2 / p1.  The kernel evaluates the exact f^2 > 1e10 rule only when
        // some pair of the wave is inside TWICE that r^2 (a liquid never is).
        const PairPot& lj = m.pairpots[0];
        const double ratio = 3.0 * lj.p2 / (1e5 * lj.p1);
        if (ratio > 0.0 && ratio < 1.0 && lj.p2 / lj.p1 < 1e5) P_.ljDropR2 = 2.0 * lj.p1 * std::pow(ratio, 1.0 / 7.0);
    }
    {   // kernel specialisation 2/3: few species, ONE potential family among lnjs / buck / p746 / bmhs for every defined pair, no
        // radius-dependent potential, electrostatics none / direct / Fennell / Ewald inside the domain of the erfcx fit
        int family = 0;
        bool uniform = m.nSpec() <= 4 && !P_.single_lj;
        for (const auto& p : m.pairpots)
        {
            if (p.type == 0) continue;
            if (p.use_radii || p.type < AZTOT_VDW_LJ || p.type > AZTOT_VDW_BHM) uniform = false;
            if (family == 0) family = p.type; else if (family != p.type) family = 5;   // 5: mixed families, type looked up per species pair
        }
        if (family == 0) family = AZTOT_VDW_LJ;          // charges only: any family does, nothing is inside a VdW cut-off
        if ((m.elec_type == AZTOT_ELEC_FENNEL || m.elec_type == AZTOT_ELEC_EWALD) && m.alpha * m.rReal > 4.0) uniform = false;
        if (debug_ & 512) uniform = false;      // debug bit 512: take the generic kernel
        P_.pad1 = uniform ? 2 : 0;
        P_.vdwFamily = uniform ? family : 0;
        if (uniform && family == AZTOT_VDW_LJ)
        {   // family 6 = Lennard-Jones where EVERY species pair has a potential and none of their cut-offs lies inside the pair test's r2Max (the usual input:
            // one cut-off for everything, C3): the per-pair cut-off test - a table read, a compare and two selects per visit - always passes and is compiled out
            bool always = true;
            for (const auto& p : m.pairpots) if (p.type != AZTOT_VDW_LJ || p.r2cut < m.r2Max) always = false;
            if (always && !(debug_ & 1024)) P_.vdwFamily = 6;          // (debug bit 1024: keep the test)
        }
        if (uniform && family == AZTOT_VDW_LJ)
        {   // the same shortcut for the Lennard-Jones family with charges: |f| <= |f_LJ| + |f_Coulomb|, each held below 0.5e5 beyond its radius.  LJ part
            // as above per species pair; Coulomb part (direct, Fennell, real-space Ewald alike): |f| <= |kqq| (1/r^3 + (2 alpha/sqrt pi)/r^2 + S2/r)
            double r2 = 0.0;
            bool ok = true;
            for (const auto& p : m.pairpots)
            {
                if (p.type == 0) continue;
                const double ratio = 3.0 * p.p2 / (0.5e5 * p.p1);
                if (ratio > 0.0 && ratio < 1.0 && p.p2 / p.p1 < 0.5e5) r2 = std::max(r2, p.p1 * std::pow(ratio, 1.0 / 7.0)); else ok = false;
            }
            double qq = 0.0;
            for (int a = 0; a < m.nSpec(); a++)
                for (int b = 0; b < m.nSpec(); b++) qq = std::max(qq, std::fabs(m.species[a].charge * m.species[b].charge) * units::Fcoul_scale);
            if (m.elec_type != AZTOT_ELEC_NONE && qq > 0.0)
            {
                const double a2 = (m.elec_type == AZTOT_ELEC_DIRECT) ? 0.0 : 2.0 * std::fabs(m.alpha) / std::sqrt(units::pi);
                const double s2 = (m.elec_type == AZTOT_ELEC_FENNEL) ? std::fabs(m.el_scale2) : 0.0;
                double r = 0.01;
                while (r < 100.0 && qq * (1.0 / (r * r * r) + a2 / (r * r) + s2 / r) > 0.5e5) r *= 1.05;
                if (r >= 100.0) ok = false;
                r2 = std::max(r2, r * r);
            }
            if (ok) P_.ljDropR2 = 2.0 * r2;
        }
        // kernel specialisation 4: ONE species with the radius-dependent surk potential and no electrostatics (case study 2)
        if (m.nSpec() == 1 && m.pairpots[0].type == AZTOT_VDW_SURK && m.pairpots[0].use_radii && m.elec_type == AZTOT_ELEC_NONE && !(debug_ & 512))
            P_.pad1 = 4;
    }
    std::memset(&S_, 0, sizeof(S_));
    for (int i = 0; i < m.nSpec(); i++)
    {
        const Species& s = m.species[i];
        S_.mass[i] = s.mass; S_.charge[i] = s.charge; S_.rMhdt[i] = s.rMass_hdt;
        S_.radA[i] = s.radA; S_.radB[i] = s.radB; S_.mxEng[i] = s.mxEng;
        S_.charged[i] = s.charged; S_.frozen[i] = s.frozen;
    }
    choose_cells();
    allocate();
    upload_bonded();
    upload_ewald();
    {
        // who applies the second half-kick on plain NVE steps (nothing is added to the pair forces, nothing rescales velocities)
        const bool plainNve = !(P_.nEq > 0) && P_.tstat == AZTOT_TSTAT_NONE && !hasBonded_ && !hasEwald_ && !(debug_ & 128);
        const int variant = pair_variant();
        // small systems / slabs are bound by launch latency: the tile kernel's epilogue does it (one kernel less: C2 0.083 -> 0.063 ms).
        // On 1 M atoms that 13-lane read-modify-write of the velocities makes L2 write lines back several times (rocprofv3: 221 MB
        // instead of 63 + 76 MB per step) for no gain, so large systems fold it into the next step's streaming k_integrate1_bin.
        fuseEpilogue_ = plainNve && variant >= 2 && capacity_ <= kFuseKickMaxAtoms && !(debug_ & 256);   // debug bit 256: large-system path
        // (bonded and reciprocal-space forces are added behind the pair kernel and are complete before the next step's integrate kernel just the same: runs with
        //  bonds / angles / the Ewald sum defer the kick too - they cannot take the pair kernel's epilogue, which would kick with the pair forces alone.
        //  M4: one 17 us launch per step less)
        const bool kickCanWait = !(P_.nEq > 0) && P_.tstat == AZTOT_TSTAT_NONE && !(debug_ & 128);
        lazyKick_ = kickCanWait && !fuseEpilogue_;
        P_.pad2 = 0;
        // next-step fusion (NextStep, pair_tile.hip.h): plain NVE (nothing happens between the forces and the next half-kick; on slab ranks the coordinate
        // exchange of the next step simply follows the pair kernel that produced the coordinates), a lazy run that walks
        // pair lists; debug bit 131072 switches it off.  Up to ~500 000 atoms per GPU, where a step is bound by launch latency (measured: 40 000 atoms 0.0230 ->
        // 0.0199 ms/step; emulated slab ranks of 143 000 / 250 000 / 333 000 atoms -9 % / -5 % / -4 %): on 1 M atoms the 13-lane stores of the epilogue cost
        // the pair kernel exactly what the streaming k_integrate1_bin<2> costs on its own (111 + 31 -> 140 us); debug bit 262144 forces it on there
        // ... and runs with the radiative thermostat (case study 1: 40 000 atoms, two launches per step - the pair kernel and the boundary kernel that closes
        // the step with the thermostat and opens the next): the thermostat acts on one atom at a time, so the lane that closes the atom's step applies it too
        // (k_pair_list<..., TSTAT>) and a step is ONE launch.  Not where the pair kernel reads radii (the thermostat rewrites them while other waves still
        // gather: case study 2's surk potential), not with bonded terms or the Ewald sum (their forces arrive after the pair kernel).
        const bool radiFuse = P_.tstat == AZTOT_TSTAT_RADI && !hasBonded_ && !hasEwald_ && !(debug_ & 128) && !P_.use_radii && P_.pad1 != 4 && (P_.single_lj || P_.pad1 == 2) &&
                              nranks_ == 1 && capacity_ <= 2 * kFuseKickMaxAtoms;
        fuseNextTstat_ = radiFuse && listsOn_ && variant == 2 && !(debug_ & 131072);
        if (((plainNve && (capacity_ <= 2 * kFuseKickMaxAtoms || (debug_ & 262144))) || radiFuse) && listsOn_ && variant == 2 && !(debug_ & 131072))
        {
            fuseNextOk_ = true;
            const size_t nd = sizeof(double) * (size_t)capacity_;
            for (int b = 0; b < 2; b++)
                for (int d = 0; d < 3; d++)
                {
                    void* p = nullptr;
                    HIP_CHECK(hipMalloc(&p, std::max<size_t>(nd, 16)));
                    allocs_.push_back(p);
                    altXyz_[b][d] = (double*)p;
                    HIP_CHECK(hipMemsetAsync(p, 0, nd, stream_));
                }
        }
    }
    // displacement bound instead of the per-atom check on plain steps: wherever k_integrate1_bin<2> opens every plain step (engines that fuse the next
    // step into the pair kernel keep the per-atom check there); debug bit 524288 switches it off
    P_.pad2 = (lazyOn_ && !fuseNextOk_ && !(debug_ & 524288)) ? 1 : 0;
    if (nranks_ > 1 && !xch_)
    {   // options.reserved[1]: loopback measurement mode (see LoopbackExchanger)
        ownedXch_.reset(new LoopbackExchanger((P_.ncxLocal - 2 * P_.hw[0]) * P_.csz[0], P_.L[0], lay_.mig_offset(), lay_.halo_offset(),
                                              sizeof(MigRec), sizeof(HaloRec)));
        xch_ = ownedXch_.get();
    }
    upload_initial();
}

Engine::~Engine() { release(); }

void Engine::destroy_graphs()
{
    for (GraphSlot& g : graphs_)
    {
        if (g.exec) (void)hipGraphExecDestroy(g.exec);
        if (g.graph) (void)hipGraphDestroy(g.graph);
    }
    graphs_.clear();
}

Engine::BufState Engine::buf_state() const
{
    BufState b;
    b.cur = cur_;
    for (int k = 0; k < 2; k++)
    {
        b.xyz[k][0] = buf_[k].x; b.xyz[k][1] = buf_[k].y; b.xyz[k][2] = buf_[k].z;
        for (int d = 0; d < 3; d++) b.alt[k][d] = altXyz_[k][d];
    }
    return b;
}

void Engine::set_buf_state(const BufState& b)
{
    cur_ = b.cur;
    for (int k = 0; k < 2; k++)
    {
        buf_[k].x = b.xyz[k][0]; buf_[k].y = b.xyz[k][1]; buf_[k].z = b.xyz[k][2];
        for (int d = 0; d < 3; d++) altXyz_[k][d] = b.alt[k][d];
    }
}

void Engine::release()
{
    if (stream_) (void)hipStreamSynchronize(stream_);
    destroy_graphs();
    for (auto& p : pending_) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
    pending_.clear();
    for (auto e : eventPool_) (void)hipEventDestroy(e);
    eventPool_.clear();
    for (void* p : allocs_) (void)hipFree(p);
    allocs_.clear();
    free_lists();
    if (evIntegrated_) (void)hipEventDestroy(evIntegrated_);
    if (evHalo_) (void)hipEventDestroy(evHalo_);
    evIntegrated_ = evHalo_ = nullptr;
    if (evUnlisted_) (void)hipEventDestroy(evUnlisted_);
    evUnlisted_ = nullptr;
    if (hUnlisted_) (void)hipHostFree(hUnlisted_);
    hUnlisted_ = nullptr;
    if (evHaloInfo_) (void)hipEventDestroy(evHaloInfo_);
    evHaloInfo_ = nullptr;
    if (hHalo_) (void)hipHostFree(hHalo_);
    hHalo_ = nullptr;
    if (hLook_) (void)hipHostFree(hLook_);
    hLook_ = nullptr;
    if (commStream_) { (void)hipStreamSynchronize(commStream_); (void)hipStreamDestroy(commStream_); }
    commStream_ = nullptr;
    if (stream_) (void)hipStreamDestroy(stream_);
    stream_ = nullptr;
}

// a rejected launch configuration (too much LDS, bad grid) is reported by hipGetLastError only: hipStreamSynchronize stays silent
void Engine::check_launch(const char* where)
{
    if (capturing_) return;
    check_hip(hipGetLastError(), where);
}

void Engine::allocate()
{
    const int N = model_.nAt;
    if (nranks_ > 1)
    {
        // Capacities from the ACTUAL initial population of the x-layers (every rank holds the whole model, so all ranks arrive at
        // the same message size): the fixed-size message is what travels every step, so padding costs link time.
        //   halo      the fullest run of hw consecutive layers + 25 %
        //   migrants  3 % of the fullest layer (an atom moves << one cell per step: v dt / edge ~ 1e-3) + 1024
        //   arrays    this rank's window (owned + ghost layers) + one ghost run + 25 %
        const int ncx = P_.nc[0], hw = P_.hw[0];
        std::vector<long long> hist(ncx, 0);
        for (int i = 0; i < N; i++)
        {
            int gx = (int)std::floor(model_.x[i] * P_.icsz[0]);
            gx %= ncx; if (gx < 0) gx += ncx;
            hist[gx]++;
        }
        long long maxLayer = 0, maxRun = 0, window = 0;
        for (int g = 0; g < ncx; g++)
        {
            maxLayer = std::max(maxLayer, hist[g]);
            long long run = 0;
            for (int k = 0; k < hw; k++) run += hist[(g + k) % ncx];
            maxRun = std::max(maxRun, run);
        }
        for (int l = 0; l < P_.ncxLocal; l++) window += hist[(((P_.cx0 + l) % ncx) + ncx) % ncx];
        // arrays: the received atoms are appended behind the owned range while last step's left ghosts still sit in front of it, so
        // the footprint is window + one more ghost run - NOT bounded by N on a short ring (2 ranks x 3-4 layers), where the window
        // alone already covers most of the box
        capacity_ = (int)(std::ceil((window + maxRun) * 1.25) + 4096);
        lay_.haloCap = (int)std::min<double>((double)N, std::ceil(maxRun * 1.25) + 1024);
        lay_.migCap = (int)std::min<double>((double)N, std::ceil(maxLayer * 0.03) + 1024);
    }
    else capacity_ = N;
    nCellAlloc_ = P_.nCellLocal;
    // + 24: a split launch rounds each of its three runs up to 8 workgroups; the clean-up launches behind k_pair_list book into their own slots
    {   // few cells (less than half a residency of waves) and a stencil with columns to share out: several waves per cell in the staging kernel
        const int cells = pair_tile_cells(P_);
        int n = 1;
        if (pair_tile_supported(P_) && !(debug_ & 67108864))
            while (n < 8 && cells * n * 2 <= 4096 && n * 2 <= P_.nOff[0] * P_.nOff[1]) n *= 2;
        if (n == 1 && pair_tile_supported(P_) && !(debug_ & 67108864))
        {   // more cells than that, but a stencil that needs several tiles per cell (dense systems, small cells: expected candidates = density x volume
            // within the cut-off of a cell): two waves per cell, each with half the columns and fewer tile flushes (measured: S40 -8 %, M4 -9 %; four: worse)
            const double r = model_.rMax, a = P_.csz[0], b = P_.csz[1], c = P_.csz[2];
            const double vol = a * b * c + 2.0 * r * (a * b + b * c + c * a) + 3.14159265358979 * r * r * (a + b + c) + 4.18879020478639 * r * r * r;
            const double density = (double)model_.nAt / (model_.L[0] * model_.L[1] * model_.L[2]);
            if (density * vol > 1.15 * kTileCap && P_.nOff[0] * P_.nOff[1] >= 2) n = 2;
        }
        if (opt_.waves_per_cell == 1 || opt_.waves_per_cell == 2 || opt_.waves_per_cell == 4 || opt_.waves_per_cell == 8) n = opt_.waves_per_cell;      // measurements: forced
        split_.n = n;
        split_.capacity = capacity_;
    }
    pairBlocks_ = std::max(div_up(capacity_, kBlock), pair_tile_grid(P_) * std::max(split_.n, kListMaxWaves) + 24 + 3 * pair_cleanup_grid(pair_tile_cells(P_)));
    maxBlocks_ = std::max(div_up(capacity_, kBlock), pairBlocks_) + 1;
    auto alloc = [&](size_t bytes) { void* p = nullptr; HIP_CHECK(hipMalloc(&p, std::max<size_t>(bytes, 16))); allocs_.push_back(p); return p; };
    const size_t nd = sizeof(double) * (size_t)capacity_, ni = sizeof(int32_t) * (size_t)capacity_;
    for (int b = 0; b < 2; b++)
    {
        AtomArrays& A = buf_[b];
        double** d[] = {&A.x, &A.y, &A.z, &A.vx, &A.vy, &A.vz, &A.fx, &A.fy, &A.fz, &A.U, &A.rad};
        for (auto pp : d) { *pp = (double*)alloc(nd); HIP_CHECK(hipMemsetAsync(*pp, 0, nd, stream_)); }
        A.type = (int32_t*)alloc(ni); A.id = (int32_t*)alloc(ni);
        HIP_CHECK(hipMemsetAsync(A.type, 0, ni, stream_)); HIP_CHECK(hipMemsetAsync(A.id, 0, ni, stream_));
    }
    dCellOf_ = (int32_t*)alloc(ni); dSlotOf_ = (int32_t*)alloc(ni);
    dTmpId_ = (int32_t*)alloc(ni); dTmpSrc_ = (int32_t*)alloc(ni); dTmpCell_ = (int32_t*)alloc(ni); dCellOfSorted_ = (int32_t*)alloc(ni);
    dCellCount_ = (int32_t*)alloc(sizeof(int32_t) * (size_t)(nCellAlloc_ + 1));
    dCellStart_ = (int32_t*)alloc(sizeof(int32_t) * (size_t)(nCellAlloc_ + 1));
    HIP_CHECK(hipMemsetAsync(dCellCount_, 0, sizeof(int32_t) * (size_t)(nCellAlloc_ + 1), stream_));
    HIP_CHECK(hipMemsetAsync(dCellStart_, 0, sizeof(int32_t) * (size_t)(nCellAlloc_ + 1), stream_));
    dPartials_ = (double*)alloc(sizeof(double) * (size_t)PS_COUNT * maxBlocks_);
    HIP_CHECK(hipMemsetAsync(dPartials_, 0, sizeof(double) * (size_t)PS_COUNT * maxBlocks_, stream_));
    dStats_ = (DevStats*)alloc(sizeof(DevStats));
    {
        DevStats zero;
        std::memset(&zero, 0, sizeof(zero));
        zero.vscale = 1.0;          // "no scaling"; only k_scale_decision ever changes it
        zero.vscaleBegin = 1.0;
        HIP_CHECK(hipMemcpy(dStats_, &zero, sizeof(DevStats), hipMemcpyHostToDevice));
    }
    if (split_.n > 1)
    {
        const size_t nb = sizeof(double) * (size_t)split_.n * capacity_;
        split_.fx = (double*)alloc(nb); split_.fy = (double*)alloc(nb); split_.fz = (double*)alloc(nb);
        split_.arrived = (int32_t*)alloc(sizeof(int32_t) * (size_t)P_.nCellLocal);
        HIP_CHECK(hipMemsetAsync(split_.arrived, 0, sizeof(int32_t) * (size_t)P_.nCellLocal, stream_));
    }
    dCounts_ = (Counts*)alloc(sizeof(Counts));
    dChunkTot_ = (int32_t*)alloc(sizeof(int32_t) * (size_t)(div_up(nCellAlloc_, kScanChunk) + 1));
    dEkGlobal_ = (double*)alloc(sizeof(double) * 2);
    dStage_ = (double*)alloc(sizeof(double) * PS_COUNT * kCollectParts);
    HIP_CHECK(hipMemsetAsync(dStage_, 0, sizeof(double) * PS_COUNT * kCollectParts, stream_));
    if (nranks_ > 1)
    {
        for (int k = 0; k < 4; k++) { dMsg_[k] = (char*)alloc(lay_.bytes()); HIP_CHECK(hipMemsetAsync(dMsg_[k], 0, lay_.bytes(), stream_)); }
        dHaloInfo_ = (int32_t*)alloc(sizeof(int32_t) * 16);
        HIP_CHECK(hipMemsetAsync(dHaloInfo_, 0, sizeof(int32_t) * 16, stream_));
    }
    const int ns = model_.nSpec();
    std::vector<DevPot> pots((size_t)ns * ns);
    for (int a = 0; a < ns; a++)
        for (int b = 0; b < ns; b++)
        {
            const PairPot& p = model_.pot(a, b);
            DevPot& d = pots[(size_t)a * ns + b];
            d.type = p.type; d.use_radii = p.use_radii; d.p0 = p.p0; d.p1 = p.p1; d.p2 = p.p2; d.p3 = p.p3; d.p4 = p.p4; d.r2cut = p.r2cut;
            // Lennard-Jones uses p0..p2 only (vdw.cpp:283-288: 4 eps, sigma^2, 24 eps); the two free slots carry the force law in powers of 1/r^2 for the
            // one-species kernel (pair_body, MODE 1): f = u^4 (A2 u^3 - A1) with u = 1/r^2, A1 = 24 eps sigma^6, A2 = 48 eps sigma^12
            if (p.type == AZTOT_VDW_LJ) { const double s6 = p.p1 * p.p1 * p.p1; d.p3 = p.p2 * s6; d.p4 = 2.0 * p.p2 * s6 * s6; }
        }
    {   // lazy re-sort: no external field (its energy is booked from wrapped coordinates), a stencil that can be widened by one cell on a violation
        // (one GPU), and a slack worth having
        const Model& m = model_;
        double slack = 1e300;
        bool widenOk = true;
        for (int k = 0; k < 3; k++)
        {
            slack = std::min(slack, 0.49 * (P_.hw[k] * P_.csz[k] - m.rMax));
            if (P_.nc[k] < 2 * (P_.hw[k] + 1) + 1) widenOk = false;
        }
        const int sortEvery = opt_.sort_every;                    // 0: adaptive, 1: every step (the reference's schedule), n: at most every n-th step
        // (a slab rank cannot widen its stencil - it holds hw ghost layers -: there a violation is repaired by going back to the last look's snapshot and
        //  running the window again with the cells rebuilt every step, Engine::replay_from_snapshot; the interval keeps a factor 2 in hand)
        lazyOn_ = sortEvery != 1 && (widenOk || nranks_ > 1) && m.rMax > 0 && slack > 1e-3 && m.E[0] == 0.0 && m.E[1] == 0.0 &&
                  m.E[2] == 0.0 && pair_tile_supported(P_);
        lazyCap_ = sortEvery > 1 ? std::min(sortEvery, kLazyCapMax) : kLazyCapMax;
        if (lazyOn_ && (debug_ & 8192)) lazyK_ = lazyCap_;
        // The slack an atom may use is half the skin (both atoms of a pair move).  With a skin (options.skin >= 0) the cells were sized for it and the box
        // may give a little more for free - taken up to a quarter above the target, beyond that it would only lengthen the lists; without one
        // (options.skin < 0) it is capped at 1.2 % of the cut-off as in round 2.
        slack = std::min(slack, skinTarget_ > 0.0 ? 0.5 * 1.25 * skinTarget_ : 0.012 * m.rMax);
        lazySlack_ = lazyOn_ ? slack : 0.0;
        P_.lazySlack2 = lazySlack_ * lazySlack_;
        const double rp = m.rMax + 2.0 * lazySlack_;
        P_.pruneR2 = rp * rp * (1.0 + 1e-12);
        if (lazyOn_)
        {
            ref_.x = (double*)alloc(nd); ref_.y = (double*)alloc(nd); ref_.z = (double*)alloc(nd);
            if (capacity_ < (1 << 26) && !(debug_ & 32768))          // atom index + 6 bits of image code in 32 bits; debug 32768: no lists
            {
                if (P_.ncxLocal > 1023 || P_.nc[1] > 1023 || P_.nc[2] > 1023) throw std::runtime_error("more than 1023 cells along an axis");
                int devLds = 0;
                HIP_CHECK(hipDeviceGetAttribute(&devLds, hipDeviceAttributeMaxSharedMemoryPerBlock, opt_.device));
                listLdsMax_ = (size_t)std::max(devLds, 64 * 1024);
                // capacities from the density: candidates = atoms within the list radius of a cell's box, iterations = pairs of a cell / 64 lanes, each
                // with room for a liquid's fluctuations (+ 30 % / + 50 %); cells that need more keep no list, and the engine grows the lists when
                // that happens to more than a few (adapt_sort_interval)
                const double a = P_.csz[0], b = P_.csz[1], c = P_.csz[2], r = rp;
                const double vol = a * b * c + 2.0 * r * (a * b + b * c + c * a) + 3.14159265358979 * r * r * (a + b + c) + 4.18879020478639 * r * r * r;
                const double density = (double)m.nAt / (m.L[0] * m.L[1] * m.L[2]);
                const double perCell = density * a * b * c, partners = density * 4.18879020478639 * r * r * r;
                // (a cell's iterations are its atoms' partners divided by the slices each atom gets, NS = 64 / atoms: sized for a cell 60 % fuller than the mean)
                const int nBig = std::min(kWave, (int)(1.6 * perCell) + 6);
                int cand = (int)(density * vol * 1.5) + 64;
                int iters = (int)(partners * 1.25 / list_slices(nBig)) + 4;
                // waves per cell in k_pair_list (they share the cell's tile and split its atoms): one where cells are many and their tiles small; more where the
                // tile of a cell is large (LDS per wave bounds the occupancy: dense systems, wide stencils) - measured: M4 (28 atoms and 720 candidates per
                // cell) 788 -> 514 us with two; on the 1 M-atom liquid (14 atoms, 320 candidates) two waves cost 102 -> 158 us: every wave pays the
                // fixed part of a cell again
                {
                    const int cells = pair_tile_cells(P_);
                    const double tileBytes = density * vol * (P_.single_lj ? 24.0 : (pair_list_tab_mode(P_) ? 25.0 : 32.0));
                    int w = 1;
                    (void)cells;             // (few cells alone do not pay for more waves: C2 13.9 -> 15.2 us, an emulated rank of 8 26.8 -> 29.2 us with two)
                    if (tileBytes > 13.0 * 1024) w = 2;
                    if (tileBytes > 26.0 * 1024) w = 4;
                    if (opt_.waves_per_cell == 1 || opt_.waves_per_cell == 2 || opt_.waves_per_cell == 4) w = opt_.waves_per_cell;
                    listWaves_ = w;
                    // (a wave of a W-wave cell serves ceil(n / W) atoms: sized for a cell 60 % fuller than the mean, as above)
                    iters = (int)(partners * 1.25 / list_slices(list_atoms_per_wave(std::min(kWave * w, (int)(1.6 * perCell) + 6), w))) + 4;
                }
                cand = std::max(kListMinCand * listWaves_, (cand + 63) & ~63);
                iters = std::max(2 * kListMinIter, (iters + 7) & ~7);
                if (const char* e = std::getenv("AZTOT_CAND_CAP")) cand = std::atoi(e);        // (experiments / tests: start from other capacities)
                if (const char* e = std::getenv("AZTOT_ITER_CAP")) iters = std::atoi(e);
                try { allocate_lists(cand, iters); }
                catch (const std::exception&)
                {   // (no room for the lists - hundreds of millions of cells: the run goes on without them, the steps between two rebuilds stage every cell)
                    (void)hipGetLastError();
                    free_lists();
                }
            }
        }
    }
    dPots_ = (DevPot*)alloc(sizeof(DevPot) * pots.size());
    HIP_CHECK(hipMemcpyAsync(dPots_, pots.data(), sizeof(DevPot) * pots.size(), hipMemcpyHostToDevice, stream_));
    HIP_CHECK(hipStreamSynchronize(stream_));
}

// Lists of the lazy re-sort (pair_list.hip.h): [nCell][candCap] candidates, [nCell][iterCap x 64] pair entries, headers.  Capacities are clipped to
// what the LDS of k_pair_list / k_build_lists and the entry format allow.
void Engine::allocate_lists(int candCap, int iterCap)
{
    free_lists();
    // (k_pair_list asks for five groups of candidate entries per wave and two chunks of 8 iterations before it knows the cell: the arrays are never smaller)
    candCap = std::max(kListMinCand * listWaves_, std::min((candCap + 63) & ~63, kListCandMax));
    iterCap = std::max(kListMinIter, std::min((iterCap + 7) & ~7, kListIterMax));
    PairLists probe;
    probe.candCap = probe.candLds = candCap; probe.iterCap = probe.iterLds = iterCap; probe.recBytes = pair_list_rec_bytes(P_); probe.waves = listWaves_;
    while (candCap > kListMinCand && (pair_list_lds_bytes(P_, probe) > listLdsMax_ || build_lists_lds_bytes(probe) > listLdsMax_))
    {
        candCap -= 64;
        probe.candCap = probe.candLds = candCap;
    }
    while (iterCap > 2 * kListMinIter && build_lists_lds_bytes(probe) > listLdsMax_) { iterCap -= 8; probe.iterCap = probe.iterLds = iterCap; }
    const size_t nc = (size_t)P_.nCellLocal;
    auto alloc = [&](size_t bytes) { void* p = nullptr; HIP_CHECK(hipMalloc(&p, std::max<size_t>(bytes, 16))); return p; };
    dCandList_ = (uint32_t*)alloc(sizeof(uint32_t) * nc * candCap);
    dListMeta_ = (int32_t*)alloc(sizeof(int32_t) * 4 * nc);
    dPairList_ = (uint16_t*)alloc(sizeof(uint16_t) * nc * (size_t)listWaves_ * (size_t)iterCap * kWave);
    dNoList_ = (int32_t*)alloc(sizeof(int32_t) * 16);
    dRel_ = (float4*)alloc(sizeof(float4) * ((size_t)capacity_ + kWave));
    HIP_CHECK(hipMemsetAsync(dRel_, 0, sizeof(float4) * ((size_t)capacity_ + kWave), stream_));
    candCap_ = candCap; iterCap_ = iterCap;
    candLds_ = candCap; iterLds_ = iterCap;          // (tightened once the builder has reported what the cells really hold: adapt_sort_interval)
    {   // header: -1 = no list ; second word: the cell's coordinates in the local grid (the kernels decode them with shifts)
        std::vector<int32_t> mx(4 * nc, 0);
        const int ncy = P_.nc[1], ncz = P_.nc[2];
        for (size_t c = 0; c < nc; c++)
        {
            const int cz = (int)(c % ncz), cy = (int)((c / ncz) % ncy), lx = (int)(c / ((size_t)ncy * ncz));
            mx[4 * c] = -1; mx[4 * c + 1] = lx | (cy << 10) | (cz << 20);
        }
        HIP_CHECK(hipMemcpyAsync(dListMeta_, mx.data(), sizeof(int32_t) * 4 * nc, hipMemcpyHostToDevice, stream_));
        HIP_CHECK(hipStreamSynchronize(stream_));
    }
    HIP_CHECK(hipMemsetAsync(dCandList_, 0, sizeof(uint32_t) * nc * candCap, stream_));      // every entry is an atom index at all times
    HIP_CHECK(hipMemsetAsync(dPairList_, 0, sizeof(uint16_t) * nc * (size_t)listWaves_ * (size_t)iterCap * kWave, stream_));   // every entry is a tile offset at all times (k_pair_list reads ahead)
    HIP_CHECK(hipMemsetAsync(dNoList_, 0, sizeof(int32_t) * 16, stream_));
    listsOn_ = true;
    listsValid_ = false;
    if (std::getenv("AZTOT_VERBOSE"))
        std::fprintf(stderr, "aztot: pair lists for %zu cells: %d candidates, %d iterations per cell (%.1f MB), LDS %zu B (list kernel) / %zu B (builder)\n", nc, candCap, iterCap,
                     (double)(nc * ((size_t)candCap * 4 + (size_t)iterCap * 128)) / 1e6, pair_list_lds_bytes(P_, pair_lists()), build_lists_lds_bytes(pair_lists()));
}

// Larger lists in the middle of a run (adapt_sort_interval).  allocate_lists frees the old ones first; if the new ones do not fit the run goes on
// without lists - the steps between two rebuilds stage every cell, as after a failed allocation at construction - instead of failing the call half-way
// through a look (on slab ranks the other ranks would be left waiting in the look's all-reduce: ADVICE round 3).
void Engine::regrow_lists(int candCap, int iterCap)
{
    try { allocate_lists(candCap, iterCap); }
    catch (const std::exception& e)
    {
        (void)hipGetLastError();
        free_lists();
        if (std::getenv("AZTOT_VERBOSE")) std::fprintf(stderr, "aztot: rank %d: no room for larger pair lists (%s): the run goes on without lists\n", rank_, e.what());
    }
}

void Engine::free_lists()
{
    if (stream_) (void)hipStreamSynchronize(stream_);
    if (dCandList_) (void)hipFree(dCandList_);
    if (dListMeta_) (void)hipFree(dListMeta_);
    if (dPairList_) (void)hipFree(dPairList_);
    if (dNoList_) (void)hipFree(dNoList_);
    if (dRel_) (void)hipFree(dRel_);
    dRel_ = nullptr;
    dCandList_ = nullptr; dListMeta_ = nullptr; dPairList_ = nullptr; dNoList_ = nullptr;
    listsOn_ = false; listsValid_ = false;
}

PairLists Engine::pair_lists() const
{
    PairLists pl;
    if (listsOn_)
    {
        pl.cand = dCandList_; pl.meta = dListMeta_; pl.pairs = dPairList_; pl.noList = dNoList_;
        pl.candCap = candCap_; pl.iterCap = iterCap_; pl.candLds = candLds_; pl.iterLds = iterLds_; pl.recBytes = pair_list_rec_bytes(P_); pl.entryScale = pair_list_entry_scale(P_); pl.waves = listWaves_;
        pl.rel = dRel_;
    }
    return pl;
}

// Static per-atom tables of the bonded terms, keyed by atom id and replicated on every rank (like the reference's
// host-side lists, cuBonds.cu / cuAngles.cu); entries of one atom keep list order, so the summation order is fixed.
void Engine::upload_bonded()
{
    const Model& m = model_;
    const size_t nb = m.bondA.size(), na = m.angC.size();
    hasBonded_ = (nb + na) > 0;
    if (!hasBonded_) return;
    const int N = m.nAt;
    auto up = [&](const void* src, size_t bytes) {
        void* p = nullptr; HIP_CHECK(hipMalloc(&p, std::max<size_t>(bytes, 16))); allocs_.push_back(p);
        if (bytes) HIP_CHECK(hipMemcpy(p, src, bytes, hipMemcpyHostToDevice));
        return p;
    };
    std::vector<int32_t> bs(N + 1, 0), as(N + 1, 0);
    for (size_t k = 0; k < nb; k++) { bs[m.bondA[k] + 1]++; bs[m.bondB[k] + 1]++; }
    for (size_t k = 0; k < na; k++) { as[m.angC[k] + 1]++; as[m.angL1[k] + 1]++; as[m.angL2[k] + 1]++; }
    for (int i = 0; i < N; i++) { bs[i + 1] += bs[i]; as[i + 1] += as[i]; }
    std::vector<BondEntry> be(2 * nb);
    std::vector<AngleEntry> ae(3 * na);
    {
        std::vector<int32_t> fill(bs.begin(), bs.end() - 1);
        for (size_t k = 0; k < nb; k++)
        {
            be[fill[m.bondA[k]]++] = BondEntry{m.bondB[k], m.bondT[k] | (1 << 30)};
            be[fill[m.bondB[k]]++] = BondEntry{m.bondA[k], m.bondT[k]};
        }
        fill.assign(as.begin(), as.end() - 1);
        for (size_t k = 0; k < na; k++)
        {
            const int32_t c = m.angC[k], l1 = m.angL1[k], l2 = m.angL2[k], t = m.angT[k] << 2;
            ae[fill[c]++] = AngleEntry{t | 0, c, l1, l2};
            ae[fill[l1]++] = AngleEntry{t | 1, c, l1, l2};
            ae[fill[l2]++] = AngleEntry{t | 2, c, l1, l2};
        }
    }
    std::vector<DevBondType> bt(m.bondTypes.size() + 1);
    std::memset(bt.data(), 0, sizeof(DevBondType) * bt.size());
    for (size_t k = 0; k < m.bondTypes.size(); k++)
    {
        const BondType& b = m.bondTypes[k];
        bt[k + 1] = DevBondType{b.type, 0, b.p[0], b.p[1], b.p[2], b.p[3], b.p[4]};
    }
    std::vector<DevAngleType> at(m.angleTypes.size() + 1, DevAngleType{0.0, 0.0});
    for (size_t k = 0; k < m.angleTypes.size(); k++) at[k + 1] = DevAngleType{m.angleTypes[k].k, m.angleTypes[k].cos0};
    bonded_.bondStart = (const int32_t*)up(bs.data(), sizeof(int32_t) * bs.size());
    bonded_.bondEnt = (const BondEntry*)up(be.data(), sizeof(BondEntry) * be.size());
    bonded_.angStart = (const int32_t*)up(as.data(), sizeof(int32_t) * as.size());
    bonded_.angEnt = (const AngleEntry*)up(ae.data(), sizeof(AngleEntry) * ae.size());
    bonded_.btypes = (const DevBondType*)up(bt.data(), sizeof(DevBondType) * bt.size());
    bonded_.atypes = (const DevAngleType*)up(at.data(), sizeof(DevAngleType) * at.size());
    std::vector<int32_t> none(N, -1);
    bonded_.idxOfId = (int32_t*)up(none.data(), sizeof(int32_t) * (size_t)N);
}

// k-vector table of the Ewald sum + work buffers.  The table is the host model's (finish_model, same loop as ewald_rec).
void Engine::upload_ewald()
{
    const Model& m = model_;
    hasEwald_ = m.elec_type == AZTOT_ELEC_EWALD && !m.kvecs.empty();
    if (!hasEwald_) return;
    std::vector<EwaldK> kv(m.kvecs.size());
    for (size_t k = 0; k < kv.size(); k++)
    {
        const KVec& a = m.kvecs[k];
        int flags = 0;
        if (k == 0 || m.kvecs[k - 1].l != a.l) flags |= EWK_NEW_L | EWK_NEW_LM;
        else if (m.kvecs[k - 1].m != a.m) flags |= EWK_NEW_LM;
        kv[k] = EwaldK{a.l, a.m, a.n, flags, a.rkx, a.rky, a.rkz, a.akk};
    }
    auto alloc = [&](size_t bytes) { void* p = nullptr; HIP_CHECK(hipMalloc(&p, std::max<size_t>(bytes, 16))); allocs_.push_back(p); return p; };
    EwaldK* dkv = (EwaldK*)alloc(sizeof(EwaldK) * kv.size());
    HIP_CHECK(hipMemcpy(dkv, kv.data(), sizeof(EwaldK) * kv.size(), hipMemcpyHostToDevice));
    ew_.kv = dkv;
    // work items of k_ewald_sfac: (l, m, |n|) with the indices of its +n / -n members
    std::map<std::tuple<int, int, int>, int> slot;
    std::vector<EwaldW> work;
    for (size_t k = 0; k < kv.size(); k++)
    {
        const auto key = std::make_tuple(kv[k].l, kv[k].m, std::abs(kv[k].n));
        auto it = slot.find(key);
        if (it == slot.end()) { slot[key] = (int)work.size(); work.push_back(EwaldW{kv[k].l, kv[k].m, std::abs(kv[k].n), -1, -1}); it = slot.find(key); }
        if (kv[k].n >= 0) work[it->second].kPlus = (int)k; else work[it->second].kMinus = (int)k;
    }
    EwaldW* dw = (EwaldW*)alloc(sizeof(EwaldW) * work.size());
    HIP_CHECK(hipMemcpy(dw, work.data(), sizeof(EwaldW) * work.size(), hipMemcpyHostToDevice));
    ew_.work = dw; ew_.nW = (int)work.size();
    // work items of k_ewald_force: runs of consecutive n with the same (l, m) (the table is in ewald_rec's nested-loop order)
    std::vector<EwaldG> groups;
    for (size_t k = 0; k < kv.size(); k++)
    {
        if (!groups.empty() && groups.back().l == kv[k].l && groups.back().m == kv[k].m && groups.back().nHi + 1 == kv[k].n) groups.back().nHi = kv[k].n;
        else groups.push_back(EwaldG{kv[k].l, kv[k].m, kv[k].n, kv[k].n, (int)k});
    }
    EwaldG* dg = (EwaldG*)alloc(sizeof(EwaldG) * groups.size());
    HIP_CHECK(hipMemcpy(dg, groups.data(), sizeof(EwaldG) * groups.size(), hipMemcpyHostToDevice));
    ew_.groups = dg; ew_.nG = (int)groups.size();
    ew_.T = (double*)alloc(sizeof(double) * 2 * kv.size());
    ew_.nK = (int)kv.size(); ew_.kx = m.ewald_k[0]; ew_.ky = m.ewald_k[1]; ew_.kz = m.ewald_k[2];
    ew_.nBlocksA = std::max(1, std::min(1024, div_up(capacity_, kEwTile)));
    ew_.partial = (double*)alloc(sizeof(double) * 2 * (size_t)ew_.nK * ew_.nBlocksA);
    ew_.S = (double*)alloc(sizeof(double) * 2 * (size_t)ew_.nK);
    HIP_CHECK(hipMemset(ew_.S, 0, sizeof(double) * 2 * (size_t)ew_.nK));
    ew_.scale = m.el_scale; ew_.scale2 = m.el_scale2;
    {   // both kernels keep the three per-atom harmonic tables in dynamic LDS: 1 KiB per harmonic (+ 12 KiB of force slices).  Beyond
        // the default 64 KiB a kernel has to be told (up to the CU's 160 KiB); beyond that the sum cannot run in this layout
        const size_t ldsTab = sizeof(double) * 2 * (size_t)kEwTile * (ew_.kx + ew_.ky + ew_.kz);
        const size_t ldsForce = ldsTab + sizeof(double) * 3 * kEwTile * kEwSlices;
        int devMax = 0;
        HIP_CHECK(hipDeviceGetAttribute(&devMax, hipDeviceAttributeMaxSharedMemoryPerBlock, opt_.device));
        const size_t hwMax = std::max<size_t>((size_t)devMax, 160 * 1024);
        if (ldsForce > hwMax)
            throw std::runtime_error("elec pme: kx + ky + kz = " + std::to_string(ew_.kx + ew_.ky + ew_.kz) + " harmonics need " + std::to_string(ldsForce) +
                                     " B of LDS per workgroup, more than the " + std::to_string(hwMax) + " B a CU has (use fewer k-vectors or elec fenn)");
        if (ldsTab > 64 * 1024) HIP_CHECK(hipFuncSetAttribute((const void*)k_ewald_sfac, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsTab));
        if (ldsForce > 64 * 1024) HIP_CHECK(hipFuncSetAttribute((const void*)k_ewald_force, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsForce));
    }
    DevStats s;
    HIP_CHECK(hipMemcpy(&s, dStats_, sizeof(DevStats), hipMemcpyDeviceToHost));
    s.engCoulConst = m.engElec1;
    HIP_CHECK(hipMemcpy(dStats_, &s, sizeof(DevStats), hipMemcpyHostToDevice));
}

// sim->add_elec = ewald_rec (main.cpp:99 ; GPU path main.cu:330-335): structure factors of the owned atoms, their sum over the
// ranks, then energy and per-atom forces (added to what the pair kernel wrote)
void Engine::launch_ewald()
{
    const size_t lds = sizeof(double) * 2 * (size_t)kEwTile * (ew_.kx + ew_.ky + ew_.kz);
    timed("ewald_sfac", [&] {
        hipLaunchKernelGGL(k_ewald_sfac, dim3(ew_.nBlocksA), dim3(256), lds, stream_, P_, S_, cur(), dCounts_, ew_);
        hipLaunchKernelGGL(k_ewald_reduce, dim3(div_up(ew_.nK, 64)), dim3(64 * kEwRedGroups), 0, stream_, ew_);
    });
    if (nranks_ > 1) timed("ewald_allreduce", [&] { xch_->allreduce_device(ew_.S, 2 * ew_.nK, stream_); });
    timed("ewald_energy", [&] { hipLaunchKernelGGL(k_ewald_energy, dim3(1), dim3(256), 0, stream_, ew_, dStats_); });
    timed("ewald_force", [&] {
        hipLaunchKernelGGL(k_ewald_force, dim3(div_up(capacity_, kEwTile)), dim3(kEwTile * kEwSlices), lds + sizeof(double) * 3 * kEwTile * kEwSlices,
                           stream_, P_, S_, cur(), dCounts_, ew_);
    });
    check_launch("Ewald kernels");
}

void Engine::upload_initial()
{
    const Model& m = model_;
    const int N = m.nAt;
    // which atoms start on this rank: all of them on one GPU, the slab's cell layers otherwise
    std::vector<int32_t> ids; ids.reserve(capacity_);
    for (int i = 0; i < N; i++)
    {
        if (nranks_ > 1)
        {
            int gx = (int)std::floor(m.x[i] * P_.icsz[0]);
            gx %= P_.nc[0]; if (gx < 0) gx += P_.nc[0];
            const int lo = P_.cx0 + P_.hw[0], hi = lo + P_.ncxLocal - 2 * P_.hw[0];
            if (gx < lo || gx >= hi) continue;
        }
        ids.push_back(i);
    }
    const int n = (int)ids.size();
    if (n > capacity_) throw std::runtime_error("slab capacity exceeded at initialisation");
    AtomArrays& A = cur();
    std::vector<double> tmp(std::max(n, 1));
    auto up = [&](double* dst, const std::vector<double>& src) {
        for (int k = 0; k < n; k++) tmp[k] = src[ids[k]];
        HIP_CHECK(hipMemcpy(dst, tmp.data(), sizeof(double) * (size_t)n, hipMemcpyHostToDevice));
    };
    up(A.x, m.x); up(A.y, m.y); up(A.z, m.z); up(A.vx, m.vx); up(A.vy, m.vy); up(A.vz, m.vz);
    std::vector<int32_t> ty(std::max(n, 1));
    for (int k = 0; k < n; k++) ty[k] = m.types[ids[k]];
    HIP_CHECK(hipMemcpy(A.type, ty.data(), sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(A.id, ids.data(), sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice));
    if (m.tstat_type == AZTOT_TSTAT_RADI)
    {   // init_cuda_tstat: cuTemp.cu:25-60 ; tables: read_tstat temperature.cpp:113-245
        std::vector<double> ph(N), ux(kNumUnitVectors), uy(kNumUnitVectors), uz(kNumUnitVectors);
        photon_engs(N, ph.data(), m.Temp, opt_.seed);
        unit_vectors(ux.data(), uy.data(), uz.data());
        auto alloc_up = [&](double*& d, const std::vector<double>& h) {
            void* p = nullptr; HIP_CHECK(hipMalloc(&p, sizeof(double) * h.size())); allocs_.push_back(p); d = (double*)p;
            HIP_CHECK(hipMemcpy(d, h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice));
        };
        alloc_up(dPhotons_, ph); alloc_up(dUvx_, ux); alloc_up(dUvy_, uy); alloc_up(dUvz_, uz);
        for (int k = 0; k < n; k++) tmp[k] = initial_radius(opt_.seed, (uint64_t)ids[k]);
        HIP_CHECK(hipMemcpy(A.rad, tmp.data(), sizeof(double) * (size_t)n, hipMemcpyHostToDevice));
    }
    Counts c{};
    c.ownedBegin = 0; c.ownedEnd = n; c.nTotal = n; c.srcBegin = 0; c.srcEnd = n;
    HIP_CHECK(hipMemcpy(dCounts_, &c, sizeof(Counts), hipMemcpyHostToDevice));
    // init_serial computes all_pairs only (sys_init.cpp:1181-1184): bonds and angles first act in step 1's force phase
    if (opt_.initial_forces) forces(false);
}

// ---------------------------------------------------------------------------------------------------
// pair kernel dispatch
// ---------------------------------------------------------------------------------------------------
// which pair kernel runs: 1 per-atom gather (any geometry), 2 one wave per cell with its own LDS tile (+ the pair lists of the lazy re-sort).
// 0 = the best one that supports the system.  (Round 2's variant 3 - four waves sharing a tile of 16-atom cell bins, measured 8 % slower than 2 - was
// retired in round 4: NOTES.md.)
int Engine::pair_variant() const
{
    int variant = opt_.pair_variant;
    if (variant == 0) variant = pair_tile_supported(P_) ? 2 : 1;
    if (variant == 2 && !pair_tile_supported(P_)) variant = 1;
    return variant;
}

void Engine::launch_pair()
{
    const int variant = pair_variant();
    if (variant == 2)
    {
        StepParams Q = P_;
        Q.fuseKick = fuseNow_ ? 1 : 0;
        // pair energies are looked at through the statistics of a call's last step only (finish_steps): the list kernel of every other step books none
        // (options.energies_every_step: every step does).  Inside a graph the last step of the cycle is the one that may be the call's last
        const bool wantEnergies = stepsLeftInRun_ == 0 || (debug_ & DBG_ENERGIES_EVERY_STEP);
        const PairLists pl = pair_lists();
        if (overlapHalo_)
        {   // interior x-layers [2 hw, ncx - 2 hw) first; then, once the neighbours' coordinates have landed, the two runs of boundary layers
            const int plane = P_.nc[1] * P_.nc[2], hw = P_.hw[0];
            PairRange in, lo, hi;
            in.first = 2 * hw * plane; in.n = (P_.ncxLocal - 4 * hw) * plane;
            lo.first = hw * plane; lo.n = hw * plane;
            hi.first = (P_.ncxLocal - 2 * hw) * plane; hi.n = hw * plane;
            const bool lists = candMode_ == 2 && pl.cand;
            int nb = 0;
            auto run = [&](PairRange& r) {
                r.blockBase = nb;
                if (lists)
                {
                    nb += launch_pair_list(Q, S_, dPots_, cur(), dCounts_, dCellStart_, dPartials_, maxBlocks_, stream_, r, pl, NextStep(), wantEnergies);
                    nb += launch_pair_cleanup(Q, S_, dPots_, cur(), dCounts_, dCellStart_, dPartials_, maxBlocks_, stream_, r, pl);
                }
                else
                {
                    launch_pair_tile(Q, S_, dPots_, cur(), dCounts_, dCellStart_, dPartials_, maxBlocks_, stream_, r, PairLists(), 0, NextStep(), split_);
                    nb += pair_range_grid(r.n) * split_.n;
                }
            };
            timed(lists ? "pair_list" : "pair_tile", [&] {
                run(in);
                HIP_CHECK(hipStreamWaitEvent(stream_, evHalo_, 0));
                run(lo);
                run(hi);
            });
            splitBlocks_ = nb;
            overlapHalo_ = false;
        }
        else
        {
            splitBlocks_ = 0;
            if (candMode_ != 0 && pl.cand)
            {   // lazy re-sort.  The step that rebuilds the cells first makes the lists (candidates of every tile, partners of every atom); then, like every
                // plain step until the next rebuild, it walks them; the clean-up launch stages the cells that keep no list
                if (candMode_ == 1)
                {   // (k_rank_gather has cleared the count of cells without a list)
                    timed("build_lists", [&] { launch_build_lists(Q, dCellStart_, stream_, PairRange(), pl); });
                    listsValid_ = true;
                    unlistedState_ = 0;
                    if (nranks_ > 1 && !capturing_)
                    {
                        if (!hUnlisted_)
                        {
                            HIP_CHECK(hipHostMalloc((void**)&hUnlisted_, sizeof(int32_t) * 4, hipHostMallocDefault));
                            HIP_CHECK(hipEventCreateWithFlags(&evUnlisted_, hipEventDisableTiming));
                        }
                        HIP_CHECK(hipMemcpyAsync(hUnlisted_, dNoList_ + 2, sizeof(int32_t), hipMemcpyDeviceToHost, stream_));
                        HIP_CHECK(hipEventRecord(evUnlisted_, stream_));
                    }
                }
                else if (nranks_ > 1 && unlistedState_ == 0 && hUnlisted_ && !capturing_ && hipEventQuery(evUnlisted_) == hipSuccess)
                    unlistedState_ = (hUnlisted_[0] == 0) ? 1 : 2;
                const bool skipCleanup = (nranks_ > 1 && unlistedState_ == 1 && candMode_ == 2 && !capturing_) || optimistic_;
                NextStep nx;
                nx.st = dStats_; nx.cnt = dCounts_; nx.R0 = ref_;
                if (fuseNext_) { nx.xn = altXyz_[cur_][0]; nx.yn = altXyz_[cur_][1]; nx.zn = altXyz_[cur_][2]; }
                if (fuseNext_ && fuseNextTstat_) { nx.photons = dPhotons_; nx.uvx = dUvx_; nx.uvy = dUvy_; nx.uvz = dUvz_; pairClosedStep_ = true; }
                else nx.pendingAfter = lazyKick_ ? 1 : -1;         // this step's second half-kick is owed to the next k_integrate1_bin (a call may open
                                                                    // with a plain step, where no scan re-arms the flag)
                timed("pair_list", [&] { splitBlocks_ = launch_pair_list(Q, S_, dPots_, cur(), dCounts_, dCellStart_, dPartials_, maxBlocks_, stream_, PairRange(), pl, nx, wantEnergies); });
                if (!skipCleanup)
                    timed("pair_cleanup", [&] { splitBlocks_ += launch_pair_cleanup(Q, S_, dPots_, cur(), dCounts_, dCellStart_, dPartials_, maxBlocks_, stream_, PairRange(), pl, nx); });
                if (fuseNext_)
                {   // the next step's positions are in the other set of coordinate arrays now
                    AtomArrays& A = cur();
                    std::swap(A.x, altXyz_[cur_][0]); std::swap(A.y, altXyz_[cur_][1]); std::swap(A.z, altXyz_[cur_][2]);
                    preIntegrated_ = true;
                }
            }
            else
                timed("pair_tile", [&] { launch_pair_tile(Q, S_, dPots_, cur(), dCounts_, dCellStart_, dPartials_, maxBlocks_, stream_, PairRange(), pl, 0, NextStep(), split_); });
        }
    }
    else
        timed("pair_atom", [&] {
            hipLaunchKernelGGL(k_pair_atom, dim3(div_up(capacity_, kBlock)), dim3(kBlock), 0, stream_, P_, S_, dPots_, cur(), dCounts_, dCellStart_,
                               dCellOfSorted_, dPartials_, maxBlocks_);
        });
    if (variant < 2) fuseNow_ = false;           // only the tile kernels have the fused epilogue (cannot happen: see the constructor)
    pairBlocksUsed_ = (variant == 2) ? (splitBlocks_ ? splitBlocks_ : pair_tile_grid(P_) * split_.n) : div_up(capacity_, kBlock);
    blocksEver_ = std::max(blocksEver_, std::max(pairBlocksUsed_, div_up(capacity_, kBlock)) + 1);
}

// slab ranks: where the boundary layers sit in the sorted arrays, as read back behind the last sort
void Engine::take_halo_info()
{
    if (!haloInfoPending_) return;
    HIP_CHECK(hipEventSynchronize(evHaloInfo_));
    haloInfoPending_ = false;
    adopt_halo_info(hHalo_);
}

// Behind every sort that opens an interval of plain steps the ranks tell each other how many boundary atoms they will send per plain step (k_rank_gather
// left {send count, ghost count} per side in dHaloInfo_[5..6] and [10..11]); the right neighbour's pair lands in [12..13], the left one's in [14..15].
// Fixed-size messages: this exchange cannot itself be mismatched.
void Engine::post_count_exchange()
{
    const int left = (rank_ + nranks_ - 1) % nranks_, right = (rank_ + 1) % nranks_;
    xch_->exchange_counts(left, right, dHaloInfo_ + 5, dHaloInfo_ + 10, dHaloInfo_ + 14, dHaloInfo_ + 12, 2, stream_);
}

// ... and before the first plain step posts a send or a receive, every rank checks that its ghost ranges are exactly as long as what the neighbours
// will send: a disagreement (a protocol error - the ranges are equal by construction) is reported as an error instead of hanging in RCCL or silently
// shifting coordinates onto the wrong atoms
void Engine::adopt_halo_info(const int32_t* h)
{
    halo_[0] = h[2]; halo_[1] = h[0]; halo_[2] = h[1]; halo_[3] = h[3]; halo_[4] = h[4];
    const int ghostsLeft = h[2], ghostsRight = h[4] - h[3], sendLeft = h[5], sendRight = h[10];
    // (both ranks of a boundary evaluate the same two equalities, so they fail together)
    if (h[12] != ghostsRight || h[13] != sendRight || h[14] != ghostsLeft || h[15] != sendLeft)
        throw std::runtime_error("slab decomposition: the neighbours disagree about their boundary atoms (left boundary: this rank sends " + std::to_string(sendLeft) +
                                 " and holds " + std::to_string(ghostsLeft) + " ghosts, the neighbour holds " + std::to_string(h[15]) + " and sends " + std::to_string(h[14]) +
                                 "; right boundary: sends " + std::to_string(sendRight) + ", holds " + std::to_string(ghostsRight) + ", the neighbour holds " +
                                 std::to_string(h[13]) + " and sends " + std::to_string(h[12]) + "): no coordinate exchange was posted");
}

// one message to each x-neighbour: migrants + halo (packed by k_integrate1_bin; protocol in slab.hip.h)
void Engine::exchange_halo()
{
    const int left = (rank_ + nranks_ - 1) % nranks_, right = (rank_ + 1) % nranks_;
    timed("exchange", [&] { xch_->exchange(left, right, dMsg_[0], dMsg_[1], dMsg_[2], dMsg_[3], lay_.bytes(), stream_); });
    const int recvCap = 2 * (lay_.migCap + lay_.haloCap);
    timed("unpack_halo", [&] {
        hipLaunchKernelGGL(k_unpack, dim3(div_up(recvCap, kBlock)), dim3(kBlock), 0, stream_, P_, cur(), dCounts_, capacity_, lay_, dMsg_[2], dMsg_[3],
                           dCellOf_, dSlotOf_, dCellCount_, dMsg_[0], dMsg_[1]);
    });
}

// ---------------------------------------------------------------------------------------------------
// iter_fastCellList (cuPairs.cu:2519-2567): histogram -> [halo] -> scan -> sort -> pair forces
// ---------------------------------------------------------------------------------------------------
void Engine::sort_and_forces(int stepMode, bool withBonded)
{
    const int gridAtoms = div_up(capacity_, kBlock);
    const bool integrate_first = stepMode != 0;
    P_.cycleStep = (stepMode == 2) ? sinceSort_ + 1 : 0;           // which step since the last rebuild (the slack-violation flag is indexed by it)
    if (stepMode == 2)
    {   // plain step of the lazy re-sort: integrate only; slots, cells and buffers stay as they are
        if (preIntegrated_) preIntegrated_ = false;                 // the previous step's pair kernel has opened this step already (NextStep)
        else if (nranks_ == 1 && P_.tstat != AZTOT_TSTAT_NOSE && !(debug_ & 16777216))
            // one GPU (the owned range starts at 0: 16-byte loads are aligned), nothing scales the velocities at the start of the step: two atoms per thread
            timed("integrate1", [&] {
                hipLaunchKernelGGL(k_integrate_plain2, dim3(div_up(div_up(capacity_, 2), kBlock)), dim3(kBlock), 0, stream_, P_, S_, cur(), dCounts_, dPartials_,
                                   maxBlocks_, dStats_, ref_);
            });
        else
        timed("integrate1", [&] {
            hipLaunchKernelGGL(k_integrate1_bin<2>, dim3(gridAtoms), dim3(kBlock), 0, stream_, P_, S_, cur(), dCounts_, dCellOf_, dSlotOf_,
                               dCellCount_, dPartials_, maxBlocks_, lay_, dMsg_[0], dMsg_[1], dStats_, ref_);
        });
        if (nranks_ > 1)
        {   // the neighbours hold this rank's boundary atoms in the order of the last sort: only their coordinates (and radii) travel
            const int left = (rank_ + nranks_ - 1) % nranks_, right = (rank_ + 1) % nranks_;
            take_halo_info();
            AtomArrays& A = cur();
            double* arr[4] = {A.x, A.y, A.z, A.rad};
            // The exchange touches the ghost ranges only, and the cells two or more layers inside the slab never read them: it runs on its
            // own stream next to the interior cells' pair forces; the boundary cells wait for it (launch_pair).  Not while kernels are being timed
            // one by one, not with the per-atom kernel, and not when there is no interior to speak of.
            const int interiorLayers = P_.ncxLocal - 4 * P_.hw[0];
            // OPT-IN (debug bit 16384): measured on one rank of 7 (loopback) the two cross-stream event waits and the two extra launches cost 30 us
            // where the exchange they hide takes 7 (0.0685 -> 0.0981 ms/step), so the default is the serial order
            overlapHalo_ = !profile_ && xch_->device_side() && pair_variant() == 2 && interiorLayers >= 1 && (debug_ & 16384);
            if (overlapHalo_)
            {
                HIP_CHECK(hipEventRecord(evIntegrated_, stream_));
                HIP_CHECK(hipStreamWaitEvent(commStream_, evIntegrated_, 0));
            }
            hipStream_t xs = overlapHalo_ ? commStream_ : stream_;
            timed("exchange_coords", [&] {
                xch_->exchange_ranges(left, right, arr, P_.use_radii ? 4 : 3, halo_[0], halo_[1] - halo_[0], halo_[2], halo_[3] - halo_[2], 0, halo_[0], halo_[3],
                                      halo_[4] - halo_[3], xs);
            });
            if (overlapHalo_) HIP_CHECK(hipEventRecord(evHalo_, commStream_));
        }
        sinceSort_++;
        candMode_ = 2;
    }
    else
    {
    candMode_ = (stepMode == 1 && lazyOn_ && lazyK_ > 1) ? 1 : 0;       // a step that opens an interval of plain steps records the lists
    if (integrate_first)
        timed("integrate1_bin", [&] {
            hipLaunchKernelGGL(k_integrate1_bin<1>, dim3(gridAtoms), dim3(kBlock), 0, stream_, P_, S_, cur(), dCounts_, dCellOf_, dSlotOf_,
                               dCellCount_, dPartials_, maxBlocks_, lay_, dMsg_[0], dMsg_[1], dStats_, ref_);
        });
    else
    {
        HIP_CHECK(hipMemsetAsync(dCellCount_, 0, sizeof(int32_t) * (size_t)(nCellAlloc_ + 1), stream_));
        timed("bin", [&] {
            hipLaunchKernelGGL(k_integrate1_bin<0>, dim3(gridAtoms), dim3(kBlock), 0, stream_, P_, S_, cur(), dCounts_, dCellOf_, dSlotOf_,
                               dCellCount_, dPartials_, maxBlocks_, lay_, dMsg_[0], dMsg_[1], dStats_, ref_);
        });
    }
    if (nranks_ > 1) exchange_halo();
    const int nChunks = div_up(P_.nCellLocal, kScanChunk);
    const int setPending = (integrate_first && lazyKick_) ? 1 : 0;
    timed("scan_cells", [&] {
        if (P_.nCellLocal <= kScanSingleMax)
            hipLaunchKernelGGL(k_scan_single, dim3(1), dim3(1024), 0, stream_, P_.nCellLocal, dCellCount_, dCellStart_, dCounts_, dStats_, setPending);
        else
        {
            hipLaunchKernelGGL(k_scan_totals, dim3(nChunks), dim3(kBlock), 0, stream_, P_.nCellLocal, dCellCount_, dChunkTot_);
            hipLaunchKernelGGL(k_scan_apply, dim3(nChunks), dim3(kBlock), 0, stream_, P_.nCellLocal, dCellCount_, dChunkTot_, dCellStart_, dCounts_, dStats_, setPending);
        }
    });
    timed("place", [&] {
        hipLaunchKernelGGL(k_place, dim3(gridAtoms), dim3(kBlock), 0, stream_, dCounts_, dCellOf_, dSlotOf_, dCellStart_, cur().id, dTmpId_, dTmpSrc_,
                           dTmpCell_);
    });
    timed("rank_gather", [&] {
        hipLaunchKernelGGL(k_rank_gather, dim3(gridAtoms), dim3(kBlock), 0, stream_, dCounts_, dCellStart_, dTmpId_, dTmpSrc_, dTmpCell_, cur(), oth(),
                           dCellOfSorted_, (P_.tstat == AZTOT_TSTAT_RADI || P_.use_radii || thermoTouched_) ? 2 : 0, P_, dCounts_, bonded_.idxOfId, ref_,
                           dHaloInfo_, (listsOn_ && stepMode == 1 && lazyOn_ && lazyK_ > 1) ? dNoList_ + 2 : nullptr, listsOn_ ? dRel_ : nullptr);
    });
    cur_ ^= 1;
    sinceSort_ = 0;
    if (!capturing_ && stepMode == 1) rebuilds_++;
    listsValid_ = false;            // (until this step's launch_pair records them)
    if (nranks_ > 1 && lazyOn_ && lazyK_ > 1)
    {   // where the boundary layers sit in the sorted arrays: [ownedBegin, end of layer 2hw-1) goes left, [start of layer ncx-2hw, ownedEnd) goes right;
        // ghosts are [0, ownedBegin) and [ownedEnd, nTotal).  One small read-back per sort.
        if (!hHalo_)
        {
            HIP_CHECK(hipHostMalloc((void**)&hHalo_, sizeof(int32_t) * 16, hipHostMallocDefault));
            HIP_CHECK(hipEventCreateWithFlags(&evHaloInfo_, hipEventDisableTiming));
        }
        post_count_exchange();
        HIP_CHECK(hipMemcpyAsync(hHalo_, dHaloInfo_, sizeof(int32_t) * 16, hipMemcpyDeviceToHost, stream_));
        HIP_CHECK(hipEventRecord(evHaloInfo_, stream_));
        haloInfoPending_ = true;      // (no stall here: take_halo_info waits when the numbers are needed)
    }
    }
    launch_pair();
    if (hasEwald_) launch_ewald();
    if (hasBonded_ && withBonded)      // exec_bondlist + exec_anglelist, main.cpp:101-104 (GPU path: main.cu:307-312,353-363)
        timed("bonded", [&] {
            if (nranks_ > 1)
                hipLaunchKernelGGL(k_bonded<true>, dim3(gridAtoms), dim3(kBlock), 0, stream_, P_, cur(), dCounts_, bonded_, dPartials_, maxBlocks_);
            else
                hipLaunchKernelGGL(k_bonded<false>, dim3(gridAtoms), dim3(kBlock), 0, stream_, P_, cur(), dCounts_, bonded_, dPartials_, maxBlocks_);
        });
    check_launch("sort + force kernels");
}

void Engine::collect_and_finalize(unsigned slotMask)
{
    timed("collect", [&] {
        hipLaunchKernelGGL(k_collect, dim3(PS_COUNT * kCollectParts), dim3(256), 0, stream_, dPartials_, maxBlocks_, div_up(capacity_, kBlock), pairBlocksUsed_,
                           dStage_, slotMask, ekinFromPair_ ? 1 : 0, blocksEver_);
    });
    timed("finalize", [&] { hipLaunchKernelGGL(k_finalize, dim3(1), dim3(64), 0, stream_, P_, dStats_, dStage_, slotMask); });
}

void Engine::forces(bool withBonded)
{
    if (!failed_.empty()) throw std::runtime_error("this handle failed in an earlier call: " + failed_);
    settle();
    snap_.valid = false;
    sinceSort_ = 1 << 30;           // a sort interval does not run on through a force call (it re-bins wrapped coordinates)
    sort_and_forces(0, withBonded);
    sinceSort_ = 1 << 30; listsValid_ = false; carryAgreed_ = false;
    // energies of this configuration; kinetic energy and wall counters are left untouched
    unsigned mask = (1u << PS_EVDW) | (1u << PS_ECOUL) | (1u << PS_DROPPED);
    if (hasBonded_ && withBonded) mask |= (1u << PS_EBOND) | (1u << PS_EANGLE);
    collect_and_finalize(mask);
    HIP_CHECK(hipMemsetAsync(dCellCount_, 0, sizeof(int32_t) * (size_t)(nCellAlloc_ + 1), stream_));
    sync();
    check_overflow();
}

// ---------------------------------------------------------------------------------------------------
// one iteration of the loop body of main.cu:281-410 (serial twin: main.cpp:89-142)
// ---------------------------------------------------------------------------------------------------
void Engine::launch_step_kernels()
{
    const int gridAtoms = div_up(capacity_, kBlock);
    if (P_.tstat == AZTOT_TSTAT_NOSE) timed("nose_begin", [&] { hipLaunchKernelGGL(k_nose_begin, dim3(1), dim3(64), 0, stream_, P_, dStats_); });
    // does this step need the all-atom kinetic energy on the device?  Nose-Hoover: every step; equilibration rescaling: the steps it acts on
    // (integrators.cpp:511-522: step <= nequil and a multiple of eqfreq) - the host knows the number of the step it is launching.  (A captured cycle is
    // replayed at many step numbers: cycles are captured only once the equilibration period is over, can_graph.)
    const long long iStep = hostStep_ + 1;
    const bool scalingDue = !capturing_ && P_.nEq > 0 && iStep <= P_.nEq && P_.freqEq > 0 && (iStep % P_.freqEq) == 0;
    const bool equil = scalingDue || P_.tstat == AZTOT_TSTAT_NOSE;
    // plain NVE steps leave integrate2 to somebody else (decided once, in the constructor): small systems / slabs -> the tile
    // kernel's epilogue (fuseEpilogue_); large ones -> the next step's k_integrate1_bin (lazyKick_, see finish_steps)
    fuseNow_ = fuseEpilogue_;
    const int stepMode = (lazyOn_ && sinceSort_ < lazyK_ - 1) ? 2 : 1;
    // is there a next step before the host looks or the cycle ends, and is it a plain one?
    const int sinceAfter = (stepMode == 2) ? sinceSort_ + 1 : 0;
    const bool nextPlain = lazyOn_ && sinceAfter < lazyK_ - 1 && stepsLeftInRun_ > 0;
    {   // does this step's pair kernel also open the next step?  Only if this step walks the lists (k_pair_list and its clean-up launch carry the epilogue)
        const bool lists = listsOn_ && lazyOn_ && lazyK_ > 1 && pair_variant() == 2;
        fuseNext_ = fuseNextOk_ && lists && nextPlain;
        // thermostat runs: only where no clean-up launch follows (it has no thermostat epilogue), on steps without equilibration scaling, never with energies on every step
        // (and one wave per cell: in the multi-wave kernels of dense systems the thermostat's registers cost the loop more than the launch saves -
        //  measured with a second radius array for case study 2's surk kernel: S40 0.076 -> 0.093 ms/step, case study 2 unchanged; not kept)
        if (fuseNextTstat_) fuseNext_ = fuseNext_ && optimistic_ && !equil && listWaves_ == 1 && !(debug_ & DBG_ENERGIES_EVERY_STEP);
    }
    pairClosedStep_ = false;
    sort_and_forces(stepMode);
    const bool fused = fuseNow_;                 // launch_pair drops the request if the tile kernel is not the one running
    fuseNow_ = false;
    ekinFromPair_ = fused;
    // radiative thermostat without equilibration scaling: nothing global happens between the second half-kick and the thermostat - one launch (debug bit
    // 4194304: two, as everywhere else)
    const bool closed = pairClosedStep_;           // (thermostat run, fused: the pair kernel has closed this step and opened the next)
    const bool kickAndPost = !closed && !fused && !lazyKick_ && !equil && P_.tstat == AZTOT_TSTAT_RADI && !(debug_ & 4194304);
    // ... and when a plain step follows, the same launch opens it (k_boundary_radi; debug bit 33554432: no)
    if (kickAndPost && nextPlain && !(debug_ & 33554432))
    {
        StepParams Q = P_;
        Q.cycleStep = sinceSort_ + 1;              // of the step being opened
        timed("boundary", [&] {
            hipLaunchKernelGGL(k_boundary_radi, dim3(gridAtoms), dim3(kBlock), 0, stream_, Q, S_, cur(), dCounts_, dPartials_, maxBlocks_, dStats_, dPhotons_, dUvx_, dUvy_,
                               dUvz_, ref_);
        });
        preIntegrated_ = true;
    }
    else if (kickAndPost)
        timed("integrate2_post", [&] {
            hipLaunchKernelGGL(k_integrate2_post, dim3(gridAtoms), dim3(kBlock), 0, stream_, P_, S_, cur(), dCounts_, dCellCount_, P_.nCellLocal, dPartials_,
                               maxBlocks_, dStats_, dPhotons_, dUvx_, dUvy_, dUvz_);
        });
    else if (!closed && !fused && !lazyKick_)
        timed("integrate2", [&] {
            hipLaunchKernelGGL(k_integrate2, dim3(gridAtoms), dim3(kBlock), 0, stream_, P_, S_, cur(), dCounts_, dCellCount_, P_.nCellLocal, dPartials_,
                               maxBlocks_, dStats_);
        });
    if (equil)
    {
        timed("reduce_kin", [&] {
            hipLaunchKernelGGL(k_reduce_kin, dim3(1), dim3(1024), 0, stream_, P_, dPartials_, maxBlocks_, gridAtoms, dStats_, dEkGlobal_);
        });
        if (nranks_ > 1)
        {   // the scaling factor needs the kinetic energy of ALL ranks (equilibration steps only)
            double ek = 0.0;
            HIP_CHECK(hipMemcpyAsync(&ek, dEkGlobal_, sizeof(double), hipMemcpyDeviceToHost, stream_));
            HIP_CHECK(hipStreamSynchronize(stream_));
            xch_->allreduce_sum(&ek, 1, stream_);
            HIP_CHECK(hipMemcpyAsync(dEkGlobal_, &ek, sizeof(double), hipMemcpyHostToDevice, stream_));
        }
        timed("scale_decision", [&] { hipLaunchKernelGGL(k_scale_decision, dim3(1), dim3(64), 0, stream_, P_, dStats_, dEkGlobal_); });
    }
    if ((equil || P_.tstat == AZTOT_TSTAT_RADI) && !kickAndPost && !closed)
        timed("post_tstat", [&] {
            hipLaunchKernelGGL(k_post, dim3(gridAtoms), dim3(kBlock), 0, stream_, P_, S_, cur(), dCounts_, dStats_, dPhotons_, dUvx_, dUvy_, dUvz_,
                               dPartials_, maxBlocks_);
        });
    if (scalingDue && P_.tstat != AZTOT_TSTAT_NOSE)
    {   // the factor has been applied: the steps that follow take the short path, which never decides one (DevStats::vscale is read by every thermostat kernel)
        static const double one = 1.0;
        HIP_CHECK(hipMemcpyAsync(&dStats_->vscale, &one, sizeof(double), hipMemcpyHostToDevice, stream_));
    }
    lastStepEquil_ = equil;
    if (!capturing_) hostStep_++;
}

// the per-block partial sums are folded into the statistics only when somebody can look at them: at the end of a
// step() call (energies are those of the last step; wall momenta / crossing counts pile up in between)
void Engine::finish_steps()
{
    if (kickOwed_)
    {   // the deferred second half-kick of the last step + its kinetic energy
        timed("integrate2", [&] {
            hipLaunchKernelGGL(k_integrate2, dim3(div_up(capacity_, kBlock)), dim3(kBlock), 0, stream_, P_, S_, cur(), dCounts_, dCellCount_, P_.nCellLocal,
                               dPartials_, maxBlocks_, dStats_);
        });
        kickOwed_ = false;
    }
    unsigned mask = (1u << PS_COUNT) - 1u;
    if (lastStepEquil_) mask &= ~(1u << PS_EKIN);   // k_reduce_kin / k_scale_decision own engKin then
    if (!(P_.nEq > 0 || P_.tstat == AZTOT_TSTAT_RADI)) mask &= ~(1u << PS_ETEMP);
    if (!hasBonded_) mask &= ~((1u << PS_EBOND) | (1u << PS_EANGLE));
    collect_and_finalize(mask);
}

// may the plain steps of the next window run without the clean-up launch?  (decided at the start of a call and after every look: the answer is baked into
// kernel arguments, so a change drops the captured graphs)
void Engine::choose_optimism()
{
    // (any size since round 4: on 1 M atoms the launch cost 6.9 us of a 123 us step - the drain and refill of the chip around a kernel that finds nothing
    //  to do -, a snapshot is 100 MB = 33 us per look, and looks are up to 256 steps apart)
    const bool want = nranks_ == 1 && lazyOn_ && listsOn_ && lazyMeasured_ && lazyK_ > 1 && pair_variant() == 2 &&
                      safeLooks_ == 0 && !unlistedAtLook_ && unlistedState_ != 2 && !(debug_ & DBG_ALWAYS_CLEANUP);
    if (want != optimistic_)
    {
        optimistic_ = want;
        P_.optimistic = want ? 1 : 0;
        destroy_graphs(); graphCycle_ = 0;
        if (!want) snap_.valid = false;
    }
}

// Slab ranks: the dynamic state as it stands (the host has just looked and found no skin violation), device to device.  The arrays are those an exact
// restart needs (aztot_state + aztot_clock): everything else is rebuilt by the step that rebuilds the cells, and the replay opens with one.
void Engine::take_snapshot()
{
    const size_t nd = sizeof(double) * (size_t)capacity_, ni = sizeof(int32_t) * (size_t)capacity_;
    if (!snap_.A.x)
    {
        auto alloc = [&](size_t bytes) { void* p = nullptr; HIP_CHECK(hipMalloc(&p, std::max<size_t>(bytes, 16))); allocs_.push_back(p); return p; };
        double** d[] = {&snap_.A.x, &snap_.A.y, &snap_.A.z, &snap_.A.vx, &snap_.A.vy, &snap_.A.vz, &snap_.A.fx, &snap_.A.fy, &snap_.A.fz, &snap_.A.U, &snap_.A.rad};
        for (auto pp : d) *pp = (double*)alloc(nd);
        snap_.A.type = (int32_t*)alloc(ni); snap_.A.id = (int32_t*)alloc(ni);
        snap_.stats = alloc(sizeof(DevStats)); snap_.counts = alloc(sizeof(Counts));
        snap_.partials = (double*)alloc(sizeof(double) * (size_t)PS_COUNT * maxBlocks_);
    }
    const AtomArrays& A = cur();
    StateCopy C;
    const double* src[] = {A.x, A.y, A.z, A.vx, A.vy, A.vz, A.fx, A.fy, A.fz, A.U, A.rad};
    double* dst[] = {snap_.A.x, snap_.A.y, snap_.A.z, snap_.A.vx, snap_.A.vy, snap_.A.vz, snap_.A.fx, snap_.A.fy, snap_.A.fz, snap_.A.U, snap_.A.rad};
    for (int k = 0; k < 11; k++) { C.srcD[k] = src[k]; C.dstD[k] = dst[k]; }
    C.srcI[0] = A.type; C.dstI[0] = snap_.A.type; C.srcI[1] = A.id; C.dstI[1] = snap_.A.id;
    C.srcP = dPartials_; C.dstP = snap_.partials; C.nP = (long long)PS_COUNT * maxBlocks_;
    C.srcS[0] = (const int32_t*)dStats_; C.dstS[0] = (int32_t*)snap_.stats; C.nS[0] = (int)(sizeof(DevStats) / 4);
    C.srcS[1] = (const int32_t*)dCounts_; C.dstS[1] = (int32_t*)snap_.counts; C.nS[1] = (int)(sizeof(Counts) / 4);
    hipLaunchKernelGGL(k_copy_state, dim3(div_up(capacity_, kBlock)), dim3(kBlock), 0, stream_, C, capacity_);
    snap_.buf = buf_state();
    snap_.hostStep = hostStep_;
    snap_.valid = true;
    stepsSinceSnap_ = 0;
}

// ... and back to it: the steps since the snapshot run again with the cells rebuilt on every one of them (no atom can leave a slack that is never used)
void Engine::replay_from_snapshot()
{
    if (!snap_.valid) throw std::runtime_error("lazy re-sort on slab ranks: an atom left its cell's slack and there is no snapshot to go back to");
    sync();
    destroy_graphs(); graphCycle_ = 0;
    set_buf_state(snap_.buf);
    const AtomArrays& A = cur();
    StateCopy C;
    double* dst[] = {A.x, A.y, A.z, A.vx, A.vy, A.vz, A.fx, A.fy, A.fz, A.U, A.rad};
    const double* src[] = {snap_.A.x, snap_.A.y, snap_.A.z, snap_.A.vx, snap_.A.vy, snap_.A.vz, snap_.A.fx, snap_.A.fy, snap_.A.fz, snap_.A.U, snap_.A.rad};
    for (int k = 0; k < 11; k++) { C.srcD[k] = src[k]; C.dstD[k] = dst[k]; }
    C.srcI[0] = snap_.A.type; C.dstI[0] = A.type; C.srcI[1] = snap_.A.id; C.dstI[1] = A.id;
    C.srcP = snap_.partials; C.dstP = dPartials_; C.nP = (long long)PS_COUNT * maxBlocks_;
    C.srcS[0] = (const int32_t*)snap_.stats; C.dstS[0] = (int32_t*)dStats_; C.nS[0] = (int)(sizeof(DevStats) / 4);
    C.srcS[1] = (const int32_t*)snap_.counts; C.dstS[1] = (int32_t*)dCounts_; C.nS[1] = (int)(sizeof(Counts) / 4);
    hipLaunchKernelGGL(k_copy_state, dim3(div_up(capacity_, kBlock)), dim3(kBlock), 0, stream_, C, capacity_);
    sinceSort_ = 1 << 30; listsValid_ = false; carryAgreed_ = false; preIntegrated_ = false; haloInfoPending_ = false; unlistedState_ = 0;
    kickOwed_ = false;
    hostStep_ = snap_.hostStep;
    if (nranks_ > 1) { if (lazyK_ != 1) lazyK_ = 1; }
    else
    {   // one GPU: the same steps with the clean-up launch behind every k_pair_list (it stages the cells without a list and, from the step of a violation
        // on, every cell with the wider stencil)
        optimistic_ = false; P_.optimistic = 0;
        rollbacks_ = std::min(rollbacks_ + 1, 12);
        safeLooks_ = 4 << rollbacks_;          // 8, 16, 32 ... looks on the safe side: a system that keeps outgrowing its lists stops trying
    }
    const long long n = stepsSinceSnap_;
    if (std::getenv("AZTOT_VERBOSE")) std::fprintf(stderr, "aztot: rank %d: a skin violation (or a cell without a list) found at a look - %lld steps run again, %s\n", rank_, n, nranks_ > 1 ? "cells rebuilt every step" : "with the clean-up launch");
    if (n > 0) run_steps((int)n);
    stepsSinceSnap_ = n;
}

// An error inside a call leaves the host's picture of the device (cycles, lists, the deferred kick, its own step count) unreliable: the handle is dead
// for stepping from then on and says why; reads (aztot_get_stats, aztot_md_to_host, aztot_get_clock) still work for a post-mortem and report what the
// device holds (a secondary "step counts differ" error used to hide the first one).
void Engine::mark_failed(const char* what) noexcept
{
    if (failed_.empty()) failed_ = what ? what : "unknown error";
    unsettled_ = false;
    snap_.valid = false;
    if (stream_) (void)hipStreamSynchronize(stream_);
    (void)hipGetLastError();
    DevStats s;
    if (dStats_ && hipMemcpy(&s, dStats_, sizeof(DevStats), hipMemcpyDeviceToHost) == hipSuccess) hostStep_ = s.step;
    (void)hipGetLastError();
}

void Engine::step(int nsteps)
{
    if (nsteps <= 0) return;
    if (!failed_.empty()) throw std::runtime_error("this handle failed in an earlier call and cannot step on: " + failed_);
    try { step_body(nsteps); }
    catch (const std::exception& e) { mark_failed(e.what()); throw; }
    catch (...) { mark_failed("unknown exception"); throw; }
}

// Does this call end with the look / statistics / read-backs right away?  Slab ranks (collectives every rank must take together), bonded terms
// (an overflow flag to read back), per-kernel timing (events to drain): always.  One GPU otherwise: only when a look at the sort interval is due
// (8, 16, ... 256 steps after the last one, or never made) - else the call returns with its kernels queued, and whoever reads state or statistics next
// (or the next look) settles: a caller that steps one step at a time (the reference's loop is per step, main.cu:281-410) pays for the second half-kick
// launch, the statistics reduction, the stream synchronisation and the snapshot once per look instead of once per step.
bool Engine::settle_now() const
{
    if (nranks_ > 1 || hasBonded_ || profile_ || (debug_ & DBG_SETTLE_EVERY_CALL)) return true;
    if (!lazyOn_) return false;
    return !lazyMeasured_ || sinceLook_ >= lazyWindow_;
}

// The end of a call, possibly deferred (settle_now): the deferred second half-kick and the statistics of the last step, the look at the sort interval
// (a window of steps run again where it finds a skin violation that nothing had repaired on the spot), overflow checks on the state the caller will see,
// what the next call needs.
void Engine::settle()
{
    if (!unsettled_ || !failed_.empty()) return;
    unsettled_ = false;
    try
    {
        const int windowCap = nranks_ > 1 ? 64 : 256;
        finish_steps();
        check_launch("end-of-call kernels");
        look_sync();
        if (lazyOn_)
        {
            if (!adapt_sort_interval())
            {
                replay_from_snapshot();
                kickOwed_ = lazyKick_;
                finish_steps();
                check_launch("step kernels (run again)");
                look_sync();
                if (!adapt_sort_interval()) throw std::runtime_error("lazy re-sort: a skin violation in a run that rebuilds its cells every step");
            }
            if (2 * sinceLook_ >= lazyWindow_) lazyWindow_ = std::min(windowCap, 2 * lazyWindow_);      // (this look stands in for the one that was due)
            sinceLook_ = 0;
            if (safeLooks_ > 0) safeLooks_--;
        }
        // (on the state the caller will see: an optimistic window that had to be run again is inexact until it has been - ADVICE round 3)
        check_overflow();
        prepare_next_call();
        if (lazyOn_)
        {   // (behind prepare_next_call: it may have recorded the engine's first lists and knows whether every cell got one)
            choose_optimism();
            // The snapshot a later look may have to go back to.  A verified one that is still young is kept (going back a few steps further costs a little
            // more in the rare case, a 100 MB copy per call costs every caller that steps in short calls)
            if (!rollback_on()) snap_.valid = false;
            else if (!snap_.valid || stepsSinceSnap_ >= kSnapshotKeepSteps) take_snapshot();
        }
    }
    catch (const std::exception& e) { mark_failed(e.what()); throw; }
}

void Engine::sync_all()
{
    settle();
    sync();
}

void Engine::step_body(int nsteps)
{
    // The sort interval is re-evaluated ("a look": a stream synchronisation and a few small read-backs, ~0.1 ms with the pipeline refill) at the end of every
    // call and, inside long calls, whenever lazyWindow_ steps have gone by since the last look - 8, 16, 32, ... then every 256 (a run that starts from rest speeds
    // up for a while: the looks are close together where that happens).  The count runs on across calls (sinceLook_), and a look is skipped when the call
    // ends within half a window anyway: a short call right behind a look - the driver's 20 steps after 5 of warm-up - pays for no look of its own.
    const int windowCap = nranks_ > 1 ? 64 : 256;     // (a slab rank repairs a skin violation by running the window again: it looks more often)
    // (a call whose end was deferred - settle_now - and that nobody has looked at since: its second half-kick is still owed on the device, DevStats::pendingKick,
    //  and this call's first integrate kernel pays it exactly as it does between two steps of one call; its statistics were never asked for)
    kickOwed_ = false;
    if (!unsettled_)
    {
        choose_optimism();
        if (rollback_on() && !snap_.valid) take_snapshot();         // (the state as the caller left it: set_state / aztot_forces / the first call)
    }
    bool roll = rollback_on();
    int left = nsteps;
    while (left > 0)
    {
        const int n = lazyOn_ ? std::min(left, std::max(1, lazyWindow_ - sinceLook_)) : left;
        run_steps(n);
        left -= n;
        sinceLook_ += n;
        stepsSinceSnap_ += n;
        if (left > 0 && sinceLook_ >= lazyWindow_)
        {
            if (2 * left > lazyWindow_)
            {
                look_sync();
                if (!adapt_sort_interval())
                {   // (slab ranks, all together) an atom left its slack somewhere since the snapshot: those steps again, exactly, then a look that cannot fail
                    replay_from_snapshot();
                    look_sync();
                    if (!adapt_sort_interval()) throw std::runtime_error("lazy re-sort: a skin violation in a run that rebuilds its cells every step");
                }
                if (safeLooks_ > 0) safeLooks_--;
                choose_optimism();
                roll = rollback_on();
                if (roll) take_snapshot();
                sinceLook_ = 0;
                lazyWindow_ = std::min(windowCap, 2 * lazyWindow_);
            }
            else { run_steps(left); sinceLook_ += left; stepsSinceSnap_ += left; left = 0; }      // the end of the call, at most half a window away, is the look
        }
    }
    kickOwed_ = lazyKick_;          // set here, not in launch_step_kernels: a replayed graph does not pass through the host code
    check_launch("step kernels");
    unsettled_ = true;
    if (settle_now()) settle();
}

// steps per graph: one sort interval; with the cells rebuilt every step the sort ping-pongs the buffers, so it takes two steps to come back
int Engine::graph_cycle() const
{
    const int K = lazyOn_ ? lazyK_ : 1;
    return (K == 1) ? 2 : K;
}

bool Engine::can_graph() const
{
    // slab ranks: only with the loopback transport and only on request (debug bit 4096) - an experiment, see DESIGN.md section 6
    const bool slabGraph = nranks_ > 1 && ownedXch_ && (debug_ & 4096) && !(P_.nEq > 0 || P_.tstat == AZTOT_TSTAT_NOSE);
    // Replaying captured cycles pays where a step is a handful of microsecond kernels.  On 1 M atoms the kernels are long enough for plain asynchronous
    // launches to keep the GPU busy, and each hipGraphLaunch costs a 40 us bubble in front of its first kernel (rocprofv3 kernel trace): 0.1594 ms/step
    // replayed, 0.1575 launched one by one; 40 000 atoms: 0.0184 replayed, 0.0187 one by one.  (Debug bit 8388608: replay whatever the size.)
    const bool worthIt = capacity_ <= 2 * kFuseKickMaxAtoms || (debug_ & 8388608);
    return opt_.use_graph && worthIt && (nranks_ == 1 || slabGraph) && !profile_ && !equil_phase();
}

// the graph of one cycle of steps for the buffer state the engine is in (captured on first use; the capture executes nothing).  One graph per buffer
// state because kernel arguments are baked in at capture time; a graph starts with a step that sorts
Engine::GraphSlot* Engine::graph_for_state(int cycle)
{
    if (graphCycle_ != cycle) { destroy_graphs(); graphCycle_ = cycle; }
    const BufState now = buf_state();
    for (GraphSlot& g : graphs_)
        if (g.before.cur == now.cur && g.before.xyz[0][0] == now.xyz[0][0] && g.before.xyz[1][0] == now.xyz[1][0]) return &g;
    GraphSlot g;
    g.before = now;
    const int sinceBefore = sinceSort_;
    const LaunchNotes notesBefore = launch_notes();      // (the capture executes nothing: what the host knows about the last EXECUTED step must survive it)
    HIP_CHECK(hipStreamBeginCapture(stream_, hipStreamCaptureModeThreadLocal));
    capturing_ = true;
    try
    {
        sinceSort_ = 1 << 30;
        preIntegrated_ = false;
        for (int k = 0; k < cycle; k++) { stepsLeftInRun_ = cycle - 1 - k; launch_step_kernels(); }
    }
    catch (...)
    {   // leave neither the stream in capture mode nor the engine believing it is capturing
        hipGraph_t broken = nullptr;
        (void)hipStreamEndCapture(stream_, &broken);
        if (broken) (void)hipGraphDestroy(broken);
        capturing_ = false;
        preIntegrated_ = false;
        sinceSort_ = sinceBefore;
        set_buf_state(now);
        adopt_launch_notes(notesBefore);
        throw;
    }
    capturing_ = false;
    HIP_CHECK(hipStreamEndCapture(stream_, &g.graph));
    HIP_CHECK(hipGraphInstantiate(&g.exec, g.graph, nullptr, nullptr, 0));
    g.after = buf_state();
    g.notes = launch_notes();         // ... of the cycle's last step
    set_buf_state(now);               // the capture itself executed nothing
    adopt_launch_notes(notesBefore);
    sinceSort_ = sinceBefore;
    stepsLeftInRun_ = 0;
    graphs_.push_back(g);
    return &graphs_.back();
}

void Engine::run_steps(int nsteps)
{
    int done = 0;
    // Lazy re-sort (one GPU): the reference rebuilds its cell list every step (main.cu:300-326); here a step re-sorts only every lazyK_-th time.  That is
    // exact as long as no atom is farther than (stencil reach - rc) / 2 from where it was when the cells were built: every pair inside rc is then still
    // found in the stencil of the cell the atoms were sorted into.  Plain steps leave slots, cells and buffers alone, keep coordinates unwrapped, count
    // wall crossings when they happen, and check every atom's displacement; should one ever leave the slack (the interval is sized with a factor 1.15 to
    // spare from the largest step seen), the pair kernels reach one cell further until the next sort - slower, still exact.  The first step of
    // every call sorts (the deferred half-kick is re-armed by the scan).
    // With pair lists a sort interval runs on from the previous call when nothing has touched the state since (the lists are those of the arrays as they
    // stand; slab ranks: when all ranks agreed at the end of the previous call that they can - a rebuild step carries the full exchange, so they must
    // take it together); the plain steps that finish it are launched one by one, whole cycles are replayed as graphs after that.
    const bool carryOn = lazyOn_ && (nranks_ == 1 || carryAgreed_) && listsOn_ && listsValid_ && lazyK_ > 1 && sinceSort_ < (1 << 29) && pair_variant() == 2;
    if (!carryOn) sinceSort_ = 1 << 30;
    preIntegrated_ = false;
    while (carryOn && done < nsteps && sinceSort_ < lazyK_ - 1) { stepsLeftInRun_ = nsteps - 1 - done; launch_step_kernels(); done++; }
    const int cycle = graph_cycle();
    if (can_graph() && nsteps - done >= cycle)
    {
        sinceSort_ = 1 << 30;
        while (nsteps - done >= cycle)
        {
            GraphSlot* slot = graph_for_state(cycle);
            HIP_CHECK(hipGraphLaunch(slot->exec, stream_));
            done += cycle;
            hostStep_ += cycle;
            rebuilds_ += (lazyOn_ && lazyK_ > 1) ? 1 : cycle;       // (a captured cycle: one sort interval, or two every-step steps)
            set_buf_state(slot->after);           // one sort per cycle (K > 1) and the coordinate-array swaps of the fused steps
            adopt_launch_notes(slot->notes);      // where the cycle's last step left its partial sums
            sinceSort_ = 1 << 30;                 // the next cycle (or the eager remainder) starts with a sort
        }
    }
    preIntegrated_ = false;
    for (; done < nsteps; done++) { stepsLeftInRun_ = nsteps - 1 - done; launch_step_kernels(); }
    stepsLeftInRun_ = 0;
}

// At the end of a call: what the next call will need and this one can provide without executing a step.  (1) The graph of a whole cycle for the buffer
// state the engine is in (capturing executes nothing).  (2) If the cells were rebuilt by the last step but no lists were recorded (the interval was 1
// until now: the engine had not measured the atoms' speed yet), the lists of the arrays as they stand - so that the next call opens with plain steps
// instead of rebuilding the same cells again.
// Records of k_pair_list's LDS tile for cells of at most maxT candidates.  LDS per wave bounds the kernel's occupancy and the hardware hands LDS out in
// blocks of 1 280 B (gfx950: 160 KiB in 128 blocks), so the tile is the largest one that costs no more blocks than the smallest one that would do
// (largest cell + 1 % + 1): C4's largest tile of 305 candidates gives 319 records in six blocks - 21 waves per CU by LDS - where "largest + 4 % + 6" took seven.
static int tile_records_for(const StepParams& P, const PairLists& base, int maxT, int candCap, bool roomy)
{
    constexpr size_t kLdsBlock = 1280;
    PairLists L = base;
    // (roomy: systems that run without the clean-up launch - a cell that outgrows the tile there costs a window of steps run again, and LDS per wave is not
    //  what bounds a step of a few thousand cells: + 6 %)
    const int need = std::max(4 * kWave, std::min(candCap, roomy ? maxT + maxT / 16 + 4 : maxT + maxT / 100 + 1));
    L.candLds = need;
    const size_t blocks = (pair_list_lds_bytes(P, L) + kLdsBlock - 1) / kLdsBlock;
    int best = need;
    for (int c = need + 1; c <= candCap; c++)
    {
        L.candLds = c;
        if ((pair_list_lds_bytes(P, L) + kLdsBlock - 1) / kLdsBlock > blocks) break;
        best = c;
    }
    return best;
}

void Engine::prepare_next_call()
{
    if (!lazyOn_ || !lazyMeasured_) return;
    if (listsOn_ && lazyK_ > 1 && !listsValid_ && sinceSort_ == 0 && pair_variant() == 2)
    {
        const PairLists pl = pair_lists();
        P_.cycleStep = 0;
        HIP_CHECK(hipMemsetAsync(dNoList_ + 2, 0, sizeof(int32_t), stream_));
        launch_build_lists(P_, dCellStart_, stream_, PairRange(), pl);
        check_launch("list building");
        sync();
        {
            int32_t nl[8];
            HIP_CHECK(hipMemcpy(nl, dNoList_, sizeof(nl), hipMemcpyDeviceToHost));
            unlistedState_ = (nl[2] == 0) ? 1 : 2;
            // the first lists of an engine's life: size the LDS tiles from what the cells really hold right away (adapt_sort_interval does the same at
            // every look) - the next call then walks them at full occupancy
            if (nl[5] == 0 && nl[6] == 0 && nl[3] > 0 && !(debug_ & 65536))
            {
                const int candLds = tile_records_for(P_, pl, nl[3], candCap_, nranks_ == 1 && capacity_ <= 2 * kFuseKickMaxAtoms);
                const int iterLds = std::max(2 * kListMinIter, std::min(iterCap_, (nl[4] + nl[4] / 8 + 2 + 7) & ~7));
                if (candLds < candLds_ || iterLds < iterLds_) { candLds_ = std::min(candLds_, candLds); iterLds_ = std::min(iterLds_, iterLds); destroy_graphs(); graphCycle_ = 0; }
            }
        }
        if (nranks_ > 1)
        {   // the plain steps' coordinate exchange needs to know where the boundary layers sit; with interval 1 nobody had asked (k_rank_gather left it ready)
            int32_t h[16];
            post_count_exchange();
            HIP_CHECK(hipStreamSynchronize(stream_));
            HIP_CHECK(hipMemcpy(h, dHaloInfo_, sizeof(h), hipMemcpyDeviceToHost));
            haloInfoPending_ = false;
            adopt_halo_info(h);
        }
        listsValid_ = true;
    }
    if (nranks_ > 1)
    {   // can every rank open the next call with plain steps?
        double cannot = (listsOn_ && listsValid_ && lazyK_ > 1 && pair_variant() == 2) ? 0.0 : 1.0;
        xch_->allreduce_sum(&cannot, 1, stream_);
        carryAgreed_ = cannot == 0.0;
    }
    if (can_graph()) (void)graph_for_state(graph_cycle());
}

// the largest step any atom made since the last look sizes the next calls' sort interval: K steps of that length use at most slack / 1.15 (Engine::lazyMargin_)
// (one GPU; a violation is handled exactly by the clean-up launch at the staging kernel's speed, so the margin is a performance choice: with half the
// slack C4 ran at K = 16, with two thirds at K = 24-28, no violation in 2 000 steps; every violation widens the margin for good - a system that heats up,
// like the Born-Mayer-Huggins melt B3, would otherwise run into one after the other) - half on slab ranks, whose repair (a window of steps run again) is dearer.
// Returns false when the slab ranks have found a violation: the caller takes every rank back to its snapshot
// A look needs the device's Counts and the list builder's report.  Two blocking copies behind a stream synchronisation were three round trips between host
// and device (~40 us of a short aztot_step call); queued into pinned memory BEFORE the synchronisation they ride on it (a small kernel writing the pinned block directly: measured, the same).
void Engine::look_sync()
{
    if (lazyOn_)
    {
        if (!hLook_) HIP_CHECK(hipHostMalloc(&hLook_, sizeof(Counts) + sizeof(int32_t) * 16, hipHostMallocDefault));
        HIP_CHECK(hipMemcpyAsync(hLook_, dCounts_, sizeof(Counts), hipMemcpyDeviceToHost, stream_));
        if (listsOn_ && dNoList_) HIP_CHECK(hipMemcpyAsync((char*)hLook_ + sizeof(Counts), dNoList_, sizeof(int32_t) * 16, hipMemcpyDeviceToHost, stream_));
    }
    sync();
}

// (called right behind look_sync: reads what it left in pinned memory)
bool Engine::adapt_sort_interval()
{
    Counts c;
    std::memcpy(&c, hLook_, sizeof(Counts));
    bool rebuildNeeded = false;                    // this rank's lists were re-allocated: the next step must rebuild (on every rank: rebuild steps carry the full exchange)
    if (listsOn_)
    {   // cells that keep no list are staged by the small clean-up launch: fine for a few, slow for many (stencils wider than one tile, cells of more than
        // 64 atoms) - then the plain steps go back to staging every cell
        int32_t nl[16];
        std::memcpy(nl, (const char*)hLook_ + sizeof(Counts), sizeof(nl));
        if ((debug_ & 2097152) && nl[1] > 0)
        {   // measurement aid: mean list length / tile size / atoms per cell over the cells recorded since the last look
            std::fprintf(stderr, "aztot: per cell: %.2f list iterations, %.1f candidates, %.2f atoms\n", (double)nl[8] / nl[1], (double)nl[9] / nl[1], (double)nl[10] / nl[1]);
            HIP_CHECK(hipMemset(dNoList_ + 8, 0, sizeof(int32_t) * 3));
        }
        if (nl[1] > 0) unlistedAtLook_ = nl[0] > 0;      // (nothing recorded since the last look: what that look found still stands)
        if (nl[1] > 0)
        {
            HIP_CHECK(hipMemsetAsync(dNoList_, 0, sizeof(int32_t) * 2, stream_));          // ([2] stays: it describes the lists in force; [3], [4] are all-time maxima)
            HIP_CHECK(hipMemsetAsync(dNoList_ + 5, 0, sizeof(int32_t) * 2, stream_));
            if (std::getenv("AZTOT_VERBOSE"))
                std::fprintf(stderr, "aztot: lists recorded since the last look: %d cells, %d of them without a list (%d: tile full, %d: list full); largest tile %d of %d (LDS %d), longest list %d of %d (LDS %d)\n",
                             nl[1], nl[0], nl[5], nl[6], nl[3], candCap_, candLds_, nl[4], iterCap_, iterLds_);
            // LDS per wave is what bounds the occupancy of k_pair_list: the tiles are sized from the largest cell ever recorded (+ 6 %), not from the
            // capacity of the arrays.  A cell that does not fit next time keeps no list for one interval (exact: the clean-up launch serves it) and is
            // counted; then the tiles grow again.
            int candLds = tile_records_for(P_, pair_lists(), nl[3], candCap_, nranks_ == 1 && capacity_ <= 2 * kFuseKickMaxAtoms);
            int iterLds = std::max(2 * kListMinIter, std::min(iterCap_, (nl[4] + nl[4] / 8 + 2 + 7) & ~7));
            if (nl[5] > 0) candLds = std::min(candCap_, std::max(candLds, candLds_ + 32));
            if (nl[6] > 0) iterLds = std::min(iterCap_, std::max(iterLds, iterLds_ + 8));
            if (debug_ & 65536) { candLds = candLds_; iterLds = iterLds_; }
            // more waves per cell where the cells turn out denser than the mean density promised (a droplet in a large box - case study 2: tiles of 1 487
            // candidates, lists of 210 iterations where the homogeneous estimate said 1 wave would do): decided from what the builder recorded
            {
                const double tileNow = (double)nl[3] * (P_.single_lj ? 24.0 : (pair_list_tab_mode(P_) ? 25.0 : 32.0));
                int wWant = listWaves_;
                if (tileNow > 13.0 * 1024 || nl[4] > 96) wWant = std::max(wWant, 2);
                if (tileNow > 26.0 * 1024 || nl[4] > 192) wWant = 4;
                const bool forced = opt_.waves_per_cell == 1 || opt_.waves_per_cell == 2 || opt_.waves_per_cell == 4;
                if (wWant > listWaves_ && !forced && !(debug_ & 65536))
                {
                    const int itersNew = std::max(2 * kListMinIter, ((int)(nl[4] * (double)listWaves_ / wWant * 1.3) + 8 + 7) & ~7);
                    const int candNew = std::max(candCap_, kListMinCand * wWant);
                    listWaves_ = wWant;
                    destroy_graphs(); graphCycle_ = 0;
                    regrow_lists(candNew, itersNew);
                    rebuildNeeded = true;
                    nl[1] = 0;                                  // (what was recorded describes lists that no longer exist: nothing more to decide now)
                }
            }
            if (nl[1] > 0)
            {
            const bool capFull = (nl[5] > 0 && candLds_ == candCap_) || (nl[6] > 0 && iterLds_ == iterCap_);
            // (one GPU: ANY cell that does not fit costs an engine that runs without the clean-up launch a window of steps run again, so the arrays grow at once
            //  while they still can; slab ranks and the last growth wait until it is more than a handful)
            if (capFull && !(debug_ & 65536) && (double)(nl[5] + nl[6]) > ((nranks_ == 1 && listGrowths_ < 3) ? 0.0 : 0.0005 * (double)nl[1]))
            {   // the arrays themselves are too small: larger ones if the limits allow (twice), and the next step rebuilds; else - cells of more than 64 atoms
                // never fit - the plain steps go back to staging once that is more than 2 % of the cells
                const int cand = nl[5] > 0 ? std::min(kListCandMax, (candCap_ * 3 / 2 + 63) & ~63) : candCap_;
                const int iters = nl[6] > 0 ? std::min(kListIterMax, (iterCap_ * 2 + 7) & ~7) : iterCap_;
                destroy_graphs(); graphCycle_ = 0;
                if (listGrowths_ < 3 && (cand > candCap_ || iters > iterCap_)) { listGrowths_++; regrow_lists(cand, iters); rebuildNeeded = true; }
            }
            else if (candLds != candLds_ || iterLds != iterLds_)
            {
                candLds_ = candLds; iterLds_ = iterLds;
                destroy_graphs(); graphCycle_ = 0;        // (launch parameters are baked into the graphs)
            }
            if (listsOn_ && (double)nl[0] > 0.02 * (double)nl[1] && !(debug_ & 65536) && nl[5] + nl[6] < nl[0] / 2)
            {   // mostly cells of more than 64 atoms: no list will ever hold them
                listsOn_ = false; listsValid_ = false; destroy_graphs(); graphCycle_ = 0;
            }
            }
        }
    }
    if (nranks_ > 1)
    {   // every rank must arrive at the same interval (sort steps carry the full exchange) and at the same verdict: one slot per rank + the flags
        double v[11] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        v[10] = rebuildNeeded ? 1.0 : 0.0;
        double mine;
        std::memcpy(&mine, &c.maxStep2, sizeof(mine));
        if (nranks_ <= 8) v[rank_] = mine; else v[0] = 0.0;
        v[8] = c.lazyViolatedEver ? 1.0 : 0.0;
        v[9] = nranks_ > 8 ? mine : 0.0;              // more than 8 ranks: sum of squares bounds the maximum from above (conservative)
        xch_->allreduce_sum(v, 11, stream_);
        rebuildNeeded = v[10] > 0.0;
        double mx = v[9];
        for (int k = 0; k < 8; k++) mx = std::max(mx, v[k]);
        std::memcpy(&c.maxStep2, &mx, sizeof(mx));
        c.lazyViolatedEver = v[8] > 0.0 ? 1 : 0;
        if (c.lazyViolatedEver)
        {
            const int32_t z = 0;
            HIP_CHECK(hipMemcpy(&dCounts_->lazyViolatedEver, &z, sizeof(z), hipMemcpyHostToDevice));
            if (lazyK_ != 1) { lazyK_ = 1; destroy_graphs(); graphCycle_ = 0; }
            lazyMeasured_ = false; lazyWindow_ = 8; sinceLook_ = 0; lazyViolations_++;
            lazyMargin_ = std::min(4.0, lazyMargin_ * (4.0 / 3.0));
            if (rebuildNeeded) { sinceSort_ = 1 << 30; listsValid_ = false; carryAgreed_ = false; }
            return false;             // every rank goes back to its snapshot (Engine::step)
        }
    }
    HIP_CHECK(hipMemsetAsync(&dCounts_->maxStep2, 0, sizeof(unsigned long long), stream_));      // (in stream order: no second round trip for the look)
    if (rebuildNeeded) { sinceSort_ = 1 << 30; listsValid_ = false; carryAgreed_ = false; }
    lazyMeasured_ = true;
    int K = lazyK_;
    // one GPU without the clean-up launch: a violation, or a cell that kept no list, means the steps since the last look were not exact
    const bool goBack = optimistic_ && (c.lazyViolatedEver != 0 || unlistedAtLook_);
    if (debug_ & 8192)
    {   // debug: fixed interval whatever the speeds (exercises the wider-stencil fallback); violations are only counted
        if (c.lazyViolatedEver)
        {
            const int32_t z = 0;
            HIP_CHECK(hipMemcpy(&dCounts_->lazyViolatedEver, &z, sizeof(z), hipMemcpyHostToDevice));
            lazyViolations_++;
        }
        K = lazyCap_;
    }
    else
    {
        const bool violated = c.lazyViolatedEver != 0;
        if (violated)
        {
            const int32_t z = 0;
            HIP_CHECK(hipMemcpy(&dCounts_->lazyViolatedEver, &z, sizeof(z), hipMemcpyHostToDevice));
            lazyViolations_++;
            lazyMargin_ = std::min(4.0, lazyMargin_ * (4.0 / 3.0));      // the speeds are growing: more room from now on (1.15 -> 1.5 -> 2.0 -> 2.7 -> 3.6 -> 4)
        }
        int fromSpeed = -1;
        if (c.maxStep2 != 0)
        {
            double ms2;
            std::memcpy(&ms2, &c.maxStep2, sizeof(ms2));
            const double len = std::sqrt(ms2);
            // (slab ranks repair a violation by running up to a window of steps again with the cells rebuilt every step - dearer than the wider stencil one GPU
            //  falls back on - so they keep a factor 2 in hand at least; round 2 kept 4 and could only report the violation)
            // ... and where the speeds are still growing (a lattice released from rest, a melt heating up: the longest step of this window against the last
            // one's) the margin grows by the same factor, at most 2: the steps ahead will be longer than the ones just seen.  (With the plain margin of 1.15
            // C3 - charges on a lattice at rest - ran into a violation in its first hundred steps; a first look has nothing to compare with and takes 1.3.)
            // (a ratio below 1.1 is the noise of a maximum over a few thousand atom-steps, not a trend)
            const double growth = lastLookLen_ > 0.0 ? (len > 1.1 * lastLookLen_ ? std::min(2.0, len / lastLookLen_) : 1.0) : 1.3 / 1.15;
            lastLookLen_ = len;
            const double raw = len > 0 ? lazySlack_ / ((nranks_ > 1 ? std::max(2.0, lazyMargin_) : lazyMargin_) * growth * len) : 1e9;
            if (std::getenv("AZTOT_VERBOSE"))
                std::fprintf(stderr, "aztot: longest step %.3e A, slack %.3e A, margin %.2f%s: interval up to %.1f steps\n", len, lazySlack_, lazyMargin_, violated ? ", violated" : "", raw);
            static const int allowed[] = {128, 112, 96, 80, 64, 56, 48, 40, 36, 32, 28, 24, 20, 16, 14, 12, 10, 8, 6, 5, 4, 3, 2, 1};
            fromSpeed = 1;
            // (engines that replay whole cycles as graphs stop at 64: what is left of a call behind its last whole cycle is launched kernel by kernel -
            //  C2 with 2 000-step calls: 0.0122 ms/step at K = 64, 0.0129 at 112)
            const int cap = can_graph() ? std::min(lazyCap_, 64) : lazyCap_;
            for (int a : allowed) if (a <= cap && (double)a <= raw) { fromSpeed = a; break; }
            // (the table keeps the number of different cycle lengths - captured graphs - small; where cycles are launched kernel by kernel any whole number will do:
            //  a thermalised 1 M-atom liquid sits between 14 and 16 and gets 15)
            if (!can_graph() && raw >= 1.0) fromSpeed = std::max(fromSpeed, std::min(cap, (int)raw));
        }
        // after a violation: half the interval, or what the speeds seen now allow if that is less (a melt that heats up outruns halving)
        if (violated) K = std::max(1, fromSpeed > 0 ? std::min(K / 2, fromSpeed) : K / 2);
        else if (fromSpeed > 0) K = fromSpeed;
    }
    if (K != lazyK_) { lazyK_ = K; destroy_graphs(); graphCycle_ = 0; }
    return !goBack;
}

void Engine::check_overflow()
{
    if (nranks_ <= 1 && !hasBonded_) return;
    Counts c;
    HIP_CHECK(hipMemcpy(&c, dCounts_, sizeof(Counts), hipMemcpyDeviceToHost));
    if (c.overflow) throw std::runtime_error("slab decomposition: a fixed-capacity halo/migration/atom buffer overflowed");
    if (c.bondedMissing)
        throw std::runtime_error("slab decomposition: a bond / angle partner was not resident on the rank that owns the atom "
                                 "(bonded terms must span less than the halo width)");
}

void Engine::get_stats(aztot_stats& out)
{
    settle();
    sync();
    DevStats s;
    HIP_CHECK(hipMemcpy(&s, dStats_, sizeof(DevStats), hipMemcpyDeviceToHost));
    // (the host decides from its own count of the steps which of them the equilibration schedule acts on: the two counts must never part)
    if (failed_.empty() && s.step != hostStep_) throw std::runtime_error("the host's step count (" + std::to_string(hostStep_) + ") and the device's (" + std::to_string(s.step) + ") differ");
    double v[24];
    v[0] = s.engKin; v[1] = s.engVdW; v[2] = s.engCoul; v[3] = s.engElecField; v[4] = s.engTemp;
    for (int k = 0; k < 6; k++) { v[5 + k] = s.mom[k]; v[11 + k] = (double)s.cross[k]; }
    v[17] = (double)s.dropped; v[18] = s.engBond; v[19] = s.engAngle;
    if (nranks_ > 1) xch_->allreduce_sum(v, 20, stream_);
    std::memset(&out, 0, sizeof(out));
    out.step = s.step;
    out.time = s.step * model_.tSt;
    out.engKin = v[0]; out.engVdW = v[1]; out.engCoul = v[2]; out.engElecField = v[3]; out.engTemp = v[4];
    out.engPot = out.engCoul + out.engVdW;
    out.engBond = v[18]; out.engAngle = v[19];
    out.engCoulRec = s.engCoulRec; out.engCoulConst = s.engCoulConst;      // global quantities: identical on every rank
    out.engTot = out.engElecField + out.engVdW + (out.engCoulConst + out.engCoulRec + out.engCoul) + out.engKin + out.engBond +
                 out.engAngle;                                             // calc_chars integrators.cpp:70-71
    out.temperature = 2.0 * out.engKin * model_.revDegFree * (1.0 / units::kB);     // integrators.cpp:67
    out.negMom[0] = v[5]; out.posMom[0] = v[6]; out.negMom[1] = v[7]; out.posMom[1] = v[8]; out.negMom[2] = v[9]; out.posMom[2] = v[10];
    out.negCross[0] = (int64_t)v[11]; out.posCross[0] = (int64_t)v[12]; out.negCross[1] = (int64_t)v[13];
    out.posCross[1] = (int64_t)v[14]; out.negCross[2] = (int64_t)v[15]; out.posCross[2] = (int64_t)v[16];
    out.pairs_dropped = (int64_t)v[17];
    out.n_cells = (int64_t)P_.nc[0] * P_.nc[1] * P_.nc[2];
    out.nose_chit = s.chit; out.nose_conint = s.conint;
    out.sort_interval = lazyOn_ ? lazyK_ : 1; out.sort_violations = lazyViolations_;
    out.pair_lists = (listsOn_ && lazyOn_ && lazyK_ > 1) ? 1 : 0;
    out.rebuilds = rebuilds_;
    out.skin = 2.0 * lazySlack_;
    out.cells_without_list = 0;
    if (listsOn_)
    {
        int32_t nl = 0;
        HIP_CHECK(hipMemcpy(&nl, dNoList_ + 2, sizeof(nl), hipMemcpyDeviceToHost));
        out.cells_without_list = nl;
    }
    // pressure from the wall momentum over the window since the previous evaluation (main.cpp:143-163)
    if (s.step - lastPresStep_ >= std::max(1, model_.stat))
    {
        const double dtw = (s.step - lastPresStep_) * model_.tSt;
        const double revS[6] = {1.0 / (model_.L[1] * model_.L[2]), 1.0 / (model_.L[1] * model_.L[2]), 1.0 / (model_.L[0] * model_.L[2]),
                                1.0 / (model_.L[0] * model_.L[2]), 1.0 / (model_.L[0] * model_.L[1]), 1.0 / (model_.L[0] * model_.L[1])};
        double p = 0.0;
        for (int k = 0; k < 6; k++) { p += 2.0 * units::pressure_factor * revS[k] * (v[5 + k] - lastMom_[k]) / dtw; lastMom_[k] = v[5 + k]; }
        pressure_ = p / 6.0;
        lastPresStep_ = s.step;
    }
    out.pressure = pressure_;
}

void Engine::species_crossings(int64_t* out, int cap)
{
    const int n = 6 * model_.nSpec();
    if (cap < n) throw std::runtime_error("species_crossings: output array too small");
    settle();
    sync();
    DevStats s;
    HIP_CHECK(hipMemcpy(&s, dStats_, sizeof(DevStats), hipMemcpyDeviceToHost));
    double v[kSpecCap * 6];
    for (int k = 0; k < n; k++) v[k] = (double)s.specCross[k];
    if (nranks_ > 1) xch_->allreduce_sum(v, n, stream_);
    for (int k = 0; k < n; k++) out[k] = (int64_t)(v[k] + 0.5);
}

// md_to_host (cuInit.cu:1212-1262).  The reference returns the arrays in cell-sorted order (SURVEY C-20);
// we put every atom back at its original index.  On several ranks each rank fills only the atoms it owns.
void Engine::md_to_host(aztot_state& out)
{
    settle();
    sync();
    Counts c;
    HIP_CHECK(hipMemcpy(&c, dCounts_, sizeof(Counts), hipMemcpyDeviceToHost));
    const int n = c.ownedEnd - c.ownedBegin;
    if (out.n_atoms < model_.nAt) throw std::runtime_error("md_to_host: output arrays too small");
    std::vector<int32_t> ids(std::max(n, 1));
    HIP_CHECK(hipMemcpy(ids.data(), cur().id + c.ownedBegin, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost));
    std::vector<double> tmp(std::max(n, 1));
    auto down = [&](double* dst, const double* src) {
        if (!dst) return;
        HIP_CHECK(hipMemcpy(tmp.data(), src + c.ownedBegin, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost));
        for (int k = 0; k < n; k++) dst[ids[k]] = tmp[k];
    };
    AtomArrays& A = cur();
    down(out.x, A.x); down(out.y, A.y); down(out.z, A.z); down(out.vx, A.vx); down(out.vy, A.vy); down(out.vz, A.vz);
    if (lazyOn_)
    {   // between two sorts of a lazy run coordinates are kept unwrapped on the device: hand out what put_periodic would have left (box.cpp:230-295)
        auto wrap = [](double x, double L) {
            const double invL = 1.0 / L;
            if (x < 0) x += ((int)(-x * invL) + 1) * L;
            else if (x > L) x -= ((int)(x * invL)) * L;
            return (x >= L) ? 0.0 : x;
        };
        double* arr[3] = {out.x, out.y, out.z};
        for (int k = 0; k < 3; k++)
            if (arr[k])
                for (int q = 0; q < n; q++) arr[k][ids[q]] = wrap(arr[k][ids[q]], model_.L[k]);
    }
    down(out.fx, A.fx); down(out.fy, A.fy); down(out.fz, A.fz); down(out.U, A.U); down(out.radius, A.rad);
    if (out.types)
    {
        std::vector<int32_t> ty(std::max(n, 1));
        HIP_CHECK(hipMemcpy(ty.data(), A.type + c.ownedBegin, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost));
        for (int k = 0; k < n; k++) out.types[ids[k]] = ty[k];
    }
    out.n_atoms = n;   // number of atoms this rank wrote
}

// restart support: step number + thermostat scalars (see aztot_clock)
void Engine::get_clock(aztot_clock& out)
{
    settle();
    sync();
    DevStats s;
    HIP_CHECK(hipMemcpy(&s, dStats_, sizeof(DevStats), hipMemcpyDeviceToHost));
    out.step = s.step; out.nose_chit = s.chit; out.nose_conint = s.conint; out.eng_kin = s.ekSim;
}

void Engine::set_clock(const aztot_clock& in)
{
    settle();
    sync();
    snap_.valid = false;
    DevStats s;
    HIP_CHECK(hipMemcpy(&s, dStats_, sizeof(DevStats), hipMemcpyDeviceToHost));
    s.step = in.step; s.stepAtSort = in.step; s.chit = in.nose_chit; s.conint = in.nose_conint; s.ekSim = in.eng_kin; s.engKin = in.eng_kin;
    s.pendingKick = 0;
    HIP_CHECK(hipMemcpy(dStats_, &s, sizeof(DevStats), hipMemcpyHostToDevice));
    lastPresStep_ = in.step;
    hostStep_ = in.step;
    sinceSort_ = 1 << 30; listsValid_ = false; carryAgreed_ = false;       // the next call rebuilds the cells
    destroy_graphs();
}

// read-back of the sorted cell list (cudaMD::firstAtomInCell + the id of the atom in every slot)
int Engine::cell_table(int32_t dims[3], int32_t* cellStart, int capCells, int32_t* atomId, int capAtoms)
{
    settle();
    sync();
    dims[0] = P_.ncxLocal; dims[1] = P_.nc[1]; dims[2] = P_.nc[2];
    const int nCell = P_.nCellLocal;
    if (cellStart)
    {
        if (capCells < nCell + 1) throw std::runtime_error("cell_table: cell_start needs n_cells + 1 entries");
        HIP_CHECK(hipMemcpy(cellStart, dCellStart_, sizeof(int32_t) * (size_t)(nCell + 1), hipMemcpyDeviceToHost));
    }
    if (atomId)
    {
        Counts c;
        HIP_CHECK(hipMemcpy(&c, dCounts_, sizeof(Counts), hipMemcpyDeviceToHost));
        if (capAtoms < c.nTotal) throw std::runtime_error("cell_table: atom_id array too small");
        HIP_CHECK(hipMemcpy(atomId, cur().id, sizeof(int32_t) * (size_t)c.nTotal, hipMemcpyDeviceToHost));
    }
    return nCell;
}

// overwrite per-atom state (indexed by ORIGINAL atom id); used for exact restarts and stage-wise tests
void Engine::set_state(const aztot_state& in)
{
    if (!failed_.empty()) throw std::runtime_error("this handle failed in an earlier call: " + failed_);
    settle();
    sync();
    snap_.valid = false;
    sinceSort_ = 1 << 30; listsValid_ = false; carryAgreed_ = false;       // the next call rebuilds the cells
    if (in.vx || in.vy || in.vz || in.fx || in.fy || in.fz)
    {   // new velocities / forces: the interval measured on the old ones says nothing about them - every step rebuilds until the first look
        if (lazyK_ != 1 && !(debug_ & 8192)) { lazyK_ = 1; destroy_graphs(); graphCycle_ = 0; }
        lazyMeasured_ = false; lazyWindow_ = 8; sinceLook_ = 0; lastLookLen_ = 0.0;
    }
    Counts c;
    HIP_CHECK(hipMemcpy(&c, dCounts_, sizeof(Counts), hipMemcpyDeviceToHost));
    const int n = c.ownedEnd - c.ownedBegin;
    std::vector<int32_t> ids(std::max(n, 1));
    HIP_CHECK(hipMemcpy(ids.data(), cur().id + c.ownedBegin, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost));
    std::vector<double> tmp(std::max(n, 1));
    auto up = [&](const double* src, double* dst) {
        if (!src) return;
        for (int k = 0; k < n; k++) tmp[k] = src[ids[k]];
        HIP_CHECK(hipMemcpy(dst + c.ownedBegin, tmp.data(), sizeof(double) * (size_t)n, hipMemcpyHostToDevice));
    };
    AtomArrays& A = cur();
    up(in.x, A.x); up(in.y, A.y); up(in.z, A.z); up(in.vx, A.vx); up(in.vy, A.vy); up(in.vz, A.vz);
    up(in.fx, A.fx); up(in.fy, A.fy); up(in.fz, A.fz); up(in.U, A.U); up(in.radius, A.rad);
    if ((in.U || in.radius) && !thermoTouched_)
    {   // from now on the sort carries the per-atom thermostat state along; a captured graph has the old carry mode baked in
        thermoTouched_ = true;
        destroy_graphs();
    }
}

}  // namespace aztot
