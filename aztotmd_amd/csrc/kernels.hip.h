// Hand-written HIP kernels (gfx950 / CDNA4, wave64) of the azTotMD per-step hot path, fp64.
//
//  reference kernel (file:line)                          ours
//  ---------------------------------------------------   ----------------------------------------------
//  clear_clist            cuMDfunc.cu:693                folded into k_integrate2 (zeroes the histogram)
//  verlet_1stage+count_cell cuMDfunc.cu:333, cuSort.cu:114  k_integrate1_bin
//  calc_firstAtomInCell   cuSort.cu:130 (1 thread)       k_scan_totals + k_scan_apply (chunked, parallel)
//  sort_atoms+refresh_arrays cuSort.cu:145,74            k_place + k_rank_gather (deterministic order)
//  cell_list5a + cell_list4b_noshared + pair_1           k_pair_atom / k_pair_tile (no atomics, no tables)
//     cuPairs.cu:2266,1474,117
//  verlet_2stage + zero_engKin cuMDfunc.cu:521, cuTemp.cu:165   k_integrate2
//  temp_scale/after_tscale cuTemp.cu:77,109              k_reduce_kin (decides the factor) + k_post
//  tstat_radi9            cuTemp.cu:689                  k_post
//  reset_quantities + calc_quantities cuMDfunc.cu:270, main.cu:121   k_finalize
//
// No MFMA anywhere: there is no dense contraction on this path.
#pragma once
#include <hip/hip_runtime.h>

#include "device_md.h"
#include "msg_layout.h"
#include "rng.h"

namespace aztot {

constexpr int kBlock = 256;
constexpr int kWave = 64;

// ------------------------------------------------------------------------------------------------
// reductions: wave shuffle -> LDS -> thread 0, fixed order (deterministic for a fixed launch shape)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, kWave);
    return v;
}

// sum over the workgroup; result valid in thread 0. `scratch` holds blockDim/64 doubles.
__device__ __forceinline__ double block_sum(double v, double* scratch)
{
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) scratch[w] = v;
    __syncthreads();
    double r = 0.0;
    if (threadIdx.x == 0)
        for (int k = 0; k < (int)(blockDim.x >> 6); k++) r += scratch[k];
    return r;
}

// maximum over the workgroup; result valid in thread 0
__device__ __forceinline__ double block_max(double v, double* scratch)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_down(v, o, kWave));
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) scratch[w] = v;
    __syncthreads();
    double r = 0.0;
    if (threadIdx.x == 0)
        for (int k = 0; k < (int)(blockDim.x >> 6); k++) r = fmax(r, scratch[k]);
    return r;
}

__device__ __forceinline__ void put_partial(double* partials, int maxBlocks, int slot, double v)
{
    partials[(size_t)slot * maxBlocks + blockIdx.x] = v;
}
// running sums (wall momenta, crossing counts, dropped pairs) pile up in the block's own entry until k_collect reads and
// clears them: the reduction runs only when the host looks at the statistics, not every step
__device__ __forceinline__ void add_partial(double* partials, int maxBlocks, int slot, double v)
{
    partials[(size_t)slot * maxBlocks + blockIdx.x] += v;
}

// Wall momenta and crossing counts of a workgroup into its partial-sum slots (put_periodic's counters, box.cpp:230-295 / cuMDfunc.cu:72-106).  A handful of
// atoms per step cross a wall in a box of a million, but the workgroup that holds one used to run twelve block reductions (24 barriers) for it - every wave of
// it -, and an integrate kernel is as slow as its slowest workgroup: k_integrate_plain2 took 31.7 us on the thermalised 1 M-atom box against 27.1 on the
// lattice at rest.  Now: one barrier for everybody, twelve wave reductions in the waves that have a crossing, one more barrier and twelve short sums in the
// workgroups that have one.  Same order of summation as block_sum (inside a wave, then the waves in order): the sums are bit-identical.
__device__ __forceinline__ void book_wall_crossings(const double (&mom)[6], const double (&cross)[6], int anyCross, double* __restrict__ partials, int maxBlocks)
{
    __shared__ double wall[kBlock / kWave][12];
    if (!__syncthreads_or(anyCross)) return;
    const int lane = threadIdx.x & (kWave - 1), w = threadIdx.x >> 6;
    const bool mine = __any(anyCross) != 0;                            // wave-uniform
#pragma unroll
    for (int k = 0; k < 6; k++)
    {
        const double a = mine ? wave_sum(mom[k]) : 0.0, b = mine ? wave_sum(cross[k]) : 0.0;
        if (lane == 0) { wall[w][k] = a; wall[w][6 + k] = b; }
    }
    __syncthreads();
    if (threadIdx.x < 6)
    {
        const int k = threadIdx.x;
        double a = 0.0, b = 0.0;
        for (int q = 0; q < (int)(blockDim.x >> 6); q++) { a += wall[q][k]; b += wall[q][6 + k]; }
        if (b != 0.0) { add_partial(partials, maxBlocks, PS_MOM_XN + k, a); add_partial(partials, maxBlocks, PS_CNT_XN + k, b); }
    }
}
__device__ __forceinline__ bool slot_accumulates(int slot) { return (slot >= PS_MOM_XN && slot <= PS_CNT_ZP) || slot == PS_DROPPED; }

// ------------------------------------------------------------------------------------------------
// geometry helpers
// ------------------------------------------------------------------------------------------------
// put_periodic: serial formula (box.cpp:230-295) + the GPU path's x >= L -> 0 safety (cuMDfunc.cu:64-69)
__device__ __forceinline__ int wrap_coord(double& x, double L, double invL)
{
    int crossed = 0;
    if (x < 0) { x += ((int)(-x * invL) + 1) * L; crossed = -1; }
    else if (x > L) { x -= ((int)(x * invL)) * L; crossed = 1; }
    if (x >= L) x = 0.0;
    return crossed;
}

__device__ __forceinline__ int cell_coord(double x, double icsz, int n)
{   // count_cell: cuSort.cu:119 - floor(x * cRevSize) in double; made robust for coordinates outside [0, L)
    int c = (int)floor(x * icsz);
    c %= n;
    if (c < 0) c += n;
    return c;
}

__device__ __forceinline__ int local_cell(const StepParams& P, double x, double y, double z, int* layer = nullptr)
{
    int gx = cell_coord(x, P.icsz[0], P.nc[0]);
    int cy = cell_coord(y, P.icsz[1], P.nc[1]);
    int cz = cell_coord(z, P.icsz[2], P.nc[2]);
    int lx = gx - P.cx0;                 // window of x-layers held by this rank (periodic unwrap)
    if (lx >= P.nc[0]) lx -= P.nc[0];
    if (lx < 0) lx += P.nc[0];
    if (layer) *layer = lx;
    return (lx * P.nc[1] + cy) * P.nc[2] + cz;
}

__device__ __forceinline__ void min_image(double& d, double L, double half)
{   // delta_periodic: box.cpp:180-207
    if (d > half) d -= L;
    else if (d < -half) d += L;
}

// ------------------------------------------------------------------------------------------------
// fast fp64 building blocks of the Coulomb terms (full double accuracy, a fraction of ocml's instruction count)
// ------------------------------------------------------------------------------------------------
// 1/x: v_rcp_f64 (about 2^-24 relative) refined by two Newton steps -> error within ~1 ulp; 5 instructions instead
// of the 11 of the IEEE division sequence.  x is a squared distance inside the cut-off: normal, positive.
__device__ __forceinline__ double fast_rcp(double x)
{
    double y = __builtin_amdgcn_rcp(x);
    const double e = fma(-x, y, 1.0);            // one third-order step y (1 + e + e^2): error e^3 ~ 2^-72, three FMAs instead of the
    return fma(y, fma(e, e, e), y);              // four of two Newton steps
}

// 1/sqrt(x): v_rsq_f64 refined (below); 6 instructions for what gives r = x y, 1/r = y and
// 1/r^2 = y y, instead of the reciprocal above plus ocml's range-scaled sqrt.  x is a squared distance: normal, positive.
__device__ __forceinline__ double fast_rsqrt(double x)
{
    // one third-order step y (1 + e/2 + 3 e^2/8), e = 1 - x y^2: error (5/16) e^3 ~ 2^-75 from v_rsq_f64's 2^-26; five instructions behind the
    // seed where two Newton steps took eight
    const double y = __builtin_amdgcn_rsq(x);
    const double e = fma(-(x * y), y, 1.0);
    return fma(y, fma(0.375, e, 0.5) * e, y);
}

// erfc(x) for 0 <= x <= 4 given ex = exp(-x*x), which the caller needs anyway (Fennell / Ewald force term):
// erfc(x) = ex * p(t), t = 3u - 2, u = 1/(1 + x/2), p = our own degree-15 fit of erfcx (tools/fit_erfcx.py; max relative error
// 7e-14 against scipy on [0, 4]; degree 16 until round 4: 8.5e-15 for one more FMA per visit and two more scalar registers in a loop
// that had run out of them).  23 instructions instead of ocml's 140-instruction erfc plus a second exp; the host selects
// this kernel only when alpha * rc <= 4.  tests/test_gpu_parity.py checks the result against the oracle's libm erfc.
// Coefficients live in constant memory so that they reach the polynomial chains through scalar registers: as 64-bit literals
// every one of them costs a v_mov_b64 per use inside the pair loop (57 of the 199 VALU instructions of the Coulomb body).
__constant__ double kCoulCoef[32] = {
    // [0..15] erfcx fit, highest degree first
    -6.42984581155711645e-10, -1.37711557461844240e-09, 1.54088012265774375e-08, -2.26846941341576436e-08, -1.28510332527685291e-07,
    7.29764795089034142e-07, -4.70995058463941374e-07, -9.92822323014682285e-06, 3.46017759998636334e-05, 1.00396835324338853e-04,
    -8.61280658405786780e-04, -1.60202365845927793e-03, 2.25095126335906476e-02, 1.42427002000104913e-01, 4.09818022175859442e-01,
    4.27583576155797507e-01,
    // [16..18] log2(e), ln2 high part (32 trailing zero bits), ln2 low part
    1.44269504088896338700e+00, 6.93147180369123816490e-01, 1.90821492927058770002e-10,
    // [19..27] q(r) of exp(r) = 1 + r (1 + r q(r)) on |r| <= ln2/2, highest degree first (degree 8 fit of (e^r - 1 - r) / r^2: 1.2e-15 relative;
    // the Taylor series to r^12 it replaced: 1.7e-16 for two more FMAs)
    2.76223814036453055e-07, 2.76251616785559953e-06, 2.48015167924052650e-05, 1.98412086507197805e-04, 1.38888889198207112e-03,
    8.33333335376915986e-03, 4.16666666666196950e-02, 1.66666666666482416e-01, 5.00000000000000222e-01,
    0.0, 0.0, 0.0, 0.0};

// exp(y) for -700 < y <= 0: n = rint(y log2 e), r = y - n ln2 (two-part), 1 + r (1 + r q(r)), scaled by 2^n.  ~18 instructions.
template <bool CLAMP = true>
__device__ __forceinline__ double exp_nonpos(double y)
{
    if (CLAMP) y = fmax(y, -700.0);                // masked-out pairs arrive with y = -1e299: keep the reduction finite (result ~1e-304).  CLAMP = false: the
                                                   // caller guarantees y > -700 (the list kernel's Coulomb bodies: pairs outside the cut-off never get here and
                                                   // alpha r <= 4)
    const double n = rint(y * kCoulCoef[16]);
    double r = fma(-n, kCoulCoef[17], y);
    r = fma(-n, kCoulCoef[18], r);
    double p = kCoulCoef[19];
#pragma unroll
    for (int k = 20; k <= 27; k++) p = fma(p, r, kCoulCoef[k]);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return ldexp(p, (int)n);
}

__device__ __forceinline__ double erfc_given_exp(double x, double ex)
{
    // (1 / (1 + x/2) by ONE Newton step behind v_rcp_f64: 2^-52 relative - t feeds a polynomial whose value moves by about as much -, one FMA less than fast_rcp)
    const double a = fma(0.5, x, 1.0);
    double u = __builtin_amdgcn_rcp(a);
    u = fma(u, fma(-a, u, 1.0), u);
    const double t = fma(3.0, u, -2.0);
    double p = kCoulCoef[0];
#pragma unroll
    for (int k = 1; k <= 15; k++) p = fma(p, t, kCoulCoef[k]);
    return ex * p;
}

// ------------------------------------------------------------------------------------------------
// pair functions: f = -(1/r) dU/dr, energy added to eV / eC.  Operation order follows the serial
// reference (vdw.cpp:16-157, elec.cpp:415-444); elin/einv/surk follow cuVdW.cu:162-257 in fp64.
// ------------------------------------------------------------------------------------------------
// Generic (any potential mix) bodies.  Divisions, square roots and exponentials go through the building blocks above (reciprocal
// / reciprocal square root with Newton steps, range-reduced Taylor exp): same values to ~1 ulp at a third of ocml's instructions.
__device__ __forceinline__ double vdw_force(const DevPot& v, double r2, double& r, double radi, double radj, double& eng)
{
    const double ir = fast_rsqrt(r2), r2i = ir * ir;
    r = r2 * ir;
    switch (v.type)
    {
    case 1:
    {   // fer_lj vdw.cpp:16-26
        double sr2 = v.p1 * r2i;
        double sr6 = sr2 * sr2 * sr2;
        eng += v.p0 * sr6 * (sr6 - 1.0);
        return v.p2 * r2i * sr6 * (2.0 * sr6 - 1.0);
    }
    case 2:
    {   // fer_buckingham vdw.cpp:60-70
        double r4i = r2i * r2i;
        double rho_i = fast_rcp(v.p1);
        double ex = v.p0 * exp_nonpos(-r * rho_i);
        eng += ex - v.p2 * r4i * r2i;
        return ex * ir * rho_i - 6.0 * v.p2 * r4i * r4i;
    }
    case 3:
    {   // fer_746 vdw.cpp:144-157
        double r4i = r2i * r2i;
        eng += r4i * (v.p0 * r2i * ir - v.p1 - v.p2 * r2i);
        return r4i * r2i * (7.0 * v.p0 * r2i * ir - 4.0 * v.p1 - 6.0 * v.p2 * r2i);
    }
    case 4:
    {   // fer_bhm vdw.cpp:102-112
        double r4i = r2i * r2i;
        double ex = v.p0 * exp_nonpos(v.p1 * (v.p2 - r));
        eng += ex - v.p3 * r4i * r2i - v.p4 * r4i * r4i;
        return v.p1 * ex * ir - 6.0 * v.p3 * r4i * r4i - 8.0 * v.p4 * r4i * r4i * r2i;
    }
    case 5:
    {   // cu_fer_elin cuVdW.cu:162-171
        double rho_i = fast_rcp(v.p1);
        double ex = v.p0 * exp_nonpos(-r * rho_i);
        eng += ex + v.p2 * r;
        return ex * ir * rho_i - v.p2 * ir;
    }
    case 6:
    {   // cu_fer_einv cuVdW.cu:200-208
        double rho_i = fast_rcp(v.p1);
        double ex = v.p0 * exp_nonpos(-r * rho_i);
        eng += ex - v.p2 * ir;
        return ex * ir * rho_i - v.p2 * ir * r2i;
    }
    case 7:
    {   // surk_pot cuVdW.cu:236-257
        double c2ir_sum = v.p1 * fast_rcp(v.p2 * radi + v.p3 * radj);
        double r_prod = radi * radj;
        double C1ab2 = r_prod * r_prod * v.p0;
        double ir6 = r2i * r2i * r2i;
        eng += r_prod * ir6 * (C1ab2 * ir - c2ir_sum);
        return r_prod * ir6 * r2i * (7.0 * C1ab2 * ir - 6.0 * c2ir_sum);
    }
    }
    return 0.0;
}

__device__ __forceinline__ double coul_force(const StepParams& P, double qq, double r2, double& eng)
{   // r is passed by value in the serial reference (elec.h:16), so the VdW part never sees it
    const double kqq = qq * P.fcoul;
    const double ir = fast_rsqrt(r2), r2i = ir * ir, r = r2 * ir;
    if (P.elec_type == 1)
    {   // direct_coul elec.cpp:415-428
        eng += kqq * ir;
        return kqq * ir * r2i;
    }
    // erfc / exp: the shared-exp erfcx fit when the whole range alpha r <= alpha rReal <= 4 is inside its domain (wave-uniform
    // choice), libm otherwise
    const bool fit = P.alpha * P.rReal <= 4.0;
    const double ar = P.alpha * r;
    const double ex = fit ? exp_nonpos(-ar * ar) : exp(-ar * ar);
    const double erfcar = fit ? erfc_given_exp(ar, ex) : erfc(ar);
    if (P.elec_type == 3)
    {   // fennel elec.cpp:430-444
        eng += kqq * (erfcar * ir - P.el_scale + P.el_scale2 * (r - P.rReal));
        return kqq * ir * ((erfcar * r2i + P.daipi2 * ex * ir) - P.el_scale2);
    }
    // coul_iter elec.cpp:344-369 (real-space Ewald term)
    eng += kqq * erfcar * ir;
    return kqq * ir * r2i * (erfcar + P.daipi2 * r * ex);
}

// one candidate pair, seen from atom i (pair_inter integrators.cpp:139-185).  Every unordered pair is
// visited from both ends, so each visit books half of the pair energy.
struct PairAcc { double fx, fy, fz, eV, eC, dropped; };

__device__ __forceinline__ void pair_visit(const StepParams& P, const SpecTable& S, const DevPot* __restrict__ pots,
                                           double dx, double dy, double dz, double r2, int ti, int tj, double radi, double radj,
                                           PairAcc& a)
{
    double r = 0.0, f = 0.0, eV = 0.0, eC = 0.0;
    if (P.elec_type != 0 && S.charged[ti] && S.charged[tj]) f += coul_force(P, S.charge[ti] * S.charge[tj], r2, eC);
    const DevPot v = pots[ti * P.nSpec + tj];
    if (v.type != 0 && r2 <= v.r2cut) f += vdw_force(v, r2, r, radi, radj, eV);
    a.eV += 0.5 * eV; a.eC += 0.5 * eC;
    if (f * f > 1e10) { a.dropped += 0.5; return; }          // integrators.cpp:170-174: energy booked, force dropped
    a.fx += f * dx; a.fy += f * dy; a.fz += f * dz;
}

// ------------------------------------------------------------------------------------------------
// K3: first half-kick + drift + wrap + wall counters + cell histogram
//     (verlet_1stage cuMDfunc.cu:333-519 ; serial integrate1_clst integrators.cpp:331-376)
// ------------------------------------------------------------------------------------------------
// index bookkeeping that lives on the device so that no step needs a host round trip
struct Counts
{
    int32_t ownedBegin, ownedEnd;   // range of the sorted arrays this rank integrates
    int32_t nTotal;                 // resident atoms (owned + ghosts) after the last sort
    int32_t srcBegin, srcEnd;       // pre-sort source range: old owned range + atoms appended by k_unpack
    int32_t nRecv;                  // atoms appended by k_unpack in the step in flight
    int32_t overflow;               // sticky: a fixed-capacity buffer was too small
    int32_t bondedMissing;          // sticky: a bond / angle partner was not resident on this rank (k_bonded)
    int32_t lazyViolated;           // lazy re-sort: 0, or the first step since the last rebuild (StepParams::cycleStep) whose positions have an atom outside the
                                    // slack of its cell (set by whoever integrates - k_integrate1_bin on plain steps, or the pair kernel's epilogue one step
                                    // ahead -, cleared on rebuild steps): from that step on the pair kernels widen their stencil by one cell until the next rebuild
    int32_t lazyViolatedEver;       // sticky copy for the host, which then shortens the sort interval
    unsigned long long maxStep2;    // bit pattern of the largest |v dt|^2 of any atom since the host last looked (non-negative doubles order like integers)
    // Displacement bound of the lazy re-sort (k_integrate1_bin<2>): on plain step s no atom can be farther from where it was at the last rebuild than
    // (s - 1) x the longest step made since + its own step, so while that stays inside the slack the per-atom check against the reference position - 24 B
    // per atom and step - is not needed.  cycMaxRun only ever grows between two rebuilds (a value read while a launch is still raising it is still an upper
    // bound of everything earlier); a rebuild lets it decay by 10 % instead of clearing it, so that only the few workgroups with a new record issue an atomic.
    unsigned long long cycMaxRun;   // bit pattern of (an upper bound of) the largest |v dt|^2 since the last rebuild
};

// has an atom left its cell's slack as of the step this launch belongs to?
__device__ __forceinline__ bool slack_violated(const StepParams& P, const Counts* c)
{
    const int v = c->lazyViolated;
    return v != 0 && v <= P.cycleStep;
}

// image index of a coordinate with exactly the case distinction of put_periodic / wrap_coord (box.cpp:243-252: x in [0, L] is image 0)
__device__ __forceinline__ int image_of(double x, double L, double invL)
{
    return (x < 0) ? -((int)(-x * invL) + 1) : ((x > L) ? (int)(x * invL) : 0);
}

__device__ __forceinline__ int wave_append(bool flag, int32_t* counter)
{   // position of this lane's element in a shared output list (one atomic per wave), -1 if the lane has nothing to append
    const unsigned long long mask = __ballot(flag);
    if (mask == 0ULL) return -1;
    const int lane = threadIdx.x & 63;
    const int leader = __ffsll((long long)mask) - 1;
    int base = 0;
    if (lane == leader) base = atomicAdd(counter, __popcll(mask));
    base = __shfl(base, leader, kWave);
    const unsigned long long below = mask & ((1ULL << lane) - 1ULL);
    return flag ? base + __popcll(below) : -1;
}

// In slab mode the same kernel also packs this rank's message to each x-neighbour (slab.hip.h describes the protocol):
// emigrants (full state) and the atoms of the hw boundary layers (position, type, id, radius), appended with one atomic per
// wave and category.
// STEPMODE 0: bin only (aztot_forces: wraps what a lazy run left unwrapped, counts nothing); 1: integrate + wrap + bin (a step that re-sorts:
// every step unless the lazy re-sort is on); 2: integrate only (a plain step of the lazy re-sort: coordinates stay UNWRAPPED until the next sort,
// wall crossings are still counted in the step they happen, and the displacement since the last sort is checked against the cells' slack)
template <int STEPMODE>
__global__ __launch_bounds__(kBlock) void k_integrate1_bin(StepParams P, SpecTable S, AtomArrays A, Counts* __restrict__ cnt,
                                                           int32_t* __restrict__ cellOf, int32_t* __restrict__ slotOf,
                                                           int32_t* __restrict__ cellCount, double* __restrict__ partials, int maxBlocks,
                                                           MsgLayout lay, char* __restrict__ sendLeft, char* __restrict__ sendRight,
                                                           DevStats* __restrict__ st, RefPos R0)
{
    constexpr bool INTEGRATE = STEPMODE != 0;
    constexpr bool BIN = STEPMODE != 2;
    __shared__ double scratch[kBlock / kWave];
    const int begin = cnt->ownedBegin, end = cnt->ownedEnd;
    const int i = begin + blockIdx.x * kBlock + threadIdx.x;
    const bool pendingKick = INTEGRATE && st->pendingKick != 0;   // written only by kernels that run between two launches of this one
    if (INTEGRATE && blockIdx.x == 0 && threadIdx.x == 0) { st->step += 1; if (STEPMODE == 1) st->stepAtSort = st->step; }   // the step in flight gets its 1-based number (main.cpp:92);
                                                                           // read only by the thermostat kernels at the end of the step
    double eField = 0.0, mom[6] = {0, 0, 0, 0, 0, 0}, cross[6] = {0, 0, 0, 0, 0, 0}, stepLen2 = 0.0;
    int anyCross = 0, myCell = 0, myLayer = 0, violated = 0;
    if (STEPMODE == 1 && blockIdx.x == 0 && threadIdx.x == 0)
    {   // the cells are rebuilt in this step: nobody has moved since
        cnt->lazyViolated = 0;
        cnt->cycMaxRun = (unsigned long long)__double_as_longlong(0.81 * __longlong_as_double((long long)cnt->cycMaxRun));
    }
    // plain step s: what the atoms can have moved by before this step
    double roomLeft = -1.0;                                          // < 0: check every atom against its reference position
    if (STEPMODE == 2 && P.pad2)
        roomLeft = sqrt(P.lazySlack2) - (double)(P.cycleStep - 1) * sqrt(__longlong_as_double((long long)__hip_atomic_load(&cnt->cycMaxRun, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)));
    if (i < end)
    {
        const int t = A.type[i];
        double x = A.x[i], y = A.y[i], z = A.z[i];
        if (STEPMODE == 0 && P.lazySlack2 > 0.0)
        {   // aztot_forces between two sorts of a lazy run: wrap first (crossings were counted when they happened)
            wrap_coord(x, P.L[0], P.invL[0]); wrap_coord(y, P.L[1], P.invL[1]); wrap_coord(z, P.L[2], P.invL[2]);
            A.x[i] = x; A.y[i] = y; A.z[i] = z;
        }
        if (INTEGRATE)
        {
            const double rM = S.rMhdt[t], m = S.mass[t];
            double vx = A.vx[i], vy = A.vy[i], vz = A.vz[i];
            if (P.tstat == 1) { const double sc = st->vscaleBegin; vx *= sc; vy *= sc; vz *= sc; }   // tstat_nose at integrators.cpp:305-306
            const double fxo = A.fx[i], fyo = A.fy[i], fzo = A.fz[i];
            if (pendingKick) { vx += rM * fxo; vy += rM * fyo; vz += rM * fzo; }   // integrate2 of the previous step, deferred (same f, same order)
            vx += rM * fxo;
            vy += rM * fyo;
            vz += rM * fzo;
            // a wall is crossed when the periodic image index changes (with every-step sorting the old index is 0 and this is put_periodic's
            // own test); the coordinate itself is wrapped only on steps that re-sort
            const int ix0 = image_of(x, P.L[0], P.invL[0]), iy0 = image_of(y, P.L[1], P.invL[1]), iz0 = image_of(z, P.L[2], P.invL[2]);
            double dx = 0.0, dy = 0.0, dz = 0.0;
            if (!S.frozen[t]) { dx = vx * P.dt; dy = vy * P.dt; dz = vz * P.dt; x += dx; y += dy; z += dz; }
            stepLen2 = dx * dx + dy * dy + dz * dz;
            int c;
            c = image_of(x, P.L[0], P.invL[0]) - ix0;
            if (c < 0) { mom[0] = m * (-vx); cross[0] = 1; anyCross = 1; } else if (c > 0) { mom[1] = m * vx; cross[1] = 1; anyCross = 1; }
            c = image_of(y, P.L[1], P.invL[1]) - iy0;
            if (c < 0) { mom[2] = m * (-vy); cross[2] = 1; anyCross = 1; } else if (c > 0) { mom[3] = m * vy; cross[3] = 1; anyCross = 1; }
            c = image_of(z, P.L[2], P.invL[2]) - iz0;
            if (c < 0) { mom[4] = m * (-vz); cross[4] = 1; anyCross = 1; } else if (c > 0) { mom[5] = m * vz; cross[5] = 1; anyCross = 1; }
            if (BIN) { wrap_coord(x, P.L[0], P.invL[0]); wrap_coord(y, P.L[1], P.invL[1]); wrap_coord(z, P.L[2], P.invL[2]); }
            else
            {   // plain step: has the atom left the slack of the cell it was sorted into?  Certainly not while the longest steps since the rebuild plus
                // its own add up to less than the slack; otherwise look at where it was
                if (!(roomLeft > 0.0 && stepLen2 < roomLeft * roomLeft))
                {
                    const double ex = x - R0.x[i], ey = y - R0.y[i], ez = z - R0.z[i];
                    if (ex * ex + ey * ey + ez * ez > P.lazySlack2) violated = 1;
                }
            }
            if (anyCross)
            {   // per-species crossing counters: specAcBoxNeg / specAcBoxPos of put_periodic (cuMDfunc.cu:35-106), the columns of
                // msd.dat.  Crossings are rare (a few atoms per step), so plain global atomics cost nothing.
#pragma unroll
                for (int d = 0; d < 6; d++) if (cross[d] != 0.0) atomicAdd(&st->specCross[t * 6 + d], 1ULL);
            }
            A.vx[i] = vx; A.vy[i] = vy; A.vz[i] = vz;
            A.x[i] = x; A.y[i] = y; A.z[i] = z;
            eField = S.charge[t] * (x * P.E[0] + y * P.E[1] + z * P.E[2]);      // integrators.cpp:374 / cuMDfunc.cu:476
        }
        if (BIN)
        {
            int lx;
            myCell = local_cell(P, x, y, z, &lx);
            cellOf[i] = myCell;
            myLayer = lx;
        }
    }
    if (INTEGRATE && P.lazySlack2 > 0.0)
    {   // largest step of any atom (the host sizes the sort interval from it) and the violation flag: one atomic per workgroup at most
        const double mx = block_max(stepLen2, scratch);
        if (threadIdx.x == 0)
        {   // thousands of workgroups on one word would serialise (~90 atomics per microsecond): look first, only a new maximum is published
            const unsigned long long bits = (unsigned long long)__double_as_longlong(mx);
            if (bits > __hip_atomic_load(&cnt->maxStep2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&cnt->maxStep2, bits);
            if (STEPMODE == 2 && P.pad2 && bits > __hip_atomic_load(&cnt->cycMaxRun, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&cnt->cycMaxRun, bits);
        }
        if (!BIN && __syncthreads_or(violated) && threadIdx.x == 0)
        {   // (an earlier violation keeps its step; all writers of one launch write the same value)
            if (cnt->lazyViolated == 0) cnt->lazyViolated = max(P.cycleStep, 1);
            cnt->lazyViolatedEver = 1;
        }
    }
    if (BIN && P.nranks > 1)
    {
        const bool live = i < end;
        const int hw = P.hw[0], lx = myLayer;
        const bool migL = live && lx < hw, migR = live && lx >= P.ncxLocal - hw;
        const bool haloL = live && lx >= hw && lx < 2 * hw, haloR = live && lx >= P.ncxLocal - 2 * hw && lx < P.ncxLocal - hw;
        SendHeader* hL = (SendHeader*)sendLeft;
        SendHeader* hR = (SendHeader*)sendRight;
        const int pML = wave_append(migL, &hL->nMig), pMR = wave_append(migR, &hR->nMig);
        const int pHL = wave_append(haloL, &hL->nHalo), pHR = wave_append(haloR, &hR->nHalo);
        if (migL || migR)
        {
            const int p = migL ? pML : pMR;
            if (p >= lay.migCap) cnt->overflow = 1;
            else
            {
                MigRec* r = (MigRec*)((migL ? sendLeft : sendRight) + lay.mig_offset()) + p;
                r->x = A.x[i]; r->y = A.y[i]; r->z = A.z[i]; r->vx = A.vx[i]; r->vy = A.vy[i]; r->vz = A.vz[i];
                r->U = A.U[i]; r->rad = A.rad[i]; r->type = A.type[i]; r->id = A.id[i]; r->pad0 = 0; r->pad1 = 0;
            }
        }
        if (haloL || haloR)
        {
            const int p = haloL ? pHL : pHR;
            if (p >= lay.haloCap) cnt->overflow = 1;
            else
            {
                HaloRec* r = (HaloRec*)((haloL ? sendLeft : sendRight) + lay.halo_offset()) + p;
                r->x = A.x[i]; r->y = A.y[i]; r->z = A.z[i]; r->rad = A.rad[i]; r->type = A.type[i]; r->id = A.id[i];
            }
        }
    }
    if (BIN)
    {
        // histogram with one atomic per RUN of equal cells inside the wave: the arrays are still in the previous
        // step's cell order, so neighbouring lanes mostly fall into the same cell (about 13 atoms per run)
        const int lane = threadIdx.x & 63;
        const int key = (i < end) ? myCell : -1 - lane;
        const int prevKey = __shfl_up(key, 1, kWave);
        const bool head = (lane == 0) || (key != prevKey);
        const unsigned long long H = __ballot(head);
        const unsigned long long upto = (lane == 63) ? ~0ULL : ((2ULL << lane) - 1ULL);
        const int start = 63 - __clzll((long long)(H & upto));
        const unsigned long long above = H & ~upto;
        const int next = above ? (__ffsll((long long)above) - 1) : 64;
        int base = 0;
        if (head && i < end) base = atomicAdd(&cellCount[myCell], next - start);
        base = __shfl(base, start, kWave);
        if (i < end) slotOf[i] = base + (lane - start);
    }
    if (INTEGRATE)
    {
        double s = 0.0;
        if (P.E[0] != 0.0 || P.E[1] != 0.0 || P.E[2] != 0.0) s = block_sum(eField, scratch);
        if (threadIdx.x == 0) put_partial(partials, maxBlocks, PS_EFIELD, s);
        book_wall_crossings(mom, cross, anyCross, partials, maxBlocks);
    }
}

// Plain step of the lazy re-sort on one GPU, two atoms per thread with 16-byte loads and stores (k_integrate1_bin<2> moves 8 bytes per lane and request;
// the arithmetic, its order and every side effect are the same - see there for the commentary).  Requires an even first index and no velocity scaling
// at the start of the step (Nose-Hoover), which is what Engine checks before choosing it.
__global__ __launch_bounds__(kBlock) void k_integrate_plain2(StepParams P, SpecTable S, AtomArrays A, Counts* __restrict__ cnt, double* __restrict__ partials, int maxBlocks,
                                                             DevStats* __restrict__ st, RefPos R0)
{
    __shared__ double scratch[kBlock / kWave];
    const int begin = cnt->ownedBegin, end = cnt->ownedEnd;
    const int i0 = begin + 2 * (blockIdx.x * kBlock + threadIdx.x);
    const bool pendingKick = st->pendingKick != 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) st->step += 1;
    double roomLeft = -1.0;
    if (P.pad2)
        roomLeft = sqrt(P.lazySlack2) - (double)(P.cycleStep - 1) * sqrt(__longlong_as_double((long long)__hip_atomic_load(&cnt->cycMaxRun, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)));
    double mom[6] = {0, 0, 0, 0, 0, 0}, cross[6] = {0, 0, 0, 0, 0, 0}, stepLen2 = 0.0;
    int anyCross = 0, violated = 0;
    if (i0 < end)
    {
        const bool two = i0 + 1 < end;
        double x[2], y[2], z[2], vx[2], vy[2], vz[2], fx[2], fy[2], fz[2];
        int t[2];
        auto ld2 = [&](const double* p, double (&o)[2]) {
            if (two) { const double2 v = *(const double2*)(p + i0); o[0] = v.x; o[1] = v.y; }
            else { o[0] = p[i0]; o[1] = 0.0; }
        };
        ld2(A.x, x); ld2(A.y, y); ld2(A.z, z); ld2(A.vx, vx); ld2(A.vy, vy); ld2(A.vz, vz); ld2(A.fx, fx); ld2(A.fy, fy); ld2(A.fz, fz);
        if (two) { const int2 tt = *(const int2*)(A.type + i0); t[0] = tt.x; t[1] = tt.y; } else { t[0] = A.type[i0]; t[1] = 0; }
#pragma unroll
        for (int q = 0; q < 2; q++)
        {
            if (q == 1 && !two) break;
            const int i = i0 + q;
            const double rM = S.rMhdt[t[q]], m = S.mass[t[q]];
            if (pendingKick) { vx[q] += rM * fx[q]; vy[q] += rM * fy[q]; vz[q] += rM * fz[q]; }
            vx[q] += rM * fx[q]; vy[q] += rM * fy[q]; vz[q] += rM * fz[q];
            const int ix0 = image_of(x[q], P.L[0], P.invL[0]), iy0 = image_of(y[q], P.L[1], P.invL[1]), iz0 = image_of(z[q], P.L[2], P.invL[2]);
            double dx = 0.0, dy = 0.0, dz = 0.0;
            if (!S.frozen[t[q]]) { dx = vx[q] * P.dt; dy = vy[q] * P.dt; dz = vz[q] * P.dt; x[q] += dx; y[q] += dy; z[q] += dz; }
            const double len2 = dx * dx + dy * dy + dz * dz;
            stepLen2 = fmax(stepLen2, len2);
            bool crossed = false;
            int c;
            c = image_of(x[q], P.L[0], P.invL[0]) - ix0;
            if (c < 0) { mom[0] += m * (-vx[q]); cross[0] += 1; crossed = true; atomicAdd(&st->specCross[t[q] * 6 + 0], 1ULL); }
            else if (c > 0) { mom[1] += m * vx[q]; cross[1] += 1; crossed = true; atomicAdd(&st->specCross[t[q] * 6 + 1], 1ULL); }
            c = image_of(y[q], P.L[1], P.invL[1]) - iy0;
            if (c < 0) { mom[2] += m * (-vy[q]); cross[2] += 1; crossed = true; atomicAdd(&st->specCross[t[q] * 6 + 2], 1ULL); }
            else if (c > 0) { mom[3] += m * vy[q]; cross[3] += 1; crossed = true; atomicAdd(&st->specCross[t[q] * 6 + 3], 1ULL); }
            c = image_of(z[q], P.L[2], P.invL[2]) - iz0;
            if (c < 0) { mom[4] += m * (-vz[q]); cross[4] += 1; crossed = true; atomicAdd(&st->specCross[t[q] * 6 + 4], 1ULL); }
            else if (c > 0) { mom[5] += m * vz[q]; cross[5] += 1; crossed = true; atomicAdd(&st->specCross[t[q] * 6 + 5], 1ULL); }
            if (crossed) anyCross = 1;
            if (!(roomLeft > 0.0 && len2 < roomLeft * roomLeft))
            {
                const double ex = x[q] - R0.x[i], ey = y[q] - R0.y[i], ez = z[q] - R0.z[i];
                if (ex * ex + ey * ey + ez * ez > P.lazySlack2) violated = 1;
            }
        }
        auto st2 = [&](double* p, const double (&o)[2]) {
            if (two) { double2 v; v.x = o[0]; v.y = o[1]; *(double2*)(p + i0) = v; }
            else p[i0] = o[0];
        };
        st2(A.vx, vx); st2(A.vy, vy); st2(A.vz, vz); st2(A.x, x); st2(A.y, y); st2(A.z, z);
    }
    {
        const double mx = block_max(stepLen2, scratch);
        if (threadIdx.x == 0)
        {
            const unsigned long long bits = (unsigned long long)__double_as_longlong(mx);
            if (bits > __hip_atomic_load(&cnt->maxStep2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&cnt->maxStep2, bits);
            if (P.pad2 && bits > __hip_atomic_load(&cnt->cycMaxRun, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&cnt->cycMaxRun, bits);
        }
        if (__syncthreads_or(violated) && threadIdx.x == 0)
        {
            if (cnt->lazyViolated == 0) cnt->lazyViolated = max(P.cycleStep, 1);
            cnt->lazyViolatedEver = 1;
        }
    }
    if (threadIdx.x == 0) put_partial(partials, maxBlocks, PS_EFIELD, 0.0);
    book_wall_crossings(mom, cross, anyCross, partials, maxBlocks);
}

// The dynamic state in one launch (Engine::take_snapshot / replay_from_snapshot: sixteen separate copies cost a short aztot_step call more host time than its
// steps): eleven arrays of doubles and two of 32-bit integers of `n` elements each, the partial sums, DevStats and Counts.
struct StateCopy
{
    const double* srcD[11]; double* dstD[11];
    const int32_t* srcI[2]; int32_t* dstI[2];
    const double* srcP; double* dstP; long long nP;            // partial sums
    const int32_t* srcS[2]; int32_t* dstS[2]; int nS[2];       // DevStats, Counts as 32-bit words
};
__global__ __launch_bounds__(kBlock) void k_copy_state(StateCopy C, int n)
{
    const long long t = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (t < n)
    {
#pragma unroll
        for (int k = 0; k < 11; k++) C.dstD[k][t] = C.srcD[k][t];
        C.dstI[0][t] = C.srcI[0][t]; C.dstI[1][t] = C.srcI[1][t];
    }
    for (long long q = t; q < C.nP; q += (long long)gridDim.x * kBlock) C.dstP[q] = C.srcP[q];
    if (blockIdx.x == 0)
        for (int k = 0; k < 2; k++)
            for (int q = threadIdx.x; q < C.nS[k]; q += kBlock) C.dstS[k][q] = C.srcS[k][q];
}

// ------------------------------------------------------------------------------------------------
// K4: exclusive prefix sum of the cell histogram (calc_firstAtomInCell cuSort.cu:130-143 is ONE thread).
//   k_scan_totals : each workgroup sums its chunk of kScanChunk cells
//   k_scan_apply  : offset = sum of the preceding chunk totals, then a local exclusive scan
// ------------------------------------------------------------------------------------------------
constexpr int kScanChunk = 1024;    // cells per workgroup (4 per thread)

__global__ __launch_bounds__(kBlock) void k_scan_totals(int nCell, const int32_t* __restrict__ cellCount, int32_t* __restrict__ chunkTot)
{
    __shared__ int wt[kBlock / kWave];
    const int c0 = blockIdx.x * kScanChunk + threadIdx.x * 4;
    int s = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) if (c0 + k < nCell) s += cellCount[c0 + k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, kWave);
    if ((threadIdx.x & 63) == 0) wt[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) chunkTot[blockIdx.x] = wt[0] + wt[1] + wt[2] + wt[3];
}

__global__ __launch_bounds__(kBlock) void k_scan_apply(int nCell, int32_t* __restrict__ cellCount, const int32_t* __restrict__ chunkTot,
                                                       int32_t* __restrict__ cellStart, Counts* cnt, DevStats* st, int setPending)
{
    __shared__ int wt[kBlock / kWave];
    __shared__ int chunkBase;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    // sum of all preceding chunk totals
    int pre = 0;
    for (int b = tid; b < (int)blockIdx.x; b += kBlock) pre += chunkTot[b];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) pre += __shfl_down(pre, o, kWave);
    if (lane == 0) wt[w] = pre;
    __syncthreads();
    if (tid == 0) chunkBase = wt[0] + wt[1] + wt[2] + wt[3];
    __syncthreads();
    const int c0 = blockIdx.x * kScanChunk + tid * 4;
    int v[4], s = 0;
#pragma unroll
    for (int k = 0; k < 4; k++)
    {   // the histogram is consumed here: leave it cleared for the next step (clear_clist, cuMDfunc.cu:693)
        v[k] = 0;
        if (c0 + k < nCell) { v[k] = cellCount[c0 + k]; cellCount[c0 + k] = 0; }
        s += v[k];
    }
    int incl = s;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { int n = __shfl_up(incl, o, kWave); if (lane >= o) incl += n; }
    __syncthreads();
    if (lane == 63) wt[w] = incl;
    __syncthreads();
    int run = chunkBase + incl - s;
    for (int k = 0; k < w; k++) run += wt[k];
#pragma unroll
    for (int k = 0; k < 4; k++) { if (c0 + k < nCell) cellStart[c0 + k] = run; run += v[k]; }
    if (blockIdx.x == gridDim.x - 1 && tid == kBlock - 1)
    {
        cellStart[nCell] = run;           // total number of resident atoms
        cnt->nTotal = run;
        // the atoms to be sorted are the old owned range plus whatever k_unpack appended behind it
        cnt->srcBegin = cnt->ownedBegin;
        cnt->srcEnd = cnt->ownedEnd + cnt->nRecv;
        cnt->nRecv = 0;
        st->pendingKick = setPending;     // whether this step leaves its second half-kick to the next k_integrate1_bin
    }
}

// Single-workgroup variant for small grids (a 40 000-atom box, a rank's slab of a 1 M-atom box): one launch instead of two, no
// cross-workgroup hand-over - these systems are bound by launch latency, not bandwidth.  The histogram is read and the
// offsets are written coalesced through an LDS copy; each of the 1024 threads scans a contiguous slice of it.
constexpr int kScanSingleMax = 16384;
__global__ __launch_bounds__(1024) void k_scan_single(int nCell, int32_t* __restrict__ cellCount, int32_t* __restrict__ cellStart, Counts* cnt,
                                                     DevStats* st, int setPending)
{
    __shared__ int32_t buf[kScanSingleMax];
    __shared__ int wt[1024 / kWave];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    for (int c = tid; c < nCell; c += 1024) { buf[c] = cellCount[c]; cellCount[c] = 0; }      // consumed: leave it cleared (clear_clist)
    __syncthreads();
    const int per = (nCell + 1023) >> 10;
    const int c0 = min(nCell, tid * per), c1 = min(nCell, c0 + per);
    int s = 0;
    for (int c = c0; c < c1; c++) s += buf[c];
    int incl = s;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { int n = __shfl_up(incl, o, kWave); if (lane >= o) incl += n; }
    if (lane == 63) wt[w] = incl;
    __syncthreads();
    int run = incl - s;
    for (int k = 0; k < w; k++) run += wt[k];
    for (int c = c0; c < c1; c++) { const int v = buf[c]; buf[c] = run; run += v; }
    __syncthreads();
    for (int c = tid; c < nCell; c += 1024) cellStart[c] = buf[c];
    if (tid == 1023)
    {   // the last thread's running total is the number of resident atoms (slices beyond nCell are empty)
        cellStart[nCell] = run;
        cnt->nTotal = run;
        cnt->srcBegin = cnt->ownedBegin;
        cnt->srcEnd = cnt->ownedEnd + cnt->nRecv;
        cnt->nRecv = 0;
        st->pendingKick = setPending;     // whether this step leaves its second half-kick to the next k_integrate1_bin
    }
}

// ------------------------------------------------------------------------------------------------
// K5a/K5b: counting-sort placement with a deterministic order inside each cell.
//   k_place  : provisional slot (atomic arrival order) -> (id, source index) pairs grouped by cell
//   k_rank_gather : rank of the atom's persistent id inside its cell -> final position; moves the state.
//   (sort_atoms cuSort.cu:145-197 carries the arrival order, which differs run to run.)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_place(const Counts* __restrict__ cnt,
                                                  const int32_t* __restrict__ cellOf, const int32_t* __restrict__ slotOf,
                                                  const int32_t* __restrict__ cellStart, const int32_t* __restrict__ idIn,
                                                  int32_t* __restrict__ tmpId, int32_t* __restrict__ tmpSrc, int32_t* __restrict__ tmpCell)
{
    const int i = cnt->srcBegin + blockIdx.x * kBlock + threadIdx.x;
    if (i >= cnt->srcEnd) return;
    const int c = cellOf[i];
    const int p = cellStart[c] + slotOf[i];
    tmpId[p] = idIn[i];
    tmpSrc[p] = i;
    tmpCell[p] = c;
}

__global__ __launch_bounds__(kBlock) void k_rank_gather(const Counts* __restrict__ cnt, const int32_t* __restrict__ cellStart,
                                                        const int32_t* __restrict__ tmpId, const int32_t* __restrict__ tmpSrc,
                                                        const int32_t* __restrict__ tmpCell, AtomArrays src, AtomArrays dst,
                                                        int32_t* __restrict__ cellOfSorted, int carryForces /* bit 0: forces, bit 1: U + radius */, StepParams P, Counts* cntOut,
                                                        int32_t* __restrict__ idxOfId, RefPos R0, int32_t* __restrict__ haloInfo,
                                                        int32_t* __restrict__ zeroMe, float4* __restrict__ rel)
{
    const int p = blockIdx.x * kBlock + threadIdx.x;
    if (p == 0)
    {   // new owned range (nobody in this launch reads it): the cells of the x-layers [hw, ncxLocal - hw) on a slab rank
        if (P.nranks > 1)
        {
            const int plane = P.nc[1] * P.nc[2];
            const int ob = cellStart[P.hw[0] * plane], oe = cellStart[(P.ncxLocal - P.hw[0]) * plane];
            cntOut->ownedBegin = ob;
            cntOut->ownedEnd = oe;
            if (haloInfo)
            {   // what the host needs to address the boundary layers in the plain steps' coordinate exchange: one small copy instead of three
                haloInfo[0] = cellStart[2 * P.hw[0] * plane]; haloInfo[1] = cellStart[(P.ncxLocal - 2 * P.hw[0]) * plane];
                haloInfo[2] = ob; haloInfo[3] = oe; haloInfo[4] = cnt->nTotal;
                // what this rank will send on every plain step until the next sort: to the left neighbour its first 2 hw owned layers, to the right one its
                // last 2 hw - each neighbour must hold exactly that many ghosts on that side (checked before the first plain step: Engine::take_halo_info)
                // (each message also carries the ghost count of that side, so that both ranks of a boundary see both numbers and arrive at the same verdict)
                haloInfo[5] = haloInfo[0] - ob; haloInfo[6] = ob;                          // to the left neighbour: {what I send leftward, my left ghosts}
                haloInfo[10] = oe - haloInfo[1]; haloInfo[11] = cnt->nTotal - oe;          // to the right neighbour: {what I send rightward, my right ghosts}
            }
        }
        else { cntOut->ownedBegin = 0; cntOut->ownedEnd = cnt->nTotal; }
        if (zeroMe) *zeroMe = 0;        // (the list builder's count of cells without a list: cleared here instead of by a memset node of its own)
    }
    if (p >= cnt->nTotal) return;
    const int c = tmpCell[p];
    const int s = cellStart[c], e = cellStart[c + 1];
    const int myId = tmpId[p];
    int rank = 0;
    for (int q = s; q < e; q++) rank += (tmpId[q] < myId) ? 1 : 0;
    const int d = s + rank;
    const int i = tmpSrc[p];
    const double x = src.x[i], y = src.y[i], z = src.z[i];
    dst.x[d] = x; dst.y[d] = y; dst.z[d] = z;
    if (R0.x) { R0.x[d] = x; R0.y[d] = y; R0.z[d] = z; }         // lazy re-sort: displacements are measured from here
    if (rel)
    {   // what the list builder stages (k_build_lists): the position relative to the centre of the cell the atom was binned into (count_cell:
        // floor(x * cRevSize)), f32, and that cell's z index - a neighbour's position relative to ANOTHER cell's centre is then this plus whole cell edges
        const int gz = cell_coord(z, P.icsz[2], P.nc[2]);
        const double cx = cell_coord(x, P.icsz[0], P.nc[0]) * P.csz[0] + 0.5 * P.csz[0];
        const double cy = cell_coord(y, P.icsz[1], P.nc[1]) * P.csz[1] + 0.5 * P.csz[1];
        const double cz = gz * P.csz[2] + 0.5 * P.csz[2];
        rel[d] = make_float4((float)(x - cx), (float)(y - cy), (float)(z - cz), (float)gz);
    }
    dst.vx[d] = src.vx[i]; dst.vy[d] = src.vy[i]; dst.vz[d] = src.vz[i];
    if (carryForces & 2) { dst.U[d] = src.U[i]; dst.rad[d] = src.rad[i]; }     // thermostat state: only when something reads it
    dst.type[d] = src.type[i]; dst.id[d] = myId;
    if (carryForces & 1) { dst.fx[d] = src.fx[i]; dst.fy[d] = src.fy[i]; dst.fz[d] = src.fz[i]; }
    cellOfSorted[d] = c;
    if (idxOfId) idxOfId[myId] = d;     // bonded terms find their partners through this map (replaces cuSort.cu:199-236)
}

// ------------------------------------------------------------------------------------------------
// K7/K8 (variant 1): per-atom gather over the neighbour-cell stencil.  Full (not half-shell) neighbour
// evaluation: forces are written, never accumulated atomically; no pair tables (the reference's are
// O(nCell^2) on the host, cuCellList.cu:516-531).  Correctness baseline for the tiled kernel.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_pair_atom(StepParams P, SpecTable S, const DevPot* __restrict__ pots, AtomArrays A,
                                                      const Counts* __restrict__ cnt, const int32_t* __restrict__ cellStart,
                                                      const int32_t* __restrict__ cellOfSorted, double* __restrict__ partials, int maxBlocks)
{
    __shared__ double scratch[kBlock / kWave];
    const int i = cnt->ownedBegin + blockIdx.x * kBlock + threadIdx.x;
    PairAcc acc = {0, 0, 0, 0, 0, 0};
    if (P.nranks == 1 && slack_violated(P, cnt))
        for (int k = 0; k < 3; k++) { P.hw[k] += 1; P.nOff[k] = min(2 * P.hw[k] + 1, P.nc[k]); }     // an atom has left its cell's slack: reach one cell further
    if (i < cnt->ownedEnd)
    {
        const double xi = A.x[i], yi = A.y[i], zi = A.z[i];
        const int ti = A.type[i];
        const double radi = P.use_radii ? A.rad[i] : 0.0;
        const int c = cellOfSorted[i];
        const int cz = c % P.nc[2], cy = (c / P.nc[2]) % P.nc[1], lx = c / (P.nc[1] * P.nc[2]);
        const double q = S.charge[ti];
        acc.fx = -q * P.E[0]; acc.fy = -q * P.E[1]; acc.fz = -q * P.E[2];          // clear_force integrators.cpp:17-39
        for (int ox = 0; ox < P.nOff[0]; ox++)
        {
            int nx;
            if (P.nranks > 1) nx = lx + ox - P.hw[0];                                  // slab window: ghost layers are resident
            else if (P.nOff[0] == P.nc[0]) nx = ox;
            else { nx = lx + ox - P.hw[0]; if (nx < 0) nx += P.nc[0]; else if (nx >= P.nc[0]) nx -= P.nc[0]; }
            for (int oy = 0; oy < P.nOff[1]; oy++)
            {
                int ny;
                if (P.nOff[1] == P.nc[1]) ny = oy;
                else { ny = cy + oy - P.hw[1]; if (ny < 0) ny += P.nc[1]; else if (ny >= P.nc[1]) ny -= P.nc[1]; }
                // the z-neighbours of one (x, y) column are contiguous in the sorted arrays unless the run wraps
                for (int oz = 0; oz < P.nOff[2]; oz++)
                {
                    int nz;
                    if (P.nOff[2] == P.nc[2]) nz = oz;
                    else { nz = cz + oz - P.hw[2]; if (nz < 0) nz += P.nc[2]; else if (nz >= P.nc[2]) nz -= P.nc[2]; }
                    const int cn = (nx * P.nc[1] + ny) * P.nc[2] + nz;
                    const int jb = cellStart[cn], je = cellStart[cn + 1];
                    for (int j = jb; j < je; j++)
                    {
                        if (j == i) continue;
                        double dx = xi - A.x[j], dy = yi - A.y[j], dz = zi - A.z[j];
                        min_image(dx, P.L[0], P.half[0]); min_image(dy, P.L[1], P.half[1]); min_image(dz, P.L[2], P.half[2]);
                        const double r2 = dx * dx + dy * dy + dz * dz;
                        if (r2 <= P.r2Max)
                            pair_visit(P, S, pots, dx, dy, dz, r2, ti, A.type[j], radi, P.use_radii ? A.rad[j] : 0.0, acc);
                    }
                }
            }
        }
        A.fx[i] = acc.fx; A.fy[i] = acc.fy; A.fz[i] = acc.fz;
    }
    double s;
    s = block_sum(acc.eV, scratch); if (threadIdx.x == 0) put_partial(partials, maxBlocks, PS_EVDW, s);
    s = block_sum(acc.eC, scratch); if (threadIdx.x == 0) put_partial(partials, maxBlocks, PS_ECOUL, s);
    s = block_sum(acc.dropped, scratch); if (threadIdx.x == 0 && s != 0.0) add_partial(partials, maxBlocks, PS_DROPPED, s);
}

// ------------------------------------------------------------------------------------------------
// K10: second half-kick + kinetic-energy partials (verlet_2stage cuMDfunc.cu:521-600 ; integrate2
//      integrators.cpp:486-531).  Also clears the cell histogram for the next step (clear_clist).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_integrate2(StepParams P, SpecTable S, AtomArrays A, const Counts* __restrict__ cnt,
                                                       int32_t* __restrict__ cellCount, int nCell, double* __restrict__ partials, int maxBlocks,
                                                       DevStats* st)
{
    __shared__ double scratch[kBlock / kWave];
    const int gid = blockIdx.x * kBlock + threadIdx.x;
    if (gid == 0) st->pendingKick = 0;       // this launch pays the kick
    for (int c = gid; c < nCell; c += gridDim.x * kBlock) cellCount[c] = 0;
    const int i = cnt->ownedBegin + gid;
    double kin = 0.0;
    if (i < cnt->ownedEnd)
    {
        const int t = A.type[i];
        const double rM = S.rMhdt[t];
        double vx = A.vx[i] + rM * A.fx[i];
        double vy = A.vy[i] + rM * A.fy[i];
        double vz = A.vz[i] + rM * A.fz[i];
        A.vx[i] = vx; A.vy[i] = vy; A.vz[i] = vz;
        kin = (vx * vx + vy * vy + vz * vz) * S.mass[t];
    }
    double s = block_sum(kin, scratch);
    if (threadIdx.x == 0) put_partial(partials, maxBlocks, PS_EKIN, 0.5 * s);
}

// sum one partial slot over blocks in a fixed order (single workgroup)
__device__ __forceinline__ double reduce_slot(const double* __restrict__ partials, int maxBlocks, int nBlocks, int slot, double* scratch)
{
    double v = 0.0;
    for (int b = threadIdx.x; b < nBlocks; b += blockDim.x) v += partials[(size_t)slot * maxBlocks + b];
    return block_sum(v, scratch);
}

// equilibration scaling (integrate2 tScale branch integrators.cpp:511-522 ; temp_scale cuTemp.cu:77-113):
// k_reduce_kin sums this rank's kinetic energy, k_scale_decision turns the all-rank value into the factor.
__global__ __launch_bounds__(1024) void k_reduce_kin(StepParams P, const double* __restrict__ partials, int maxBlocks, int nBlocks, DevStats* st,
                                                     double* __restrict__ ekOut)
{
    __shared__ double scratch[16];
    double ek = reduce_slot(partials, maxBlocks, nBlocks, PS_EKIN, scratch);
    if (threadIdx.x == 0) { st->local[PS_EKIN] = ek; ekOut[0] = ek; }
}

// tstat_nose (temperature.cpp:339-360) on the scalars; the per-atom velocity scaling happens in the next per-atom kernel
__device__ __forceinline__ double nose_update(const StepParams& P, DevStats* st, double ek, double* kinEout)
{
    st->chit += P.dt * (ek - P.tKin) * P.rQmass;
    const double scale = 1 - P.dt * st->chit;
    const double kinE = ek * scale * scale;
    st->conint += P.dt * st->chit * P.qMassTau2;
    st->chit += P.dt * (kinE - P.tKin) * P.rQmass;
    *kinEout = kinE;
    return scale;
}

// beginning of a step (integrate1 / integrate1_clst, integrators.cpp:305,340): Nose-Hoover scales the velocities with the kinetic
// energy the previous integrate2 left in sim->engKin; the energy it returns is discarded there
__global__ void k_nose_begin(StepParams P, DevStats* st)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double unused;
    st->vscaleBegin = nose_update(P, st, st->ekSim, &unused);
}

// end of a step (integrate2, integrators.cpp:507-529): equilibration rescaling first, then Nose-Hoover
__global__ void k_scale_decision(StepParams P, DevStats* st, const double* __restrict__ ekGlobal)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const long long iStep = st->step;                             // 1-based index of the step in flight (set by k_integrate1_bin)
    double ek = ekGlobal[0];
    double k = 1.0;
    bool changed = false;
    if (P.nEq > 0 && iStep <= P.nEq && P.freqEq > 0 && (iStep % P.freqEq) == 0 && ek != 0.0)
    {
        const double c = (P.tstat == 2) ? 0.25 : 1.0;             // cuTemp.cu:90-94
        k = sqrt(c * P.tKin / ek);
        ek = P.tKin;                                              // engKin := tKin (integrators.cpp:521, cuTemp.cu:112)
        changed = true;
    }
    if (P.tstat == 1)
    {
        double kinE;
        k *= nose_update(P, st, ek, &kinE);
        ek = kinE;
        changed = true;
    }
    if (changed) st->local[PS_EKIN] = ek / (double)P.nranks;
    st->ekSim = ek;
    st->vscale = k;
}

// ------------------------------------------------------------------------------------------------
// K11 + K12: velocity scaling (equilibration) and the radiative thermostat
//   tstat_radi9 cuTemp.cu:689-773, adsorb_rand_photon :484-507, radiate_photon3 :631-685,
//   get_angled_vector :395-453 - fp64, counter-based RNG, fixes of SURVEY Appendix C-9..C-12.
// ------------------------------------------------------------------------------------------------
// The construction of the basis (v2, v3) around v is ill-conditioned when |v_x| << |v|: v2.x = -(v1.y + v1.z) / v1.x amplifies the last-bit
// noise of v1 by 1 / |v1.x| (the preset unit vectors contain such directions, e.g. (2e-13, 0.7071, -0.7071), and the first emission of an
// atom at rest is aimed along one of them).  No FMA contraction here, so that the same v gives the same basis, bit for bit, as the
// scalar C arithmetic of the CPU restatement; both bases are equally valid physically (the azimuth is uniformly random).
__device__ __forceinline__ void angled_vector(const double v[3], double cos_phi, double theta, double out[3])
{
#pragma clang fp contract(off)
    const double l1 = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    double v1[3] = {v[0] / l1, v[1] / l1, v[2] / l1}, v2[3], v3[3];
    if (v1[0] != 0.0) { v2[1] = 1.0; v2[2] = 1.0; v2[0] = -(v1[1] * v2[1] + v1[2] * v2[2]) / v1[0]; }
    else if (v1[1] != 0.0) { v2[0] = 1.0; v2[2] = 1.0; v2[1] = -(v1[2] * v2[2]) / v1[1]; }
    else { v2[0] = 1.0; v2[1] = 0.0; v2[2] = 0.0; }
    v3[0] = v1[1] * v2[2] - v1[2] * v2[1];
    v3[1] = -v1[0] * v2[2] + v1[2] * v2[0];
    v3[2] = v1[0] * v2[1] - v1[1] * v2[0];
    const double l2 = sqrt(v2[0] * v2[0] + v2[1] * v2[1] + v2[2] * v2[2]);
    const double l3 = sqrt(v3[0] * v3[0] + v3[1] * v3[1] + v3[2] * v3[2]);
    for (int k = 0; k < 3; k++) { v2[k] /= l2; v3[k] /= l3; }
    const double sinPhi = sqrt(1 - cos_phi * cos_phi), sinTh = sin(theta), cosTh = cos(theta);
    for (int k = 0; k < 3; k++) out[k] = v1[k] * cos_phi + sinPhi * (cosTh * v2[k] + sinTh * v3[k]);
}

// what k_post does to one atom: equilibration scaling and the radiative thermostat (tstat_radi9 cuTemp.cu:689-773 with adsorb_rand_photon :484-507 and
// radiate_photon3 :631-685); v comes in and goes out through registers, the atom's internal energy and radius through memory.  Returns the atom's U.
__device__ __forceinline__ double post_tstat_atom(const StepParams& P, const SpecTable& S, const AtomArrays& A, const DevStats* __restrict__ st,
                                                  const double* __restrict__ photons, const double* __restrict__ uvx, const double* __restrict__ uvy,
                                                  const double* __restrict__ uvz, int i, double& vx, double& vy, double& vz, long long stepNumber = -1)
{
    double uSum = 0.0;
    const double k = st->vscale;
    if (k != 1.0) { vx *= k; vy *= k; vz *= k; }
    if (P.tstat == 2)
    {
        const uint64_t step = (uint64_t)(stepNumber >= 0 ? stepNumber : st->step);
        const uint64_t id = (uint64_t)A.id[i];
        const int tp = A.type[i];
        const double m = S.mass[tp];
        double U = A.U[i];
        const double pe = photons[(id + step) % (uint64_t)P.nAtGlobal];
        {   // absorb
            const uint32_t rnd = rng_draw(P.seed, step, id, 1) % 3072u;
            const double v02 = vx * vx + vy * vy + vz * vz;
            const double ermc = pe * P.revLight / m;
            vx += ermc * uvx[rnd]; vy += ermc * uvy[rnd]; vz += ermc * uvz[rnd];
            const double v12 = vx * vx + vy * vy + vz * vz;
            U += pe + 0.5 * m * (v02 - v12);
        }
        if (U > P.radThr)
        {   // radiate
            const double u0 = U;
            const double v[3] = {vx, vy, vz};
            const double v02 = vx * vx + vy * vy + vz * vz, v0 = sqrt(v02);
            const double ph = P.radFrac * u0;
            const double ermc = ph * P.revLight / m;
            double d[3];
            if (v0 == 0.0)
            {
                const uint32_t rnd = rng_draw(P.seed, step, id, 2) % 3072u;
                d[0] = uvx[rnd]; d[1] = uvy[rnd]; d[2] = uvz[rnd];
            }
            else
            {
                const double ermcv0 = ermc / v0;
                if (ermcv0 >= 1.0) { d[0] = -v[0] / v0; d[1] = -v[1] / v0; d[2] = -v[2] / v0; }
                else
                {
                    const uint32_t r1 = rng_draw(P.seed, step, id, 2) % 2048u;
                    double cos_phi = (double)r1 / 1024.0 * (1.0 - ermcv0);
                    cos_phi -= 1.0;
                    const uint32_t r2 = rng_draw(P.seed, step, id, 3) % 2048u;
                    const double theta = (double)r2 / 1024.0 * P.numPi;
                    angled_vector(v, cos_phi, theta, d);
                }
            }
            vx += ermc * d[0]; vy += ermc * d[1]; vz += ermc * d[2];
            const double v12 = vx * vx + vy * vy + vz * vz;
            U -= (ph + 0.5 * m * (v12 - v02));
        }
        const double restrE = U < S.mxEng[tp] ? U : S.mxEng[tp];
        A.rad[i] = S.radA[tp] / (S.radB[tp] - restrE);
        A.U[i] = U;
        uSum = U;
    }
    return uSum;
}

__global__ __launch_bounds__(kBlock) void k_post(StepParams P, SpecTable S, AtomArrays A, const Counts* __restrict__ cnt,
                                                 const DevStats* __restrict__ st, const double* __restrict__ photons,
                                                 const double* __restrict__ uvx, const double* __restrict__ uvy, const double* __restrict__ uvz,
                                                 double* __restrict__ partials, int maxBlocks)
{
    __shared__ double scratch[kBlock / kWave];
    const int i = cnt->ownedBegin + blockIdx.x * kBlock + threadIdx.x;
    double uSum = 0.0;
    if (i < cnt->ownedEnd)
    {
        double vx = A.vx[i], vy = A.vy[i], vz = A.vz[i];
        uSum = post_tstat_atom(P, S, A, st, photons, uvx, uvy, uvz, i, vx, vy, vz);
        A.vx[i] = vx; A.vy[i] = vy; A.vz[i] = vz;
    }
    double s = block_sum(uSum, scratch);
    if (threadIdx.x == 0) put_partial(partials, maxBlocks, PS_ETEMP, s);
}

// k_integrate2 and k_post in one launch, for runs whose thermostat needs nothing global between the two (radiative thermostat without equilibration
// scaling: case studies 1 and 2).  Same operations on the same values in the same order - the kinetic energy is booked before the thermostat acts, as in
// the two-launch form - so the results are bit-identical; one launch less per step (6 us of the 24 a step of the 40 000-atom gas takes).
__global__ __launch_bounds__(kBlock) void k_integrate2_post(StepParams P, SpecTable S, AtomArrays A, const Counts* __restrict__ cnt, int32_t* __restrict__ cellCount,
                                                            int nCell, double* __restrict__ partials, int maxBlocks, DevStats* st, const double* __restrict__ photons,
                                                            const double* __restrict__ uvx, const double* __restrict__ uvy, const double* __restrict__ uvz)
{
    __shared__ double scratch[kBlock / kWave];
    const int gid = blockIdx.x * kBlock + threadIdx.x;
    if (gid == 0) st->pendingKick = 0;       // this launch pays the kick
    for (int c = gid; c < nCell; c += gridDim.x * kBlock) cellCount[c] = 0;
    const int i = cnt->ownedBegin + gid;
    double kin = 0.0, uSum = 0.0;
    if (i < cnt->ownedEnd)
    {
        const int t = A.type[i];
        const double rM = S.rMhdt[t];
        double vx = A.vx[i] + rM * A.fx[i];
        double vy = A.vy[i] + rM * A.fy[i];
        double vz = A.vz[i] + rM * A.fz[i];
        kin = (vx * vx + vy * vy + vz * vz) * S.mass[t];
        uSum = post_tstat_atom(P, S, A, st, photons, uvx, uvy, uvz, i, vx, vy, vz);
        A.vx[i] = vx; A.vy[i] = vy; A.vz[i] = vz;
    }
    double s = block_sum(kin, scratch);
    if (threadIdx.x == 0) put_partial(partials, maxBlocks, PS_EKIN, 0.5 * s);
    s = block_sum(uSum, scratch);
    if (threadIdx.x == 0) put_partial(partials, maxBlocks, PS_ETEMP, s);
}

// The boundary between two plain steps of a lazy run with the radiative thermostat (no equilibration scaling), in one launch: what k_integrate2_post
// does to close step t (second half-kick, kinetic energy, thermostat) and what k_integrate1_bin<2> does to open step t + 1 (first half-kick, drift, wall
// counters, displacement check) - per atom, in that order, with the same operations, so the trajectory is bit-identical to the two launches.  P.cycleStep
// is that of the step being OPENED; the number of the step being closed, which keys the thermostat's random draws, is derived from it and from
// DevStats::stepAtSort (the counter itself is advanced by one thread of this launch and must not be read by the others).
__global__ __launch_bounds__(kBlock) void k_boundary_radi(StepParams P, SpecTable S, AtomArrays A, Counts* __restrict__ cnt, double* __restrict__ partials, int maxBlocks,
                                                          DevStats* st, const double* __restrict__ photons, const double* __restrict__ uvx,
                                                          const double* __restrict__ uvy, const double* __restrict__ uvz, RefPos R0)
{
    __shared__ double scratch[kBlock / kWave];
    const int gid = blockIdx.x * kBlock + threadIdx.x;
    const int begin = cnt->ownedBegin, end = cnt->ownedEnd;
    const int i = begin + gid;
    const long long closing = st->stepAtSort + (long long)(P.cycleStep - 1);
    if (gid == 0) { st->pendingKick = 0; st->step = closing + 1; }
    double roomLeft = -1.0;
    if (P.pad2)
        roomLeft = sqrt(P.lazySlack2) - (double)(P.cycleStep - 1) * sqrt(__longlong_as_double((long long)__hip_atomic_load(&cnt->cycMaxRun, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)));
    double kin = 0.0, uSum = 0.0, mom[6] = {0, 0, 0, 0, 0, 0}, cross[6] = {0, 0, 0, 0, 0, 0}, stepLen2 = 0.0;
    int anyCross = 0, violated = 0;
    if (i < end)
    {
        const int t = A.type[i];
        const double rM = S.rMhdt[t], m = S.mass[t];
        const double fx = A.fx[i], fy = A.fy[i], fz = A.fz[i];
        // ---- close step t (k_integrate2_post)
        double vx = A.vx[i] + rM * fx;
        double vy = A.vy[i] + rM * fy;
        double vz = A.vz[i] + rM * fz;
        kin = (vx * vx + vy * vy + vz * vz) * m;
        uSum = post_tstat_atom(P, S, A, st, photons, uvx, uvy, uvz, i, vx, vy, vz, closing);
        // ---- open step t + 1 (k_integrate1_bin<2>; nothing is pending: the kick above was this step's)
        double x = A.x[i], y = A.y[i], z = A.z[i];
        vx += rM * fx;
        vy += rM * fy;
        vz += rM * fz;
        const int ix0 = image_of(x, P.L[0], P.invL[0]), iy0 = image_of(y, P.L[1], P.invL[1]), iz0 = image_of(z, P.L[2], P.invL[2]);
        double dx = 0.0, dy = 0.0, dz = 0.0;
        if (!S.frozen[t]) { dx = vx * P.dt; dy = vy * P.dt; dz = vz * P.dt; x += dx; y += dy; z += dz; }
        stepLen2 = dx * dx + dy * dy + dz * dz;
        int c;
        c = image_of(x, P.L[0], P.invL[0]) - ix0;
        if (c < 0) { mom[0] = m * (-vx); cross[0] = 1; anyCross = 1; } else if (c > 0) { mom[1] = m * vx; cross[1] = 1; anyCross = 1; }
        c = image_of(y, P.L[1], P.invL[1]) - iy0;
        if (c < 0) { mom[2] = m * (-vy); cross[2] = 1; anyCross = 1; } else if (c > 0) { mom[3] = m * vy; cross[3] = 1; anyCross = 1; }
        c = image_of(z, P.L[2], P.invL[2]) - iz0;
        if (c < 0) { mom[4] = m * (-vz); cross[4] = 1; anyCross = 1; } else if (c > 0) { mom[5] = m * vz; cross[5] = 1; anyCross = 1; }
        if (!(roomLeft > 0.0 && stepLen2 < roomLeft * roomLeft))
        {
            const double ex = x - R0.x[i], ey = y - R0.y[i], ez = z - R0.z[i];
            if (ex * ex + ey * ey + ez * ez > P.lazySlack2) violated = 1;
        }
        if (anyCross)
        {
#pragma unroll
            for (int d = 0; d < 6; d++) if (cross[d] != 0.0) atomicAdd(&st->specCross[t * 6 + d], 1ULL);
        }
        A.vx[i] = vx; A.vy[i] = vy; A.vz[i] = vz;
        A.x[i] = x; A.y[i] = y; A.z[i] = z;
    }
    double s = block_sum(kin, scratch);
    if (threadIdx.x == 0) put_partial(partials, maxBlocks, PS_EKIN, 0.5 * s);
    s = block_sum(uSum, scratch);
    if (threadIdx.x == 0) put_partial(partials, maxBlocks, PS_ETEMP, s);
    {
        const double mx = block_max(stepLen2, scratch);
        if (threadIdx.x == 0)
        {
            const unsigned long long bits = (unsigned long long)__double_as_longlong(mx);
            if (bits > __hip_atomic_load(&cnt->maxStep2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&cnt->maxStep2, bits);
            if (P.pad2 && bits > __hip_atomic_load(&cnt->cycMaxRun, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&cnt->cycMaxRun, bits);
        }
        if (__syncthreads_or(violated) && threadIdx.x == 0)
        {
            if (cnt->lazyViolated == 0) cnt->lazyViolated = max(P.cycleStep, 1);
            cnt->lazyViolatedEver = 1;
        }
    }
    if (threadIdx.x == 0) put_partial(partials, maxBlocks, PS_EFIELD, 0.0);
    book_wall_crossings(mom, cross, anyCross, partials, maxBlocks);
}

// ------------------------------------------------------------------------------------------------
// K1 + K13: fold the per-block partials into this rank's per-step sums (fixed order)
// ------------------------------------------------------------------------------------------------
// two-level fixed-order sum: kCollectParts workgroups per slot each reduce a contiguous share of the per-block
// partials, k_finalize adds the kCollectParts sub-totals in order
constexpr int kCollectParts = 16;

__global__ __launch_bounds__(256) void k_collect(double* __restrict__ partials, int maxBlocks, int nBlocksAtoms, int nBlocksPair,
                                                 double* __restrict__ stage, unsigned slotMask, int ekinFromPair, int nBlocksEver)
{
    __shared__ double scratch[4];
    const int slot = blockIdx.x / kCollectParts, part = blockIdx.x % kCollectParts;
    double v = 0.0;
    if ((slotMask >> slot) & 1u)
    {
        // the accumulating slots (wall counters, dropped pairs) are read over every entry any launch has ever booked into (nBlocksEver: the rows are sized for
        // four waves per cell, most engines use a quarter of that): the pair kernels' launch layout may have changed since a drop was booked (split
        // launches of a slab rank, more waves per cell), and idle entries hold zeros
        const int nb = slot_accumulates(slot) ? min(maxBlocks, nBlocksEver)
                       : ((slot == PS_EVDW || slot == PS_ECOUL || (slot == PS_EKIN && ekinFromPair)) ? nBlocksPair : nBlocksAtoms);
        const int per = (nb + kCollectParts - 1) / kCollectParts;
        const int b0 = part * per, b1 = min(nb, b0 + per);
        const bool clear = slot_accumulates(slot);
        for (int b = b0 + threadIdx.x; b < b1; b += blockDim.x)
        {
            const size_t idx = (size_t)slot * maxBlocks + b;
            v += partials[idx];
            if (clear) partials[idx] = 0.0;
        }
    }
    v = block_sum(v, scratch);
    if (threadIdx.x == 0) stage[slot * kCollectParts + part] = v;
}

// reset_quantities + calc_quantities (cuMDfunc.cu:270, main.cu:121-194 ; serial calc_chars integrators.cpp:63-73).
// Works on this rank's sums; the cross-rank sum is taken when the host asks for statistics (Engine::get_stats).
__global__ void k_finalize(StepParams P, DevStats* st, const double* __restrict__ stage, unsigned slotMask)
{
    // one lane per slot adds the slot's sub-totals in their fixed order (the loads of all slots travel together: one thread walking 20 x 16 of them in turn took
    // 13 us at the end of every aztot_step call); lane 0 then does the bookkeeping
    __shared__ double sums[PS_COUNT];
    if (blockIdx.x != 0) return;
    if (threadIdx.x < PS_COUNT)
    {
        const int slot = threadIdx.x;
        double v;
        if ((slotMask >> slot) & 1u)
        {
            v = 0.0;
            for (int k = 0; k < kCollectParts; k++) v += stage[slot * kCollectParts + k];
            st->local[slot] = v;
        }
        else v = st->local[slot];
        sums[slot] = v;
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    st->engElecField = sums[PS_EFIELD];
    st->engVdW = sums[PS_EVDW];
    st->engCoul = sums[PS_ECOUL];
    st->engKin = sums[PS_EKIN];
    if ((slotMask >> PS_EKIN) & 1u) st->ekSim = sums[PS_EKIN];    // (steps that did not go through k_scale_decision: the reference's sim->engKin is the last integrate2's)
    st->engTemp = sums[PS_ETEMP];
    // running totals grow only by what THIS call collected: a slot outside slotMask still holds the previous call's window in
    // local[] (aztot_forces after aztot_step collects the energies only and must leave the wall counters alone)
    for (int k = 0; k < 6; k++)
    {
        if ((slotMask >> (PS_MOM_XN + k)) & 1u) st->mom[k] += sums[PS_MOM_XN + k];
        if ((slotMask >> (PS_CNT_XN + k)) & 1u) st->cross[k] += (long long)(sums[PS_CNT_XN + k] + 0.5);
    }
    if ((slotMask >> PS_DROPPED) & 1u) st->dropped += (long long)(sums[PS_DROPPED] + 0.5);
    st->temperature = 2.0 * st->engKin * P.revDegFree * P.rkB;
    st->engBond = sums[PS_EBOND];
    st->engAngle = sums[PS_EANGLE];
    st->engPot = st->engCoul + st->engVdW;
    st->engTot = st->engElecField + st->engVdW + (st->engCoulConst + st->engCoulRec + st->engCoul) + st->engKin + st->engBond + st->engAngle;
}

}  // namespace aztot
