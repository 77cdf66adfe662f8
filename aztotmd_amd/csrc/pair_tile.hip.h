// Pair-force kernel, variant 2: one wave64 per cell, neighbour atoms staged through LDS.
//
// Replaces the reference's cell_list5a + cell_list4b_noshared + pair_1 (cuPairs.cu:2266,1474,117), which walk
// host-built O(nCell^2) cell-pair tables, re-read both atoms from global memory for every pair and add every
// pair force with three global float atomics.  Here:
//   * the neighbour stencil (2hw+1)^3 is walked on the fly; the periodic image shift is applied once per
//     neighbour run while it is staged, so the inner loop has no minimum-image selects;
//   * staging is a coalesced SoA load of whole z-runs (cells that are contiguous in the sorted arrays), pruned
//     against the centre cell's bounding box and packed into LDS with wave ballot + popcount prefix; the runs are
//     looked up by all lanes in parallel and loaded in groups of four so that a wave pays ~4 memory round trips
//     per cell instead of ~20;
//   * lanes are split (i-slot, j-slice): 16 atoms x 4 slices for the typical 13-atom cell, so ~85 % of the lanes
//     work instead of 21 %; slices are folded with two xor-shuffles in a fixed order;
//   * every atom's force is the sum over ALL its neighbours (no Newton-3 halving): written once, no atomics,
//     bit-reproducible; each pair visit books half of the pair energy;
//   * workgroup -> cell mapping keeps each XCD on a contiguous eighth of the cell list (private L2 per XCD).
// One matrix instruction: the distance filter of pass 1 (v_mfma_f32_16x16x4_f32 as a 256-wide comparator, tile_filter); the potential is vector fp64.
// Two roles (template parameter of k_pair_tile): the force kernel of runs that rebuild their cells every step, and the clean-up launch behind
// k_pair_list (CLEANUP; pair_list.hip.h holds the list kernel and the kernel that builds its lists).
#pragma once
#include "kernels.hip.h"

namespace aztot {

constexpr int kTileCap = 320;      // candidates resident in LDS per wave
constexpr int kTilePad = 16;       // far-away dummies behind the last candidate (4 unrolled iterations x 4 slices)
constexpr int kTileLds = kTileCap + kTilePad;     // = 16 (mod 32): the three coordinate arrays start 32 banks apart, so the 16 + 16 (+ 16) lanes of
                                                  // the matrix-operand read below hit 64 different banks
static_assert(kTileLds % 32 == 16, "coordinate arrays must be staggered by half the LDS banks");

typedef float float4_t __attribute__((ext_vector_type(4)));


// the tile kernel needs every neighbour cell to be reached through exactly one periodic image
inline bool pair_tile_supported(const StepParams& P)
{
    for (int k = 0; k < 3; k++)
    {
        const bool slabX = (k == 0 && P.nranks > 1);
        if (!slabX && P.nc[k] < 2 * P.hw[k] + 1) return false;
    }
    return true;
}

inline int pair_tile_cells(const StepParams& P)
{
    const int plane = P.nc[1] * P.nc[2];
    return (P.nranks > 1) ? (P.ncxLocal - 2 * P.hw[0]) * plane : P.nCellLocal;
}
inline int pair_tile_grid(const StepParams& P) { return 8 * ((pair_tile_cells(P) + 7) / 8); }

// number of set bits of a 64-bit ballot below this lane: two v_mbcnt instructions
__device__ __forceinline__ int lanes_below(unsigned long long mask)
{
    return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
}

// 8-byte element j of a device array through a 32-bit byte offset (scalar base + vector offset addressing)
__device__ __forceinline__ double ld_f64(const double* __restrict__ base, int j)
{
    return *(const double*)((const char*)base + ((unsigned)j << 3));
}
__device__ __forceinline__ int ld_i32(const int32_t* __restrict__ base, int j)
{
    return *(const int32_t*)((const char*)base + ((unsigned)j << 2));
}

// The two passes over one staged chunk; LG = log2(i-slots), NS = 64 >> LG = lanes (slices) per i-atom.
//   pass 1  distance tests only (7 fp64 ops per candidate); hits are recorded in three per-lane 32-bit masks
//   pass 2  every lane pops its own hits (from whichever word still has one), so the potential runs on densely
//           filled waves: only ~20 % of the candidates are inside the cut-off, and evaluating inline would execute
//           the potential on nearly every wave-iteration with 80 % of the lanes masked off.
// rocprof shows the kernel VALU-issue bound (VALU busy ~85 %), so this is all about instructions per wave: constant
// LDS offsets (NS is a template parameter), no bounds clamps (far-away dummies pad the tile), Newton-refined
// v_rcp_f64, branch-free potential.
constexpr int kPairTabStride = 8;   // {p0..p4, r2cut, kqq, potential type} per ordered species pair.  Not one byte more: the Coulomb
                                    // kernels sit at 10 192 B of LDS, and 10 240 B is the limit for 16 waves per CU (+64 B cost 7 %)
constexpr int kLjSpecMax = 4;      // MODE 2 keeps the per-species-pair Lennard-Jones / charge-product table in LDS

// One pair visit of the specialised tile kernels: potential + electrostatics of the pair (i, candidate) at separation (dx, dy, dz), r2 = |d|^2.
// `live` = the lane really has a candidate (only the unmasked Coulomb forms of tile_passes pass false).  Shared by the staging kernel below and
// by the pair-list kernel (pair_list.hip.h), so both evaluate a pair with exactly the same operations.
// The two wave-uniform numbers the table-driven bodies need in every visit.  The list kernel keeps them in VECTOR registers (pair_hot_in_vgprs): with the 31
// polynomial coefficients of exp and erfc in scalar registers the compiler otherwise re-loads them from the kernel arguments inside the pair loop, and the
// s_waitcnt lgkmcnt(0) that follows also waits for the LDS reads of the next candidate issued just before - the software pipelining was gone.
// ljA2 (one-species Lennard-Jones body): 48 eps sigma^12, the coefficient of the force polynomial's FMA - a vector register in the list kernel, because an FMA
// takes ONE scalar operand and the other constant (-24 eps sigma^6) already is one: left to itself the compiler copied a constant into vector registers on every visit
// elScale2, daipi2 (Fennell / Ewald bodies): round 4 found the list kernel's Fennell loop re-loading both from the kernel arguments on EVERY visit (and reading two
// spilled polynomial coefficients back with v_readlane): the loop had run out of scalar registers again.
struct PairHot { double r2Max, alpha, ljA2, elScale2, daipi2; };
__device__ __forceinline__ PairHot pair_hot(const StepParams& P, const DevPot& lj) { return PairHot{P.r2Max, P.alpha, lj.p4, P.el_scale2, P.daipi2}; }
__device__ __forceinline__ PairHot pair_hot_in_vgprs(const StepParams& P, const DevPot& lj)
{
    PairHot h{P.r2Max, P.alpha, lj.p4, P.el_scale2, P.daipi2};
    asm volatile("" : "+v"(h.r2Max), "+v"(h.alpha), "+v"(h.elScale2), "+v"(h.daipi2));
    return h;
}
__device__ __forceinline__ PairHot pair_hot_lj_in_vgprs(const StepParams& P, const DevPot& lj)
{
    PairHot h{P.r2Max, P.alpha, lj.p4, P.el_scale2, P.daipi2};
    asm volatile("" : "+v"(h.ljA2));
    return h;
}

// MASKED (list kernel: nearly every entry of a list is inside the cut-off): pairs outside the cut-off leave through the EXEC mask instead of being carried
// along with r^2 = 1e300 and multiplied away - four selects less per visit, same arithmetic for the pairs that count.
template <int MODE, int VDW, bool MASKED = false>
__device__ __forceinline__ void pair_body(const StepParams& P, const SpecTable& S, const DevPot* __restrict__ pots, const DevPot& lj, const double* pairTab,
                                          bool live, double dx, double dy, double dz, double r2, int ti, int tj, double radi, double radj, double ljDropR2,
                                          int& nDropHalf, PairAcc& ra, const PairHot& H)
{
    if (MODE == 1)
    {   // fer_lj vdw.cpp:16-26 ; pair_inter integrators.cpp:139-185.  The atom itself (r2 == 0 exactly: its own
        // cell is part of the tile, unshifted), candidates the conservative filter let through and dead lanes
        // are pushed out to a huge r2, where sr6 underflows to exactly 0 and with it energy and force.
        // The exact cut-off test and the self pair also go through the EXEC mask: the filter is conservative by 1e-5, so next to the
        // atom's own copy (one hit in ~57) practically every lane passes and nothing diverges
        // (MASKED = the list kernel: an atom is never on its own list and idle lanes never meet the dummy candidate at r = 0, so r^2 > 0 needs no test)
        if ((MASKED || r2 > 0.0) & (r2 <= lj.r2cut))
        {
            // fer_lj (vdw.cpp:16-26) computes sr2 = sigma^2 / r^2, sr6 = sr2^3, f = 24 eps / r^2 * sr6 * (2 sr6 - 1): seven multiplications behind the
            // reciprocal.  The same polynomial in u = 1 / r^2 with the constants folded on the host (Engine::allocate: lj.p3 = 24 eps sigma^6,
            // lj.p4 = 48 eps sigma^12) takes five - u^2, u^3, u^4, one FMA, one product -, each result within a few ulp of the reference's
            // (the parity tests hold forces to 1e-11): 24 -> 22 vector instructions per visit of the list kernel
            const double u = fast_rcp(r2);
            const double u2 = u * u, u3 = u2 * u;
            double fm = (u2 * u2) * fma(H.ljA2, u3, -lj.p3);
            {   // energy (launches that book energies only: the compiler drops it elsewhere): half of 4 eps sr6 (sr6 - 1), sr6 = sigma^6 u^3, as
                // u^3 (E2 u^3 - E1) with E1 = 2 eps sigma^6, E2 = 2 eps sigma^12 (wave-uniform, computed once per wave): two instructions per visit
                const double s6 = lj.p1 * lj.p1 * lj.p1, e1 = 0.5 * lj.p0 * s6, e2 = e1 * s6;
                ra.eV = fma(u3, fma(e2, u3, -e1), ra.eV);
            }
            // integrators.cpp:170-174: a pair with f^2 > 1e10 is dropped.  |f| grows monotonically as r shrinks below the minimum, so
            // the exact test is only reached (wave-uniform branch, practically never) when some lane is inside a generous radius
            if (__builtin_expect(__any(r2 < ljDropR2), 0))
            {
                const bool tooBig = fm * fm > 1e10;
                nDropHalf += tooBig ? 1 : 0;
                fm = tooBig ? 0.0 : fm;
            }
            ra.fx = fma(fm, dx, ra.fx); ra.fy = fma(fm, dy, ra.fy); ra.fz = fma(fm, dz, ra.fz);
        }
    }
    else if (MODE == 4)
    {   // one species, radius-dependent 'surk' potential (surk_pot cuVdW.cu:236-257; cuPairs.cu:145-146), no electrostatics: case
        // study 2.  U = a b r^-6 (C1 a^2 b^2 / r - C2 / (ka a + kb b)) with a, b the radii the radiative thermostat writes
        // (cuTemp.cu:757-759); same operation order as the generic kernel's vdw_force, branch-free like the Lennard-Jones body
        const bool pairOk = live & (r2 > 0.0) & (r2 <= lj.r2cut);
        const double r2s = pairOk ? r2 : 1e300;             // r^-6 underflows to 0: no energy, no force
        const double ir = fast_rsqrt(r2s), r2i = ir * ir;
        const double c2ir_sum = lj.p1 * fast_rcp(lj.p2 * radi + lj.p3 * radj);
        const double r_prod = radi * radj;
        const double C1ab2 = r_prod * r_prod * lj.p0;
        const double ir6 = r2i * r2i * r2i;
        ra.eV = fma(0.5, r_prod * ir6 * (C1ab2 * ir - c2ir_sum), ra.eV);
        const double f = r_prod * ir6 * r2i * (7.0 * C1ab2 * ir - 6.0 * c2ir_sum);
        const bool tooBig = f * f > 1e10;                               // integrators.cpp:170-174: pair dropped
        nDropHalf += tooBig ? 1 : 0;
        const double fm = tooBig ? 0.0 : f;
        ra.fx = fma(fm, dx, ra.fx); ra.fy = fma(fm, dy, ra.fy); ra.fz = fma(fm, dz, ra.fz);
    }
    else if (MODE >= 2)
    {   // one potential family for every species pair (VDW: 1 lnjs, 2 buck, 3 p746, 4 bmhs - fer_* of vdw.cpp:16-157), electrostatics
        // none, direct, Fennell/DSF (fennel elec.cpp:430-444) or the real-space Ewald term; parameters per species pair come from a
        // small LDS table {p0..p4, r2cut, kqq, aux}.  Branch-free: a pair outside its potential's cut-off is multiplied away.
        // MODE 5 = MODE 2 with the electrostatics known when the kernel is compiled (Fennell/DSF): no wave-uniform branches inside the pair loop, which cost
        // MODE 2 its instruction scheduling (every table read waited for on the spot) and, with three variants of the body alive, its scalar registers
        const double* pp = pairTab + (ti * P.nSpec + tj) * kPairTabStride;
        // (Lennard-Jones family: slots 3 and 4 hold the force polynomial's constants 24 eps sigma^6 and 48 eps sigma^12, Engine::allocate - see MODE 1)
        constexpr bool kLJ = (VDW == 1 || VDW == 6);                  // VDW 6: Lennard-Jones whose per-pair cut-off test always passes (Engine::construct)
        const double tabP1 = kLJ ? pp[3] : pp[1], tabP2 = kLJ ? pp[4] : pp[2], tabCut = (VDW == 6) ? 0.0 : pp[5], tabKqq = pp[6];      // read whatever the pair turns out to be: the reads travel together
        // (MASKED = the list kernel: an atom is never on its own list and idle lanes never meet the dummy candidate at r = 0, so r^2 > 0 needs no test)
        const bool pairOk = live & (MASKED || r2 > 0.0) & (r2 <= H.r2Max);
        if (MASKED && !pairOk) return;
        const double r2s = (MASKED || pairOk) ? r2 : 1e300;
        const bool coul = (MODE == 3) || (MODE == 5) || (P.elec_type != 0);               // wave-uniform
        const bool needR = coul || !kLJ;
        const double ir = needR ? fast_rsqrt(r2s) : 0.0;
        const double r2i = needR ? ir * ir : fast_rcp(r2s);
        const double r = r2s * ir;
        const bool vdwOk = (VDW == 6) || r2s <= tabCut;
        double f;
        if (kLJ)
        {   // fer_lj vdw.cpp:16-26 (p0 = 4 eps, p1 = sigma^2, p2 = 24 eps) as the polynomial f = u^4 (A2 u^3 - A1) in u = 1 / r^2, as in MODE 1
            const double u2 = r2i * r2i, u3 = u2 * r2i;
            const double fl = (u2 * u2) * fma(tabP2, u3, -tabP1);
            f = vdwOk ? fl : 0.0;
            {   // energy (launches that book energies only)
                const double s2 = pp[1];
                const double sr6 = vdwOk ? (s2 * s2 * s2) * u3 : 0.0;
                ra.eV = fma(0.5 * pp[0], sr6 * (sr6 - 1.0), ra.eV);
            }
        }
        else
        {
            const double w = vdwOk ? 1.0 : 0.0;
            const double r4i = r2i * r2i, r6i = r4i * r2i;
            const int pt = (VDW == 5) ? (int)pp[7] : VDW;                  // VDW 5: the families are mixed - per-pair type, divergent
            double e;
            if (pt == 1)
            {   // fer_lj vdw.cpp:16-26
                const double sr2 = pp[1] * r2i, sr6 = sr2 * sr2 * sr2;
                e = pp[0] * sr6 * (sr6 - 1.0);
                f = pp[2] * r2i * sr6 * (2.0 * sr6 - 1.0);
            }
            else if (pt == 2)
            {   // fer_buckingham vdw.cpp:60-70: A exp(-r/rho) - C/r^6 ; 1/rho in slot 3
                const double ex = pp[0] * exp_nonpos(-r * pp[3]);
                e = ex - pp[2] * r6i;
                f = ex * ir * pp[3] - 6.0 * pp[2] * r4i * r4i;
            }
            else if (pt == 3)
            {   // fer_746 vdw.cpp:144-157: p0/r^7 - p1/r^4 - p2/r^6
                e = r4i * (pp[0] * r2i * ir - pp[1] - pp[2] * r2i);
                f = r6i * (7.0 * pp[0] * r2i * ir - 4.0 * pp[1] - 6.0 * pp[2] * r2i);
            }
            else if (pt == 4)
            {   // fer_bhm vdw.cpp:102-112: A exp(B (sigma - r)) - C/r^6 - D/r^8
                const double ex = pp[0] * exp_nonpos(pp[1] * (pp[2] - r));
                e = ex - pp[3] * r6i - pp[4] * r4i * r4i;
                f = pp[1] * ex * ir - 6.0 * pp[3] * r4i * r4i - 8.0 * pp[4] * r4i * r4i * r2i;
            }
            else { e = 0.0; f = 0.0; }                                       // no potential for this species pair
            ra.eV = fma(0.5 * w, e, ra.eV);
            f *= w;
        }
        if (MODE == 5 || (MODE == 2 && P.elec_type == 3))
        {
            const double kqq = (MASKED || pairOk) ? tabKqq : 0.0;
            const double ar = H.alpha * r;
            const double ex = exp_nonpos<!MASKED>(-ar * ar);
            const double erfcar = erfc_given_exp(ar, ex);
            ra.eC = fma(0.5 * kqq, erfcar * ir - P.el_scale + H.elScale2 * (r - P.rReal), ra.eC);
            f = fma(kqq * ir, (erfcar * r2i + H.daipi2 * ex * ir) - H.elScale2, f);
        }
        else if (MODE == 3)
        {   // real-space term of the Ewald sum: coul_iter elec.cpp:344-369 (real_ewald cuElec.cu:94-113)
            const double kqq = (MASKED || pairOk) ? tabKqq : 0.0;
            const double ar = H.alpha * r;
            const double ex = exp_nonpos<!MASKED>(-ar * ar);
            const double erfcar = erfc_given_exp(ar, ex);
            ra.eC = fma(0.5 * kqq, erfcar * ir, ra.eC);
            f = fma(kqq * ir * r2i, fma(H.daipi2 * r, ex, erfcar), f);
        }
        else if (MODE == 2 && P.elec_type == 1)
        {   // direct_coul elec.cpp:415-428
            const double kqq = (MASKED || pairOk) ? tabKqq : 0.0;
            ra.eC = fma(0.5 * kqq, ir, ra.eC);
            f = fma(kqq * ir, r2i, f);
        }
        // integrators.cpp:170-174: a pair with f^2 > 1e10 is dropped.  Lennard-Jones family: beyond ljDropR2 neither part of the force can get there
        // (Engine::construct), so the exact test is a wave-uniform branch a liquid never takes; the other families test every pair
        double fm = f;
        if (!kLJ || __builtin_expect(__any(r2s < ljDropR2), 0))
        {
            const bool tooBig = f * f > 1e10;
            nDropHalf += tooBig ? 1 : 0;
            fm = tooBig ? 0.0 : f;
        }
        ra.fx = fma(fm, dx, ra.fx); ra.fy = fma(fm, dy, ra.fy); ra.fz = fma(fm, dz, ra.fz);
    }
    else if (live && r2 > 0.0 && r2 <= P.r2Max)
        pair_visit(P, S, pots, dx, dy, dz, r2, ti, tj, radi, radj, ra);
}

// Pass 1 of one round (NW words of 32 candidates per lane, starting at per-lane candidate rb): distance tests only; the hits come back as per-lane bit
// masks m[0..NW-1], candidate number b of a word at bit 31 - b.  LG = log2(atom slots), NS = 64 >> LG lanes (slices) per atom; STRIDE: distance (in
// entries) between the x, y and z arrays of the tile.  Shared by the force kernels (threshold rc^2) and by the list-building launch (deal_hits, threshold (rc + 2 slack)^2).
template <int LG, int STRIDE, int NW>
__device__ __forceinline__ void tile_filter(const double* tx, const double* ty, const double* tz, const float* tw, int rb, int iters, int slice,
                                            double xi, double yi, double zi, float filtB, double r2Filter, uint32_t (&m)[4])
{
    constexpr int NS = kWave >> LG;
#pragma unroll
    for (int w = 0; w < NW; w++)
    {
        const int tb = rb + w * 32;
        const int nb = min(32, iters - tb);            // multiple of 4, may be <= 0
        uint32_t miss = 0xFFFFFFFFu;                   // one bit per candidate, 1 = outside the cut-off
        if (LG == 4)
        {   // 16 atoms x 4 slices: the distance filter of a block of 16 candidates x 16 atoms is ONE v_mfma_f32_16x16x4_f32:
            //   D[cand][atom] = -|rj|^2 + (xj, yj, zj, 1) . (2 xi, 2 yi, 2 zi, thr - |ri|^2) = thr - |ri - rj|^2
            // in f32 on cell-relative coordinates (|x| < cell/2 + rc), thr widened by the f32 error bound, so the filter stays
            // conservative; pass 2 applies the exact fp64 test.  The matrix instruction holds the vector pipe for ~38 cycles per 256
            // tests where the fp64 VALU form below needs ~140 (tools/ubench/valu_rates.hip); an fp64 MFMA would not help: it runs on
            // the same fp64 units as v_fma_f64 (measured: additive).  Operand maps (gfx950): A[row l&15][k l>>4], B[k l>>4][col l&15],
            // C/D[row 4 (l>>4) + reg][col l&15].  Rows are fed in the order perm(c) = (c >> 2) + 4 (c & 3), which makes register r of the
            // lane (atom l&15, slice l>>4) the candidate 16 B + slice + 4 r: exactly the lane's own interleaved candidates; the
            // -|rj|^2 of those four candidates arrive as the C operand.
            const int lane = threadIdx.x & (kWave - 1);
            const int c = lane & 15, k = lane >> 4;
            const int permc = (c >> 2) + ((c & 3) << 2);
            // lanes k = 3 feed the constant 1; they read (and ignore) the y array: banks 32-63, away from the z lanes of their half-wave
            const double* pa = tx + (k == 3 ? 1 : k) * STRIDE + tb * NS + permc;
            const float* pc = tw + tb * NS + k;        // -|rj|^2 of the lane's candidates k, k + 4, k + 8, k + 12 of every block
            const int nblk = nb >> 2;                  // blocks of 16 candidates in this word (<= 8)
#pragma unroll
            for (int q = 0; q < 8; q++)
            {
                if (q < nblk)
                {
                    const float av = (float)pa[q * 16];
                    const float a = (k == 3) ? 1.0f : av;
                    const float4_t cw = {pc[q * 16], pc[q * 16 + 4], pc[q * 16 + 8], pc[q * 16 + 12]};      // two ds_read2_b32
                    const float4_t d = __builtin_amdgcn_mfma_f32_16x16x4f32(a, filtB, cw, 0, 0, 0);
                    miss = __builtin_amdgcn_alignbit(miss, __float_as_uint(d[0]), 31);
                    miss = __builtin_amdgcn_alignbit(miss, __float_as_uint(d[1]), 31);
                    miss = __builtin_amdgcn_alignbit(miss, __float_as_uint(d[2]), 31);
                    miss = __builtin_amdgcn_alignbit(miss, __float_as_uint(d[3]), 31);   // newest candidate in bit 0
                }
            }
            m[w] = (nb > 0) ? (~miss << (32 - nb)) : 0u;
            continue;
        }
        const double* px = &tx[tb * NS + slice];
        const double* py = &ty[tb * NS + slice];
        const double* pz = &tz[tb * NS + slice];
        for (int b = 0; b < nb; b += 4)
        {
#pragma unroll
            for (int u = 0; u < 4; u++)
            {
                // 7 VALU per candidate: 3 subtractions, 3 FMAs that leave d = rc^2(1+eps) - r^2, and one v_alignbit
                // that shifts the sign of d into the mask.  The filter is conservative by eps; pass 2 applies the
                // exact r^2 <= rc^2 test of the reference (pair_inter integrators.cpp:148).
                const double dx = xi - px[u * NS], dy = yi - py[u * NS], dz = zi - pz[u * NS];
                const double d = fma(-dz, dz, fma(-dy, dy, fma(-dx, dx, r2Filter)));
                miss = __builtin_amdgcn_alignbit(miss, (uint32_t)(__double_as_longlong(d) >> 32), 31);   // newest candidate in bit 0
            }
            px += 4 * NS; py += 4 * NS; pz += 4 * NS;
        }
        // left-align: candidate number b of this word sits at bit 31 - b, whatever the word length
        m[w] = (nb > 0) ? (~miss << (32 - nb)) : 0u;
    }
}

// The two passes over one staged chunk.  NW: 32-candidate mask words per round (3 for the one-wave tile of <= 320 candidates, 4 for the shared tile of
// a retired kernel whose windows held ~400)
template <int MODE, int VDW, int LG, int STRIDE = kTileLds, int NW = 3>
__device__ __forceinline__ void tile_passes(const StepParams& P, const SpecTable& S, const DevPot* __restrict__ pots, const DevPot& lj,
                                            const double* tx, const double* ty, const double* tz, const float* tw, const uint8_t* ttyp, const double* trad,
                                            const double* pairTab, int T, int slice, double xi, double yi, double zi, int ti, double radi,
                                            float filtB, float filtC, PairAcc& acc)
{
    constexpr int NS = kWave >> LG;
    const int iters = ((T + NS - 1) / NS + 3) & ~3;        // per-lane candidates, rounded up to the unroll factor
    const double r2Filter = P.r2Max * (1.0 + 1e-13);       // conservative pass-1 threshold
    const double ljDropR2 = P.ljDropR2;                   // MODE 1: no pair beyond this r^2 can break the f^2 > 1e10 rule (Engine::construct)
    int nDropHalf = 0;
    if (P.pad0 & 2048) return;                             // measurement aid (bench.py --debug 2048): staging only, forces are wrong
    for (int rb = 0; rb < iters; rb += 32 * NW)
    {
        uint32_t m[4] = {0u, 0u, 0u, 0u};
        tile_filter<LG, STRIDE, NW>(tx, ty, tz, tw, rb, iters, slice, xi, yi, zi, filtB, r2Filter, m);
        // pass 2: every lane pops its own hits.  `cur` is the word being drained, `nxt`/`lst` the ones still waiting; when
        // `cur` runs dry the next word slides in (selects, no branches), so a lane keeps busy as long as ANY of its three
        // words has hits left - the wave loops max-over-lanes(hits per lane) times, not sum-over-words(max per word).
        // Lanes whose words are empty sit the iteration out (EXEC mask: two scalar instructions, where selecting a dummy candidate and
        // carrying a `live` flag through the cut-off test cost three vector ones); the body itself is branch-free.  The loop is a
        // do-while on purpose: with the exit test at the top the compiler copies the five 64-bit accumulators in every iteration.
        PairAcc& ra = acc;
        uint32_t cur = m[0], nxt = m[1], lst = m[2], ult = m[3];
        int kbase = rb * NS + slice;                       // tile index of bit 31 of `cur`
        if (__any((cur | nxt | lst | ult) != 0u))
        do
        {
            const bool dry = cur == 0u;
            cur = dry ? nxt : cur;
            nxt = dry ? lst : nxt;
            if (NW > 3) { lst = dry ? ult : lst; ult = dry ? 0u : ult; }
            else lst = dry ? 0u : lst;
            kbase += dry ? 32 * NS : 0;
            // (the Coulomb bodies are the exception: measured on C3 they lose 3 % to the EXEC-masked form, so they keep the dummy candidate)
            constexpr bool kMaskBody = (MODE == 1 || MODE == 4);
            const bool live = cur != 0u;
            if (kMaskBody ? live : true)
            {
            const int b = __clz(cur | 1u);
            cur &= ~(0x80000000u >> b);                    // for a dry word this clears bit 0 of zero: harmless
            const int k = (kMaskBody || live) ? kbase + b * NS : T;       // dead lanes of the unmasked form: the first dummy
            const double dx = xi - tx[k], dy = yi - ty[k], dz = zi - tz[k];
            const double r2 = dx * dx + dy * dy + dz * dz;
            int tj = 0;
            double radj = 0.0;
            if (MODE == 0 || MODE == 2 || MODE == 3 || MODE == 5) tj = ttyp[k];
            if (MODE == 0 || MODE == 4) radj = trad[k];
            pair_body<MODE, VDW>(P, S, pots, lj, pairTab, live, dx, dy, dz, r2, ti, tj, radi, radj, ljDropR2, nDropHalf, ra, pair_hot(P, lj));
            }
        } while (__any((cur | nxt | lst | ult) != 0u));
    }

    if (MODE != 0) acc.dropped += 0.5 * (double)nDropHalf;     // dropped pairs, counted per lane in "half pair" units (every pair is visited from both ends)
}

// Lists of the lazy re-sort, written by k_build_lists and read by k_pair_list (both in pair_list.hip.h); the clean-up launch of k_pair_tile looks at the
// headers to find the cells that keep no list.  Sizes are per engine (Engine::allocate sizes them from density, cut-off and skin; they grow when too
// many cells turn out not to fit).
struct PairLists
{
    uint32_t* cand = nullptr;      // [nCell][candCap]: atom index | image code << 26 of every candidate, in tile order; padded with valid entries to a multiple of 64
    int32_t* meta = nullptr;       // [nCell][4]: {candidates T | list iterations << 12 | slices per atom << 20, cell coordinates lx | cy << 10 | cz << 20 (written once by the host),
                                   //          the cell's first atom, atoms in the cell | (65536 / slices + 1) << 12} ;
                                   //          first word -1: this cell keeps no list (more candidates than the tile holds, more than 64 atoms, or more
                                   //          iterations than iterCap) - such cells are staged in full on every step by the clean-up launch
    uint16_t* pairs = nullptr;     // [nCell][iterCap * 64]: chunks of 8 iterations x 64 lanes; an entry is the byte offset of the candidate's record in the LDS tile
                                   //          of k_pair_list: candidate k sits in record k + 1, record 0 is the far-away dummy (entry 0 = "no candidate")
    int32_t* noList = nullptr;     // [0], [1]: cells recorded without a list / cells recorded since the host last looked ; [2]: cells without a list in the lists in force
                                   //      (zeroed before every recording launch) ; [3], [4]: largest T / largest iteration count ever recorded ; [5], [6]: cells that did not
                                   //      fit the tile / the list since the host last looked ; [8..10] statistics (debug)
    int32_t candCap = 0;           // candidates per cell in `cand` (multiple of 64)
    int32_t iterCap = 0;           // list iterations per cell in `pairs` (multiple of 8)
    int32_t candLds = 0;           // candidates the LDS tiles of k_pair_list / k_build_lists hold (>= 256, <= candCap): sized by the engine from the largest T
                                   //      seen, because LDS per wave is what bounds the occupancy of k_pair_list (7.7 KiB: 95 us, 10.8 KiB: 106 us on the 1 M-atom box)
    int32_t iterLds = 0;           // iterations the builder's LDS list buffer holds (multiple of 8, <= iterCap)
    int32_t waves = 1;             // waves per cell in k_pair_list (1, 2 or 4): they share ONE tile and split the cell's atoms - where a cell's candidates are
                                   //      many and the cells few (dense systems, slab ranks), LDS per wave and wave lifetime are what bounds the kernel.
                                   //      `pairs` then holds `waves` lists per cell ([nCell][waves][iterCap * 64])
    int32_t recBytes = 0;          // bytes per LDS record in k_pair_list: 24 {x, y, z}
    int32_t entryScale = 0;        // list entry of record n: n * entryScale (the record's byte offset, or its number in the table-driven modes)
    const float4* rel = nullptr;   // [atoms] written by the sort: position relative to the centre of the atom's own cell (f32) + its cell's z index; what the builder stages
};
// does this cell walk its list?  (the same test in k_pair_list and in the clean-up launch: every cell is served by exactly one of them)
__device__ __forceinline__ bool list_usable(const PairLists& L, int header) { return header >= 0 && (header & 0xFFF) <= L.candLds; }

__device__ __forceinline__ int wave_max_int(int v)
{
#pragma unroll
    for (int o = kWave >> 1; o >= 1; o >>= 1) v = max(v, __shfl_xor(v, o, kWave));
    return v;
}

// The next step's first half, fused into the epilogue of this step's pair kernel (one GPU, plain NVE steps of a lazy run that walks pair lists): the lane that
// has just written an atom's force also completes this step's second half-kick, applies the next step's first one and drifts the atom - everything
// k_integrate1_bin<2> would do in a launch of its own (31 us on the 1 M-atom box), operation for operation: deferred kick, kick, drift, wall-crossing
// counters, longest step, displacement against the slack.  The new position goes into a SECOND set of coordinate arrays: other waves still read this step's
// positions; the engine swaps the two sets after the launch.
struct NextStep
{
    double *xn = nullptr, *yn = nullptr, *zn = nullptr;     // where the next step's positions go (nullptr: no fusion in this launch)
    RefPos R0{};                                             // positions at the last rebuild
    DevStats* st = nullptr;
    Counts* cnt = nullptr;
    // radiative thermostat (k_pair_list<..., TSTAT>): the fused epilogue closes the step the way k_integrate2_post does - second half-kick, then the thermostat
    // (post_tstat_atom, keyed by the number of the step being closed) - before it opens the next one; nullptr: plain NVE
    const double *photons = nullptr, *uvx = nullptr, *uvy = nullptr, *uvz = nullptr;
    int pendingAfter = -1;                                   // launches that do NOT fuse, in an engine that sometimes does: what k_pair_list leaves in DevStats::pendingKick
                                                             // (1: this step's second half-kick is owed to the next k_integrate1_bin / k_integrate2 ; -1: hands off)
};

// per-lane part; the wave-level bookkeeping follows in next_step_finish.  (Scalars on purpose: as two arrays of six the wall sums ended up in scratch
// memory, which cost every launch of the kernel 40 %.)
struct NextAcc { double mx, my, mz, stepLen2; int cx, cy, cz, violated; };      // wall crossed along x / y / z: -1 lower wall, +1 upper wall; m v of the crossing
__device__ __forceinline__ void next_acc_clear(NextAcc& a) { a.mx = a.my = a.mz = 0.0; a.stepLen2 = 0.0; a.cx = a.cy = a.cz = 0; a.violated = 0; }

// v holds the velocity with this step's second half-kick already applied (same operand order as k_integrate1_bin's deferred kick)
__device__ __forceinline__ void next_step_atom(const StepParams& P, const SpecTable& S, const NextStep& N, int i, int t, double x, double y, double z,
                                               double fx, double fy, double fz, double& vx, double& vy, double& vz, double rx, double ry, double rz, NextAcc& a)
{
    const double rM = S.rMhdt[t], m = S.mass[t];
    vx += rM * fx; vy += rM * fy; vz += rM * fz;                   // the next step's first half-kick (integrators.cpp:293-330 ; verlet_1stage cuMDfunc.cu:333-470)
    const int ix0 = image_of(x, P.L[0], P.invL[0]), iy0 = image_of(y, P.L[1], P.invL[1]), iz0 = image_of(z, P.L[2], P.invL[2]);
    double dx = 0.0, dy = 0.0, dz = 0.0;
    if (!S.frozen[t]) { dx = vx * P.dt; dy = vy * P.dt; dz = vz * P.dt; x += dx; y += dy; z += dz; }
    a.stepLen2 = dx * dx + dy * dy + dz * dz;
    int c;
    c = image_of(x, P.L[0], P.invL[0]) - ix0;
    if (c < 0) { a.mx = m * (-vx); a.cx = -1; } else if (c > 0) { a.mx = m * vx; a.cx = 1; }
    c = image_of(y, P.L[1], P.invL[1]) - iy0;
    if (c < 0) { a.my = m * (-vy); a.cy = -1; } else if (c > 0) { a.my = m * vy; a.cy = 1; }
    c = image_of(z, P.L[2], P.invL[2]) - iz0;
    if (c < 0) { a.mz = m * (-vz); a.cz = -1; } else if (c > 0) { a.mz = m * vz; a.cz = 1; }
    const double ex = x - rx, ey = y - ry, ez = z - rz;             // (rx, ry, rz): where the atom was when the cells were rebuilt
    if (ex * ex + ey * ey + ez * ez > P.lazySlack2) a.violated = 1;
    // per-species crossing counters (specAcBoxNeg / specAcBoxPos of put_periodic, cuMDfunc.cu:35-106): crossings are rare, plain atomics
    if (a.cx) atomicAdd(&N.st->specCross[t * 6 + (a.cx < 0 ? 0 : 1)], 1ULL);
    if (a.cy) atomicAdd(&N.st->specCross[t * 6 + (a.cy < 0 ? 2 : 3)], 1ULL);
    if (a.cz) atomicAdd(&N.st->specCross[t * 6 + (a.cz < 0 ? 4 : 5)], 1ULL);
    N.xn[i] = x; N.yn[i] = y; N.zn[i] = z;
}

__device__ __forceinline__ void next_wall_sum(double m, int c, int want, int slot, double* __restrict__ partials, int maxBlocks, size_t pb)
{
    const double s = wave_sum(c == want ? m : 0.0), b = wave_sum(c == want ? 1.0 : 0.0);
    if ((threadIdx.x & (kWave - 1)) == 0 && b != 0.0) { partials[(size_t)(PS_MOM_XN + slot) * maxBlocks + pb] += s; partials[(size_t)(PS_CNT_XN + slot) * maxBlocks + pb] += b; }
}

// once per wave, all lanes: longest step, violation flag (for the NEXT step: cycleStep + 1), wall momenta / crossings into this workgroup's partial slots
__device__ __forceinline__ void next_step_finish(const StepParams& P, const NextStep& N, NextAcc& a, double* __restrict__ partials, int maxBlocks, size_t pb)
{
    const int lane = threadIdx.x & (kWave - 1);
    double mx = a.stepLen2;
#pragma unroll
    for (int o = kWave >> 1; o >= 1; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o, kWave));
    if (lane == 0)
    {
        const unsigned long long bits = (unsigned long long)__double_as_longlong(mx);
        if (bits > __hip_atomic_load(&N.cnt->maxStep2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&N.cnt->maxStep2, bits);
    }
    if (__any(a.violated) && lane == 0)
    {
        if (N.cnt->lazyViolated == 0) N.cnt->lazyViolated = P.cycleStep + 1;
        N.cnt->lazyViolatedEver = 1;
    }
    if (__any((a.cx | a.cy | a.cz) != 0))
    {   // slots in the order Xn, Xp, Yn, Yp, Zn, Zp (PartialSlot)
        next_wall_sum(a.mx, a.cx, -1, 0, partials, maxBlocks, pb); next_wall_sum(a.mx, a.cx, 1, 1, partials, maxBlocks, pb);
        next_wall_sum(a.my, a.cy, -1, 2, partials, maxBlocks, pb); next_wall_sum(a.my, a.cy, 1, 3, partials, maxBlocks, pb);
        next_wall_sum(a.mz, a.cz, -1, 4, partials, maxBlocks, pb); next_wall_sum(a.mz, a.cz, 1, 5, partials, maxBlocks, pb);
    }
}

template <int MODE, int VDW, bool CLEANUP>
                      // MODE 0: generic (any mix, radii) ; 1: one species, Lennard-Jones only ; 2: <= 4 species, one potential family VDW, elec none|dir|Fennell ;
                      // 3: as 2 with the real-space term of the Ewald sum ; 4: one species, surk with thermostat radii (case study 2).
                      // CLEANUP: the clean-up launch behind k_pair_list - a small grid that strides over the cells and stages those without a list.
__global__ __launch_bounds__(kWave, (MODE == 2 || MODE == 3 || MODE == 5) ? 3 : 1) void k_pair_tile(StepParams P, SpecTable S, const DevPot* __restrict__ pots, AtomArrays A,
                                                     const int32_t* __restrict__ cellStart, int firstCell, int nCellsRun,
                                                     double* __restrict__ partials, int maxBlocks, const Counts* __restrict__ counts, int blockBase,
                                                     PairLists L, NextStep N, SplitArgs Z)
{
    constexpr bool onlyUnlisted = CLEANUP;           // (a template parameter: with the strided loop of the clean-up launch in it the full launch lost 7 %)
    // lazy re-sort (Engine::step): an atom has left the slack of the cell it was sorted into - until the next sort the stencil reaches one cell
    // further (rare; Engine::lazy_allowed guarantees that the wider stencil still sees every cell through one image only)
    const bool widened = P.lazySlack2 > 0.0 && P.nranks == 1 && slack_violated(P, counts);     // (a slab rank has no ghost layers to widen into: Engine reports the violation)
    if (widened)
    {   // nothing is known about how far the atoms have strayed beyond the slack (less than a cell, the host halves the interval at once): no pruning
        for (int k = 0; k < 3; k++) { P.hw[k] += 1; P.nOff[k] = 2 * P.hw[k] + 1; }
        P.pruneR2 = 1e300;
    }
    constexpr bool kOneSpecies = (MODE == 1 || MODE == 4);           // no species table, no type ids
    __shared__ double txyz[3 * kTileLds];                            // candidate coordinates RELATIVE to the centre of the centre cell
    __shared__ float tw[kTileLds];                                   // -(x^2 + y^2 + z^2) of the same, f32: 4th operand row of the filter
    double* const tx = txyz;
    double* const ty = txyz + kTileLds;
    double* const tz = txyz + 2 * kTileLds;
    __shared__ uint8_t ttyp[!kOneSpecies ? kTileLds : 1];               // species ids (< 16)
    __shared__ double trad[(MODE == 0 || MODE == 4) ? kTileLds : 1];
    __shared__ double pairTab[(MODE == 2 || MODE == 3 || MODE == 5) ? kLjSpecMax * kLjSpecMax * kPairTabStride : 1];
    __shared__ int32_t entJ[kWave], entN[kWave], entC[kWave];       // staging table: first atom, count (<= 64), image-shift code

    const int lane = threadIdx.x;
    PairAcc acc = {0, 0, 0, 0, 0, 0};
    double eV = 0.0, eC = 0.0, dropped = 0.0, eK = 0.0;
    NextAcc nacc;
    next_acc_clear(nacc);
    const DevPot lj = pots[0];
    if (MODE == 2 || MODE == 3 || MODE == 5)
    {
        const int np = P.nSpec * P.nSpec;
        if (lane < np)
        {
            const DevPot v = pots[lane];
            const int a = lane / P.nSpec, b = lane - a * P.nSpec;
            double* q = pairTab + lane * kPairTabStride;
            q[0] = v.p0; q[1] = v.p1; q[2] = v.p2; q[3] = v.p3; q[4] = v.p4;
            q[5] = v.type ? v.r2cut : -1.0;                                   // no potential for this pair: never inside the cut-off
            q[6] = (S.charged[a] && S.charged[b]) ? S.charge[a] * S.charge[b] * P.fcoul : 0.0;
            if (v.type == 2) q[3] = 1.0 / v.p1;                              // buck uses p0..p2 only: 1/rho rides in the p3 slot
            q[7] = (double)v.type;
        }
        __builtin_amdgcn_wave_barrier();
    }
    // XCD-aware mapping: workgroups b and b+8 share an XCD (and its L2); give each XCD a contiguous run of cells.  A full launch has one workgroup per
    // cell (the loop runs once); the clean-up launch behind k_pair_list (onlyUnlisted) is a small grid that strides over the cells and stages those
    // that keep no list - or all of them after a slack violation
    const int per = (nCellsRun + 7) >> 3;
    const int rowStep = (int)(gridDim.x >> 3);
    // which of this workgroup's cells need work: lane k of batch b looks at row (blockIdx >> 3) + (64 b + k) * rowStep of this XCD's share.  A full launch
    // has exactly one cell; the clean-up launch (a few workgroups that stride over all cells) takes those without a list - or, after a slack violation, all
    bool cleanupIdle = false;
    if (onlyUnlisted && !widened) cleanupIdle = L.noList[2] == 0;       // the last recording left no cell without a list: nothing to clean up
    const int nSplit = CLEANUP ? 1 : Z.n;
    const int sub = (int)(blockIdx.x >> 3) & (nSplit - 1);           // which share of the stencil's columns this wave takes (all shares of a cell on one XCD)
    for (int rowBase = (int)(blockIdx.x >> 3) / nSplit; rowBase < per && !cleanupIdle; rowBase += CLEANUP ? kWave * rowStep : per)
    {
    unsigned long long todo = 1ULL;                                  // full launch: this workgroup's one cell
    if (CLEANUP)
    {
        const int myRow = rowBase + lane * rowStep;
        const int myCr = (blockIdx.x & 7) * per + myRow;
        bool need = myRow < per && myCr < nCellsRun;
        if (!widened && need) need = !list_usable(L, L.meta[4 * (firstCell + myCr)]);
        todo = __ballot(need);
    }
    else if ((blockIdx.x & 7) * per + rowBase >= nCellsRun) todo = 0ULL;
    while (todo != 0ULL)
    {
        const int rowK = CLEANUP ? __ffsll((long long)todo) - 1 : 0;
        todo &= todo - 1ULL;
        const int cr = (blockIdx.x & 7) * per + rowBase + rowK * rowStep;
        const int cell = firstCell + cr;
        const int ncy = P.nc[1], ncz = P.nc[2];
        const int cz = cell % ncz, cy = (cell / ncz) % ncy, lx = cell / (ncy * ncz);
        const int ib = cellStart[cell], ie = cellStart[cell + 1];
        // centre of the centre cell (global coordinates) and its half edges: everything in the tile is relative to this point, which keeps
        // the magnitudes small enough for the f32 distance filter and costs nothing in accuracy (differences of cell-relative fp64
        // coordinates are exact or within one ulp of 16 A, below the ulp of a global coordinate in any box wider than 32 A)
        const double h0 = 0.5 * P.csz[0], h1 = 0.5 * P.csz[1], h2 = 0.5 * P.csz[2];
        const double cc0 = (lx + P.cx0) * P.csz[0] + h0, cc1 = cy * P.csz[1] + h1, cc2 = cz * P.csz[2] + h2;
        // f32 filter threshold: rc^2 + error bound.  Operands are rounded to f32 (2^-24 relative), products and the 4-term sum are f32:
        // |error| <= 2^-21 (|ri|^2 + |rj|^2 + 2 |ri.rj| + rc^2) with |r|^2 <= sum (h + rc)^2 - bounded here with a factor 4 to spare
        const double rprune = sqrt(P.pruneR2), rcut = rprune + 0.5 * (rprune - sqrt(P.r2Max));     // staged atoms reach rc + 2 slack from the box, the cell's own atoms slack
        double ext2 = (h0 + rcut) * (h0 + rcut) + (h1 + rcut) * (h1 + rcut) + (h2 + rcut) * (h2 + rcut);
        if (widened)
        {   // unpruned: candidates sit anywhere in the (widened) stencil
            const double w0 = (2 * P.hw[0] + 1) * h0, w1 = (2 * P.hw[1] + 1) * h1, w2 = (2 * P.hw[2] + 1) * h2;
            ext2 = w0 * w0 + w1 * w1 + w2 * w2;
        }
        const double filtThr = P.r2Max + 1.9073486328125e-06 * (4.0 * ext2 + P.r2Max);      // 2^-19
        for (int i0 = ib; i0 < ie; i0 += kWave)
        {
            const int nthis = min(kWave, ie - i0);
            const int lg = nthis <= 16 ? 4 : (nthis <= 32 ? 5 : 6);     // log2(i-slots)
            const int islots = 1 << lg;
            const int il = lane & (islots - 1), slice = lane >> lg;
            const bool validI = il < nthis;
            const int myi = i0 + il;
            double xi = 1e30, yi = 1e30, zi = 1e30, radi = 0.0;
            int ti = 0;
            if (validI)
            {
                xi = A.x[myi] - cc0; yi = A.y[myi] - cc1; zi = A.z[myi] - cc2;
                if (!kOneSpecies) ti = A.type[myi];
                if ((MODE == 0 && P.use_radii) || MODE == 4) radi = A.rad[myi];
            }
            // matrix-filter operand of this lane (used when the cell has <= 16 atoms): component `slice` of (2 xi, 2 yi, 2 zi, thr - |ri|^2).
            // Idle atom slots sit at 1e30: their column of the filter is -inf ("outside") whatever the candidate
            const float filtB = (slice == 0) ? (float)(2.0 * xi) : (slice == 1) ? (float)(2.0 * yi) : (slice == 2) ? (float)(2.0 * zi)
                                                                                                    : (float)(filtThr - (xi * xi + yi * yi + zi * zi));
            const float filtC = 0.0f;
            acc.fx = 0.0; acc.fy = 0.0; acc.fz = 0.0; acc.eV = 0.0; acc.eC = 0.0; acc.dropped = 0.0;
            int T = 0;
            // one LDS chunk = two passes per round of 96 candidates per lane:
            //   pass 1  distance tests only (7 fp64 ops per candidate); hits are recorded in per-lane bit masks
            //   pass 2  every lane pops its own hits, so the expensive potential runs on densely filled waves
            //           (about 20 % of the candidates are inside the cut-off: evaluating the potential inline would
            //            execute it for nearly every wave-iteration with 80 % of the lanes masked off)
            auto process = [&]() {
                // far-away, FINITE dummies behind the last candidate: the passes need no bounds checks, and dead lanes have a
                // harmless candidate to chew on (uninitialised LDS could hold NaN patterns: 0 * NaN would poison a force)
                if (lane < kTilePad)
                {
                    tx[T + lane] = -1e30; ty[T + lane] = 0.0; tz[T + lane] = 0.0; tw[T + lane] = -3e38f;
                    if (!kOneSpecies) ttyp[T + lane] = 0;          // a valid species: the parameter table is indexed with it
                    if (MODE == 0 || MODE == 4) trad[T + lane] = 1.0;
                }
                __builtin_amdgcn_wave_barrier();
                if (lg == 4) tile_passes<MODE, VDW, 4>(P, S, pots, lj, tx, ty, tz, tw, ttyp, trad, pairTab, T, slice, xi, yi, zi, ti, radi, filtB, filtC, acc);
                else if (lg == 5) tile_passes<MODE, VDW, 5>(P, S, pots, lj, tx, ty, tz, tw, ttyp, trad, pairTab, T, slice, xi, yi, zi, ti, radi, filtB, filtC, acc);
                else tile_passes<MODE, VDW, 6>(P, S, pots, lj, tx, ty, tz, tw, ttyp, trad, pairTab, T, slice, xi, yi, zi, ti, radi, filtB, filtC, acc);
                __builtin_amdgcn_wave_barrier();
                T = 0;
            };

            // ---- staging.  Memory latency, not bandwidth, is what staging costs (each wave would otherwise walk ~10
            // dependent s_load -> global_load round trips), so it is organised as few, wide round trips:
            //   1. the lanes look up ALL z-runs of the stencil in parallel (one lane per (x-offset, y-offset, segment));
            //   2. non-empty runs are packed into a small LDS table (<= 64 atoms per entry);
            //   3. entries are consumed in groups of up to four: all their coordinate loads are issued before the first
            //      one is used.
            const int nSegTot = P.nOff[0] * P.nOff[1] * 3;
            const int rcpOff1 = 65536 / P.nOff[1] + 1;
            for (int sg0 = 0; sg0 < nSegTot; sg0 += kWave)
            {
                int rjb = 0, rn = 0, rcode = 0;
                {
                    const int sg = sg0 + lane;
                    // sg = (ox * nOff1 + oy) * 3 + seg, decoded with multiply-shift reciprocals (operands are < 2^10)
                    const int col = (sg * 21846) >> 16, seg = sg - 3 * col;
                    const int ox = (col * rcpOff1) >> 16, oy = col - ox * P.nOff[1];
                    int nx = lx + ox - P.hw[0], cxs = 1;                      // shift codes: 0 -> -L, 1 -> 0, 2 -> +L
                    if (P.nranks > 1)
                    {   // slab window: ghost layers are resident; their coordinates are global, so shift across the seam
                        const int gx = nx + P.cx0;
                        if (gx < 0) cxs = 0; else if (gx >= P.nc[0]) cxs = 2;
                    }
                    else if (nx < 0) { nx += P.nc[0]; cxs = 0; }
                    else if (nx >= P.nc[0]) { nx -= P.nc[0]; cxs = 2; }
                    int ny = cy + oy - P.hw[1], cys = 1;
                    if (ny < 0) { ny += ncy; cys = 0; } else if (ny >= ncy) { ny -= ncy; cys = 2; }
                    const int zlo = cz - P.hw[2], zhi = cz + P.hw[2];
                    int zs, ze, czs = 1;
                    bool have = sg < nSegTot && (col & (nSplit - 1)) == sub;
                    if (seg == 0) { zs = max(zlo, 0); ze = min(zhi, ncz - 1); }
                    else if (seg == 1) { have = have && zlo < 0; zs = zlo + ncz; ze = ncz - 1; czs = 0; }
                    else { have = have && zhi >= ncz; zs = 0; ze = zhi - ncz; czs = 2; }
                    if (have)
                    {
                        const int colBase = (nx * ncy + ny) * ncz;
                        rjb = cellStart[colBase + zs];
                        rn = cellStart[colBase + ze + 1] - rjb;
                        rcode = cxs | (cys << 2) | (czs << 4);
                    }
                }
                while (__any(rn > 0))
                {
                    // one table entry (<= 64 atoms) per run that still has atoms
                    const unsigned long long emask = __ballot(rn > 0);
                    const int nEnt = __popcll(emask);
                    if (rn > 0)
                    {
                        const int pe = lanes_below(emask);
                        entJ[pe] = rjb; entN[pe] = min(rn, kWave); entC[pe] = rcode;
                        rjb += kWave; rn -= kWave;
                    }
                    __builtin_amdgcn_wave_barrier();
                    int e = 0;
                    while (e < nEnt)
                    {
                        // up to four entries at a time, as many as are sure to fit: the tile must have room for all their atoms, counted before pruning (an
                        // exact count per run, taken after the loads, chains the runs of a group one behind the other: 60-100 us on the 1 M-atom box if done always)
                        const int le = min(e + (lane & 3), nEnt - 1);
                        const int vj = entJ[le], vn = entN[le], vc = entC[le];
                        int g = 0;
                        {
                            int room = kTileCap - T;
#pragma unroll
                            for (int u = 0; u < 4; u++)
                            {
                                const int nu = __builtin_amdgcn_readlane(vn, u);
                                if (g == u && e + u < nEnt && nu <= room) { g = u + 1; room -= nu; }
                            }
                        }
                        if (g == 0)
                        {   // The next run is not sure to fit by its atom count: stage it alone and look at what the pruning really keeps of it (rare: the last
                            // runs of a crowded tile).  Only if that does not fit either is the tile full: run the passes on what is there, then go on filling
                            const int j = __builtin_amdgcn_readlane(vj, 0) + lane, n0 = __builtin_amdgcn_readlane(vn, 0), code0 = __builtin_amdgcn_readlane(vc, 0);
                            double xj = 0.0, yj = 0.0, zj = 0.0, rad0 = 0.0;
                            int typ0 = 0;
                            if (lane < n0)
                            {
                                xj = ld_f64(A.x, j); yj = ld_f64(A.y, j); zj = ld_f64(A.z, j);
                                if (!kOneSpecies) typ0 = ld_i32(A.type, j);
                                if ((MODE == 0 && P.use_radii) || MODE == 4) rad0 = ld_f64(A.rad, j);
                            }
                            if (code0 != 0x15)
                            {
                                const int c0 = code0 & 3, c1 = (code0 >> 2) & 3, c2 = (code0 >> 4) & 3;
                                xj += c0 == 0 ? -P.L[0] : (c0 == 2 ? P.L[0] : 0.0);
                                yj += c1 == 0 ? -P.L[1] : (c1 == 2 ? P.L[1] : 0.0);
                                zj += c2 == 0 ? -P.L[2] : (c2 == 2 ? P.L[2] : 0.0);
                            }
                            xj -= cc0; yj -= cc1; zj -= cc2;
                            const double bx = fmax(fabs(xj) - h0, 0.0), by = fmax(fabs(yj) - h1, 0.0), bz = fmax(fabs(zj) - h2, 0.0);
                            const bool keep = (lane < n0) && (bx * bx + by * by + bz * bz) <= P.pruneR2;
                            const unsigned long long mask = __ballot(keep);
                            const int nk = __popcll(mask);
                            if (T + nk > kTileCap) { process(); continue; }
                            if (keep)
                            {
                                const int pp = T + lanes_below(mask);
                                tx[pp] = xj; ty[pp] = yj; tz[pp] = zj;
                                tw[pp] = -(float)(xj * xj + yj * yj + zj * zj);
                                if (!kOneSpecies) ttyp[pp] = (uint8_t)typ0;
                                if (MODE == 0 || MODE == 4) trad[pp] = rad0;
                            }
                            T += nk;
                            e += 1;
                            continue;
                        }
                        double gx[4], gy[4], gz[4], grad[4];
                        int gtyp[4], gjn[4], gcode[4];
#pragma unroll
                        for (int u = 0; u < 4; u++)
                        {
                            const int j = __builtin_amdgcn_readlane(vj, u) + lane;
                            gjn[u] = (u < g) ? __builtin_amdgcn_readlane(vn, u) : 0;
                            gcode[u] = __builtin_amdgcn_readlane(vc, u);
                            gx[u] = gy[u] = gz[u] = 0.0; grad[u] = 0.0; gtyp[u] = 0;
                            if (lane < gjn[u])
                            {
                                gx[u] = ld_f64(A.x, j); gy[u] = ld_f64(A.y, j); gz[u] = ld_f64(A.z, j);
                                if (!kOneSpecies) gtyp[u] = ld_i32(A.type, j);
                                if ((MODE == 0 && P.use_radii) || MODE == 4) grad[u] = ld_f64(A.rad, j);
                            }
                        }
#pragma unroll
                        for (int u = 0; u < 4; u++)
                        {
                            if (u < g)
                            {
                                double xj = gx[u], yj = gy[u], zj = gz[u];
                                if (gcode[u] != 0x15)
                                {   // this run is a periodic image (wave-uniform branch; only boundary cells take it)
                                    const int c0 = gcode[u] & 3, c1 = (gcode[u] >> 2) & 3, c2 = (gcode[u] >> 4) & 3;
                                    xj += c0 == 0 ? -P.L[0] : (c0 == 2 ? P.L[0] : 0.0);
                                    yj += c1 == 0 ? -P.L[1] : (c1 == 2 ? P.L[1] : 0.0);
                                    zj += c2 == 0 ? -P.L[2] : (c2 == 2 ? P.L[2] : 0.0);
                                }
                                xj -= cc0; yj -= cc1; zj -= cc2;
                                // distance from the centre cell's box: atoms farther than the cut-off cannot reach any atom in it
                                const double bx = fmax(fabs(xj) - h0, 0.0);
                                const double by = fmax(fabs(yj) - h1, 0.0);
                                const double bz = fmax(fabs(zj) - h2, 0.0);
                                const bool keep = (lane < gjn[u]) && (bx * bx + by * by + bz * bz) <= P.pruneR2;
                                const unsigned long long mask = __ballot(keep);
                                if (keep)
                                {
                                    const int pp = T + lanes_below(mask);
                                    tx[pp] = xj; ty[pp] = yj; tz[pp] = zj;
                                    tw[pp] = -(float)(xj * xj + yj * yj + zj * zj);
                                    if (!kOneSpecies) ttyp[pp] = (uint8_t)gtyp[u];
                                    if (MODE == 0 || MODE == 4) trad[pp] = grad[u];
                                }
                                T += __popcll(mask);
                            }
                        }
                        e += g;
                    }
                    __builtin_amdgcn_wave_barrier();
                }
            }
            process();

            // fold the j-slices (fixed order) and write the force: clear_force + pair sums
            for (int o = kWave >> 1; o >= islots; o >>= 1)
            {
                acc.fx += __shfl_xor(acc.fx, o, kWave);
                acc.fy += __shfl_xor(acc.fy, o, kWave);
                acc.fz += __shfl_xor(acc.fz, o, kWave);
            }
            if (nSplit > 1)
            {   // this wave saw a share of the stencil only: deliver the partial force (finished below by the last share to arrive)
                if (validI && slice == 0)
                {
                    const size_t o = (size_t)sub * Z.capacity + myi;
                    // (device-coherent stores and, below, loads: they bypass the caches that are not coherent between CUs / XCDs, so that no fence - which
                    //  on this chip writes the whole L2 back: 40 us per wave - is needed around the arrival count)
                    __hip_atomic_store(&Z.fx[o], acc.fx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(&Z.fy[o], acc.fy, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(&Z.fz[o], acc.fz, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            else if (validI && slice == 0)
            {
                double q = 0.0;
                if (!kOneSpecies) q = S.charge[ti];
                const double fxi = -q * P.E[0] + acc.fx;   // clear_force integrators.cpp:17-39
                const double fyi = -q * P.E[1] + acc.fy;
                const double fzi = -q * P.E[2] + acc.fz;
                A.fx[myi] = fxi; A.fy[myi] = fyi; A.fz[myi] = fzi;
                if (P.fuseKick || (CLEANUP && N.xn))
                {   // second half-kick + kinetic energy of integrate2 (integrators.cpp:486-531 ; verlet_2stage cuMDfunc.cu:521-600),
                    // fused here on plain NVE steps: the force is still in registers
                    const double rM = S.rMhdt[ti], m = S.mass[ti];
                    double vx = A.vx[myi] + rM * fxi, vy = A.vy[myi] + rM * fyi, vz = A.vz[myi] + rM * fzi;
                    if (P.fuseKick) eK += (vx * vx + vy * vy + vz * vz) * m;
                    if (CLEANUP && N.xn) next_step_atom(P, S, N, myi, ti, A.x[myi], A.y[myi], A.z[myi], fxi, fyi, fzi, vx, vy, vz, N.R0.x[myi], N.R0.y[myi], N.R0.z[myi], nacc);      // (NextStep)
                    A.vx[myi] = vx; A.vy[myi] = vy; A.vz[myi] = vz;
                }
            }
            eV += acc.eV; eC += acc.eC; dropped += acc.dropped;
        }
        if (nSplit > 1)
        {
            // Hand-off without fences (a release / acquire pair at agent scope makes gfx950 write the whole L2 back: 40 us per wave, measured).  Why the weaker
            // form is safe HERE: the partial forces are agent-scope atomic stores (they go to the level all CUs see, not to a CU-private cache); a wave issues
            // in order and s_waitcnt vmcnt(0) holds it until every one of those stores has been acknowledged by that level, so the arrival count cannot
            // go up before they are visible; the last arriver reads them back with agent-scope atomic loads, which do not hit in its own non-coherent caches.
            // The compiler cannot move the stores past the asm (memory clobber) nor the loads above the atomic's result they depend on (control dependence
            // through `before`).
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the partial forces above have reached the coherent level before the count goes up
            int before = 0;
            if (lane == 0) before = atomicAdd(&Z.arrived[cell], 1);
            before = __shfl(before, 0, kWave);
            if (before == nSplit - 1)
            {   // every share is in: add them up in the order of the shares, and finish the cell's atoms as the one-wave form does
                if (lane == 0) __hip_atomic_store(&Z.arrived[cell], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                for (int i = ib + lane; i < ie; i += kWave)
                {
                    double fx = 0.0, fy = 0.0, fz = 0.0;
                    for (int q = 0; q < nSplit; q++)
                    {
                        const size_t o = (size_t)q * Z.capacity + i;
                        fx += __hip_atomic_load(&Z.fx[o], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        fy += __hip_atomic_load(&Z.fy[o], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        fz += __hip_atomic_load(&Z.fz[o], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    int ti = 0;
                    double qi = 0.0;
                    if (!kOneSpecies) { ti = A.type[i]; qi = S.charge[ti]; }
                    const double fxi = -qi * P.E[0] + fx, fyi = -qi * P.E[1] + fy, fzi = -qi * P.E[2] + fz;
                    A.fx[i] = fxi; A.fy[i] = fyi; A.fz[i] = fzi;
                    if (P.fuseKick)
                    {
                        const double rM = S.rMhdt[ti], m = S.mass[ti];
                        const double vx = A.vx[i] + rM * fxi, vy = A.vy[i] + rM * fyi, vz = A.vz[i] + rM * fzi;
                        A.vx[i] = vx; A.vy[i] = vy; A.vz[i] = vz;
                        eK += (vx * vx + vy * vy + vz * vz) * m;
                    }
                }
            }
        }
    }
    }   // batches of rows
    eV = wave_sum(eV); eC = wave_sum(eC); dropped = wave_sum(dropped);
    if (lane == 0)
    {
        // (blockBase: split launches - interior / boundary cells of a slab rank, the clean-up launch behind k_pair_list - book into their own runs of slots)
        const size_t pb = (size_t)blockBase + blockIdx.x;
        partials[(size_t)PS_EVDW * maxBlocks + pb] = eV;
        partials[(size_t)PS_ECOUL * maxBlocks + pb] = eC;
        if (dropped != 0.0) partials[(size_t)PS_DROPPED * maxBlocks + pb] += dropped;
    }
    if (P.fuseKick)
    {
        eK = wave_sum(eK);
        if (lane == 0) partials[(size_t)PS_EKIN * maxBlocks + blockBase + blockIdx.x] = 0.5 * eK;
    }
    if (CLEANUP && N.xn) next_step_finish(P, N, nacc, partials, maxBlocks, (size_t)blockBase + blockIdx.x);
}

// a run of cells for one launch: first cell, number of cells, first partial-sum slot; n < 0: all the cells this rank owns
struct PairRange { int first = 0, n = -1, blockBase = 0; };
inline int pair_range_grid(int nCells) { return 8 * ((nCells + 7) / 8); }
constexpr int kCleanupGrid = 4096;             // workgroups of the clean-up launch behind k_pair_list: as many as are resident at once (16 per CU).  It has work to do
                                               // only for cells without a list and on the rare steps after a slack violation - then it stages every cell at the
                                               // staging kernel's full speed; idle it costs 5.1 us on MI355X (256 workgroups: 4.9)
inline int pair_cleanup_grid(int nCells) { return std::min(pair_range_grid(nCells), kCleanupGrid); }
inline void pair_range_default(const StepParams& P, PairRange& R)
{
    const int plane = P.nc[1] * P.nc[2];
    if (R.n < 0) { R.n = pair_tile_cells(P); R.first = (P.nranks > 1) ? P.hw[0] * plane : 0; R.blockBase = 0; }
}

// listMode 0: stage every cell ; 2: clean-up launch - a small grid that stages the cells without a list
template <int MODE, int VDW>
inline void launch_pair_tile_as(const StepParams& P, const SpecTable& S, const DevPot* pots, AtomArrays A, const Counts* cnt, const int32_t* cellStart, double* partials,
                                int maxBlocks, hipStream_t stream, PairRange R, PairLists L, int listMode, NextStep N, SplitArgs Z)
{
    pair_range_default(P, R);
    if (R.n == 0) return;
    if (listMode == 2)
        hipLaunchKernelGGL((k_pair_tile<MODE, VDW, true>), dim3(pair_cleanup_grid(R.n)), dim3(kWave), 0, stream, P, S, pots, A, cellStart, R.first, R.n, partials, maxBlocks,
                           cnt, R.blockBase, L, N, SplitArgs());
    else
        hipLaunchKernelGGL((k_pair_tile<MODE, VDW, false>), dim3(pair_range_grid(R.n) * Z.n), dim3(kWave), 0, stream, P, S, pots, A, cellStart, R.first, R.n, partials,
                           maxBlocks, cnt, R.blockBase, L, NextStep(), Z);
}

// dispatch on the potential set.  P.pad1 == 2: every defined pair potential belongs to the family P.vdwFamily (1 lnjs, 2 buck, 3 p746, 4 bmhs; 5 = a mix of
// them, selected per species pair), <= 4 species, no radii, electrostatics none / direct / Fennell / Ewald with alpha rReal <= 4 (Engine::Engine decides)
#define AZTOT_PAIR_DISPATCH(LAUNCH, ...)                                                                   \
    do {                                                                                                   \
        if (P.single_lj) { LAUNCH<1, 1>(__VA_ARGS__); return; }                                            \
        if (P.pad1 == 4) { LAUNCH<4, 7>(__VA_ARGS__); return; }     /* one species, surk + radii */        \
        if (P.pad1 == 2)                                                                                   \
        {                                                                                                  \
            const bool ew = P.elec_type == 2;                                                              \
            switch (P.vdwFamily)                                                                           \
            {                                                                                              \
            case 1: if (ew) LAUNCH<3, 1>(__VA_ARGS__); else if (P.elec_type == 3) LAUNCH<5, 1>(__VA_ARGS__); else LAUNCH<2, 1>(__VA_ARGS__); return; \
            case 2: if (ew) LAUNCH<3, 2>(__VA_ARGS__); else if (P.elec_type == 3) LAUNCH<5, 2>(__VA_ARGS__); else LAUNCH<2, 2>(__VA_ARGS__); return; \
            case 3: if (ew) LAUNCH<3, 3>(__VA_ARGS__); else if (P.elec_type == 3) LAUNCH<5, 3>(__VA_ARGS__); else LAUNCH<2, 3>(__VA_ARGS__); return; \
            case 4: if (ew) LAUNCH<3, 4>(__VA_ARGS__); else if (P.elec_type == 3) LAUNCH<5, 4>(__VA_ARGS__); else LAUNCH<2, 4>(__VA_ARGS__); return; \
            case 5: if (ew) LAUNCH<3, 5>(__VA_ARGS__); else if (P.elec_type == 3) LAUNCH<5, 5>(__VA_ARGS__); else LAUNCH<2, 5>(__VA_ARGS__); return; \
            case 6: if (ew) LAUNCH<3, 6>(__VA_ARGS__); else if (P.elec_type == 3) LAUNCH<5, 6>(__VA_ARGS__); else LAUNCH<2, 6>(__VA_ARGS__); return; \
            }                                                                                              \
        }                                                                                                  \
        LAUNCH<0, 0>(__VA_ARGS__);                                                                         \
    } while (0)

inline void launch_pair_tile(const StepParams& P, const SpecTable& S, const DevPot* pots, AtomArrays A, const Counts* cnt, const int32_t* cellStart,
                             double* partials, int maxBlocks, hipStream_t stream, PairRange R = PairRange(), PairLists L = PairLists(), int listMode = 0,
                             NextStep N = NextStep(), SplitArgs Z = SplitArgs())
{
    if (!L.cand) listMode = 0;
    AZTOT_PAIR_DISPATCH(launch_pair_tile_as, P, S, pots, A, cnt, cellStart, partials, maxBlocks, stream, R, L, listMode, N, Z);
}

}  // namespace aztot
