// Pair-force kernel of a lazy run (every step: plain steps and - after the lists have been made - the rebuild steps too): one wave64 per cell,
// driven by the lists the last rebuild of the cells recorded.
//
// Replaces the reference's cell_list5a + cell_list4b_noshared + pair_1 (cuPairs.cu:2266,1474,117).  The reference rebuilds everything every step
// (main.cu:300-326); here the step that rebuilds the cells (BUILD instantiation of k_pair_tile, pair_tile.hip.h) leaves two lists per cell:
//   * the candidates of the cell's tile (atom index + periodic image code), and
//   * for every atom of the cell the candidates within rc + 2 slack, dealt round-robin to the lanes that serve the atom.
// Until the next rebuild atoms keep their slots and nobody moves farther than the slack (checked every step by whoever integrates; a violation
// makes this kernel stand down and the clean-up launch of k_pair_tile stage everything with a wider stencil), so a step is
//   1. gather the candidates into LDS (coordinates relative to the cell centre, as in k_pair_tile: same numbers, same arithmetic);
//   2. every lane walks ITS list: entry -> LDS byte offset -> exact r^2 <= rc^2 test -> potential (pair_body, shared with k_pair_tile);
//   3. optionally (NextStep, small systems) the epilogue opens the next step: second half-kick, first half-kick, drift.
// No run table, no pruning, no compaction, no distance filter, no bit masks; and because an atom's partners were dealt evenly, the lanes of a
// wave finish together (the mask-popping loop of k_pair_tile runs max-over-lanes = 21 iterations for a mean of 14 on the 1 M-atom box; here 14).
// Forces: written once per atom, no atomics, fixed summation order => bit-reproducible.  HBM traffic: the lists are streamed once per step
// (coalesced: 4 B per candidate, 2 B per pair entry), which is what buys the 3x in vector instructions.
#pragma once
#include "pair_tile.hip.h"

namespace aztot {

// ENG = false: the launch books no energies (steps whose energies nobody can see: all but the last step of an aztot_step call - the statistics are those of
// the last step, finish_steps; the reference prints them every `stat` steps, cuStat.cu:308-330).  Forces are the same instructions either way: the energy
// terms feed nothing else, the compiler drops them and their two wave reductions (C4: 603 -> 540 vector instructions per cell).
template <int MODE, int VDW, bool ENG>
__global__ __launch_bounds__(kWave) void k_pair_list(StepParams P, SpecTable S, const DevPot* __restrict__ pots, AtomArrays A,
                                                     const int32_t* __restrict__ cellStart, int firstCell, int nCellsRun,
                                                     double* __restrict__ partials, int maxBlocks, const Counts* __restrict__ counts, int blockBase, PairLists L,
                                                     NextStep N)
{
    constexpr bool kOneSpecies = (MODE == 1 || MODE == 4);           // no species table, no type ids
    constexpr bool kRadii = (MODE == 0 || MODE == 4);
    __shared__ double txyz[3 * kTileLds];                            // candidate coordinates RELATIVE to the centre of the centre cell
    __shared__ uint8_t ttyp[!kOneSpecies ? kTileLds : 1];            // species ids (< 16)
    __shared__ double trad[kRadii ? kTileLds : 1];
    __shared__ double pairTab[(MODE == 2 || MODE == 3 || MODE == 5) ? kLjSpecMax * kLjSpecMax * kPairTabStride : 1];

    const int lane = threadIdx.x;
    const int per = (nCellsRun + 7) >> 3;
    const int cr = (blockIdx.x & 7) * per + (blockIdx.x >> 3);       // XCD-aware: each XCD owns a contiguous eighth of the cells
    double eV = 0.0, eC = 0.0, dropped = 0.0, eK = 0.0;
    NextAcc nacc;                                                    // (touched only in launches that fuse the next step: no initialisation on the others)
    if (N.xn) next_acc_clear(nacc);
    // after a slack violation (one GPU) the clean-up launch stages every cell with the wider stencil: stand down
    const bool violated = P.nranks == 1 && slack_violated(P, counts);
    // everything that depends on the cell number only is requested at once, before anything is known about the cell (the loads stay inside the
    // arrays whatever they return): list header, the five groups of candidate entries, the lane's first two list chunks and entry count.  A wave's
    // life is then two memory round trips (these, then the coordinates) and the loop
    const int cell = firstCell + min(cr, nCellsRun - 1);
    const uint32_t* const myList = L.cand + (size_t)cell * kTileCap + lane;
    const uint4* const pl = (const uint4*)(L.pairs + (size_t)cell * kListStride16) + lane;
    constexpr int kRounds = kTileCap / kWave;
    uint32_t ent[kRounds];
#pragma unroll
    for (int u = 0; u < kRounds; u++) ent[u] = myList[u * kWave];
    uint4 w = pl[0];
    uint4 w1 = pl[kWave];                                           // (the second chunk too: 16 iterations cover a liquid's cells, and a chunk asked for only
                                                                    //  8 iterations ahead arrives late)
    const int2 mx = ((const int2*)L.meta)[cell];                   // {list header, cell coordinates lx | cy << 10 | cz << 20}: one scalar load, no integer divisions
    int meta = mx.x;
    if (violated || cr >= nCellsRun) meta = -1;
    if (meta > 0)
    {
        const int T = meta & 0xFFF, nIter = meta >> 12;
        const int ncy = P.nc[1], ncz = P.nc[2];
        const int lx = mx.y & 1023, cy = (mx.y >> 10) & 1023, cz = (mx.y >> 20) & 1023;
        const int ib = cellStart[cell], ie = cellStart[cell + 1];
        const double cc0 = (lx + P.cx0) * P.csz[0] + 0.5 * P.csz[0], cc1 = cy * P.csz[1] + 0.5 * P.csz[1], cc2 = cz * P.csz[2] + 0.5 * P.csz[2];
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
        }
        // ---- the cell's own atoms: lanes are (atom slot, slice) as in k_pair_tile
        const int nthis = ie - ib;                                     // <= 64 (cells with more keep no list)
        const int lg = nthis <= 16 ? 4 : (nthis <= 32 ? 5 : 6);
        const int islots = 1 << lg;
        const int il = lane & (islots - 1), slice = lane >> lg;
        const bool validI = il < nthis;
        const int myi = ib + il;
        // (every lane loads - idle atom slots from the cell's first atom - so that the three loads travel together; a conditional load made the compiler
        //  wait for each of them in turn)
        const int myl = validI ? myi : ib;
        const double xr = A.x[myl], yr = A.y[myl], zr = A.z[myl];
        // NextStep: what the epilogue will need is asked for now - loaded there, at the end of the wave's life, it cost a memory round trip per wave (+30 us)
        double v0x, v0y, v0z, r0x, r0y, r0z;                         // (defined exactly where they are used: with fuseKick / NextStep)
        if (P.fuseKick || N.xn) { v0x = A.vx[myl]; v0y = A.vy[myl]; v0z = A.vz[myl]; }
        if (N.xn) { r0x = N.R0.x[myl]; r0y = N.R0.y[myl]; r0z = N.R0.z[myl]; }
        double radi = 0.0;
        int ti = 0;
        if (!kOneSpecies) ti = A.type[myl];
        if ((MODE == 0 && P.use_radii) || MODE == 4) radi = A.rad[myl];
        // ---- gather the candidates (groups of 64; the record padded the last group with a valid atom).  Four groups are gathered whatever T is (stale
        // entries are atom indices too: the array starts out zeroed and only indices are ever written), the fifth when T > 256; all loads travel together
        {
            const int gx0 = lx + P.cx0;
            // does any neighbour cell lie across a periodic boundary (or the seam of a slab ring)?  wave-uniform
            const bool images = gx0 - P.hw[0] < 0 || gx0 + P.hw[0] >= P.nc[0] || cy - P.hw[1] < 0 || cy + P.hw[1] >= ncy || cz - P.hw[2] < 0 || cz + P.hw[2] >= ncz;
            struct Cand { double x, y, z, rad; int typ; };
            auto fetch = [&](uint32_t e) {
                Cand c;
                const int j = (int)(e & 0x3FFFFFFu);
                c.x = ld_f64(A.x, j); c.y = ld_f64(A.y, j); c.z = ld_f64(A.z, j);
                c.typ = kOneSpecies ? 0 : ld_i32(A.type, j);
                c.rad = ((MODE == 0 && P.use_radii) || MODE == 4) ? ld_f64(A.rad, j) : 0.0;
                return c;
            };
            auto put = [&](int u, uint32_t e, const Cand& c) {
                double xj = c.x, yj = c.y, zj = c.z;
                if (images)
                {   // image code per axis: 0 -> -L, 1 -> 0, 2 -> +L (exact: the product is +-L or 0)
                    xj += (double)((int)((e >> 26) & 3u) - 1) * P.L[0];
                    yj += (double)((int)((e >> 28) & 3u) - 1) * P.L[1];
                    zj += (double)((int)((e >> 30) & 3u) - 1) * P.L[2];
                }
                const int pq = u * kWave + lane;
                txyz[pq] = xj - cc0; txyz[kTileLds + pq] = yj - cc1; txyz[2 * kTileLds + pq] = zj - cc2;
                if (!kOneSpecies) ttyp[pq] = (uint8_t)c.typ;
                if (kRadii) trad[pq] = c.rad;
            };
            const Cand c0 = fetch(ent[0]), c1 = fetch(ent[1]), c2 = fetch(ent[2]), c3 = fetch(ent[3]);
            if (T > 4 * kWave)
            {
                const Cand c4 = fetch(ent[4]);
                put(0, ent[0], c0); put(1, ent[1], c1); put(2, ent[2], c2); put(3, ent[3], c3); put(4, ent[4], c4);
            }
            else { put(0, ent[0], c0); put(1, ent[1], c1); put(2, ent[2], c2); put(3, ent[3], c3); }
        }
        const double xi = validI ? xr - cc0 : 1e30, yi = validI ? yr - cc1 : 1e30, zi = validI ? zr - cc2 : 1e30;
        // the dummy candidate the unused list entries point at (kListDummy): far outside any cut-off, with a species and a radius the potentials can digest
        if (lane == 0)
        {
            txyz[kTileLds - 1] = 1e30; txyz[2 * kTileLds - 1] = 1e30; txyz[3 * kTileLds - 1] = 1e30;
            if (!kOneSpecies) ttyp[kTileLds - 1] = 0;
            if (kRadii) trad[kTileLds - 1] = 1.0;
        }
        __builtin_amdgcn_wave_barrier();

        // ---- every lane walks its list
        PairAcc acc = {0, 0, 0, 0, 0, 0};
        int nDropHalf = 0;
        const double ljDropR2 = P.ljDropR2;
        const PairHot hot = (MODE == 2 || MODE == 3 || MODE == 5) ? pair_hot_in_vgprs(P) : pair_hot(P);
        const char* const tb = (const char*)txyz;
        const int nChunks = (nIter + 7) >> 3;
        // software-pipelined: the candidate of iteration t + 1 is read from LDS before the potential of iteration t is evaluated (entries behind a lane's
        // last one point at the dummy - the record fills the list buffer with kListDummy - so the read ahead is always a valid one)
        double xj, yj, zj, radj = 0.0;
        int tj = 0;
        auto fetch = [&](uint32_t ko, double& x, double& y, double& z, int& ty, double& rd) {
            x = *(const double*)(tb + ko);
            y = *(const double*)(tb + ko + kTileLds * 8);
            z = *(const double*)(tb + ko + 2 * kTileLds * 8);
            if (!kOneSpecies) ty = ttyp[ko >> 3];
            if (kRadii) rd = *(const double*)((const char*)trad + ko);
        };
        fetch(nIter > 0 ? (w.x & 0xFFFFu) : kListDummy, xj, yj, zj, tj, radj);
        for (int c = 0; c < nChunks; c++)
        {
            uint4 wn = w1;
            if (c + 2 < nChunks) w1 = pl[(c + 2) * kWave];
            const uint32_t ww[5] = {w.x, w.y, w.z, w.w, wn.x};
#pragma unroll
            for (int u = 0; u < 8; u++)
            {
                const int t = c * 8 + u;
                if (t >= nIter) break;                                  // wave-uniform
                const uint32_t kn = ((u + 1) & 1) ? (ww[(u + 1) >> 1] >> 16) : (ww[(u + 1) >> 1] & 0xFFFFu);      // byte offset of the next candidate in the tile
                double xn, yn, zn, radn = 0.0;
                int tn = 0;
                fetch(kn, xn, yn, zn, tn, radn);
                {   // (no test of the lane's own entry count: behind its last entry a lane finds dummies)
                    const double dx = xi - xj, dy = yi - yj, dz = zi - zj;
                    const double r2 = dx * dx + dy * dy + dz * dz;
                    pair_body<MODE, VDW, true>(P, S, pots, lj, pairTab, true, dx, dy, dz, r2, ti, tj, radi, radj, ljDropR2, nDropHalf, acc, hot);
                }
                xj = xn; yj = yn; zj = zn; tj = tn; radj = radn;
            }
            w = wn;
        }
        if (MODE != 0) acc.dropped += 0.5 * (double)nDropHalf;

        // fold the j-slices (fixed order) and write the force: clear_force + pair sums
        for (int o = kWave >> 1; o >= islots; o >>= 1)
        {
            acc.fx += __shfl_xor(acc.fx, o, kWave);
            acc.fy += __shfl_xor(acc.fy, o, kWave);
            acc.fz += __shfl_xor(acc.fz, o, kWave);
        }
        if (validI && slice == 0)
        {
            double q = 0.0;
            if (!kOneSpecies) q = S.charge[ti];
            const double fxi = -q * P.E[0] + acc.fx;   // clear_force integrators.cpp:17-39
            const double fyi = -q * P.E[1] + acc.fy;
            const double fzi = -q * P.E[2] + acc.fz;
            A.fx[myi] = fxi; A.fy[myi] = fyi; A.fz[myi] = fzi;
            if (P.fuseKick || N.xn)
            {   // second half-kick + kinetic energy of integrate2 (integrators.cpp:486-531 ; verlet_2stage cuMDfunc.cu:521-600); with NextStep also the
                // next step's k_integrate1_bin<2>
                const double rM = S.rMhdt[ti], m = S.mass[ti];
                double vx = v0x + rM * fxi, vy = v0y + rM * fyi, vz = v0z + rM * fzi;
                if (ENG && P.fuseKick) eK += (vx * vx + vy * vy + vz * vz) * m;     // (the kinetic energy too is looked at after a call's last step only)
                if (N.xn) next_step_atom(P, S, N, myi, ti, xr, yr, zr, fxi, fyi, fzi, vx, vy, vz, r0x, r0y, r0z, nacc);
                A.vx[myi] = vx; A.vy[myi] = vy; A.vz[myi] = vz;
            }
        }
        if (ENG) { eV = acc.eV; eC = acc.eC; }
        dropped = acc.dropped;
    }
    // (the reductions are ~14 vector instructions each: skipped where the sum is known to be zero)
    if (ENG) eV = wave_sum(eV);
    if (ENG && MODE != 1 && MODE != 4) eC = wave_sum(eC);
    if (__any(dropped != 0.0)) dropped = wave_sum(dropped);
    if (lane == 0)
    {
        const size_t pb = (size_t)blockBase + blockIdx.x;
        if (ENG)
        {
            partials[(size_t)PS_EVDW * maxBlocks + pb] = eV;
            partials[(size_t)PS_ECOUL * maxBlocks + pb] = eC;
        }
        if (dropped != 0.0) partials[(size_t)PS_DROPPED * maxBlocks + pb] += dropped;
    }
    if (ENG && P.fuseKick)
    {
        eK = wave_sum(eK);
        if (lane == 0) partials[(size_t)PS_EKIN * maxBlocks + blockBase + blockIdx.x] = 0.5 * eK;
    }
    if (N.xn)
    {
        next_step_finish(P, N, nacc, partials, maxBlocks, (size_t)blockBase + blockIdx.x);
        // the step counter of the step being opened (main.cpp:92), and: its second half-kick is NOT owed to the next k_integrate1_bin
        if (blockIdx.x == 0 && lane == 0) { N.st->step += 1; N.st->pendingKick = 0; }
    }
    else if (N.pendingAfter >= 0 && blockIdx.x == 0 && lane == 0) N.st->pendingKick = N.pendingAfter;
}

template <int MODE, int VDW>
inline void launch_pair_list_as(const StepParams& P, const SpecTable& S, const DevPot* pots, AtomArrays A, const Counts* cnt, const int32_t* cellStart, double* partials,
                                int maxBlocks, hipStream_t stream, PairRange R, PairLists L, NextStep N, bool energies)
{
    if (energies)
        hipLaunchKernelGGL((k_pair_list<MODE, VDW, true>), dim3(pair_range_grid(R.n)), dim3(kWave), 0, stream, P, S, pots, A, cellStart, R.first, R.n, partials, maxBlocks,
                           cnt, R.blockBase, L, N);
    else
        hipLaunchKernelGGL((k_pair_list<MODE, VDW, false>), dim3(pair_range_grid(R.n)), dim3(kWave), 0, stream, P, S, pots, A, cellStart, R.first, R.n, partials, maxBlocks,
                           cnt, R.blockBase, L, N);
}

// a plain step: the list kernel for every cell (launch_pair_list), then the clean-up launch of the staging kernel for the cells that keep no list
// (launch_pair_cleanup; it books into the partial-sum slots behind the list kernel's).  Each returns the number of partial-sum slots it uses.
inline int launch_pair_list(const StepParams& P, const SpecTable& S, const DevPot* pots, AtomArrays A, const Counts* cnt, const int32_t* cellStart,
                            double* partials, int maxBlocks, hipStream_t stream, PairRange R, PairLists L, NextStep N = NextStep(), bool energies = true)
{
    pair_range_default(P, R);
    if (R.n == 0) return 0;
    auto list = [&]() { AZTOT_PAIR_DISPATCH(launch_pair_list_as, P, S, pots, A, cnt, cellStart, partials, maxBlocks, stream, R, L, N, energies); };
    list();
    return pair_range_grid(R.n);
}

inline int launch_pair_cleanup(const StepParams& P, const SpecTable& S, const DevPot* pots, AtomArrays A, const Counts* cnt, const int32_t* cellStart,
                               double* partials, int maxBlocks, hipStream_t stream, PairRange R, PairLists L, NextStep N = NextStep())
{
    pair_range_default(P, R);
    if (R.n == 0) return 0;
    R.blockBase += pair_range_grid(R.n);
    launch_pair_tile(P, S, pots, A, cnt, cellStart, partials, maxBlocks, stream, R, L, 2, N);
    return pair_cleanup_grid(R.n);
}

}  // namespace aztot
