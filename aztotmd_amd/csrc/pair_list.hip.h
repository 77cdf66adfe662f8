// Pair lists of the lazy re-sort: the kernel that builds them on the step that rebuilds the cells (k_build_lists) and the pair-force kernel that walks
// them on every step of a lazy run (k_pair_list).  One wave64 per cell in both.
//
// Replaces the reference's cell_list5a + cell_list4b_noshared + pair_1 (cuPairs.cu:2266,1474,117).  The reference rebuilds everything every step
// (main.cu:300-326); here the cells are rebuilt every K-th step only (Engine::step) and the rebuild leaves two lists per cell:
//   * the candidates: every atom of the cell's stencil within rc + skin of the cell's box (atom index + periodic image code, tile order), and
//   * for every atom of the cell its partners among them - the candidates within rc + skin of the atom -, dealt round-robin to the lanes that
//     serve the atom.  An entry is the byte offset of the candidate's record in the LDS tile of k_pair_list.
// Until the next rebuild atoms keep their slots and nobody moves farther than skin / 2 (checked every step by whoever integrates; a violation makes
// k_pair_list stand down and the clean-up launch of k_pair_tile stage everything with a wider stencil), so no pair inside rc can be missing and a step is
//   1. gather the candidates into LDS (coordinates relative to the cell centre: same numbers, same arithmetic as k_pair_tile);
//   2. every lane walks ITS list: entry -> LDS record -> exact r^2 <= rc^2 test -> potential (pair_body, shared with k_pair_tile);
//   3. optionally (NextStep, small systems) the epilogue opens the next step: second half-kick, first half-kick, drift.
// Lanes are (atom slot, slice) with slot = lane / NS and NS = min(8, 64 / atoms in the cell) slices per atom: 13-16 atoms run 4 slices each, 17-21
// three, 22-32 two - whatever the cell holds, at least two thirds of the lanes work (the power-of-two layout of round 2 fell to half at 17 atoms, which
// is what a liquid's cells of rc + skin hold).  Slices of one atom are neighbouring lanes: they are folded with lane shifts in a fixed order.
// Forces: written once per atom, no atomics, fixed summation order => bit-reproducible.  All list sizes are per engine (PairLists::candCap / iterCap,
// dynamic LDS), so dense systems (hundreds of partners per atom, a thousand candidates per cell) walk lists too.
#pragma once
#include <type_traits>
#include "pair_tile.hip.h"

namespace aztot {

constexpr int kListPreload = 5;            // groups of 64 candidate entries every wave of k_pair_list asks for before it knows the cell's T (candCap >= 320)
constexpr int kListMinCand = 64 * kListPreload;
constexpr int kListMinIter = 16;           // two chunks of 8 iterations are asked for up front
constexpr int kListMaxSlices = 32;         // slices per atom (a cell of one or two atoms still uses half a wave)

constexpr int kListMaxWaves = 4;
// a cell of n atoms served by W waves: every wave takes ceil(n / W) atoms (the last ones fewer, possibly none), each with NS = 64 / that many slices
__host__ __device__ inline int list_atoms_per_wave(int nAtoms, int W) { return (nAtoms + W - 1) >> (W >> 1); }      // (W is 1, 2 or 4: a shift, not a division)
__host__ __device__ inline int list_slices(int nAtomsOfWave) { const int n = kWave / (nAtomsOfWave > 0 ? nAtomsOfWave : 1); return n > kListMaxSlices ? kListMaxSlices : (n < 1 ? 1 : n); }

// LDS of k_pair_list by kernel mode (LDS per wave bounds its occupancy, so every byte per candidate counts):
//   one species, LJ (MODE 1)            records {x, y, z} of 24 B; a list entry is the record's byte offset
//   table-driven modes (2, 3, 5)        [pair table 1 KiB][species bytes, kListTypeBytes][records of 24 B]; a list entry is the record NUMBER (the species byte
//                                       sits at a fixed offset + number, the record at number x 24 + base: one v_mad more per visit, 8 B less per candidate)
//   radii (MODE 4) / generic (MODE 0)   [records of 24 B][radii, 8 B each][species bytes (MODE 0)]; entry = record number.  (Records of 32 B {x, y, z, radius}
//                                       were tried first: a stride of 8 banks leaves 8 distinct bank groups for 64 lanes - 90 % of the LDS cycles of the
//                                       case-study-2 kernel were bank conflicts; 24 B = 6 banks gives 32)
inline bool pair_list_tab_mode(const StepParams& P) { return P.pad1 == 2 && !P.single_lj; }
inline size_t pair_list_lds_bytes(const StepParams& P, const PairLists& L)
{
    const bool tab = pair_list_tab_mode(P);
    const bool generic = !P.single_lj && P.pad1 != 2 && P.pad1 != 4;
    const bool radii = !P.single_lj && !tab;                       // MODE 4 and MODE 0
    size_t b = (size_t)(L.candLds + 1) * 24;
    if (tab) b += sizeof(double) * kLjSpecMax * kLjSpecMax * kPairTabStride + (((size_t)L.candLds + 1 + 15) & ~(size_t)15);
    if (radii) b += sizeof(double) * (size_t)(L.candLds + 1);
    if (generic) b += (size_t)(L.candLds + 1 + 7) & ~(size_t)7;
    return b;
}
inline int pair_list_rec_bytes(const StepParams&) { return 24; }
inline int pair_list_entry_scale(const StepParams& P) { return P.single_lj ? 24 : 1; }     // what the builder multiplies a record number by: byte offset, or the number itself

// ENG = false: the launch books no energies (steps whose energies nobody can see: all but the last step of an aztot_step call - the statistics are those of
// the last step, finish_steps; the reference prints them every `stat` steps, cuStat.cu:308-330).  Forces are the same instructions either way: the energy
// terms feed nothing else, the compiler drops them and their two wave reductions.
// MULTI = false: one wave per cell, known when the kernel is compiled (the wave number and every multiple of W fold away: 585 -> 515 vector instructions per
// cell on the 1 M-atom liquid); true: W = PairLists::waves waves share the cell's tile.
// TSTAT = true (launches that fuse the next step in a run with the radiative thermostat; never with ENG: the call's last step does not fuse): the epilogue
// applies the thermostat between closing this step and opening the next, operation for operation what k_boundary_radi does in a launch of its own.
template <int MODE, int VDW, bool ENG, bool MULTI, bool TSTAT = false>
__global__ __launch_bounds__(MULTI ? kWave * kListMaxWaves : kWave) void k_pair_list(StepParams P, SpecTable S, const DevPot* __restrict__ pots, AtomArrays A,
                                                     const int32_t* __restrict__ cellStart, int firstCell, int nCellsRun,
                                                     double* __restrict__ partials, int maxBlocks, const Counts* __restrict__ counts, int blockBase, PairLists L,
                                                     NextStep N)
{
    constexpr bool kOneSpecies = (MODE == 1 || MODE == 4);           // no species table, no type ids
    constexpr bool kRadii = (MODE == 0 || MODE == 4);
    constexpr bool kTab = (MODE == 2 || MODE == 3 || MODE == 5);
    constexpr int RECB = 24;                                         // record {x, y, z} RELATIVE to the centre of the cell
    constexpr int ESCALE = (MODE == 1) ? 1 : 24;                     // list entry -> byte offset of the record (only the one-species LJ kernel keeps byte offsets)
    constexpr int kTabDoubles = kTab ? kLjSpecMax * kLjSpecMax * kPairTabStride : 0;
    extern __shared__ double ldsList[];
    double* const pairTab = ldsList;
    uint8_t* const ttypT = (uint8_t*)(ldsList + kTabDoubles);        // table modes: species ids by record number, at a compile-time offset
    char* const tb = (char*)(ldsList + kTabDoubles) + (kTab ? ((L.candLds + 1 + 15) & ~15) : 0);
    double* const trad = (double*)(tb + (size_t)(L.candLds + 1) * RECB);         // MODE 0 / 4: radii (by record number)
    uint8_t* const ttyp0 = (uint8_t*)(trad + (L.candLds + 1));                   // MODE 0 only: species ids (by record number)

    const int lane = threadIdx.x & (kWave - 1), wave = MULTI ? (int)(threadIdx.x >> 6) : 0;
    const int W = MULTI ? L.waves : 1;                               // waves of this workgroup = waves per cell (they share the tile, each serves its share of the atoms)
    const int per = (nCellsRun + 7) >> 3;
    const int cr = (blockIdx.x & 7) * per + (blockIdx.x >> 3);       // XCD-aware: each XCD owns a contiguous eighth of the cells
    double eV = 0.0, eC = 0.0, dropped = 0.0, eK = 0.0;
    NextAcc nacc;                                                    // (touched only in launches that fuse the next step: no initialisation on the others)
    if (N.xn) next_acc_clear(nacc);
    // everything that depends on the cell number only is requested at once, before anything is known about the cell (the loads stay inside the
    // arrays whatever they return): list header, five groups of candidate entries, the lane's first two list chunks.  A wave's life is then two
    // memory round trips (these, then the coordinates) and the loop
    const int cell = firstCell + min(cr, nCellsRun - 1);
    // (candidate groups are dealt to the waves round-robin: wave w gathers groups w, w + W, ...; candCap >= 64 * 5 * W keeps the five preloads inside the array)
    const uint32_t* const myList = L.cand + (size_t)cell * L.candCap + lane;
    const uint4* const pl = (const uint4*)(L.pairs + ((size_t)cell * W + wave) * L.iterCap * kWave) + lane;
    uint32_t ent[kListPreload];
#pragma unroll
    for (int u = 0; u < kListPreload; u++) ent[u] = myList[(wave + u * W) * kWave];
    uint4 w = pl[0];
    uint4 w1 = pl[kWave];                                           // (the second chunk too: 16 iterations cover a liquid's cells, and a chunk asked for only
                                                                    //  8 iterations ahead arrives late)
    // {list header, cell coordinates lx | cy << 10 | cz << 20, the cell's first atom, atoms | rcpNS << 12}: one scalar load - no look at cellStart (a second,
    // dependent round trip), no integer divisions (the builder leaves slices per atom and their reciprocal in the header)
    const int4 mx = ((const int4*)L.meta)[cell];
    asm volatile("" ::: "memory");                                  // (what follows is issued behind the loads above, not in front of them)
    // after a slack violation (one GPU) the clean-up launch stages every cell with the wider stencil: stand down
    const bool violated = P.nranks == 1 && !P.optimistic && slack_violated(P, counts);
    int meta = mx.x;
    if (violated || cr >= nCellsRun || !list_usable(L, meta)) meta = -1;         // (a cell that does not walk its list is served by the clean-up launch)
    if (meta > 0)
    {
        const int T = meta & 0xFFF, nIter = (meta >> 12) & 0xFF;
        const int ncy = P.nc[1], ncz = P.nc[2];
        const int lx = mx.y & 1023, cy = (mx.y >> 10) & 1023, cz = (mx.y >> 20) & 1023;
        const int ib = mx.z;
        const double cc0 = (lx + P.cx0) * P.csz[0] + 0.5 * P.csz[0], cc1 = cy * P.csz[1] + 0.5 * P.csz[1], cc2 = cz * P.csz[2] + 0.5 * P.csz[2];
        const DevPot lj = pots[0];
        if (kTab)
        {
            const int np = P.nSpec * P.nSpec;
            if (threadIdx.x < np)
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
        // ---- the cell's own atoms: this wave serves atoms [wave * aw, wave * aw + aw) of the cell; lane = slot * NS + slice
        const int ncell = mx.w & 0xFFF;                                // 1 .. 64 W (cells with more keep no list)
        const int aw = list_atoms_per_wave(ncell, W);
        const int nthis = max(0, min(aw, ncell - wave * aw));           // atoms of this wave (the last waves of a small cell may have none)
        const int NS = (meta >> 20) & 63;                               // slices per atom, list_slices(aw), and 65536 / NS + 1: left in the header by the builder
        const int rcpNS = mx.w >> 12;
        const int slot = (lane * rcpNS) >> 16, slice = lane - slot * NS;
        const bool validI = slot < nthis;
        const int myi = ib + wave * aw + slot;
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
        // ---- gather the candidates (groups of 64; the builder padded the last group with a valid atom).  Five groups are gathered whatever T is (stale
        // entries are atom indices too: the array starts out zeroed and only indices are ever written); all their loads travel together
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
            auto put = [&](auto IMG, int u, uint32_t e, const Cand& c) {
                // (the tile holds candLds records, not whole groups of 64: LDS per wave bounds the occupancy; the first four groups always fit)
                if ((MULTI || u >= 4) && u * kWave + lane >= L.candLds) return;      // (one wave: its first four groups always fit, candLds >= 256)
                double xj = c.x, yj = c.y, zj = c.z;
                if (decltype(IMG)::value)
                {   // image code per axis: 0 -> -L, 1 -> 0, 2 -> +L (exact: the product is +-L or 0)
                    xj += (double)((int)((e >> 26) & 3u) - 1) * P.L[0];
                    yj += (double)((int)((e >> 28) & 3u) - 1) * P.L[1];
                    zj += (double)((int)((e >> 30) & 3u) - 1) * P.L[2];
                }
                char* const r = tb + (size_t)(u * kWave + lane + 1) * RECB;       // candidate k -> record k + 1 (record 0 is the dummy)
                *(double*)r = xj - cc0; *(double*)(r + 8) = yj - cc1; *(double*)(r + 16) = zj - cc2;
                if (kTab) ttypT[u * kWave + lane + 1] = (uint8_t)c.typ;
                if (kRadii) trad[u * kWave + lane + 1] = c.rad;
                if (MODE == 0) ttyp0[u * kWave + lane + 1] = (uint8_t)c.typ;
            };
            // this wave's groups are wave + q W, q = 0, 1, ...  Five of them are gathered whatever T is - straight-line code, fifteen loads in flight together
            // (the builder fills all five groups of the array: behind the last candidate with the cell's first atom, one cache line for the whole wave)
            const Cand c0 = fetch(ent[0]), c1 = fetch(ent[1]), c2 = fetch(ent[2]), c3 = fetch(ent[3]), c4 = fetch(ent[4]);
            using ImgYes = std::integral_constant<bool, true>;
            using ImgNo = std::integral_constant<bool, false>;
            // (two copies of the straight-line code rather than a wave-uniform branch inside every put: interior cells - nearly all - shift nothing)
            if (images) { put(ImgYes(), wave, ent[0], c0); put(ImgYes(), wave + W, ent[1], c1); put(ImgYes(), wave + 2 * W, ent[2], c2); put(ImgYes(), wave + 3 * W, ent[3], c3); put(ImgYes(), wave + 4 * W, ent[4], c4); }
            else { put(ImgNo(), wave, ent[0], c0); put(ImgNo(), wave + W, ent[1], c1); put(ImgNo(), wave + 2 * W, ent[2], c2); put(ImgNo(), wave + 3 * W, ent[3], c3); put(ImgNo(), wave + 4 * W, ent[4], c4); }
            if (T > kListPreload * W * kWave)
            {   // dense systems: the rest of the tile, four groups per round trip
                for (int q0 = kListPreload; (wave + q0 * W) * kWave < T; q0 += 4)
                {
                    uint32_t en[4];
#pragma unroll
                    for (int q = 0; q < 4; q++) en[q] = ((wave + (q0 + q) * W) * kWave < T) ? myList[(wave + (q0 + q) * W) * kWave] : ent[0];
                    const Cand d0 = fetch(en[0]), d1 = fetch(en[1]), d2 = fetch(en[2]), d3 = fetch(en[3]);
                    put(ImgYes(), wave + q0 * W, en[0], d0);           // (image code 1 on every axis shifts by exactly 0)
                    if ((wave + (q0 + 1) * W) * kWave < T) put(ImgYes(), wave + (q0 + 1) * W, en[1], d1);
                    if ((wave + (q0 + 2) * W) * kWave < T) put(ImgYes(), wave + (q0 + 2) * W, en[2], d2);
                    if ((wave + (q0 + 3) * W) * kWave < T) put(ImgYes(), wave + (q0 + 3) * W, en[3], d3);
                }
            }
        }
        // idle atom slots sit far away on the other side of the dummy candidate, so that nothing they meet is inside a cut-off
        const double xi = validI ? xr - cc0 : 1e30, yi = validI ? yr - cc1 : 1e30, zi = validI ? zr - cc2 : 1e30;
        // the dummy candidate the unused list entries (0) point at: far outside any cut-off, with a species and a radius the potentials can digest
        constexpr uint32_t dummyOff = 0u;
        if (threadIdx.x == 0)
        {
            char* const r = tb;
            *(double*)r = -1e30; *(double*)(r + 8) = -1e30; *(double*)(r + 16) = -1e30;
            if (kTab) ttypT[0] = 0;
            if (kRadii) trad[0] = 1.0;
            if (MODE == 0) ttyp0[0] = 0;
        }
        __syncthreads();                                                // (one wave: no more than the wave barrier it replaces; the branch is uniform over the workgroup)

        // ---- every lane walks its list
        PairAcc acc = {0, 0, 0, 0, 0, 0};
        int nDropHalf = 0;
        double ljDropR2 = P.ljDropR2;
        if (kTab) asm volatile("" : "+v"(ljDropR2));                   // (table-driven modes: a vector register, like the uniforms of PairHot - see there)
        const PairHot hot = kTab ? pair_hot_in_vgprs(P, lj) : (MODE == 1 ? pair_hot_lj_in_vgprs(P, lj) : pair_hot(P, lj));
        const int nChunks = (nIter + 7) >> 3;
        // software-pipelined: the candidate of iteration t + 1 is read from LDS before the potential of iteration t is evaluated (entries behind a lane's
        // last one point at the dummy - the builder fills the list buffer with it - so the read ahead is always a valid one)
        double xj, yj, zj, radj = 0.0;
        int tj = 0;
        auto fetch = [&](uint32_t en, double& x, double& y, double& z, int& ty, double& rd) {
            const uint32_t ko = en * ESCALE;                           // (table modes: one v_mad_u32_u24 with the records' base)
            x = *(const double*)(tb + ko);
            y = *(const double*)(tb + ko + 8);
            z = *(const double*)(tb + ko + 16);
            if (kTab) ty = ttypT[en];
            if (kRadii) rd = trad[en];
            if (MODE == 0) ty = ttyp0[en];
        };
        fetch(nIter > 0 ? (w.x & 0xFFFFu) : dummyOff, xj, yj, zj, tj, radj);
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
                uint32_t kn = ((u + 1) & 1) ? (ww[(u + 1) >> 1] >> 16) : (ww[(u + 1) >> 1] & 0xFFFFu);      // byte offset of the next candidate's record
                if (u == 7 && c + 1 >= nChunks) kn = dummyOff;           // (behind the list's last chunk there is nothing the builder wrote: stale entries of an older, larger tile)
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

        // fold the slices of every atom (neighbouring lanes) in a fixed tree order and write the force: clear_force + pair sums
        if (NS == 4)
        {   // (the usual case: two lane exchanges inside a quad)
            acc.fx += __shfl_xor(acc.fx, 1, kWave); acc.fy += __shfl_xor(acc.fy, 1, kWave); acc.fz += __shfl_xor(acc.fz, 1, kWave);
            acc.fx += __shfl_xor(acc.fx, 2, kWave); acc.fy += __shfl_xor(acc.fy, 2, kWave); acc.fz += __shfl_xor(acc.fz, 2, kWave);
        }
        else
        {   // n partial sums left in slices 0 .. n - 1: slice i < s takes slice i + s (if there is one), then n = s; at most five steps for 32 slices
            int n = NS;
            for (int s = kListMaxSlices / 2; s >= 1; s >>= 1)
            {
                if (s >= n) continue;                                   // wave-uniform
                const double ux = __shfl_down(acc.fx, s, kWave), uy = __shfl_down(acc.fy, s, kWave), uz = __shfl_down(acc.fz, s, kWave);
                if (slice < s && slice + s < n) { acc.fx += ux; acc.fy += uy; acc.fz += uz; }
                n = s;
            }
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
                if (TSTAT && N.xn)      // close the step as k_integrate2_post does: the thermostat acts on the fully kicked velocity; its draws are keyed by this step's number
                    (void)post_tstat_atom(P, S, A, N.st, N.photons, N.uvx, N.uvy, N.uvz, myi, vx, vy, vz, N.st->stepAtSort + (long long)P.cycleStep);
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
    // every wave books into its own partial-sum slot (fixed order of the final sums whatever W is)
    const size_t pb = (size_t)blockBase + (size_t)blockIdx.x * W + wave;
    if (lane == 0)
    {
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
        if (lane == 0) partials[(size_t)PS_EKIN * maxBlocks + pb] = 0.5 * eK;
    }
    if (N.xn)
    {
        next_step_finish(P, N, nacc, partials, maxBlocks, pb);
        // the step counter of the step being opened (main.cpp:92), and: its second half-kick is NOT owed to the next k_integrate1_bin
        if (blockIdx.x == 0 && threadIdx.x == 0) { N.st->step += 1; N.st->pendingKick = 0; }
    }
    else if (N.pendingAfter >= 0 && blockIdx.x == 0 && threadIdx.x == 0) N.st->pendingKick = N.pendingAfter;
}

// ---------------------------------------------------------------------------------------------------------------------------------------------
// k_build_lists: the step that rebuilds the cells makes the lists (geometry only - independent of the potential set; no forces: k_pair_list
// computes that step's forces like any other's, so a run has ONE force kernel and one summation order).  Everything here is f32: the lists only
// have to be a SUPERSET of the pairs within rc + skin (k_pair_list applies the exact fp64 cut-off test every step), and the sort leaves every atom's
// position relative to the centre of its own cell as one float4 (PairLists::rel) - so a neighbour's position relative to THIS cell's centre is that
// plus a whole number of cell edges, whatever the periodic wrap: no image shifts, no fp64, one 16-byte load per candidate.  Per cell, one wave:
//   1. the z-runs of the stencil are looked up by the lanes in parallel; the loads of up to eight runs travel together (a wave's life is two memory
//      round trips and the arithmetic);
//   2. candidates are pruned against the cell's box and packed into LDS with wave ballot + popcount prefix, together with their list entries
//      (atom index + image code, which k_pair_list needs for its fp64 gather);
//   3. per group of 16 atoms of the cell ONE v_mfma_f32_16x16x4_f32 per 16 candidates tests 256 pairs against the list radius
//      (D = thr - |ri - rj|^2 with a threshold widened by the f32 error bound: conservative);
//   4. every lane pops its hits into its atom's stretch of a compact per-cell array; then every lane of k_pair_list's layout
//      (lane = atom * NS + slice) reads ITS entries out of that array (entry slice + t NS of its atom) and the list leaves as whole 1 KiB chunks.
// A cell whose candidates exceed the tile, whose lists exceed iterCap or that holds more than 64 atoms keeps no list (header -1): the clean-up
// launch of k_pair_tile stages it on every step, and the engine grows the capacities when that happens.
// ---------------------------------------------------------------------------------------------------------------------------------------------
struct BuildLds
{
    int capS;            // floats per coordinate array: candLds rounded up to whole mask words (the matrix filter reads 8 blocks of 16 candidates at a time)
    int nWords;          // 32-bit hit-mask words per lane and atom group: capS / 128
    int hitCap;          // entries of the compact hit array: iterLds x 64 x waves per cell
    int nTab;            // entries of each of the three small tables (staging table: <= 64 runs ; per atom of the cell: <= 64 x waves)
    __host__ __device__ BuildLds(int candLds, int iterLds, int waves)
        : capS((candLds + 127) & ~127), nWords((candLds + 127) / 128), hitCap(iterLds * kWave * waves), nTab(kWave * waves) {}
    __host__ __device__ size_t union_bytes() const { const size_t a = sizeof(uint32_t) * (size_t)capS, b = sizeof(uint16_t) * (size_t)hitCap; return ((a > b ? a : b) + 15) & ~(size_t)15; }
    __host__ __device__ size_t bytes() const { return sizeof(float) * 4 * (size_t)capS + union_bytes() + sizeof(uint32_t) * (size_t)nWords * kWave + sizeof(int32_t) * 3 * (size_t)nTab; }
};

__global__ __launch_bounds__(kWave) void k_build_lists(StepParams P, const int32_t* __restrict__ cellStart, int firstCell, int nCellsRun, PairLists L)
{
    extern __shared__ float ldsBuild[];
    const int W = L.waves;
    const BuildLds G(L.candLds, L.iterLds, W);
    float4* const tile = (float4*)ldsBuild;                         // {x, y, z, -(x^2 + y^2 + z^2)} of every candidate, relative to the cell centre: one 16-byte store each;
                                                                    // lane (row c, component k) of the matrix operand reads float 4 (16 b + perm(c)) + k: 64 different banks
    uint32_t* const tent = (uint32_t*)(tile + G.capS);              // list entries of the candidates (atom index | image code << 26), tile order ...
    uint16_t* const hits = (uint16_t*)tent;                         // ... and, once those have left, the compact hit array (entries of k_pair_list's lists, atom by atom)
    uint32_t* const maskBuf = (uint32_t*)((char*)tent + G.union_bytes());      // [nWords][64] hit masks of the atom group in flight
    int32_t* const entJ = (int32_t*)(maskBuf + (size_t)G.nWords * kWave);      // staging table: first atom, count (<= 64), codes ; later, per atom of the cell:
    int32_t* const entN = entJ + G.nTab;                            //   tile slot / offset / hit count
    int32_t* const entC = entN + G.nTab;

    const int lane = threadIdx.x;
    const int per = (nCellsRun + 7) >> 3;
    const int cr = (blockIdx.x & 7) * per + (blockIdx.x >> 3);       // the mapping of k_pair_list: a cell is built and walked on the same XCD
    if (blockIdx.x == 0 && lane == 0) atomicAdd(&L.noList[1], nCellsRun);      // cells recorded (one atomic per launch)
    if (cr >= nCellsRun) return;
    const int cell = firstCell + cr;
    const int ncy = P.nc[1], ncz = P.nc[2];
    const int cxyz = L.meta[4 * cell + 1];
    const int lx = cxyz & 1023, cy = (cxyz >> 10) & 1023, cz = (cxyz >> 20) & 1023;
    const int ib = cellStart[cell], ie = cellStart[cell + 1];
    const int nthis = ie - ib;
    if (nthis == 0) { if (lane == 0) L.meta[4 * cell] = 0; return; }            // an empty cell: a list with nothing in it
    auto no_list = [&](int why) { if (lane == 0) { L.meta[4 * cell] = -1; atomicAdd(&L.noList[0], 1); atomicAdd(&L.noList[2], 1); if (why) atomicAdd(&L.noList[why], 1); } };
    if (nthis > kWave * W) { no_list(0); return; }                 // (W waves of k_pair_list share the cell: up to 64 atoms each)
    const int RECB = L.entryScale;                                  // record number -> list entry (k_pair_list's mode decides: byte offset or number)
    const int candLds = L.candLds;
    const float cs0 = (float)P.csz[0], cs1 = (float)P.csz[1], cs2 = (float)P.csz[2];
    // f32 pruning radius: the list radius + what rounding can do to a coordinate (an atom's own-cell offset is rounded to f32, |x| < a few cell edges:
    // 2^-22 relative on the square is far more than that)
    const float pruneF = (float)(P.pruneR2 * (1.0 + 1e-5));
    // The candidates are pruned against the bounding box of the cell's ATOMS, not against the cell: a dozen atoms leave on average an eighth of the cell's
    // edge empty on either side, and the box dilated by the list radius holds 15 % fewer candidates (C4: 299 -> 255) - less to filter here, and in
    // k_pair_list a smaller LDS tile (its occupancy), a fifth group of candidates that is mostly padding, shorter gathers.  Conservative: an atom farther than
    // the list radius from the box is farther than that from every atom inside it.
    float bc0, bc1, bc2, bh0, bh1, bh2;
    {
        float lo0 = 3e38f, lo1 = 3e38f, lo2 = 3e38f, hi0 = -3e38f, hi1 = -3e38f, hi2 = -3e38f;
        for (int a = lane; a < nthis; a += kWave)
        {
            const float4 m = L.rel[ib + a];
            lo0 = fminf(lo0, m.x); hi0 = fmaxf(hi0, m.x); lo1 = fminf(lo1, m.y); hi1 = fmaxf(hi1, m.y); lo2 = fminf(lo2, m.z); hi2 = fmaxf(hi2, m.z);
        }
#pragma unroll
        for (int o = kWave >> 1; o >= 1; o >>= 1)
        {
            lo0 = fminf(lo0, __shfl_xor(lo0, o, kWave)); hi0 = fmaxf(hi0, __shfl_xor(hi0, o, kWave));
            lo1 = fminf(lo1, __shfl_xor(lo1, o, kWave)); hi1 = fmaxf(hi1, __shfl_xor(hi1, o, kWave));
            lo2 = fminf(lo2, __shfl_xor(lo2, o, kWave)); hi2 = fmaxf(hi2, __shfl_xor(hi2, o, kWave));
        }
        // (half-widths rounded up by more than the f32 error of centre and difference)
        bc0 = 0.5f * (lo0 + hi0); bh0 = 0.5f * (hi0 - lo0) * 1.000001f + 1e-5f;
        bc1 = 0.5f * (lo1 + hi1); bh1 = 0.5f * (hi1 - lo1) * 1.000001f + 1e-5f;
        bc2 = 0.5f * (lo2 + hi2); bh2 = 0.5f * (hi2 - lo2) * 1.000001f + 1e-5f;
    }

    // ---- staging: all z-runs are looked up by the lanes in parallel, packed into a table, loaded eight at a time
    int T = 0;
    bool overflow = false;
    const int nSegTot = P.nOff[0] * P.nOff[1] * 3;
    const int rcpOff1 = 65536 / P.nOff[1] + 1;
    for (int sg0 = 0; sg0 < nSegTot && !overflow; sg0 += kWave)
    {
        int rjb = 0, rn = 0, rcode = 0;
        {
            const int sg = sg0 + lane;
            const int col = (sg * 21846) >> 16, seg = sg - 3 * col;
            const int ox = (col * rcpOff1) >> 16, oy = col - ox * P.nOff[1];
            int nx = lx + ox - P.hw[0], cxs = 1;                      // image codes (for k_pair_list's fp64 gather): 0 -> -L, 1 -> 0, 2 -> +L
            if (P.nranks > 1)
            {   // slab window: ghost layers are resident; their coordinates are global, so k_pair_list shifts across the seam
                const int gx = nx + P.cx0;
                if (gx < 0) cxs = 0; else if (gx >= P.nc[0]) cxs = 2;
            }
            else if (nx < 0) { nx += P.nc[0]; cxs = 0; }
            else if (nx >= P.nc[0]) { nx -= P.nc[0]; cxs = 2; }
            int ny = cy + oy - P.hw[1], cys = 1;
            if (ny < 0) { ny += ncy; cys = 0; } else if (ny >= ncy) { ny -= ncy; cys = 2; }
            const int zlo = cz - P.hw[2], zhi = cz + P.hw[2];
            int zs, ze, czs = 1;
            bool have = sg < nSegTot;
            if (seg == 0) { zs = max(zlo, 0); ze = min(zhi, ncz - 1); }
            else if (seg == 1) { have = have && zlo < 0; zs = zlo + ncz; ze = ncz - 1; czs = 0; }
            else { have = have && zhi >= ncz; zs = 0; ze = zhi - ncz; czs = 2; }
            if (have)
            {
                const int colBase = (nx * ncy + ny) * ncz;
                rjb = cellStart[colBase + zs];
                rn = cellStart[colBase + ze + 1] - rjb;
                // image code (6 bits) | x offset of the column in cells + 32 (6 bits) | y offset + 32 (6 bits)
                rcode = cxs | (cys << 2) | (czs << 4) | ((ox - P.hw[0] + 32) << 6) | ((oy - P.hw[1] + 32) << 12);
            }
        }
        while (__any(rn > 0) && !overflow)
        {
            const unsigned long long emask = __ballot(rn > 0);
            const int nEnt = __popcll(emask);
            if (rn > 0)
            {
                const int pe = lanes_below(emask);
                entJ[pe] = rjb; entN[pe] = min(rn, kWave); entC[pe] = rcode;
                rjb += kWave; rn -= kWave;
            }
            __builtin_amdgcn_wave_barrier();
            for (int e = 0; e < nEnt && !overflow; e += 8)
            {
                const int le = min(e + (lane & 7), nEnt - 1);
                const int vj = entJ[le], vn = entN[le], vc = entC[le];
                float4 g[8];
                int gjn[8];
#pragma unroll
                for (int u = 0; u < 8; u++)
                {
                    const int j = __builtin_amdgcn_readlane(vj, u) + lane;
                    gjn[u] = (e + u < nEnt) ? __builtin_amdgcn_readlane(vn, u) : 0;
                    g[u] = L.rel[j];                                   // (unconditional: the array is padded by a wave's width, lanes beyond the run are dropped below)
                }
#pragma unroll
                for (int u = 0; u < 8; u++)
                {
                    if (gjn[u] > 0 && !overflow)
                    {
                        const int code = __builtin_amdgcn_readlane(vc, u);
                        const float offx = (float)(((code >> 6) & 63) - 32) * cs0, offy = (float)(((code >> 12) & 63) - 32) * cs1;
                        const float dzc = (float)((((code >> 4) & 3) - 1) * ncz - cz);          // cells along z between the candidate's cell (its own index rides in .w) and this one
                        const float xf = g[u].x + offx, yf = g[u].y + offy, zf = fmaf(g[u].w + dzc, cs2, g[u].z);
                        // distance from the box of the cell's atoms: atoms farther than the list radius cannot be anybody's partner
                        const float bx = fmaxf(fabsf(xf - bc0) - bh0, 0.0f), by = fmaxf(fabsf(yf - bc1) - bh1, 0.0f), bz = fmaxf(fabsf(zf - bc2) - bh2, 0.0f);
                        const bool keep = (lane < gjn[u]) && (bx * bx + by * by + bz * bz) <= pruneF;
                        const unsigned long long mask = __ballot(keep);
                        const int nk = __popcll(mask);
                        if (T + nk > candLds) { overflow = true; }
                        else
                        {
                            if (keep)
                            {
                                const int pp = T + lanes_below(mask);
                                tile[pp] = make_float4(xf, yf, zf, -(xf * xf + yf * yf + zf * zf));
                                tent[pp] = (uint32_t)(__builtin_amdgcn_readlane(vj, u) + lane) | ((uint32_t)(code & 63) << 26);
                            }
                            T += nk;
                        }
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    if (overflow) { no_list(5); return; }
    if ((P.pad0 & 3) == 1) return;                                 // (phase timing: staging only)

    // ---- candidates: written out in whole groups of 64 (k_pair_list gathers whole groups: the last one is filled with a valid atom, the cell's first)
    {
        uint32_t* const myList = L.cand + (size_t)cell * L.candCap;
        const int Tpad = max((T + kWave - 1) & ~(kWave - 1), kListPreload * W * kWave);      // (k_pair_list gathers five groups per wave whatever T is)
        for (int q = lane; q < Tpad; q += kWave) myList[q] = (q < T) ? tent[q] : ((uint32_t)ib | (0x15u << 26));
    }
    // far-away, finite dummies behind the last candidate up to the end of its mask word: the filter reads whole words of 8 x 16 candidates, without guards
    for (int q = T + lane; q < ((T + 127) & ~127); q += kWave) tile[q] = make_float4(-1e18f, 0.0f, 0.0f, -3e38f);
    // where the cell's own atoms sit in the tile (they are candidates too, unshifted): found by their list entries
    for (int q = lane; q < T; q += kWave)
    {
        const uint32_t en = tent[q];
        const int rel = (int)(en & 0x3FFFFFFu) - ib;
        if ((en >> 26) == 0x15u && rel >= 0 && rel < nthis) entJ[rel] = q;
    }
    __builtin_amdgcn_wave_barrier();                                // (from here on the entries' place holds the hit array)

    // ---- filter + compaction, one group of 16 atoms at a time.  Matrix operand maps (gfx950, v_mfma_f32_16x16x4_f32): A[row l & 15][k l >> 4],
    // B[k l >> 4][col l & 15], C/D[row 4 (l >> 4) + reg][col l & 15].  Rows are fed in the order perm(c) = (c >> 2) + 4 (c & 3), which makes register r
    // of lane (atom l & 15, quarter kq = l >> 4) the candidate 16 b + kq + 4 r of block b; the -|rj|^2 of those four arrive as the C operand.
    const double h0 = 0.5 * P.csz[0], h1 = 0.5 * P.csz[1], h2 = 0.5 * P.csz[2];
    const double rList = sqrt(P.pruneR2);
    const double ext2 = (h0 + rList) * (h0 + rList) + (h1 + rList) * (h1 + rList) + (h2 + rList) * (h2 + rList);
    const float thrList = (float)(P.pruneR2 + 1.9073486328125e-06 * (4.0 * ext2 + P.pruneR2));       // list radius^2 + 2^-19 x a bound of the f32 error
    const int aw = list_atoms_per_wave(nthis, W);                   // k_pair_list: wave w serves atoms [w aw, w aw + aw), lane = (atom - w aw) * NS + slice
    const int NS = list_slices(aw);
    const int rcpNS = 65536 / NS + 1;
    const int nBlk = (T + 15) >> 4;
    const int nW = (nBlk + 7) >> 3;
    const int c16 = lane & 15, kq = lane >> 4;
    const int permc = (c16 >> 2) + ((c16 & 3) << 2);
    // A[cand][k] = (xj, yj, zj, -|rj|^2), B[k][atom] = (2 xi, 2 yi, 2 zi, 1), C[cand][atom] = thr - |ri|^2 (the same in a lane's four registers, for every block):
    // the per-candidate term rides in the A operand, so a block costs ONE LDS read per lane and no operand selects
    const float* const pa = (const float*)tile + 4 * permc + kq;
    int nIter = 0, hitBase = 0, tooLong = 0;
    for (int g0 = 0; g0 < nthis; g0 += 16)
    {
        const int a = g0 + c16;
        const bool validA = a < nthis;
        // idle atom slots sit at 1e18: their column of the filter is hugely negative ("outside") whatever the candidate
        float4 me = make_float4(1e18f, 1e18f, 1e18f, 0.f);
        if (validA) me = L.rel[ib + a];
        const float fB = (kq == 0) ? 2.0f * me.x : (kq == 1) ? 2.0f * me.y : (kq == 2) ? 2.0f * me.z : 1.0f;
        const float cAtom = thrList - (me.x * me.x + me.y * me.y + me.z * me.z);
        const float4_t cw = {cAtom, cAtom, cAtom, cAtom};
        const int kSelf = validA ? entJ[a] : -1;
        int h = 0;
        for (int wd = 0; wd < nW; wd++)
        {
            // all operands of the word's eight blocks first (the tile is padded with dummies to the end of the word), then the eight matrix instructions,
            // then the sign bits: no guards, so the reads of one block do not wait for the arithmetic of the one before
            float av[8];
#pragma unroll
            for (int q = 0; q < 8; q++) av[q] = pa[(8 * wd + q) * 64];
            uint32_t miss = 0u;                                      // one bit per candidate of the lane, 1 = outside; newest candidate in bit 0
#pragma unroll
            for (int q = 0; q < 8; q++)
            {
                const float4_t d = __builtin_amdgcn_mfma_f32_16x16x4f32(av[q], fB, cw, 0, 0, 0);
                miss = __builtin_amdgcn_alignbit(miss, __float_as_uint(d[0]), 31);
                miss = __builtin_amdgcn_alignbit(miss, __float_as_uint(d[1]), 31);
                miss = __builtin_amdgcn_alignbit(miss, __float_as_uint(d[2]), 31);
                miss = __builtin_amdgcn_alignbit(miss, __float_as_uint(d[3]), 31);
            }
            // candidate number p of this word (p = 4 (block & 7) + r) sits at bit 31 - p; the dummies behind the last candidate are misses
            uint32_t m = ~miss;
            // the atom itself is a candidate of its own tile (always a hit): not a partner
            if (kSelf >= 0 && (kSelf & 3) == kq && (kSelf >> 7) == wd) m &= ~(0x80000000u >> ((kSelf & 127) >> 2));
            maskBuf[wd * kWave + lane] = m;
            h += __popc(m);
        }
        // where this lane's hits go: behind those of the lower quarters of its atom, which come behind the atoms before it
        int below = 0, total = 0;
#pragma unroll
        for (int q = 0; q < 4; q++)
        {
            const int hq = __shfl(h, c16 | (q << 4), kWave);
            below += (q < kq) ? hq : 0;
            total += hq;
        }
        if (!validA) total = 0;
        int incl = total;                                            // inclusive scan over the 16 atoms of the group (every quarter computes the same)
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) { const int up = __shfl_up(incl, o, 16); if (c16 >= o) incl += up; }
        const int myOff = hitBase + incl - total;
        hitBase += __shfl(incl, 15, 16);
        if (validA && kq == 0) { entN[a] = myOff; entC[a] = total; }
        nIter = max(nIter, ((total + NS - 1) * rcpNS) >> 16);
        if (hitBase > G.hitCap) { tooLong = 1; break; }              // (wave-uniform)
        if ((P.pad0 & 3) == 2) continue;                           // (phase timing: no compaction)
        // candidate k = 128 wd + kq + 4 p sits in record k + 1 of k_pair_list's tile
        uint16_t* dst = hits + myOff + below;
        for (int wd = 0; wd < nW; wd++)
        {
            uint32_t cur = maskBuf[wd * kWave + lane];
            const int base = (128 * wd + kq + 1) * RECB;
            while (cur != 0u)
            {
                const int p = __clz(cur);
                cur &= ~(0x80000000u >> p);
                *dst++ = (uint16_t)(base + p * 4 * RECB);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    if ((P.pad0 & 3) >= 2) return;                                 // (phase timing: no read-out)
    nIter = wave_max_int(nIter);
    // (debug bit 65536, tests: lists hold 14 iterations only - part of a liquid's cells then keep no list and go through the clean-up launch)
    const bool usable = !tooLong && nIter <= ((P.pad0 & 65536) ? 14 : L.iterCap);
    if (usable)
    {   // every lane of k_pair_list's layout collects ITS entries: lane = slot * NS + slice of wave w walks entries slice, slice + NS, ... of atom w aw + slot;
        // 0 = no candidate
        const int slot = (lane * rcpNS) >> 16, slice = lane - slot * NS;
        for (int w = 0; w < W; w++)
        {
            const int atom = w * aw + slot;
            int off = 0, cnt = 0;
            if (slot < aw && atom < nthis) { off = entN[atom]; cnt = entC[atom]; }
            const uint16_t* const src = hits + off;
            uint4* const out = (uint4*)(L.pairs + ((size_t)cell * W + w) * L.iterCap * kWave) + lane;
            for (int c = 0; c * 8 < nIter; c++)
            {
                uint32_t v[8];
#pragma unroll
                for (int u = 0; u < 8; u++)
                {
                    const int e = (c * 8 + u) * NS + slice;
                    v[u] = (e < cnt) ? (uint32_t)src[e] : 0u;
                }
                out[c * kWave] = make_uint4(v[0] | (v[1] << 16), v[2] | (v[3] << 16), v[4] | (v[5] << 16), v[6] | (v[7] << 16));
            }
        }
    }
    if (lane == 0)
    {
        // header: candidates | iterations << 12 | slices per atom << 20 ; then the cell's first atom and atoms | (65536 / slices + 1) << 12 - all k_pair_list needs to
        // know about the cell arrives in one 16-byte scalar load
        L.meta[4 * cell] = usable ? (T | (nIter << 12) | (NS << 20)) : -1;
        L.meta[4 * cell + 2] = ib;
        L.meta[4 * cell + 3] = nthis | (rcpNS << 12);
        if (T > L.noList[3]) atomicMax(&L.noList[3], T);           // (a read first: after the first few cells nobody has a new record to report)
        if (usable && nIter > L.noList[4]) atomicMax(&L.noList[4], nIter);
        if (P.pad0 & 2097152) { atomicAdd(&L.noList[8], nIter); atomicAdd(&L.noList[9], T); atomicAdd(&L.noList[10], nthis); }      // measurement aid (slow)
        if (!usable) { atomicAdd(&L.noList[0], 1); atomicAdd(&L.noList[2], 1); atomicAdd(&L.noList[6], 1); }
    }
}

// the step that rebuilds the cells: candidates and pair lists of every cell (no forces; k_pair_list follows)
inline void launch_build_lists(const StepParams& P, const int32_t* cellStart, hipStream_t stream, PairRange R, PairLists L)
{
    pair_range_default(P, R);
    if (R.n == 0) return;
    const BuildLds G(L.candLds, L.iterLds, L.waves);
    hipLaunchKernelGGL(k_build_lists, dim3(pair_range_grid(R.n)), dim3(kWave), G.bytes(), stream, P, cellStart, R.first, R.n, L);
}
inline size_t build_lists_lds_bytes(const PairLists& L) { return BuildLds(L.candLds, L.iterLds, L.waves).bytes(); }

template <int MODE, int VDW>
inline void launch_pair_list_as(const StepParams& P, const SpecTable& S, const DevPot* pots, AtomArrays A, const Counts* cnt, const int32_t* cellStart, double* partials,
                                int maxBlocks, hipStream_t stream, PairRange R, PairLists L, NextStep N, bool energies)
{
    const size_t lds = pair_list_lds_bytes(P, L);
    const dim3 grid(pair_range_grid(R.n)), block(kWave * L.waves);
#define AZTOT_LAUNCH_LIST(E, M, T) hipLaunchKernelGGL((k_pair_list<MODE, VDW, E, M, T>), grid, block, lds, stream, P, S, pots, A, cellStart, R.first, R.n, partials, maxBlocks, cnt, R.blockBase, L, N)
    if (N.photons)
    {   // thermostat-fused epilogue: modes that do not read radii (the thermostat rewrites them while other waves would still be gathering), never on a step
        // whose energies are wanted (Engine::launch_step_kernels)
        if (energies || MODE == 0 || MODE == 4) throw std::runtime_error("k_pair_list: the thermostat cannot be fused into this launch");
        if (MODE != 0 && MODE != 4) { if (L.waves == 1) AZTOT_LAUNCH_LIST(false, false, (MODE != 0 && MODE != 4)); else AZTOT_LAUNCH_LIST(false, true, (MODE != 0 && MODE != 4)); }
    }
    else if (L.waves == 1) { if (energies) AZTOT_LAUNCH_LIST(true, false, false); else AZTOT_LAUNCH_LIST(false, false, false); }
    else { if (energies) AZTOT_LAUNCH_LIST(true, true, false); else AZTOT_LAUNCH_LIST(false, true, false); }
#undef AZTOT_LAUNCH_LIST
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
    return pair_range_grid(R.n) * L.waves;
}

inline int launch_pair_cleanup(const StepParams& P, const SpecTable& S, const DevPot* pots, AtomArrays A, const Counts* cnt, const int32_t* cellStart,
                               double* partials, int maxBlocks, hipStream_t stream, PairRange R, PairLists L, NextStep N = NextStep())
{
    pair_range_default(P, R);
    if (R.n == 0) return 0;
    R.blockBase += pair_range_grid(R.n) * L.waves;
    launch_pair_tile(P, S, pots, A, cnt, cellStart, partials, maxBlocks, stream, R, L, 2, N);
    return pair_cleanup_grid(R.n);
}

}  // namespace aztot
