// Pair-force kernel, variant 3: kQuadR waves = kQuadR z-consecutive cells share ONE staged tile made of whole 16-atom bins.
//
// Same job and same arithmetic as k_pair_tile (pair_tile.hip.h; reference: cell_list5a + cell_list4b_noshared + pair_1,
// cuPairs.cu:2266,1474,117), different staging.  rocprofv3 on the 1 M-atom box showed the one-wave kernel spending 795 of its 1 870
// vector instructions per cell on STAGING the 27 neighbour cells (139 us of 286): every wave loads, shifts, prunes and packs its own 3 x 3 x 3
// stencil atom by atom, although z-neighbouring cells share two thirds of it.  Here
//   * the sort (k_rank_gather) also files every atom in its cell's fixed-size BINS of 16 (CellBins, device_md.h): coordinates relative to
//     the centre of the atom's own cell, the tail of the last bin filled with far-away dummies;
//   * a workgroup of kQuadR waves owns kQuadR cells of one z-column and stages the union 3 x 3 x (kQuadR + 2) once, as a list of whole
//     bins in LAYER-major order (a layer = the 9 cells of one z): 64 lanes copy 4 bins per instruction, adding the bin's cell offset (a
//     multiple of the cell edge - the periodic image shift is implicit in cell-relative coordinates).  The position of every bin follows
//     from one wave-wide prefix sum over the 9 (kQuadR + 2) cell counts, which every wave computes for itself: no compaction, no ballots,
//     no LDS atomics, ONE barrier;
//   * the stencil of the cell at z is the three consecutive layers z-1, z, z+1, one contiguous window of the tile that starts on a bin
//     boundary: wave w runs the unchanged two-pass machinery (tile_passes) on layers w .. w+2;
//   * nothing is pruned: the matrix-instruction distance filter tests 256 pairs in ~70 cycles, so dropping the atoms outside rc + cell
//     (25 %) is no longer worth per-atom bookkeeping.
// The order of the candidates inside the tile - and with it the order of every floating-point sum - is fixed by the cell table, so
// results stay bit-reproducible.  A union that does not fit the tile (dense systems) is staged column by column; atoms of a cell beyond
// its bins come from the per-atom arrays (same arithmetic, slow path); cells with more than 64 atoms take the batch loop of k_pair_tile.
// Stencil half-width 1 only (cell edge >= cut-off) and no generic potential mix; everything else stays with k_pair_tile.
#pragma once
#include "pair_tile.hip.h"

namespace aztot {

#ifndef AZTOT_QUAD_R
#define AZTOT_QUAD_R 4
#endif
constexpr int kQuadR = AZTOT_QUAD_R;             // cells = waves per workgroup (2, 3 or 4)
constexpr int kQuadLayers = kQuadR + 2;          // z-layers of the union stencil
constexpr int kQuadCells = 9 * kQuadLayers;      // cells of the union = lanes of the cell table
constexpr int kQuadBins = kQuadR == 4 ? 64 : (kQuadR == 3 ? 52 : 40);   // tile capacity in bins: ~1.15 bins per cell of the union + slack
constexpr int kQuadTile = kQuadBins * 16 + 16;   // entries: + one bin of dummies behind the last one
static_assert(kQuadTile % 32 == 16, "coordinate arrays must be staggered by half the LDS banks");
static_assert(kQuadCells < kWave, "the cell table lives in one wave");

inline bool pair_quad_supported(const StepParams& P)
{
    if (!pair_tile_supported(P)) return false;
    for (int k = 0; k < 3; k++)
        if (P.hw[k] != 1 || P.nOff[k] != 3) return false;
    return true;
}
// bins to reserve per cell for an average of `avg` atoms per cell: twice the average + 1 bin.  One z-column of the union must fit the
// tile even if every cell used all its bins, which caps the density this kernel takes (~70 atoms per cell)
inline int pair_quad_bins_per_cell(double avg)
{
    int b = (int)(2.0 * avg / 16.0) + 2;
    return b < 2 ? 2 : b;
}
inline bool pair_quad_density_ok(int binsPerCell) { return binsPerCell * kQuadLayers <= kQuadBins; }
inline int pair_quad_groups(const StepParams& P) { return (P.nc[2] + kQuadR - 1) / kQuadR; }
inline int pair_quad_workgroups(const StepParams& P)
{
    const int ncx = (P.nranks > 1) ? (P.ncxLocal - 2 * P.hw[0]) : P.ncxLocal;
    return ncx * P.nc[1] * pair_quad_groups(P);
}
inline int pair_quad_grid(const StepParams& P) { return 8 * ((pair_quad_workgroups(P) + 7) / 8); }

// one entry of a bin: from the bins, or (slow path) from the per-atom arrays for the part of a cell beyond its bins.  Same arithmetic as
// k_rank_gather, so an atom has the same cell-relative coordinates whichever way it arrives
__device__ __forceinline__ void quad_fetch(const StepParams& P, const AtomArrays& A, const CellBins& B, int cell, int cellStartC, int cntC, int binInCell,
                                           int slot, double& x, double& y, double& z, int& t)
{
    if (binInCell < B.perCell)
    {
        const size_t g = ((size_t)cell * B.perCell + binInCell) * 16 + slot;
        x = B.x[g]; y = B.y[g]; z = B.z[g]; t = B.type[g];
        return;
    }
    const int r = binInCell * 16 + slot;
    if (r < cntC)
    {
        const int j = cellStartC + r;
        x = A.x[j]; y = A.y[j]; z = A.z[j]; t = A.type[j];
        x -= cell_coord(x, P.icsz[0], P.nc[0]) * P.csz[0] + 0.5 * P.csz[0];
        y -= cell_coord(y, P.icsz[1], P.nc[1]) * P.csz[1] + 0.5 * P.csz[1];
        z -= cell_coord(z, P.icsz[2], P.nc[2]) * P.csz[2] + 0.5 * P.csz[2];
    }
    else { x = -1e30; y = 0.0; z = 0.0; t = 0; }
}

template <int MODE, int VDW>
__global__ __launch_bounds__(kWave* kQuadR, 4) void k_pair_quad(StepParams P, SpecTable S, const DevPot* __restrict__ pots, AtomArrays A, CellBins B,
                                                                const int32_t* __restrict__ cellStart, int firstLayerX, int nWorkgroups,
                                                                double* __restrict__ partials, int maxBlocks, Counts* __restrict__ counts)
{
    static_assert(MODE >= 1, "the generic potential mix stays with k_pair_tile");
    __shared__ double txyz[3 * kQuadTile];                           // candidate coordinates relative to the centre of the group of cells
    __shared__ float tw[kQuadTile];                                  // -(x^2 + y^2 + z^2): 4th operand row of the distance filter
    __shared__ uint8_t ttyp[MODE != 1 ? kQuadTile : 1];
    __shared__ double pairTab[MODE >= 2 ? kLjSpecMax * kLjSpecMax * kPairTabStride : 1];
    __shared__ int32_t binCell[kQuadBins];                           // per tile bin: cell-table lane it belongs to | bin number inside the cell << 8
    __shared__ double red[kQuadR][4];
    double* const tx = txyz;
    double* const ty = txyz + kQuadTile;
    double* const tz = txyz + 2 * kQuadTile;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // XCD-aware mapping: workgroups b and b + 8 share an XCD (and its L2); give each XCD a contiguous run of z-columns
    const int per = (nWorkgroups + 7) >> 3;
    const int cr = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    const bool wgValid = cr < nWorkgroups;
    const int ncy = P.nc[1], ncz = P.nc[2];
    const int nG = (ncz + kQuadR - 1) / kQuadR;
    const int g = wgValid ? cr % nG : 0, colIdx = wgValid ? cr / nG : 0;
    const int cy = colIdx % ncy, lx = firstLayerX + colIdx / ncy;
    const int z0 = g * kQuadR;
    const int nActive = wgValid ? min(kQuadR, ncz - z0) : 0;           // waves that own a cell
    const bool activeWave = wave < nActive;
    double eV = 0.0, eC = 0.0, dropped = 0.0, eK = 0.0;

    // f32 filter threshold (see k_pair_tile): tile coordinates are relative to the centre of the group's box
    const double h0 = 0.5 * P.csz[0], h1 = 0.5 * P.csz[1], h2 = 0.5 * P.csz[2];
    const double rcut = sqrt(P.r2Max);
    const double e0 = h0 + rcut, e1 = h1 + rcut, e2 = kQuadR * h2 + rcut;
    const double filtThr = P.r2Max + 1.9073486328125e-06 * (4.0 * (e0 * e0 + e1 * e1 + e2 * e2) + P.r2Max);      // 2^-19: f32 error bound
    const DevPot lj = pots[0];
    if (MODE >= 2 && wave == 0)
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

    // ---- the cell table of the union: lane = layer * 9 + column, i.e. in TILE ORDER; every wave computes the same table
    int tCell = 0, tStart = 0, tCnt = 0;
    double tox = 0.0, toy = 0.0, toz = 0.0;                               // offset of the cell's centre from the group's centre
    {
        const int k = lane / 9, col = lane - 9 * k;
        const int ox = col / 3, oy = col - 3 * ox;
        int nx = lx + ox - 1;
        if (P.nranks == 1) { if (nx < 0) nx += P.nc[0]; else if (nx >= P.nc[0]) nx -= P.nc[0]; }   // slab ranks hold their ghost layers
        int ny = cy + oy - 1;
        if (ny < 0) ny += ncy; else if (ny >= ncy) ny -= ncy;
        int zz = z0 - 1 + k;
        if (zz < 0) zz += ncz; else if (zz >= ncz) zz -= ncz;
        tox = (ox - 1) * P.csz[0]; toy = (oy - 1) * P.csz[1]; toz = (k - 0.5 - 0.5 * kQuadR) * P.csz[2];
        // layer k is the stencil of the waves k-2 .. k: beyond the last active wave nobody reads it
        if (wgValid && lane < kQuadCells && k <= nActive + 1 && zz >= 0 && zz < ncz)
        {
            tCell = (nx * ncy + ny) * ncz + zz;
            tStart = cellStart[tCell];
            tCnt = cellStart[tCell + 1] - tStart;
        }
    }
    const int tBins = (tCnt + 15) >> 4;
    // own cell: column 4 (ox = oy = 1) of layer wave + 1
    const int myLane = activeWave ? 9 * (wave + 1) + 4 : 0;
    const int myCell = __builtin_amdgcn_readlane(tCell, myLane), ib = __builtin_amdgcn_readlane(tStart, myLane);
    const int nMine = activeWave ? __builtin_amdgcn_readlane(tCnt, myLane) : 0;
    const double myOz = (wave + 0.5 - 0.5 * kQuadR) * P.csz[2];
    int nMax = 0;
#pragma unroll
    for (int w = 0; w < kQuadR; w++) nMax = max(nMax, __builtin_amdgcn_readlane(tCnt, 9 * (w + 1) + 4));
    // all nine columns at once if the union fits the tile, else column by column (one column always fits: pair_quad_density_ok, and
    // the part of a cell beyond its bins is bounded by the same count)
    int total = tBins;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) total += __shfl_xor(total, o, kWave);
    const int cpb = (total <= kQuadBins) ? 9 : 1;
    bool fits = true;
    if (cpb == 1)
    {   // bins per column, maximum over the columns
        int colBins = tBins;
        const int col = lane % 9;
#pragma unroll
        for (int k = 1; k < kQuadLayers; k++) colBins += __shfl(tBins, col + 9 * k, kWave);
        fits = !__any(lane < 9 && colBins > kQuadBins);
        if (!fits && threadIdx.x == 0) counts->overflow = 1;               // > 170 atoms per cell in one column: this box belongs to k_pair_tile
    }

    for (int i0 = 0; fits && i0 < max(nMax, 1); i0 += kWave)
    {
        const int nthis = max(0, min(kWave, nMine - i0));
        const int lg = nthis <= 16 ? 4 : (nthis <= 32 ? 5 : 6);     // log2(i-slots)
        const int islots = 1 << lg;
        const int il = lane & (islots - 1), slice = lane >> lg;
        const bool validI = il < nthis;
        const int myi = ib + i0 + il;
        double xi = 1e30, yi = 1e30, zi = 1e30;
        int ti = 0;
        if (validI)
        {   // from the bins, exactly as the tile holds it: the atom's own copy among the candidates is then at distance 0.0 exactly
            quad_fetch(P, A, B, myCell, ib, nMine, (i0 + il) >> 4, (i0 + il) & 15, xi, yi, zi, ti);
            zi += myOz;
        }
        const float filtB = (slice == 0) ? (float)(2.0 * xi) : (slice == 1) ? (float)(2.0 * yi) : (slice == 2) ? (float)(2.0 * zi)
                                                                                                : (float)(filtThr - (xi * xi + yi * yi + zi * zi));
        PairAcc acc = {0, 0, 0, 0, 0, 0};

        for (int c0 = 0; c0 < 9; c0 += cpb)
        {
            // position of every cell's first bin in the tile: exclusive prefix sum of the bin counts over the table lanes (= tile order)
            const int col = lane % 9;
            const int nb = (col >= c0 && col < c0 + cpb) ? tBins : 0;
            int incl = nb;
#pragma unroll
            for (int d = 1; d < kWave; d <<= 1)
            {
                const int t = __shfl_up(incl, d, kWave);
                if (lane >= d) incl += t;
            }
            const int pos = incl - nb;
            const int nBinsTot = __builtin_amdgcn_readlane(incl, kWave - 1);
            if (i0 > 0 || c0 > 0) __syncthreads();                        // the previous window passes have finished with the tile
            if (wave == 0)
            {
                for (int b = 0; b < nb; b++) binCell[pos + b] = lane | (b << 8);
                if (lane < 16)
                {   // one bin of far-away, finite dummies behind the last bin: what dead lanes of pass 2 chew on
                    const int idx = nBinsTot * 16 + lane;
                    tx[idx] = -1e30; ty[idx] = 0.0; tz[idx] = 0.0; tw[idx] = -3e38f;
                    if (MODE != 1) ttyp[idx] = 0;
                }
            }
            __syncthreads();

            // ---- staging: 64 lanes copy 4 bins per round; rounds are dealt to the waves in turn
            const int slot = lane & 15;
            for (int q0 = wave * 4; q0 < nBinsTot; q0 += 4 * kQuadR)
            {   // (wave-uniform loop: the cross-lane reads below need every lane of the cell table switched on)
                const int q = q0 + (lane >> 4);
                const bool have = q < nBinsTot;
                const int bc = binCell[have ? q : 0];
                const int tl = bc & 255, bIn = bc >> 8;
                const int cell = __shfl(tCell, tl, kWave), cs = __shfl(tStart, tl, kWave), cn = __shfl(tCnt, tl, kWave);
                const double ox = __shfl(tox, tl, kWave), oy = __shfl(toy, tl, kWave), oz = __shfl(toz, tl, kWave);
                if (have)
                {
                    double xj, yj, zj;
                    int tj;
                    quad_fetch(P, A, B, cell, cs, cn, bIn, slot, xj, yj, zj, tj);
                    xj += ox; yj += oy; zj += oz;
                    const int pp = q * 16 + slot;
                    tx[pp] = xj; ty[pp] = yj; tz[pp] = zj;
                    tw[pp] = -(float)(xj * xj + yj * yj + zj * zj);
                    if (MODE != 1) ttyp[pp] = (uint8_t)tj;
                }
            }
            __syncthreads();

            // ---- the two passes over this wave's window: layers wave .. wave + 2
            if (activeWave)
            {
                const int ws = 16 * __builtin_amdgcn_readlane(pos, 9 * wave);
                const int we = 16 * __builtin_amdgcn_readlane(pos, 9 * (wave + 3));   // lane 9 (R + 2) holds the total: beyond the table
                const int T = we - ws;
                if (lg == 4)
                    tile_passes<MODE, VDW, 4, kQuadTile, 4>(P, S, pots, lj, tx + ws, ty + ws, tz + ws, tw + ws, ttyp + (MODE != 1 ? ws : 0), nullptr, pairTab, T, slice, xi, yi, zi, ti, 0.0, filtB, 0.0f, acc);
                else if (lg == 5)
                    tile_passes<MODE, VDW, 5, kQuadTile, 4>(P, S, pots, lj, tx + ws, ty + ws, tz + ws, tw + ws, ttyp + (MODE != 1 ? ws : 0), nullptr, pairTab, T, slice, xi, yi, zi, ti, 0.0, filtB, 0.0f, acc);
                else
                    tile_passes<MODE, VDW, 6, kQuadTile, 4>(P, S, pots, lj, tx + ws, ty + ws, tz + ws, tw + ws, ttyp + (MODE != 1 ? ws : 0), nullptr, pairTab, T, slice, xi, yi, zi, ti, 0.0, filtB, 0.0f, acc);
            }
        }

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
            if (MODE != 1) q = S.charge[ti];
            const double fxi = -q * P.E[0] + acc.fx;   // clear_force integrators.cpp:17-39
            const double fyi = -q * P.E[1] + acc.fy;
            const double fzi = -q * P.E[2] + acc.fz;
            A.fx[myi] = fxi; A.fy[myi] = fyi; A.fz[myi] = fzi;
            if (P.fuseKick)
            {   // second half-kick + kinetic energy of integrate2 (integrators.cpp:486-531 ; verlet_2stage cuMDfunc.cu:521-600),
                // fused here on plain NVE steps: the force is still in registers
                const double rM = S.rMhdt[ti], m = S.mass[ti];
                const double vx = A.vx[myi] + rM * fxi, vy = A.vy[myi] + rM * fyi, vz = A.vz[myi] + rM * fzi;
                A.vx[myi] = vx; A.vy[myi] = vy; A.vz[myi] = vz;
                eK += (vx * vx + vy * vy + vz * vz) * m;
            }
        }
        eV += acc.eV; eC += acc.eC; dropped += acc.dropped;
    }
    // per-workgroup partial sums in a fixed order: lanes -> wave (shuffles), waves -> thread 0 (LDS)
    eV = wave_sum(eV); eC = wave_sum(eC); dropped = wave_sum(dropped); eK = wave_sum(eK);
    if (lane == 0) { red[wave][0] = eV; red[wave][1] = eC; red[wave][2] = dropped; red[wave][3] = eK; }
    __syncthreads();
    if (threadIdx.x == 0)
    {
        double r[4] = {0, 0, 0, 0};
        for (int w = 0; w < kQuadR; w++)
            for (int k = 0; k < 4; k++) r[k] += red[w][k];
        put_partial(partials, maxBlocks, PS_EVDW, r[0]);
        put_partial(partials, maxBlocks, PS_ECOUL, r[1]);
        if (r[2] != 0.0) add_partial(partials, maxBlocks, PS_DROPPED, r[2]);
        if (P.fuseKick) put_partial(partials, maxBlocks, PS_EKIN, 0.5 * r[3]);
    }
}

template <int MODE, int VDW>
inline void launch_pair_quad_as(const StepParams& P, const SpecTable& S, const DevPot* pots, AtomArrays A, const CellBins& B, Counts* cnt, const int32_t* cellStart,
                                double* partials, int maxBlocks, hipStream_t stream)
{
    const int first = (P.nranks > 1) ? P.hw[0] : 0;
    hipLaunchKernelGGL((k_pair_quad<MODE, VDW>), dim3(pair_quad_grid(P)), dim3(kWave * kQuadR), 0, stream, P, S, pots, A, B, cellStart, first, pair_quad_workgroups(P),
                       partials, maxBlocks, cnt);
}

// same dispatch as launch_pair_tile; the caller has checked pair_quad_supported(P) and that the generic kernel (MODE 0) is not needed
inline bool launch_pair_quad(const StepParams& P, const SpecTable& S, const DevPot* pots, AtomArrays A, const CellBins& B, Counts* cnt, const int32_t* cellStart,
                             double* partials, int maxBlocks, hipStream_t stream)
{
    if (P.single_lj) { launch_pair_quad_as<1, 1>(P, S, pots, A, B, cnt, cellStart, partials, maxBlocks, stream); return true; }
    if (P.pad1 == 2)
    {
        const bool ew = P.elec_type == 2;
        switch (P.vdwFamily)
        {
        case 1: if (ew) launch_pair_quad_as<3, 1>(P, S, pots, A, B, cnt, cellStart, partials, maxBlocks, stream); else launch_pair_quad_as<2, 1>(P, S, pots, A, B, cnt, cellStart, partials, maxBlocks, stream); return true;
        case 2: if (ew) launch_pair_quad_as<3, 2>(P, S, pots, A, B, cnt, cellStart, partials, maxBlocks, stream); else launch_pair_quad_as<2, 2>(P, S, pots, A, B, cnt, cellStart, partials, maxBlocks, stream); return true;
        case 3: if (ew) launch_pair_quad_as<3, 3>(P, S, pots, A, B, cnt, cellStart, partials, maxBlocks, stream); else launch_pair_quad_as<2, 3>(P, S, pots, A, B, cnt, cellStart, partials, maxBlocks, stream); return true;
        case 4: if (ew) launch_pair_quad_as<3, 4>(P, S, pots, A, B, cnt, cellStart, partials, maxBlocks, stream); else launch_pair_quad_as<2, 4>(P, S, pots, A, B, cnt, cellStart, partials, maxBlocks, stream); return true;
        case 5: if (ew) launch_pair_quad_as<3, 5>(P, S, pots, A, B, cnt, cellStart, partials, maxBlocks, stream); else launch_pair_quad_as<2, 5>(P, S, pots, A, B, cnt, cellStart, partials, maxBlocks, stream); return true;
        }
    }
    return false;
}

}  // namespace aztot
