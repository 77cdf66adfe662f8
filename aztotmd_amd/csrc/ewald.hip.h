// Reciprocal-space part of the Ewald sum ('elec pme'; "next" row f4 of SURVEY section 8).
//
// Reference: recip_ewald + ewald_force (cuElec.cu:151-382; one thread per slice of atoms, fp32, per-atom q*exp(ikr) of EVERY
// k-vector kept in global memory: nAt x nKvec float2) and the serial ewald_rec (elec.cpp:167-335, fp64), whose arithmetic
// this follows: exp(i 2pi x/a), exp(i 2pi y/b), exp(i 2pi z/c) once per atom, higher harmonics by complex multiplication,
// negative m / n by conjugation, S(k) = sum_i q_i exp(i k r_i), E = scale * sum_k akk |S(k)|^2,
// F_i = scale2 * sum_k akk * Im(conj(S(k)) q_i exp(i k r_i)) * k.
//
// Here nothing per (atom, k) ever touches HBM: both kernels rebuild the three small per-atom harmonic tables in LDS.
//   k_ewald_sfac    threads <-> k-vectors, 64 atoms per tile in LDS ([atom][harmonic]: the 64 lanes read the same atom, different
//                   harmonics -> broadcasts); per-block partial S(k) in a fixed order -> bit-reproducible
//   k_ewald_reduce  S(k) = sum over blocks (fixed order)                     [multi-GPU: all-reduce of S over the ranks follows]
//   k_ewald_force   threads <-> atoms, the k loop is wave-uniform (k-vector data and S(k) come through scalar loads), harmonic
//                   tables in LDS as [harmonic][lane] (conflict-free); exp(i(lx+my)) is reused along the inner n loop
//   k_ewald_energy  E = scale * sum_k akk |S(k)|^2
// No MFMA: S(k) is a sum of products of three per-atom table entries, not a contraction of stored matrices.
#pragma once
#include <hip/hip_runtime.h>

#include "device_md.h"
#include "kernels.hip.h"

namespace aztot {

constexpr int kEwTile = 64;                 // atoms per LDS tile (k_ewald_sfac) = lanes per block (k_ewald_force)
struct cplx { double c, s; };
__device__ __forceinline__ cplx cmul(cplx a, cplx b) { return cplx{a.c * b.c - a.s * b.s, a.s * b.c + b.s * a.c}; }

// harmonics 0..n-1 of exp(i * arg) by the reference's recurrence (elec.cpp:213-226); out[h * stride]
__device__ __forceinline__ void ew_harmonics(double arg, int n, cplx* out, int stride, double pref)
{
    cplx e1; sincos(arg, &e1.s, &e1.c);
    cplx cur = cplx{1.0, 0.0};
    for (int h = 0; h < n; h++)
    {
        out[h * stride] = cplx{pref * cur.c, pref * cur.s};
        cur = (h == 0) ? e1 : cmul(cur, e1);
    }
}

// work item of k_ewald_sfac: the pair of k-vectors (l, m, +n), (l, m, -n) - they share exp(i(lx + my)) and, up to conjugation,
// exp(inz), which halves the LDS traffic per k-vector - or a single one (n == 0, or l == m == 0 where only n >= 1 is visited)
__global__ __launch_bounds__(256) void k_ewald_sfac(StepParams P, SpecTable S, AtomArrays A, const Counts* __restrict__ cnt, EwaldTables E)
{
    extern __shared__ double ew_lds[];
    cplx* tab = (cplx*)ew_lds;                                  // [kEwTile][nH]
    const int nH = E.kx + E.ky + E.kz;
    const int nOwned = cnt->ownedEnd - cnt->ownedBegin;
    const double twopi = 2.0 * 3.14159265359;                   // const.h:11-12
    double* mine = E.partial + (size_t)blockIdx.x * E.nK * 2;
    for (int it = 0;; it++)
    {
        const int base = (blockIdx.x + it * gridDim.x) * kEwTile;
        if (it > 0 && base >= nOwned) break;                    // the first round always runs: it initialises this block's row
        const int nAt = max(0, min(kEwTile, nOwned - base));
        __syncthreads();
        if (threadIdx.x < 3 * kEwTile)
        {   // one thread per (atom, axis): that axis' harmonics; the charge rides on the x-table; the tile is padded with
            // zero-charge atoms so that the accumulation loop below has a constant trip count
            const int a = threadIdx.x % kEwTile, ax = threadIdx.x / kEwTile;
            cplx* row = tab + a * nH;
            const int i = cnt->ownedBegin + base + min(a, max(nAt - 1, 0));
            const bool real = a < nAt;
            if (ax == 0) ew_harmonics(real ? twopi * A.x[i] * P.invL[0] : 0.0, E.kx, row, 1, real ? S.charge[A.type[i]] : 0.0);
            else if (ax == 1) ew_harmonics(real ? twopi * A.y[i] * P.invL[1] : 0.0, E.ky, row + E.kx, 1, 1.0);
            else ew_harmonics(real ? twopi * A.z[i] * P.invL[2] : 0.0, E.kz, row + E.kx + E.ky, 1, 1.0);
        }
        __syncthreads();
        for (int w = threadIdx.x; w < E.nW; w += blockDim.x)
        {
            const EwaldW wi = E.work[w];
            const cplx* px = tab + wi.l;
            const cplx* pm = tab + E.kx + abs(wi.m);
            const cplx* pn = tab + E.kx + E.ky + wi.n;
            const double sm = wi.m < 0 ? -1.0 : 1.0;
            double pc = 0.0, ps = 0.0, mc = 0.0, ms = 0.0;      // sums for +n and -n
#pragma unroll 8
            for (int a = 0; a < kEwTile; a++)
            {
                const cplx ex = px[a * nH];
                cplx em = pm[a * nH];
                const cplx en = pn[a * nH];
                em.s *= sm;                                     // negative m: complex conjugate (elec.cpp:258-262)
                const cplx lm = cmul(ex, em);
                const double cc = lm.c * en.c, ss = lm.s * en.s, sc = lm.s * en.c, cs = lm.c * en.s;
                pc += cc - ss; ps += sc + cs;                   // (l, m, +n): elec.cpp:287-288
                mc += cc + ss; ms += sc - cs;                   // (l, m, -n): conjugate of exp(inz), elec.cpp:300-301
            }
            if (wi.kPlus >= 0)
            {
                if (it == 0) { mine[2 * wi.kPlus] = pc; mine[2 * wi.kPlus + 1] = ps; }
                else { mine[2 * wi.kPlus] += pc; mine[2 * wi.kPlus + 1] += ps; }
            }
            if (wi.kMinus >= 0)
            {
                if (it == 0) { mine[2 * wi.kMinus] = mc; mine[2 * wi.kMinus + 1] = ms; }
                else { mine[2 * wi.kMinus] += mc; mine[2 * wi.kMinus + 1] += ms; }
            }
        }
    }
}

// S(k) = sum over the blocks' partial rows, fixed order: 64 k-vectors x 16 row groups per workgroup (coalesced 1 KB row
// segments), the 16 group sums folded through LDS in group order
constexpr int kEwRedGroups = 16;
__global__ __launch_bounds__(64 * kEwRedGroups) void k_ewald_reduce(EwaldTables E)
{
    __shared__ double red[kEwRedGroups][64][2];
    const int kl = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int k = blockIdx.x * 64 + kl;
    double sc = 0.0, ss = 0.0;
    if (k < E.nK)
        for (int b = g; b < E.nBlocksA; b += kEwRedGroups)
        {
            const double* row = E.partial + ((size_t)b * E.nK + k) * 2;
            sc += row[0]; ss += row[1];
        }
    red[g][kl][0] = sc; red[g][kl][1] = ss;
    __syncthreads();
    if (g == 0 && k < E.nK)
    {
        double c = 0.0, s2 = 0.0;
        for (int q = 0; q < kEwRedGroups; q++) { c += red[q][kl][0]; s2 += red[q][kl][1]; }
        E.S[2 * k] = c; E.S[2 * k + 1] = s2;
    }
}

// E = scale * sum_k akk |S(k)|^2, and the per-k factors of the force kernel: T(k) = scale2 * akk * S(k)
__global__ __launch_bounds__(256) void k_ewald_energy(EwaldTables E, DevStats* st)
{
    __shared__ double scratch[4];
    double e = 0.0;
    for (int k = threadIdx.x; k < E.nK; k += blockDim.x)
    {
        const double sc = E.S[2 * k], ss = E.S[2 * k + 1];
        const double akk = E.kv[k].akk;
        e += akk * (sc * sc + ss * ss);
        E.T[2 * k] = (akk * E.scale2) * sc; E.T[2 * k + 1] = (akk * E.scale2) * ss;
    }
    e = block_sum(e, scratch);
    if (threadIdx.x == 0) st->engCoulRec = E.scale * e;         // engElec2, elec.cpp:333
}

// Forces.  KS waves per block share one tile of 64 atoms (lane <-> atom); wave `slice` takes every KS-th (l, m) group of the
// k-vector list, so the chip sees KS x nAtoms/64 waves.  Inside a group n runs over a contiguous range: exp(i(lx + my)) is
// formed once, rkx and rky are constant, rkz = n * 2pi/c - per k-vector one LDS read (exp(inz)), one 16-byte scalar load
// (T(k)), a complex product and 4 more flops:  x = Im(conj(T) q e^{ikr}) ; F += (rkx, rky, n 2pi/c) x   (elec.cpp:317-322)
constexpr int kEwSlices = 8;

__global__ __launch_bounds__(kEwTile* kEwSlices) void k_ewald_force(StepParams P, SpecTable S, AtomArrays A, const Counts* __restrict__ cnt,
                                                                     EwaldTables E)
{
    extern __shared__ double ew_lds[];
    cplx* tab = (cplx*)ew_lds;                                  // [nH][kEwTile]
    const int nH = E.kx + E.ky + E.kz;
    double* red = ew_lds + 2 * (size_t)nH * kEwTile;            // [kEwSlices][3][kEwTile]
    const int lane = threadIdx.x & (kEwTile - 1);
    const int slice = __builtin_amdgcn_readfirstlane(threadIdx.x / kEwTile);     // wave-uniform: keeps the k loops on scalar registers
    const int i0 = cnt->ownedBegin + blockIdx.x * kEwTile;
    const double twopi = 2.0 * 3.14159265359;
    if (threadIdx.x < 3 * kEwTile)
    {
        const int a = threadIdx.x % kEwTile, ax = threadIdx.x / kEwTile;
        const bool real = i0 + a < cnt->ownedEnd;
        const int i = real ? i0 + a : cnt->ownedBegin;
        if (ax == 0) ew_harmonics(real ? twopi * A.x[i] * P.invL[0] : 0.0, E.kx, tab + a, kEwTile, real ? S.charge[A.type[i]] : 0.0);
        else if (ax == 1) ew_harmonics(real ? twopi * A.y[i] * P.invL[1] : 0.0, E.ky, tab + E.kx * kEwTile + a, kEwTile, 1.0);
        else ew_harmonics(real ? twopi * A.z[i] * P.invL[2] : 0.0, E.kz, tab + (E.kx + E.ky) * kEwTile + a, kEwTile, 1.0);
    }
    __syncthreads();
    double fx = 0.0, fy = 0.0, fz = 0.0;
    const cplx* tm = tab + E.kx * kEwTile + lane;
    const cplx* tn = tab + (E.kx + E.ky) * kEwTile + lane;
    const double* __restrict__ T = E.T;
    for (int g = slice; g < E.nG; g += kEwSlices)
    {
        const EwaldG G = E.groups[g];                           // wave-uniform: scalar load
        const cplx ex = tab[G.l * kEwTile + lane];
        cplx em = tm[abs(G.m) * kEwTile];
        if (G.m < 0) em.s = -em.s;
        const cplx lm = cmul(ex, em);
        double sx = 0.0, sz = 0.0;
        const double* Tg = T + 2 * (size_t)(G.kStart - G.nLo);
#pragma unroll 4
        for (int n = G.nLo; n <= G.nHi; n++)
        {
            cplx en = tn[abs(n) * kEwTile];
            if (n < 0) en.s = -en.s;
            const cplx ck = cmul(lm, en);
            const double x = ck.s * Tg[2 * n] - ck.c * Tg[2 * n + 1];
            sx += x;
            sz = fma((double)n, x, sz);
        }
        fx = fma((double)G.l * (twopi * P.invL[0]), sx, fx);
        fy = fma((double)G.m * (twopi * P.invL[1]), sx, fy);
        fz += sz;
    }
    fz *= twopi * P.invL[2];
    red[(slice * 3 + 0) * kEwTile + lane] = fx;
    red[(slice * 3 + 1) * kEwTile + lane] = fy;
    red[(slice * 3 + 2) * kEwTile + lane] = fz;
    __syncthreads();
    if (threadIdx.x < 3 * kEwTile)
    {
        const int a = threadIdx.x % kEwTile, c = threadIdx.x / kEwTile;
        double f = 0.0;
        for (int sl = 0; sl < kEwSlices; sl++) f += red[(sl * 3 + c) * kEwTile + a];
        const int i = i0 + a;
        if (i < cnt->ownedEnd)
        {
            double* dst = (c == 0) ? A.fx : (c == 1 ? A.fy : A.fz);
            dst[i] += f;
        }
    }
}

}  // namespace aztot
