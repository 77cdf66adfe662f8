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
        {   // one thread per (atom, axis): that axis' harmonics; the charge rides on the x-table
            const int a = threadIdx.x % kEwTile, ax = threadIdx.x / kEwTile;
            if (a < nAt)
            {
                const int i = cnt->ownedBegin + base + a;
                cplx* row = tab + a * nH;
                if (ax == 0) ew_harmonics(twopi * A.x[i] * P.invL[0], E.kx, row, 1, S.charge[A.type[i]]);
                else if (ax == 1) ew_harmonics(twopi * A.y[i] * P.invL[1], E.ky, row + E.kx, 1, 1.0);
                else ew_harmonics(twopi * A.z[i] * P.invL[2], E.kz, row + E.kx + E.ky, 1, 1.0);
            }
        }
        __syncthreads();
        for (int k = threadIdx.x; k < E.nK; k += blockDim.x)
        {
            const EwaldK kv = E.kv[k];
            const int io = kv.l, im = E.kx + abs(kv.m), in = E.kx + E.ky + abs(kv.n);
            const double sm = kv.m < 0 ? -1.0 : 1.0, sn = kv.n < 0 ? -1.0 : 1.0;
            double sc = 0.0, ss = 0.0;
            for (int a = 0; a < nAt; a++)
            {
                const cplx* row = tab + a * nH;
                const cplx ex = row[io];
                cplx em = row[im], en = row[in];
                em.s *= sm; en.s *= sn;                         // negative m / n: complex conjugate (elec.cpp:258-262,296-307)
                const cplx ck = cmul(cmul(ex, em), en);
                sc += ck.c; ss += ck.s;
            }
            if (it == 0) { mine[2 * k] = sc; mine[2 * k + 1] = ss; }
            else { mine[2 * k] += sc; mine[2 * k + 1] += ss; }
        }
    }
}

__global__ __launch_bounds__(256) void k_ewald_reduce(EwaldTables E)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= E.nK) return;
    double sc = 0.0, ss = 0.0;
    for (int b = 0; b < E.nBlocksA; b++)
    {
        const double* row = E.partial + ((size_t)b * E.nK + k) * 2;
        sc += row[0]; ss += row[1];
    }
    E.S[2 * k] = sc; E.S[2 * k + 1] = ss;
}

__global__ __launch_bounds__(256) void k_ewald_energy(EwaldTables E, DevStats* st)
{
    __shared__ double scratch[4];
    double e = 0.0;
    for (int k = threadIdx.x; k < E.nK; k += blockDim.x)
    {
        const double sc = E.S[2 * k], ss = E.S[2 * k + 1];
        e += E.kv[k].akk * (sc * sc + ss * ss);
    }
    e = block_sum(e, scratch);
    if (threadIdx.x == 0) st->engCoulRec = E.scale * e;         // engElec2, elec.cpp:333
}

__global__ __launch_bounds__(kEwTile) void k_ewald_force(StepParams P, SpecTable S, AtomArrays A, const Counts* __restrict__ cnt, EwaldTables E)
{
    extern __shared__ double ew_lds[];
    cplx* tab = (cplx*)ew_lds;                                  // [nH][kEwTile]
    const int lane = threadIdx.x;
    const int i = cnt->ownedBegin + blockIdx.x * kEwTile + lane;
    const bool valid = i < cnt->ownedEnd;
    const double twopi = 2.0 * 3.14159265359;
    if (valid)
    {
        ew_harmonics(twopi * A.x[i] * P.invL[0], E.kx, tab + lane, kEwTile, S.charge[A.type[i]]);
        ew_harmonics(twopi * A.y[i] * P.invL[1], E.ky, tab + E.kx * kEwTile + lane, kEwTile, 1.0);
        ew_harmonics(twopi * A.z[i] * P.invL[2], E.kz, tab + (E.kx + E.ky) * kEwTile + lane, kEwTile, 1.0);
    }
    else
        for (int h = 0; h < E.kx + E.ky + E.kz; h++) tab[h * kEwTile + lane] = cplx{0.0, 0.0};
    // a lane only ever reads its own column: no barrier needed
    double fx = 0.0, fy = 0.0, fz = 0.0;
    cplx ex = cplx{0.0, 0.0}, lm = cplx{0.0, 0.0};
    const cplx* tm = tab + E.kx * kEwTile + lane;
    const cplx* tn = tab + (E.kx + E.ky) * kEwTile + lane;
    for (int k = 0; k < E.nK; k++)
    {
        const EwaldK kv = E.kv[k];                              // wave-uniform: scalar loads
        if (kv.flags & EWK_NEW_L) ex = tab[kv.l * kEwTile + lane];
        if (kv.flags & EWK_NEW_LM)
        {
            cplx em = tm[abs(kv.m) * kEwTile];
            if (kv.m < 0) em.s = -em.s;
            lm = cmul(ex, em);
        }
        cplx en = tn[abs(kv.n) * kEwTile];
        if (kv.n < 0) en.s = -en.s;
        const cplx ck = cmul(lm, en);
        const double x = (kv.akk * E.scale2) * (ck.s * E.S[2 * k] - ck.c * E.S[2 * k + 1]);   // elec.cpp:317-319
        fx = fma(kv.rkx, x, fx); fy = fma(kv.rky, x, fy); fz = fma(kv.rkz, x, fz);
    }
    if (valid) { A.fx[i] += fx; A.fy[i] += fy; A.fz[i] += fz; }
}

}  // namespace aztot
