// Bonded terms: constant bonds + harmonic-cosine angles ("next" row f2 of SURVEY section 8, spec in Appendix G).
//
// Reference: apply_const_bonds cuBonds.cu:709-796 (one thread per slice of the bond list, 6 fp32 atomics per bond) and
// apply_angles cuAngles.cu:169-228 (9 atomics per angle); serial twins exec_bondlist bonds.cpp:1069-1218 and
// exec_anglelist angles.cpp:229-242.  Here: one thread per OWNED atom gathers the terms that atom takes part in from a
// static CSR table keyed by atom id (a bond is evaluated at both ends, an angle at its three atoms), so forces are
// accumulated in registers in a fixed order - no atomics, deterministic - and added to what the pair kernel wrote.
// Partners are found through idxOfId, which k_rank_gather refreshes with every sort; on a slab rank a partner is an
// owned or a ghost atom, and a partner that is not resident (bond longer than the halo) raises Counts::bondedMissing.
#pragma once
#include <hip/hip_runtime.h>

#include "device_md.h"
#include "kernels.hip.h"

namespace aztot {

// bond_iter bonds.cpp:731-787 / bond_harm..bond_e6812 cuBonds.cu:1123-1224: returns -(1/r) dU/dr, adds U to eng
__device__ __forceinline__ double bond_force(const DevBondType& b, double r2, double& eng)
{
    const double r = sqrt(r2);
    switch (b.type)
    {
    case 1:
    {
        const double x = r - b.p1;
        eng += 0.5 * b.p0 * x * x;
        return -b.p0 / r * x;
    }
    case 2:
    {
        const double x = exp(-b.p1 * (r - b.p2)), y = 1.0 - x;
        eng += b.p0 * y * y - b.p3;
        return -2.0 * b.p0 * b.p1 * x * y / r;
    }
    case 3:
    {
        const double x = exp(-b.p1 * (r - b.p2)), y = 1.0 - x;
        const double ir2 = 1.0 / r2;
        double irn = ir2 * ir2; irn = irn * irn * irn;
        eng += b.p0 * y * y - b.p3 - b.p4 * irn;
        return -2.0 * b.p0 * b.p1 * x * y / r - 12.0 * b.p4 * irn * ir2;
    }
    case 4:
    {
        const double ir2 = 1.0 / r2, irn = ir2 * ir2, e = b.p0 * exp(-r / b.p1);
        eng += e - b.p2 * irn * ir2;
        return e / r / b.p1 - 6.0 * b.p2 * irn * irn;
    }
    case 5:
    {
        const double ir2 = 1.0 / r2, irn = ir2 * ir2, e = b.p0 * exp(-r / b.p1);
        eng += e - b.p2 * irn * ir2 - b.p3 * irn * irn - b.p4 * irn * irn * irn;
        return e / r / b.p1 - 6.0 * b.p2 * irn * irn - 8.0 * b.p3 * irn * irn * ir2 - 12.0 * b.p4 * irn * irn * irn * ir2;
    }
    }
    return 0.0;
}

// VERIFY (slab ranks): the map may hold a stale entry for an atom that left this rank, so the id is read back; on one GPU every
// atom is resident and every entry was written by the sort that has just run
template <bool VERIFY>
__device__ __forceinline__ int resident_index(const BondedTables& B, const AtomArrays& A, int id, int nTotal)
{
    const int j = B.idxOfId[id];
    if (!VERIFY) return j;
    return (j >= 0 && j < nTotal && A.id[j] == id) ? j : -1;
}

template <bool VERIFY>
__global__ __launch_bounds__(kBlock) void k_bonded(StepParams P, AtomArrays A, Counts* __restrict__ cnt, BondedTables B,
                                                   double* __restrict__ partials, int maxBlocks)
{
    __shared__ double scratch[kBlock / kWave];
    const int i = cnt->ownedBegin + blockIdx.x * kBlock + threadIdx.x;
    const int nTotal = cnt->nTotal;
    double eB = 0.0, eA = 0.0;
    if (i < cnt->ownedEnd)
    {
        const int me = A.id[i];
        const double xi = A.x[i], yi = A.y[i], zi = A.z[i];
        double fx = 0.0, fy = 0.0, fz = 0.0;
        bool missing = false;
        for (int k = B.bondStart[me], ke = B.bondStart[me + 1]; k < ke; k++)
        {
            const BondEntry en = B.bondEnt[k];
            const int j = resident_index<VERIFY>(B, A, en.partner, nTotal);
            if (j < 0) { missing = true; continue; }
            double dx = xi - A.x[j], dy = yi - A.y[j], dz = zi - A.z[j];       // sqr_distance_proj, box.cpp:327-335
            min_image(dx, P.L[0], P.half[0]); min_image(dy, P.L[1], P.half[1]); min_image(dz, P.L[2], P.half[2]);
            double e = 0.0;
            const double f = bond_force(B.btypes[en.typeFirst & 0xffff], dx * dx + dy * dy + dz * dz, e);
            fx += f * dx; fy += f * dy; fz += f * dz;
            if (en.typeFirst >> 30) eB += e;                                      // the energy is booked once, at at1
        }
        for (int k = B.angStart[me], ke = B.angStart[me + 1]; k < ke; k++)
        {   // angle_iter angles.cpp:179-227 ; angle_hcos cuAngles.cu:230-284
            const AngleEntry en = B.angEnt[k];
            const int role = en.roleType & 3;
            const int jc = (role == 0) ? i : resident_index<VERIFY>(B, A, en.c, nTotal);
            const int j1 = (role == 1) ? i : resident_index<VERIFY>(B, A, en.l1, nTotal);
            const int j2 = (role == 2) ? i : resident_index<VERIFY>(B, A, en.l2, nTotal);
            if (jc < 0 || j1 < 0 || j2 < 0) { missing = true; continue; }
            const DevAngleType at = B.atypes[en.roleType >> 2];
            const double xc = A.x[jc], yc = A.y[jc], zc = A.z[jc];
            double xij = A.x[j1] - xc, yij = A.y[j1] - yc, zij = A.z[j1] - zc;
            min_image(xij, P.L[0], P.half[0]); min_image(yij, P.L[1], P.half[1]); min_image(zij, P.L[2], P.half[2]);
            const double r2ij = xij * xij + yij * yij + zij * zij, rij = sqrt(r2ij);
            double xik = A.x[j2] - xc, yik = A.y[j2] - yc, zik = A.z[j2] - zc;
            min_image(xik, P.L[0], P.half[0]); min_image(yik, P.L[1], P.half[1]); min_image(zik, P.L[2], P.half[2]);
            const double r2ik = xik * xik + yik * yik + zik * zik, rik = sqrt(r2ik);
            const double cos_th = (xij * xik + yij * yik + zij * zik) / rij / rik;
            const double dCos = cos_th - at.cos0;
            const double c1 = -at.k * dCos, c2 = 1.0 / rij / rik;
            if (role == 0)
            {
                fx += -c1 * (xik * c2 + xij * c2 - cos_th * (xij / r2ij + xik / r2ik));
                fy += -c1 * (yik * c2 + yij * c2 - cos_th * (yij / r2ij + yik / r2ik));
                fz += -c1 * (zik * c2 + zij * c2 - cos_th * (zij / r2ij + zik / r2ik));
                eA += 0.5 * at.k * dCos * dCos;                                   // booked once, at the central atom
            }
            else if (role == 1)
            {
                fx += c1 * (xik * c2 - cos_th * xij / r2ij);
                fy += c1 * (yik * c2 - cos_th * yij / r2ij);
                fz += c1 * (zik * c2 - cos_th * zij / r2ij);
            }
            else
            {
                fx += c1 * (xij * c2 - cos_th * xik / r2ik);
                fy += c1 * (yij * c2 - cos_th * yik / r2ik);
                fz += c1 * (zij * c2 - cos_th * zik / r2ik);
            }
        }
        A.fx[i] += fx; A.fy[i] += fy; A.fz[i] += fz;
        if (missing) atomicOr(&cnt->bondedMissing, 1);
    }
    const double sB = block_sum(eB, scratch);
    const double sA = block_sum(eA, scratch);
    if (threadIdx.x == 0)
    {
        put_partial(partials, maxBlocks, PS_EBOND, sB);
        put_partial(partials, maxBlocks, PS_EANGLE, sA);
    }
}

}  // namespace aztot
