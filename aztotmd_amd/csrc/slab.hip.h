// Slab (x) domain decomposition across the GPUs of one node: device side.
// The reference is single-GPU (device 0 hard-coded, cuInit.cu:688); this part is new design.
//
// Per step and per x-neighbour ONE fixed-capacity message travels:
//     [SendHeader | migrant records | halo records]
// * migrants: atoms this rank integrated whose new cell layer belongs to the neighbour (full state);
//   the sender KEEPS them as ghosts (they sit exactly in its ghost layer), so nothing comes back;
// * halo: this rank's owned atoms in its hw boundary layers (position, type, id, radius only).
// Packing (inside k_integrate1_bin) uses wave64 ballot + popcount prefix with one atomic per wave and category; the receiver reads
// the counts from the header on the device, so a step needs no host round trip.  Ghost coordinates stay
// global: distances go through the same minimum-image arithmetic as on one GPU, and together with the
// id-ordered cells this makes N-GPU forces bit-identical to 1-GPU forces.
#pragma once
#include "kernels.hip.h"
#include "msg_layout.h"

namespace aztot {

// append what the two neighbours sent behind the owned range and add it to the cell histogram
__global__ __launch_bounds__(kBlock) void k_unpack(StepParams P, AtomArrays A, Counts* cnt, int capacity, MsgLayout lay,
                                                   const char* __restrict__ fromLeft, const char* __restrict__ fromRight,
                                                   int32_t* __restrict__ cellOf, int32_t* __restrict__ slotOf, int32_t* __restrict__ cellCount,
                                                   char* __restrict__ sendLeft, char* __restrict__ sendRight)
{
    const SendHeader hL = *(const SendHeader*)fromLeft, hR = *(const SendHeader*)fromRight;
    const int nML = min(hL.nMig, lay.migCap), nHL = min(hL.nHalo, lay.haloCap);
    const int nMR = min(hR.nMig, lay.migCap), nHR = min(hR.nHalo, lay.haloCap);
    const int total = nML + nHL + nMR + nHR;
    const int base = cnt->ownedEnd;
    int t = blockIdx.x * kBlock + threadIdx.x;
    if (t == 0)
    {
        // the send buffers have travelled: clear their counters for the next step's packing
        SendHeader z = {0, 0, 0, 0};
        *(SendHeader*)sendLeft = z; *(SendHeader*)sendRight = z;
        if (base + total > capacity) cnt->overflow = 1;
        cnt->nRecv = min(total, max(capacity - base, 0));
    }
    if (t >= total || base + t >= capacity) return;
    const int d = base + t;
    double x, y, z, rad; int type, id;
    const MigRec* mig = nullptr;
    const HaloRec* halo = nullptr;
    if (t < nML) mig = (const MigRec*)(fromLeft + lay.mig_offset()) + t;
    else if ((t -= nML) < nHL) halo = (const HaloRec*)(fromLeft + lay.halo_offset()) + t;
    else if ((t -= nHL) < nMR) mig = (const MigRec*)(fromRight + lay.mig_offset()) + t;
    else halo = (const HaloRec*)(fromRight + lay.halo_offset()) + (t - nMR);
    if (mig)
    {
        x = mig->x; y = mig->y; z = mig->z; rad = mig->rad; type = mig->type; id = mig->id;
        A.vx[d] = mig->vx; A.vy[d] = mig->vy; A.vz[d] = mig->vz; A.U[d] = mig->U;
    }
    else
    {
        x = halo->x; y = halo->y; z = halo->z; rad = halo->rad; type = halo->type; id = halo->id;
        A.vx[d] = 0.0; A.vy[d] = 0.0; A.vz[d] = 0.0; A.U[d] = 0.0;
    }
    A.x[d] = x; A.y[d] = y; A.z[d] = z; A.rad[d] = rad; A.type[d] = type; A.id[d] = id;
    const int c = local_cell(P, x, y, z);
    cellOf[d] = c;
    slotOf[d] = atomicAdd(&cellCount[c], 1);
}

}  // namespace aztot
