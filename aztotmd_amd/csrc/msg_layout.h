// Layout of the fixed-capacity slab message (see slab.hip.h).
#pragma once
#include <cstddef>
#include <cstdint>

#if defined(__HIPCC__)
#define AZTOT_LAYOUT_FN __host__ __device__
#else
#define AZTOT_LAYOUT_FN
#endif

namespace aztot {

struct SendHeader { int32_t nMig, nHalo, pad0, pad1; };
struct MigRec { double x, y, z, vx, vy, vz, U, rad; int32_t type, id, pad0, pad1; };   // 80 B
struct HaloRec { double x, y, z, rad; int32_t type, id; };                             // 40 B

struct MsgLayout
{
    int32_t migCap = 0, haloCap = 0;
    AZTOT_LAYOUT_FN size_t mig_offset() const { return sizeof(SendHeader); }
    AZTOT_LAYOUT_FN size_t halo_offset() const { return sizeof(SendHeader) + sizeof(MigRec) * (size_t)migCap; }
    AZTOT_LAYOUT_FN size_t bytes() const { return halo_offset() + sizeof(HaloRec) * (size_t)haloCap; }
};

}  // namespace aztot
