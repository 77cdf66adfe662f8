// Counter-based RNG shared by host tables and device kernels.
// Replaces the reference's racy xorshift128 on an uninitialised seed (cuUtils.cu:89-105,
// cuStruct.h:255; SURVEY C-11): a draw is a pure function of (seed, step, persistent atom id,
// draw index), so results do not depend on sort order, launch geometry or GPU count.
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#define AZTOT_HD __host__ __device__ inline
#else
#define AZTOT_HD inline
#endif

namespace aztot {

AZTOT_HD uint64_t mix64(uint64_t z)
{
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ULL;
    z ^= z >> 27; z *= 0x94D049BB133111EBULL;
    z ^= z >> 31;
    return z;
}

AZTOT_HD uint32_t rng_draw(uint64_t seed, uint64_t step, uint64_t id, uint64_t draw)
{
    uint64_t z = mix64(seed + 0x9E3779B97F4A7C15ULL * (step + 1));
    z = mix64(z ^ (0xD1B54A32D192ED03ULL * (id + 1)));
    z = mix64(z ^ (0x8CB92BA72F3D8DD7ULL * (draw + 1)));
    return (uint32_t)(z >> 32);
}

constexpr uint64_t kRngStreamTables = 0xFFFFFFFFFFFFFFF0ULL;   // "step" reserved for table generation
constexpr uint64_t kRngStreamInitVel = 0xFFFFFFFFFFFFFFF1ULL;  // "step" reserved for init_vel

}  // namespace aztot
