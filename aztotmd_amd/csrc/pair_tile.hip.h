// LDS-tiled pair kernel (variant 2) - placeholder until the tiled kernel lands.
#pragma once
#include "kernels.hip.h"

namespace aztot {
inline bool pair_tile_supported(const StepParams&) { return false; }
inline int pair_tile_grid(const StepParams&) { return 0; }
inline void launch_pair_tile(const StepParams&, const SpecTable&, const DevPot*, AtomArrays, const Counts*, const int32_t*, double*, int, hipStream_t) {}
}  // namespace aztot
