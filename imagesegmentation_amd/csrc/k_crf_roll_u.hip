// Rolling-window mean-field update, window constants uniform (the trained parameters): compile-time radii 10 and
// 15 (the larger windows of the 1280x960 configuration) and a run-time radius for every other one.
#include "k_crf_roll.h"

namespace fs {

namespace {

template <int R>
hipError_t by_class(int channel_class, const FusedLaunch &L)
{
    switch (channel_class) {
    case 1:
    case 2: return launch_roll_t<R, 2, 0, true>(L);
    case 3:
    case 4: return launch_roll_t<R, 4, 0, true>(L);
    case 8: return launch_roll_t<R, 8, 0, true>(L);
    case 16: return launch_roll_t<R, 16, 0, true>(L);
    default: return hipErrorInvalidValue;
    }
}

}  // namespace

hipError_t launch_roll_uniform(int channel_class, const FusedLaunch &L)
{
    if (L.g.r == 10) return by_class<10>(channel_class, L);
    if (L.g.r == 15) return by_class<15>(channel_class, L);
    return by_class<0>(channel_class, L);
}

}  // namespace fs
