// Diagnostic builds of the fused mean-field updates (window constants uniform, every radius): the same kernel templates
// with TRACE set, which also write the softmax argument unary + weight * compatibility (find_surfaces.py:361) and the new
// probabilities of all S labels of every pixel with a depth.  Used by fs_find_surfaces_trace, through which the tests
// compare the production arithmetic of every update with a CPU evaluation; no other entry point launches them.
#include "k_crf_fused.h"
#include "k_crf_roll.h"

namespace fs {

namespace {

template <int R>
hipError_t roll_by_class(int channel_class, const FusedLaunch &L)
{
    switch (channel_class) {
    case 1:
    case 2: return launch_roll_t<R, 2, 0, true, true>(L);
    case 3:
    case 4: return launch_roll_t<R, 4, 0, true, true>(L);
    case 8: return launch_roll_t<R, 8, 0, true, true>(L);
    case 16: return launch_roll_t<R, 16, 0, true, true>(L);
    default: return hipErrorInvalidValue;
    }
}

}  // namespace

hipError_t launch_fused_trace(int channel_class, const FusedLaunch &L)
{
    if (L.g.r == 7) {
        switch (channel_class) {
        case 1: return launch_fused_t<7, 16, 1, 0, true, true>(L);
        case 2: return launch_fused_t<7, 16, 2, 0, true, true>(L);
        case 3:
        case 4: return launch_fused_t<7, 16, 4, 0, true, true>(L);
        case 8: return launch_fused_t<7, 16, 8, 0, true, true>(L);
        case 16: return launch_fused_t<7, 16, 16, 0, true, true>(L);
        default: return hipErrorInvalidValue;
        }
    }
    if (L.g.r == 10) return roll_by_class<10>(channel_class, L);
    if (L.g.r == 15) return roll_by_class<15>(channel_class, L);
    return roll_by_class<0>(channel_class, L);
}

}  // namespace fs
