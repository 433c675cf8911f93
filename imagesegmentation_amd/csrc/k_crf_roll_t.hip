// Rolling-window mean-field update with per-offset constants: parameter sets whose position terms do not vanish
// (variant 1) and the logistic-regression similarity of conv_crf_LR (variant 2); run-time radius.  Also the passes
// of wide updates (tiles with more channels than the kernels keep in registers), for every variant.
#include "k_crf_roll.h"

namespace fs {

hipError_t launch_roll_table(int variant, int channel_class, const FusedLaunch &L)
{
    if (variant == 2)       // two channel classes only: it is not the tuned path
        return channel_class <= 2 ? launch_roll_t<0, 2, 1, false>(L)
                                  : (channel_class <= 16 ? launch_roll_t<0, 16, 1, false>(L) : hipErrorInvalidValue);
    if (variant != 1) return hipErrorInvalidValue;
    switch (channel_class) {
    case 1:
    case 2: return launch_roll_t<0, 2, 0, false>(L);
    case 3:
    case 4: return launch_roll_t<0, 4, 0, false>(L);
    case 8: return launch_roll_t<0, 8, 0, false>(L);
    case 16: return launch_roll_t<0, 16, 0, false>(L);
    default: return hipErrorInvalidValue;
    }
}

// Passes of a wide update: run-time radius, 128 channel slots per pass, every variant (it is not a tuned path: tiles
// with more than 128 channels come from frames cut into hundreds of small patches).
hipError_t launch_roll_wide(int variant, const FusedLaunch &L)
{
    switch (variant) {
    case 0: return launch_roll_t<0, 16, 0, true, false, true>(L);
    case 1: return launch_roll_t<0, 16, 0, false, false, true>(L);
    case 2: return launch_roll_t<0, 16, 1, false, false, true>(L);
    default: return hipErrorInvalidValue;
    }
}

}  // namespace fs
