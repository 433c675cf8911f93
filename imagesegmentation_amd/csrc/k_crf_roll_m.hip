// Rolling-window mean-field update for tiles with many labels around them: all of a tile's channels (12, 16 or 32 per
// walk of the window, as many as the ring holds at that radius) take their messages from one evaluation of the
// similarity, instead of 8 per walk with the similarity evaluated again in every walk.  Trained parameters (uniform
// window constants), compile-time radii 7, 10 and 15: a tile of a scene with many small surfaces sees 20 labels at r = 7 and
// 30 to 50 with the larger windows of the 1280x960 configuration.
#include "k_crf_roll.h"

namespace fs {

// quads: quad planes of Q in the ring (4 channels each).  3 and 4 run two blocks per CU, 8 one block per CU with two
// walks (64 channel slots per tile).
hipError_t launch_roll_multi(int quads, const FusedLaunch &L)
{
    if (L.g.r == 7) {
        if (quads == 4) return launch_roll_t<7, 4, 0, true, false, false, 4>(L);
    } else if (L.g.r == 10) {
        if (quads == 4) return launch_roll_t<10, 4, 0, true, false, false, 4>(L);
    } else if (L.g.r == 15) {
        if (quads == 3) return launch_roll_t<15, 4, 0, true, false, false, 3>(L);
    }
    return hipErrorInvalidValue;
}

}  // namespace fs
