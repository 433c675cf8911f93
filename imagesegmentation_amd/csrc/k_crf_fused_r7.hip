// Fused mean-field update, window radius 7 (the reference's standard_kernel_size, find_surfaces.py:23) with the
// window's constants in scalar registers: the production variants, one per channel class.
#include "k_crf_fused.h"

namespace fs {

hipError_t launch_fused_r7_uniform(int channel_class, const FusedLaunch &L)
{
    switch (channel_class) {
    case 1: return launch_fused_t<7, 16, 1, 0, true>(L);
    case 2: return launch_fused_t<7, 16, 2, 0, true>(L);
    case 3:
    case 4: return launch_fused_t<7, 16, 4, 0, true>(L);
    case 8: return launch_fused_t<7, 16, 8, 0, true>(L);
    case 16: return launch_fused_t<7, 16, 16, 0, true>(L);
    default: return hipErrorInvalidValue;
    }
}

}  // namespace fs

#if FS_TILE_CLOCKS
// experiment build only: the phase clocks of every r = 7 update since the last call (k_crf_fused.h), then cleared
extern "C" __attribute__((visibility("default"))) int fs_debug_phase_clocks(unsigned long long out[12])
{
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(fs::g_phase_clocks), 12 * sizeof(unsigned long long)) != hipSuccess) return -1;
    const unsigned long long zero[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    return hipMemcpyToSymbol(HIP_SYMBOL(fs::g_phase_clocks), zero, sizeof(zero)) == hipSuccess ? 0 : -1;
}
#endif
