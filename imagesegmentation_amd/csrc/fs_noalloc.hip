// fs_connected_components_noalloc: connected components for callers that cannot allocate device memory
// (code under HIP graph capture, serving loops, harnesses that forbid hipMalloc).  The work space of the
// labelling kernels (union-find forest, component sizes, new ids, block sums: what fs_connected_components takes
// from the context's grow-only buffers) lives in a file-scope device array instead, so the call touches no
// allocator.  Inputs and outputs are the caller's device memory.
//
// One copy of the array exists per GPU, at its own address, for the life of the process.  Every call on a GPU
// shares that GPU's copy.  Calls made through this entry point are ordered among themselves: each one makes its
// stream wait for an event that the previous call on the same GPU recorded (created at set-up; waiting and
// recording allocate nothing), so two contexts or threads on one GPU take turns instead of racing.  A stream under
// capture cannot wait for an event recorded outside its capture: there the ordering is left out and overlapping
// calls race, as they do with anything else that writes the caller's arrays.
// The array keeps the previous call's contents; the kernels write every word they later read (parents and local
// sizes of all pixels, counts at the roots, all block sums), so nothing has to be cleared between calls.
#include <mutex>

#include "../../include/fs_hip.h"
#include "fs_common.h"

namespace fs {

namespace {

constexpr size_t kNoallocMaxPixels = (size_t)1280 * 960;        // all frames of one call together
constexpr size_t kNoallocBlocks = kNoallocMaxPixels / 1024 + 64;            // block sums: one per 1024 pixels, rounded up per frame
constexpr size_t kNoallocInts = 3 * kNoallocMaxPixels + 2 * kNoallocBlocks;  // 14.1 MiB

// parent | count | newid | blocksum | label counts (when the caller does not want them)
static __device__ __align__(256) int32_t scratch[kNoallocInts];

constexpr int kDevices = 64;
int32_t *g_scratch_ptr[kDevices] = {};       // device address of `scratch` on each GPU (the array's name is not one)
hipEvent_t g_scratch_free[kDevices] = {};    // recorded behind the last call that used the GPU's array
std::mutex g_scratch_mutex;
std::mutex g_call_mutex[kDevices];           // wait, launches and record of one call stay together

}  // namespace

// Looks up this GPU's copy of the work space.  `device` must be the current device.
hipError_t noalloc_scratch_init(int device, int32_t **out)
{
    if (device < 0 || device >= kDevices) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> g(g_scratch_mutex);
    if (!g_scratch_ptr[device]) {
        int32_t *p = nullptr;
        hipError_t e = hipGetSymbolAddress(reinterpret_cast<void **>(&p), HIP_SYMBOL(scratch));
        if (e != hipSuccess) return e;
        if (!p) return hipErrorInvalidSymbol;
        hipEvent_t ev = nullptr;
        e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
        if (e != hipSuccess) return e;
        g_scratch_free[device] = ev;
        g_scratch_ptr[device] = p;
    }
    if (out) *out = g_scratch_ptr[device];
    return hipSuccess;
}

// n frames of H x W pixels -> labels / label counts, work space from this GPU's static array.  Returns
// hipErrorInvalidValue when the frames do not fit.
hipError_t noalloc_connected_components(int device, const uint8_t *d_mask, int min_size_exclusive, int64_t *d_labels,
                                        int32_t *d_num_labels, int n, int H, int W, hipStream_t st)
{
    const size_t px = (size_t)n * H * W;
    const size_t blocks = (size_t)n * cc_blocks_per_image(H, W);
    if (px > kNoallocMaxPixels || blocks > kNoallocBlocks) return hipErrorInvalidValue;
    int32_t *base = nullptr;
    hipError_t e = noalloc_scratch_init(device, &base);
    if (e != hipSuccess) return e;
    CcWork w;
    w.parent = base;
    w.count = base + kNoallocMaxPixels;
    w.newid = base + 2 * kNoallocMaxPixels;
    w.blocksum = base + 3 * kNoallocMaxPixels;
    w.nlabels = d_num_labels ? d_num_labels : base + 3 * kNoallocMaxPixels + kNoallocBlocks;   // n <= blocks
    std::lock_guard<std::mutex> turn(g_call_mutex[device]);
    hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &capturing) != hipSuccess) { (void)hipGetLastError(); capturing = hipStreamCaptureStatusNone; }
    const bool ordered = capturing == hipStreamCaptureStatusNone;
    if (ordered) {
        e = hipStreamWaitEvent(st, g_scratch_free[device], 0);      // the previous call on this GPU is done with the array
        if (e != hipSuccess) return e;
    }
    launch_connected_components(d_mask, w, min_size_exclusive, d_labels, nullptr, n, H, W, st);
    e = hipGetLastError();
    if (e == hipSuccess && ordered) e = hipEventRecord(g_scratch_free[device], st);
    return e;
}

}  // namespace fs

extern "C" {

size_t fs_noalloc_max_pixels(void) { return fs::kNoallocMaxPixels; }

int fs_connected_components_noalloc(fs_ctx *ctx, int n, const uint8_t *d_mask, int min_size_exclusive, int64_t *d_labels,
                                    int32_t *d_num_labels)
{
    if (!ctx) return -1;
    const fs::CtxView v = fs::ctx_view(ctx);
    if (n < 1 || !d_mask || !d_labels) return fs::ctx_fail(ctx, "fs_connected_components_noalloc: bad arguments");
    if ((size_t)n * v.H * v.W > fs::kNoallocMaxPixels || (size_t)n * fs::cc_blocks_per_image(v.H, v.W) > fs::kNoallocBlocks)
        return fs::ctx_fail(ctx, "fs_connected_components_noalloc: n * height * width exceeds the static work space of " +
                                     std::to_string(fs::kNoallocMaxPixels) + " pixels; use fs_connected_components");
    hipError_t e = hipSetDevice(v.device);
    if (e == hipSuccess)
        e = fs::noalloc_connected_components(v.device, d_mask, min_size_exclusive, d_labels, d_num_labels, n, v.H, v.W, v.stream);
    if (e != hipSuccess) return fs::ctx_fail(ctx, std::string("fs_connected_components_noalloc: ") + hipGetErrorString(e));
    return 0;
}

}  // extern "C"
