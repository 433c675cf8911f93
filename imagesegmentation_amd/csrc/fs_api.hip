// C ABI of libfs_hip.so (declared in include/fs_hip.h): context, device buffers
// and the orchestration of the kernels in k_stencils.hip, k_cc.hip, k_crf.hip.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <immintrin.h>
#include <functional>
#include <mutex>
#include <thread>

#include "../../include/fs_hip.h"
#include "fs_common.h"

using namespace fs;

#ifndef FS_STAGE_PAGEABLE
#define FS_STAGE_PAGEABLE 1
#endif

namespace {
thread_local std::string g_create_error;   // fs_create has no context to hold its message yet

enum { G_STENCIL = 0, G_CC = 1, G_SIM = 2, G_INIT = 3, G_ITER = 4, G_COUNT = 5 };

struct EventPair { hipEvent_t a, b; };
}  // namespace

struct fs_ctx {
    int device = 0, H = 0, W = 0, max_batch = 0;
    int r = 7;
    double view_x = 62.0 / 180 * M_PI, view_y = 48.6 / 180 * M_PI;   // config.py:6-7
    bool have_params = false;
    CrfParams params{};
    hipStream_t own_stream = nullptr, stream = nullptr;
    hipStream_t in_stream = nullptr, out_stream = nullptr;   // copy streams of the host-buffer entry point
    std::vector<hipEvent_t> pipe_events;
    std::string err;

    // tables
    DevBuf tanx, tany, ax, ay, pos_terms, wtab;
    bool use_planes = false;   // FS_CRF_PLANES=1: CRF update from precomputed similarity planes instead of the fused kernel
    // frame buffers, each sized for max_batch frames
    DevBuf depth16, rgb8, depthf, smoothed, points, normals, raw, angles, curv, scores, mask, edges, patches64, patches32,
        surfaces64;
    DevBuf cc_parent, cc_count, cc_newid, cc_blocksum;
    // CRF buffers (one frame in flight)
    DevBuf simH, qa, qb, pot, unary, f1, f2, f3, rows, descs, featA, featB;
    DevBuf msg;                        // message planes of a wide update (one frame at a time)
    std::string debug_fill;            // fs_set_debug_fill: work buffers to overwrite in front of every call (tests)
    HostBuf descs_host;                // page-locked staging of the frame descriptors; descs_copied: its last upload
    hipEvent_t descs_copied = nullptr;
    DevBuf crf_queue;                  // tile queues of the fused CRF update
    DevBuf tile_order;                 // sorted tile lists of the launches that want them (crf_tile_order_ints)
    bool queue_clean = false;          // all queue slots are zero and no launch has been given one since
    DevBuf curv_todo;                  // pixels whose curvature test needs the correctly rounded evaluation
    bool todo_clean = false;           // its per-frame counters are zero (the chain of kernels leaves them so)
    DevBuf tile_recs, counts;          // per-tile active labels [n][tiles][rec]; counts = nlabels[n] | max_active[n]
    HostBuf counts_host;               // page-locked: nlabels[0..n) then max_active[0..n)
    // One frame per call: the three updates are enqueued before the label counts reach the host (CrfFrameDesc::spec).
    // The counts then come through a page-locked word that the first update's first block writes and the host polls
    // (CrfFrameDesc::publish), not through a copy command that the updates would queue behind.
    bool speculate = true;                         // FS_NO_SPECULATION=1 switches it off (measurements, tests)
    hipEvent_t spec_copied = nullptr, counts_copied = nullptr;
    hipEvent_t stats_up_events[4] = {nullptr, nullptr, nullptr, nullptr};   // ... and for its four groups of inputs
    hipEvent_t stats_events[3] = {nullptr, nullptr, nullptr};   // per-frame statistics: the same for its three groups of outputs
    hipEvent_t early_events[2] = {nullptr, nullptr};   // host-buffer calls of one chunk: outputs leave as they are completed
    bool early_downloads = true;                   // FS_NO_EARLY_DOWNLOADS=1 switches that off (measurements)
    HostBuf publish_host;                          // the word
    // Host-buffer calls of several chunks send the label images as 16-bit patches and surfaces and 8-bit edges (narrow_dev,
    // copies of a chunk's int64 / int32 images) to page-locked staging (narrow_host) and widen them into the caller's
    // arrays on worker threads: 1.5 instead of 6.2 MB of a frame's 15.8 cross the link, which bounds that entry point.
    bool narrow_downloads = true;                  // FS_NO_NARROW_DOWNLOADS=1 switches that off (measurements)
    int widen_threads = 3;                         // FS_WIDEN_THREADS (two already keep up with the link)
    DevBuf narrow_dev;
    HostBuf narrow_host;
    std::vector<hipEvent_t> narrow_events;         // per chunk: its narrow images have arrived
    HostBuf frames_in_host;                        // frames of a small host-buffer call that arrive in pageable memory
    HostBuf stats_in_host;                         // ... and its inputs that arrive in pageable memory, on their way up
    HostBuf stats_small_host;                      // per-frame statistics: the small outputs on their way to the caller's arrays
    DevBuf publish_seq;                            // device word: one-frame calls so far (the kernels count)
    unsigned publish_expected = 0;                 // ... as the host counts them
    DevBuf spec_descs;
    // fs_find_surfaces_trace: while trace_cap != 0 the updates run as their diagnostic builds and leave the softmax
    // arguments and the probabilities of every update in trace_buf ([update][2][S][N] floats), S <= trace_cap
    DevBuf trace_buf;
    int trace_cap = 0, trace_S = 0;
    DevBuf apply_stats;                            // fs_apply_crf: per-frame label maxima and flags
    HostBuf apply_stats_host;
    HostBuf spec_descs_host;
    std::vector<CrfFrameDesc> spec_shadow;         // what spec_descs holds
    int spec_cap = 0;                              // label channels the Q planes held when the last such launch went out
    int64_t spec_held = 0, spec_repeated = 0;
    // per-surface statistics (one frame in flight)
    DevBuf st_in, st_work, st_scratch, st_out, st_pool, st_descs, st_lin;
    std::vector<StatsFrame> st_descs_host;

    // timing
    unsigned timing = 0;   // bit g set: record events around kernel group g
    std::vector<EventPair> ev[G_COUNT];
    size_t ev_used[G_COUNT] = {0, 0, 0, 0, 0};

    int fail(hipError_t e, const char *what)
    {
        err = std::string(what) + ": " + hipGetErrorString(e);
        return -1;
    }
    int fail(const std::string &m)
    {
        err = m;
        return -1;
    }
    size_t N() const { return (size_t)H * W; }
};

namespace fs {
CtxView ctx_view(const fs_ctx *ctx) { return CtxView{ctx->device, ctx->H, ctx->W, ctx->stream}; }
int ctx_fail(fs_ctx *ctx, const std::string &message) { return ctx->fail(message); }
}  // namespace fs

#define FS_HIP(expr)                                           \
    do {                                                       \
        hipError_t e_ = (expr);                                \
        if (e_ != hipSuccess) return ctx->fail(e_, #expr);     \
    } while (0)

namespace {

struct ScopedTimer {
    fs_ctx *c;
    int g;
    EventPair *ep = nullptr;
    ScopedTimer(fs_ctx *ctx, int group) : c(ctx), g(group)
    {
        if (!(c->timing >> g & 1u)) return;
        if (c->ev_used[g] == c->ev[g].size()) {
            EventPair p;
            if (hipEventCreate(&p.a) != hipSuccess || hipEventCreate(&p.b) != hipSuccess) return;
            c->ev[g].push_back(p);
        }
        ep = &c->ev[g][c->ev_used[g]++];
        (void)hipEventRecord(ep->a, c->stream);
    }
    ~ScopedTimer()
    {
        if (ep) (void)hipEventRecord(ep->b, c->stream);
    }
};

// page-locked host memory (hipHostMalloc, hipHostRegister) as opposed to ordinary pageable memory
bool host_memory_is_page_locked(const void *p)
{
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) {
        (void)hipGetLastError();            // an address the runtime does not know: pageable
        return false;
    }
    return a.type == hipMemoryTypeHost;
}

// Sources of a call's uploads: arrays in pageable memory are copied into the context's page-locked staging block first
// (the runtime's own path for pageable sources holds the calling thread several times longer than that memcpy).
// want() every source, then begin(), then from() in the same order.  The entry points that use it end with the stream
// drained, so the block is free again when the next call begins.
class StagedSources {
  public:
    explicit StagedSources(HostBuf &b) : buf_(b) {}
    void want(const void *src, size_t bytes)
    {
        if (bytes && !host_memory_is_page_locked(src)) total_ += (bytes + 255) / 256 * 256;
    }
    hipError_t begin() { return buf_.ensure(total_); }
    const void *from(const void *src, size_t bytes)
    {
        if (!bytes || host_memory_is_page_locked(src)) return src;
        char *dst = buf_.as<char>() + used_;
        std::memcpy(dst, src, bytes);
        used_ += (bytes + 255) / 256 * 256;
        return dst;
    }
    // room for `bytes` that the caller fills in itself (declare it with want_block() before begin())
    void want_block(size_t bytes) { total_ += (bytes + 255) / 256 * 256; }
    void *block(size_t bytes)
    {
        char *dst = buf_.as<char>() + used_;
        used_ += (bytes + 255) / 256 * 256;
        return dst;
    }

  private:
    HostBuf &buf_;
    size_t total_ = 0, used_ = 0;
};

int set_device(fs_ctx *ctx)
{
    FS_HIP(hipSetDevice(ctx->device));
    return 0;
}

// fs_set_debug_fill(<letters>) (initially the value of FS_DEBUG_NAN): float work buffers named by the letters are filled with NaN in front of every find_surfaces
// call (q: both sets of probability planes, a / b: the two halves of the feature records, s: smoothed depth, r: raw
// angles; t: the tile records of the CRF update, as -1; p c n k u o m v e l: one integer work buffer each, as 1; w, read by the statistics stage: its work
// arrays, as 1), so that
// a kernel that reads a value nobody wrote in this call shows up in the results.
int debug_nan_fill(fs_ctx *ctx)
{
    if (ctx->debug_fill.empty()) return 0;
    const char *which = ctx->debug_fill.c_str();
    auto fill = [&](DevBuf &b) -> hipError_t {
        return b.p ? hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(b.p), 0x7fc00000, b.cap / 4, ctx->stream) : hipSuccess;
    };
    for (const char *c = which; *c; ++c) {
        if (*c == 'q') { FS_HIP(fill(ctx->qa)); FS_HIP(fill(ctx->qb)); }
        if (*c == 'a') FS_HIP(fill(ctx->featA));
        if (*c == 'b') FS_HIP(fill(ctx->featB));
        if (*c == 's') FS_HIP(fill(ctx->smoothed));
        if (*c == 'r') FS_HIP(fill(ctx->raw));
        // integer work buffers, as the in-range value 1 (the components' parents, sizes, ranks and block sums, the
        // undecided-pixel lists, the label counts, mask, curvature decisions, edges and the int32 patch labels)
        DevBuf *ints[] = {&ctx->cc_parent, &ctx->cc_count, &ctx->cc_newid, &ctx->cc_blocksum, &ctx->curv_todo, &ctx->counts,
                          &ctx->mask, &ctx->curv, &ctx->edges, &ctx->patches32};
        const char *names = "pcnkuomvel";
        for (int i = 0; names[i]; ++i)
            if (*c == names[i] && ints[i]->p) {
                FS_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(ints[i]->p), 1, ints[i]->cap / 4, ctx->stream));
                if (names[i] == 'u') ctx->todo_clean = false;       // (the lists' counters are cleared again by the call)
            }
        if (*c == 't' && ctx->tile_recs.p) FS_HIP(hipMemsetAsync(ctx->tile_recs.p, 0xff, ctx->tile_recs.cap, ctx->stream));   // tile records: -1
    }
    return 0;
}

// get_angle_arrays (image_processing_models_GPU.py:137-149) and the lateral pixel
// size factors |tan[i-1] - tan[i+1]| of find_surfaces.py:237-238, on the host in
// fp32 so that the tables are the same numbers a CPU evaluation would use.
int build_tables(fs_ctx *ctx)
{
    const int H = ctx->H, W = ctx->W;
    std::vector<float> tx(W), ty(H), ax(W, 0.0f), ay(H, 0.0f);
    const float x_val = (float)(ctx->view_x / 2.0), y_val = (float)(ctx->view_y / 2.0);
    const float dx = (float)(W / 2.0 - 0.5), dy = (float)(H / 2.0 - 0.5);
    for (int c = 0; c < W; ++c) {
        float v = ((float)c / dx) - 1.0f;
        tx[c] = tanf(x_val * v);
    }
    for (int r = 0; r < H; ++r) {
        float v = ((float)(H - 1 - r) / dy) - 1.0f;
        ty[r] = tanf(y_val * v);
    }
    for (int c = 1; c < W - 1; ++c) ax[c] = fabsf(tx[c - 1] - tx[c + 1]);
    for (int r = 1; r < H - 1; ++r) ay[r] = fabsf(ty[r - 1] - ty[r + 1]);
    FS_HIP(ctx->tanx.ensure(W * sizeof(float)));
    FS_HIP(ctx->tany.ensure(H * sizeof(float)));
    FS_HIP(ctx->ax.ensure(W * sizeof(float)));
    FS_HIP(ctx->ay.ensure(H * sizeof(float)));
    FS_HIP(hipMemcpy(ctx->tanx.p, tx.data(), W * sizeof(float), hipMemcpyHostToDevice));
    FS_HIP(hipMemcpy(ctx->tany.p, ty.data(), H * sizeof(float), hipMemcpyHostToDevice));
    FS_HIP(hipMemcpy(ctx->ax.p, ax.data(), W * sizeof(float), hipMemcpyHostToDevice));
    FS_HIP(hipMemcpy(ctx->ay.p, ay.data(), H * sizeof(float), hipMemcpyHostToDevice));
    return 0;
}

// Position terms of the three Gaussian kernels for every forward window offset:
// theta^2 * ((dy/40)^2) + theta^2 * ((dx/40)^2), evaluated in the graph's order
// (find_surfaces.py:343-350 with grid features, exponent mask 0 => divisor 1).
int build_pos_terms(fs_ctx *ctx)
{
    const int r = ctx->r;
    const size_t nh = crf_half_offsets(r);
    std::vector<float> pt(3 * nh);
    const float np_ = (float)(1.0 / 40.0);
    for (int dy = 0; dy <= r; ++dy)
        for (int dx = (dy == 0 ? 1 : -r); dx <= r; ++dx) {
            const int oi = dy == 0 ? dx - 1 : r + (dy - 1) * (2 * r + 1) + (dx + r);
            const float d0 = ((float)dy / 1.0f) * np_, d1 = ((float)dx / 1.0f) * np_;
            pt[3 * oi + 0] = (d0 * d0) * ctx->params.t11 + (d1 * d1) * ctx->params.t11;
            pt[3 * oi + 1] = (d0 * d0) * ctx->params.t21 + (d1 * d1) * ctx->params.t21;
            pt[3 * oi + 2] = (d0 * d0) * ctx->params.t31 + (d1 * d1) * ctx->params.t31;
        }
    FS_HIP(ctx->pos_terms.ensure(pt.size() * sizeof(float)));
    FS_HIP(hipMemcpy(ctx->pos_terms.p, pt.data(), pt.size() * sizeof(float), hipMemcpyHostToDevice));
    // per-offset constants of the fused update over the full window: (w1', w2', w3', bias) with
    // w_k' = w_k * exp(-position term of kernel k); zeros at the centre (excluded from the message,
    // find_surfaces.py:300-302)
    const int kw = 2 * r + 1;
    std::vector<float> wt((size_t)4 * kw * kw, 0.0f);
    for (int dy = -r; dy <= r; ++dy)
        for (int dx = -r; dx <= r; ++dx) {
            if (dy == 0 && dx == 0) continue;
            const double q = ((double)dy / 40.0) * ((double)dy / 40.0) + ((double)dx / 40.0) * ((double)dx / 40.0);
            float *w = &wt[4 * ((size_t)(dy + r) * kw + dx + r)];
            if (ctx->params.mode == 1) {
                // linear similarity: weights of |relative depth difference| * 6, angle distance * 2, colour distance / 50;
                // the position distance |offset| / 40 is a constant of the offset and joins the bias
                w[0] = 6.0f * ctx->params.lr_coef[0];
                w[1] = 2.0f * ctx->params.lr_coef[1];
                w[2] = (float)(ctx->params.lr_coef[2] / 50.0);
                w[3] = (float)((double)ctx->params.bias + (double)ctx->params.lr_coef[3] * (std::sqrt((double)(dy * dy + dx * dx)) / 40.0));
                continue;
            }
            w[0] = (float)(ctx->params.w1 * std::exp(-(double)ctx->params.t11 * q));
            w[1] = (float)(ctx->params.w2 * std::exp(-(double)ctx->params.t21 * q));
            w[2] = (float)(ctx->params.w3 * std::exp(-(double)ctx->params.t31 * q));
            w[3] = ctx->params.bias;
        }
    FS_HIP(ctx->wtab.ensure(wt.size() * sizeof(float)));
    FS_HIP(hipMemcpy(ctx->wtab.p, wt.data(), wt.size() * sizeof(float), hipMemcpyHostToDevice));
    return 0;
}

int check_batch(fs_ctx *ctx, int n)
{
    if (!ctx) return -1;
    if (n < 1 || n > ctx->max_batch) return ctx->fail("batch size " + std::to_string(n) + " outside [1, max_batch=" + std::to_string(ctx->max_batch) + "]");
    return set_device(ctx);
}

// list capacity per frame: one pixel in 16 (about one in a thousand is undecided on sensor data)
int curv_todo_cap(const fs_ctx *ctx) { return (int)(ctx->N() / 16); }

int ensure_frame_buffers(fs_ctx *ctx)
{
    const size_t n = ctx->max_batch, N = ctx->N();
    FS_HIP(ctx->smoothed.ensure(n * N * 4));
    FS_HIP(ctx->points.ensure(n * N * 12));
    FS_HIP(ctx->normals.ensure(n * N * 12));
    FS_HIP(ctx->raw.ensure(n * N * 8));
    FS_HIP(ctx->angles.ensure(n * N * 8));
    FS_HIP(ctx->curv.ensure(n * N));
    FS_HIP(ctx->mask.ensure(n * N));
    FS_HIP(ctx->edges.ensure(n * N * 4));
    FS_HIP(ctx->patches64.ensure(n * N * 8));
    FS_HIP(ctx->patches32.ensure(n * N * 4));
    FS_HIP(ctx->surfaces64.ensure(n * N * 8));
    FS_HIP(ctx->cc_parent.ensure(n * N * 4));
    FS_HIP(ctx->cc_count.ensure(n * N * 4));
    FS_HIP(ctx->cc_newid.ensure(n * N * 4));
    FS_HIP(ctx->cc_blocksum.ensure(n * cc_blocks_per_image(ctx->H, ctx->W) * 4));
    FS_HIP(ctx->counts.ensure(2 * n * 4));
    FS_HIP(ctx->crf_queue.ensure(kCrfQueueSlots * kCrfQueueBytes));
    FS_HIP(ctx->curv_todo.ensure(curvature_todo_ints((int)n, curv_todo_cap(ctx)) * sizeof(int32_t)));
    ctx->todo_clean = false;           // new memory, or another layout (the lists follow max_batch counters)
    ctx->queue_clean = false;
    FS_HIP(ctx->counts_host.ensure(2 * n * sizeof(int32_t)));
    if (!ctx->publish_seq.p) {
        // (coherent: the GPU's store must become visible while the stream is still at work, whatever HIP_HOST_COHERENT says)
        FS_HIP(ctx->publish_host.ensure(sizeof(unsigned long long), hipHostMallocCoherent));
        *ctx->publish_host.as<unsigned long long>() = 0;
        FS_HIP(ctx->publish_seq.ensure(sizeof(unsigned)));
        FS_HIP(hipMemset(ctx->publish_seq.p, 0, sizeof(unsigned)));
    }
    if (crf_fused_tile_height(ctx->r)) {
        int tx, ty;
        crf_fused_tile_grid(ctx->H, ctx->W, ctx->r, &tx, &ty);
        FS_HIP(ctx->tile_recs.ensure(n * tx * ty * crf_tile_rec_ints() * 4));
        FS_HIP(ctx->featA.ensure(n * N * 16));
        FS_HIP(ctx->featB.ensure(n * N * 8));
    }
    return 0;
}

CcWork cc_work(fs_ctx *ctx)
{
    CcWork w;
    w.parent = ctx->cc_parent.as<int32_t>();
    w.count = ctx->cc_count.as<int32_t>();
    w.newid = ctx->cc_newid.as<int32_t>();
    w.blocksum = ctx->cc_blocksum.as<int32_t>();
    w.nlabels = ctx->counts.as<int32_t>();
    return w;
}

// smoothed depth -> patches / edges (device), label counts to the host
int run_patches(fs_ctx *ctx, int n, const float *d_smoothed, int64_t *d_patches64, int32_t *d_patches32, int32_t *d_edges,
                float *d_scores, uint8_t *d_curv, int read_counts)      // 0: no, 1: copied and waited for, 2: left to the first update (wait_published_counts), 3: copied, counts_copied marks the arrival
{
    const int H = ctx->H, W = ctx->W;
    const float threshold = (float)(0.007003343675404672 * 10.5);   // find_surfaces.py:229
    {
        ScopedTimer t(ctx, G_STENCIL);
        // the list counters of the curvature pass are zeroed by the majority pass that follows it, so that a call
        // needs no memset of its own once the first one has run
        const bool used = launch_curvature_mask(d_smoothed, ctx->ax.as<float>(), ctx->ay.as<float>(), threshold, d_curv, d_scores,
                                                ctx->curv_todo.as<int32_t>(), curv_todo_cap(ctx), ctx->max_batch, ctx->todo_clean, n, H, W,
                                                ctx->stream);
        if (used) ctx->todo_clean = false;
        launch_majority_edges(d_curv, d_smoothed, ctx->ax.as<float>(), ctx->ay.as<float>(), 3.0f, ctx->mask.as<uint8_t>(), d_edges,
                              used ? ctx->curv_todo.as<int32_t>() : nullptr, n, H, W, ctx->stream);
        if (used) ctx->todo_clean = true;
    }
    int32_t *maxa = ctx->counts.as<int32_t>() + n;
    {
        ScopedTimer t(ctx, G_CC);
        CcWork w = cc_work(ctx);
        if (read_counts) w.clear = maxa;        // the per-frame maxima the tile scan accumulates into start at zero
        launch_connected_components(ctx->mask.as<uint8_t>(), w, 20, d_patches64, d_patches32, n, H, W, ctx->stream);
    }
    FS_HIP(hipGetLastError());
    if (read_counts) {
        // per-tile active patch labels for the CRF (needs only the patches), then one readback of
        // label counts and per-frame maxima
        if (const int th = crf_fused_tile_height(ctx->r)) {
            ScopedTimer t(ctx, G_INIT);
            // the tile queues of the updates are cleared on the way: one memset less in front of the CRF
            launch_crf_tile_active(d_patches32, n, H, W, 32, th, ctx->r * 3, ctx->tile_recs.as<int32_t>(), maxa, ctx->stream,
                                   ctx->crf_queue.as<unsigned>(), (int)(kCrfQueueSlots * kCrfQueueBytes / sizeof(unsigned)));
            ctx->queue_clean = true;
        }
        if (read_counts == 1 || read_counts == 3) {
            FS_HIP(hipMemcpyAsync(ctx->counts_host.p, ctx->counts.p, 2 * n * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
            if (read_counts == 1) {
                FS_HIP(hipStreamSynchronize(ctx->stream));
            } else {
                if (!ctx->counts_copied) FS_HIP(hipEventCreateWithFlags(&ctx->counts_copied, hipEventDisableTiming));
                FS_HIP(hipEventRecord(ctx->counts_copied, ctx->stream));
            }
        }
    }
    return 0;
}

// the label counts of a one-frame call whose updates went out ahead of them (launch_crf_speculative) -> ctx->counts_host
int wait_published_counts(fs_ctx *ctx)
{
    const volatile unsigned long long *pub = ctx->publish_host.as<unsigned long long>();
    const unsigned long long want = ctx->publish_expected & 0xffffu;
    unsigned long long word;
    for (unsigned spins = 0; ((word = *pub) >> 48) != want; ++spins) {
#if defined(__x86_64__)
        __builtin_ia32_pause();                       // (a polling core leaves its sibling thread the execution units)
#endif
        if ((spins & 0xffffu) != 0xffffu) continue;
        // now and then: is the stream still at work?  Once it has drained, the word is either there or will never come.
        const hipError_t e = hipStreamQuery(ctx->stream);
        if (e == hipErrorNotReady) continue;
        if (e != hipSuccess) return ctx->fail(e, "waiting for the label counts");
        if (((word = *pub) >> 48) != want) return ctx->fail("the label counts did not arrive although the stream has drained");
        break;
    }
    int32_t *dst = ctx->counts_host.as<int32_t>();
    dst[0] = (int32_t)(word & 0xffffffffu);
    dst[1] = (int32_t)(word >> 32 & 0xffffu);
    return 0;
}

// Similarity planes for `frames` frames of Npix pixels: [slack | planes*Npix | slack | planes*Npix | slack ...],
// slack zeroed once and never written.  Returns plane 0 of frame 0 and the frame stride (floats).
int ensure_sim(fs_ctx *ctx, size_t Npix, int Wimg, int frames, float **plane0, size_t *stride)
{
    const size_t slack = crf_sim_slack(Wimg, ctx->r);
    *stride = crf_half_offsets(ctx->r) * Npix + slack;
    const size_t floats = slack + (size_t)frames * *stride;
    if (floats * sizeof(float) > ctx->simH.cap) {
        FS_HIP(ctx->simH.ensure(floats * sizeof(float)));
        FS_HIP(hipMemsetAsync(ctx->simH.p, 0, ctx->simH.cap, ctx->stream));
    }
    *plane0 = ctx->simH.as<float>() + slack;
    return 0;
}

constexpr int kSpecLabels = 64;                // Q planes a one-frame call allocates before it knows the frame's label count
constexpr int kSpecClass = 2;                  // kernel variant of the updates launched ahead of the counts (16 channels per tile)
constexpr int kCrfGroup = 128;                 // most frames per batched CRF launch (few launches: each has a ramp and a tail)
constexpr int kCrfGroupPlanes = 37;            // ... when the similarity planes of the group are kept as well
constexpr size_t kCrfGroupBytes = (size_t)24 << 30;  // ... and most bytes of one Q buffer of a group (there are two)
constexpr int kPipeChunk = 16;  // frames per upload / compute / download step of the host-buffer entry point

// ---- narrow copies of the label images for the host-buffer entry point ---------------------------------------------
// patch labels and final labels lie in [0, kMaxLabels), depth edges are 0 or 1: 16 and 8 bits hold them
__global__ __launch_bounds__(256) void narrow_patches_edges_kernel(const int32_t *__restrict__ patches, const int32_t *__restrict__ edges,
                                                                   int16_t *__restrict__ p16, uint8_t *__restrict__ e8, size_t n4)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    if (p16) {
        const int4 v = reinterpret_cast<const int4 *>(patches)[i];
        reinterpret_cast<short4 *>(p16)[i] = make_short4((short)v.x, (short)v.y, (short)v.z, (short)v.w);
    }
    if (e8) {
        const int4 v = reinterpret_cast<const int4 *>(edges)[i];
        reinterpret_cast<uchar4 *>(e8)[i] = make_uchar4((unsigned char)v.x, (unsigned char)v.y, (unsigned char)v.z, (unsigned char)v.w);
    }
}
__global__ __launch_bounds__(256) void narrow_surfaces_kernel(const int64_t *__restrict__ surfaces, int16_t *__restrict__ s16, size_t n4)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const longlong2 a = reinterpret_cast<const longlong2 *>(surfaces)[2 * i], b = reinterpret_cast<const longlong2 *>(surfaces)[2 * i + 1];
    reinterpret_cast<short4 *>(s16)[i] = make_short4((short)a.x, (short)a.y, (short)b.x, (short)b.y);
}
// host side: back to the caller's int64 / int32 arrays.  The wide arrays are written once and not read here, so on a CPU
// with AVX2 the stores bypass the cache (a plain store first reads the line it is about to overwrite: twice the memory
// traffic, and with one process per GPU on a node the host's memory is what all of them share).
__attribute__((target("avx2"))) void widen_i16_to_i64_avx2(const int16_t *src, int64_t *dst, size_t n)
{
    size_t i = 0;
    for (; i < n && (reinterpret_cast<uintptr_t>(dst + i) & 31u); ++i) dst[i] = src[i];
    for (; i + 16 <= n; i += 16) {
        const __m128i a = _mm_loadu_si128(reinterpret_cast<const __m128i *>(src + i));
        const __m128i b = _mm_loadu_si128(reinterpret_cast<const __m128i *>(src + i + 8));
        _mm256_stream_si256(reinterpret_cast<__m256i *>(dst + i), _mm256_cvtepi16_epi64(a));
        _mm256_stream_si256(reinterpret_cast<__m256i *>(dst + i + 4), _mm256_cvtepi16_epi64(_mm_srli_si128(a, 8)));
        _mm256_stream_si256(reinterpret_cast<__m256i *>(dst + i + 8), _mm256_cvtepi16_epi64(b));
        _mm256_stream_si256(reinterpret_cast<__m256i *>(dst + i + 12), _mm256_cvtepi16_epi64(_mm_srli_si128(b, 8)));
    }
    for (; i < n; ++i) dst[i] = src[i];
    _mm_sfence();
}
__attribute__((target("avx2"))) void widen_u8_to_i32_avx2(const uint8_t *src, int32_t *dst, size_t n)
{
    size_t i = 0;
    for (; i < n && (reinterpret_cast<uintptr_t>(dst + i) & 31u); ++i) dst[i] = src[i];
    for (; i + 16 <= n; i += 16) {
        const __m128i a = _mm_loadu_si128(reinterpret_cast<const __m128i *>(src + i));
        _mm256_stream_si256(reinterpret_cast<__m256i *>(dst + i), _mm256_cvtepu8_epi32(a));
        _mm256_stream_si256(reinterpret_cast<__m256i *>(dst + i + 8), _mm256_cvtepu8_epi32(_mm_srli_si128(a, 8)));
    }
    for (; i < n; ++i) dst[i] = src[i];
    _mm_sfence();
}
const bool g_cpu_has_avx2 = __builtin_cpu_supports("avx2");
void widen_i16_to_i64(const int16_t *src, int64_t *dst, size_t n)
{
    if (g_cpu_has_avx2) return widen_i16_to_i64_avx2(src, dst, n);
    for (size_t i = 0; i < n; ++i) dst[i] = src[i];
}
void widen_u8_to_i32(const uint8_t *src, int32_t *dst, size_t n)
{
    if (g_cpu_has_avx2) return widen_u8_to_i32_avx2(src, dst, n);
    for (size_t i = 0; i < n; ++i) dst[i] = src[i];
}
// The worker threads of one host-buffer call: chunk k's narrow images may be widened once the main thread has recorded the
// event behind their copies (`posted` counts the chunks for which it has) and that event has completed.  Every chunk is cut
// into `parts` jobs so that all workers share a chunk; a job is claimed through `next`.
struct WidenJobs {
    std::mutex m;
    std::condition_variable cv;
    int posted = 0;
    bool abort = false;
    std::atomic<int> next{0};
    std::atomic<int> failed{0};
    std::vector<std::thread> workers;
    // waits for the workers; `ok` false: they give up whatever has not been posted
    void finish(bool ok)
    {
        if (workers.empty()) return;
        {
            std::lock_guard<std::mutex> lk(m);
            if (!ok) abort = true;
        }
        cv.notify_all();
        for (std::thread &t : workers) t.join();
        workers.clear();
    }
    ~WidenJobs() { finish(false); }            // (a call that leaves early)
};

// CRF of n frames resident on the device: similarity planes, initial probabilities from
// `labels32`, `iterations` mean-field updates, argmax.  S[i] label channels and max_active[i]
// listed labels per tile (host arrays).  Frames are processed in groups; inside a group all
// frames of one kernel variant (channel class) share a launch.
int run_crf_batch(fs_ctx *ctx, int n, const uint16_t *d_depth, const uint8_t *d_rgb, const float *d_angles,
                  const int32_t *d_labels32, const int *S, const int *max_active, int iterations, int64_t *d_surfaces,
                  bool features_packed = false)
{
    const int H = ctx->H, W = ctx->W;
    const size_t N = ctx->N();
    const bool fused = !ctx->use_planes && crf_fused_tile_height(ctx->r) != 0;
    if (!fused && ctx->params.mode == 1) return ctx->fail("the logistic-regression similarity is only available in the fused update");
    int tiles_x = 0, tiles_y = 0;
    if (fused) crf_fused_tile_grid(H, W, ctx->r, &tiles_x, &tiles_y);
    const size_t rec_stride = (size_t)tiles_x * tiles_y * crf_tile_rec_ints();
    for (int i = 0; i < n; ++i)
        if (S[i] > kMaxLabels) return ctx->fail("frame has " + std::to_string(S[i]) + " label channels; at most " + std::to_string(kMaxLabels) + " are supported");
    // the update kernels address a frame's probability planes with 32-bit element offsets
    for (int i = 0; i < n; ++i)
        if ((uint64_t)S[i] * N >= ((uint64_t)1 << 32))
            return ctx->fail("a frame of " + std::to_string(N) + " pixels with " + std::to_string(S[i]) + " label channels exceeds 2^32 probabilities");
    // groups: consecutive frames, limited in number and in the size of their Q planes
    std::vector<int> gstart{0};
    size_t qmax = 0;
    int gmax = 0;
    {
        // (two passes: the first counts the groups, the second deals the frames out evenly, so that the last group is not a
        // small one whose launches leave most of the chip idle)
        int cap = fused ? kCrfGroup : kCrfGroupPlanes;
        for (int pass = 0; pass < 2; ++pass) {
            gstart.assign(1, 0);
            qmax = 0;
            gmax = 0;
            size_t q = 0;
            for (int i = 0; i < n; ++i) {
                const size_t qi = (size_t)S[i] * N;
                if (i > gstart.back() && (i - gstart.back() == cap || (q + qi) * sizeof(float) > kCrfGroupBytes)) {
                    gstart.push_back(i);
                    q = 0;
                }
                q += qi;
                if (q > qmax) qmax = q;
                if (i + 1 - gstart.back() > gmax) gmax = i + 1 - gstart.back();
            }
            const int groups = (int)gstart.size();
            gstart.push_back(n);
            cap = std::min(cap, (n + groups - 1) / groups);
        }
    }
    const int ngroups = (int)gstart.size() - 1;
    float *sim0 = nullptr;
    size_t sim_stride = 0;
    if (!fused && ensure_sim(ctx, N, W, gmax, &sim0, &sim_stride)) return -1;
    FS_HIP(ctx->qa.ensure(qmax * sizeof(float)));
    FS_HIP(ctx->qb.ensure(qmax * sizeof(float)));
    if (fused) FS_HIP(ctx->tile_order.ensure(crf_tile_order_ints((size_t)tiles_x * tiles_y * gmax) * sizeof(int32_t)));
    const bool trace = ctx->trace_cap != 0;
    if (trace) {
        if (n != 1 || !fused) return ctx->fail("the diagnostic updates take one frame and the fused kernels");
        ctx->trace_S = S[0];
        if (S[0] > ctx->trace_cap)
            return ctx->fail("the frame has " + std::to_string(S[0]) + " label channels, the caller's arrays hold " + std::to_string(ctx->trace_cap));
        const size_t bytes = (size_t)iterations * 2 * S[0] * N * sizeof(float);
        FS_HIP(ctx->trace_buf.ensure(bytes));
        FS_HIP(hipMemsetAsync(ctx->trace_buf.p, 0, bytes, ctx->stream));
    }

    // descriptors: [group][iteration][frames of the group ordered by channel class]
    // page-locked, so that the upload is a plain DMA transfer; the previous upload must have left it
    const size_t ndesc = (size_t)n * iterations;
    if (ctx->descs_copied) FS_HIP(hipEventSynchronize(ctx->descs_copied));
    else FS_HIP(hipEventCreateWithFlags(&ctx->descs_copied, hipEventDisableTiming));
    FS_HIP(ctx->descs_host.ensure(ndesc * sizeof(CrfFrameDesc)));
    CrfFrameDesc *dh = ctx->descs_host.as<CrfFrameDesc>();
    std::fill(dh, dh + ndesc, CrfFrameDesc{});
    struct Launch { size_t first; int count, cls, wide_channels, max_listed; };
    size_t msg_floats = 0;                       // message planes of the largest wide frame
    std::vector<std::vector<Launch>> launches((size_t)ngroups * iterations);
    for (int gi = 0; gi < ngroups; ++gi) {
        const int g0 = gstart[gi], gn = gstart[gi + 1] - g0;
        std::vector<int> order(gn), cls(gn);
        std::vector<size_t> qoff(gn);
        size_t q = 0;
        for (int k = 0; k < gn; ++k) {
            order[k] = k;
            int ce = fused ? max_active[g0 + k] + 3 : S[g0 + k];   // without per-tile lists every label is carried
            if (ce > S[g0 + k]) ce = S[g0 + k];
            if (ce > kMaxTileChannels && !fused)
                return ctx->fail("a tile of frame " + std::to_string(g0 + k) + " needs " + std::to_string(ce) + " label channels; at most " +
                                 std::to_string(kMaxTileChannels) + " are supported by the plane-based update");
            cls[k] = crf_channel_class(ce);
            if (cls[k] == kCrfWideClass) {
                if (trace) return ctx->fail("the diagnostic updates carry at most " + std::to_string(kMaxTileChannels) + " channels per tile");
                msg_floats = std::max(msg_floats, (size_t)S[g0 + k] * N);
            }
            qoff[k] = q;
            q += (size_t)S[g0 + k] * N;
        }
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return cls[a] < cls[b]; });
        for (int it = 0; it < iterations; ++it) {
            const bool last = it == iterations - 1;
            float *qin = (it & 1) ? ctx->qb.as<float>() : ctx->qa.as<float>();
            float *qout = (it & 1) ? ctx->qa.as<float>() : ctx->qb.as<float>();
            const size_t base = (size_t)g0 * iterations + (size_t)it * gn;
            for (int k = 0; k < gn; ++k) {
                const int f = order[k], fi = g0 + f;
                CrfFrameDesc &d = dh[base + k];
                d.sim0 = fused ? nullptr : sim0 + (size_t)f * sim_stride;
                d.featA = fused ? ctx->featA.as<float4>() + (size_t)fi * N : nullptr;
                d.featB = fused ? ctx->featB.as<float2>() + (size_t)fi * N : nullptr;
                d.Qin = qin + qoff[f];
                d.Qout = last ? nullptr : qout + qoff[f];
                d.potentials = trace ? ctx->trace_buf.as<float>() + (size_t)(2 * it) * S[0] * N : nullptr;
                d.trace_q = trace ? ctx->trace_buf.as<float>() + (size_t)(2 * it + 1) * S[0] * N : nullptr;
                d.unary = nullptr;
                d.patches = d_labels32 + (size_t)fi * N;
                d.depth_raw = d_depth + (size_t)fi * N;
                d.surfaces = last ? d_surfaces + (size_t)fi * N : nullptr;
                d.tile_recs = fused ? ctx->tile_recs.as<int32_t>() + (size_t)fi * rec_stride : nullptr;
                d.S = S[fi];
                // frames of one class share a launch; a wide frame (more channels around a tile than the kernels keep in
                // registers) goes alone, because all of them use the context's one set of message planes in turn
                const bool wide = cls[f] == kCrfWideClass;
                if (k == 0 || cls[order[k]] != cls[order[k - 1]] || wide)
                    launches[(size_t)gi * iterations + it].push_back(Launch{base + k, 1, cls[f], wide ? std::min(S[fi], max_active[fi] + 3) : 0, 0});
                else
                    launches[(size_t)gi * iterations + it].back().count++;
                Launch &lk = launches[(size_t)gi * iterations + it].back();
                lk.max_listed = std::max(lk.max_listed, fused ? max_active[fi] : 0);
            }
        }
    }
    if (msg_floats) {
        FS_HIP(ctx->msg.ensure(msg_floats * sizeof(float)));
        for (size_t i = 0; i < ndesc; ++i) dh[i].msg = ctx->msg.as<float>();       // (only the wide launches read it)
    }
    FS_HIP(ctx->descs.ensure(ndesc * sizeof(CrfFrameDesc)));
    FS_HIP(hipMemcpyAsync(ctx->descs.p, dh, ndesc * sizeof(CrfFrameDesc), hipMemcpyHostToDevice, ctx->stream));
    FS_HIP(hipEventRecord(ctx->descs_copied, ctx->stream));
    const CrfFrameDesc *dd = ctx->descs.as<CrfFrameDesc>();
    const CrfGeom geom{H, W, 0, 0, H, W, ctx->r};
    if (fused && !features_packed) {
        ScopedTimer t(ctx, G_SIM);
        launch_crf_pack_features(d_depth, d_rgb, d_angles, ctx->featA.p, ctx->featB.p, (size_t)n * N, ctx->params, ctx->stream);
    }
    // tile queues: one zeroed slot per launch, all cleared by one memset (which the caller may have issued already)
    int queue_slot = ctx->queue_clean ? 0 : kCrfQueueSlots;
    ctx->queue_clean = false;
    for (int gi = 0; gi < ngroups; ++gi) {
        const int g0 = gstart[gi], gn = gstart[gi + 1] - g0;
        if (!fused) {
            ScopedTimer t(ctx, G_SIM);
            launch_crf_sim(d_depth + (size_t)g0 * N, d_rgb + (size_t)g0 * N * 3, d_angles + (size_t)g0 * N * 2, ctx->params,
                           ctx->pos_terms.as<float>(), sim0, sim_stride, gn, H, W, ctx->r, ctx->stream);
        }
        if (!fused) {
            ScopedTimer t(ctx, G_INIT);
            launch_crf_init(dd + (size_t)g0 * iterations, gn, H, W, ctx->stream);
        }
        for (int it = 0; it < iterations; ++it) {
            ScopedTimer t(ctx, G_ITER);
            for (const Launch &l : launches[(size_t)gi * iterations + it]) {
                const int mode = (it == iterations - 1 ? CRF_LAST : 0) | (it == 0 ? CRF_FIRST : 0);
                if (!fused) {
                    const hipError_t e = launch_crf_iter(dd + l.first, l.count, l.cls, geom, ctx->params.weight, ctx->params.evidence, mode, ctx->stream);
                    if (e != hipSuccess) return ctx->fail(e, "crf iteration launch");
                    continue;
                }
                // one launch, or one per group of tiles that share a kernel variant (crf_update_plan), each with its own queue
                CrfSplit plan[kCrfMaxSplits];
                const int nplan = crf_update_plan(ctx->params, ctx->r, l.cls, l.max_listed, trace, plan);
                for (int s = 0; s < nplan; ++s) {
                    if (queue_slot == kCrfQueueSlots) {
                        FS_HIP(hipMemsetAsync(ctx->crf_queue.p, 0, kCrfQueueSlots * kCrfQueueBytes, ctx->stream));
                        queue_slot = 0;
                    }
                    unsigned *queue = ctx->crf_queue.as<unsigned>() + queue_slot++ * (kCrfQueueBytes / sizeof(unsigned));
                    int32_t *order = nullptr;
                    if (plan[s].channel_class & kCrfMultiClass) order = ctx->tile_order.as<int32_t>();
                    const hipError_t e = launch_crf_iter_fused(dd + l.first, l.count, plan[s].channel_class, geom, ctx->params, ctx->wtab.as<float>(), mode, queue,
                                                               ctx->stream, trace, l.wide_channels, plan[s].a_min, plan[s].a_max, order);
                    if (e != hipSuccess) return ctx->fail(e, "crf iteration launch");
                }
            }
        }
    }
    return 0;
}



// does a frame with S label channels and max_active listed labels in its fullest tile fit a launch made ahead of the counts?
bool spec_fits(int S, int max_active, int cap) { return S <= cap && crf_channel_class(std::min(S, max_active + 3)) == kSpecClass; }

// One frame, fused update: `iterations` mean-field updates enqueued behind the kernels that produce the patch labels,
// before the label count has reached the host.  The updates read the count from ctx->counts (CrfFrameDesc::spec) and
// leave the frame alone when it does not fit (spec_fits, evaluated by the kernels and, afterwards, by the caller).
int launch_crf_speculative(fs_ctx *ctx, const uint16_t *d_depth, const int32_t *d_labels32, int iterations, int64_t *d_surfaces)
{
    const int H = ctx->H, W = ctx->W;
    const size_t N = ctx->N();
    FS_HIP(ctx->qa.ensure((size_t)kSpecLabels * N * sizeof(float)));
    FS_HIP(ctx->qb.ensure((size_t)kSpecLabels * N * sizeof(float)));
    ctx->spec_cap = (int)std::min<size_t>(std::min(ctx->qa.cap, ctx->qb.cap) / (N * sizeof(float)), (size_t)kMaxLabels);
    ctx->spec_cap = (int)std::min<uint64_t>((uint64_t)ctx->spec_cap, (((uint64_t)1 << 32) - 1) / N);     // 32-bit plane offsets in the kernels
    std::vector<CrfFrameDesc> want((size_t)iterations);
    for (int it = 0; it < iterations; ++it) {
        const bool last = it == iterations - 1;
        CrfFrameDesc &d = want[it];
        d = CrfFrameDesc{};
        d.featA = ctx->featA.as<float4>();
        d.featB = ctx->featB.as<float2>();
        d.Qin = (it & 1) ? ctx->qb.as<float>() : ctx->qa.as<float>();
        d.Qout = last ? nullptr : ((it & 1) ? ctx->qa.as<float>() : ctx->qb.as<float>());
        d.patches = d_labels32;
        d.depth_raw = d_depth;
        d.surfaces = last ? d_surfaces : nullptr;
        d.tile_recs = ctx->tile_recs.as<int32_t>();
        d.spec = ctx->counts.as<int32_t>();          // n = 1: [0] largest patch label, [1] most listed labels of a tile
        d.spec_cap = ctx->spec_cap;
        d.publish = it == 0 ? ctx->publish_host.as<unsigned long long>() : nullptr;
        d.publish_seq = ctx->publish_seq.as<unsigned>();
    }
    // the descriptors change only when a buffer moves or the caller passes other arrays: uploaded then, in stream order
    // behind the launches that still read the old ones
    const size_t bytes = want.size() * sizeof(CrfFrameDesc);
    if (ctx->spec_shadow.size() != want.size() || std::memcmp(ctx->spec_shadow.data(), want.data(), bytes) != 0) {
        if (ctx->spec_copied) FS_HIP(hipEventSynchronize(ctx->spec_copied));       // the staging block's previous upload
        else FS_HIP(hipEventCreateWithFlags(&ctx->spec_copied, hipEventDisableTiming));
        FS_HIP(ctx->spec_descs_host.ensure(bytes));
        FS_HIP(ctx->spec_descs.ensure(bytes));
        std::memcpy(ctx->spec_descs_host.p, want.data(), bytes);
        FS_HIP(hipMemcpyAsync(ctx->spec_descs.p, ctx->spec_descs_host.p, bytes, hipMemcpyHostToDevice, ctx->stream));
        FS_HIP(hipEventRecord(ctx->spec_copied, ctx->stream));
        ctx->spec_shadow = want;
    }
    const CrfFrameDesc *dd = ctx->spec_descs.as<CrfFrameDesc>();
    const CrfGeom geom{H, W, 0, 0, H, W, ctx->r};
    if (!ctx->queue_clean) FS_HIP(hipMemsetAsync(ctx->crf_queue.p, 0, kCrfQueueSlots * kCrfQueueBytes, ctx->stream));
    ctx->queue_clean = false;
    for (int it = 0; it < iterations; ++it) {
        ScopedTimer t(ctx, G_ITER);
        const int mode = (it == iterations - 1 ? CRF_LAST : 0) | (it == 0 ? CRF_FIRST : 0);
        unsigned *queue = ctx->crf_queue.as<unsigned>() + (size_t)it * (kCrfQueueBytes / sizeof(unsigned));
        hipError_t e = launch_crf_iter_fused(dd + it, 1, kSpecClass, geom, ctx->params, ctx->wtab.as<float>(), mode, queue, ctx->stream);
        if (e != hipSuccess) return ctx->fail(e, "crf iteration launch");
        if (it == 0) ++ctx->publish_expected;        // the launch is out: its first block will count it too
    }
    return 0;
}

}  // namespace

extern "C" {

int fs_create(fs_ctx **out, int device, int height, int width, int max_batch)
{
    if (!out) return -1;
    *out = nullptr;
    if (height < 16 || width < 16 || max_batch < 1) {
        g_create_error = "fs_create: need height, width >= 16 and max_batch >= 1";
        return -1;
    }
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || device < 0 || device >= count) {
        g_create_error = std::string("fs_create: no such HIP device ") + std::to_string(device) + " (" + (e != hipSuccess ? hipGetErrorString(e) : "out of range") + ")";
        return -1;
    }
    fs_ctx *ctx = new fs_ctx();
    ctx->device = device;
    ctx->H = height;
    ctx->W = width;
    ctx->max_batch = max_batch;
    auto bail = [&](int) {
        g_create_error = ctx->err;
        fs_destroy(ctx);
        return -1;
    };
    if (hipSetDevice(device) != hipSuccess) { ctx->err = "hipSetDevice failed"; return bail(0); }
    if (hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking) != hipSuccess) { ctx->err = "hipStreamCreate failed"; return bail(0); }
    ctx->stream = ctx->own_stream;
    if (hipStreamCreateWithFlags(&ctx->in_stream, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&ctx->out_stream, hipStreamNonBlocking) != hipSuccess) { ctx->err = "hipStreamCreate failed"; return bail(0); }
    if (const char *e = getenv("FS_DEBUG_NAN")) ctx->debug_fill = e;          // (environment switches are read here, once)
    if (const char *e = getenv("FS_CRF_PLANES")) ctx->use_planes = e[0] == '1';
    if (const char *e = getenv("FS_NO_SPECULATION")) ctx->speculate = e[0] != '1';
    if (const char *e = getenv("FS_NO_EARLY_DOWNLOADS")) ctx->early_downloads = e[0] != '1';
    if (const char *e = getenv("FS_NO_NARROW_DOWNLOADS")) ctx->narrow_downloads = e[0] != '1';
    if (const char *e = getenv("FS_WIDEN_THREADS")) ctx->widen_threads = std::max(1, std::min(64, atoi(e)));
    if (build_tables(ctx)) return bail(0);
    if (ensure_frame_buffers(ctx)) return bail(0);
    // this GPU's copy of the static work space of fs_connected_components_noalloc: looked up here, once per
    // device, so that the entry point itself does nothing but launch
    if (hipError_t se = noalloc_scratch_init(device, nullptr); se != hipSuccess) { ctx->fail(se, "static work space"); return bail(0); }
    *out = ctx;
    return 0;
}

void fs_destroy(fs_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    DevBuf *bufs[] = {&ctx->trace_buf, &ctx->spec_descs, &ctx->apply_stats, &ctx->publish_seq, &ctx->tanx, &ctx->tany, &ctx->ax, &ctx->ay, &ctx->pos_terms, &ctx->wtab, &ctx->depth16, &ctx->rgb8, &ctx->depthf,
                      &ctx->smoothed, &ctx->points, &ctx->normals, &ctx->raw, &ctx->angles, &ctx->curv, &ctx->scores,
                      &ctx->mask, &ctx->edges, &ctx->patches64, &ctx->patches32, &ctx->surfaces64, &ctx->cc_parent,
                      &ctx->cc_count, &ctx->cc_newid, &ctx->cc_blocksum, &ctx->counts, &ctx->crf_queue, &ctx->curv_todo, &ctx->tile_recs, &ctx->st_in, &ctx->st_work, &ctx->st_descs, &ctx->st_lin,
                      &ctx->st_scratch, &ctx->st_out, &ctx->st_pool, &ctx->simH, &ctx->qa, &ctx->qb,
                      &ctx->pot, &ctx->unary, &ctx->f1, &ctx->f2, &ctx->f3, &ctx->rows, &ctx->descs, &ctx->featA, &ctx->featB, &ctx->msg, &ctx->tile_order, &ctx->narrow_dev};
    for (DevBuf *b : bufs) b->release();
    for (auto &v : ctx->ev)
        for (auto &p : v) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    if (ctx->in_stream) (void)hipStreamDestroy(ctx->in_stream);
    if (ctx->out_stream) (void)hipStreamDestroy(ctx->out_stream);
    for (hipEvent_t e : ctx->pipe_events) (void)hipEventDestroy(e);
    if (ctx->descs_copied) (void)hipEventDestroy(ctx->descs_copied);
    if (ctx->spec_copied) (void)hipEventDestroy(ctx->spec_copied);
    if (ctx->counts_copied) (void)hipEventDestroy(ctx->counts_copied);
    for (hipEvent_t e : ctx->early_events)
        if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : ctx->stats_events)
        if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : ctx->stats_up_events)
        if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : ctx->narrow_events) (void)hipEventDestroy(e);
    ctx->narrow_host.release();
    ctx->spec_descs_host.release();
    ctx->publish_host.release();
    ctx->stats_small_host.release();
    ctx->apply_stats_host.release();
    ctx->stats_in_host.release();
    ctx->frames_in_host.release();
    ctx->descs_host.release();
    ctx->counts_host.release();
    delete ctx;
}

const char *fs_last_error(const fs_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int fs_set_view_angles(fs_ctx *ctx, double view_x, double view_y)
{
    if (!ctx) return -1;
    if (set_device(ctx)) return -1;
    FS_HIP(hipStreamSynchronize(ctx->stream));     // kernels in flight still read the tables this call rewrites
    ctx->view_x = view_x;
    ctx->view_y = view_y;
    return build_tables(ctx);
}

int fs_set_gauss_params(fs_ctx *ctx, const float p[13])
{
    if (!ctx || !p) return -1;
    if (set_device(ctx)) return -1;
    FS_HIP(hipStreamSynchronize(ctx->stream));     // kernels in flight still read the tables this call rewrites
    CrfParams &q = ctx->params;
    q.w1 = p[0]; q.w2 = p[1]; q.w3 = p[2]; q.bias = p[3];
    q.t11 = p[4] * p[4]; q.t21 = p[5] * p[5]; q.t31 = p[6] * p[6];
    q.t12 = p[7] * p[7]; q.t22 = p[8] * p[8]; q.t32 = p[9] * p[9];
    q.t23 = p[10] * p[10]; q.t33 = p[11] * p[11];
    q.weight = p[12];
    q.evidence = 0.00005f;
    q.mode = 0;
    ctx->have_params = true;
    return build_pos_terms(ctx);
}

int fs_set_lr_params(fs_ctx *ctx, const float p[7])
{
    if (!ctx || !p) return -1;
    if (set_device(ctx)) return -1;
    FS_HIP(hipStreamSynchronize(ctx->stream));     // kernels in flight still read the tables this call rewrites
    CrfParams &q = ctx->params;
    for (int i = 0; i < 4; ++i) q.lr_coef[i] = p[i];
    q.bias = p[4];
    q.weight = p[5];
    q.evidence = p[6];
    q.mode = 1;
    ctx->have_params = true;
    return build_pos_terms(ctx);
}

int fs_set_kernel_size(fs_ctx *ctx, int kernel_size)
{
    if (!ctx) return -1;
    if (kernel_size < 1 || kernel_size > kMaxRadius) return ctx->fail("kernel_size must be in [1, " + std::to_string(kMaxRadius) + "]");
    if (set_device(ctx)) return -1;
    FS_HIP(hipStreamSynchronize(ctx->stream));     // kernels in flight still read the tables this call rewrites
    ctx->r = kernel_size;
    if (ensure_frame_buffers(ctx)) return -1;
    return ctx->have_params ? build_pos_terms(ctx) : 0;
}

int fs_set_max_batch(fs_ctx *ctx, int max_batch)
{
    if (!ctx) return -1;
    if (max_batch < 1) return ctx->fail("fs_set_max_batch: need max_batch >= 1");
    if (set_device(ctx)) return -1;
    if (max_batch <= ctx->max_batch) return 0;          // buffers only grow
    FS_HIP(hipStreamSynchronize(ctx->stream));             // kernels in flight still use the buffers this call replaces
    const int old = ctx->max_batch;
    ctx->max_batch = max_batch;
    if (ensure_frame_buffers(ctx)) { ctx->max_batch = old; return -1; }
    return 0;
}

int fs_set_stream(fs_ctx *ctx, void *hip_stream)
{
    if (!ctx) return -1;
    if (set_device(ctx)) return -1;
    // nothing of this context stays behind on the stream it leaves (a handle the caller has destroyed already
    // makes this fail, which is as good as drained)
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) (void)hipGetLastError();
    ctx->stream = reinterpret_cast<hipStream_t>(hip_stream);   // a null handle is the legacy default stream
    return 0;
}

int fs_use_own_stream(fs_ctx *ctx)
{
    if (!ctx) return -1;
    if (set_device(ctx)) return -1;
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) (void)hipGetLastError();
    ctx->stream = ctx->own_stream;
    return 0;
}

int fs_synchronize(fs_ctx *ctx)
{
    if (!ctx) return -1;
    if (set_device(ctx)) return -1;
    FS_HIP(hipStreamSynchronize(ctx->stream));
    return 0;
}

void *fs_stream(fs_ctx *ctx) { return ctx ? reinterpret_cast<void *>(ctx->stream) : nullptr; }

int fs_set_debug_fill(fs_ctx *ctx, const char *letters)
{
    if (!ctx) return -1;
    ctx->debug_fill = letters ? letters : "";
    return 0;
}

int fs_set_timing(fs_ctx *ctx, int enable)
{
    if (!ctx) return -1;
    ctx->timing = (unsigned)enable;
    return 0;
}

int fs_get_speculation_counts(fs_ctx *ctx, int64_t *held, int64_t *repeated)
{
    if (!ctx) return -1;
    if (held) *held = ctx->spec_held;
    if (repeated) *repeated = ctx->spec_repeated;
    return 0;
}

int fs_get_timing(fs_ctx *ctx, int group, double *total_ms, int64_t *launches)
{
    if (!ctx || group < 0 || group >= G_COUNT) return -1;
    if (set_device(ctx)) return -1;
    FS_HIP(hipStreamSynchronize(ctx->stream));
    double tot = 0.0;
    for (size_t i = 0; i < ctx->ev_used[group]; ++i) {
        float ms = 0.0f;
        FS_HIP(hipEventElapsedTime(&ms, ctx->ev[group][i].a, ctx->ev[group][i].b));
        tot += ms;
    }
    if (total_ms) *total_ms = tot;
    if (launches) *launches = (int64_t)ctx->ev_used[group];
    ctx->ev_used[group] = 0;
    return 0;
}

int fs_get_tile_labels(fs_ctx *ctx, int n, int32_t *listed, int *tiles_x, int *tiles_y, int *tile_w, int *tile_h)
{
    if (check_batch(ctx, n)) return -1;
    const int th = crf_fused_tile_height(ctx->r);
    if (!th) return ctx->fail("fs_get_tile_labels: no per-tile label lists at this window radius");
    int tx = 0, ty = 0;
    crf_fused_tile_grid(ctx->H, ctx->W, ctx->r, &tx, &ty);
    if (tiles_x) *tiles_x = tx;
    if (tiles_y) *tiles_y = ty;
    if (tile_w) *tile_w = 32;
    if (tile_h) *tile_h = th;
    if (listed) {
        const int total = n * tx * ty;
        FS_HIP(ctx->st_pool.ensure((size_t)total * sizeof(int32_t)));
        launch_crf_tile_counts(ctx->tile_recs.as<int32_t>(), ctx->st_pool.as<int32_t>(), total, ctx->stream);
        FS_HIP(hipGetLastError());
        FS_HIP(hipMemcpyAsync(listed, ctx->st_pool.p, (size_t)total * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
        FS_HIP(hipStreamSynchronize(ctx->stream));
    }
    return 0;
}

int fs_smooth_depth(fs_ctx *ctx, int n, const float *depth, float *out)
{
    if (check_batch(ctx, n)) return -1;
    if (!depth || !out) return ctx->fail("fs_smooth_depth: null argument");
    const size_t bytes = (size_t)n * ctx->N() * 4;
    FS_HIP(ctx->depthf.ensure(bytes));
    FS_HIP(hipMemcpyAsync(ctx->depthf.p, depth, bytes, hipMemcpyHostToDevice, ctx->stream));
    launch_smooth_depth_f32(ctx->depthf.as<float>(), ctx->smoothed.as<float>(), n, ctx->H, ctx->W, ctx->stream);
    FS_HIP(hipGetLastError());
    FS_HIP(hipMemcpyAsync(out, ctx->smoothed.p, bytes, hipMemcpyDeviceToHost, ctx->stream));
    FS_HIP(hipStreamSynchronize(ctx->stream));
    return 0;
}

int fs_normals(fs_ctx *ctx, int n, const float *smoothed, float *angles, float *vectors, float *points, float *raw_angles)
{
    if (check_batch(ctx, n)) return -1;
    if (!smoothed) return ctx->fail("fs_normals: null argument");
    const size_t px = (size_t)n * ctx->N();
    FS_HIP(hipMemcpyAsync(ctx->smoothed.p, smoothed, px * 4, hipMemcpyHostToDevice, ctx->stream));
    launch_normals(ctx->smoothed.as<float>(), ctx->tanx.as<float>(), ctx->tany.as<float>(), ctx->points.as<float>(),
                   ctx->normals.as<float>(), ctx->raw.as<float>(), n, ctx->H, ctx->W, ctx->stream);
    launch_angle_filter(ctx->raw.as<float>(), ctx->smoothed.as<float>(), ctx->angles.as<float>(), nullptr, 1.0f, n, ctx->H, ctx->W, ctx->stream);
    FS_HIP(hipGetLastError());
    if (angles) FS_HIP(hipMemcpyAsync(angles, ctx->angles.p, px * 8, hipMemcpyDeviceToHost, ctx->stream));
    if (vectors) FS_HIP(hipMemcpyAsync(vectors, ctx->normals.p, px * 12, hipMemcpyDeviceToHost, ctx->stream));
    if (points) FS_HIP(hipMemcpyAsync(points, ctx->points.p, px * 12, hipMemcpyDeviceToHost, ctx->stream));
    if (raw_angles) FS_HIP(hipMemcpyAsync(raw_angles, ctx->raw.p, px * 8, hipMemcpyDeviceToHost, ctx->stream));
    FS_HIP(hipStreamSynchronize(ctx->stream));
    return 0;
}

int fs_patches(fs_ctx *ctx, int n, const float *smoothed, int64_t *patches, int32_t *edges, float *curvature_scores,
               uint8_t *curvature_mask)
{
    if (check_batch(ctx, n)) return -1;
    if (!smoothed) return ctx->fail("fs_patches: null argument");
    const size_t px = (size_t)n * ctx->N();
    FS_HIP(hipMemcpyAsync(ctx->smoothed.p, smoothed, px * 4, hipMemcpyHostToDevice, ctx->stream));
    if (curvature_scores) FS_HIP(ctx->scores.ensure(px * 4));
    if (run_patches(ctx, n, ctx->smoothed.as<float>(), ctx->patches64.as<int64_t>(), nullptr, ctx->edges.as<int32_t>(),
                    curvature_scores ? ctx->scores.as<float>() : nullptr, ctx->curv.as<uint8_t>(), false))
        return -1;
    if (patches) FS_HIP(hipMemcpyAsync(patches, ctx->patches64.p, px * 8, hipMemcpyDeviceToHost, ctx->stream));
    if (edges) FS_HIP(hipMemcpyAsync(edges, ctx->edges.p, px * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (curvature_scores) FS_HIP(hipMemcpyAsync(curvature_scores, ctx->scores.p, px * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (curvature_mask) FS_HIP(hipMemcpyAsync(curvature_mask, ctx->curv.p, px, hipMemcpyDeviceToHost, ctx->stream));
    FS_HIP(hipStreamSynchronize(ctx->stream));
    return 0;
}

int fs_connected_components(fs_ctx *ctx, int n, const uint8_t *mask, int min_size_exclusive, int64_t *labels, int32_t *num_labels)
{
    if (check_batch(ctx, n)) return -1;
    if (!mask) return ctx->fail("fs_connected_components: null argument");
    const size_t px = (size_t)n * ctx->N();
    FS_HIP(hipMemcpyAsync(ctx->mask.p, mask, px, hipMemcpyHostToDevice, ctx->stream));
    launch_connected_components(ctx->mask.as<uint8_t>(), cc_work(ctx), min_size_exclusive, ctx->patches64.as<int64_t>(), nullptr, n, ctx->H, ctx->W, ctx->stream);
    FS_HIP(hipGetLastError());
    if (labels) FS_HIP(hipMemcpyAsync(labels, ctx->patches64.p, px * 8, hipMemcpyDeviceToHost, ctx->stream));
    if (num_labels) FS_HIP(hipMemcpyAsync(num_labels, ctx->counts.p, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    FS_HIP(hipStreamSynchronize(ctx->stream));
    return 0;
}

int fs_crf_step(fs_ctx *ctx, int B, int h, int w, int S, const float *unary, const float *Q, const float *f1, const float *f2,
                const float *f3, const float *val, float *out, float *potentials)
{
    if (!ctx) return -1;
    if (set_device(ctx)) return -1;
    if (!ctx->have_params || ctx->params.mode != 0) return ctx->fail("fs_crf_step: call fs_set_gauss_params first");
    if (B < 1 || h < 1 || w < 1 || S < 2) return ctx->fail("fs_crf_step: bad shape");
    if (S > kMaxTileChannels) return ctx->fail("fs_crf_step: at most " + std::to_string(kMaxTileChannels) + " label channels");
    if (!unary || !Q || !f1 || !f2 || !f3 || !val || !out) return ctx->fail("fs_crf_step: null argument");
    for (int l = 0; l < S; ++l)
        if (val[l] != (l == S - 1 ? 1.0f : 0.0f)) return ctx->fail("fs_crf_step: val must be the one-hot of the last label");
    const int r = ctx->r, hp = h + 2 * r, wp = w + 2 * r;
    const size_t NP = (size_t)hp * wp, NO = (size_t)h * w;
    float *sim0 = nullptr;
    size_t sim_stride = 0;
    if (ensure_sim(ctx, NP, wp, 1, &sim0, &sim_stride)) return -1;
    FS_HIP(ctx->rows.ensure(NP * S * 4));
    FS_HIP(ctx->qa.ensure(NP * S * 4));
    FS_HIP(ctx->qb.ensure(NO * S * 4));
    FS_HIP(ctx->unary.ensure(NO * S * 4));
    FS_HIP(ctx->pot.ensure(NO * S * 4));
    FS_HIP(ctx->f1.ensure(NP * 3 * 4));
    FS_HIP(ctx->f2.ensure(NP * 6 * 4));
    FS_HIP(ctx->f3.ensure(NP * 5 * 4));
    FS_HIP(ctx->descs.ensure(sizeof(CrfFrameDesc)));
    CrfFrameDesc d{};
    d.sim0 = sim0;
    d.Qin = ctx->qa.as<float>();
    d.Qout = ctx->qb.as<float>();
    d.potentials = potentials ? ctx->pot.as<float>() : nullptr;
    d.unary = ctx->unary.as<float>();
    d.S = S;
    FS_HIP(hipMemcpyAsync(ctx->descs.p, &d, sizeof(d), hipMemcpyHostToDevice, ctx->stream));
    FS_HIP(hipStreamSynchronize(ctx->stream));
    const CrfGeom geom{hp, wp, r, r, h, w, r};
    float *rows = ctx->rows.as<float>();
    for (int b = 0; b < B; ++b) {
        // pixel-major host tensors -> label planes
        FS_HIP(hipMemcpyAsync(rows, Q + (size_t)b * NP * S, NP * S * 4, hipMemcpyHostToDevice, ctx->stream));
        launch_rows_to_planes(rows, ctx->qa.as<float>(), NP, S, ctx->stream);
        FS_HIP(hipMemcpyAsync(rows, unary + (size_t)b * NO * S, NO * S * 4, hipMemcpyHostToDevice, ctx->stream));
        launch_rows_to_planes(rows, ctx->unary.as<float>(), NO, S, ctx->stream);
        FS_HIP(hipMemcpyAsync(ctx->f1.p, f1 + (size_t)b * NP * 3, NP * 3 * 4, hipMemcpyHostToDevice, ctx->stream));
        FS_HIP(hipMemcpyAsync(ctx->f2.p, f2 + (size_t)b * NP * 6, NP * 6 * 4, hipMemcpyHostToDevice, ctx->stream));
        FS_HIP(hipMemcpyAsync(ctx->f3.p, f3 + (size_t)b * NP * 5, NP * 5 * 4, hipMemcpyHostToDevice, ctx->stream));
        launch_crf_sim_explicit(ctx->f1.as<float>(), ctx->f2.as<float>(), ctx->f3.as<float>(), ctx->params, sim0, hp, wp, r, ctx->stream);
        hipError_t e = launch_crf_iter(ctx->descs.as<CrfFrameDesc>(), 1, crf_channel_class(S), geom, ctx->params.weight, ctx->params.evidence, 0, ctx->stream);
        if (e != hipSuccess) return ctx->fail(e, "fs_crf_step: launch");
        launch_planes_to_rows(ctx->qb.as<float>(), rows, NO, S, ctx->stream);
        FS_HIP(hipMemcpyAsync(out + (size_t)b * NO * S, rows, NO * S * 4, hipMemcpyDeviceToHost, ctx->stream));
        if (potentials) {
            launch_planes_to_rows(ctx->pot.as<float>(), rows, NO, S, ctx->stream);
            FS_HIP(hipMemcpyAsync(potentials + (size_t)b * NO * S, rows, NO * S * 4, hipMemcpyDeviceToHost, ctx->stream));
        }
        FS_HIP(hipStreamSynchronize(ctx->stream));
    }
    return 0;
}

int fs_crf_step_lr(fs_ctx *ctx, int B, int h, int w, int S, const float *unary, const float *Q, const float *f1, const float *f2,
                   const float *f3, const float *f4, const float *val, float *out, float *potentials)
{
    if (!ctx) return -1;
    if (set_device(ctx)) return -1;
    if (!ctx->have_params || ctx->params.mode != 1) return ctx->fail("fs_crf_step_lr: call fs_set_lr_params first");
    if (B < 1 || h < 1 || w < 1 || S < 2 || S > kMaxLabels) return ctx->fail("fs_crf_step_lr: bad shape");
    if (!unary || !Q || !f1 || !f2 || !f3 || !f4 || !val || !out) return ctx->fail("fs_crf_step_lr: null argument");
    for (int l = 0; l < S; ++l)
        if (val[l] != (l == S - 1 ? 1.0f : 0.0f)) return ctx->fail("fs_crf_step_lr: val must be the one-hot of the last label");
    const int r = ctx->r;
    const size_t NP = (size_t)(h + 2 * r) * (w + 2 * r), NO = (size_t)h * w;
    FS_HIP(ctx->qa.ensure(NP * S * 4));
    FS_HIP(ctx->qb.ensure(NO * S * 4));
    FS_HIP(ctx->unary.ensure(NO * S * 4));
    FS_HIP(ctx->pot.ensure(NO * S * 4));
    FS_HIP(ctx->f1.ensure(NP * 3 * 4));      // f1 | f4 share this buffer: 1 + 2 floats per pixel
    FS_HIP(ctx->f2.ensure(NP * 2 * 4));
    FS_HIP(ctx->f3.ensure(NP * 3 * 4));
    float *d_f1 = ctx->f1.as<float>(), *d_f4 = d_f1 + NP;
    for (int b = 0; b < B; ++b) {
        FS_HIP(hipMemcpyAsync(ctx->qa.p, Q + (size_t)b * NP * S, NP * S * 4, hipMemcpyHostToDevice, ctx->stream));
        FS_HIP(hipMemcpyAsync(ctx->unary.p, unary + (size_t)b * NO * S, NO * S * 4, hipMemcpyHostToDevice, ctx->stream));
        FS_HIP(hipMemcpyAsync(d_f1, f1 + (size_t)b * NP, NP * 4, hipMemcpyHostToDevice, ctx->stream));
        FS_HIP(hipMemcpyAsync(ctx->f2.p, f2 + (size_t)b * NP * 2, NP * 2 * 4, hipMemcpyHostToDevice, ctx->stream));
        FS_HIP(hipMemcpyAsync(ctx->f3.p, f3 + (size_t)b * NP * 3, NP * 3 * 4, hipMemcpyHostToDevice, ctx->stream));
        FS_HIP(hipMemcpyAsync(d_f4, f4 + (size_t)b * NP * 2, NP * 2 * 4, hipMemcpyHostToDevice, ctx->stream));
        launch_crf_lr_explicit(ctx->unary.as<float>(), ctx->qa.as<float>(), d_f1, ctx->f2.as<float>(), ctx->f3.as<float>(), d_f4,
                               ctx->params.lr_coef, ctx->params.bias, ctx->params.weight, ctx->params.evidence, ctx->qb.as<float>(),
                               potentials ? ctx->pot.as<float>() : nullptr, h, w, r, S, ctx->stream);
        FS_HIP(hipGetLastError());
        FS_HIP(hipMemcpyAsync(out + (size_t)b * NO * S, ctx->qb.p, NO * S * 4, hipMemcpyDeviceToHost, ctx->stream));
        if (potentials) FS_HIP(hipMemcpyAsync(potentials + (size_t)b * NO * S, ctx->pot.p, NO * S * 4, hipMemcpyDeviceToHost, ctx->stream));
        FS_HIP(hipStreamSynchronize(ctx->stream));
    }
    return 0;
}

}  // extern "C"

namespace {
// The stage on n frames resident in device memory.  after_stage (may be empty) is called when the kernels that
// complete a group of outputs have been enqueued: 0 = angles, vectors, points; 1 = patches, edges.
using StageHook = std::function<int(int)>;
int find_surfaces_dev(fs_ctx *ctx, int n, const uint16_t *d_depth, const uint8_t *d_rgb, float *d_angles, float *d_vectors,
                      float *d_points, int64_t *d_patches, int32_t *d_edges, int64_t *d_surfaces, int32_t *num_surfaces,
                      const StageHook &after_stage)
{
    if (check_batch(ctx, n)) return -1;
    if (!ctx->have_params) return ctx->fail("fs_find_surfaces: call fs_set_gauss_params first");
    if (!d_depth || !d_rgb) return ctx->fail("fs_find_surfaces: depth and rgb are required");
    const int H = ctx->H, W = ctx->W;
    float *angles = d_angles ? d_angles : ctx->angles.as<float>();
    float *vectors = d_vectors ? d_vectors : ctx->normals.as<float>();
    float *points = d_points ? d_points : ctx->points.as<float>();
    int64_t *patches = d_patches ? d_patches : ctx->patches64.as<int64_t>();
    int32_t *edges = d_edges ? d_edges : ctx->edges.as<int32_t>();
    int64_t *surfaces = d_surfaces ? d_surfaces : ctx->surfaces64.as<int64_t>();
    const bool fused_crf = !ctx->use_planes && crf_fused_tile_height(ctx->r) != 0;     // as run_crf_batch decides
    // One frame per call: the updates go out before the label count has come back (launch_crf_speculative).
    const bool speculate = ctx->speculate && n == 1 && fused_crf && ctx->trace_cap == 0;
    if (debug_nan_fill(ctx)) return -1;
    {
        ScopedTimer t(ctx, G_STENCIL);
        // (the colour half of the fused CRF update's feature records is written on the way: it needs the same depth)
        launch_smooth_depth_u16(d_depth, ctx->smoothed.as<float>(), n, H, W, ctx->stream, fused_crf ? d_rgb : nullptr,
                                fused_crf ? ctx->featA.p : nullptr, crf_colour_feature_scale(ctx->params));   // find_surfaces.py:505
        launch_normals(ctx->smoothed.as<float>(), ctx->tanx.as<float>(), ctx->tany.as<float>(), points, vectors,
                       ctx->raw.as<float>(), n, H, W, ctx->stream);                                            // :507
        // the fused CRF update wants the angles once more, scaled, in its feature records: the filter writes both
        launch_angle_filter(ctx->raw.as<float>(), ctx->smoothed.as<float>(), angles, fused_crf ? ctx->featB.p : nullptr,
                            crf_angle_feature_scale(ctx->params), n, H, W, ctx->stream);
    }
    if (after_stage && after_stage(0)) return -1;
    if (run_patches(ctx, n, ctx->smoothed.as<float>(), patches, ctx->patches32.as<int32_t>(), edges, nullptr,
                    ctx->curv.as<uint8_t>(), speculate ? 2 : (after_stage ? 3 : 1)))                           // :513
        return -1;
    if (after_stage && after_stage(1)) return -1;
    if (!speculate && after_stage) {
        // (the hook's commands went in between the count copy and its wait)
        FS_HIP(hipEventSynchronize(ctx->counts_copied));
    }
    if (speculate) {
        if (launch_crf_speculative(ctx, d_depth, ctx->patches32.as<int32_t>(), 3, surfaces)) return -1;        // :515-532
        if (wait_published_counts(ctx)) return -1;
    }
    std::vector<int> S(n), maxa(n);
    const int32_t *counts = ctx->counts_host.as<int32_t>();
    for (int i = 0; i < n; ++i) {
        const int ns = counts[i] + 1;                                                                          // :514
        if (num_surfaces) num_surfaces[i] = ns;
        S[i] = ns + 1;
        maxa[i] = counts[n + i];
    }
    // the updates in flight are the frame's if the frame fits them; otherwise they gave up and the regular path runs
    if (speculate) {
        if (spec_fits(S[0], maxa[0], ctx->spec_cap)) {
            ++ctx->spec_held;
            return 0;
        }
        ++ctx->spec_repeated;
    }
    if (run_crf_batch(ctx, n, d_depth, d_rgb, angles, ctx->patches32.as<int32_t>(), S.data(), maxa.data(), 3, surfaces, fused_crf))   // :515-532
        return -1;
    FS_HIP(hipGetLastError());
    return 0;
}

}  // namespace

extern "C" {

int fs_find_surfaces_batch_dev(fs_ctx *ctx, int n, const uint16_t *d_depth, const uint8_t *d_rgb, float *d_angles,
                               float *d_vectors, float *d_points, int64_t *d_patches, int32_t *d_edges, int64_t *d_surfaces,
                               int32_t *num_surfaces)
{
    return find_surfaces_dev(ctx, n, d_depth, d_rgb, d_angles, d_vectors, d_points, d_patches, d_edges, d_surfaces, num_surfaces, StageHook());
}

int fs_find_surfaces_batch(fs_ctx *ctx, int n, const uint16_t *depth, const uint8_t *rgb, float *angles, float *vectors,
                           float *points, int64_t *patches, int32_t *edges, int64_t *surfaces, int32_t *num_surfaces)
{
    if (check_batch(ctx, n)) return -1;
    if (!depth || !rgb) return ctx->fail("fs_find_surfaces: depth and rgb are required");
    const size_t N = ctx->N();
    FS_HIP(ctx->depth16.ensure((size_t)ctx->max_batch * N * 2));
    FS_HIP(ctx->rgb8.ensure((size_t)ctx->max_batch * N * 3));
    // Frames go through in chunks: the upload of chunk k+1 and the download of chunk k-1 run on their
    // own streams beside the kernels of chunk k.  With pinned host memory (fs_host_alloc) the copies are
    // true DMA transfers; with pageable memory the runtime stages them and the overlap is partial.
    const int C = kPipeChunk;
    const int chunks = (n + C - 1) / C;
    while (ctx->pipe_events.size() < (size_t)2 * chunks) {
        hipEvent_t e;
        FS_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        ctx->pipe_events.push_back(e);
    }
    // Frames of a small call that arrive in pageable memory are copied into page-locked memory first (the runtime's own
    // path for pageable sources holds the calling thread far longer than a memcpy of the same bytes); a large call is
    // expected to bring page-locked arrays (fs_host_alloc) and is passed through as it is.
    const uint16_t *depth_src = depth;
    const uint8_t *rgb_src = rgb;
    if (chunks == 1 && FS_STAGE_PAGEABLE) {
        const bool dl = host_memory_is_page_locked(depth), cl = host_memory_is_page_locked(rgb);
        if (!dl || !cl) {
            FS_HIP(ctx->frames_in_host.ensure((size_t)n * N * 5));
            char *stage = ctx->frames_in_host.as<char>();
            if (!dl) { std::memcpy(stage, depth, (size_t)n * N * 2); depth_src = reinterpret_cast<const uint16_t *>(stage); }
            if (!cl) { std::memcpy(stage + (size_t)n * N * 2, rgb, (size_t)n * N * 3); rgb_src = reinterpret_cast<const uint8_t *>(stage + (size_t)n * N * 2); }
        }
    }
    auto upload = [&](int k) -> int {
        const size_t f0 = (size_t)k * C, cn = std::min<size_t>(C, n - f0);
        FS_HIP(hipMemcpyAsync(ctx->depth16.as<uint16_t>() + f0 * N, depth_src + f0 * N, cn * N * 2, hipMemcpyHostToDevice, ctx->in_stream));
        FS_HIP(hipMemcpyAsync(ctx->rgb8.as<uint8_t>() + f0 * N * 3, rgb_src + f0 * N * 3, cn * N * 3, hipMemcpyHostToDevice, ctx->in_stream));
        FS_HIP(hipEventRecord(ctx->pipe_events[2 * k], ctx->in_stream));
        return 0;
    };
    // the previous call's kernels may still read the input buffers when the caller drives the stream itself
    FS_HIP(hipStreamSynchronize(ctx->stream));
    const bool early = ctx->early_downloads;
    for (hipEvent_t &e : ctx->early_events)
        if (early && !e) FS_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    // a call that fails leaves nothing in flight from or towards the caller's arrays
    auto drained = [&]() {
        (void)hipStreamSynchronize(ctx->in_stream);
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipStreamSynchronize(ctx->out_stream);
        return -1;
    };
    // Calls of several chunks: the label images cross the link narrow and are widened by worker threads (see narrow_dev)
    const bool narrow = ctx->narrow_downloads && chunks > 1 && N % 4 == 0 && (patches || edges || surfaces);
    int16_t *p16d = nullptr, *s16d = nullptr, *p16h = nullptr, *s16h = nullptr;
    uint8_t *e8d = nullptr, *e8h = nullptr;
    WidenJobs jobs;
    std::vector<std::thread> &workers = jobs.workers;
    const int parts = narrow ? ctx->widen_threads : 0;
    if (narrow) {
        const size_t total = (size_t)n * N;
        FS_HIP(ctx->narrow_dev.ensure(total * 5));
        FS_HIP(ctx->narrow_host.ensure(total * 5));
        p16d = ctx->narrow_dev.as<int16_t>(); s16d = p16d + total; e8d = reinterpret_cast<uint8_t *>(s16d + total);
        p16h = ctx->narrow_host.as<int16_t>(); s16h = p16h + total; e8h = reinterpret_cast<uint8_t *>(s16h + total);
        while (ctx->narrow_events.size() < (size_t)chunks) {
            hipEvent_t e;
            FS_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            ctx->narrow_events.push_back(e);
        }
        const int device = ctx->device;
        std::vector<hipEvent_t> *events = &ctx->narrow_events;
        try {
        for (int t = 0; t < parts; ++t)
            workers.emplace_back([&jobs, events, device, chunks, parts, C, n, N, p16h, s16h, e8h, patches, surfaces, edges]() {
                if (hipSetDevice(device) != hipSuccess) jobs.failed = 1;
                for (;;) {
                    const int j = jobs.next.fetch_add(1);
                    if (j >= chunks * parts) return;
                    const int k = j / parts, part = j - k * parts;
                    {
                        std::unique_lock<std::mutex> lk(jobs.m);
                        jobs.cv.wait(lk, [&] { return jobs.posted > k || jobs.abort; });
                        if (jobs.abort) return;
                    }
                    if (hipEventSynchronize((*events)[k]) != hipSuccess) { jobs.failed = 1; continue; }
                    const size_t f0 = (size_t)k * C, cn = std::min<size_t>(C, n - f0), px = cn * N;
                    const size_t lo = f0 * N + px * part / parts, hi = f0 * N + px * (part + 1) / parts;
                    if (patches) widen_i16_to_i64(p16h + lo, patches + lo, hi - lo);
                    if (surfaces) widen_i16_to_i64(s16h + lo, surfaces + lo, hi - lo);
                    if (edges) widen_u8_to_i32(e8h + lo, edges + lo, hi - lo);
                }
            });
        } catch (const std::exception &e) {              // (no thread to be had: nothing is in flight yet)
            jobs.finish(false);
            return ctx->fail(std::string("fs_find_surfaces: cannot start the worker threads: ") + e.what());
        }
    }
    auto finish_workers = [&](bool ok) { jobs.finish(ok); };
    auto drained_all = [&]() {
        finish_workers(false);
        return drained();
    };
    if (upload(0)) return drained_all();
    for (int k = 0; k < chunks; ++k) {
        const size_t f0 = (size_t)k * C, cn = std::min<size_t>(C, n - f0);
        if (k + 1 < chunks && upload(k + 1)) return drained_all();
        FS_HIP(hipStreamWaitEvent(ctx->stream, ctx->pipe_events[2 * k], 0));
        const size_t px = cn * N, o = f0 * N;
        // Each group of a chunk's outputs starts its way to the host as soon as the kernels that complete it are
        // enqueued (normals, angles and points after the angle filter, patches and edges after the components), and
        // only the labels wait for the updates: a call of one chunk has nothing else to overlap its downloads with,
        // and with several chunks the link, which bounds the call, starts on a chunk a CRF's length earlier (111
        // frames: 35.9 -> 33.5 ms).
        StageHook hook;
        auto narrow_patches_edges = [&]() -> int {      // on the compute stream, behind the components
            if (!narrow || (!patches && !edges)) return 0;
            const size_t n4 = px / 4;
            hipLaunchKernelGGL(narrow_patches_edges_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, ctx->stream,
                               ctx->patches32.as<int32_t>(), ctx->edges.as<int32_t>() + o, patches ? p16d + o : nullptr,      // (patches32: the chunk's, from 0)
                               edges ? e8d + o : nullptr, n4);
            FS_HIP(hipGetLastError());
            return 0;
        };
        auto download_patches_edges = [&]() -> int {    // on the download stream
            if (narrow) {
                if (patches) FS_HIP(hipMemcpyAsync(p16h + o, p16d + o, px * 2, hipMemcpyDeviceToHost, ctx->out_stream));
                if (edges) FS_HIP(hipMemcpyAsync(e8h + o, e8d + o, px, hipMemcpyDeviceToHost, ctx->out_stream));
            } else {
                if (patches) FS_HIP(hipMemcpyAsync(patches + o, ctx->patches64.as<int64_t>() + o, px * 8, hipMemcpyDeviceToHost, ctx->out_stream));
                if (edges) FS_HIP(hipMemcpyAsync(edges + o, ctx->edges.as<int32_t>() + o, px * 4, hipMemcpyDeviceToHost, ctx->out_stream));
            }
            return 0;
        };
        if (early) hook = [&](int stage) -> int {
            if (stage == 1 && narrow_patches_edges()) return -1;
            FS_HIP(hipEventRecord(ctx->early_events[stage], ctx->stream));
            FS_HIP(hipStreamWaitEvent(ctx->out_stream, ctx->early_events[stage], 0));
            if (stage == 0) {
                if (angles) FS_HIP(hipMemcpyAsync(angles + o * 2, ctx->angles.as<float>() + o * 2, px * 8, hipMemcpyDeviceToHost, ctx->out_stream));
                if (vectors) FS_HIP(hipMemcpyAsync(vectors + o * 3, ctx->normals.as<float>() + o * 3, px * 12, hipMemcpyDeviceToHost, ctx->out_stream));
                if (points) FS_HIP(hipMemcpyAsync(points + o * 3, ctx->points.as<float>() + o * 3, px * 12, hipMemcpyDeviceToHost, ctx->out_stream));
            } else {
                if (download_patches_edges()) return -1;
            }
            return 0;
        };
        if (find_surfaces_dev(ctx, (int)cn, ctx->depth16.as<uint16_t>() + f0 * N, ctx->rgb8.as<uint8_t>() + f0 * N * 3,
                              ctx->angles.as<float>() + f0 * N * 2, ctx->normals.as<float>() + f0 * N * 3,
                              ctx->points.as<float>() + f0 * N * 3, ctx->patches64.as<int64_t>() + f0 * N,
                              ctx->edges.as<int32_t>() + f0 * N, ctx->surfaces64.as<int64_t>() + f0 * N,
                              num_surfaces ? num_surfaces + f0 : nullptr, hook))
            return drained_all();
        if (narrow) {
            if (!early && narrow_patches_edges()) return drained_all();
            if (surfaces) {
                const size_t n4 = px / 4;
                hipLaunchKernelGGL(narrow_surfaces_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, ctx->stream,
                                   ctx->surfaces64.as<int64_t>() + o, s16d + o, n4);
                FS_HIP(hipGetLastError());
            }
        }
        FS_HIP(hipEventRecord(ctx->pipe_events[2 * k + 1], ctx->stream));
        FS_HIP(hipStreamWaitEvent(ctx->out_stream, ctx->pipe_events[2 * k + 1], 0));
        if (!early) {
            if (angles) FS_HIP(hipMemcpyAsync(angles + o * 2, ctx->angles.as<float>() + o * 2, px * 8, hipMemcpyDeviceToHost, ctx->out_stream));
            if (vectors) FS_HIP(hipMemcpyAsync(vectors + o * 3, ctx->normals.as<float>() + o * 3, px * 12, hipMemcpyDeviceToHost, ctx->out_stream));
            if (points) FS_HIP(hipMemcpyAsync(points + o * 3, ctx->points.as<float>() + o * 3, px * 12, hipMemcpyDeviceToHost, ctx->out_stream));
            if (download_patches_edges()) return drained_all();
        }
        if (narrow) {
            if (surfaces) FS_HIP(hipMemcpyAsync(s16h + o, s16d + o, px * 2, hipMemcpyDeviceToHost, ctx->out_stream));
            FS_HIP(hipEventRecord(ctx->narrow_events[k], ctx->out_stream));
            {
                std::lock_guard<std::mutex> lk(jobs.m);
                jobs.posted = k + 1;
            }
            jobs.cv.notify_all();
        } else if (surfaces) {
            FS_HIP(hipMemcpyAsync(surfaces + o, ctx->surfaces64.as<int64_t>() + o, px * 8, hipMemcpyDeviceToHost, ctx->out_stream));
        }
    }
    finish_workers(true);
    FS_HIP(hipStreamSynchronize(ctx->out_stream));
    FS_HIP(hipStreamSynchronize(ctx->stream));
    if (jobs.failed) return ctx->fail("fs_find_surfaces: a worker thread could not wait for its chunk's copies");
    return 0;
}

int fs_find_surfaces_trace(fs_ctx *ctx, const uint16_t *depth, const uint8_t *rgb, int max_channels, float *potentials,
                           float *probabilities, int64_t *patches, int64_t *surfaces, int32_t *num_surfaces)
{
    if (check_batch(ctx, 1)) return -1;
    if (!depth || !rgb || max_channels < 2) return ctx->fail("fs_find_surfaces_trace: depth, rgb and max_channels >= 2 are required");
    const size_t N = ctx->N();
    FS_HIP(ctx->depth16.ensure((size_t)ctx->max_batch * N * 2));
    FS_HIP(ctx->rgb8.ensure((size_t)ctx->max_batch * N * 3));
    FS_HIP(hipStreamSynchronize(ctx->stream));
    FS_HIP(hipMemcpyAsync(ctx->depth16.p, depth, N * 2, hipMemcpyHostToDevice, ctx->stream));
    FS_HIP(hipMemcpyAsync(ctx->rgb8.p, rgb, N * 3, hipMemcpyHostToDevice, ctx->stream));
    ctx->trace_cap = max_channels;
    ctx->trace_S = 0;
    int32_t ns = 0;
    const int rc = find_surfaces_dev(ctx, 1, ctx->depth16.as<uint16_t>(), ctx->rgb8.as<uint8_t>(), nullptr, nullptr, nullptr, nullptr, nullptr,
                                     nullptr, &ns, StageHook());
    ctx->trace_cap = 0;
    if (num_surfaces) *num_surfaces = ns;
    if (rc) { (void)hipStreamSynchronize(ctx->stream); return -1; }
    const size_t plane_set = (size_t)ctx->trace_S * N;
    for (int it = 0; it < 3; ++it) {
        const float *src = ctx->trace_buf.as<float>() + (size_t)(2 * it) * plane_set;
        if (potentials) FS_HIP(hipMemcpyAsync(potentials + it * plane_set, src, plane_set * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
        if (probabilities) FS_HIP(hipMemcpyAsync(probabilities + it * plane_set, src + plane_set, plane_set * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    }
    if (patches) FS_HIP(hipMemcpyAsync(patches, ctx->patches64.p, N * 8, hipMemcpyDeviceToHost, ctx->stream));
    if (surfaces) FS_HIP(hipMemcpyAsync(surfaces, ctx->surfaces64.p, N * 8, hipMemcpyDeviceToHost, ctx->stream));
    FS_HIP(hipStreamSynchronize(ctx->stream));
    return 0;
}

void *fs_host_alloc(size_t bytes)
{
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) return nullptr;
    return p;
}

void fs_host_free(void *p)
{
    if (p) (void)hipHostFree(p);
}

int fs_apply_crf(fs_ctx *ctx, int n, const int64_t *labels, const uint16_t *depth, const uint8_t *rgb, const float *angles,
                 int64_t *out)
{
    if (check_batch(ctx, n)) return -1;
    if (!ctx->have_params) return ctx->fail("fs_apply_crf: call fs_set_gauss_params first");
    if (!labels || !depth || !rgb || !angles || !out) return ctx->fail("fs_apply_crf: null argument");
    const size_t N = ctx->N(), px = (size_t)n * N;
    StagedSources up(ctx->stats_in_host);
    const bool out_locked = host_memory_is_page_locked(out);
    if (!out_locked) up.want_block(px * 8);                 // a pageable result array is filled from page-locked memory as well
    up.want(labels, px * 8);
    up.want(depth, px * 2);
    up.want(rgb, px * 3);
    up.want(angles, px * 8);
    FS_HIP(up.begin());
    int64_t *out_stage = out_locked ? out : static_cast<int64_t *>(up.block(px * 8));
    FS_HIP(ctx->depth16.ensure((size_t)ctx->max_batch * N * 2));
    FS_HIP(ctx->rgb8.ensure((size_t)ctx->max_batch * N * 3));
    // the int64 labels go up as they are (surfaces64 holds them until it receives the result) and are narrowed,
    // range-checked and maximised on the device: the counts come back with the tile scan's in one read-back
    FS_HIP(hipMemcpyAsync(ctx->surfaces64.p, up.from(labels, px * 8), px * 8, hipMemcpyHostToDevice, ctx->stream));
    FS_HIP(hipMemcpyAsync(ctx->depth16.p, up.from(depth, px * 2), px * 2, hipMemcpyHostToDevice, ctx->stream));
    FS_HIP(hipMemcpyAsync(ctx->rgb8.p, up.from(rgb, px * 3), px * 3, hipMemcpyHostToDevice, ctx->stream));
    FS_HIP(hipMemcpyAsync(ctx->angles.p, up.from(angles, px * 8), px * 8, hipMemcpyHostToDevice, ctx->stream));
    FS_HIP(ctx->apply_stats.ensure(3 * (size_t)ctx->max_batch * sizeof(int32_t)));
    FS_HIP(ctx->apply_stats_host.ensure(3 * (size_t)ctx->max_batch * sizeof(int32_t)));
    int32_t *d_stats = ctx->apply_stats.as<int32_t>();      // [n] largest label | [n] out of range | [n] most listed labels of a tile
    int32_t *maxa = d_stats + 2 * n;
    FS_HIP(hipMemsetAsync(d_stats, 0, 3 * n * sizeof(int32_t), ctx->stream));
    launch_crf_labels_to_int32(ctx->surfaces64.as<int64_t>(), ctx->patches32.as<int32_t>(), d_stats, n, (int)N, kMaxLabels - 2, ctx->stream);
    if (const int th = crf_fused_tile_height(ctx->r))
        launch_crf_tile_active(ctx->patches32.as<int32_t>(), n, ctx->H, ctx->W, 32, th, ctx->r, ctx->tile_recs.as<int32_t>(), maxa, ctx->stream);
    FS_HIP(hipMemcpyAsync(ctx->apply_stats_host.p, d_stats, 3 * n * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    FS_HIP(hipStreamSynchronize(ctx->stream));
    const int32_t *hs = ctx->apply_stats_host.as<int32_t>();
    std::vector<int> S(n), ma(n);
    for (int i = 0; i < n; ++i) {
        if (hs[n + i]) return ctx->fail("fs_apply_crf: a label of frame " + std::to_string(i) + " lies outside [0, " + std::to_string(kMaxLabels - 2) + "]");
        S[i] = hs[i] + 2;     // val = max + 1 surfaces, + the "unassigned" channel (post_processing.py:10-11)
        ma[i] = hs[2 * n + i];
    }
    if (run_crf_batch(ctx, n, ctx->depth16.as<uint16_t>(), ctx->rgb8.as<uint8_t>(), ctx->angles.as<float>(), ctx->patches32.as<int32_t>(),
                      S.data(), ma.data(), 1, ctx->surfaces64.as<int64_t>()))
        return -1;
    FS_HIP(hipMemcpyAsync(out_stage, ctx->surfaces64.p, px * 8, hipMemcpyDeviceToHost, ctx->stream));
    FS_HIP(hipStreamSynchronize(ctx->stream));
    if (!out_locked) std::memcpy(out, out_stage, px * 8);
    return 0;
}

namespace {

constexpr int kStatsOutputs = 15;

// bytes of output k for a frame of N pixels with S surfaces and T texture components
void stats_output_sizes(size_t N, size_t S, size_t T, size_t sizes[kStatsOutputs])
{
    const size_t v[kStatsOutputs] = {N * 8, (S + 1) * 4, N * 8, S * 16, S * 8, S * 12, S * 8, 65536 * S * 8, 1600 * S * 8, S * T * 8, S * 16,
                                     S * S * 8, S * S * 16, N * 4, S * 32};
    for (int k = 0; k < kStatsOutputs; ++k) sizes[k] = v[k];
}

// Outputs of at most this many bytes travel to a host caller together (one copy into page-locked memory, then a
// memcpy each): a copy command of their own costs each of them 30-70 us whatever its size, longer than any kernel here.
bool stats_small_output(size_t bytes) { return bytes <= ((size_t)512 << 10); }

struct StatsInputs {                 // device pointers, frames back to back
    const int64_t *surfaces, *patches;
    const int32_t *edges;
    const float *points_3d, *angles;
    const double *lab;               // or null when rgb is given
    const uint8_t *rgb;
    const float *texture;            // or null
    double *lab_out;                 // or null: where to leave the converted colours
};

// Shared body of the statistics entry points: n frames, frame i with S[i] surfaces.  `in` and the non-null entries of
// user_out are device pointers (arrays of all frames back to back, frame i's part sized by S[i]); outputs the caller
// did not ask for go to the context's staging buffer.  od[k] receives the device address of output k, out_bytes[k] its
// total size.  Synchronises once to read the error flag.
// after_stage: see launch_surface_stats_batch; od and out_bytes are filled in before the first call of it.
int surface_stats_core(fs_ctx *ctx, const char *who, int n, const int32_t *S, int T, const StatsInputs &in, void *const user_out[kStatsOutputs],
                       char *od[kStatsOutputs], size_t out_bytes[kStatsOutputs], const StatsStageHook &after_stage = StatsStageHook())
{
    const int H = ctx->H, W = ctx->W;
    const size_t N = ctx->N();
    const int nblk = stats_blocks_per_frame(H, W);
    auto up = [](size_t b) { return (b + 255) / 256 * 256; };
    // sizes
    std::vector<size_t> off((size_t)kStatsOutputs * (n + 1), 0);           // off[k * (n+1) + i]: byte offset of frame i in output k
    StatsScratchSizes tot{0, 0, 0, 0};
    std::vector<StatsScratchSizes> ss(n);
    for (int i = 0; i < n; ++i) {
        size_t sz[kStatsOutputs];
        stats_output_sizes(N, (size_t)S[i], (size_t)T, sz);
        for (int k = 0; k < kStatsOutputs; ++k) off[(size_t)k * (n + 1) + i + 1] = off[(size_t)k * (n + 1) + i] + sz[k];
        ss[i] = stats_scratch_sizes(S[i], T, nblk);
        tot.zero += ss[i].zero; tot.ones += ss[i].ones; tot.ext += ss[i].ext; tot.plain += ss[i].plain;
    }
    size_t staged = 256;                                                    // the error flag
    for (int k = 0; k < kStatsOutputs; ++k) {
        out_bytes[k] = off[(size_t)k * (n + 1) + n];
        if (!user_out[k]) staged += up(out_bytes[k]);
    }
    FS_HIP(ctx->st_out.ensure(staged));
    FS_HIP(ctx->st_work.ensure(4 * up((size_t)n * N * 4)));
    FS_HIP(ctx->st_scratch.ensure(up(tot.zero) + up(tot.ones) + up(tot.ext) + up(tot.plain)));
    FS_HIP(ctx->st_descs.ensure((size_t)n * sizeof(StatsFrame)));
    if (in.rgb) {
        if (!ctx->st_lin.p) {
            // sRGB byte -> linear light, the first step of the colour conversion (utils.rgb_to_Lab)
            double lin[256];
            for (int v = 0; v < 256; ++v) {
                const double c = (double)v / 255.0;
                lin[v] = c > 0.04045 ? std::pow((c + 0.055) / 1.055, 2.4) : c / 12.92;
            }
            float lin32[256];                                  // the same values in float32 behind them
            for (int v = 0; v < 256; ++v) lin32[v] = (float)lin[v];
            FS_HIP(ctx->st_lin.ensure(sizeof(lin) + sizeof(lin32)));
            FS_HIP(hipMemcpy(ctx->st_lin.p, lin, sizeof(lin), hipMemcpyHostToDevice));
            FS_HIP(hipMemcpy((char *)ctx->st_lin.p + sizeof(lin), lin32, sizeof(lin32), hipMemcpyHostToDevice));
        }
    }
    char *p = ctx->st_out.as<char>();
    int *d_err = reinterpret_cast<int *>(p);
    p += 256;
    // staged outputs: the small ones first and next to each other, so that a caller on the host can fetch them all with
    // one copy (stats_small_output)
    for (int pass = 0; pass < 2; ++pass)
        for (int k = 0; k < kStatsOutputs; ++k) {
            if (user_out[k]) { od[k] = static_cast<char *>(user_out[k]); continue; }
            if (stats_small_output(out_bytes[k]) == (pass == 0)) { od[k] = p; p += up(out_bytes[k]); }
        }
    StatsBatchWork w{};
    {
        char *q = ctx->st_work.as<char>();
        void **slots[4] = {(void **)&w.surf32, (void **)&w.pat32, (void **)&w.parent, (void **)&w.count};
        for (auto *sl : slots) { *sl = q; q += up((size_t)n * N * 4); }
    }
    char *zero = ctx->st_scratch.as<char>(), *ones = zero + up(tot.zero), *ext = ones + up(tot.ones), *plain = ext + up(tot.ext);
    w.lin = ctx->st_lin.as<double>();
    w.lin32 = reinterpret_cast<const float *>(w.lin + 256);
    w.ext_all = reinterpret_cast<uint32_t *>(ext);
    w.ext_words = tot.ext / 4;
    w.err = d_err;
    if (ctx->debug_fill.find('w') != std::string::npos) {
        // the statistics stage's work arrays and the part of its scratch that no memset prepares, as 32-bit ones
        FS_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(ctx->st_work.p), 1, ctx->st_work.cap / 4, ctx->stream));
        FS_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(plain), 1, tot.plain / 4, ctx->stream));
    }
    FS_HIP(hipMemsetAsync(zero, 0, tot.zero, ctx->stream));
    FS_HIP(hipMemsetAsync(ones, 0xff, tot.ones, ctx->stream));
    FS_HIP(hipMemsetAsync(d_err, 0, 4, ctx->stream));
    std::vector<StatsFrame> &fh = ctx->st_descs_host;
    fh.assign(n, StatsFrame{});
    // colours given as rgb: the conversion writes the Lab image where the caller wants it, or, when nobody wants it, only the
    // pixels' histogram bins (into the work array the component pass no longer uses for sizes)
    double *lab_work = in.rgb ? in.lab_out : nullptr;
    uint32_t *color_bins = in.rgb && !in.lab_out ? reinterpret_cast<uint32_t *>(w.count) : nullptr;
    for (int i = 0; i < n; ++i) {
        StatsFrame &f = fh[i];
        f.S = S[i];
        f.surfaces = in.surfaces + (size_t)i * N;
        f.patches = in.patches + (size_t)i * N;
        f.edges = in.edges + (size_t)i * N;
        f.points_3d = in.points_3d + (size_t)i * N * 3;
        f.angles = in.angles + (size_t)i * N * 2;
        f.rgb = in.rgb ? in.rgb + (size_t)i * N * 3 : nullptr;
        f.lab_work = lab_work ? lab_work + (size_t)i * N * 3 : nullptr;
        f.color_bin = color_bins ? color_bins + (size_t)i * N : nullptr;
        f.lab = in.rgb ? f.lab_work : in.lab + (size_t)i * N * 3;
        f.texture = in.texture ? in.texture + (size_t)i * (N / 16) * T : nullptr;
        stats_carve_scratch(f, T, nblk, zero, ones, ext, plain);
        zero += ss[i].zero; ones += ss[i].ones; ext += ss[i].ext; plain += ss[i].plain;
        auto at = [&](int k) { return od[k] + off[(size_t)k * (n + 1) + i]; };
        f.surfaces_out = reinterpret_cast<int64_t *>(at(0));
        f.offsets = reinterpret_cast<int32_t *>(at(1));
        f.patch_points = reinterpret_cast<float *>(at(2));
        f.avg_pos = reinterpret_cast<double *>(at(3));
        f.counts = reinterpret_cast<double *>(at(4));
        f.centroids = reinterpret_cast<float *>(at(5));
        f.centroid_idx = reinterpret_cast<int32_t *>(at(6));
        f.hist_color_out = reinterpret_cast<double *>(at(7));
        f.hist_angle_out = reinterpret_cast<double *>(at(8));
        f.hist_texture_out = reinterpret_cast<double *>(at(9));
        f.avg_angles = reinterpret_cast<double *>(at(10));
        f.neighbors = reinterpret_cast<double *>(at(11));
        f.border_centers = reinterpret_cast<double *>(at(12));
        f.norm = reinterpret_cast<float *>(at(13));
        f.depth_extend = reinterpret_cast<float *>(at(14));
    }
    FS_HIP(hipMemcpyAsync(ctx->st_descs.p, fh.data(), (size_t)n * sizeof(StatsFrame), hipMemcpyHostToDevice, ctx->stream));
    hipError_t e = launch_surface_stats_batch(ctx->st_descs.as<StatsFrame>(), fh.data(), n, w, H, W, T, ctx->stream, after_stage);
    if (e != hipSuccess) return ctx->fail(e, (std::string(who) + ": launch").c_str());
    int err = 0;
    FS_HIP(hipMemcpyAsync(&err, d_err, 4, hipMemcpyDeviceToHost, ctx->stream));
    FS_HIP(hipStreamSynchronize(ctx->stream));
    if (err == 1) return ctx->fail(std::string(who) + ": a label lies outside [0, num_surfaces)");
    if (err == 2) return ctx->fail(std::string(who) + ": Lab a/b + 127 rounds outside [0, 255]");
    if (err == 3) return ctx->fail(std::string(who) + ": angle outside [-pi, pi]");
    return 0;
}

int stats_check_args(fs_ctx *ctx, const char *who, int n, const int32_t *S, int T, bool has_texture, bool inputs_ok)
{
    if (n < 1 || !S) return ctx->fail(std::string(who) + ": need at least one frame and its num_surfaces");
    for (int i = 0; i < n; ++i)
        if (S[i] < 1 || S[i] > kMaxLabels) return ctx->fail(std::string(who) + ": num_surfaces outside [1, " + std::to_string(kMaxLabels) + "]");
    if (T < 1 || T > 4096) return ctx->fail(std::string(who) + ": texture_dim outside [1, 4096]");
    if (has_texture && (ctx->H % 4 || ctx->W % 4)) return ctx->fail(std::string(who) + ": texture input needs frame sides divisible by 4");
    if (!inputs_ok) return ctx->fail(std::string(who) + ": null input");
    return 0;
}

}  // namespace

namespace {

// host-pointer statistics of one frame: colours as lab (float64) or as rgb (uint8, converted on the device)
int surface_stats_host(fs_ctx *ctx, const char *who, int num_surfaces, const int64_t *surfaces, const int64_t *patches,
                       const int32_t *depth_edges, const float *points_3d, const float *angles, const double *lab, const uint8_t *rgb,
                       const float *texture, int texture_dim, double *lab_out, void *const host_out[kStatsOutputs])
{
    if (!ctx) return -1;
    if (set_device(ctx)) return -1;
    const int32_t S = num_surfaces;
    const int T = texture_dim;
    if (stats_check_args(ctx, who, 1, &S, T, texture != nullptr, surfaces && patches && depth_edges && points_3d && angles && (lab || rgb)))
        return -1;
    const size_t N = ctx->N();
    auto up8 = [](size_t b) { return (b + 255) / 256 * 256; };
    const size_t in_sizes[8] = {N * 8, N * 8, N * 4, N * 12, N * 8, lab ? N * 24 : 0, lab ? 0 : N * 3, texture ? N / 16 * T * 4 : 0};
    const void *in_host[8] = {surfaces, patches, depth_edges, points_3d, angles, lab, rgb, texture};
    size_t in_total = 0;
    for (size_t b : in_sizes) in_total += up8(b);
    if (!lab && lab_out) in_total += up8(N * 24);
    FS_HIP(ctx->st_in.ensure(in_total));
    char *in_dev[8];
    char *p = ctx->st_in.as<char>();
    for (int i = 0; i < 8; ++i) {
        in_dev[i] = p;
        p += up8(in_sizes[i]);
    }
    // The inputs go up on the upload stream in the order the kernels want them (colours for the conversion, the two
    // label images for the relabelling, the points for the centroids, the rest for the accumulation pass), and the
    // kernels wait for their group only: most of the 13 MB travel beside the first half of the kernel chain.
    // (The previous call ended with both streams drained: nothing still reads these buffers.)
    for (hipEvent_t &e : ctx->stats_up_events)
        if (!e) FS_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    // An array in ordinary (pageable) memory is first copied into page-locked memory here: the runtime's own path for
    // such a source held the calling thread for 340 us per 0.9 MB colour image.
    size_t pageable = 0;
    bool locked[8];
    for (int i = 0; i < 8; ++i) {
        locked[i] = !in_sizes[i] || host_memory_is_page_locked(in_host[i]);
        if (!locked[i]) pageable += up8(in_sizes[i]);
    }
    FS_HIP(ctx->stats_in_host.ensure(pageable));
    char *stage = ctx->stats_in_host.as<char>();
    const int up_order[8] = {6, 0, 1, 3, 2, 4, 5, 7}, up_group[8] = {0, 1, 1, 2, 3, 3, 3, 3};
    for (int k = 0; k < 8; ++k) {
        const int i = up_order[k];
        const void *src = in_host[i];
        if (!locked[i]) {
            std::memcpy(stage, in_host[i], in_sizes[i]);
            src = stage;
            stage += up8(in_sizes[i]);
        }
        if (in_sizes[i]) FS_HIP(hipMemcpyAsync(in_dev[i], src, in_sizes[i], hipMemcpyHostToDevice, ctx->in_stream));
        if (k == 7 || up_group[k + 1] != up_group[k]) FS_HIP(hipEventRecord(ctx->stats_up_events[up_group[k]], ctx->in_stream));
    }
    StatsInputs in{};
    in.surfaces = reinterpret_cast<const int64_t *>(in_dev[0]);
    in.patches = reinterpret_cast<const int64_t *>(in_dev[1]);
    in.edges = reinterpret_cast<const int32_t *>(in_dev[2]);
    in.points_3d = reinterpret_cast<const float *>(in_dev[3]);
    in.angles = reinterpret_cast<const float *>(in_dev[4]);
    in.lab = lab ? reinterpret_cast<const double *>(in_dev[5]) : nullptr;
    in.rgb = lab ? nullptr : reinterpret_cast<const uint8_t *>(in_dev[6]);
    in.texture = texture ? reinterpret_cast<const float *>(in_dev[7]) : nullptr;
    in.lab_out = (!lab && lab_out) ? reinterpret_cast<double *>(p) : nullptr;
    void *const none[kStatsOutputs] = {};
    char *od[kStatsOutputs];
    size_t out_sizes[kStatsOutputs];
    // Each group of outputs starts its way to the host when the kernels that complete it have been enqueued (the
    // converted colours after the first kernel, the images and point lists after the component pass, the histograms
    // at the end), on the download stream beside the kernels that follow; the downloads of a frame's 27 MB otherwise
    // take longer than all its kernels.
    const int stage_of[kStatsOutputs] = {1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2};
    hipError_t hook_error = hipSuccess;
    for (hipEvent_t &e : ctx->stats_events)
        if (!e) FS_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    size_t small_hi = 0;                                    // extent of the small outputs in the staging buffer (device), bytes from small_base
    char *small_base = nullptr;
    auto hook = [&](int stage) {
        auto ok = [&](hipError_t e) { if (e != hipSuccess && hook_error == hipSuccess) hook_error = e; return e == hipSuccess; };
        if (stage >= 10) {                                  // a group of inputs is about to be read
            ok(hipStreamWaitEvent(ctx->stream, ctx->stats_up_events[stage - 10], 0));
            return;
        }
        if (!ok(hipEventRecord(ctx->stats_events[stage], ctx->stream)) || !ok(hipStreamWaitEvent(ctx->out_stream, ctx->stats_events[stage], 0))) return;
        if (stage == 0 && in.lab_out) ok(hipMemcpyAsync(lab_out, in.lab_out, N * 24, hipMemcpyDeviceToHost, ctx->out_stream));
        for (int i = 0; i < kStatsOutputs; ++i)
            if (stage_of[i] == stage && host_out[i] && !stats_small_output(out_sizes[i]))
                ok(hipMemcpyAsync(host_out[i], od[i], out_sizes[i], hipMemcpyDeviceToHost, ctx->out_stream));
        if (stage != 2) return;
        // the small outputs lie next to each other at the start of the staging buffer: one copy for all of them
        for (int i = 0; i < kStatsOutputs; ++i) {
            if (!stats_small_output(out_sizes[i])) continue;
            if (!small_base || od[i] < small_base) small_base = od[i];
        }
        for (int i = 0; i < kStatsOutputs; ++i)
            if (stats_small_output(out_sizes[i])) small_hi = std::max(small_hi, (size_t)(od[i] - small_base) + out_sizes[i]);
        if (small_base && ok(ctx->stats_small_host.ensure(small_hi)))
            ok(hipMemcpyAsync(ctx->stats_small_host.p, small_base, small_hi, hipMemcpyDeviceToHost, ctx->out_stream));
    };
    if (surface_stats_core(ctx, who, 1, &S, T, in, none, od, out_sizes, hook)) {
        (void)hipStreamSynchronize(ctx->in_stream);        // nothing of this call stays in flight from or towards the caller's arrays
        (void)hipStreamSynchronize(ctx->out_stream);
        return -1;
    }
    if (hook_error != hipSuccess) {
        (void)hipStreamSynchronize(ctx->in_stream);
        (void)hipStreamSynchronize(ctx->out_stream);
        return ctx->fail(hook_error, "statistics downloads");
    }
    FS_HIP(hipStreamSynchronize(ctx->out_stream));
    for (int i = 0; i < kStatsOutputs; ++i)
        if (host_out[i] && stats_small_output(out_sizes[i]))
            std::memcpy(host_out[i], ctx->stats_small_host.as<char>() + (od[i] - small_base), out_sizes[i]);
    return 0;
}

}  // namespace

int fs_surface_stats(fs_ctx *ctx, int num_surfaces, const int64_t *surfaces, const int64_t *patches, const int32_t *depth_edges,
                     const float *points_3d, const float *angles, const double *lab, const float *texture, int texture_dim,
                     int64_t *surfaces_out, int32_t *patch_point_offsets, float *patch_points, double *avg_pos, double *counts,
                     float *centroids, int32_t *centroid_indices, double *hist_color, double *hist_angle, double *hist_texture,
                     double *avg_angles, double *neighbors, double *border_centers, float *norm, float *depth_extend)
{
    void *const host_out[kStatsOutputs] = {surfaces_out, patch_point_offsets, patch_points, avg_pos, counts, centroids, centroid_indices,
                                           hist_color, hist_angle, hist_texture, avg_angles, neighbors, border_centers, norm, depth_extend};
    if (ctx && !lab) return ctx->fail("fs_surface_stats: null input");
    return surface_stats_host(ctx, "fs_surface_stats", num_surfaces, surfaces, patches, depth_edges, points_3d, angles, lab, nullptr, texture,
                              texture_dim, nullptr, host_out);
}

int fs_surface_stats_rgb(fs_ctx *ctx, int num_surfaces, const int64_t *surfaces, const int64_t *patches, const int32_t *depth_edges,
                         const float *points_3d, const float *angles, const uint8_t *rgb, const float *texture, int texture_dim,
                         double *lab_out, int64_t *surfaces_out, int32_t *patch_point_offsets, float *patch_points, double *avg_pos,
                         double *counts, float *centroids, int32_t *centroid_indices, double *hist_color, double *hist_angle,
                         double *hist_texture, double *avg_angles, double *neighbors, double *border_centers, float *norm,
                         float *depth_extend)
{
    void *const host_out[kStatsOutputs] = {surfaces_out, patch_point_offsets, patch_points, avg_pos, counts, centroids, centroid_indices,
                                           hist_color, hist_angle, hist_texture, avg_angles, neighbors, border_centers, norm, depth_extend};
    if (ctx && !rgb) return ctx->fail("fs_surface_stats_rgb: null input");
    return surface_stats_host(ctx, "fs_surface_stats_rgb", num_surfaces, surfaces, patches, depth_edges, points_3d, angles, nullptr, rgb,
                              texture, texture_dim, lab_out, host_out);
}

int fs_surface_stats_dev(fs_ctx *ctx, int num_surfaces, const int64_t *d_surfaces, const int64_t *d_patches,
                         const int32_t *d_depth_edges, const float *d_points_3d, const float *d_angles, const double *d_lab,
                         const float *d_texture, int texture_dim, int64_t *d_surfaces_out, int32_t *d_patch_point_offsets,
                         float *d_patch_points, double *d_avg_pos, double *d_counts, float *d_centroids, int32_t *d_centroid_indices,
                         double *d_hist_color, double *d_hist_angle, double *d_hist_texture, double *d_avg_angles,
                         double *d_neighbors, double *d_border_centers, float *d_norm, float *d_depth_extend)
{
    const int32_t S = num_surfaces;
    return fs_surface_stats_batch_dev(ctx, 1, &S, d_surfaces, d_patches, d_depth_edges, d_points_3d, d_angles, d_lab, nullptr, d_texture,
                                      texture_dim, nullptr, d_surfaces_out, d_patch_point_offsets, d_patch_points, d_avg_pos, d_counts,
                                      d_centroids, d_centroid_indices, d_hist_color, d_hist_angle, d_hist_texture, d_avg_angles, d_neighbors,
                                      d_border_centers, d_norm, d_depth_extend);
}

int fs_surface_stats_batch_dev(fs_ctx *ctx, int n, const int32_t *num_surfaces, const int64_t *d_surfaces, const int64_t *d_patches,
                               const int32_t *d_depth_edges, const float *d_points_3d, const float *d_angles, const double *d_lab,
                               const uint8_t *d_rgb, const float *d_texture, int texture_dim, double *d_lab_out, int64_t *d_surfaces_out,
                               int32_t *d_patch_point_offsets, float *d_patch_points, double *d_avg_pos, double *d_counts,
                               float *d_centroids, int32_t *d_centroid_indices, double *d_hist_color, double *d_hist_angle,
                               double *d_hist_texture, double *d_avg_angles, double *d_neighbors, double *d_border_centers, float *d_norm,
                               float *d_depth_extend)
{
    if (!ctx) return -1;
    if (set_device(ctx)) return -1;
    const int T = texture_dim;
    if (stats_check_args(ctx, "fs_surface_stats_batch_dev", n, num_surfaces, T, d_texture != nullptr,
                         d_surfaces && d_patches && d_depth_edges && d_points_3d && d_angles && (d_lab || d_rgb)))
        return -1;
    StatsInputs in{};
    in.surfaces = d_surfaces;
    in.patches = d_patches;
    in.edges = d_depth_edges;
    in.points_3d = d_points_3d;
    in.angles = d_angles;
    in.lab = d_lab;
    in.rgb = d_lab ? nullptr : d_rgb;
    in.texture = d_texture;
    in.lab_out = d_lab_out;
    void *const user_out[kStatsOutputs] = {d_surfaces_out, d_patch_point_offsets, d_patch_points, d_avg_pos, d_counts, d_centroids,
                                           d_centroid_indices, d_hist_color, d_hist_angle, d_hist_texture, d_avg_angles, d_neighbors,
                                           d_border_centers, d_norm, d_depth_extend};
    char *od[kStatsOutputs];
    size_t out_sizes[kStatsOutputs];
    return surface_stats_core(ctx, "fs_surface_stats_batch_dev", n, num_surfaces, T, in, user_out, od, out_sizes) ? -1 : 0;
}

int fs_chi_squared_distances(fs_ctx *ctx, const double *hist, int A, int B, int S, int pool_h, int pool_w, int stride_h, int stride_w,
                             float *out)
{
    if (!ctx) return -1;
    if (set_device(ctx)) return -1;
    if (!hist || !out || A < 1 || B < 1 || S < 1 || pool_h < 1 || pool_w < 1 || stride_h < 1 || stride_w < 1 || pool_h > A || pool_w > B)
        return ctx->fail("fs_chi_squared_distances: bad shape");
    if (S > 65535) return ctx->fail("fs_chi_squared_distances: too many surfaces");
    const int PA = (A - pool_h) / stride_h + 1, PB = (B - pool_w) / stride_w + 1;
    const size_t in_bytes = (size_t)A * B * S * 8, pooled_bytes = (size_t)S * PA * PB * 4, out_bytes = (size_t)S * S * 4;
    FS_HIP(ctx->st_in.ensure(in_bytes));
    FS_HIP(ctx->st_scratch.ensure(pooled_bytes));
    FS_HIP(ctx->st_out.ensure(out_bytes));
    StagedSources up(ctx->stats_in_host);
    up.want(hist, in_bytes);
    FS_HIP(up.begin());
    FS_HIP(hipMemcpyAsync(ctx->st_in.p, up.from(hist, in_bytes), in_bytes, hipMemcpyHostToDevice, ctx->stream));
    launch_chi_squared(ctx->st_in.as<double>(), ctx->st_scratch.as<float>(), ctx->st_out.as<float>(), A, B, S, pool_h, pool_w, stride_h,
                       stride_w, ctx->stream);
    FS_HIP(hipGetLastError());
    FS_HIP(hipMemcpyAsync(out, ctx->st_out.p, out_bytes, hipMemcpyDeviceToHost, ctx->stream));
    FS_HIP(hipStreamSynchronize(ctx->stream));
    return 0;
}

int fs_chi_squared_distances_dev(fs_ctx *ctx, const double *d_hist, int A, int B, int S, int pool_h, int pool_w, int stride_h,
                                 int stride_w, float *d_out)
{
    if (!ctx) return -1;
    if (set_device(ctx)) return -1;
    if (!d_hist || !d_out || A < 1 || B < 1 || S < 1 || pool_h < 1 || pool_w < 1 || stride_h < 1 || stride_w < 1 || pool_h > A ||
        pool_w > B)
        return ctx->fail("fs_chi_squared_distances_dev: bad shape");
    if (S > 65535) return ctx->fail("fs_chi_squared_distances_dev: too many surfaces");
    const int PA = (A - pool_h) / stride_h + 1, PB = (B - pool_w) / stride_w + 1;
    // a buffer of its own: st_scratch may still hold what an earlier call on this stream left for a later one
    FS_HIP(ctx->st_pool.ensure((size_t)S * PA * PB * 4));
    launch_chi_squared(d_hist, ctx->st_pool.as<float>(), d_out, A, B, S, pool_h, pool_w, stride_h, stride_w, ctx->stream);
    FS_HIP(hipGetLastError());
    return 0;
}

int fs_overlap_counts(fs_ctx *ctx, const int64_t *annotation, const int64_t *detected, int num_annotations, int num_detections,
                      double *overlap)
{
    if (!ctx) return -1;
    if (set_device(ctx)) return -1;
    if (!annotation || !detected || !overlap || num_annotations < 1 || num_detections < 1)
        return ctx->fail("fs_overlap_counts: bad arguments");
    const size_t N = ctx->N(), cells = (size_t)num_annotations * num_detections;
    FS_HIP(ctx->st_in.ensure(2 * N * 8));
    FS_HIP(ctx->st_out.ensure(cells * 4 + 8));
    int64_t *d_a = ctx->st_in.as<int64_t>(), *d_o = d_a + N;
    unsigned *d_ov = ctx->st_out.as<unsigned>();
    int *d_err = reinterpret_cast<int *>(d_ov + cells);
    StagedSources up(ctx->stats_in_host);
    up.want(annotation, N * 8);
    up.want(detected, N * 8);
    FS_HIP(up.begin());
    FS_HIP(hipMemcpyAsync(d_a, up.from(annotation, N * 8), N * 8, hipMemcpyHostToDevice, ctx->stream));
    FS_HIP(hipMemcpyAsync(d_o, up.from(detected, N * 8), N * 8, hipMemcpyHostToDevice, ctx->stream));
    FS_HIP(hipMemsetAsync(d_ov, 0, cells * 4 + 8, ctx->stream));
    launch_overlap_counts(d_a, d_o, d_ov, d_err, (int)N, num_detections, num_annotations, ctx->stream);
    FS_HIP(hipGetLastError());
    std::vector<unsigned> host(cells + 1);
    FS_HIP(hipMemcpyAsync(host.data(), d_ov, cells * 4 + 4, hipMemcpyDeviceToHost, ctx->stream));
    FS_HIP(hipStreamSynchronize(ctx->stream));
    if (host[cells]) return ctx->fail("fs_overlap_counts: a label lies outside the given counts");
    for (size_t i = 0; i < cells; ++i) overlap[i] = (double)host[i];
    return 0;
}

int fs_nearest_points(fs_ctx *ctx, int groups, const float *points, const int32_t *point_offsets, const float *targets,
                      const int32_t *target_offsets, int32_t *out)
{
    if (!ctx) return -1;
    if (set_device(ctx)) return -1;
    if (groups < 1 || !points || !point_offsets || !targets || !target_offsets || !out) return ctx->fail("fs_nearest_points: bad arguments");
    const int P = point_offsets[groups], T = target_offsets[groups];
    if (point_offsets[0] != 0 || target_offsets[0] != 0 || P < 0 || T < 0) return ctx->fail("fs_nearest_points: offsets must start at 0");
    std::vector<int32_t> group_of(T);
    for (int g = 0; g < groups; ++g) {
        if (point_offsets[g + 1] < point_offsets[g] || target_offsets[g + 1] < target_offsets[g])
            return ctx->fail("fs_nearest_points: offsets must not decrease");
        if (target_offsets[g + 1] > target_offsets[g] && point_offsets[g + 1] == point_offsets[g])
            return ctx->fail("fs_nearest_points: group " + std::to_string(g) + " has targets but no points");
        for (int t = target_offsets[g]; t < target_offsets[g + 1]; ++t) group_of[t] = g;
    }
    if (T == 0) return 0;
    auto up8 = [](size_t b) { return (b + 255) / 256 * 256; };
    const size_t bp = up8((size_t)P * 8), bo = up8((size_t)(groups + 1) * 4), bt = up8((size_t)T * 8), bg = up8((size_t)T * 4);
    FS_HIP(ctx->st_in.ensure(bp + bo + bt + bg));
    FS_HIP(ctx->st_out.ensure((size_t)T * 8));
    char *base = ctx->st_in.as<char>();
    FS_HIP(hipMemcpyAsync(base, points, (size_t)P * 8, hipMemcpyHostToDevice, ctx->stream));
    FS_HIP(hipMemcpyAsync(base + bp, point_offsets, (size_t)(groups + 1) * 4, hipMemcpyHostToDevice, ctx->stream));
    FS_HIP(hipMemcpyAsync(base + bp + bo, targets, (size_t)T * 8, hipMemcpyHostToDevice, ctx->stream));
    FS_HIP(hipMemcpyAsync(base + bp + bo + bt, group_of.data(), (size_t)T * 4, hipMemcpyHostToDevice, ctx->stream));
    launch_nearest_points(reinterpret_cast<const float *>(base), reinterpret_cast<const int32_t *>(base + bp),
                          reinterpret_cast<const float *>(base + bp + bo), reinterpret_cast<const int32_t *>(base + bp + bo + bt), T,
                          ctx->st_out.as<int32_t>(), ctx->stream);
    FS_HIP(hipGetLastError());
    FS_HIP(hipMemcpyAsync(out, ctx->st_out.p, (size_t)T * 8, hipMemcpyDeviceToHost, ctx->stream));
    FS_HIP(hipStreamSynchronize(ctx->stream));
    return 0;
}

}  // extern "C"
