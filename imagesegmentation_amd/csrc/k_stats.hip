// Per-surface statistics of a labelled frame (gfx950): the pixel-level loops of
// extract_information_from_surface_data_and_preprocess_surfaces (process_surfaces.py:672-687)
//   extract_individual_points_from_surfaces            :96-108   stable sort of pixel indices by patch label
//   relabel_surfaces                                   :236-242
//   determine_pixel_counts_and_average_position_...    :111-129  integer sums, exact
//   determine_surface_patch_centroids                  :81-93    first minimum of an fp64 distance
//   remove_disconnected_components                     :8-28     label components instead of flood fills
//   determine_histograms_and_average_normals           :245-297
//   determine_neighboring_surfaces_and_border_centers  :31-68
//   calculate_depth_extend                             :332-346
// All of it is HBM-bound integer / byte work: one pass per dependency level, per-block
// privatisation in LDS of everything indexed by label, one global atomic per (block, label).
// Every kernel works on a batch: blockIdx.y (blockIdx.x for the one-block-per-frame kernels) is the frame,
// frames[] holds its label count, its label-indexed scratch and its output pointers; the int32 working copies of
// the label images are [n][H*W] arrays.
#include <algorithm>

#include "fs_common.h"

namespace fs {

namespace {

constexpr int SB = 256;            // threads per block
constexpr int SPB = 1024;          // pixels per block (4 per thread, interleaved for coalescing)
constexpr int kWinSlots = 8;       // labels per block of the accumulation pass whose colour counts are gathered in LDS

__device__ __forceinline__ unsigned enc_f32(float f)      // order-preserving float -> unsigned
{
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float dec_f32(unsigned u)
{
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

// hist[idx] += 1 for every active lane, one atomic per distinct idx of the wave: the pixels of a
// planar, evenly coloured surface fall into very few bins, and thousands of atomics on one address
// would serialise in L2.  Every lane of the wave must call this (inactive ones with active = false).
__device__ __forceinline__ void wave_histogram_add(uint32_t *__restrict__ hist, uint32_t idx, bool active)
{
    unsigned long long todo = __ballot(active);
    const int lane = threadIdx.x & 63;
    // (rounds bounded at three, the remaining lanes adding their own counts, measured a third slower on the OSD frames:
    // the atomics, not the rounds, are what the pass waits for)
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const uint32_t key = __shfl(idx, leader);
        const unsigned long long same = __ballot(active && idx == key) & todo;
        if (lane == leader) atomicAdd(hist + key, (uint32_t)__popcll(same));
        todo &= ~same;
    }
}

// counts[key] += number of active lanes with that key, one LDS atomic per distinct key of the wave instead of one per
// lane (lanes of a wave mostly share their label, and atomics on one LDS address go through one after the other).
// Every lane of the wave must call this.
__device__ __forceinline__ void wave_count_add(uint32_t *counts, int key, bool active)
{
    unsigned long long todo = __ballot(active);
    while (todo) {
        const int k = __builtin_amdgcn_readlane(key, __ffsll((long long)todo) - 1);
        const unsigned long long same = __ballot(active && key == k) & todo;
        if ((threadIdx.x & 63) == 0) atomicAdd(&counts[k], (uint32_t)__popcll(same));
        todo &= ~same;
    }
}

// CIE L*a*b* of an sRGB pixel (D65, 2-degree observer): the definition utils.rgb_to_Lab restates for
// skimage.color.rgb2lab (utils.py:7-8 of the reference).  lin[v] = linearised channel value of byte v.
// When the Lab image itself is not an output, only what the colour histogram needs of it leaves the kernel: the
// pixel's bin round(a + 127) << 8 | round(b + 127) (process_surfaces.py:265-267), 4 bytes instead of 24 written and 16 read.
//
// The bins alone do not need float64: a and b in float32 are off by less than 3e-4 (the cube roots carry an absolute
// error below 2.5e-7 each -- 2^-24 relative in the table, three more roundings in the matrix product, a third of their sum
// through the root plus the root's own two units -- times 500 for a, 200 for b, plus 2e-5 of rounding in the difference
// and the offset), so a pixel whose a + 127 and b + 127 both lie further than kLabBand from a rounding boundary, and
// whose X, Y, Z stay clear of the switch between root and straight line, gets its bins from float32; the others (one
// in four hundred) go through the float64 evaluation.  All 2^24 colours are enumerated by the tests.
constexpr float kLabBand = 6e-4f;

__device__ __forceinline__ bool lab_bins_float(float r, float g, float b, int &bin_a, int &bin_b)
{
    const float x = fmaf(b, 0.180423f, fmaf(g, 0.357580f, r * 0.412453f)) / 0.95047f;
    const float y = fmaf(b, 0.072169f, fmaf(g, 0.715160f, r * 0.212671f));
    const float z = fmaf(b, 0.950227f, fmaf(g, 0.119193f, r * 0.019334f)) / 1.08883f;
    const float v[3] = {x, y, z};
    float t[3];
    bool clear = true;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        clear &= fabsf(v[c] - 0.008856f) > 1e-6f;
        t[c] = v[c] > 0.008856f ? cbrtf(v[c]) : fmaf(7.787f, v[c], 16.0f / 116.0f);
    }
    const float av = fmaf(500.0f, t[0] - t[1], 127.0f), bv = fmaf(200.0f, t[1] - t[2], 127.0f);
    const float fa = av - floorf(av), fb = bv - floorf(bv);
    clear &= fabsf(fa - 0.5f) > kLabBand && fabsf(fb - 0.5f) > kLabBand;
    bin_a = (int)rintf(av);
    bin_b = (int)rintf(bv);
    return clear;
}

__device__ __forceinline__ void lab_float64(double r, double g, double b, double &L, double &a_, double &b_)
{
    const double xyz[3] = {((r * 0.412453 + g * 0.357580) + b * 0.180423) / 0.95047, ((r * 0.212671 + g * 0.715160) + b * 0.072169) / 1.0,
                           ((r * 0.019334 + g * 0.119193) + b * 0.950227) / 1.08883};
    double t[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) t[c] = xyz[c] > 0.008856 ? cbrt(xyz[c]) : 7.787 * xyz[c] + 16.0 / 116.0;
    L = 116.0 * t[1] - 16.0;
    a_ = 500.0 * (t[0] - t[1]);
    b_ = 200.0 * (t[1] - t[2]);
}

// the Lab image of the frames that want one
__global__ __launch_bounds__(SB) void stats_lab_kernel(const StatsFrame *__restrict__ frames, const double *__restrict__ lin, int N)
{
    const StatsFrame &f = frames[blockIdx.y];
    const int p = blockIdx.x * SB + threadIdx.x;
    if (p >= N || !f.rgb || !f.lab_work) return;
    double *o = f.lab_work + 3 * (size_t)p;
    lab_float64(lin[f.rgb[3 * (size_t)p]], lin[f.rgb[3 * (size_t)p + 1]], lin[f.rgb[3 * (size_t)p + 2]], o[0], o[1], o[2]);
}

// the colour bins of the frames that do not: four pixels per lane (twelve colour bytes in three words when the image
// is aligned for that, one 16-byte store), the two tables in LDS
__global__ __launch_bounds__(SB) void stats_color_bins_kernel(const StatsFrame *__restrict__ frames, const double *__restrict__ lin,
                                                              const float *__restrict__ lin32, int N)
{
    __shared__ double s_lin[256];
    __shared__ float s_lin32[256];
    const StatsFrame &f = frames[blockIdx.y];
    if (!f.rgb || f.lab_work) return;
    static_assert(SB == 256, "one table entry per thread");
    s_lin[threadIdx.x] = lin[threadIdx.x];
    s_lin32[threadIdx.x] = lin32[threadIdx.x];
    __syncthreads();
    const int p0 = 4 * (blockIdx.x * SB + (int)threadIdx.x);
    if (p0 >= N) return;
    const uint8_t *__restrict__ rgb = f.rgb + 3 * (size_t)p0;
    uint8_t c[12];
    const bool whole = p0 + 4 <= N;
    if (whole && ((uintptr_t)rgb & 3u) == 0) {
        const uint32_t w0 = reinterpret_cast<const uint32_t *>(rgb)[0], w1 = reinterpret_cast<const uint32_t *>(rgb)[1],
                       w2 = reinterpret_cast<const uint32_t *>(rgb)[2];
#pragma unroll
        for (int k = 0; k < 4; ++k) { c[k] = (uint8_t)(w0 >> (8 * k)); c[4 + k] = (uint8_t)(w1 >> (8 * k)); c[8 + k] = (uint8_t)(w2 >> (8 * k)); }
    } else {
#pragma unroll
        for (int k = 0; k < 12; ++k) c[k] = p0 + k / 3 < N ? rgb[k] : 0;
    }
    uint32_t bins[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int a, b;
        if (!lab_bins_float(s_lin32[c[3 * k]], s_lin32[c[3 * k + 1]], s_lin32[c[3 * k + 2]], a, b)) {
            double L, a_, b_;
            lab_float64(s_lin[c[3 * k]], s_lin[c[3 * k + 1]], s_lin[c[3 * k + 2]], L, a_, b_);
            a = (int)rint(a_ + 127.0);
            b = (int)rint(b_ + 127.0);
        }
        // (an sRGB colour never leaves the range; the accumulation pass reports the flag if one does)
        bins[k] = a >= 0 && a <= 255 && b >= 0 && b <= 255 ? (uint32_t)(a * 256 + b) : 0xffffffffu;
    }
    uint32_t *out = f.color_bin + p0;
    if (whole && ((uintptr_t)out & 15u) == 0) *reinterpret_cast<uint4 *>(out) = make_uint4(bins[0], bins[1], bins[2], bins[3]);
    else
        for (int k = 0; k < 4 && p0 + k < N; ++k) out[k] = bins[k];
}

// sum of the lane numbers of a wave mask (scalar arithmetic: the mask is the same in every lane)
__device__ __forceinline__ unsigned lane_number_sum(unsigned long long m)
{
    return (unsigned)(__popcll(m & 0xaaaaaaaaaaaaaaaaull) + 2 * __popcll(m & 0xccccccccccccccccull) + 4 * __popcll(m & 0xf0f0f0f0f0f0f0f0ull) +
                      8 * __popcll(m & 0xff00ff00ff00ff00ull) + 16 * __popcll(m & 0xffff0000ffff0000ull) + 32 * __popcll(m & 0xffffffff00000000ull));
}

// relabel, int32 working copies, interior counts and position sums, patch-pixel counts per block (the input of
// the counting sort that builds the point lists).  A wave's 64 consecutive pixels carry one to three labels, and 64
// LDS atomics on one address go through one after the other: the wave adds up per label instead.  Its pixels are
// consecutive, so with a frame at least a wave wide (at most one row change inside the wave) the sums of x and y over
// a set of lanes follow from the set's mask without any exchange between lanes.
__global__ __launch_bounds__(SB) void stats_prepare_kernel(const StatsFrame *__restrict__ frames, int32_t *__restrict__ surf32_all,
                                                           int32_t *__restrict__ pat32_all, int *__restrict__ err, int H, int W)
{
    extern __shared__ uint32_t sh[];             // [4][S]: interior count, sum x, sum y, patch count
    const StatsFrame &f = frames[blockIdx.y];
    const int N = H * W, S = f.S;
    const int64_t *__restrict__ surfaces = f.surfaces, *__restrict__ patches = f.patches;
    int32_t *__restrict__ surf32 = surf32_all + (size_t)blockIdx.y * N, *__restrict__ pat32 = pat32_all + (size_t)blockIdx.y * N;
    unsigned long long *__restrict__ acc = f.acc;
    for (int i = threadIdx.x; i < 4 * S; i += SB) sh[i] = 0u;
    const int lane = threadIdx.x & 63;
    int64_t pls[SPB / SB], sls[SPB / SB];
#pragma unroll
    for (int k = 0; k < SPB / SB; ++k) {         // every load of the block is on its way before the first label is looked at
        const int p = blockIdx.x * SPB + k * SB + threadIdx.x;
        pls[k] = p < N ? patches[p] : 0;
        sls[k] = p < N ? surfaces[p] : 0;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < SPB / SB; ++k) {
        const int pw = __builtin_amdgcn_readfirstlane(blockIdx.x * SPB + k * SB + (int)(threadIdx.x & ~63u));   // the wave's first pixel
        const int p = pw + lane;
        const bool in = p < N;
        const int64_t pl = pls[k], sl = sls[k];
        const bool good = in && !(pl < 0 || pl >= S || sl < 0 || sl >= S);
        if (in && !good) *err = 1;                             // every lane that writes stores the same value
        const int s = good ? (pl != 0 ? (int)pl : (int)sl) : 0, q = good ? (int)pl : 0;
        if (in) {
            surf32[p] = s;
            pat32[p] = q;
        }
        if (W >= 64) {
            const int yw = pw / W, xw = pw - yw * W;
            const int wrap = W - xw;                           // first lane of the next row, if below 64
            const unsigned long long next_row = wrap < 64 ? ~0ull << wrap : 0ull;
            const int y = yw + (lane >= wrap ? 1 : 0), x = xw + lane - (lane >= wrap ? W : 0);
            const bool interior = good && y >= 1 && y < H - 1 && x >= 1 && x < W - 1;
            wave_count_add(sh + 3 * S, q, good);               // patch pixels per label
            unsigned long long todo = __ballot(interior);
            while (todo) {                                     // interior pixels of the relabelled image and their positions
                const int leader = __ffsll((long long)todo) - 1;
                const int key = __builtin_amdgcn_readlane(s, leader);
                const unsigned long long same = __ballot(interior && s == key) & todo;
                const unsigned c = (unsigned)__popcll(same), c_next = (unsigned)__popcll(same & next_row);
                if (lane == 0) {
                    atomicAdd(&sh[key], c);
                    atomicAdd(&sh[S + key], c * (unsigned)xw + lane_number_sum(same) - c_next * (unsigned)W);
                    atomicAdd(&sh[2 * S + key], c * (unsigned)yw + c_next);
                }
                todo &= ~same;
            }
        } else if (good) {                                     // narrow frames: a wave spans several rows
            atomicAdd(&sh[3 * S + q], 1u);
            const int y = p / W, x = p - y * W;
            if (y >= 1 && y < H - 1 && x >= 1 && x < W - 1) {
                atomicAdd(&sh[s], 1u);
                atomicAdd(&sh[S + s], (uint32_t)x);
                atomicAdd(&sh[2 * S + s], (uint32_t)y);
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 4 * S; i += SB)
        if (sh[i]) atomicAdd(acc + i, (unsigned long long)sh[i]);
    for (int i = threadIdx.x; i < S; i += SB) f.blockcnt[(size_t)blockIdx.x * S + i] = sh[3 * S + i];
}

// one block per frame: patch-point offsets (exclusive scan over the labels), counts and average positions, and
// the counting sort's second step: blockcnt[b][l] becomes the position of block b's first pixel of label l
__global__ __launch_bounds__(1024) void stats_offsets_kernel(const StatsFrame *__restrict__ frames, int nblk)
{
    __shared__ int part[1024];
    const StatsFrame &f = frames[blockIdx.x];
    const int S = f.S;
    const unsigned long long *__restrict__ acc = f.acc;
    int32_t *__restrict__ offsets = f.offsets;
    double *__restrict__ avg_pos = f.avg_pos, *__restrict__ counts = f.counts;
    const int t = threadIdx.x;
    const int v = t < S ? (int)acc[3 * S + t] : 0;
    part[t] = v;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const int u = t >= off ? part[t - off] : 0;
        __syncthreads();
        part[t] += u;
        __syncthreads();
    }
    if (t < S) {
        offsets[t] = part[t] - v;
        if (t == S - 1) offsets[S] = part[t];
        const double c = (double)acc[t];
        counts[t] = c;
        avg_pos[2 * t] = c > 0 ? (double)acc[S + t] / c : 0.0;
        avg_pos[2 * t + 1] = c > 0 ? (double)acc[2 * S + t] / c : 0.0;
    }
    // exclusive scan of every label's block counts, starting at the label's offset: a wave per label, 64 blocks per step
    // (one lane per label walked the frame's few hundred blocks one dependent load at a time: 50 us for one frame)
    __shared__ int base[1024];
    base[t] = part[t] - v;
    __syncthreads();
    const int lane = t & 63;
    for (int l = t >> 6; l < S; l += 1024 / 64) {
        uint32_t run = (uint32_t)base[l];
        for (int b0 = 0; b0 < nblk; b0 += 64) {
            const int b = b0 + lane;
            const uint32_t c0 = b < nblk ? f.blockcnt[(size_t)b * S + l] : 0u;
            uint32_t x = c0;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t y = __shfl_up(x, off);
                if (lane >= off) x += y;
            }
            if (b < nblk) f.blockcnt[(size_t)b * S + l] = run + x - c0;
            run += __shfl(x, 63);
        }
    }
}

// extract_individual_points_from_surfaces (:96-108) is a stable counting sort of the pixels by patch label.  A block
// holds 1024 consecutive pixels as 16 segments of 64 (one wave, one pass); a pixel's position in its label's list is
// blockcnt[block][label] + the label's pixels in the block's earlier segments + those in lower lanes of its own wave.
__global__ __launch_bounds__(SB) void stats_points_kernel(const StatsFrame *__restrict__ frames, const int32_t *__restrict__ pat32_all,
                                                          int N, int W)
{
    extern __shared__ uint32_t seg[];            // [16][S]
    const StatsFrame &f = frames[blockIdx.y];
    const int S = f.S;
    const int32_t *__restrict__ pat32 = pat32_all + (size_t)blockIdx.y * N;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 16 * S; i += SB) seg[i] = 0u;
    __syncthreads();
    int lab[SPB / SB], rank[SPB / SB];
#pragma unroll
    for (int k = 0; k < SPB / SB; ++k) {
        const int p = blockIdx.x * SPB + k * SB + threadIdx.x;
        const bool in = p < N;
        lab[k] = in ? pat32[p] : -1;
        rank[k] = 0;
        unsigned long long todo = __ballot(in);
        while (todo) {                           // a wave of consecutive pixels holds few labels
            const int leader = __ffsll((long long)todo) - 1;
            const int key = __shfl(lab[k], leader);
            const unsigned long long same = __ballot(lab[k] == key) & todo;
            if (lab[k] == key) rank[k] = __popcll(same & ((1ull << lane) - 1ull));
            if (lane == leader) seg[(k * 4 + wv) * S + key] = (uint32_t)__popcll(same);
            todo &= ~same;
        }
    }
    __syncthreads();
    for (int l = threadIdx.x; l < S; l += SB) {
        uint32_t run = f.blockcnt[(size_t)blockIdx.x * S + l];
        for (int sg = 0; sg < 16; ++sg) {
            const uint32_t c0 = seg[sg * S + l];
            seg[sg * S + l] = run;
            run += c0;
        }
    }
    __syncthreads();
    float2 *__restrict__ points = reinterpret_cast<float2 *>(f.patch_points);
#pragma unroll
    for (int k = 0; k < SPB / SB; ++k) {
        if (lab[k] < 0) continue;
        const int p = blockIdx.x * SPB + k * SB + threadIdx.x;
        const int y = p / W;
        points[seg[(k * 4 + wv) * S + lab[k]] + rank[k]] = make_float2((float)(p - y * W), (float)y);
    }
}

__device__ __forceinline__ unsigned long long dist_bits(int p, int W, const double *avg_pos, int l)
{
    const int y = p / W, x = p - y * W;
    const double dx = (double)x - avg_pos[2 * l], dy = (double)y - avg_pos[2 * l + 1];
    return (unsigned long long)__double_as_longlong(dx * dx + dy * dy);   // non-negative doubles order like their bit patterns
}

// pass 1: smallest squared distance per label; pass 2: smallest pixel index among the pixels that attain it
template <int PASS>
__global__ __launch_bounds__(SB) void stats_centroid_kernel(const StatsFrame *__restrict__ frames, const int32_t *__restrict__ pat32_all,
                                                            int N, int W)
{
    extern __shared__ unsigned long long shb[];  // [S]
    const StatsFrame &f = frames[blockIdx.y];
    const int S = f.S;
    const int32_t *__restrict__ pat32 = pat32_all + (size_t)blockIdx.y * N;
    const double *__restrict__ avg_pos = f.avg_pos;
    unsigned long long *__restrict__ best = f.best;
    for (int i = threadIdx.x; i < S; i += SB) shb[i] = ~0ull;
    __syncthreads();
    for (int k = 0; k < SPB / SB; ++k) {
        const int p = blockIdx.x * SPB + k * SB + threadIdx.x;
        if (p >= N) continue;
        const int l = pat32[p];
        const unsigned long long d = dist_bits(p, W, avg_pos, l);
        if (PASS == 1) {
            if (d < shb[l]) atomicMin(&shb[l], d);
        } else if (d == best[l]) {
            atomicMin(&shb[l], (unsigned long long)p);
        }
    }
    __syncthreads();
    unsigned long long *dst = PASS == 1 ? best : best + S;
    for (int i = threadIdx.x; i < S; i += SB)
        if (shb[i] != ~0ull) atomicMin(dst + i, shb[i]);
}

__global__ __launch_bounds__(SB) void stats_centroid_finish_kernel(const StatsFrame *__restrict__ frames, int W)
{
    const StatsFrame &f = frames[blockIdx.y];
    const int S = f.S;
    const unsigned long long *__restrict__ best = f.best;
    const float *__restrict__ points_3d = f.points_3d;
    float *__restrict__ centroids = f.centroids;
    int32_t *__restrict__ centroid_idx = f.centroid_idx, *__restrict__ cpix = f.cpix;
    const int l = blockIdx.x * SB + threadIdx.x;
    if (l >= S) return;
    const unsigned long long b = best[S + l];
    if (b == ~0ull) {
        cpix[l] = -1;
        centroid_idx[2 * l] = centroid_idx[2 * l + 1] = -1;
        centroids[3 * l] = centroids[3 * l + 1] = centroids[3 * l + 2] = -1.0f;
        return;
    }
    const int p = (int)b, y = p / W;
    cpix[l] = p;
    centroid_idx[2 * l] = y;
    centroid_idx[2 * l + 1] = p - y * W;
    for (int c = 0; c < 3; ++c) centroids[3 * l + c] = points_3d[3 * (size_t)p + c];
}

// keep a pixel iff it lies in the component of its label's centroid pixel (or is the start pixel of label 0's fill)
__global__ __launch_bounds__(SB) void stats_keep_kernel(const StatsFrame *__restrict__ frames, int32_t *__restrict__ surf32_all,
                                                        const int32_t *__restrict__ parent_all, int N)
{
    const StatsFrame &f = frames[blockIdx.y];
    int32_t *__restrict__ surf32 = surf32_all + (size_t)blockIdx.y * N;
    const int32_t *__restrict__ parent = parent_all + (size_t)blockIdx.y * N, *__restrict__ cpix = f.cpix;
    int64_t *__restrict__ surfaces_out = f.surfaces_out;
    const int p = blockIdx.x * SB + threadIdx.x;
    if (p >= N) return;
    const int s = surf32[p];
    int keep = 0;
    if (s != 0) {
        const int c = cpix[s];
        keep = (c >= 0 && parent[p] == parent[c]) || p == cpix[0];
    }
    const int out = keep ? s : 0;
    surf32[p] = out;
    surfaces_out[p] = out;
}

// histograms, angle sums, neighbour counts and border sums, norm and per-label extents
__global__ __launch_bounds__(SB) void stats_accumulate_kernel(const StatsFrame *__restrict__ frames, const int32_t *__restrict__ surf32_all,
                                                              const int32_t *__restrict__ pat32_all, int *__restrict__ err, int H, int W,
                                                              int pixels_per_block, int window_table)
{
    const StatsFrame &f = frames[blockIdx.y];
    const int S = f.S;
    const int32_t *__restrict__ surf32 = surf32_all + (size_t)blockIdx.y * H * W, *__restrict__ pat32 = pat32_all + (size_t)blockIdx.y * H * W;
    const int32_t *__restrict__ edges = f.edges;
    const double *__restrict__ lab = f.lab;
    const uint32_t *__restrict__ color_bin_of = f.color_bin;
    const float *__restrict__ angles = f.angles, *__restrict__ points_3d = f.points_3d;
    uint32_t *__restrict__ hist_color = f.hist_color, *__restrict__ hist_angle = f.hist_angle, *__restrict__ cnt = f.cnt;
    uint32_t *__restrict__ nb_count = f.nb_count, *__restrict__ extent = f.extent;
    double *__restrict__ angle_sum = f.angle_sum;
    unsigned long long *__restrict__ nb_sum = f.nb_sum;
    float *__restrict__ norm = f.norm;
    // per-block copies of everything indexed by label: global atomics on one address per label would
    // serialise (a patch has ~1e5 pixels), LDS atomics do not
    extern __shared__ double shd[];              // [S][2] angle sums | then uint32: [S][8] min/max of (norm, x, y, z) | [S] surface count | [S] patch count
    double *s_ang = shd;
    uint32_t *she = reinterpret_cast<uint32_t *>(shd + 2 * S);
    uint32_t *s_cnt = she + 8 * S, *p_cnt = s_cnt + S;
    // Colour counts of a fat block (window_table): a surface's colours lie close together, so the first kWinSlots labels
    // the block meets get a window of 32 x 32 bins each in LDS, placed around the first colour seen of the label, with
    // 16-bit counters (a block has at most 8 192 pixels); a pixel inside its label's window costs an LDS atomic, the
    // windows go to the frame's histogram once at the block's end (a few hundred global atomics instead of a few
    // thousand), and pixels outside, or of a label without a window, add to the frame's histogram directly as before.
    int *win_slot = reinterpret_cast<int *>(p_cnt + S);               // [S] slot of a label: -1 none yet, -2 being set up, -3 none left
    int *win_origin = win_slot + S, *win_label = win_origin + kWinSlots, *win_used = win_label + kWinSlots;
    uint32_t *win_tab = reinterpret_cast<uint32_t *>(win_used + 1);   // [kWinSlots][32][32] counters, two per word
    const int N = H * W;
    for (int i = threadIdx.x; i < 8 * S; i += SB) she[i] = (i & 1) ? 0u : ~0u;
    for (int i = threadIdx.x; i < 2 * S; i += SB) { s_cnt[i] = 0u; s_ang[i] = 0.0; }
    if (window_table) {
        for (int i = threadIdx.x; i < S; i += SB) win_slot[i] = -1;
        for (int i = threadIdx.x; i < kWinSlots * 512; i += SB) win_tab[i] = 0u;
        if (threadIdx.x == 0) *win_used = 0;
    }
    __syncthreads();
    for (int k = 0; k < pixels_per_block / SB; ++k) {
        const int p = blockIdx.x * pixels_per_block + k * SB + threadIdx.x;
        const bool in = p < N;
        const int y = in ? p / W : 0, x = in ? p - y * W : 0;
        const int s = in ? surf32[p] : 0;
        uint32_t color_bin = 0, angle_bin = 0;
        bool color_on = false, angle_on = false;
        if (in) {
            // depth extent (every label, 0 included), :332-346
            const float px = points_3d[3 * (size_t)p], py = points_3d[3 * (size_t)p + 1], pz = points_3d[3 * (size_t)p + 2];
            const float n = sqrtf((px * px + py * py) + pz * pz);
            norm[p] = n;
            if (n != 0.0f) {
                const float v[4] = {n, px, py, pz};
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const unsigned e = enc_f32(v[c]);
                    if (e < she[s * 8 + 2 * c]) atomicMin(&she[s * 8 + 2 * c], e);
                    if (e > she[s * 8 + 2 * c + 1]) atomicMax(&she[s * 8 + 2 * c + 1], e);
                }
            }
        }
        if (s > 0) {
            // colour histogram, :265-267
            uint32_t ab;
            if (color_bin_of) {
                ab = color_bin_of[p];
            } else {
                const long a = (long)rint(lab[3 * (size_t)p + 1] + 127.0), b = (long)rint(lab[3 * (size_t)p + 2] + 127.0);
                ab = a < 0 || a > 255 || b < 0 || b > 255 ? 0xffffffffu : (uint32_t)(a * 256 + b);
            }
            if (ab == 0xffffffffu) *err = 2;
            else { color_on = true; color_bin = ab * (uint32_t)S + (uint32_t)s; }
            if (window_table && color_on) {
                const int a = (int)(ab >> 8), b = (int)(ab & 255u);
                int slot = win_slot[s];
                if (slot == -1 && atomicCAS(&win_slot[s], -1, -2) == -1) {        // this lane sets the label's window up
                    const int k = atomicAdd(win_used, 1);
                    if (k < kWinSlots) {
                        win_origin[k] = min(max(a - 16, 0), 224) << 8 | min(max(b - 16, 0), 224);
                        win_label[k] = s;
                        __threadfence_block();
                        slot = k;
                    } else {
                        slot = -3;
                    }
                    __hip_atomic_store(&win_slot[s], slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                if (slot >= 0) {
                    const int o = win_origin[slot];
                    const unsigned da = (unsigned)(a - (o >> 8)), db = (unsigned)(b - (o & 255));
                    if (da < 32u && db < 32u) {
                        const unsigned idx = (unsigned)slot * 1024u + da * 32u + db;
                        atomicAdd(&win_tab[idx >> 1], 1u << (16u * (idx & 1u)));
                        color_on = false;                                        // counted
                    }
                }
            }
            atomicAdd(&s_cnt[s], 1u);
            // neighbours, :39-55 (pairs with label 0 and pairs of equal labels are zeroed afterwards: skipped)
            if (edges[p] != 0) {
                if (y < H - 1) {
                    const int o = edges[p + W] != 0 ? surf32[p + W] : 0;
                    if (o != 0 && o != s) {
                        atomicAdd(nb_count + (size_t)s * S + o, 1u);
                        atomicAdd(nb_sum + 2 * ((size_t)s * S + o), (unsigned long long)(2 * x + 1));
                        atomicAdd(nb_sum + 2 * ((size_t)s * S + o) + 1, (unsigned long long)(2 * y));
                    }
                }
                if (x < W - 1) {
                    const int o = edges[p + 1] != 0 ? surf32[p + 1] : 0;
                    if (o != 0 && o != s) {
                        atomicAdd(nb_count + (size_t)s * S + o, 1u);
                        atomicAdd(nb_sum + 2 * ((size_t)s * S + o), (unsigned long long)(2 * x));
                        atomicAdd(nb_sum + 2 * ((size_t)s * S + o) + 1, (unsigned long long)(2 * y + 1));
                    }
                }
            }
            // angle histogram and sums over patch pixels, :269-276
            const int pl = pat32[p];
            if (pl > 0) {
                const float a0 = angles[2 * (size_t)p], a1 = angles[2 * (size_t)p + 1];
                const int i0 = (int)(((double)a0 / (2 * M_PI) + 0.5) * 39), i1 = (int)(((double)a1 / (2 * M_PI) + 0.5) * 39);
                if (i0 < 0 || i0 > 39 || i1 < 0 || i1 > 39) *err = 3;
                else { angle_on = true; angle_bin = (uint32_t)((i0 * 40 + i1) * S + pl); }
                atomicAdd(&s_ang[2 * pl], (double)a0);
                atomicAdd(&s_ang[2 * pl + 1], (double)a1);
                atomicAdd(&p_cnt[pl], 1u);
            }
        }
        wave_histogram_add(hist_color, color_bin, color_on);
        wave_histogram_add(hist_angle, angle_bin, angle_on);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 8 * S; i += SB) {
        const uint32_t v = she[i];
        if (i & 1) { if (v != 0u) atomicMax(extent + i, v); }
        else if (v != ~0u) atomicMin(extent + i, v);
    }
    for (int i = threadIdx.x; i < 2 * S; i += SB) {
        if (s_cnt[i]) atomicAdd(cnt + i, s_cnt[i]);
        if (s_ang[i] != 0.0) atomicAdd(angle_sum + i, s_ang[i]);
    }
    if (window_table) {
        const int used = min(*win_used, kWinSlots);
        for (int i = threadIdx.x; i < used * 512; i += SB) {
            const uint32_t w = win_tab[i];
            if (!w) continue;
            const int slot = i >> 9, o = win_origin[slot], label = win_label[slot];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const uint32_t v = (w >> (16 * h)) & 0xffffu;
                const int idx = 2 * i + h, a = (o >> 8) + ((idx >> 5) & 31), b = (o & 255) + (idx & 31);
                if (v) atomicAdd(hist_color + (size_t)(a * 256 + b) * S + label, v);
            }
        }
    }
}

// texture vectors of every 4th pixel of a patch, :278-283: one block per texture row, a thread per component
__global__ __launch_bounds__(SB) void stats_texture_kernel(const StatsFrame *__restrict__ frames, const int32_t *__restrict__ surf32_all,
                                                           const int32_t *__restrict__ pat32_all, int N, int W, int TW, int T)
{
    const StatsFrame &f = frames[blockIdx.y];
    const float *__restrict__ texture = f.texture;
    if (!texture) return;
    const int32_t *__restrict__ surf32 = surf32_all + (size_t)blockIdx.y * N, *__restrict__ pat32 = pat32_all + (size_t)blockIdx.y * N;
    double *__restrict__ tex_sum = f.tex_sum;
    uint32_t *__restrict__ tex_cnt = f.tex_cnt;
    const int ty = blockIdx.x;
    // consecutive texture pixels of one label are summed in registers before touching memory
    for (int c = threadIdx.x; c < T; c += SB) {
        double run = 0.0;
        int cur = 0, curn = 0;
        for (int tx = 0; tx < TW; ++tx) {
            const size_t p = (size_t)(4 * ty) * W + 4 * tx;
            const int s = surf32[p] > 0 ? pat32[p] : 0;
            if (s != cur) {
                if (cur > 0) {
                    atomicAdd(tex_sum + (size_t)cur * T + c, run);
                    if (c == 0) atomicAdd(tex_cnt + cur, (uint32_t)curn);
                }
                cur = s; run = 0.0; curn = 0;
            }
            if (s > 0) {
                run += (double)texture[((size_t)ty * TW + tx) * T + c];
                ++curn;
            }
        }
        if (cur > 0) {
            atomicAdd(tex_sum + (size_t)cur * T + c, run);
            if (c == 0) atomicAdd(tex_cnt + cur, (uint32_t)curn);
        }
    }
}

// normalisations of :284-295 and conversion to the float64 outputs
__global__ __launch_bounds__(SB) void stats_finalize_kernel(const StatsFrame *__restrict__ frames, int T)
{
    const StatsFrame &f = frames[blockIdx.y];
    const int S = f.S;
    const uint32_t *__restrict__ hist_color = f.hist_color, *__restrict__ hist_angle = f.hist_angle, *__restrict__ cnt = f.cnt;
    const uint32_t *__restrict__ tex_cnt = f.tex_cnt, *__restrict__ extent = f.extent;
    const double *__restrict__ angle_sum = f.angle_sum, *__restrict__ tex_sum = f.tex_sum;
    double *__restrict__ out_color = f.hist_color_out, *__restrict__ out_angle = f.hist_angle_out, *__restrict__ out_texture = f.hist_texture_out;
    double *__restrict__ out_avg_angles = f.avg_angles;
    float *__restrict__ out_extent = f.depth_extend;
    // (the largest index is 256 * 256 * kMaxLabels < 2^32: label and bin come from 32-bit divisions, a 64-bit one costs
    // several times the rest of the element's work; a bin nobody fell into is 0 without its float64 division)
    const uint32_t i = blockIdx.x * (uint32_t)SB + threadIdx.x;
    const uint32_t n_color = 256u * 256u * (uint32_t)S, n_angle = 40u * 40u * (uint32_t)S, n_tex = (uint32_t)S * (uint32_t)T;
    // colour histogram: four bins per lane (one 16-byte load, two 16-byte stores) when the arrays allow it; it is
    // nine tenths of the kernel's bytes
    auto color_value = [&](uint32_t idx, uint32_t h) {
        if (h == 0u) return 0.0;
        const uint32_t s = idx % (uint32_t)S;
        double norm = (double)cnt[s] / 10;
        if (norm == 0) norm = 1;
        return (double)h / norm;
    };
    if ((((uintptr_t)hist_color | (uintptr_t)out_color) & 15u) == 0) {
        if (4u * i < n_color) {                                  // 256 * 256 * S is a multiple of four
            const uint4 h = *reinterpret_cast<const uint4 *>(hist_color + 4u * i);
            double2 *o = reinterpret_cast<double2 *>(out_color + 4u * i);
            o[0] = make_double2(color_value(4u * i, h.x), color_value(4u * i + 1u, h.y));
            o[1] = make_double2(color_value(4u * i + 2u, h.z), color_value(4u * i + 3u, h.w));
        }
    } else {
        for (uint32_t j = 4u * i; j < 4u * i + 4u && j < n_color; ++j) out_color[j] = color_value(j, hist_color[j]);
    }
    if (i < n_angle) {
        const uint32_t h = hist_angle[i];
        double v = 0.0;
        if (h != 0u) {
            const uint32_t s = i % (uint32_t)S;
            double pc = (double)cnt[S + s];
            if (pc == 0) pc = 1;
            v = (double)h / (pc / 10);
        }
        out_angle[i] = v;
    }
    if (i < n_tex) {
        const uint32_t s = i / (uint32_t)T;
        double c = (double)tex_cnt[s];
        if (c == 0) c = 1;
        out_texture[i] = tex_sum[i] / c * 10;
    }
    if (i < 2u * (uint32_t)S) {
        double pc = (double)cnt[S + i / 2];
        if (pc == 0) pc = 1;
        out_avg_angles[i] = angle_sum[i] / pc;
    }
    if (i < 8u * (uint32_t)S) {
        const uint32_t e = extent[i];
        out_extent[i] = (i & 1) ? (e == 0u ? -INFINITY : dec_f32(e)) : (e == ~0u ? INFINITY : dec_f32(e));
    }
}

// The S x S tail of determine_neighboring_surfaces_and_border_centers (process_surfaces.py:56-68).  Every adjacent
// pair was counted once, in the direction it was met; the reference adds it to both entries, zeroes row 0, column 0
// and the diagonal, replaces every count below 4 by 1 (the diagonal and row/column 0 included: they are 0 by then)
// and divides the coordinate sums by twice the count.
__global__ __launch_bounds__(SB) void stats_neighbors_kernel(const StatsFrame *__restrict__ frames)
{
    const StatsFrame &f = frames[blockIdx.y];
    const int S = f.S;
    const uint32_t *__restrict__ nb_count = f.nb_count;
    const unsigned long long *__restrict__ nb_sum = f.nb_sum;
    double *__restrict__ neighbors = f.neighbors, *__restrict__ border_centers = f.border_centers;
    const size_t a = (size_t)blockIdx.x * SB + threadIdx.x;
    if (a >= (size_t)S * S) return;
    const int i = (int)(a / S), j = (int)(a - (size_t)i * S);
    const size_t b = (size_t)j * S + i;
    const bool dead = i == 0 || j == 0 || i == j;
    double n = dead ? 0.0 : (double)(nb_count[a] + nb_count[b]);
    const double sx = dead ? 0.0 : (double)(nb_sum[2 * a] + nb_sum[2 * b]);
    const double sy = dead ? 0.0 : (double)(nb_sum[2 * a + 1] + nb_sum[2 * b + 1]);
    if (n < 4) n = 1;
    neighbors[a] = n;
    border_centers[2 * a] = sx / (2 * n);
    border_centers[2 * a + 1] = sy / (2 * n);
}

// extent[i] = ~0 (minimum slots) / 0 (maximum slots) for the whole batch
__global__ __launch_bounds__(SB) void stats_fill_kernel(uint32_t *__restrict__ a, size_t n, uint32_t even, uint32_t odd)
{
    const size_t i = (size_t)blockIdx.x * SB + threadIdx.x;
    if (i < n) a[i] = (i & 1) ? odd : even;
}

}  // namespace

// Label-indexed scratch of one frame, in four regions of the batch's scratch buffer so that the whole batch is
// initialised by three calls: `zero` (cleared), `ones` (0xff: the centroid search), `ext` (min / max pattern) and
// `plain` (written before it is read).
StatsScratchSizes stats_scratch_sizes(int S, int T, int nblk)
{
    StatsScratchSizes z;
    // hist_color u32 (first: every frame's starts on a 16-byte boundary, which the finalisation's four-bin loads want) |
    // acc[4S] u64 | nb_sum[2SS] u64 | angle_sum[2S] f64 | tex_sum[S*T] f64 | hist_angle u32 | nb_count[SS] u32 | cnt[2S] u32 |
    // tex_cnt[S] u32 (+ pad to 16 bytes)
    z.zero = 8 * ((size_t)4 * S + 2 * (size_t)S * S + 2 * S + (size_t)S * T) +
             4 * ((size_t)256 * 256 * S + 1600 * (size_t)S + (size_t)S * S + 2 * S + S);
    z.zero = (z.zero + 15) / 16 * 16;
    z.ones = 8 * (size_t)2 * S;                                   // best[2S] u64
    z.ext = 4 * (size_t)8 * S;                                    // extent[8S] u32
    z.plain = 4 * (((size_t)S + (size_t)nblk * S + 1) / 2 * 2);   // cpix[S] i32 | blockcnt[nblk][S] u32
    return z;
}

void stats_carve_scratch(StatsFrame &f, int T, int nblk, char *zero, char *ones, char *ext, char *plain)
{
    const size_t S = f.S;
    auto take = [](char *&base, size_t bytes) { char *p = base; base += bytes; return p; };
    f.hist_color = reinterpret_cast<uint32_t *>(take(zero, 4 * (size_t)256 * 256 * S));
    f.acc = reinterpret_cast<unsigned long long *>(take(zero, 8 * 4 * S));
    f.nb_sum = reinterpret_cast<unsigned long long *>(take(zero, 8 * 2 * S * S));
    f.angle_sum = reinterpret_cast<double *>(take(zero, 8 * 2 * S));
    f.tex_sum = reinterpret_cast<double *>(take(zero, 8 * S * T));
    f.hist_angle = reinterpret_cast<uint32_t *>(take(zero, 4 * (size_t)1600 * S));
    f.nb_count = reinterpret_cast<uint32_t *>(take(zero, 4 * S * S));
    f.cnt = reinterpret_cast<uint32_t *>(take(zero, 4 * 2 * S));
    f.tex_cnt = reinterpret_cast<uint32_t *>(take(zero, 4 * S));
    f.best = reinterpret_cast<unsigned long long *>(ones);
    f.extent = reinterpret_cast<uint32_t *>(ext);
    f.cpix = reinterpret_cast<int32_t *>(take(plain, 4 * S));
    f.blockcnt = reinterpret_cast<uint32_t *>(take(plain, 4 * (size_t)nblk * S));
}

int stats_blocks_per_frame(int H, int W) { return (H * W + SPB - 1) / SPB; }

// frames: device array of n descriptors (host copy in frames_host for the launch geometry); surf32 / pat32 / parent /
// count: [n][H*W] int32 work arrays; lin: 256 linearised sRGB channel values (used when a frame gives rgb instead of lab);
// ext_all / ext_words: the batch's extent region; err: one int, cleared by the caller.
hipError_t launch_surface_stats_batch(const StatsFrame *frames, const StatsFrame *frames_host, int n, const StatsBatchWork &w, int H, int W,
                                      int T, hipStream_t st, const StatsStageHook &after_stage)
{
    const int N = H * W;
    const int nblk = stats_blocks_per_frame(H, W);
    int Smax = 1;
    bool any_rgb = false, any_texture = false, any_lab_out = false, any_bins_only = false;
    for (int i = 0; i < n; ++i) {
        Smax = std::max(Smax, frames_host[i].S);
        any_rgb |= frames_host[i].rgb != nullptr;
        any_lab_out |= frames_host[i].rgb && frames_host[i].lab_work;
        any_bins_only |= frames_host[i].rgb && !frames_host[i].lab_work;
        any_texture |= frames_host[i].texture != nullptr;
    }
    const dim3 by_pixel((N + SB - 1) / SB, n), by_block(nblk, n);
    hipLaunchKernelGGL(stats_fill_kernel, dim3((unsigned)((w.ext_words + SB - 1) / SB)), dim3(SB), 0, st, w.ext_all, w.ext_words, ~0u, 0u);
    if (any_rgb && after_stage) after_stage(10);
    if (any_lab_out) hipLaunchKernelGGL(stats_lab_kernel, by_pixel, dim3(SB), 0, st, frames, w.lin, N);
    if (any_bins_only) hipLaunchKernelGGL(stats_color_bins_kernel, dim3((N + 4 * SB - 1) / (4 * SB), n), dim3(SB), 0, st, frames, w.lin, w.lin32, N);
    if (any_rgb && after_stage) after_stage(0);
    if (after_stage) after_stage(11);
    hipLaunchKernelGGL(stats_prepare_kernel, by_block, dim3(SB), 4 * Smax * sizeof(uint32_t), st, frames, w.surf32, w.pat32, w.err, H, W);
    hipLaunchKernelGGL(stats_offsets_kernel, dim3(n), dim3(1024), 0, st, frames, nblk);
    hipLaunchKernelGGL(stats_points_kernel, by_block, dim3(SB), 16 * Smax * sizeof(uint32_t), st, frames, w.pat32, N, W);
    hipLaunchKernelGGL(stats_centroid_kernel<1>, by_block, dim3(SB), Smax * sizeof(unsigned long long), st, frames, w.pat32, N, W);
    hipLaunchKernelGGL(stats_centroid_kernel<2>, by_block, dim3(SB), Smax * sizeof(unsigned long long), st, frames, w.pat32, N, W);
    if (after_stage) after_stage(12);
    hipLaunchKernelGGL(stats_centroid_finish_kernel, dim3((Smax + SB - 1) / SB, n), dim3(SB), 0, st, frames, W);
    launch_label_components(w.surf32, w.parent, nullptr, n, H, W, st);      // the keep pass compares roots; sizes are not needed
    hipLaunchKernelGGL(stats_keep_kernel, by_pixel, dim3(SB), 0, st, frames, w.surf32, w.parent, N);
    if (after_stage) after_stage(1);
    if (after_stage) after_stage(13);
    // Pixels per block of the accumulation pass: a block's pixels go through one after the other, each a chain of loads
    // and atomics, and every block ends by flushing its per-label sums into the frame's.  Few frames want many small
    // blocks (one pixel per lane: the chip is not full otherwise), a large batch few fat ones (fewer flushes; on 111
    // frames 256 / 1 024 / 8 192 pixels per block measured 4.34 / 4.00 / 3.75 ms for the whole stage): about 4 096 blocks
    // in all, between 256 and 8 192 pixels each.
    int acc_ppb = SB;
    while (acc_ppb < 8192 && (size_t)n * N / (2 * (size_t)acc_ppb) >= 4096) acc_ppb *= 2;
    const int window_table = acc_ppb >= 2048;      // (setting the windows up and sweeping them costs a block about what 500 pixels do)
    hipLaunchKernelGGL(stats_accumulate_kernel, dim3((N + acc_ppb - 1) / acc_ppb, n), dim3(SB),
                       10 * Smax * sizeof(uint32_t) + 2 * Smax * sizeof(double) +
                           (window_table ? (Smax + 2 * kWinSlots + 1 + kWinSlots * 512) * sizeof(uint32_t) : 0),
                       st, frames, w.surf32, w.pat32, w.err, H, W, acc_ppb, window_table);
    if (any_texture)
        hipLaunchKernelGGL(stats_texture_kernel, dim3(H / 4, n), dim3(SB), 0, st, frames, w.surf32, w.pat32, N, W, W / 4, T);
    const size_t n_fin = std::max<size_t>((size_t)256 * 256 * Smax / 4, std::max<size_t>((size_t)Smax * T, (size_t)1600 * Smax));   // lanes: four colour bins each
    hipLaunchKernelGGL(stats_finalize_kernel, dim3((unsigned)((n_fin + SB - 1) / SB), n), dim3(SB), 0, st, frames, T);
    hipLaunchKernelGGL(stats_neighbors_kernel, dim3((unsigned)(((size_t)Smax * Smax + SB - 1) / SB), n), dim3(SB), 0, st, frames);
    if (after_stage) after_stage(2);
    return hipGetLastError();
}

}  // namespace fs
