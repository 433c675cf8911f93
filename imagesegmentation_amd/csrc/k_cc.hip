// Connected-component labelling of the smooth-patch mask (gfx950).
//
// Replaces the two tfa.image.connected_components calls and the bincount size
// filter of find_surfaces_model_GPU (find_surfaces.py:281-285,
// image_processing_models_GPU.py:74-80): 4-connectivity, ids 1..n in row-major
// order of each component's first pixel, components of <= 20 pixels removed.
//
// Algorithm (all integer, schedule independent):
//   1. local   components inside 64 x 16 tiles in LDS: a wave labels one tile row, a
//              ballot of the mask gives every lane the start of its horizontal run,
//              which becomes its parent (so horizontal unions are free); vertical
//              neighbours are joined by a union-find with LDS atomicMin (skipped
//              when the pixel to the left already carries the same link);
//   2. border  the links across tile borders with the same union-find on the flat
//              indices in global memory;
//              the root of a tree is always its smallest flat index, i.e. the
//              component's first pixel in raster order;
//   3. sizes   every tile-local root points at its component's root and adds the size
//              of its tile-local component to the root's count (the other pixels stay
//              two steps from the root);
//   4. rank    surviving roots (count > threshold) are numbered by an exclusive
//              prefix count in raster order (block sums -> scan -> in-block
//              ballot/popcount), which is exactly the id the second
//              connected_components call of the reference would assign, because
//              deleting whole components does not change the others;
//   5. relabel label = newid[root].
#include "fs_common.h"

namespace fs {

namespace {

constexpr int CC_BLOCK = 1024;   // pixels per block in every kernel of this file

__device__ __forceinline__ int ld_parent(const int32_t *L, int i)
{
    // other workgroups update parents with atomics in the same launch: read at agent scope
    return __hip_atomic_load(L + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ int uf_find(const int32_t *L, int a)
{
    int p = ld_parent(L, a);
    while (p != a) {
        a = p;
        p = ld_parent(L, a);
    }
    return a;
}

__device__ __forceinline__ void uf_union(int32_t *L, int a, int b)
{
    bool done = false;
    while (!done) {
        a = uf_find(L, a);
        b = uf_find(L, b);
        if (a < b) {
            int old = atomicMin(L + b, a);
            done = (old == b);
            b = old;
        } else if (b < a) {
            int old = atomicMin(L + a, b);
            done = (old == a);
            a = old;
        } else {
            done = true;
        }
    }
}

// ---- components inside a 64 x 16 tile, in LDS -------------------------------------------------------
constexpr int LTW = 64, LTH = 16;            // one wave per tile row

__device__ __forceinline__ int lds_parent(const int *L, int i)
{
    return __hip_atomic_load(L + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ int lds_find(const int *L, int a)
{
    int p = lds_parent(L, a);
    while (p != a) {
        a = p;
        p = lds_parent(L, a);
    }
    return a;
}
__device__ __forceinline__ void lds_union(int *L, int a, int b)
{
    bool done = false;
    while (!done) {
        a = lds_find(L, a);
        b = lds_find(L, b);
        if (a < b) {
            int old = atomicMin(L + b, a);
            done = (old == b);
            b = old;
        } else if (b < a) {
            int old = atomicMin(L + a, b);
            done = (old == a);
            a = old;
        } else {
            done = true;
        }
    }
}

// T = uint8_t (binary mask) or int32_t (label image): pixels are connected when their values are
// equal and non-zero.
// Every pixel of a tile ends up pointing at the first pixel (raster order) of its component within the
// tile; the links across tile borders are left to cc_border_kernel.  Tile-local indices and flat frame
// indices order the pixels of a tile the same way, so the smallest index stays the root in both.
// SIZES: lsize receives the size of the tile-local component at its root and 0 elsewhere, and only the roots'
// counts are zeroed (cc_sum_sizes_kernel then works on the local roots only); otherwise every pixel's count is
// zeroed for cc_flatten_count_kernel.
template <typename T, bool SIZES>
__global__ __launch_bounds__(LTW * LTH) void cc_local_kernel(const T *__restrict__ mask, int32_t *__restrict__ parent,
                                                             int32_t *__restrict__ count, int32_t *__restrict__ lsize, int H, int W)
{
    __shared__ int lab[LTW * LTH];
    __shared__ int val[LTW * LTH];
    const size_t fo = (size_t)blockIdx.z * H * W;
    const int lane = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int x0 = blockIdx.x * LTW, y0 = blockIdx.y * LTH;
    const int x = x0 + lane, y = y0 + ty;
    const bool in = x < W && y < H;
    const int m = in ? (int)mask[fo + (size_t)y * W + x] : 0;
    const int l = ty * LTW + lane;
    // row runs by ballot: a lane's parent is the start of its horizontal run
    const int mprev = __shfl_up(m, 1);
    const bool start = m && (lane == 0 || mprev != m);
    const unsigned long long sb = __ballot(start);
    int par = -1;
    if (m) {
        const unsigned long long below = sb & ((2ull << lane) - 1ull);   // starts at lanes <= lane
        par = ty * LTW + (63 - __clzll(below));
    }
    lab[l] = par;
    val[l] = m;
    __syncthreads();
    if (m && ty > 0 && val[l - LTW] == m) {
        const bool covered = lane > 0 && val[l - 1] == m && val[l - LTW - 1] == m;   // the left pixel makes the same link
        if (!covered) lds_union(lab, l, l - LTW);
    }
    __syncthreads();
    const int rl = m ? lds_find(lab, l) : -1;
    if (in) {
        parent[fo + (size_t)y * W + x] = m ? (y0 + rl / LTW) * W + x0 + (rl & (LTW - 1)) : -1;
        if (!SIZES && count) count[fo + (size_t)y * W + x] = 0;
    }
    if (SIZES) {
        // sizes of the tile-local components: one LDS add per distinct root of a wave (a row holds few)
        __syncthreads();
        val[l] = 0;
        __syncthreads();
        unsigned long long todo = __ballot(rl >= 0);
        while (todo) {
            const int leader = __ffsll((long long)todo) - 1;
            const int r = __shfl(rl, leader);
            const unsigned long long same = __ballot(rl == r) & todo;
            if (lane == leader) atomicAdd(val + r, __popcll(same));
            todo &= ~same;
        }
        __syncthreads();
        if (in) lsize[fo + (size_t)y * W + x] = rl == l ? val[l] : 0;      // > 0 marks a tile-local root
        if (rl == l) count[fo + (size_t)y * W + x] = 0;
    }
}

// Links across the tile borders of cc_local_kernel: lock-free union-find with atomicMin on the flat
// indices (the root of a tree is always its smallest index, i.e. the component's first pixel in raster
// order).  One lane per pixel of a tile's top row (link to the pixel above) or left column (link to the
// pixel on the left); a link is skipped when the neighbouring border pixel of the same tile already makes it.
template <typename T>
__global__ __launch_bounds__(256) void cc_border_kernel(const T *__restrict__ mask, int32_t *__restrict__ parent, int H, int W)
{
    const int rows = (H - 1) / LTH, cols = (W - 1) / LTW;       // tile borders inside the frame
    const int t = blockIdx.x * 256 + threadIdx.x;
    const size_t fo = (size_t)blockIdx.y * H * W;
    const T *M = mask + fo;
    int32_t *L = parent + fo;
    if (t < rows * W) {
        const int y = (t / W + 1) * LTH, x = t % W, p = y * W + x;
        const T m = M[p];
        if (m && M[p - W] == m) {
            const bool covered = (x % LTW) != 0 && M[p - 1] == m && M[p - W - 1] == m;
            if (!covered) uf_union(L, p, p - W);
        }
    } else if (t < rows * W + cols * H) {
        const int u = t - rows * W;
        const int x = (u / H + 1) * LTW, y = u % H, p = y * W + x;
        const T m = M[p];
        if (m && M[p - 1] == m) {
            const bool covered = (y % LTH) != 0 && M[p - W] == m && M[p - W - 1] == m;
            if (!covered) uf_union(L, p, p - 1);
        }
    }
}

// After the border links: every tile-local root looks up its component's root, points at it directly and adds
// its local size to the root's count.  The other pixels keep pointing at their local root, which is one step
// from the root afterwards (cc_relabel_kernel takes the two steps).
__global__ __launch_bounds__(CC_BLOCK) void cc_sum_sizes_kernel(int32_t *__restrict__ parent, int32_t *__restrict__ count,
                                                                const int32_t *__restrict__ lsize, int N)
{
    const size_t fo = (size_t)blockIdx.y * N;
    const int p = blockIdx.x * CC_BLOCK + threadIdx.x;
    if (p >= N) return;
    const int ls = lsize[fo + p];
    if (ls <= 0) return;
    // only roots were changed by the border links (to smaller indices), so the way up leads over former roots
    int a = p;
    int q = ld_parent(parent + fo, a);
    while (q != a) { a = q; q = ld_parent(parent + fo, a); }
    if (a != p) parent[fo + p] = a;
    atomicAdd(count + fo + a, ls);
}

// every pixel points at its component's root afterwards (no sizes: for callers that only compare roots)
__global__ __launch_bounds__(CC_BLOCK) void cc_flatten_kernel(int32_t *__restrict__ parent, int N)
{
    const size_t fo = (size_t)blockIdx.y * N;
    const int p = blockIdx.x * CC_BLOCK + threadIdx.x;
    if (p >= N) return;
    const int par = parent[fo + p];
    if (par < 0) return;
    int a = par, q = parent[fo + a];
    while (q != a) { a = q; q = parent[fo + a]; }
    if (a != par) parent[fo + p] = a;           // (writes only shorten paths: concurrent readers still reach the root)
}

__global__ __launch_bounds__(CC_BLOCK) void cc_flatten_count_kernel(int32_t *__restrict__ parent, int32_t *__restrict__ count, int N)
{
    const size_t fo = (size_t)blockIdx.y * N;
    const int p = blockIdx.x * CC_BLOCK + threadIdx.x;
    int root = -1;
    if (p < N) {
        int par = parent[fo + p];
        if (par >= 0) {
            // parents written by the previous launch: plain loads suffice; every
            // pixel ends up pointing at a root, and writes only shorten paths
            int a = par;
            int q = parent[fo + a];
            while (q != a) { a = q; q = parent[fo + a]; }
            root = a;
            parent[fo + p] = root;
        }
    }
    // one atomicAdd per distinct root per wave
    unsigned long long todo = __ballot(root >= 0);
    const int lane = threadIdx.x & 63;
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const int r = __shfl(root, leader);
        const unsigned long long same = __ballot(root == r) & todo;
        if (lane == leader) atomicAdd(count + fo + r, __popcll(same));
        todo &= ~same;
    }
}

__device__ __forceinline__ bool is_survivor(const int32_t *parent, const int32_t *count, size_t i, int p, int thr)
{
    return parent[i] == p && count[i] > thr;
}

__global__ __launch_bounds__(CC_BLOCK) void cc_blocksum_kernel(const int32_t *__restrict__ parent, const int32_t *__restrict__ count,
                                                               int32_t *__restrict__ blocksum, int N, int thr)
{
    __shared__ int wsum[CC_BLOCK / 64];
    const size_t fo = (size_t)blockIdx.y * N;
    const int p = blockIdx.x * CC_BLOCK + threadIdx.x;
    const bool s = p < N && is_survivor(parent, count, fo + p, p, thr);
    const unsigned long long b = __ballot(s);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = __popcll(b);
    __syncthreads();
    if (threadIdx.x == 0) {
        int t = 0;
#pragma unroll
        for (int i = 0; i < CC_BLOCK / 64; ++i) t += wsum[i];
        blocksum[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = t;
    }
}

// one block per image: exclusive scan of the per-block survivor counts
__global__ __launch_bounds__(1024) void cc_scan_kernel(int32_t *__restrict__ blocksum, int32_t *__restrict__ nlabels,
                                                       int32_t *__restrict__ clear, int nblk)
{
    __shared__ int part[1024];
    int32_t *bs = blocksum + (size_t)blockIdx.x * nblk;
    const int per = (nblk + 1023) / 1024;
    const int b0 = threadIdx.x * per;
    int t = 0;
    for (int i = 0; i < per; ++i)
        if (b0 + i < nblk) t += bs[b0 + i];
    part[threadIdx.x] = t;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {   // Hillis-Steele inclusive scan
        int v = (int)threadIdx.x >= off ? part[threadIdx.x - off] : 0;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    int run = part[threadIdx.x] - t;             // exclusive prefix of this thread's chunk
    for (int i = 0; i < per; ++i)
        if (b0 + i < nblk) {
            int v = bs[b0 + i];
            bs[b0 + i] = run;
            run += v;
        }
    if (threadIdx.x == 1023) {
        nlabels[blockIdx.x] = part[1023];
        if (clear) clear[blockIdx.x] = 0;
    }
}

template <int NT>
__device__ __forceinline__ int blocks_in_front(const int32_t *__restrict__ bs, int nblk_before, int *s_part);

template <bool SCAN>
__global__ __launch_bounds__(CC_BLOCK) void cc_rank_kernel(const int32_t *__restrict__ parent, const int32_t *__restrict__ count,
                                                           const int32_t *__restrict__ blocksum, int32_t *__restrict__ newid,
                                                           int N, int thr, int32_t *__restrict__ nlabels, int32_t *__restrict__ clear)
{
    __shared__ int wsum[CC_BLOCK / 64];
    __shared__ int s_part[CC_BLOCK / 64];
    const int32_t *bs = blocksum + (size_t)blockIdx.y * gridDim.x;
    const int front = SCAN ? blocks_in_front<CC_BLOCK>(bs, blockIdx.x, s_part) : bs[blockIdx.x];
    if (SCAN && blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
        nlabels[blockIdx.y] = front + bs[blockIdx.x];
        if (clear) clear[blockIdx.y] = 0;
    }
    const size_t fo = (size_t)blockIdx.y * N;
    const int p = blockIdx.x * CC_BLOCK + threadIdx.x;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const bool s = p < N && is_survivor(parent, count, fo + p, p, thr);
    const unsigned long long b = __ballot(s);
    if (lane == 0) wsum[wv] = __popcll(b);
    __syncthreads();
    int before = front;
    for (int i = 0; i < wv; ++i) before += wsum[i];
    before += __popcll(b & ((1ull << lane) - 1ull));
    if (p < N) newid[fo + p] = s ? before + 1 : 0;
}

__global__ __launch_bounds__(CC_BLOCK) void cc_relabel_kernel(const int32_t *__restrict__ parent, const int32_t *__restrict__ newid,
                                                              int64_t *__restrict__ labels64, int32_t *__restrict__ labels32, int N)
{
    const size_t fo = (size_t)blockIdx.y * N;
    const int p = blockIdx.x * CC_BLOCK + threadIdx.x;
    if (p >= N) return;
    int root = parent[fo + p];                          // tile-local root, itself one step from the root
    if (root >= 0) root = parent[fo + root];
    const int id = root >= 0 ? newid[fo + root] : 0;
    if (labels64) labels64[fo + p] = id;
    if (labels32) labels32[fo + p] = id;
}

// ---- the same three passes with four consecutive pixels per lane (16-byte loads and stores) for frames whose
// pixel count is a multiple of four; the block still covers CC_BLOCK pixels, so the block sums and their scan are
// those of the one-pixel kernels ----
constexpr int CC_VT = CC_BLOCK / 4;     // threads per block of the four-pixel kernels

__device__ __forceinline__ void survivors4(const int32_t *parent, const int32_t *count, size_t i, int p, int thr, bool (&s)[4])
{
    const int4 par = *reinterpret_cast<const int4 *>(parent + i);
    const int4 cnt = *reinterpret_cast<const int4 *>(count + i);   // meaningful at roots only; the memory exists everywhere
    s[0] = par.x == p && cnt.x > thr;
    s[1] = par.y == p + 1 && cnt.y > thr;
    s[2] = par.z == p + 2 && cnt.z > thr;
    s[3] = par.w == p + 3 && cnt.w > thr;
}

__global__ __launch_bounds__(CC_VT) void cc_sum_sizes4_kernel(int32_t *__restrict__ parent, int32_t *__restrict__ count,
                                                              const int32_t *__restrict__ lsize, int N)
{
    const size_t fo = (size_t)blockIdx.y * N;
    const int p = blockIdx.x * CC_BLOCK + threadIdx.x * 4;
    if (p >= N) return;
    const int4 l4 = *reinterpret_cast<const int4 *>(lsize + fo + p);
    const int ls[4] = {l4.x, l4.y, l4.z, l4.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (ls[k] <= 0) continue;
        int a = p + k;
        int q = ld_parent(parent + fo, a);
        while (q != a) { a = q; q = ld_parent(parent + fo, a); }
        if (a != p + k) parent[fo + p + k] = a;
        atomicAdd(count + fo + a, ls[k]);
    }
}

__global__ __launch_bounds__(CC_VT) void cc_blocksum4_kernel(const int32_t *__restrict__ parent, const int32_t *__restrict__ count,
                                                             int32_t *__restrict__ blocksum, int N, int thr)
{
    __shared__ int wsum[CC_VT / 64];
    const size_t fo = (size_t)blockIdx.y * N;
    const int p = blockIdx.x * CC_BLOCK + threadIdx.x * 4;
    bool s[4] = {false, false, false, false};
    if (p < N) survivors4(parent, count, fo + p, p, thr, s);
    int t = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) t += __popcll(__ballot(s[k]));
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = t;
    __syncthreads();
    if (threadIdx.x == 0) {
        int tot = 0;
#pragma unroll
        for (int i = 0; i < CC_VT / 64; ++i) tot += wsum[i];
        blocksum[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = tot;
    }
}

// SCAN: blocksum holds the blocks' own survivor counts and every block adds up those of the blocks in front of it
// itself (a frame has a few hundred blocks: cheaper than a scan kernel between two launches); the frame's last block
// also writes the label count and clears clear[frame].  Otherwise blocksum holds the exclusive prefix (cc_scan_kernel).
template <int NT>
__device__ __forceinline__ int blocks_in_front(const int32_t *__restrict__ bs, int nblk_before, int *s_part)
{
    int t = 0;
    for (int b = threadIdx.x; b < nblk_before; b += NT) t += bs[b];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) t += __shfl_down(t, off);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = t;
    __syncthreads();
    int sum = 0;
#pragma unroll
    for (int i = 0; i < NT / 64; ++i) sum += s_part[i];
    return sum;
}

template <bool SCAN>
__global__ __launch_bounds__(CC_VT) void cc_rank4_kernel(const int32_t *__restrict__ parent, const int32_t *__restrict__ count,
                                                         const int32_t *__restrict__ blocksum, int32_t *__restrict__ newid, int N,
                                                         int thr, int32_t *__restrict__ nlabels, int32_t *__restrict__ clear)
{
    __shared__ int wsum[CC_VT / 64];
    __shared__ int s_part[CC_VT / 64];
    const size_t fo = (size_t)blockIdx.y * N;
    const int p = blockIdx.x * CC_BLOCK + threadIdx.x * 4;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int32_t *bs = blocksum + (size_t)blockIdx.y * gridDim.x;
    const int front = SCAN ? blocks_in_front<CC_VT>(bs, blockIdx.x, s_part) : bs[blockIdx.x];
    if (SCAN && blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
        nlabels[blockIdx.y] = front + bs[blockIdx.x];
        if (clear) clear[blockIdx.y] = 0;
    }
    bool s[4] = {false, false, false, false};
    if (p < N) survivors4(parent, count, fo + p, p, thr, s);
    const unsigned long long low = (1ull << lane) - 1ull;
    int t = 0, mine = 0;                        // survivors of the wave, and of the lanes below this one
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const unsigned long long b = __ballot(s[k]);
        t += __popcll(b);
        mine += __popcll(b & low);
    }
    if (lane == 0) wsum[wv] = t;
    __syncthreads();
    int run = front + mine;
    for (int i = 0; i < wv; ++i) run += wsum[i];
    if (p < N) {
        int4 id;
        id.x = s[0] ? ++run : 0;
        id.y = s[1] ? ++run : 0;
        id.z = s[2] ? ++run : 0;
        id.w = s[3] ? ++run : 0;
        *reinterpret_cast<int4 *>(newid + fo + p) = id;
    }
}

__global__ __launch_bounds__(CC_VT) void cc_relabel4_kernel(const int32_t *__restrict__ parent, const int32_t *__restrict__ newid,
                                                            int64_t *__restrict__ labels64, int32_t *__restrict__ labels32, int N)
{
    const size_t fo = (size_t)blockIdx.y * N;
    const int p = blockIdx.x * CC_BLOCK + threadIdx.x * 4;
    if (p >= N) return;
    const int4 par = *reinterpret_cast<const int4 *>(parent + fo + p);
    int root[4] = {par.x, par.y, par.z, par.w}, id[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (root[k] >= 0) root[k] = parent[fo + root[k]];      // tile-local root -> root
#pragma unroll
    for (int k = 0; k < 4; ++k) id[k] = root[k] >= 0 ? newid[fo + root[k]] : 0;
    if (labels64) {
        longlong2 *o = reinterpret_cast<longlong2 *>(labels64 + fo + p);
        o[0] = make_longlong2(id[0], id[1]);
        o[1] = make_longlong2(id[2], id[3]);
    }
    if (labels32) *reinterpret_cast<int4 *>(labels32 + fo + p) = make_int4(id[0], id[1], id[2], id[3]);
}

}  // namespace

template <typename T, bool SIZES>
static void launch_local_and_border(const T *mask, int32_t *parent, int32_t *count, int32_t *lsize, int n, int H, int W, hipStream_t st)
{
    hipLaunchKernelGGL((cc_local_kernel<T, SIZES>), dim3((W + LTW - 1) / LTW, (H + LTH - 1) / LTH, n), dim3(LTW * LTH), 0, st, mask, parent,
                       count, lsize, H, W);
    const int links = ((H - 1) / LTH) * W + ((W - 1) / LTW) * H;
    if (links > 0)
        hipLaunchKernelGGL(cc_border_kernel<T>, dim3((links + 255) / 256, n), dim3(256), 0, st, mask, parent, H, W);
}

size_t cc_blocks_per_image(int H, int W) { return ((size_t)H * W + CC_BLOCK - 1) / CC_BLOCK; }

void launch_connected_components(const uint8_t *mask, const CcWork &w, int min_size_exclusive, int64_t *labels64,
                                 int32_t *labels32, int n, int H, int W, hipStream_t st)
{
    const int N = H * W;
    const int nblk = (int)cc_blocks_per_image(H, W);
    const dim3 grid(nblk, n), block(CC_BLOCK);
    launch_local_and_border<uint8_t, true>(mask, w.parent, w.count, w.newid, n, H, W, st);   // newid holds the local sizes until cc_rank
    // frames of up to a few thousand blocks: every block of the ranking pass adds up the counts in front of it itself
    const bool own_scan = nblk <= 4096;
    // four pixels per lane when every frame starts on a 16-byte boundary in all arrays
    const bool vec = N % 4 == 0 && reinterpret_cast<uintptr_t>(w.parent) % 16 == 0 && reinterpret_cast<uintptr_t>(w.count) % 16 == 0 &&
                     reinterpret_cast<uintptr_t>(w.newid) % 16 == 0 && reinterpret_cast<uintptr_t>(labels64) % 16 == 0 &&
                     reinterpret_cast<uintptr_t>(labels32) % 16 == 0;
    if (vec) {
        const dim3 block4(CC_VT);
        hipLaunchKernelGGL(cc_sum_sizes4_kernel, grid, block4, 0, st, w.parent, w.count, w.newid, N);
        hipLaunchKernelGGL(cc_blocksum4_kernel, grid, block4, 0, st, w.parent, w.count, w.blocksum, N, min_size_exclusive);
        if (own_scan) {
            hipLaunchKernelGGL(cc_rank4_kernel<true>, grid, block4, 0, st, w.parent, w.count, w.blocksum, w.newid, N, min_size_exclusive, w.nlabels, w.clear);
        } else {
            hipLaunchKernelGGL(cc_scan_kernel, dim3(n), dim3(1024), 0, st, w.blocksum, w.nlabels, w.clear, nblk);
            hipLaunchKernelGGL(cc_rank4_kernel<false>, grid, block4, 0, st, w.parent, w.count, w.blocksum, w.newid, N, min_size_exclusive, nullptr, nullptr);
        }
        hipLaunchKernelGGL(cc_relabel4_kernel, grid, block4, 0, st, w.parent, w.newid, labels64, labels32, N);
        return;
    }
    hipLaunchKernelGGL(cc_sum_sizes_kernel, grid, block, 0, st, w.parent, w.count, w.newid, N);
    hipLaunchKernelGGL(cc_blocksum_kernel, grid, block, 0, st, w.parent, w.count, w.blocksum, N, min_size_exclusive);
    if (own_scan) {
        hipLaunchKernelGGL(cc_rank_kernel<true>, grid, block, 0, st, w.parent, w.count, w.blocksum, w.newid, N, min_size_exclusive, w.nlabels, w.clear);
    } else {
        hipLaunchKernelGGL(cc_scan_kernel, dim3(n), dim3(1024), 0, st, w.blocksum, w.nlabels, w.clear, nblk);
        hipLaunchKernelGGL(cc_rank_kernel<false>, grid, block, 0, st, w.parent, w.count, w.blocksum, w.newid, N, min_size_exclusive, nullptr, nullptr);
    }
    hipLaunchKernelGGL(cc_relabel_kernel, grid, block, 0, st, w.parent, w.newid, labels64, labels32, N);
}

void launch_label_components(const int32_t *labels, int32_t *parent, int32_t *count, int n, int H, int W, hipStream_t st)
{
    const int N = H * W;
    const dim3 grid((unsigned)cc_blocks_per_image(H, W), n), block(CC_BLOCK);
    launch_local_and_border<int32_t, false>(labels, parent, count, nullptr, n, H, W, st);
    if (count) hipLaunchKernelGGL(cc_flatten_count_kernel, grid, block, 0, st, parent, count, N);
    else hipLaunchKernelGGL(cc_flatten_kernel, grid, block, 0, st, parent, N);
}

}  // namespace fs
