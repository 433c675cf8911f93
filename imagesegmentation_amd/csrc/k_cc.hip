// Connected-component labelling of the smooth-patch mask (gfx950).
//
// Replaces the two tfa.image.connected_components calls and the bincount size
// filter of find_surfaces_model_GPU (find_surfaces.py:281-285,
// image_processing_models_GPU.py:74-80): 4-connectivity, ids 1..n in row-major
// order of each component's first pixel, components of <= 20 pixels removed.
//
// Algorithm (all integer, schedule independent):
//   1. init    each wave labels 64 consecutive pixels of a row: a ballot of the
//              mask gives every lane the start of its horizontal run, which
//              becomes its parent (so horizontal unions inside a wave are free);
//   2. merge   lock-free union-find with atomicMin: runs that continue across a
//              wave boundary, and vertical neighbours (skipped when the pixel to
//              the left already carries the same vertical link);
//              the root of a tree is always its smallest flat index, i.e. the
//              component's first pixel in raster order;
//   3. flatten every pixel points at its root; per-root pixel counts with one
//              atomicAdd per distinct root per wave;
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

// T = uint8_t (binary mask) or int32_t (label image): pixels are connected when their values are
// equal and non-zero.
template <typename T>
__global__ __launch_bounds__(CC_BLOCK) void cc_init_kernel(const T *__restrict__ mask, int32_t *__restrict__ parent,
                                                           int32_t *__restrict__ count, int N, int W)
{
    const size_t fo = (size_t)blockIdx.y * N;
    const int p = blockIdx.x * CC_BLOCK + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const bool in = p < N;
    const int m = in ? (int)mask[fo + p] : 0;
    const int x = in ? p % W : 0;
    const int mprev = __shfl_up(m, 1);
    const bool start = m && (lane == 0 || x == 0 || mprev != m);
    const unsigned long long sb = __ballot(start);
    if (in) {
        int par = -1;
        if (m) {
            unsigned long long below = sb & ((2ull << lane) - 1ull);   // starts at lanes <= lane
            int s = 63 - __clzll(below);
            par = p - (lane - s);
        }
        parent[fo + p] = par;
        count[fo + p] = 0;
    }
}

template <typename T>
__global__ __launch_bounds__(CC_BLOCK) void cc_merge_kernel(const T *__restrict__ mask, int32_t *__restrict__ parent,
                                                            int N, int W)
{
    const size_t fo = (size_t)blockIdx.y * N;
    const int p = blockIdx.x * CC_BLOCK + threadIdx.x;
    if (p >= N) return;
    const T *M = mask + fo;
    int32_t *L = parent + fo;
    const T m = M[p];
    if (!m) return;
    const int x = p % W;
    const bool left = x > 0 && M[p - 1] == m;
    if (left && (threadIdx.x & 63) == 0) uf_union(L, p, p - 1);   // run continues across the wave boundary
    if (p >= W && M[p - W] == m) {
        const bool covered = left && M[p - W - 1] == m;            // the left pixel makes the same vertical link
        if (!covered) uf_union(L, p, p - W);
    }
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
__global__ __launch_bounds__(1024) void cc_scan_kernel(int32_t *__restrict__ blocksum, int32_t *__restrict__ nlabels, int nblk)
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
    if (threadIdx.x == 1023) nlabels[blockIdx.x] = part[1023];
}

__global__ __launch_bounds__(CC_BLOCK) void cc_rank_kernel(const int32_t *__restrict__ parent, const int32_t *__restrict__ count,
                                                           const int32_t *__restrict__ blocksum, int32_t *__restrict__ newid,
                                                           int N, int thr)
{
    __shared__ int wsum[CC_BLOCK / 64];
    const size_t fo = (size_t)blockIdx.y * N;
    const int p = blockIdx.x * CC_BLOCK + threadIdx.x;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const bool s = p < N && is_survivor(parent, count, fo + p, p, thr);
    const unsigned long long b = __ballot(s);
    if (lane == 0) wsum[wv] = __popcll(b);
    __syncthreads();
    int before = blocksum[(size_t)blockIdx.y * gridDim.x + blockIdx.x];
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
    const int root = parent[fo + p];
    const int id = root >= 0 ? newid[fo + root] : 0;
    if (labels64) labels64[fo + p] = id;
    if (labels32) labels32[fo + p] = id;
}

}  // namespace

size_t cc_blocks_per_image(int H, int W) { return ((size_t)H * W + CC_BLOCK - 1) / CC_BLOCK; }

void launch_connected_components(const uint8_t *mask, const CcWork &w, int min_size_exclusive, int64_t *labels64,
                                 int32_t *labels32, int n, int H, int W, hipStream_t st)
{
    const int N = H * W;
    const int nblk = (int)cc_blocks_per_image(H, W);
    const dim3 grid(nblk, n), block(CC_BLOCK);
    hipLaunchKernelGGL(cc_init_kernel<uint8_t>, grid, block, 0, st, mask, w.parent, w.count, N, W);
    hipLaunchKernelGGL(cc_merge_kernel<uint8_t>, grid, block, 0, st, mask, w.parent, N, W);
    hipLaunchKernelGGL(cc_flatten_count_kernel, grid, block, 0, st, w.parent, w.count, N);
    hipLaunchKernelGGL(cc_blocksum_kernel, grid, block, 0, st, w.parent, w.count, w.blocksum, N, min_size_exclusive);
    hipLaunchKernelGGL(cc_scan_kernel, dim3(n), dim3(1024), 0, st, w.blocksum, w.nlabels, nblk);
    hipLaunchKernelGGL(cc_rank_kernel, grid, block, 0, st, w.parent, w.count, w.blocksum, w.newid, N, min_size_exclusive);
    hipLaunchKernelGGL(cc_relabel_kernel, grid, block, 0, st, w.parent, w.newid, labels64, labels32, N);
}

void launch_label_components(const int32_t *labels, int32_t *parent, int32_t *count, int n, int H, int W, hipStream_t st)
{
    const int N = H * W;
    const dim3 grid((unsigned)cc_blocks_per_image(H, W), n), block(CC_BLOCK);
    hipLaunchKernelGGL(cc_init_kernel<int32_t>, grid, block, 0, st, labels, parent, count, N, W);
    hipLaunchKernelGGL(cc_merge_kernel<int32_t>, grid, block, 0, st, labels, parent, N, W);
    hipLaunchKernelGGL(cc_flatten_count_kernel, grid, block, 0, st, parent, count, N);
}

}  // namespace fs
