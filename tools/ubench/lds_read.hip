// LDS read throughput microbenchmark (gfx950): ds_read_b128 / b64 with lane-consecutive addresses, alone and next to
// packed VALU work, one or two waves per SIMD.   hipcc --offload-arch=gfx950 -O3 lds_read.hip -o lds_read
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(unsigned long long *out, float seed, int iters)
{
    extern __shared__ float4 lds[];
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = make_float4(seed, seed, seed, seed);
    __syncthreads();
    unsigned a0 = (threadIdx.x * 16) & 0xffff, a1 = a0 + 16 * 300, a2 = a0 + 16 * 700, a3 = a0 + 16 * 1100;
    float4 r0, r1, r2, r3;
    f2 p0 = {seed, 1}, p1 = {seed, 2}, p2 = {seed, 3}, p3 = {seed, 4}, p4 = p0, p5 = p1, p6 = p2, p7 = p3;
    const f2 c = {0.999f, 1.001f}, d = {1e-3f, -1e-3f};
    float acc = 0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {            // 4 x ds_read_b128 per iteration, nothing else
            asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %5\n ds_read_b128 %2, %6\n ds_read_b128 %3, %7\n s_waitcnt lgkmcnt(0)"
                         : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "memory");
        } else if (MODE == 1) {     // the walk's reads (b128, b64, b128, b128) + 32 packed ops of VALU work
            asm volatile("ds_read_b128 %0, %12\n ds_read_b64 %1, %13\n ds_read_b128 %2, %14\n ds_read_b128 %3, %15\n"
                         "v_pk_fma_f32 %4, %4, %16, %17\n v_pk_fma_f32 %5, %5, %16, %17\n v_pk_fma_f32 %6, %6, %16, %17\n v_pk_fma_f32 %7, %7, %16, %17\n"
                         "v_pk_fma_f32 %8, %8, %16, %17\n v_pk_fma_f32 %9, %9, %16, %17\n v_pk_fma_f32 %10, %10, %16, %17\n v_pk_fma_f32 %11, %11, %16, %17\n"
                         "v_pk_fma_f32 %4, %4, %16, %17\n v_pk_fma_f32 %5, %5, %16, %17\n v_pk_fma_f32 %6, %6, %16, %17\n v_pk_fma_f32 %7, %7, %16, %17\n"
                         "v_pk_fma_f32 %8, %8, %16, %17\n v_pk_fma_f32 %9, %9, %16, %17\n v_pk_fma_f32 %10, %10, %16, %17\n v_pk_fma_f32 %11, %11, %16, %17\n"
                         "v_pk_fma_f32 %4, %4, %16, %17\n v_pk_fma_f32 %5, %5, %16, %17\n v_pk_fma_f32 %6, %6, %16, %17\n v_pk_fma_f32 %7, %7, %16, %17\n"
                         "v_pk_fma_f32 %8, %8, %16, %17\n v_pk_fma_f32 %9, %9, %16, %17\n v_pk_fma_f32 %10, %10, %16, %17\n v_pk_fma_f32 %11, %11, %16, %17\n"
                         "v_pk_fma_f32 %4, %4, %16, %17\n v_pk_fma_f32 %5, %5, %16, %17\n v_pk_fma_f32 %6, %6, %16, %17\n v_pk_fma_f32 %7, %7, %16, %17\n"
                         "v_pk_fma_f32 %8, %8, %16, %17\n v_pk_fma_f32 %9, %9, %16, %17\n v_pk_fma_f32 %10, %10, %16, %17\n v_pk_fma_f32 %11, %11, %16, %17\n"
                         "s_waitcnt lgkmcnt(0)"
                         : "=v"(r0), "=v"(*(f2 *)&r1), "=v"(r2), "=v"(r3), "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7)
                         : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(c), "v"(d) : "memory");
        } else if (MODE == 2) {     // the 32 packed ops alone
            asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                         "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
                         "v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                         "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
                         "v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                         "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
                         "v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                         "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(c), "v"(d));
        }
        if (MODE != 2) acc += r0.x + r1.x + r2.x + r3.x;
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (acc + p0.x + p1.x + p2.x + p3.x + p4.x + p5.x + p6.x + p7.x == 12345.f) out[0] = 0;
}

template <int MODE>
static void run(const char *what, int blocks_per_cu)
{
    unsigned long long *d;
    const int blocks = 256 * blocks_per_cu, iters = 2000;
    hipMalloc(&d, blocks * sizeof(*d));
    k<MODE><<<blocks, 256, 65536>>>(d, 0.5f, 10);
    k<MODE><<<blocks, 256, 65536>>>(d, 0.5f, iters);
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), d, blocks * sizeof(*d), hipMemcpyDeviceToHost);
    double s = 0;
    for (auto v : h) s += (double)v;
    printf("%-64s %d block(s)/CU: %7.1f cycles per iteration of one wave\n", what, blocks_per_cu, s / blocks / iters);
    hipFree(d);
}

int main()
{
    hipFuncSetAttribute((const void *)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipFuncSetAttribute((const void *)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipFuncSetAttribute((const void *)k<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    for (int w = 1; w <= 2; ++w) {
        run<0>("4 x ds_read_b128 (4 KB per wave)", w);
        run<2>("32 x v_pk_fma_f32", w);
        run<1>("b128 + b64 + b128 + b128 (3.5 KB per wave) beside 32 x v_pk_fma_f32", w);
    }
    return 0;
}
