// Issue-rate microbenchmark (gfx950): cycles per instruction of v_exp_f32, v_rcp_f32, v_pk_fma_f32 and of the mix the CRF
// window walk executes, with one or two waves per SIMD.   hipcc --offload-arch=gfx950 -O3 valu_mix.hip -o valu_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(unsigned long long *out, float seed, int iters)
{
    float e0 = seed + threadIdx.x * 1e-3f, e1 = e0 + 1, e2 = e0 + 2, e3 = e0 + 3, e4 = e0 + 4, e5 = e0 + 5, e6 = e0 + 6, e7 = e0 + 7;
    f2 p0 = {e0, e1}, p1 = {e2, e3}, p2 = {e4, e5}, p3 = {e6, e7}, p4 = p0 * 2.f, p5 = p1 * 2.f, p6 = p2 * 2.f, p7 = p3 * 2.f;
    const f2 c = {0.999f, 1.001f}, d = {1e-3f, -1e-3f};
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {          // 8 independent v_exp
            asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7"
                         : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3), "+v"(e4), "+v"(e5), "+v"(e6), "+v"(e7));
        } else if (MODE == 1) {   // 8 independent v_pk_fma
            asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                         "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(c), "v"(d));
        } else if (MODE == 2) {   // the walk's mix: 8 transcendentals among 24 packed ops
            asm volatile("v_exp_f32 %8, %8\n v_pk_fma_f32 %0, %0, %16, %17\n v_pk_fma_f32 %1, %1, %16, %17\n v_pk_fma_f32 %2, %2, %16, %17\n"
                         "v_exp_f32 %9, %9\n v_pk_fma_f32 %3, %3, %16, %17\n v_pk_fma_f32 %4, %4, %16, %17\n v_pk_fma_f32 %5, %5, %16, %17\n"
                         "v_exp_f32 %10, %10\n v_pk_fma_f32 %6, %6, %16, %17\n v_pk_fma_f32 %7, %7, %16, %17\n v_pk_fma_f32 %0, %0, %16, %17\n"
                         "v_exp_f32 %11, %11\n v_pk_fma_f32 %1, %1, %16, %17\n v_pk_fma_f32 %2, %2, %16, %17\n v_pk_fma_f32 %3, %3, %16, %17\n"
                         "v_exp_f32 %12, %12\n v_pk_fma_f32 %4, %4, %16, %17\n v_pk_fma_f32 %5, %5, %16, %17\n v_pk_fma_f32 %6, %6, %16, %17\n"
                         "v_exp_f32 %13, %13\n v_pk_fma_f32 %7, %7, %16, %17\n v_pk_fma_f32 %0, %0, %16, %17\n v_pk_fma_f32 %1, %1, %16, %17\n"
                         "v_rcp_f32 %14, %14\n v_pk_fma_f32 %2, %2, %16, %17\n v_pk_fma_f32 %3, %3, %16, %17\n v_pk_fma_f32 %4, %4, %16, %17\n"
                         "v_rcp_f32 %15, %15\n v_pk_fma_f32 %5, %5, %16, %17\n v_pk_fma_f32 %6, %6, %16, %17\n v_pk_fma_f32 %7, %7, %16, %17"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7), "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3),
                           "+v"(e4), "+v"(e5), "+v"(e6), "+v"(e7)
                         : "v"(c), "v"(d));
        } else if (MODE == 3) {   // the same instructions, transcendentals back to back
            asm volatile("v_exp_f32 %8, %8\n v_exp_f32 %9, %9\n v_exp_f32 %10, %10\n v_exp_f32 %11, %11\n v_exp_f32 %12, %12\n v_exp_f32 %13, %13\n"
                         "v_rcp_f32 %14, %14\n v_rcp_f32 %15, %15\n"
                         "v_pk_fma_f32 %0, %0, %16, %17\n v_pk_fma_f32 %1, %1, %16, %17\n v_pk_fma_f32 %2, %2, %16, %17\n v_pk_fma_f32 %3, %3, %16, %17\n"
                         "v_pk_fma_f32 %4, %4, %16, %17\n v_pk_fma_f32 %5, %5, %16, %17\n v_pk_fma_f32 %6, %6, %16, %17\n v_pk_fma_f32 %7, %7, %16, %17\n"
                         "v_pk_fma_f32 %0, %0, %16, %17\n v_pk_fma_f32 %1, %1, %16, %17\n v_pk_fma_f32 %2, %2, %16, %17\n v_pk_fma_f32 %3, %3, %16, %17\n"
                         "v_pk_fma_f32 %4, %4, %16, %17\n v_pk_fma_f32 %5, %5, %16, %17\n v_pk_fma_f32 %6, %6, %16, %17\n v_pk_fma_f32 %7, %7, %16, %17\n"
                         "v_pk_fma_f32 %0, %0, %16, %17\n v_pk_fma_f32 %1, %1, %16, %17\n v_pk_fma_f32 %2, %2, %16, %17\n v_pk_fma_f32 %3, %3, %16, %17\n"
                         "v_pk_fma_f32 %4, %4, %16, %17\n v_pk_fma_f32 %5, %5, %16, %17\n v_pk_fma_f32 %6, %6, %16, %17\n v_pk_fma_f32 %7, %7, %16, %17"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7), "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3),
                           "+v"(e4), "+v"(e5), "+v"(e6), "+v"(e7)
                         : "v"(c), "v"(d));
        } else if (MODE == 5) {   // the walk's mix since the reciprocal went: 6 exponentials among 25 packed ops
            asm volatile("v_exp_f32 %8, %8\n v_pk_fma_f32 %0, %0, %14, %15\n v_pk_fma_f32 %1, %1, %14, %15\n v_pk_fma_f32 %2, %2, %14, %15\n v_pk_fma_f32 %3, %3, %14, %15\n"
                         "v_exp_f32 %9, %9\n v_pk_fma_f32 %4, %4, %14, %15\n v_pk_fma_f32 %5, %5, %14, %15\n v_pk_fma_f32 %6, %6, %14, %15\n v_pk_fma_f32 %7, %7, %14, %15\n"
                         "v_exp_f32 %10, %10\n v_pk_fma_f32 %0, %0, %14, %15\n v_pk_fma_f32 %1, %1, %14, %15\n v_pk_fma_f32 %2, %2, %14, %15\n v_pk_fma_f32 %3, %3, %14, %15\n"
                         "v_exp_f32 %11, %11\n v_pk_fma_f32 %4, %4, %14, %15\n v_pk_fma_f32 %5, %5, %14, %15\n v_pk_fma_f32 %6, %6, %14, %15\n v_pk_fma_f32 %7, %7, %14, %15\n"
                         "v_exp_f32 %12, %12\n v_pk_fma_f32 %0, %0, %14, %15\n v_pk_fma_f32 %1, %1, %14, %15\n v_pk_fma_f32 %2, %2, %14, %15\n v_pk_fma_f32 %3, %3, %14, %15\n"
                         "v_exp_f32 %13, %13\n v_pk_fma_f32 %4, %4, %14, %15\n v_pk_fma_f32 %5, %5, %14, %15\n v_pk_fma_f32 %6, %6, %14, %15\n v_pk_fma_f32 %7, %7, %14, %15\n"
                         "v_pk_fma_f32 %0, %0, %14, %15"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7), "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3),
                           "+v"(e4), "+v"(e5)
                         : "v"(c), "v"(d));
        } else if (MODE == 4) {   // plain v_fma_f32 x 8
            asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                         "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9"
                         : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3), "+v"(e4), "+v"(e5), "+v"(e6), "+v"(e7) : "v"(seed), "v"(seed));
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float sink = e0 + e1 + e2 + e3 + e4 + e5 + e6 + e7 + p0.x + p1.x + p2.x + p3.x + p4.y + p5.y + p6.y + p7.y;
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (sink == 12345.f) out[0] = 0;
}

template <int MODE>
static void run(const char *what, int insts, int blocks_per_cu)
{
    unsigned long long *d;
    const int blocks = 256 * blocks_per_cu, iters = 2000;
    hipMalloc(&d, blocks * sizeof(*d));
    k<MODE><<<blocks, 256>>>(d, 0.5f, 10);
    k<MODE><<<blocks, 256>>>(d, 0.5f, iters);
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), d, blocks * sizeof(*d), hipMemcpyDeviceToHost);
    double s = 0;
    for (auto v : h) s += (double)v;
    printf("%-52s %d wave(s)/SIMD: %7.2f cycles per instruction of one wave (%d per iteration)\n", what, blocks_per_cu, s / blocks / iters / insts, insts);
    hipFree(d);
}

int main()
{
    for (int w = 1; w <= 2; ++w) {
        run<4>("v_fma_f32 x8", 8, w);
        run<1>("v_pk_fma_f32 x8", 8, w);
        run<0>("v_exp_f32 x8", 8, w);
        run<2>("walk mix, transcendentals spread (8 + 24 packed)", 32, w);
        run<3>("walk mix, transcendentals back to back", 32, w);
        run<5>("walk mix without the reciprocals (6 + 25 packed)", 31, w);
    }
    return 0;
}
