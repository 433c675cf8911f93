// Shader clock the chip sustains under the CRF window walk's instruction mix (gfx950): the cycle counter (s_memtime, shader
// clocks) against the constant 100 MHz counter (s_memrealtime) around a long run of the mix of valu_mix.hip, with 1 block on
// the chip, with one wave per SIMD on every CU and with two.   hipcc --offload-arch=gfx950 -O3 clock_rate.hip -o clock_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void k(unsigned long long *out, float seed, int iters)
{
    float e0 = seed + threadIdx.x * 1e-3f, e1 = e0 + 1, e2 = e0 + 2, e3 = e0 + 3, e4 = e0 + 4, e5 = e0 + 5, e6 = e0 + 6, e7 = e0 + 7;
    f2 p0 = {e0, e1}, p1 = {e2, e3}, p2 = {e4, e5}, p3 = {e6, e7}, p4 = p0 * 2.f, p5 = p1 * 2.f, p6 = p2 * 2.f, p7 = p3 * 2.f;
    const f2 c = {0.999f, 1.001f}, d = {1e-3f, -1e-3f};
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter(), r0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
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
    }
    const unsigned long long t1 = __builtin_readcyclecounter(), r1 = wall_clock64();
    float sink = e0 + e1 + e2 + e3 + e4 + e5 + e6 + e7 + p0.x + p1.x + p2.x + p3.x + p4.y + p5.y + p6.y + p7.y;
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = t1 - t0; out[2 * blockIdx.x + 1] = r1 - r0; }
    if (sink == 12345.f) out[0] = 0;
}

static void run(const char *what, int blocks, int iters)
{
    unsigned long long *d;
    hipMalloc(&d, 2 * blocks * sizeof(*d));
    k<<<blocks, 256>>>(d, 0.5f, 10);
    hipDeviceSynchronize();
    int wall_khz = 0;
    hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0);
    for (int rep = 0; rep < 3; ++rep) {
        k<<<blocks, 256>>>(d, 0.5f, iters);
        std::vector<unsigned long long> h(2 * blocks);
        hipMemcpy(h.data(), d, 2 * blocks * sizeof(*d), hipMemcpyDeviceToHost);
        double cyc = 0, wall = 0;
        for (int b = 0; b < blocks; ++b) { cyc += (double)h[2 * b]; wall += (double)h[2 * b + 1]; }
        const double seconds = wall / blocks / (wall_khz * 1e3);
        printf("%-34s run %d: %7.1f ms, %8.1f cycle-counter ticks per us, %6.2f ticks per instruction of one wave\n", what, rep, seconds * 1e3,
               cyc / blocks / (seconds * 1e6), cyc / blocks / iters / 32.0);
    }
    hipFree(d);
}

int main()
{
    int clk_khz = 0, wall_khz = 0;
    hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
    hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0);
    printf("device clock rate attribute %d kHz, wall clock rate %d kHz\n", clk_khz, wall_khz);
    run("one block on the chip", 1, 400000);
    run("one wave per SIMD on every CU", 256, 400000);
    run("two waves per SIMD on every CU", 512, 200000);
    return 0;
}
