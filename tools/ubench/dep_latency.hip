// Dependent-issue latency microbenchmark (gfx950): cycles per instruction of one wave when every instruction depends on
// the one N instructions before it (N chains), for v_fma_f32, v_pk_fma_f32, v_pk_add_f32, v_exp_f32 and v_rcp_f32.
//   hipcc --offload-arch=gfx950 -O3 dep_latency.hip -o dep_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))

template <int OP, int CHAINS>
__global__ __launch_bounds__(256) void k(unsigned long long *out, float seed, int iters)
{
    f2 p[16];
    float e[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { p[i] = f2{seed + i, seed - i}; e[i] = seed + i * 0.01f + threadIdx.x * 1e-4f; }
    const f2 c = {0.999f, 1.001f}, d = {1e-3f, -1e-3f};
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 64; ++u) {           // 64 instructions per iteration, round-robin over the chains
            const int i = u % CHAINS;
            if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(e[i]) : "v"(seed), "v"(seed));
            if (OP == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(c), "v"(d));
            if (OP == 2) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(d));
            if (OP == 3) asm volatile("v_exp_f32 %0, %0" : "+v"(e[i]));
            if (OP == 4) asm volatile("v_rcp_f32 %0, %0" : "+v"(e[i]));
            if (OP == 5) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(c));
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += p[i].x + p[i].y + e[i];
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (s == 12345.f) out[0] = 0;
}

template <int OP, int CHAINS>
static double run(int blocks_per_cu)
{
    unsigned long long *d;
    const int blocks = 256 * blocks_per_cu, iters = 400;
    hipMalloc(&d, blocks * sizeof(*d));
    k<OP, CHAINS><<<blocks, 256>>>(d, 0.5f, 4);
    k<OP, CHAINS><<<blocks, 256>>>(d, 0.5f, iters);
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), d, blocks * sizeof(*d), hipMemcpyDeviceToHost);
    double s = 0;
    for (auto v : h) s += (double)v;
    hipFree(d);
    return s / blocks / iters / 64;
}

template <int OP>
static void row(const char *name)
{
    for (int w = 1; w <= 2; ++w)
        printf("%-14s %d wave(s)/SIMD  chains 1: %5.1f  2: %5.1f  4: %5.1f  8: %5.1f  16: %5.1f   cycles per instruction of one wave\n", name, w,
               run<OP, 1>(w), run<OP, 2>(w), run<OP, 4>(w), run<OP, 8>(w), run<OP, 16>(w));
}

int main()
{
    row<0>("v_fma_f32");
    row<1>("v_pk_fma_f32");
    row<2>("v_pk_add_f32");
    row<5>("v_pk_mul_f32");
    row<3>("v_exp_f32");
    row<4>("v_rcp_f32");
    return 0;
}
