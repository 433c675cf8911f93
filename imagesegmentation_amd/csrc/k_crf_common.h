// Pieces shared by the CRF update kernels of k_crf.hip and the fused-update instance files.
#pragma once
#include <cmath>
#include <cstdlib>
#include <mutex>

#include "fs_common.h"
// Timing experiments only (tools/build_variant.sh compiles the r = 7 instance file with one of them set and
// tools/run_variants.sh times the builds side by side; the product build defines none): FS_SKIPMAIN=1 drops the window
// walk of the fused update, FS_SKIPSIM=1 only the similarity's arithmetic inside it, FS_SKIPQ=1 the staging of Q,
// FS_SKIPFIN=1 the epilogue; FS_DEPTH_SERIES=0 evaluates the relative depth difference with a reciprocal per neighbour
// instead of the series (see fused_accumulate); FS_CRF_MAX_BLOCKS=256u launches one block per CU;
// FS_ASM_STEP=0 leaves the window walks to the compiler instead of the hand-scheduled assembly of walk_similarity (same
// bits, about 2 % slower at r = 7 and 5 % at r = 15; the Makefile builds libfs_hip_cwalk.so that way for the test that
// compares the two), FS_ASM_ABL=1..4 are ablations of the assembly (wrong results, for pricing its instructions);
// FS_TILE_CLOCKS=1 adds cycle counters to the phases of the r = 7 kernel's tiles (tools/phase_clocks.py).
#ifndef FS_CRF_MAX_BLOCKS
#define FS_CRF_MAX_BLOCKS 512u      // resident blocks of the r = 7 kernel: two per CU (256u: one per CU, one wave per SIMD)
#endif
#ifndef FS_DEPTH_SERIES
#define FS_DEPTH_SERIES 1
#endif
#ifndef FS_SKIPSIM
#define FS_SKIPSIM 0
#endif
#ifndef FS_SKIPQ
#define FS_SKIPQ 0
#endif
#ifndef FS_SKIPFIN
#define FS_SKIPFIN 0
#endif
#ifndef FS_SKIPMAIN
#define FS_SKIPMAIN 0
#endif
#ifndef FS_TILE_CLOCKS
#define FS_TILE_CLOCKS 0             // 1: per-phase cycle counts of the r = 7 kernel's tiles (fs_debug_phase_clocks)
#endif
#ifndef FS_ASM_STEP
#define FS_ASM_STEP 1               // window walks in assembly (walk_similarity); 0: the compiler's code for step()
#endif
#ifndef FS_ASM_ABL
#define FS_ASM_ABL 0                // with FS_ASM_STEP: 1 no exponentials, 2 no messages, 3 / 4 a v_mov / v_mul in place of each exponential
#endif

namespace fs {

// constants of the fused update: exponent factors of the relative depth difference for the three Gaussian kernels,
// the evidence of the "unassigned" label; uw: -log2 of the window's common (w1', w2', w3') when uniform
struct FusedConsts { float K1, K2, K3, evidence; float uw[4]; };
// one launch of the fused update over `nframes` frames (descs: device array)
struct FusedLaunch {
    const CrfFrameDesc *descs;
    int nframes;
    CrfGeom g;
    FusedConsts K;
    const float4 *wtab;      // per-offset weights [(2r+1)^2] = (w1', w2', w3', bias), zeros at the centre
    float weight;
    int first, last;
    unsigned *queue;
    hipStream_t st;
    bool trace = false;      // the diagnostic kernels, which also write CrfFrameDesc::potentials and trace_q
    int c0 = 0;              // wide passes (launch_roll_wide): first channel slot of the pass
    // the launch works on the tiles that list a_min .. a_max labels and leaves the others alone: a frame whose fullest
    // tile needs a kernel variant with many channels runs its other tiles through the variant that fits them
    int a_min = 0, a_max = 1 << 30;
    // rolling kernels: the launch's tiles in the order to work on them, those that take the most walks of the window first
    // (crf_tile_order_kernel: four counts, then four lists of `tiles` entries each); null: every tile, by number
    const int32_t *order = nullptr;
};
// Instance files (k_crf_fused_*.hip): each compiles the kernel variants it names and returns
// hipErrorInvalidValue for a combination it does not carry.
//   variant: 0 = Gaussian similarity, window constants uniform; 1 = Gaussian, per-offset weight table;
//            2 = logistic-regression similarity
hipError_t launch_fused_r7_uniform(int channel_class, const FusedLaunch &L);      // static tile + halo in LDS, r = 7
hipError_t launch_roll_uniform(int channel_class, const FusedLaunch &L);          // rolling window, variant 0, any radius
hipError_t launch_roll_table(int variant, int channel_class, const FusedLaunch &L);   // rolling window, variants 1 and 2
// one pass of a wide update (more channels per tile than kMaxTileChannels): the messages of channel slots
// [L.c0, L.c0 + kMaxTileChannels) of every tile to descs[].msg; variant as above, any radius (k_crf_roll_t.hip)
hipError_t launch_roll_wide(int variant, const FusedLaunch &L);
// tiles with many labels around them, all channels of a walk fed by one evaluation of the similarity: `quads` quad planes
// of Q in the ring; variant 0, r = 10 (quads 4, 8) and r = 15 (quads 3, 8) (k_crf_roll_m.hip)
hipError_t launch_roll_multi(int quads, const FusedLaunch &L);
// the same kernels built with their diagnostic outputs (variant 0 at any radius; k_crf_trace.hip)
hipError_t launch_fused_trace(int channel_class, const FusedLaunch &L);

constexpr int FTW = 32;                       // tile width of the fused kernel

namespace {

typedef float f2 __attribute__((ext_vector_type(2)));      // packed pair: arithmetic lowers to v_pk_*_f32
typedef __attribute__((address_space(1))) float gfloat;   // pointer known to be global: saddr + 32-bit offset loads
// A pointer that came out of a descriptor in memory is generic to the compiler, and loads through it are flat loads,
// which count as LDS operations as well: every wait for an LDS read then also waits for the loads in flight.  The
// arrays of a descriptor are device memory, so the kernels say so where they use them.
template <typename T>
__device__ __forceinline__ const __attribute__((address_space(1))) T *global_ptr(const T *p)
{
    return (const __attribute__((address_space(1))) T *)p;
}
template <typename T>
__device__ __forceinline__ __attribute__((address_space(1))) T *global_ptr(T *p)
{
    return (__attribute__((address_space(1))) T *)p;
}
typedef float f4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 global_load4(const float4 *p, unsigned i)
{
    const f4v v = ((const __attribute__((address_space(1))) f4v *)p)[i];
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ float2 global_load2(const float2 *p, unsigned i)
{
    const f2 v = ((const __attribute__((address_space(1))) f2 *)p)[i];
    return make_float2(v.x, v.y);
}
constexpr unsigned kNoPlane = 0xffffffffu;

#if FS_ASM_STEP
// One window position of a row that both pixels of a lane's pair see, in the instruction order it should issue in (Gaussian
// similarity, weights folded into the exponents, depth series; the arithmetic is described at fused_accumulate in
// k_crf_fused.h): the neighbour's record (fa01 = depth, r; fa23 = g, b; fb = angles) against the lane's pixel pair, 26 vector
// instructions for the similarity, then one per channel for the messages of the position before (walk_messages):
// acc[k] += sim * q[k] on (pixel 0, pixel 1) pairs.  The same operations on the same operands as the C++ form (step), so
// the same bits; written out because the compiler's code for it carries two to four register moves per position (a packed
// operand that needs the second register of a pair in both halves), keeps the six exponentials back to back and fills the
// waits between dependent instructions with s_nop.  Here a component is selected by op_sel, the exponentials sit between
// the packed operations of the other kernels, no instruction reads a register that the instruction in front of it wrote
// (gfx950 wants one wait state between a packed or transcendental write and its use), and a position's last additions
// and its message updates run inside the next position, where they wait for nothing.  The state that crosses positions:
// s12 = 2^-x1 + 2^-x2 and s3 = 2^-x3 of the position just evaluated (pinned to two register pairs so that their halves
// have names), and that position's channel vector, which the caller keeps.  sim_prev receives the finished similarity of
// the position before.  The fourteen fixed registers are v[242:255]: for kernels built for at most two blocks per CU
// (a kernel built for three has 168 registers per lane; the rolling kernel at r = 10 was tried with v[154:167] and spilled
// 21 registers for 1 % of time, so it keeps the compiler's code).
// wx, wy, wz: -log2 of the three weights for (pixel 0, pixel 1), +inf where the position is the pixel itself or the pixel
// has no neighbour there.
#define FS_SUBQ(d, a, b, sel) "v_pk_add_f32 " d ", " a ", " b " " sel " neg_lo:[0,1] neg_hi:[0,1]\n"
#define FS_LO "op_sel_hi:[0,1]"
#define FS_HI "op_sel:[1,0]"
#if FS_ASM_ABL == 1
#define FS_EXP(r) ""
#elif FS_ASM_ABL == 3
#define FS_EXP(r) "v_mov_b32_e32 " r ", " r "\n"
#elif FS_ASM_ABL == 4
#define FS_EXP(r) "v_mul_f32_e32 " r ", " r ", " r "\n"
#else
#define FS_EXP(r) "v_exp_f32_e64 " r ", -" r "\n"
#endif
// P0 .. P6: the seven register pairs; P1L/P1H, P3L/P3H, P6L/P6H: the halves of three of them
#define FS_WALK_SIM_TEXT(P0, P1, P2, P3, P4, P5, P6, P1L, P1H, P3L, P3H, P6L, P6H)                                      \
    "v_pk_add_f32 %[svp], " P6 ", " P3 "\n"                            /* the position before, finished   */           \
    FS_SUBQ(P0, "%[fa01]", "%[cd]", FS_LO)                             /* e  = depth difference           */           \
    FS_SUBQ(P1, "%[fa01]", "%[cr]", FS_HI)                             /* e0 = red                        */           \
    "v_pk_fma_f32 " P2 ", " P0 ", %[pC], %[pB]\n"                                                                       \
    FS_SUBQ(P3, "%[fa23]", "%[cg]", FS_LO)                             /* e1 = green                      */           \
    "v_pk_mul_f32 " P4 ", " P0 ", " P0 "\n"                                                                             \
    "v_pk_fma_f32 " P2 ", " P0 ", " P2 ", 1.0 op_sel_hi:[1,1,0]\n"                                                      \
    "v_pk_fma_f32 " P1 ", " P1 ", " P1 ", %[wy]\n"                                                                      \
    "v_pk_mul_f32 " P4 ", " P4 ", " P2 "\n"                            /* u / a^2                         */           \
    FS_SUBQ(P5, "%[fa23]", "%[cb]", FS_HI)                             /* e2 = blue                       */           \
    "v_pk_fma_f32 " P6 ", %[K1], " P4 ", %[wx]\n"                      /* x1                              */           \
    "v_pk_fma_f32 " P1 ", " P3 ", " P3 ", " P1 "\n"                                                                     \
    FS_EXP(P6L)                                                                                                         \
    FS_SUBQ(P3, "%[fb]", "%[ca0]", FS_LO)                              /* g0                              */           \
    FS_EXP(P6H)                                                                                                         \
    "v_pk_fma_f32 " P1 ", " P5 ", " P5 ", " P1 "\n"                                                                     \
    FS_SUBQ(P5, "%[fb]", "%[ca1]", FS_HI)                              /* g1                              */           \
    "v_pk_fma_f32 " P1 ", %[K2], " P4 ", " P1 "\n"                     /* x2                              */           \
    "v_pk_fma_f32 " P3 ", " P3 ", " P3 ", %[wz]\n"                                                                      \
    FS_EXP(P1L)                                                                                                         \
    "v_pk_fma_f32 " P3 ", " P5 ", " P5 ", " P3 "\n"                                                                     \
    FS_EXP(P1H)                                                                                                         \
    "v_pk_fma_f32 " P3 ", %[K3], " P4 ", " P3 "\n"                     /* x3                              */           \
    "v_pk_add_f32 " P6 ", " P6 ", " P1 "\n"                                                                             \
    FS_EXP(P3L)                                                                                                         \
    FS_EXP(P3H)
#define FS_WALK_SIM_IN                                                                                                   \
    [fa01] "v"(fa01), [fa23] "v"(fa23), [fb] "v"(fb), [cd] "v"(cd), [cr] "v"(cr), [cg] "v"(cg), [cb] "v"(cb), [ca0] "v"(ca0),  \
        [ca1] "v"(ca1), [pB] "v"(pB), [pC] "v"(pC), [K1] "v"(KA1), [K2] "v"(KA2), [K3] "v"(KA3), [wx] "s"(wx), [wy] "s"(wy),     \
        [wz] "s"(wz)
__device__ __forceinline__ void walk_similarity(f2 &sim_prev, f2 &s12, f2 &s3, f2 fa01, f2 fa23, f2 fb, f2 cd, f2 cr, f2 cg, f2 cb, f2 ca0,
                                                f2 ca1, f2 pB, f2 pC, f2 KA1, f2 KA2, f2 KA3, f2 wx, f2 wy, f2 wz)
{
    asm volatile(FS_WALK_SIM_TEXT("v[242:243]", "v[244:245]", "v[246:247]", "v[248:249]", "v[250:251]", "v[252:253]", "v[254:255]",
                                  "v244", "v245", "v248", "v249", "v254", "v255")
                 : [svp] "=&v"(sim_prev), "+{v[254:255]}"(s12), "+{v[248:249]}"(s3)
                 : FS_WALK_SIM_IN
                 : "v242", "v243", "v244", "v245", "v246", "v247", "v250", "v251", "v252", "v253");
}
#undef FS_WALK_SIM_TEXT
#undef FS_WALK_SIM_IN
#undef FS_SUBQ
#undef FS_LO
#undef FS_HI
#undef FS_EXP
// acc[k] += sim * q[k], k = 0 .. 3, for both pixels of the pair: one quad of channels (q01 = channels 0, 1; q23 = 2, 3), the
// channel taken from its register pair by op_sel
__device__ __forceinline__ void walk_messages_quad(f2 &a0, f2 &a1, f2 &a2, f2 &a3, f2 sim, f2 q01, f2 q23)
{
    if (FS_ASM_ABL == 2) return;
    asm volatile("v_pk_fma_f32 %[a0], %[sv], %[q01], %[a0] op_sel_hi:[1,0,1]\n"
                 "v_pk_fma_f32 %[a1], %[sv], %[q01], %[a1] op_sel:[0,1,0]\n"
                 "v_pk_fma_f32 %[a2], %[sv], %[q23], %[a2] op_sel_hi:[1,0,1]\n"
                 "v_pk_fma_f32 %[a3], %[sv], %[q23], %[a3] op_sel:[0,1,0]\n"
                 : [a0] "+v"(a0), [a1] "+v"(a1), [a2] "+v"(a2), [a3] "+v"(a3)
                 : [sv] "v"(sim), [q01] "v"(q01), [q23] "v"(q23));
}
#endif
   // s_src entry of a channel that reads zeros

// per-tile record of crf_tile_active_kernel (TILE_REC ints): [0] = count A of listed labels,
// [REC_MASK .. REC_MASK+MASK_WORDS) = bit mask, [REC_LAB .. REC_LAB+A) = the labels, ascending
constexpr int MASK_WORDS = kMaxLabels / 32;
constexpr int REC_MASK = 8, REC_LAB = REC_MASK + MASK_WORDS;
constexpr int TILE_REC = REC_LAB + kMaxLabels;

// Smallest label >= 1 whose bit is clear in a tile's label mask (MASK_WORDS words), found by 16 consecutive lanes of a
// wave, one mask word each (lane16 = the lane's position in its group); every lane of the group gets the result, a large
// number if every bit is set.
static_assert(MASK_WORDS == 16, "one lane per mask word");
__device__ __forceinline__ int first_unlisted_label(const int *mask, int lane16)
{
    unsigned w = ~(unsigned)mask[lane16];
    if (lane16 == 0) w &= ~1u;                                   // label 0 does not count
    int cand = w ? lane16 * 32 + __ffs((int)w) - 1 : 0x7fffffff;
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) cand = min(cand, __shfl_xor(cand, off, 16));
    return cand;
}

// Label channels of a frame for a kernel variant that carries at most `tile_channels` channels per tile; 0 when the
// descriptor was written ahead of the label count (CrfFrameDesc::spec) and the frame turned out not to fit.
__device__ __forceinline__ int frame_channels(const CrfFrameDesc &d, int tile_channels)
{
    if (!d.spec) return d.S;
    const int S = d.spec[0] + 2;                              // num_surfaces + 1, num_surfaces = largest patch label + 1
    const int ce = min(S, d.spec[1] + 3);                     // listed labels + background + representative + "unassigned"
    return S <= d.spec_cap && ce <= tile_channels ? S : 0;
}

// First block of a launch made ahead of the label count: hands the two device words to the host (CrfFrameDesc::publish).
__device__ __forceinline__ void publish_counts(const CrfFrameDesc *descs)
{
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    const CrfFrameDesc &d = descs[0];
    if (!d.spec || !d.publish) return;
    const unsigned seq = *d.publish_seq + 1u;
    *d.publish_seq = seq;
    const unsigned long long word = (unsigned long long)(unsigned)d.spec[0] | (unsigned long long)((unsigned)d.spec[1] & 0xffffu) << 32 |
                                    (unsigned long long)(seq & 0xffffu) << 48;
    __hip_atomic_store(d.publish, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Dynamic LDS a kernel instantiation has been allowed so far.  The attribute belongs to (device, function), so
// the record is kept per device; several contexts (one per GPU, or several host threads) may launch at once.
struct LdsAllowance {
    static constexpr int kDevices = 64;
    std::mutex m;
    size_t have[kDevices] = {};
    hipError_t ensure(const void *kern, size_t lds)
    {
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        std::lock_guard<std::mutex> g(m);
        const bool tracked = dev >= 0 && dev < kDevices;
        if (tracked && lds <= have[dev]) return hipSuccess;
        e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e == hipSuccess && tracked) have[dev] = lds;
        return e;
    }
};

// Diagnostic build only: sign * v[c] of every channel slot written to the plane of the slot's label, for the lane's two
// pixels where they have a depth, and the representative's value to the plane of every label the tile does not list
// (s_list[REC_MASK ..] has the bits of the listed labels, of 0 and of S - 1).
template <int NC>
__device__ __forceinline__ void trace_planes(const f2 (&v)[NC], float sign, float *planes, const int *s_list, int S, size_t No,
                                             const unsigned (&p)[2], const bool (&ok)[2], const bool (&valid)[2])
{
    const int CE = s_list[0], rep_pos = s_list[2];
    f2 rep = f2{0.0f, 0.0f};
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        if (c < CE) {
            const int lab = s_list[REC_LAB + c];
            const f2 t = v[c] * f2{sign, sign};
            if (c == rep_pos) rep = t;
            if (ok[0] && valid[0]) planes[(size_t)lab * No + p[0]] = t.x;
            if (ok[1] && valid[1]) planes[(size_t)lab * No + p[1]] = t.y;
        }
    }
    if (rep_pos < 0) return;
    for (int l = 1; l < S - 1; ++l) {
        if ((unsigned)s_list[REC_MASK + (l >> 5)] >> (l & 31) & 1u) continue;
        if (ok[0] && valid[0]) planes[(size_t)l * No + p[0]] = rep.x;
        if (ok[1] && valid[1]) planes[(size_t)l * No + p[1]] = rep.y;
    }
}

// Potts compatibility, softmax and overrides of the fused update for the two vertically adjacent pixels a lane owns
// (compact planes, synthesised unary).  m[c] = messages of channel slot c as a packed pair
// (pixel 0, pixel 1), so the arithmetic is v_pk_*_f32 like the window walk.  The work follows the tile's channel
// count: slots are visited in groups of four behind one uniform test per group, and what depends on a slot's role
// comes from s_tab[c] = (multiplicity, evidence, unary, plane offset): multiplicity 1 for a listed label, n_rest for the
// representative, 0 for an unused slot of the last group; evidence of the constant "unassigned" input at the last
// channel (find_surfaces.py:356-357, 451-452); unary potential of the slot's label at a pixel that does not belong to
// it: 10 for label 0, 1 otherwise (:459-472), +inf for an unused slot (probability 0); label * pixels as the bits of
// an unsigned (below 2^32: the host checks), where the slot's plane starts in Qout.
// TRACE (diagnostic builds): d.potentials and d.trace_q, where not null, receive the softmax argument and the new
// probabilities of all S labels.
template <int NC, bool TRACE = false>
__device__ __forceinline__ void finish_pair_compact(f2 (&m)[NC], const int *s_list, const float4 *s_tab, const CrfFrameDesc &d, int S,
                                                    size_t No, const unsigned (&p)[2], const int (&own)[2], const bool (&ok)[2],
                                                    const bool (&valid)[2], float weight, int last)
{
    static_assert(NC % 4 == 0, "channel slots come in groups of four");
    const int CE = s_list[0];
    const float pv = (float)(1.0 / (double)(S - 1));
    // a pixel's own patch label as a key no slot matches when the pixel lies in no patch
    const int key0 = own[0] != 0 ? own[0] : -1, key1 = own[1] != 0 ? own[1] : -1;
    // total over all S labels (the representative counts n_rest times), compensated so that
    // compatibility[l] = total - msg[l] keeps full relative accuracy
    f2 tot = f2{0.0f, 0.0f}, tot_lo = f2{0.0f, 0.0f};
#pragma unroll
    for (int g = 0; g < NC / 4; ++g) {
        if (g * 4 < CE) {
#pragma unroll
            for (int c = g * 4; c < g * 4 + 4; ++c) {
                const float4 role = s_tab[c];
                m[c] = m[c] + f2{role.y, role.y};
                const f2 term = m[c] * f2{role.x, role.x};
                const f2 t = tot + term;
                const f2 bp = t - tot;
                tot_lo += (tot - (t - bp)) + (term - bp);
                tot = t;
            }
        }
    }
    float mx0 = -INFINITY, mx1 = -INFINITY;
#pragma unroll
    for (int g = 0; g < NC / 4; ++g) {
        if (g * 4 < CE) {
#pragma unroll
            for (int c = g * 4; c < g * 4 + 4; ++c) {
                const int lab = s_list[REC_LAB + c];
                const float base = s_tab[c].z;
                const f2 comp = (tot - m[c]) + tot_lo;             // Potts: sum of the other labels' messages (:297-298, 360)
                const f2 un = f2{lab == key0 ? 0.0f : base, lab == key1 ? 0.0f : base};   // :459-472
                m[c] = -(un + f2{weight, weight} * comp);          // :361
                mx0 = fmaxf(mx0, m[c].x);
                mx1 = fmaxf(mx1, m[c].y);
            }
        }
    }
    // Diagnostic build only: the softmax argument unary + weight * compatibility of every one of the S labels, as label
    // planes like Q; a label the tile does not list carries the representative's value.
    if (TRACE && d.potentials) trace_planes<NC>(m, -1.0f, d.potentials, s_list, S, No, p, ok, valid);
    f2 den = f2{0.0f, 0.0f};
    const f2 mx = f2{mx0, mx1};
#pragma unroll
    for (int g = 0; g < NC / 4; ++g) {
        if (g * 4 < CE) {
#pragma unroll
            for (int c = g * 4; c < g * 4 + 4; ++c) {
                const f2 a = (m[c] - mx) * f2{1.44269504088896341f, 1.44269504088896341f};
                m[c] = f2{__builtin_amdgcn_exp2f(a.x), __builtin_amdgcn_exp2f(a.y)};      // softmax(-add), :363
                const float mult = s_tab[c].x;
                den = __builtin_elementwise_fma(m[c], f2{mult, mult}, den);
            }
        }
    }
    if (last) {
        // Only the label is wanted: the first maximum of the float32 probabilities (:531-532).  Labels whose arguments
        // lie so close to the largest one that their quotients round to the same float are tied there, and the lowest
        // of them wins; taking the maximum of the arguments instead would let the last bit of an argument decide.
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (!ok[i]) continue;
            int best = 0;
            float bestv = -1.0f;
            const float deni = i == 0 ? den.x : den.y;
            if (valid[i]) {                                              // Q[depth == 0] = 0: argmax of zeros is label 0
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    if (c < CE) {
                        const int lab = s_list[REC_LAB + c];
                        const float v = (i == 0 ? m[c].x : m[c].y) / deni;
                        if (TRACE) { if (i == 0) m[c].x = v; else m[c].y = v; }
                        if (v > bestv || (v == bestv && lab < best)) { bestv = v; best = lab; }
                    }
                }
            }
            global_ptr(d.surfaces)[p[i]] = best == S - 1 ? 0 : best;
        }
        if (TRACE && d.trace_q) trace_planes<NC>(m, 1.0f, d.trace_q, s_list, S, No, p, ok, valid);
        return;
    }
    const f2 inv_den = f2{1.0f / den.x, 1.0f / den.y};
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        if (c < CE) {
            const int lab = s_list[REC_LAB + c];
            const f2 q = m[c] * inv_den;
            if (TRACE) m[c] = q;
            const float ov = lab == 0 ? 0.0f : pv;                       // Q[depth == 0] = prob, :520-525
            const unsigned plane = __float_as_uint(s_tab[c].w);
            if (ok[0]) global_ptr(d.Qout)[plane + p[0]] = valid[0] ? q.x : ov;
            if (ok[1]) global_ptr(d.Qout)[plane + p[1]] = valid[1] ? q.y : ov;
        }
    }
    if (TRACE && d.trace_q) trace_planes<NC>(m, 1.0f, d.trace_q, s_list, S, No, p, ok, valid);
}

}  // namespace

}  // namespace fs
