// b9_diag.hip.h -- every DIAGNOSTIC hook of the kernels, in one place: time stamps, per-workgroup start / end records,
// assembly markers and event counters.  None of the macros below is defined in the shipped library (each hook then
// compiles to nothing); tools/build_variant.py builds the instrumented variants (-DB9_STAMPS, -DB9_GANTT, -DB9_ASM_MARKERS,
// -DB9_MARG_STATS, -DB9_MARG_LIFE) into build/variants/, and the tools that read them say which one they need.
// No hook changes a result.  Part of the single translation unit b9_kernels.hip; gfx950 only.
#pragma once

// Diagnostic build only (-DB9_STAMPS): per-wave s_memtime stamps of the hot kernel's phases,
// written to a buffer of their own that no kernel reads.  Never defined in the shipped library.
#ifdef B9_STAMPS
#define B9_NSTAMP 12
__device__ unsigned long long g_stamps[8192 * B9_NSTAMP];
#define B9_STAMP_MASK 0xFFF       // which stamps are live (bit k); the rest compile to nothing
#define STAMP(k)                                                                                   \
    if ((B9_STAMP_MASK >> (k)) & 1)                                                                \
    do {                                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        unsigned long long t_;                                                                     \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        const unsigned wv_ = blockIdx.x * 4 + (threadIdx.x >> 6);                                  \
        if ((threadIdx.x & 63) == 0 && wv_ < 8192) g_stamps[wv_ * B9_NSTAMP + (k)] = t_;           \
    } while (0)
extern "C" int b9_debug_read_stamps(unsigned long long *out, int n_waves)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * B9_NSTAMP * n_waves);
}
extern "C" int b9_debug_clear_stamps(void)
{
    static unsigned long long zeros[8192 * B9_NSTAMP];
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), zeros, sizeof zeros);
}
#else
#define STAMP(k) do {} while (0)
#endif

// -DB9_ASM_MARKERS (diagnostic): comments in the generated assembly that delimit a role's code (tools/role_isa.py)
#ifdef B9_ASM_MARKERS
#define B9_MARK(name) asm volatile("; b9-mark " name)
#else
#define B9_MARK(name) do {} while (0)
#endif

#ifdef B9_GANTT      // diagnostic build only: per-workgroup start / end times (s_memrealtime, 100 MHz) of 8 consecutive launches
#define B9_GANTT_WG 8192
__device__ unsigned long long g_gantt[8 * B9_GANTT_WG * 4];
__device__ unsigned long long g_gantt_heavy[64 * 8];          // phase stamps of the heavy role (one slot per workgroup id < 64)
extern "C" int b9_debug_read_gantt(unsigned long long *out)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_gantt), sizeof(unsigned long long) * 8 * B9_GANTT_WG * 4);
}
extern "C" int b9_debug_read_gantt_heavy(unsigned long long *out)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_gantt_heavy), sizeof(unsigned long long) * 64 * 8);
}
__device__ unsigned long long g_gantt_heavy2[64 * 16];        // stamps inside one star's evaluation (lane HS2_LANE of wave 0 of workgroups < 64)
extern "C" int b9_debug_read_gantt_heavy2(unsigned long long *out)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_gantt_heavy2), sizeof(unsigned long long) * 64 * 16);
}
__device__ unsigned long long g_gantt_walk[8];                 // the tree walk of workgroup 0: kernel entry | loads issued | loads landed | walk done
extern "C" int b9_debug_read_gantt_walk(unsigned long long *out)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_gantt_walk), sizeof(unsigned long long) * 8);
}
#define WSTAMP(k) do { if (blockIdx.x == 0 && threadIdx.x == 0 && g_gantt_walk[7] == 1ull) g_gantt_walk[k] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define WSTAMP_ON(v) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_gantt_walk[7] = (v); } while (0)
#define HS2_LANE 0
#define HS2(k) do { if (threadIdx.x == HS2_LANE && blockIdx.x < 64) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); g_gantt_heavy2[blockIdx.x * 16 + (k)] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#define HSTAMP(k) do { if (threadIdx.x == 0 && blockIdx.x < 64) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); g_gantt_heavy[blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); } } while (0)
__device__ unsigned long long g_gantt_star[64 * 4 * 8];       // phase stamps of the marginalised star role: [role-relative workgroup id < 64][wave][phase]
extern "C" int b9_debug_read_gantt_star(unsigned long long *out)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_gantt_star), sizeof(unsigned long long) * 64 * 4 * 8);
}
#ifndef B9_SSTAMP_BASE          // (-DB9_SSTAMP_BASE=n: the window of 64 dispatch positions that is recorded)
#define B9_SSTAMP_BASE 0
#endif
#define SSTAMP(id, k) do { if ((threadIdx.x & 63) == 0 && (unsigned)((id) - B9_SSTAMP_BASE) < 64u) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); g_gantt_star[(((id) - B9_SSTAMP_BASE) * 4 + (threadIdx.x >> 6)) * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#define WSTAMP_LANDED(a, b) do { WSTAMP(a); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); WSTAMP(b); } while (0)
// start / end of every workgroup of the fused-step and tree kernels: [t_in, t_out, role | XCC id << 8 | HW_ID << 16, launch index]
#define B9_GANTT_ENTER() const unsigned long long t_in_ = __builtin_amdgcn_s_memrealtime()
#define B9_GANTT_EXIT(launch, role)                                                                                            \
    do {                                                                                                                       \
        __syncthreads();          /* the workgroup's last wave */                                                              \
        if (threadIdx.x == 0 && blockIdx.x < B9_GANTT_WG) {                                                                    \
            const unsigned long long l_ = (launch);                                                                            \
            unsigned long long *g_ = g_gantt + ((l_ & 7ull) * B9_GANTT_WG + blockIdx.x) * 4;                                   \
            unsigned xcc_, hw_;       /* HW_ID: CU (bits 8-11), SH (12), SE (13-15) of the workgroup's first wave */           \
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_));                                                \
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_));                                                  \
            g_[0] = t_in_; g_[1] = __builtin_amdgcn_s_memrealtime();                                                           \
            g_[2] = (unsigned long long)(role) | ((unsigned long long)(xcc_ & 15u) << 8) | ((unsigned long long)(hw_ & 0xFFFFu) << 16); \
            g_[3] = l_;                                                                                                        \
        }                                                                                                                      \
    } while (0)
#else
#define B9_GANTT_ENTER() do {} while (0)
#define B9_GANTT_EXIT(launch, role) (void)(role)
#define WSTAMP_LANDED(a, b) do {} while (0)
#define HS2(k) do {} while (0)
#define HSTAMP(k) do {} while (0)
#define SSTAMP(id, k) do {} while (0)
#define WSTAMP(k) do {} while (0)
#define WSTAMP_ON(v) do {} while (0)
#endif

#ifdef B9_MARG_STATS      // diagnostic build only (tools/marg_stats.py): what the marginalised kernel executes per star
__device__ unsigned long long g_marg_stats[8];
#define MSTAT(k, v) do { if ((threadIdx.x & 63) == 0) atomicAdd(&g_marg_stats[k], (unsigned long long)(v)); } while (0)
extern "C" int b9_debug_marg_stats(unsigned long long *out, int clear)
{
    int rc = (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_marg_stats), sizeof(unsigned long long) * 8);
    if (clear) { unsigned long long z[8] = {0}; rc |= (int)hipMemcpyToSymbol(HIP_SYMBOL(g_marg_stats), z, sizeof z); }
    return rc;
}
#define MSTAT_LIVE(k, pred) do { const unsigned long long lm_ = __ballot(pred); MSTAT(k, __popcll(lm_)); } while (0)
#else
#define MSTAT(k, v) do {} while (0)
#define MSTAT_LIVE(k, pred) do {} while (0)
#endif
#ifdef B9_MARG_LIFE       // diagnostic build only (tools/marg_life.py): start / end of every workgroup, units evaluated by its wave 0
__device__ unsigned long long g_marg_life[16384 * 4];
#define MLIFE(k, v) do { if (threadIdx.x == 0 && blockIdx.x < 16384) g_marg_life[blockIdx.x * 4 + (k)] = (v); } while (0)
#define MLIFE_UNIT() do { if (threadIdx.x == 0 && blockIdx.x < 16384) g_marg_life[blockIdx.x * 4 + 2] += 1; } while (0)
extern "C" int b9_debug_marg_life(unsigned long long *out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_marg_life), sizeof(unsigned long long) * 16384 * 4); }
#else
#define MLIFE(k, v) do {} while (0)
#define MLIFE_UNIT() do {} while (0)
#endif

// -DB9_MSTEP_NO_FRONT (no writer, no builders) / -DB9_MSTEP_NO_BUILD / -DB9_MSTEP_NO_STARS / -DB9_MSTEP_FIXED_CAND (the stars always against the prologue's tables) / -DB9_MSTEP_NO_DECIDE / -DB9_MSTEP_WAVES=n (timing probes of k_marg_step; WRONG chains): the table builders
// return at once / the star roles take candidate 0 without a decision / the kernel is built for n waves per SIMD
#ifdef B9_MSTEP_NO_BUILD
#define MSTEP_NO_BUILD 1
#else
#define MSTEP_NO_BUILD 0
#endif
#ifdef B9_MSTEP_NO_FRONT
#define MSTEP_NO_FRONT 1
#else
#define MSTEP_NO_FRONT 0
#endif
#ifdef B9_MSTEP_NO_STARS
#define MSTEP_NO_STARS 1
#else
#define MSTEP_NO_STARS 0
#endif
#ifdef B9_MSTEP_FIXED_CAND
#define MSTEP_FIXED_CAND 1
#else
#define MSTEP_FIXED_CAND 0
#endif
#ifdef B9_MSTEP_NO_DECIDE
#define MSTEP_NO_DECIDE 1
#else
#define MSTEP_NO_DECIDE 0
#endif
#ifndef B9_MSTEP_WAVES
#define B9_MSTEP_WAVES(NFP, NPOPS) B9_MARG_WAVES(NFP, NPOPS, false)
#endif
