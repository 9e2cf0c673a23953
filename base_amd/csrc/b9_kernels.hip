// b9_kernels.hip -- hand-written gfx950 kernels of the BASE-9 per-step log-posterior path.
//
//   k_derive_iso   SURVEY 8a rows a3, a8  one workgroup per (walker, population, slice of EEPs): grid
//                                     brackets by wave ballot, EEP-range intersection, EEP-wise
//                                     tri-linear interpolation (one value per thread).  In the
//                                     device-resident sampler it first FINISHES the previous step
//                                     (fixed-order sum of partials + prior + Metropolis accept,
//                                     recomputed identically by every workgroup of the walker) and
//                                     draws the new proposal (Philox4x32-10 + Box-Muller)
//   k_star_like    rows a4-a7, a9      given-mass mode, the dominant kernel.  HOT workgroups: one
//                                     LANE per star -- 8-ary search in the LDS-staged mass column,
//                                     magnitude rows from L2, binary flux combination, Gaussian
//                                     chi^2, population mixture, product-form field-star mixture,
//                                     one partial per wave.  HEAVY workgroups (first in the grid):
//                                     the stars above the AGB tip (IFMR -> WD cooling -> WD
//                                     atmosphere, or NS/BH) through the general per-star code
//   k_star_marg    row a6 (marg.)      marginalised mode: ONE WAVEFRONT PER STAR integrating over
//                                     primary mass and mass ratio, rigorous pruning, online
//                                     log-sum-exp, wavefront-shuffle merge
//   k_mcmc_step    rows a4-a9, 8f-1    the fused sampler step, ONE launch per MCMC step (given-mass mode):
//                                     accept/reject of step t-1 (redundantly per wave), k_star_like's
//                                     roles on step t's proposal, and both candidate isochrone sets
//                                     of step t+1 on extra workgroups
//   k_finalize     row a8              fixed-order sum of the partials + cluster prior (+ the accept
//                                     of a two-launch sampler block's last step)
//
// The kernels live in the *.hip.h files included below (one translation unit); this file holds
// k_finalize and the host-callable launch wrappers.
//
// The reference source is not mounted (/root/reference/README.md:4), so none of this can
// cite a reference file:line; DESIGN.md "Math" is the normative restatement and
// oracle/b9_oracle.c the CPU checker.  Floating-point contract: built with
// -ffp-contract=off; every interpolation is an explicit fma(t, b - a, a), which makes the
// derived isochrone bit-identical to the oracle's.
//
// No MFMA anywhere: there is no dense contraction on this path (BASELINE.json north_star).
// Diagnostic-only macros (never defined in the shipped library): B9_STAMPS (per-phase s_memtime
// stamps), B9_ABL_* (ablation builds used for the attribution in docs/LABNOTES.md section 8).
#include "b9_device.h"
#include "b9_launch.h"
#include <algorithm>
#include <cstring>
#include "../../include/base9_hip.h"

#include "b9_diag.hip.h"
#include "b9_common.hip.h"
#include "b9_derive.hip.h"
#include "b9_star.hip.h"
#include "b9_star_like.hip.h"
#include "b9_mcmc_step.hip.h"
#define B9_TREE_KD B9_TREE_KD_SMALL
namespace tree_kd3 {
#include "b9_mcmc_tree.hip.h"
}
#undef B9_TREE_KD
#undef B9_TREE_KV
#define B9_TREE_KD B9_TREE_KD_LARGE
namespace tree_kd5 {
#include "b9_mcmc_tree.hip.h"
}
#undef B9_TREE_KD
#include "b9_star_marg.hip.h"
#include "b9_marg_step.hip.h"

// ------------------------------------------------------------------------------------------
// k_finalize: one workgroup per walker: fixed-order sum of the partials + prior -> logpost[w]
// (SURVEY 8a row a8); -inf for a walker outside the grid; with mc.enabled also the accept/reject
// of the block's last step.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_finalize(const IsoHdr *__restrict__ hdr, const double *__restrict__ partial,
                                                   int n_partial, long long partial_stride, int n_pops,
                                                   const double *__restrict__ params, DevPriors pr,
                                                   double *__restrict__ logpost, double *__restrict__ perstar,
                                                   int n_stars, McmcDev mc, unsigned long long *__restrict__ done_flag, unsigned long long done_seq)
{
    __shared__ double s_red[4], s_cur[B9_NPARAM], s_lp;
    const int w = blockIdx.x, tid = threadIdx.x;
    const double *row = params + (size_t)w * B9_NPARAM;
    bool in_support;
    const double lp = finish_logpost(hdr, partial + (size_t)w * partial_stride, n_partial, row, pr, n_pops, w, s_red, &in_support);
    if (tid == 0) {
        logpost[w] = lp;
        // b9_logpost's completion word (mapped host memory, behind the value it announces): the host polls it instead of
        // waiting for the stream's completion signal
        if (done_flag) { __threadfence_system(); __hip_atomic_store(done_flag + w, done_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
    }
    // a walker inside the grid whose prior is -inf still has per-star values from the star kernel;
    // the oracle reports -inf for them as well
    if (perstar && !in_support)
        for (int i = tid; i < n_stars; i += 256) perstar[(size_t)w * n_stars + i] = NEG_INF;
    if (mc.enabled) metropolis_accept(mc, w, mc.step, mc.row, row, lp, true, s_cur, &s_lp);
}

// ------------------------------------------------------------------------------------------
// launch wrappers
// ------------------------------------------------------------------------------------------
hipError_t b9k_derive_iso(const DevPack &pk, double *d_params, int n_walkers, int n_pops,
                          IsoHdr *hdr, double *iso_data, long long iso_stride, int mass_cap,
                          const McmcDev &mc, const DevPriors &pr, const B9Prev &prev, hipStream_t stream)
{
    const int gy = (mass_cap * (pk.nfp + 1) + 255) / 256;
    hipLaunchKernelGGL(k_derive_iso, dim3(n_walkers * n_pops, gy), dim3(256), 0, stream,
                       pk, d_params, n_pops, hdr, iso_data, iso_stride, mass_cap, mc, pr,
                       prev.partial, prev.n_partial, prev.partial_stride, prev.hdr, prev.params);
    return hipGetLastError();
}

hipError_t b9k_derive_iso_rows(const DevPack &pk, const double *host_rows, double *d_params, int n_walkers, int n_pops,
                               IsoHdr *hdr, double *iso_data, long long iso_stride, int mass_cap, hipStream_t stream)
{
    if (n_walkers < 1 || n_walkers > 8) return hipErrorInvalidValue;
    HostRows rows{};
    std::memcpy(rows.v, host_rows, sizeof(double) * B9_NPARAM * n_walkers);
    const int gy = (mass_cap * (pk.nfp + 1) + 255) / 256;
    hipLaunchKernelGGL(k_derive_iso_rows, dim3(n_walkers * n_pops, gy), dim3(256), 0, stream,
                       pk, rows, d_params, n_pops, hdr, iso_data, iso_stride, mass_cap);
    return hipGetLastError();
}

// LDS of the heavy-star role: the WD axes (the cooling tracks' concatenated age axes only while they fit
// B9_WC_AGE_LDS_MAX doubles; longer ones are searched in L2) + per (candidate, population) the four AGB-tip columns
// and the mass column + each candidate's parameter row
static size_t heavy_lds_doubles(const DevPack &pk, int n_pops, int n_cand, int mass_cap)
{
    // [8 words of reduction scratch][the packed axes, DevPack::heavy_const][AGB tips: the whole table, or the corner columns]
    // [mass columns][parameter rows][find_bracket's over-read]
    const size_t n_tips = (size_t)pk.n_feh * pk.n_y * pk.n_age;
    const size_t tips = n_tips <= B9_TIPS_LDS_MAX ? n_tips : (size_t)4 * n_pops * n_cand * pk.n_age;
    return 8 + (size_t)pk.hc_len + tips + (size_t)n_pops * n_cand * mass_cap + (size_t)n_cand * B9_NPARAM + 8;
}

template <int NFP, int NPOPS>
static hipError_t launch_star_like(const DevPack &pk, const DevStars &st, const IsoHdr *hdr,
                                   const double *iso_data, long long iso_stride, int mass_cap,
                                   const double *d_params, int n_walkers, double *partial, long long partial_stride,
                                   double *perstar, const B9Groups &gr, int heavy_parts, hipStream_t stream)
{
    // + 8: find_bracket's last stage may read up to 6 entries past a column's end (masked out)
    const size_t lds = sizeof(double) * std::max((size_t)NPOPS * mass_cap + 8, heavy_lds_doubles(pk, NPOPS, 1, mass_cap));
    auto kern = k_star_like<NFP, NPOPS>;
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    const int hot = 8 * ((gr.n_blocks * NPOPS + 7) / 8) * n_walkers;     // padded so every XCD sees whole walker sets (two populations: a workgroup per half tile)
    const int heavy = (n_walkers * heavy_parts + 7) / 8 * 8;     // heavy-star workgroups lead the grid
    hipLaunchKernelGGL(kern, dim3(heavy + hot), dim3(256), lds, stream, pk, st, hdr, iso_data,
                       iso_stride, mass_cap, d_params, n_walkers, partial, partial_stride, gr.n_groups, gr.group_tiles,
                       gr.groups_per_block, gr.n_blocks, perstar, heavy, heavy_parts);
    return hipGetLastError();
}

#define B9_SWITCH_NFP(CALL2, CALL1)                 \
    switch (pk.nfp) {                               \
    case 4:  if (n_pops == 2) { return CALL2(4); } else { return CALL1(4); }   \
    case 8:  if (n_pops == 2) { return CALL2(8); } else { return CALL1(8); }   \
    case 16: if (n_pops == 2) { return CALL2(16); } else { return CALL1(16); } \
    default: return hipErrorInvalidValue;           \
    }

hipError_t b9k_star_like(const DevPack &pk, const DevStars &st, const IsoHdr *hdr,
                         const double *iso_data, long long iso_stride, int mass_cap,
                         const double *d_params, int n_walkers, int n_pops,
                         double *partial, long long partial_stride, double *perstar, const B9Groups &gr,
                         int heavy_parts, hipStream_t stream)
{
#define SL_ARGS pk, st, hdr, iso_data, iso_stride, mass_cap, d_params, n_walkers, partial, partial_stride, perstar, gr, heavy_parts, stream
#define SL2(NFP) launch_star_like<NFP, 2>(SL_ARGS)
#define SL1(NFP) launch_star_like<NFP, 1>(SL_ARGS)
    B9_SWITCH_NFP(SL2, SL1)
#undef SL1
#undef SL2
#undef SL_ARGS
}

hipError_t b9k_finalize(const IsoHdr *hdr, const double *partial, int n_partial, long long partial_stride,
                        int n_pops, const double *d_params, const DevPriors &pr, int n_walkers, double *d_logpost,
                        double *perstar, int n_stars, const McmcDev &mc, hipStream_t stream, unsigned long long *done_flag, unsigned long long done_seq)
{
    hipLaunchKernelGGL(k_finalize, dim3(n_walkers), dim3(256), 0, stream, hdr, partial, n_partial, partial_stride,
                       n_pops, d_params, pr, d_logpost, perstar, n_stars, mc, done_flag, done_seq);
    return hipGetLastError();
}

// doubles of one (walker, population)'s node table (MargLayout, b9_device.h)
long long b9k_marg_table_doubles(int nfp, int mass_cap, int K, int Q) { return marg_layout(nfp, mass_cap, K, Q).total; }
// Does a catalogue of n_star_chunks x n_pops split its star chunks' windows over several workgroups (DevStars::mg_piece)?  Below 512
// chunk-populations (32k stars of one population) -- a function of the CATALOGUE, never of the walkers on the GPU: it decides how a
// star's sum rounds.  How many workgroups each chunk gets is the catalogue plan's business (b9_capi_margplan.cpp).  Measured in round 4
// with a uniform split, ms per b9_logpost call at 4 x 4, one workgroup per chunk -> split: 10k stars x 1 walker 0.18 -> 0.07, 200 stars
// 0.14 -> 0.06, 20k x 8 walkers 0.21 -> 0.19; 30k x 2 populations x 8 walkers would lose (0.27 -> 0.37: 938 chunk-populations, unsplit).
int b9k_marg_split(int n_star_chunks, int n_pops) { return n_star_chunks * n_pops < 512 ? 1 : 0; }
long long b9k_marg_shares_doubles(int n_pieces, int n_pops) { return (long long)std::max(1, n_pieces) * n_pops * 128; }      // per walker
long long b9k_marg_wd_table_doubles(int nfp, int K) { return (long long)8 * K * (2 * nfp + 1); }       // per (walker, population)

// Does a launch of `wgs` star workgroups leave the chip nearly empty (at most five waves per SIMD on average)?  Then its waves
// are latency-bound on the row loop: the SPARSE tile setting (star_marg_body, TILE = 2: same bits, speed only).
static bool marg_sparse(long long wgs)
{
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu < 1) n_cu = 256;
    }
    return wgs * 4 <= (long long)5 * n_cu * 4;
}

template <int NFP, int NPOPS, bool SAMPLE>
static hipError_t launch_star_marg_t(const DevPack &pk, const DevStars &st, const IsoHdr *hdr, const double *iso_data,
                                     long long iso_stride, int mass_cap, const double *d_params, int n_walkers,
                                     double *partial, long long partial_stride, double *perstar, int K, int Q, const B9MargSample *smp, bool prune,
                                     double *tab, double *wd_tab, double *shares, hipStream_t stream)
{
    MargSample ms{};
    if (SAMPLE) { ms.mass = smp->mass; ms.ratio = smp->ratio; ms.member = smp->member; ms.pop = smp->pop; ms.k0 = smp->k0; ms.k1 = smp->k1; ms.row0 = smp->row0; }
    if (!tab) return hipErrorInvalidValue;
    const MargLayout L = marg_layout(NFP, mass_cap, K, Q);
    // the call's node table: one workgroup per (walker-population, 64-node chunk)
    const size_t lds = sizeof(double) * ((size_t)mass_cap + 8 + 8 * NFP);
    if (lds > 64 * 1024) {
        if (lds > 160 * 1024) return hipErrorInvalidValue;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_marg_table<NFP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((k_marg_table<NFP>), dim3(n_walkers * NPOPS, L.n_chunks), dim3(256), lds, stream, pk, hdr, iso_data,
                       iso_stride, mass_cap, NPOPS, d_params, K, Q, tab, L);
    // the stars: one workgroup (four waves sharing the node table's sub-chunks) per (64-star chunk, walker), dispatched in DevStars::marg_order
    // XCD placement (k_star_marg's head): two walker groups x four star-chunk groups.  Each XCD's L2 fetches its group's node
    // tables and its share of the stars once: 8 x (tables / wsplit) + wsplit x (star copy) bytes in all -- 50k stars x 8
    // walkers, 4 x 4 grid: 42.5 MB per launch at wsplit 1, 35.2 at 2, 41.3 at 4 (rocprofv3 FETCH_SIZE; the launch time is the
    // same for all three: 164 us -- the kernel is VALU-bound, the placement only decides how much crosses the fabric)
    const int wsplit = n_walkers % 2 == 0 ? 2 : 1;
    const int csplit = 8 / wsplit, n_chunks = st.mg_pad / 64;
    // SPLIT: a small catalogue's launch lasts as long as its heaviest star chunk (giants: 100 us where the median chunk takes
    // 40) while most of the chip idles, so several workgroups share a chunk's window (DevStars::mg_piece; k_marg_merge).  A
    // function of the CATALOGUE only -- it decides how a star's sum rounds, and a walker's chain must not depend on how many
    // walkers share the GPU.  The sampleMass draws keep one workgroup per chunk.
    const bool split = !SAMPLE && st.mg_n_pieces > 0;
    if (split && !shares) return hipErrorInvalidValue;
    const int per_xcd = (((split ? st.mg_n_pieces : n_chunks) + csplit - 1) / csplit) * (n_walkers / wsplit);
    const double cut2 = prune ? 2.0 * B9_MARG_CUT : __builtin_inf();
    // rows through LDS tiles or through scalar registers (star_marg_body, TILE): measured per instance -- 8 (4) filters x one
    // population, unsplit, is the one shape the scalar path still wins in this kernel (2.20 against 2.17e9 star-evals/s)
    const bool sparse = !SAMPLE && split && marg_sparse((long long)st.mg_n_pieces * n_walkers);
    const bool tiled = !SAMPLE && (split || NPOPS == 2 || NFP >= 16);
    const size_t tile_lds = sizeof(double) * 4 * B9_TILE_DOUBLES(NFP);
    if (sparse)
        hipLaunchKernelGGL((k_star_marg<NFP, NPOPS, SAMPLE, !SAMPLE, false, SAMPLE ? 0 : 2>), dim3(8 * per_xcd), dim3(256), tile_lds, stream, pk, st, hdr, iso_data, iso_stride,
                           mass_cap, d_params, partial, partial_stride, perstar, K, Q, ms, tab, L, n_walkers, cut2, wsplit, shares);
    else if (split)
        hipLaunchKernelGGL((k_star_marg<NFP, NPOPS, SAMPLE, !SAMPLE, false, SAMPLE ? 0 : 1>), dim3(8 * per_xcd), dim3(256), tile_lds, stream, pk, st, hdr, iso_data, iso_stride,
                           mass_cap, d_params, partial, partial_stride, perstar, K, Q, ms, tab, L, n_walkers, cut2, wsplit, shares);
    else if (tiled)
        hipLaunchKernelGGL((k_star_marg<NFP, NPOPS, SAMPLE, false, false, SAMPLE ? 0 : 1>), dim3(8 * per_xcd), dim3(256), tile_lds, stream, pk, st, hdr, iso_data, iso_stride,
                           mass_cap, d_params, partial, partial_stride, perstar, K, Q, ms, tab, L, n_walkers, cut2, wsplit, shares);
    else
        hipLaunchKernelGGL((k_star_marg<NFP, NPOPS, SAMPLE, false>), dim3(8 * per_xcd), dim3(256), 0, stream, pk, st, hdr, iso_data, iso_stride,
                           mass_cap, d_params, partial, partial_stride, perstar, K, Q, ms, tab, L, n_walkers, cut2, wsplit, shares);
    if (split)
        hipLaunchKernelGGL((k_marg_merge<NPOPS>), dim3(n_chunks, n_walkers), dim3(64), 0, stream, st, hdr, d_params, partial, partial_stride,
                           perstar, shares);
    if (st.n_wd > 0) {        // the catalogue's WD-stage stars: their node table (2 x 8 K WD chains per walker and population), then a wave per star
        if (!wd_tab) return hipErrorInvalidValue;
        hipLaunchKernelGGL((k_marg_wd_table<NFP>), dim3(n_walkers * NPOPS, (8 * K + 63) / 64), dim3(128), 0, stream, pk, hdr, iso_data, iso_stride,
                           mass_cap, NPOPS, d_params, K, wd_tab, n_walkers * NPOPS);
        hipLaunchKernelGGL((k_star_marg_wd<NFP, NPOPS, SAMPLE>), dim3((st.n_wd + 3) / 4, n_walkers), dim3(256), 0, stream, pk, st, hdr,
                           iso_data, iso_stride, mass_cap, d_params, partial, partial_stride, perstar, K, ms, wd_tab);
    }
    return hipGetLastError();
}

// The catalogue plan's counting pass: the unsplit star kernel on ONE row (the reference row), every wave leaving the number of
// (16 nodes x one mass ratio) units it evaluated in cost[star chunk][4].
template <int NFP, int NPOPS>
static hipError_t launch_star_marg_cost(const DevPack &pk, const DevStars &st, const IsoHdr *hdr, int mass_cap, const double *d_params,
                                        double *partial, long long partial_stride, int K, int Q, bool prune, const double *tab, unsigned *cost, hipStream_t stream)
{
    MargSample ms{};
    ms.cost = cost;
    const MargLayout L = marg_layout(NFP, mass_cap, K, Q);
    const int n_chunks = st.mg_pad / 64;
    const double cut2 = prune ? 2.0 * B9_MARG_CUT : __builtin_inf();
    hipLaunchKernelGGL((k_star_marg<NFP, NPOPS, false, false, true>), dim3(8 * ((n_chunks + 7) / 8)), dim3(256), 0, stream, pk, st, hdr, (const double *)nullptr, 0ll,
                       mass_cap, d_params, partial, partial_stride, (double *)nullptr, K, Q, ms, tab, L, 1, cut2, 1, (double *)nullptr);
    return hipGetLastError();
}

hipError_t b9k_star_marg_cost(const DevPack &pk, const DevStars &st, const IsoHdr *hdr, int mass_cap, const double *d_params, int n_pops,
                              double *partial, long long partial_stride, int K, int Q, bool prune, const double *tab, unsigned *cost, hipStream_t stream)
{
#define CS_ARGS pk, st, hdr, mass_cap, d_params, partial, partial_stride, K, Q, prune, tab, cost, stream
#define CS2(NFP) launch_star_marg_cost<NFP, 2>(CS_ARGS)
#define CS1(NFP) launch_star_marg_cost<NFP, 1>(CS_ARGS)
    B9_SWITCH_NFP(CS2, CS1)
#undef CS1
#undef CS2
#undef CS_ARGS
}

template <int NFP, int NPOPS>
static hipError_t launch_star_marg(const DevPack &pk, const DevStars &st, const IsoHdr *hdr, const double *iso_data,
                                   long long iso_stride, int mass_cap, const double *d_params, int n_walkers,
                                   double *partial, long long partial_stride, double *perstar, int K, int Q, const B9MargSample *smp, bool prune, double *tab, double *wd_tab, double *shares, hipStream_t stream)
{
    return smp ? launch_star_marg_t<NFP, NPOPS, true>(pk, st, hdr, iso_data, iso_stride, mass_cap, d_params, n_walkers, partial, partial_stride, perstar, K, Q, smp, prune, tab, wd_tab, shares, stream)
               : launch_star_marg_t<NFP, NPOPS, false>(pk, st, hdr, iso_data, iso_stride, mass_cap, d_params, n_walkers, partial, partial_stride, perstar, K, Q, smp, prune, tab, wd_tab, shares, stream);
}

hipError_t b9k_star_marg(const DevPack &pk, const DevStars &st, const IsoHdr *hdr, const double *iso_data,
                         long long iso_stride, int mass_cap, const double *d_params, int n_walkers, int n_pops,
                         double *partial, long long partial_stride, double *perstar, int K, int Q, const B9MargSample *smp, bool prune, double *tab, double *wd_tab, double *shares, hipStream_t stream)
{
#define SM_ARGS pk, st, hdr, iso_data, iso_stride, mass_cap, d_params, n_walkers, partial, partial_stride, perstar, K, Q, smp, prune, tab, wd_tab, shares, stream
#define SM2(NFP) launch_star_marg<NFP, 2>(SM_ARGS)
#define SM1(NFP) launch_star_marg<NFP, 1>(SM_ARGS)
    B9_SWITCH_NFP(SM2, SM1)
#undef SM1
#undef SM2
#undef SM_ARGS
}

// ---- the marginalised mode's fused sampler step (k_marg_step) -----------------------------------------------------------
// The node tables of a set of derived isochrones WITHOUT the star launch: the prologue of a fused block (its first
// proposal comes from k_derive_iso; every later one is built inside k_marg_step).
template <int NFP>
static hipError_t launch_marg_tables(const DevPack &pk, const IsoHdr *hdr, const double *iso_data, long long iso_stride, int mass_cap,
                                     const double *d_params, int n_walkers, int n_pops, int K, int Q, double *tab, double *wd_tab, hipStream_t stream)
{
    const MargLayout L = marg_layout(NFP, mass_cap, K, Q);
    const size_t lds = sizeof(double) * ((size_t)mass_cap + 8 + 8 * NFP);
    if (lds > 64 * 1024) {
        if (lds > 160 * 1024) return hipErrorInvalidValue;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_marg_table<NFP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((k_marg_table<NFP>), dim3(n_walkers * n_pops, L.n_chunks), dim3(256), lds, stream, pk, hdr, iso_data,
                       iso_stride, mass_cap, n_pops, d_params, K, Q, tab, L);
    if (wd_tab)
        hipLaunchKernelGGL((k_marg_wd_table<NFP>), dim3(n_walkers * n_pops, (8 * K + 63) / 64), dim3(128), 0, stream, pk, hdr, iso_data, iso_stride,
                           mass_cap, n_pops, d_params, K, wd_tab, n_walkers * n_pops);
    return hipGetLastError();
}

hipError_t b9k_marg_tables(const DevPack &pk, const IsoHdr *hdr, const double *iso_data, long long iso_stride, int mass_cap,
                           const double *d_params, int n_walkers, int n_pops, int K, int Q, double *tab, double *wd_tab, hipStream_t stream)
{
    switch (pk.nfp) {
    case 4:  return launch_marg_tables<4>(pk, hdr, iso_data, iso_stride, mass_cap, d_params, n_walkers, n_pops, K, Q, tab, wd_tab, stream);
    case 8:  return launch_marg_tables<8>(pk, hdr, iso_data, iso_stride, mass_cap, d_params, n_walkers, n_pops, K, Q, tab, wd_tab, stream);
    case 16: return launch_marg_tables<16>(pk, hdr, iso_data, iso_stride, mass_cap, d_params, n_walkers, n_pops, K, Q, tab, wd_tab, stream);
    default: return hipErrorInvalidValue;
    }
}

// Dynamic LDS of k_marg_step (its table builders' tiles; every workgroup of the launch gets it): the fused step runs only while it
// leaves the star role its workgroups per CU (B9_MSTEP_LDS_MAX: 160 KB / 7 less the kernel's static arrays, with 8 filters)
size_t b9k_marg_step_lds(int nfp, int mass_cap) { return sizeof(double) * B9_MSTEP_LDS_DOUBLES(nfp, mass_cap); }

template <int NFP, int NPOPS>
static hipError_t launch_marg_step(const DevPack &pk, const DevStars &st, const StepDev &sd, const DevPriors &pr, int K, int Q, bool prune,
                                   double *tab, double *wd_tab, long long wd_stride, double *shares, hipStream_t stream)
{
    const int W = sd.n_walkers;
    MargStep mx{};
    mx.K = K; mx.Q = Q; mx.L = marg_layout(NFP, sd.mass_cap, K, Q);
    mx.n_chunks_cap = mx.L.n_chunks;
    mx.n_wd_blocks = st.n_wd > 0 ? (8 * K + 127) / 128 : 0;
    mx.wsplit = W % 2 == 0 ? 2 : 1;
    const int csplit = 8 / mx.wsplit, n_chunks = st.mg_pad / 64;
    const bool split = st.mg_n_pieces > 0;
    if (split && !shares) return hipErrorInvalidValue;
    if (st.n_wd > 0 && !wd_tab) return hipErrorInvalidValue;
    mx.cut2 = prune ? 2.0 * B9_MARG_CUT : __builtin_inf();
    mx.tab = tab; mx.wd_tab = wd_tab; mx.wd_stride = wd_stride; mx.shares = shares;
    const int front = (W + W * 2 * NPOPS * (mx.n_chunks_cap + mx.n_wd_blocks) + 7) / 8 * 8;
    const int stars = 8 * ((((split ? st.mg_n_pieces : n_chunks) + csplit - 1) / csplit) * (W / mx.wsplit));
    const int wd = st.n_wd > 0 ? ((st.n_wd + 3) / 4) * W : 0;
    const size_t lds = b9k_marg_step_lds(NFP, sd.mass_cap);
    if (lds > B9_MSTEP_LDS_MAX(NFP)) return hipErrorInvalidValue;
    // this parity's two candidates, as the star roles read them (see MargStepSel)
    const size_t rows = (size_t)W * NPOPS, c0 = (size_t)sd.set * 2;
    const IsoHdr *hdr_rd = sd.cand_hdr + c0 * rows;
    const double *par_rd = sd.cand_par + c0 * W * B9_NPARAM, *tab_rd = tab + c0 * rows * mx.L.total, *wd_rd = wd_tab ? wd_tab + c0 * wd_stride : nullptr;
    static_assert(B9_MSTEP_LDS_DOUBLES(NFP, 2) >= 4 * B9_TILE_DOUBLES(NFP), "the star role's row tiles borrow the builders' dynamic LDS");
    if (split) {
        if (marg_sparse((long long)st.mg_n_pieces * W))
            hipLaunchKernelGGL((k_marg_step<NFP, NPOPS, true, 2>), dim3(front + stars + wd), dim3(256), lds, stream, pk, st, sd, pr, mx, front, stars,
                               hdr_rd, par_rd, tab_rd, wd_rd);
        else
            hipLaunchKernelGGL((k_marg_step<NFP, NPOPS, true>), dim3(front + stars + wd), dim3(256), lds, stream, pk, st, sd, pr, mx, front, stars,
                               hdr_rd, par_rd, tab_rd, wd_rd);
        hipLaunchKernelGGL((k_marg_step_merge<NPOPS>), dim3(n_chunks, W), dim3(64), 0, stream, st, sd, mx);
    } else {
        hipLaunchKernelGGL((k_marg_step<NFP, NPOPS, false>), dim3(front + stars + wd), dim3(256), lds, stream, pk, st, sd, pr, mx, front, stars,
                           hdr_rd, par_rd, tab_rd, wd_rd);
    }
    return hipGetLastError();
}

hipError_t b9k_marg_step(const DevPack &pk, const DevStars &st, const StepDev &sd, const DevPriors &pr, int K, int Q, bool prune,
                         double *tab, double *wd_tab, long long wd_stride, double *shares, hipStream_t stream)
{
    const int n_pops = sd.n_pops;
#define GS_ARGS pk, st, sd, pr, K, Q, prune, tab, wd_tab, wd_stride, shares, stream
#define GS2(NFP) launch_marg_step<NFP, 2>(GS_ARGS)
#define GS1(NFP) launch_marg_step<NFP, 1>(GS_ARGS)
    B9_SWITCH_NFP(GS2, GS1)
#undef GS1
#undef GS2
#undef GS_ARGS
}

// dynamic LDS of k_mcmc_step: the hot role's mass columns of both candidates (+ 8: find_bracket's masked over-read), or
// the heavy role's axes, whichever is larger
template <int NFP, int NPOPS>
static size_t mcmc_step_lds(const DevPack &pk, int mass_cap)
{
    return sizeof(double) * std::max((size_t)2 * NPOPS * mass_cap + 8, heavy_lds_doubles(pk, NPOPS, 2, mass_cap));
}

template <int NFP, int NPOPS>
static hipError_t mcmc_step_occupancy(const DevPack &pk, int mass_cap, int *blocks_per_cu)
{
    const size_t lds = mcmc_step_lds<NFP, NPOPS>(pk, mass_cap);
    auto kern = k_mcmc_step<NFP, NPOPS>;
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, reinterpret_cast<const void *>(kern), 256, lds);
}

// Workgroups of the fused step that one CU holds at once for the loaded pack (registers, LDS, wave slots of THIS
// instantiation): the launch plan sizes its single occupancy round from it instead of assuming a machine.
hipError_t b9k_mcmc_step_occupancy(const DevPack &pk, int n_pops, int mass_cap, int *blocks_per_cu)
{
#define OC2(NFP) mcmc_step_occupancy<NFP, 2>(pk, mass_cap, blocks_per_cu)
#define OC1(NFP) mcmc_step_occupancy<NFP, 1>(pk, mass_cap, blocks_per_cu)
    B9_SWITCH_NFP(OC2, OC1)
#undef OC1
#undef OC2
}

template <int NFP, int NPOPS>
static hipError_t launch_mcmc_step(const DevPack &pk, const DevStars &st, const StepDev &sd, const DevPriors &pr,
                                   const B9Groups &gr, int heavy_parts, int derive_parts, int derive_order, hipStream_t stream)
{
    const size_t lds = mcmc_step_lds<NFP, NPOPS>(pk, sd.mass_cap);
    auto kern = k_mcmc_step<NFP, NPOPS>;
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    const int W = sd.n_walkers;
    const int hot = 8 * ((gr.n_blocks * NPOPS + 7) / 8) * W;
    const int derive_first = derive_order >= 0 ? (derive_order == 1 ? 2 : 1) : 0;
    const int n_derive = W * 2 * NPOPS * derive_parts;
    const int front = (W * heavy_parts + W + (derive_first ? n_derive : 0) + 7) / 8 * 8;     // heavy, writers, (derivation), pad
    const int back = (!derive_first && sd.derive_next) ? n_derive : 0;
    hipLaunchKernelGGL(kern, dim3(front + hot + back), dim3(256), lds, stream, pk, st, sd, pr, gr.group_tiles, gr.n_groups,
                       gr.groups_per_block, gr.n_blocks, front, hot, heavy_parts, derive_parts, derive_first);
    return hipGetLastError();
}

hipError_t b9k_mcmc_step(const DevPack &pk, const DevStars &st, const StepDev &sd, const DevPriors &pr,
                         const B9Groups &gr, int heavy_parts, int derive_parts, int derive_order, hipStream_t stream)
{
    const int n_pops = sd.n_pops;
#define MS_ARGS pk, st, sd, pr, gr, heavy_parts, derive_parts, derive_order, stream
#define MS2(NFP) launch_mcmc_step<NFP, 2>(MS_ARGS)
#define MS1(NFP) launch_mcmc_step<NFP, 1>(MS_ARGS)
    B9_SWITCH_NFP(MS2, MS1)
#undef MS1
#undef MS2
#undef MS_ARGS
}

// ---- tree-speculative step (k_mcmc_tree) -------------------------------------------------------------------------
template <int NFP, int NPOPS>
static size_t mcmc_tree_lds(const DevPack &pk, int mass_cap)
{
    return sizeof(double) * std::max((size_t)NPOPS * mass_cap + 8, heavy_lds_doubles(pk, NPOPS, 1, mass_cap));
}

template <int NFP, int NPOPS>
static hipError_t mcmc_tree_occupancy(const DevPack &pk, int mass_cap, int n_groups, int *blocks_per_cu)
{
    const size_t lds = mcmc_tree_lds<NFP, NPOPS>(pk, mass_cap);
    auto kern = n_groups > 16 * B9_TREE_KD_SMALL ? tree_kd5::k_mcmc_tree<NFP, NPOPS> : tree_kd3::k_mcmc_tree<NFP, NPOPS>;
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, reinterpret_cast<const void *>(kern), 256, lds);
}

hipError_t b9k_mcmc_tree_occupancy(const DevPack &pk, int n_pops, int mass_cap, int n_groups, int *blocks_per_cu)
{
#define OC2(NFP) mcmc_tree_occupancy<NFP, 2>(pk, mass_cap, n_groups, blocks_per_cu)
#define OC1(NFP) mcmc_tree_occupancy<NFP, 1>(pk, mass_cap, n_groups, blocks_per_cu)
    B9_SWITCH_NFP(OC2, OC1)
#undef OC1
#undef OC2
}

template <int NFP, int NPOPS>
static hipError_t launch_mcmc_tree(const DevPack &pk, const DevStars &st, const TreeDev &td, const DevPriors &pr, int group_tiles,
                                   int derive_parts, hipStream_t stream)
{
    const size_t lds = mcmc_tree_lds<NFP, NPOPS>(pk, td.mass_cap);
    auto kern = td.n_groups > 16 * B9_TREE_KD_SMALL ? tree_kd5::k_mcmc_tree<NFP, NPOPS> : tree_kd3::k_mcmc_tree<NFP, NPOPS>;
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    const int W = td.n_walkers, NN = (1 << td.depth) - 1, NO = td.derive_mode == 2 ? 1 : (1 << td.depth);
    const int writers = W;                                   // (prologue: the step-table workgroups)
    const int n_derive = td.derive_mode == 0 ? 0 : W * NO * NN * NPOPS * derive_parts;
    const int heavy = td.levels > 0 ? W * NN * td.heavy_parts : 0;
    const int front = (writers + n_derive + heavy + 7) / 8 * 8;
    const int hot = td.levels > 0 ? 8 * ((td.n_groups * NPOPS + 7) / 8) * W * NN : 0;
    hipLaunchKernelGGL(kern, dim3(front + hot), dim3(256), lds, stream, pk, st, td, pr, group_tiles, front, derive_parts);
    return hipGetLastError();
}

hipError_t b9k_mcmc_tree(const DevPack &pk, const DevStars &st, const TreeDev &td, const DevPriors &pr, int group_tiles,
                         int derive_parts, hipStream_t stream)
{
    const int n_pops = td.n_pops;
#define MT2(NFP) launch_mcmc_tree<NFP, 2>(pk, st, td, pr, group_tiles, derive_parts, stream)
#define MT1(NFP) launch_mcmc_tree<NFP, 1>(pk, st, td, pr, group_tiles, derive_parts, stream)
    B9_SWITCH_NFP(MT2, MT1)
#undef MT1
#undef MT2
}

hipError_t b9k_tree_finish(const TreeDev &td, const DevPriors &pr, hipStream_t stream)
{
    if (td.n_groups > 16 * B9_TREE_KD_SMALL) hipLaunchKernelGGL(tree_kd5::k_tree_finish, dim3(td.n_walkers), dim3(256), 0, stream, td, pr);
    else hipLaunchKernelGGL(tree_kd3::k_tree_finish, dim3(td.n_walkers), dim3(256), 0, stream, td, pr);
    return hipGetLastError();
}

hipError_t b9k_tree_begin(const double *host_up, double *dev, int up_words, const double *prev_final, double *state, int n_walkers, hipStream_t stream)
{
    hipLaunchKernelGGL(tree_kd3::k_tree_begin, dim3(1), dim3(256), 0, stream, host_up, dev, up_words, prev_final, state, n_walkers);
    return hipGetLastError();
}

hipError_t b9k_mcmc_begin(const double *host_up, double *dev, int up_words, const double *prev_final, double *cur0, double *lp0,
                          double *state0, int n_walkers, hipStream_t stream)
{
    hipLaunchKernelGGL(k_mcmc_begin, dim3(1), dim3(256), 0, stream, host_up, dev, up_words, prev_final, cur0, lp0, state0, n_walkers);
    return hipGetLastError();
}

hipError_t b9k_mcmc_continue(const double *prev_final, double *cur0, double *lp0, double *state0, int n_walkers, hipStream_t stream)
{
    hipLaunchKernelGGL(k_mcmc_continue, dim3(n_walkers), dim3(64), 0, stream, prev_final, cur0, lp0, state0);
    return hipGetLastError();
}

hipError_t b9k_mcmc_finish(const DevPack &pk, const StepDev &sd, const DevPriors &pr, hipStream_t stream)
{
    hipLaunchKernelGGL(k_mcmc_finish, dim3(sd.n_walkers), dim3(256), 0, stream, pk, sd, pr);
    return hipGetLastError();
}

hipError_t b9k_chain_rows(const StepDev &sd, const double *cur_fin, const double *lp_fin, hipStream_t stream)
{
    hipLaunchKernelGGL(k_chain_rows, dim3(sd.n_walkers), dim3(256), 0, stream, sd, cur_fin, lp_fin);
    return hipGetLastError();
}

// An empty kernel: bracketing it with HIP events measures what an event bracket adds to a kernel's
// own duration (dispatch boundary + event processing); b9_calibrate_timing subtracts nothing by
// itself, it only reports the figure.
__global__ void k_noop(int *p) { if (p && threadIdx.x == 1024) *p = 0; }
// keeps the queue busy for ~`ticks` s_memrealtime ticks (100 MHz) so that work enqueued behind it
// executes back to back (bounded spin; one wave)
__global__ void k_spin(unsigned long long ticks, int *p)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    int guard = 0;
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks && guard < (1 << 24)) ++guard;
    if (p && guard < 0) *p = guard;
}
hipError_t b9k_spin(double microseconds, hipStream_t stream)
{
    hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, stream, (unsigned long long)(microseconds * 100.0), (int *)nullptr);
    return hipGetLastError();
}
hipError_t b9k_noop(hipStream_t stream)
{
    hipLaunchKernelGGL(k_noop, dim3(1), dim3(64), 0, stream, (int *)nullptr);
    return hipGetLastError();
}

// (shader-cycle counter, 100 MHz reference counter) stamped per COMPUTE UNIT by whichever of the launch's workgroups landed
// there: two such stamps bracket a stretch of stream work, and delta(s_memtime) / delta(s_memrealtime) x 100 MHz of one CU
// is the clock its shader engine actually ran at over it (MI355X_MICROARCH.md: the in-kernel clock, not pp_dpm_sclk).
// The cycle counters of different CUs are offset against each other by arbitrary amounts (measured: tens of millions of
// cycles), so a difference is only ever formed between two stamps of the SAME CU.  out[cu slot] = {t, tr}.
__global__ void k_clock_stamp(unsigned long long *out)
{
    if (threadIdx.x != 0) return;
    unsigned xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    unsigned long long t, tr;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t), "=s"(tr) :: "memory");
    const unsigned slot = ((xcc & 7u) << 8) | ((hw >> 8) & 0xFFu);      // HW_ID[15:8] = shader engine, shader array, CU
    // several workgroups land on one CU: the FIRST to claim the (zeroed) slot writes both words, so a pair is one workgroup's
    if (atomicCAS(out + 2 * slot, 0ull, t) == 0ull) out[2 * slot + 1] = tr;
}
hipError_t b9k_clock_stamp(unsigned long long *d_out, hipStream_t stream)
{
    hipLaunchKernelGGL(k_clock_stamp, dim3(2048), dim3(64), 0, stream, d_out);
    return hipGetLastError();
}
