// b9_launch.h -- host-callable launch wrappers of the kernels in b9_kernels.hip.
#pragma once
#include "b9_device.h"

// what k_derive_iso needs to FINISH the previous MCMC step before proposing the next one
struct B9Prev {
    const double *partial;      // the star kernel's partials of the previous launch
    int n_partial;
    long long partial_stride;
    const IsoHdr *hdr;          // the previous step's isochrone headers (validity)
    const double *params;       // the previous step's proposal rows
};

hipError_t b9k_derive_iso(const DevPack &pk, double *d_params, int n_walkers, int n_pops,
                          IsoHdr *hdr, double *iso_data, long long iso_stride, int mass_cap,
                          const McmcDev &mc, const DevPriors &pr, const B9Prev &prev, hipStream_t stream);

// up to 8 HOST rows carried in the kernel arguments (no upload); also stored to d_params for the later launches
hipError_t b9k_derive_iso_rows(const DevPack &pk, const double *host_rows, double *d_params, int n_walkers, int n_pops,
                               IsoHdr *hdr, double *iso_data, long long iso_stride, int mass_cap, hipStream_t stream);

// How a launch's hot workgroups take the catalogue's CANONICAL tile groups (b9_star_like.hip.h): n_groups groups of group_tiles
// tiles are fixed by the catalogue (so is the partial-sum layout: 4 per group); a workgroup takes groups_per_block of them,
// n_blocks workgroups per walker.
struct B9Groups { int group_tiles, n_groups, groups_per_block, n_blocks; };

hipError_t b9k_star_like(const DevPack &pk, const DevStars &st, const IsoHdr *hdr,
                         const double *iso_data, long long iso_stride, int mass_cap,
                         const double *d_params, int n_walkers, int n_pops,
                         double *partial, long long partial_stride, double *perstar, const B9Groups &gr,
                         int heavy_parts, hipStream_t stream);

hipError_t b9k_finalize(const IsoHdr *hdr, const double *partial, int n_partial, long long partial_stride,
                        int n_pops, const double *d_params, const DevPriors &pr, int n_walkers, double *d_logpost,
                        double *perstar, int n_stars, const McmcDev &mc, hipStream_t stream,
                        unsigned long long *done_flag = nullptr /* [n_walkers] in mapped host memory: done_seq is stored there behind the log-posterior */,
                        unsigned long long done_seq = 0);

// b9_sample_mass: where the per-star draws go (device pointers), RNG key and the global index of row 0
struct B9MargSample {
    double *mass, *ratio, *member;
    int *pop;
    unsigned k0, k1;
    long long row0;
};

// smp == nullptr: the plain marginal likelihood; else every star also draws one (mass, ratio[, population]) node
hipError_t b9k_star_marg(const DevPack &pk, const DevStars &st, const IsoHdr *hdr, const double *iso_data,
                         long long iso_stride, int mass_cap, const double *d_params, int n_walkers, int n_pops,
                         double *partial /* per walker: one sum per 64-star chunk, then one value per WD-stage star */, long long partial_stride,
                         double *perstar, int K, int Q, const B9MargSample *smp, bool prune /* false: every node of every star is evaluated */,
                         double *tab /* the call's node table: n_walkers * n_pops * b9k_marg_table_doubles(nfp, mass_cap, K, Q) doubles */,
                         double *wd_tab /* the WD-stage stars' node table: n_walkers * n_pops * b9k_marg_wd_table_doubles(nfp, K) doubles (used when the catalogue has WD-stage stars) */,
                         double *shares /* per-star shares of split launches: n_walkers * b9k_marg_shares_doubles(pieces, n_pops) doubles */,
                         hipStream_t stream);
int b9k_marg_split(int n_star_chunks, int n_pops);           // 1: the catalogue's star chunks are split into pieces (DevStars::mg_piece)
long long b9k_marg_shares_doubles(int n_pieces, int n_pops);   // per walker
// the catalogue plan's counting pass (one row, unsplit): cost[star chunk][4] = units each wave evaluated
hipError_t b9k_star_marg_cost(const DevPack &pk, const DevStars &st, const IsoHdr *hdr, int mass_cap, const double *d_params, int n_pops,
                              double *partial, long long partial_stride, int K, int Q, bool prune, const double *tab, unsigned *cost, hipStream_t stream);
long long b9k_marg_table_doubles(int nfp, int mass_cap, int K, int Q);
long long b9k_marg_wd_table_doubles(int nfp, int K);

// marginalised mode, fused sampler step (b9_marg_step.hip.h): decision of step t-1 + stars of step t against one of its two candidate
// node tables + both candidate tables of step t+1 (+ the merge launch of a split catalogue).  sd: StepDev with heavy_parts = 0,
// n_partial = star chunks + WD-stage stars, cand_iso unused.  tab / wd_tab: [2 parities][2 candidates] blocks of n_walkers * n_pops
// tables (b9k_marg_table_doubles / wd_stride doubles each).
hipError_t b9k_marg_step(const DevPack &pk, const DevStars &st, const StepDev &sd, const DevPriors &pr, int K, int Q, bool prune,
                         double *tab, double *wd_tab, long long wd_stride, double *shares, hipStream_t stream);
// the tables alone (k_marg_table [+ k_marg_wd_table when wd_tab]) of derived isochrones: the fused block's prologue
hipError_t b9k_marg_tables(const DevPack &pk, const IsoHdr *hdr, const double *iso_data, long long iso_stride, int mass_cap,
                           const double *d_params, int n_walkers, int n_pops, int K, int Q, double *tab, double *wd_tab, hipStream_t stream);
size_t b9k_marg_step_lds(int nfp, int mass_cap);
// the most dynamic LDS k_marg_step may take and keep the star role's occupancy (7 workgroups per CU up to 8 filters, 4 with 16)
#define B9_MSTEP_LDS_MAX(nfp) ((nfp) > 8 ? (size_t)30 * 1024 : (size_t)15 * 1024)

// fused sampler step (given-mass mode): decision of step t-1 + stars of step t + candidates of step t+1
// (tiles_per_block < 0: a workgroup's |tiles_per_block| tiles are strided n_groups apart instead of consecutive)
hipError_t b9k_mcmc_step(const DevPack &pk, const DevStars &st, const StepDev &sd, const DevPriors &pr,
                         const B9Groups &gr, int heavy_parts, int derive_parts,
                         int derive_order /* >= 0: derivation workgroups lead the grid, < 0: they trail it */, hipStream_t stream);
// resident workgroups per CU of k_mcmc_step's instantiation for this pack (occupancy query; no launch)
hipError_t b9k_mcmc_step_occupancy(const DevPack &pk, int n_pops, int mass_cap, int *blocks_per_cu);
// B9_BLOCK_CONTINUE: previous block's final state rows -> this block's starting buffers
// the block's opening: upload from the mapped host mirror (+ the previous block's final state when continuing), one launch
hipError_t b9k_mcmc_begin(const double *host_up, double *dev, int up_words, const double *prev_final, double *cur0, double *lp0,
                          double *state0, int n_walkers, hipStream_t stream);
hipError_t b9k_mcmc_continue(const double *prev_final, double *cur0, double *lp0, double *state0, int n_walkers, hipStream_t stream);
// the block's last decision only (one workgroup per walker)
hipError_t b9k_mcmc_finish(const DevPack &pk, const StepDev &sd, const DevPriors &pr, hipStream_t stream);

// tree-speculative step (given-mass mode, few walkers): one launch = `depth` steps of every chain (TreeDev in b9_device.h)
hipError_t b9k_mcmc_tree(const DevPack &pk, const DevStars &st, const TreeDev &td, const DevPriors &pr, int group_tiles /* tiles of a canonical group */,
                         int derive_parts, hipStream_t stream);
hipError_t b9k_mcmc_tree_occupancy(const DevPack &pk, int n_pops, int mass_cap, int n_groups /* of the catalogue: selects the kernel build */, int *blocks_per_cu);
hipError_t b9k_tree_finish(const TreeDev &td, const DevPriors &pr, hipStream_t stream);
hipError_t b9k_tree_begin(const double *host_up, double *dev, int up_words, const double *prev_final, double *state, int n_walkers, hipStream_t stream);

// summary rows of a two-launch block from its chain record on the device (sd: d, n_walkers, n_steps, samples, free_idx, row_origin, rows)
hipError_t b9k_chain_rows(const StepDev &sd, const double *cur_fin, const double *lp_fin, hipStream_t stream);

hipError_t b9k_noop(hipStream_t stream);
hipError_t b9k_spin(double microseconds, hipStream_t stream);
constexpr int B9_CLOCK_SLOTS = 8 * 256;      // (XCD, HW_ID[15:8]) -> one slot per compute unit
hipError_t b9k_clock_stamp(unsigned long long *d_out /* [B9_CLOCK_SLOTS][2] */, hipStream_t stream);
