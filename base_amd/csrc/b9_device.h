// b9_device.h -- device-side data layout shared by the kernels and the C-ABI host code.
// gfx950 only.  See DESIGN.md "Data layout in HBM".
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define B9_MAX_FILT 16          // padded filter count limit (NFP in {4, 8, 16})
#define B9_WAVE 64
#define B9_TIPS_LDS_MAX 2048       // the whole AGB-tip table (n_feh * n_y * n_age) is staged in LDS up to this many doubles
#define B9_WC_AGE_LDS_MAX 6144     // cooling-track age axes (all tracks, concatenated) are staged in LDS up to this many doubles

// Model pack, device view.  Passed to kernels by value (kernarg segment).
struct DevPack {
    int nf, nfp;                     // real / padded filter count
    int n_feh, n_y, n_age;
    int max_eep;                     // longest isochrone in the pack
    const double *feh, *y, *log_age;
    const int *first, *cnt;          // per isochrone
    const long long *off;
    const double *mass;              // [n_points]
    const double *mags;              // [n_points * nfp]  (rows padded with zeros to nfp)
    const double *tips;              // [n_iso] mass of each isochrone's last point
    double abs_m1[B9_MAX_FILT];      // abs_coeff[f] - 1   (0 in padded columns)
    // WD cooling
    // WD cooling: one track per (carbonicity, mass) node with its own age axis (ragged); track t = ic * n_wc_mass + im
    int n_wc_carb, n_wc_mass, n_wc_points;
    int wc_uniform;                  // every track has the SAME age axis (a rectangular table): one bracket search serves all tracks
    int wc_n0, wc_off0;              // ... of wc_n0 points, one copy of which starts at point wc_off0
    const double *wc_track;          // [n_wc_carb * n_wc_mass] per track: (points | first point << 32) in the bits of a double, so that
                                     // the heavy-star role stages it in LDS with its other (double) axes
    const double *wc_carb, *wc_mass, *wc_log_age, *wc_log_teff, *wc_log_radius;   // the last three: [n_wc_points]
    // WD atmospheres, rows padded to nfp
    int n_at_type, n_at_logg, n_at_teff;
    const double *at_logg, *at_log_teff, *at_mags;
    // The axes the heavy-star role searches, packed back to back in the order [log_age | cooling-age axes (the shared one of
    // a rectangular table, all tracks' of a ragged one; absent when too long for LDS) | wc_mass | wc_carb | at_log_teff |
    // at_logg | wc_track]: the role copies this ONE run to LDS at its entry (7 runs of 5-100 entries cost 7 loads per thread).
    const double *heavy_const;
    int hc_len, hc_off[7];           // entries; where each axis starts; hc_age_staged: the cooling-age axes are part of it
    int hc_age_staged;
    int ifmr_id;
    double m_wd_up;
    double log_mass_norm;
};

// Stars, device view: structure of arrays in the slot order chosen at load time (64-star
// chunks of mass-sorted singles and of mass-sorted binaries, interleaved; see build_stars),
// padded to whole 256-star tiles.  Empty slots have mass1 = +inf and perm = -1.
struct DevStars {
    int n, n_pad;                    // real stars / slots
    // Observations and weights, CHUNK-major: element (filter f, slot i) is at ((i >> 6) * nfp + f) * 64 + (i & 63) -- a wave's
    // 64 stars of one filter are 512 contiguous bytes and a star's filters sit 512 bytes apart, so the hot role forms ONE
    // address per array and star and reaches the filters through the loads' immediate offsets (with [nfp][n_pad] every one
    // of its 16 loads needed its own 64-bit multiply-add: a tenth of the role's VALU instructions).  B9_SIDX(st, nfp, f, i).
    const double *obs;               // [n_pad / 64][nfp][64]
    const double *w;                 // [n_pad / 64][nfp][64]  1/sigma^2, 0 = filter unused
    const double *mass1, *q;         // [n_pad]
    const double *c0;                // [n_pad] log p + logPriorMass(mass1) + sum_f -0.5 log(2 pi sigma_f^2)
    const double *c0m;               // [n_pad] log p + sum_f -0.5 log(2 pi sigma_f^2)   (marginalised mode)
    const double *la;                // [n_pad] log((1-p) fsLike)  (-inf when p == 1)
    const double *ea;                // [n_pad] (1-p) fsLike = exp(la)   (the hot kernel's form)
    const int *flags;                // [n_pad] bit0 = DB atmosphere, bits 8.. = stage
    const int *perm;                 // [n_pad] original index of the star in slot i, -1 = empty
    const double *heavy_mass;        // [n] primary masses in descending order ...
    const int *heavy_slot;           // [n] ... and the slots that hold them
    // A second copy of the per-star arrays IN THAT DESCENDING-MASS ORDER (stride hv_pad), read by the heavy-star role:
    // star j of the list is element j of every array, so the role's first memory round trip fetches everything a star
    // needs (through heavy_slot it took one trip for the slot and another for the data behind it).
    int hv_pad;
    const double *hv_obs, *hv_w;     // [nfp][hv_pad]
    const double *hv_q, *hv_c0, *hv_la;   // [hv_pad]
    const int *hv_flags, *hv_perm;   // [hv_pad]  (hv_perm: original index of the star)
    // the slots of the stars the catalogue marks as white dwarfs (stage WD), ascending: the marginalised mode evaluates
    // them in a launch of their own (k_star_marg_wd)
    int n_wd;
    const int *wd_slot;              // [max(1, n_wd)]
    // Marginalised mode: its own copy of the stars (WD-stage stars excluded: they are k_star_marg_wd's), in 64-star chunks of
    // stars that are PHOTOMETRIC neighbours -- sorted by the first principal component of the catalogue's magnitudes.  A wave
    // of k_star_marg evaluates 64 stars against the union of the node windows that matter for them, so its cost is that
    // union's size: neighbours in brightness share their windows (neighbours in catalogue mass do not: a binary sits with
    // singles of another brightness, and the catalogue's masses are only hints in this mode).  Speed only.
    int mg_pad;                      // slots of the copy (whole chunks)
    const double *mg_so, *mg_sw;     // [mg_pad / 64][nfp][64]: sqrt(w) obs, sqrt(w) (w = 1 / sigma^2; 0 = unused filter)
    const double *mg_c0m, *mg_la;    // [mg_pad]
    const int *mg_perm;              // [mg_pad] original index of the star, -1 = empty
    // ... and the order in which the chunks are dispatched: most expensive first, so that the launch does not end on its
    // heaviest workgroups.  Cost = the node-table units a chunk's waves evaluate at the catalogue's REFERENCE parameter row
    // (the prior means), counted once per (catalogue, pack, priors, options) by a pass of the kernel itself
    // (b9_capi_margplan.cpp); before that pass, and when the reference row lies outside the grid: photometric spread.
    const int *marg_order;           // [mg_pad / 64]
    // Small catalogues (fewer star chunks than the chip has workgroup slots): a chunk's window is shared by 1 .. 16 workgroups
    // ("pieces": piece s of n takes the node chunks s, s + n, ...), as many as its measured cost asks for, so that the pieces
    // cost about the same and the launch does not last as long as its heaviest chunk of giants.  mg_piece: dispatch order
    // (most expensive piece first) of (chunk | piece << 20 | pieces of the chunk << 25); mg_share_base[c]: first piece id of
    // chunk c (prefix sums; [n_chunks + 1]).  mg_n_pieces == 0: no split -- one workgroup per chunk, in marg_order.
    int mg_n_pieces;
    const int *mg_piece;             // [mg_n_pieces]
    const int *mg_share_base;        // [mg_pad / 64 + 1]
};

#define B9_SIDX(nfp, f, i) ((((size_t)((i) >> 6) * (nfp)) + (f)) * 64 + ((i) & 63))

// ------------------------------------------------------------------------------------------
// Marginalised mode: layout of ONE (walker, population)'s per-call node table (k_marg_table writes it, k_star_marg
// reads it through SCALAR loads: a wave evaluates 64 stars against one node at a time, so a node's words are
// wave-uniform).  Nodes come in chunks of 64 = 4 sub-chunks of 16; npad = whole chunks of the longest isochrone.
// Everything starts on a 64-byte boundary (offsets in doubles, multiples of 8).
//   rows    [chunk][sub][j = 0..Q-1][node 0..15][f]   APPARENT magnitude of the system (node, mass ratio j / Q) in filter f
//                                                     (j = 0: the primary alone); modulus and absorption already added
//   nb      [node]                                    -2 log(prior(m1) dM / Q)  (+inf: no such node / empty EEP interval)
//   box2    [chunk][sub][j]{lo[f], hi[f]}             brightest / faintest magnitude among the unit's 16 rows
//   nbmin16 [chunk][sub]                              smallest nb of the sub-chunk
//   box1    [chunk]{lo[f], hi[f]}                     the same over the chunk's 64 nodes and all mass ratios
//   nbmin64 [chunk]
//   box2f, box1f                                      the same boxes as FLOATS rounded outward (nfp doubles' worth each: the
//                                                     packed-fp32 box test, b9_star_marg.hip.h box_bound32)
// ------------------------------------------------------------------------------------------
struct MargLayout {
    int npad, n_chunks, nfp, Q;
    long long o_rows, o_nb, o_box2, o_nbmin16, o_box1, o_nbmin64, o_box2f, o_box1f, total;
};

static inline __host__ __device__ MargLayout marg_layout(int nfp, int mass_cap, int K, int Q)
{
    MargLayout L;
    L.n_chunks = ((mass_cap - 1) * K + 63) / 64;
    if (L.n_chunks < 1) L.n_chunks = 1;
    L.npad = L.n_chunks * 64; L.nfp = nfp; L.Q = Q;
    long long o = 0;
    L.o_rows = o;    o += (long long)L.npad * Q * nfp;
    L.o_nb = o;      o += L.npad;
    L.o_box2 = o;    o += (long long)L.n_chunks * 4 * Q * 2 * nfp;
    L.o_nbmin16 = o; o += (L.n_chunks * 4 + 7) / 8 * 8;
    L.o_box1 = o;    o += (long long)L.n_chunks * 2 * nfp;
    L.o_nbmin64 = o; o += (L.n_chunks + 7) / 8 * 8;
    L.o_box2f = o;   o += (long long)L.n_chunks * 4 * Q * nfp;
    L.o_box1f = o;   o += (long long)L.n_chunks * nfp;
    L.total = o;
    return L;
}

// Header of one derived isochrone (one per walker x population).
struct IsoHdr {
    int valid, first_eep, n, i_feh, i_y, i_age;
    double agb_tip, t_feh, t_y, t_age;
};

struct DevPriors {
    double mean[12], var[12];
    double log_age_min, log_age_max;
};

// Device-resident Metropolis state of the local walkers (b9_mcmc_run_block), TWO-LAUNCH step (marginalised
// mode; given-mass mode uses the fused step, StepDev below).  Passed by value;
// enabled == 0 makes the kernels behave as the plain log-posterior path.
//
// State is kept twice (ping-pong): the derive kernel of step t first FINISHES step t-1 -- every
// workgroup of a walker re-sums that walker's partials, adds the prior and takes the same
// accept/reject decision from state half `pin`; one of them stores the new state into the other
// half -- and then draws the proposal of step t from it.  A step is therefore two launches.
struct McmcDev {
    int enabled, d;                  // d = number of free parameters (<= 11)
    int has_prev;                    // k_derive_iso: finish (sum + prior + accept) the previous step first
    int pin;                         // half of cur / lp_cur that holds the state on entry
    int n_walkers;                   // W: stride of the halves and of the chain records
    int row;                         // row of the chain record written by THIS launch's accept
    double *cur;                     // [2][W][12] positions
    double *lp_cur;                  // [2][W]
    const double *chol;              // [d][d] row-major proposal factor
    const int *free_idx;             // [d]
    const int *walker_ids;           // [W] global walker ids (RNG streams)
    unsigned k0, k1;                 // Philox key (seed)
    unsigned long long step;         // k_derive_iso: step being PROPOSED; k_finalize: step being ACCEPTED
    double *samples;                 // [n_steps][W][d] or null
    double *lps;                     // [n_steps][W] or null
    unsigned long long *n_acc;       // accepted proposals
};

// ------------------------------------------------------------------------------------------
// Fused sampler step (given-mass mode): ONE launch per MCMC step.
//
// Launch K(t) evaluates the proposal of step t.  Every workgroup first takes -- redundantly, with
// identical bits -- the accept/reject decision of step t-1 from that step's partial sums; the star
// workgroups then evaluate step t's proposal, and a few extra workgroups derive, speculatively,
// BOTH candidate isochrone sets of step t+1 (proposal drawn from the state if step t is rejected /
// from step t's proposal if it is accepted; the normals of step t+1 do not depend on the outcome).
// K(t+1) picks the candidate that matches its own decision.  The chain is the one the sequential
// algorithm produces (same draws, same sums, same comparisons).
//
// Everything ping-pongs on the step's parity `set` = t & 1:
//   state[set]    written by K(t):   cur_t, lp_t, proposal p_t, its log-prior, which candidate it was
//   partial[set]  written by K(t):   per-wave partial sums of p_t's star likelihoods
//   cand[set]     read by K(t):      two candidate (params, headers, isochrones) per walker
//   cand[set ^ 1] written by K(t):   the candidates of step t+1
// and K(t) reads state[set ^ 1] / partial[set ^ 1] (step t-1) for its decision.
// ------------------------------------------------------------------------------------------
#define B9_STATE_STRIDE 32       // doubles per walker and parity:
#define B9_ST_CUR 0              //   [0..11]  position after the last finished step
#define B9_ST_LP 12              //   [12]     its log-posterior
#define B9_ST_PROP 13            //   [13..24] the proposal the writing launch evaluates
#define B9_ST_LPRIOR 25          //   [25]     log-prior of that proposal; -inf = outside the grid or the prior's support
#define B9_ST_SEL 26             //   [26]     0 / 1: which candidate that proposal was
#define B9_ST_LOGU 27            //   [27]     log u of that proposal's accept test (drawn by the writer, so no other workgroup has to)
#define B9_ST_NACC 28            //   [28]     proposals of this walker accepted so far in the block (carried from row to row: no atomic)

struct StepDev {
    int d, n_walkers, n_pops;
    int has_prev;                    // 0 for the first launch of a block: no decision to take, candidate 0
    int derive_next;                 // 0 for the launch that only finishes the block's last step
    int set;                         // parity of the step this launch evaluates
    int row;                         // chain row the decision of this launch appends (step t-1)
    int n_partial, mass_cap;         // n_partial: partials one decision adds = hot waves' + heavy_parts (of the evaluated candidate)
    int heavy_parts;                 // heavy-star workgroups per walker; each writes one partial PER CANDIDATE: the partial row holds
                                     // [hot: n_partial - heavy_parts][heavy, candidate 0: heavy_parts][heavy, candidate 1: heavy_parts]
    unsigned k0, k1;                 // Philox key
    unsigned long long step;         // global index of the step this launch evaluates
    long long partial_stride;        // doubles between walkers; the two parities sit partial_stride / 2 apart
    long long iso_stride;
    double *state;                   // [2][W][B9_STATE_STRIDE]
    double *partial;                 // [W][partial_stride]
    double *cand_par;                // [2 sets][2 candidates][W][12]
    IsoHdr *cand_hdr;                // [2][2][W * pops]
    double *cand_iso;                // [2][2][W * pops][iso_stride]
    const double *chol;              // [d][d]
    const int *free_idx;             // [d]
    const int *walker_ids;           // [W]
    double *samples;                 // [n_steps][W][d] or null
    double *lps;                     // [n_steps][W] or null
    unsigned long long *n_acc;
    unsigned long long *decided;     // [W]: (step << 1 | sel) once this launch's writer has taken walker w's decision
    // summary rows of the block (k_mcmc_finish only; null = none): one row of B9_ROW_DOUBLES(d) per walker
    double *rows;                    // [W][15 + d + d*d]
    const double *row_origin;        // [d] common origin of the moments
    int n_steps;                     // steps of the block = rows of `samples` the summary covers
    // k_mcmc_finish only, both nullable: the block's pinned host mirror mapped into the device.  The final state rows and
    // the summary rows are ALSO written there, so a block whose chain record stays on the device needs no download copy.
    double *host_state;              // [W][B9_STATE_STRIDE] (the mirror's rows of the final parity)
    double *host_rows;               // [W][B9_ROW_LEN(d)]
};

// Summary row of one walker over one block (b9_mcmc_block::rows; include/base9_hip.h documents the layout).
#define B9_ROW_LP 0
#define B9_ROW_POS 1                 // [1..12] position after the block
#define B9_ROW_MOVED 13              // steps after the block's first on which the walker moved
#define B9_ROW_N 14                  // steps of the block
#define B9_ROW_SUM 15                // [15 .. 15+d) sum of x,  then [15+d .. 15+d+d*d) sum of x x^T;  x = sample - origin
#define B9_ROW_LEN(d) (15 + (d) + (d) * (d))

// ------------------------------------------------------------------------------------------
// Tree-speculative sampler step (given-mass mode, few walkers per GPU): ONE launch advances every chain by `depth` steps.
//
// A chain's next `depth` steps form a binary tree of proposals: the proposal of step T+j depends only on which of the
// steps T+1 .. T+j-1 were accepted (its base is the latest accepted proposal, or the state x after step T) and on the
// step's own normals, which are counter-based -- so all 2^depth - 1 proposals of the tree are known before any of them
// is evaluated.  Launch K(m) evaluates the WHOLE tree rooted at the chain's state (one set of star workgroups per node,
// on the workgroup slots a one-walker launch leaves idle); the next launch walks it -- level 1's accept test, then the
// level-2 node that outcome selects, ... -- which is exactly the sequential algorithm's sequence of tests on exactly its
// proposals: same draws, same sums, same comparisons, same chain.  As in the one-step fused launch, K(m) also derives the
// candidate isochrones of K(m+1) speculatively: one tree per possible outcome of its own tree (2^depth outcomes x
// 2^depth - 1 nodes; depth 1 is the one-step scheme's two candidates).
//
// Node n of a tree: level j = floor(log2(n + 1)) + 1, prefix p = n + 1 - 2^(j-1) = the accept bits of levels 1 .. j-1
// (level 1 most significant).  Outcome o of a launch = the accept bits of its levels (level 1 most significant).
// Everything ping-pongs on the launch's parity `set`:
//   state[set]   written by K(m): x (state after the decision K(m) took), its log-posterior, and -- for the tree K(m)
//                evaluates -- every node's proposal, its log-prior, every level's log u
//   partial[set] written by K(m): per node, the per-wave partial sums of its star likelihoods
//   cand[set]    read by K(m): per outcome of K(m-1), the tree's (params, headers, isochrones); K(m) writes cand[set ^ 1]
// ------------------------------------------------------------------------------------------
#define B9_TREE_MAX_DEPTH 3
#define B9_TREE_MAX_NODES 7
// partials per lane and node the walk reads in its one round trip (B9_TREE_KD of the tree kernels' two builds, b9_mcmc_tree.hip.h):
// a node has at most 64 KD hot partials = 16 KD canonical tile groups
#define B9_TREE_KD_SMALL 3
#define B9_TREE_KD_LARGE 5
#define B9_TREE_MAX_GROUPS (16 * B9_TREE_KD_LARGE)
#define B9_TS_CUR 0              // [0..11]  state after the decision this launch took
#define B9_TS_LP 12              // its log-posterior
#define B9_TS_NACC 13            // proposals accepted so far in the block
#define B9_TS_LOGU 14            // [14..16] log u of levels 1..3 of the tree this launch evaluates
#define B9_TS_LPRIOR 17          // [17..23] log-prior of node n's proposal (-inf: outside the grid or the prior's support)
#define B9_TS_PROP 24            // [24 + 12 n ..] node n's proposal
#define B9_TREE_STATE_STRIDE (24 + 12 * B9_TREE_MAX_NODES + 4)      // 112 doubles
#define B9_TREE_TAB_ROW 12       // step table row: [0..10] delta of sampled parameter t, [11] log u

struct TreeDev {
    int d, n_walkers, n_pops;
    int depth;                       // levels of a full tree (1..3)
    int levels_prev;                 // levels of the tree the PREVIOUS launch evaluated (0: none -- first launch of a block, prologue)
    int levels;                      // levels this launch evaluates (< depth only for a block's last launch)
    int derive_mode;                 // 0: derive nothing (finish), 1: next trees for every outcome of this launch's tree,
                                     // 2: prologue -- the block's first tree from the starting state (outcome slot 0 only)
    int set;                         // parity of this launch
    int row;                         // chain row of the first step the decision of this launch appends
    int n_groups, part_stride;       // tile groups per node; doubles per (walker, node) partial row = 4 n_groups + heavy_parts (padded)
    int mass_cap, heavy_parts;
    unsigned k0, k1;                 // Philox key
    unsigned long long step;         // global index of the step at level 1 of the tree this launch evaluates
    unsigned long long next_step;    // ... of the tree(s) this launch derives
    long long iso_stride;
    double *state;                   // [2][W][B9_TREE_STATE_STRIDE]
    double *partial;                 // [2][W][nodes][part_stride]
    double *cand_par;                // [2][W][outcomes][nodes][12]
    IsoHdr *cand_hdr;                // [2][W][outcomes][nodes][pops]
    double *cand_iso;                // [2][W][outcomes][nodes][pops][iso_stride]
    const double *chol;              // [d][d]
    const int *free_idx;             // [d]
    const int *walker_ids;           // [W]
    double *samples;                 // [n_steps][W][d] or null
    double *lps;                     // [n_steps][W] or null
    // Per-block table of everything about a STEP that does not depend on the chain: [W][tab_steps][B9_TREE_TAB_ROW] --
    // per walker and step s (0 = the block's first): delta[t] = sum_j chol[t][j] z_j of sampled parameter t (the proposal's
    // increment) and the accept test's log u.  Written once per block by the prologue launch; the K launches' derivation
    // workgroups and writers READ it (the Philox + Box-Muller of six steps on one wave was 1.6 us of every derivation's chain).
    double *step_tab;
    int tab_steps;                   // n_steps + B9_TREE_MAX_DEPTH (a tree may reach past the block's end; those rows are never used for a decision)
    unsigned long long block_step0;  // global index of the block's first step
    // the finish launch only (as StepDev's)
    double *rows;
    const double *row_origin;
    int n_steps;
    double *host_state;              // [W][B9_TREE_STATE_STRIDE]
    double *host_rows;
};
