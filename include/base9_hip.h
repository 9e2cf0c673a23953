/*
 * base9_hip.h -- C ABI of the MI355X-native BASE-9 per-step log-posterior path.
 *
 * This is the drop-in boundary of the hot path.  In the reference
 * the boundary is an in-process C++ call -- the MCMC driver calls "logPost(proposed cluster)"
 * once per step, which derives one isochrone from the loaded model pack and loops over the
 * cluster's stars.  That source is NOT mounted (/root/reference/README.md:4 redirects to
 * BayesianStellarEvolution/base-cpp), so no entry point below can cite a reference file:line;
 * each cites instead the SURVEY.md section-8a row it implements and, tagged [RECALL], the
 * upstream function it is believed to replace.  Parity with BASE-9 is therefore UNPINNED.
 *
 * Conventions
 *  - plain C structs, plain pointers and sizes; no C++/torch types cross this boundary;
 *  - the caller owns every host buffer passed in; the context owns all device memory;
 *  - every function returns B9_OK (0) or a negative b9_status; b9_last_error() gives text;
 *  - proposed parameters outside the model grid are NOT an error: that walker's
 *    log-posterior is -INFINITY (the reference rejects such a step [RECALL]);
 *  - one context per GPU, not thread-safe, all work stream-ordered on the context's stream;
 *  - there is NO CPU fallback: without a HIP device b9_ctx_create fails with B9_ERR_NO_DEVICE.
 */
#ifndef BASE9_HIP_H
#define BASE9_HIP_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define B9_ABI_VERSION 6

/* ---- status codes ------------------------------------------------------------------ */
typedef enum b9_status {
    B9_OK = 0,
    B9_ERR_NO_DEVICE = -1,   /* no HIP device / HIP runtime failure at create            */
    B9_ERR_INVALID   = -2,   /* bad argument (NULL, negative size, unsorted axis ...)    */
    B9_ERR_STATE     = -3,   /* call order: pack/stars not loaded yet                    */
    B9_ERR_HIP       = -4,   /* a HIP call failed; see b9_last_error                     */
    B9_ERR_CAPACITY  = -5    /* table too large for the kernel's LDS plan                */
} b9_status;

/* ---- cluster parameter row ([RECALL] Cluster::getParam order) ------ */
enum {
    B9_P_LOGAGE = 0,      /* log10(age/yr)                                              */
    B9_P_Y = 1,           /* helium mass fraction (population A in a two-pop run)        */
    B9_P_FEH = 2,         /* [Fe/H]                                                      */
    B9_P_MOD = 3,         /* distance modulus (m-M)_V, includes A_V                      */
    B9_P_ABS = 4,         /* absorption A_V                                              */
    B9_P_CARBONICITY = 5, /* WD core carbon fraction                                     */
    B9_P_IFMR_INTERCEPT = 6,
    B9_P_IFMR_SLOPE = 7,
    B9_P_IFMR_QUAD = 8,
    B9_P_Y2 = 9,          /* helium of population B (two-pop only; [RECALL] multiPopMcmc)*/
    B9_P_LAMBDA = 10,     /* fraction of stars in population A (two-pop only)            */
    B9_P_RESERVED = 11,
    B9_NPARAM = 12        /* row stride, in doubles                                      */
};

/* ---- star status codes ([RECALL] .phot "stage" column) -------------------------------- */
enum { B9_STAGE_MSRG = 1, B9_STAGE_WD = 3, B9_STAGE_NSBH = 4, B9_STAGE_BD = 5, B9_STAGE_DNE = 9 };

/* ---- IFMR ids ----------------------------------------------------- */
enum {
    B9_IFMR_WEIDEMANN = 0, B9_IFMR_WILLIAMS = 1, B9_IFMR_SALARIS_LIN = 2,
    B9_IFMR_SALARIS_PW = 3, B9_IFMR_LINEAR = 4, B9_IFMR_QUADRATIC = 5
};

/* Magnitude assigned to "contributes no flux" ([RECALL] 99.999 sentinel). */
#define B9_MAG_NOFLUX 99.999

/*
 * Model pack: the tables a loaded MS/RGB model + WD cooling model + WD atmosphere model +
 * filter set expose to the hot path ([RECALL] MsRgbModel /
 * WdCoolingModel / WdAtmosphereModel / Model aggregate).  All axes strictly ascending.
 *
 * Isochrone (ifeh, iy, iage) has index  iso = (ifeh * n_y + iy) * n_age + iage,  holds
 * iso_n_eep[iso] evolutionary points whose EEP ids are iso_first_eep[iso] + 0,1,2,...  and
 * whose data start at point offset iso_offset[iso] in mass[] / mags[] (mags point-major:
 * mags[(off + k) * n_filt + f]).  Masses ascend along an isochrone.
 */
typedef struct b9_pack {
    int32_t n_filt;
    /* MS/RGB grid */
    int32_t n_feh, n_y, n_age;
    const double *feh;            /* [n_feh]                                             */
    const double *y;              /* [n_y]   (n_y == 1: helium is not a grid axis)        */
    const double *log_age;        /* [n_age]                                             */
    const int32_t *iso_first_eep; /* [n_feh*n_y*n_age]                                   */
    const int32_t *iso_n_eep;     /* [n_feh*n_y*n_age]                                   */
    const int64_t *iso_offset;    /* [n_feh*n_y*n_age]                                   */
    int64_t n_points;             /* total evolutionary points                            */
    const double *mass;           /* [n_points]                                          */
    const double *mags;           /* [n_points*n_filt]                                   */
    const double *abs_coeff;      /* [n_filt]  A_f / A_V                                  */
    /* WD cooling model (ABI 2): one cooling track per (carbonicity, WD mass) node, EACH WITH ITS OWN cooling-age axis --
     * real cooling tracks are ragged: every mass was evolved over its own sequence of ages.  Track (ic, im) has index
     * t = ic * n_wc_mass + im and holds wc_n_age[t] >= 2 points (log cooling age strictly ascending) starting at point
     * offset wc_offset[t] of wc_log_age / wc_log_teff / wc_log_radius.  A lookup brackets the cooling age in each of
     * the (2 carbonicities x) 2 masses' own axes, interpolates along each track, then across mass, then across
     * carbonicity.  n_wc_mass == 0: no WD models.  n_wc_carb <= 1: carbonicity is not an axis. */
    int32_t n_wc_carb, n_wc_mass;
    const double *wc_carb;        /* [n_wc_carb]                                         */
    const double *wc_mass;        /* [n_wc_mass] WD mass, Msun                            */
    const int32_t *wc_n_age;      /* [max(1,n_wc_carb)*n_wc_mass] points of each track      */
    const int64_t *wc_offset;     /* [max(1,n_wc_carb)*n_wc_mass] first point of each track */
    int64_t n_wc_points;          /* total cooling-track points                           */
    const double *wc_log_age;     /* [n_wc_points] log10 cooling age / yr                 */
    const double *wc_log_teff;    /* [n_wc_points]                                       */
    const double *wc_log_radius;  /* [n_wc_points] log10 R / cm                           */
    /* WD atmospheres: [n_at_type][n_at_logg][n_at_teff][n_filt], type 0 = DA, 1 = DB      */
    int32_t n_at_type, n_at_logg, n_at_teff;
    const double *at_logg;        /* [n_at_logg]                                         */
    const double *at_log_teff;    /* [n_at_teff]                                         */
    const double *at_mags;        /* [n_at_type*n_at_logg*n_at_teff*n_filt]              */
    /* IFMR + mass limits */
    int32_t ifmr_id;
    int32_t reserved0;
    double m_wd_up;               /* upper ZAMS mass that still makes a WD ([RECALL] M_wd_up) */
} b9_pack;

/*
 * Stars: what the photometry file gives per stellar system ([RECALL]
 * StellarSystem).  obs/sigma are star-major as read from the file: obs[i * n_filt + f].
 * sigma < 0 (or == 0) marks a filter as unused for that star.
 */
typedef struct b9_stars {
    int32_t n_stars, n_filt;
    const double *obs;            /* [n_stars*n_filt] observed magnitudes                 */
    const double *sigma;          /* [n_stars*n_filt] 1-sigma errors, <=0 -> unused       */
    const double *mass1;          /* [n_stars] primary ZAMS mass, Msun                    */
    const double *mass_ratio;     /* [n_stars] secondary/primary, 0 -> single             */
    const double *clust_prior;    /* [n_stars] prior cluster-membership probability       */
    const int32_t *stage;         /* [n_stars] B9_STAGE_* (used by the marginalised mode) */
    const int32_t *wd_type;       /* [n_stars] 0 DA, 1 DB; may be NULL (all DA)           */
    const double *filter_prior_min; /* [n_filt] field-star magnitude range ...           */
    const double *filter_prior_max; /* [n_filt] ... fsLike = prod_f 1/(max-min)           */
} b9_stars;

/* Cluster-level priors (SURVEY section 2 "Priors"; scalar per walker, added on device).   */
typedef struct b9_priors {
    double mean[B9_NPARAM];       /* Gaussian prior means                                 */
    double var[B9_NPARAM];        /* variances; <= 0 -> flat (no term)                    */
    double log_age_min, log_age_max; /* flat prior support in logAge; outside -> -inf      */
} b9_priors;

/* Evaluation modes */
enum {
    B9_MODE_GIVEN_MASS   = 0,  /* BASELINE.json north_star: one interpolation per star at its (mass1, mass_ratio) */
    B9_MODE_MARGINALISED = 1   /* [RECALL] marg.cpp: integrate each star over primary mass and mass ratio        */
};

typedef struct b9_options {
    int32_t mode;             /* B9_MODE_*                                                */
    int32_t n_pops;           /* 1 (singlePopMcmc) or 2 (multiPopMcmc)                    */
    int32_t marg_iso_increm;  /* sub-steps per EEP interval in the primary-mass integral  */
    int32_t marg_n_q;         /* mass-ratio quadrature nodes                              */
} b9_options;

/*
 * Launch-plan tuning (ABI 4).  Every field 0 = automatic (what a context starts with); results are the same for every
 * setting to the stated tolerance -- these only regroup the same work (tests/test_gpu_mcmc.py runs the combinations).
 * The one field that touches the ROUNDING of a walker's log-posterior is tiles_per_block: the catalogue's tiles are dealt
 * into canonical groups of that many tiles and every wave forms one partial per group.  Its automatic value is a function
 * of the catalogue, the pack, the options and the device ONLY -- not of the number of walkers on the GPU -- so a walker's
 * chain is the same bits whatever the number of GPUs the walkers are spread over (since ABI 4; before, the automatic value
 * followed the local walker count and a driver had to pin it).
 * The same fields can be set through the environment, read ONCE when the context is created: B9_TILES_PER_BLOCK,
 * B9_DERIVE_PARTS, B9_DERIVE_ORDER (historical coding: 1 default, 0 heavy first, < 0 derivation trails), B9_HEAVY_PARTS,
 * B9_TWO_LAUNCH_STEPS, B9_MARG_NO_PRUNING, B9_TIMING_GROUP, B9_PLAN_DEBUG, B9_TREE_DEPTH;
 * B9_STREAM_PRIORITY=default gives the context's stream the default priority instead of the lowest.
 */
typedef struct b9_tuning {
    int32_t tiles_per_block;   /* star tiles (256 stars) per CANONICAL tile group of k_star_like / k_mcmc_step / k_mcmc_tree  */
    int32_t derive_parts;      /* fused step: workgroups per candidate isochrone                                         */
    int32_t derive_order;      /* fused step grid: 1 writers + derivation lead (default), 2 heavy-star workgroups lead,  */
                               /* 3 derivation workgroups trail the hot ones                                            */
    int32_t heavy_parts;       /* workgroups per walker for the stars above the AGB tip (default: sized from the catalogue) */
    int32_t two_launch_steps;  /* 1: the derive + star launch pair per sampler step also in given-mass mode                */
    int32_t marg_no_pruning;   /* 1: marginalised kernel evaluates every node of every star (no floor, no boxes): the      */
                               /* brute-force statement of the same sum on the GPU, for tests                            */
    int32_t timing_group;      /* launches per HIP-event bracket of b9_enable_timing in the fused step (default 8)        */
    int32_t plan_debug;        /* 1: print the fused step's launch plan to stderr whenever it changes; 2: also the marginalised catalogue's pieces */
    int32_t tree_depth;        /* given-mass sampler blocks: Metropolis steps per launch.  1 = the one-step fused launch;   */
                               /* 2 / 3 = the tree-speculative launch (every proposal of the chain's next 2 / 3 steps --   */
                               /* 3 / 7 of them -- evaluated at once, same chain); default: the deepest tree whose          */
                               /* workgroups are all resident at once AND whose estimated cost per step beats the one-step  */
                               /* launch's (few walkers per GPU, catalogues that leave the chip under-filled), else 1.      */
                               /* Env: B9_TREE_DEPTH                                                                        */
    int32_t marg_piece_units;  /* marginalised mode, small catalogues: the smallest share of a star chunk's node window worth a   */
                               /* workgroup of its own, in (16 nodes x one mass ratio) units per wave (default 4; 6 from 8 mass ratios on).  Part of what a */
                               /* star's sum rounds like: every rank of a run must use the same value.  Env: B9_MARG_PIECE_UNITS   */
    int32_t reserved[6];
} b9_tuning;

typedef struct b9_ctx b9_ctx;

/* ---- lifecycle ---------------------------------------------------------------------- */
int         b9_abi_version(void);
/* device_id < 0 -> current device.  Fails with B9_ERR_NO_DEVICE when no GPU is present.  */
int         b9_ctx_create(int device_id, b9_ctx **out);
void        b9_ctx_destroy(b9_ctx *ctx);
const char *b9_last_error(const b9_ctx *ctx);   /* ctx may be NULL: last create error      */

/* ---- staging (cold; once per run) ---------------------------------------------------- */
/* Replaces [RECALL] Model construction: copies the pack's tables to HBM.                   */
int b9_load_pack(b9_ctx *ctx, const b9_pack *pack);
/* Replaces [RECALL] reading the .phot into vector<StellarSystem>: SoA + staged to HBM.     */
int b9_load_stars(b9_ctx *ctx, const b9_stars *stars);
int b9_set_priors(b9_ctx *ctx, const b9_priors *priors);
int b9_set_options(b9_ctx *ctx, const b9_options *opt);
/* tuning NULL = everything automatic again.  Not while a block is outstanding. */
int b9_set_tuning(b9_ctx *ctx, const b9_tuning *tuning);
/* The tuning in force: the B9_* environment overrides read at creation, then whatever the last b9_set_tuning passed -- a
 * caller that wants to change ONE field reads, modifies and writes back (ABI 4). */
int b9_get_tuning(const b9_ctx *ctx, b9_tuning *out);

/* ---- the hot path -------------------------------------------------------------------- */
/*
 * Log-posterior of n_walkers proposed parameter rows ([RECALL]
 * MpiMcmcApplication::logPostStep).  params: host, [n_walkers * B9_NPARAM].
 * out_logpost: host, [n_walkers].  out_perstar: host, [n_walkers * n_stars] per-star
 * log( (1-p) fsLike + p L_i ) in the ORIGINAL star order, or NULL.
 * Synchronous (returns after the results are on the host).
 */
int b9_logpost(b9_ctx *ctx, const double *params, int32_t n_walkers,
               double *out_logpost, double *out_perstar);

/*
 * Same, device-resident and asynchronous: d_params / d_logpost (/ d_perstar, nullable) are
 * DEVICE pointers; launches on `stream` (a hipStream_t passed as void*, NULL = the context's
 * stream) and returns without synchronising.  This is the entry the walker-parallel driver
 * uses so that the RCCL all-gather reads log-posteriors straight from HBM.
 */
int b9_logpost_device(b9_ctx *ctx, const double *d_params, int32_t n_walkers,
                      double *d_logpost, double *d_perstar, void *stream);

/*
 * Device-resident Metropolis block: advances the n_walkers local chains n_steps steps without
 * returning to the host between steps (SURVEY 8f row 1 -- the per-step caller of the hot path;
 * [RECALL] the body of MpiMcmcApplication's sampling loop: propose from the adapted covariance,
 * logPost, accept/reject).  Per step and walker w: z ~ N(0, I_d) and u ~ U(0,1) from the
 * counter-based stream Philox4x32-10(key = seed, counter = (step, walker_ids[w], draw));
 * proposal = current, proposal[free_idx[i]] += sum_j chol[i*d+j] z_j; accepted when
 * log u < logpost(proposal) - logpost(current).  All pointers are HOST pointers; params and
 * logpost are updated in place; samples ([n_steps][n_walkers][n_free]) and lps
 * ([n_steps][n_walkers]) receive the chain and may be NULL.  Synchronous.
 */
typedef struct b9_mcmc_block {
    int32_t n_walkers, n_free;
    const int32_t *free_idx;     /* [n_free] parameter indices being sampled                  */
    const double *chol;          /* [n_free*n_free] row-major proposal factor                 */
    const int32_t *walker_ids;   /* [n_walkers] global walker ids (random-number streams)     */
    uint64_t seed;
    int64_t step0;               /* global number of the block's first step                   */
    int32_t n_steps, flags;      /* flags: B9_BLOCK_* (0 = start from params/logpost, synchronous)          */
    double *params;              /* [n_walkers*B9_NPARAM] in/out                              */
    double *logpost;             /* [n_walkers] in/out                                        */
    double *samples;             /* out, nullable                                             */
    double *lps;                 /* out, nullable                                             */
    int64_t n_accept;            /* out                                                       */
    /* ---- block summary rows (ABI 2; both evaluation modes since ABI 3).  row_origin != NULL: the block's last launch also condenses
     * every walker's chain into ONE row of B9_ROW_DOUBLES(n_free) doubles on the device,
     *     [0] log-posterior after the block   [1..12] position after the block
     *     [13] steps after the block's first on which the walker moved   [14] n_steps
     *     [15 .. 15+d) sum_s x_s    [15+d .. 15+d+d*d) sum_s x_s x_s^T (row-major),   x_s = sample_s - row_origin,
     * sums over the steps in ascending order, plain multiply and add.  These rows are what a walker-parallel driver
     * exchanges between GPUs for the adaptive proposal: d_rows is their DEVICE address (valid until the second-next
     * block of this context is enqueued) and rows_ready (flag B9_BLOCK_ROWS_EVENT) a hipEvent_t recorded on the context's
     * stream once they are written, so a collective on another stream can read them from HBM without a host round trip.  `rows` (host,
     * nullable) receives a copy when the block is collected. */
    const double *row_origin;    /* [n_free] or NULL                                          */
    double *rows;                /* out, nullable: [n_walkers][B9_ROW_DOUBLES(n_free)]         */
    void *d_rows;                /* out: device pointer to the same rows                       */
    void *rows_ready;            /* out: hipEvent_t                                            */
} b9_mcmc_block;
#define B9_ROW_DOUBLES(d) (15 + (d) + (d) * (d))
int b9_mcmc_run_block(b9_ctx *ctx, b9_mcmc_block *blk);

/*
 * Pipelining blocks (both evaluation modes; a block may only continue a block of the same mode).  B9_BLOCK_ASYNC: b9_mcmc_run_block returns as soon as the block is
 * enqueued on the context's stream; b9_mcmc_wait(ctx, blk) -- same blk, whose host arrays must stay valid --
 * blocks until it has run and fills params / logpost / samples / lps / n_accept.  B9_BLOCK_CONTINUE: the block
 * starts from the state the PREVIOUS block of this context left on the device (its params / logpost inputs are
 * ignored; n_walkers must match), so it can be enqueued before that block has finished.  At most two blocks
 * may be outstanding; they are collected in the order they were enqueued.  While a block is outstanding it owns the
 * context's work buffers: b9_logpost / b9_sample_mass / b9_derive_isochrone, the staging calls, b9_set_options and a block
 * with another n_walkers return B9_ERR_STATE until it has been collected.  A driver that adapts the proposal
 * from block b-1 while block b runs keeps the GPU's queue non-empty: enqueue b+1 (CONTINUE | ASYNC), wait(b), ...
 */
#define B9_BLOCK_CONTINUE 1
#define B9_BLOCK_ASYNC 2
#define B9_BLOCK_ROWS_EVENT 4   /* with row_origin: also record rows_ready (a multi-GPU exchange waits on it); off, one event less in the stream */
int b9_mcmc_wait(b9_ctx *ctx, b9_mcmc_block *blk);

/*
 * Per-star mass posterior draws (the sampleMass counterpart; [RECALL] sampleMass
 * re-reads the cluster chain and, for every saved row and every star, draws the primary mass and the
 * mass ratio from the star's conditional posterior on a grid, and reports the membership
 * probability).  The grid is the one of the marginalised mode (b9_options.marg_iso_increm sub-steps
 * per EEP interval x marg_n_q mass ratios; WD-stage stars: 8 * marg_iso_increm steps above the AGB
 * tip), whatever b9_options.mode says.  For row r (a full parameter row, as in b9_logpost) and star
 * i ONE node is drawn by the Gumbel-max rule: argmax over nodes of
 *     log( prior(M1) dM / n_q * like_i(M1, q) [* lambda_k] )  -  log(-log u),
 *     u = Philox4x32-10(key = (seed lo, seed hi ^ row hi), counter = (row lo, i, node lo, 2 * node hi + k)),
 * node = (EEP interval * marg_iso_increm + sub-step) * n_q + j for main-sequence/giant nodes and the
 * step number for WD nodes, k = population; row = row0 + r.  An argmax does not depend on the order
 * in which nodes are visited, so any implementation picks the same node.
 * Outputs, host, [n_rows][n_stars] in the caller's star order: out_mass (primary mass of the node),
 * out_ratio (j / n_q; 0 for WD nodes), out_member = p_i L_i / (p_i L_i + (1 - p_i) fieldLike),
 * out_pop (0 / 1; may be NULL).  A row outside the grid gives mass = ratio = member = 0.
 */
int b9_sample_mass(b9_ctx *ctx, const double *params, int32_t n_rows, uint64_t seed, int64_t row0,
                   double *out_mass, double *out_ratio, double *out_member, int32_t *out_pop);

/*
 * Derive the isochrone for one parameter row ([RECALL]
 * MsRgbModel::deriveIsochrone; this is all that makeCMD needs).  Outputs, host:
 * out_mass[cap], out_mags[cap*n_filt] (absolute magnitudes, no modulus/absorption),
 * *out_first_eep, *out_n (0 when the row is outside the grid).  pop = 0 uses B9_P_Y,
 * pop = 1 uses B9_P_Y2.
 */
int b9_derive_isochrone(b9_ctx *ctx, const double *param_row, int32_t pop, int32_t cap,
                        double *out_mass, double *out_mags,
                        int32_t *out_first_eep, int32_t *out_n, double *out_agb_tip);

/* ---- introspection (used by bench/tests; no compute) --------------------------------- */
int b9_max_eep(const b9_ctx *ctx);           /* longest isochrone in the loaded pack        */
int b9_device_id(const b9_ctx *ctx);
/* Algorithmic bytes one star-eval moves in the loaded layout ("bytes per unit"). */
int b9_bytes_per_star_eval(const b9_ctx *ctx);
/* Star tiles per hot workgroup the fused sampler step would use for n_walkers local walkers with the loaded pack, stars
 * and tuning (> 0), or a negative b9_status.  The summation grouping of a walker's log-posterior follows from it (see
 * b9_tuning.tiles_per_block). */
int b9_step_tiles_per_block(b9_ctx *ctx, int32_t n_walkers);
/* Metropolis steps one launch of a given-mass sampler block advances n_walkers local chains by (b9_tuning.tree_depth). */
int b9_step_depth(b9_ctx *ctx, int32_t n_walkers);
/* Elapsed ms of the dominant (star-likelihood) kernel over the TIMED launches since the last
 * call with reset != 0, measured with HIP events on the launch stream; *n_launches receives
 * their count.  b9_enable_timing(ctx, n): n = 0 off, n > 0 opens an event bracket at every n-th
 * launch.  In the sampler's fused step a bracket spans 8 consecutive launches of the kernel (never
 * past the block's end; B9_TIMING_GROUP), so that the two event records cost an eighth of what a
 * bracket around a single launch adds (~2-3 us): total_ms / n_launches is the kernel's launch period. */
int b9_enable_timing(b9_ctx *ctx, int on);
int b9_kernel_time_ms(b9_ctx *ctx, int reset, double *total_ms, int32_t *n_launches);
/* Mean elapsed ms of the same event bracket around an EMPTY kernel: what the bracket adds to a
 * kernel's own duration (one dispatch boundary + event processing).  Reported, never applied. */
int b9_calibrate_timing(b9_ctx *ctx, double *bracket_overhead_ms);
/* Shader clock observed over a stretch of the context's stream (ABI 5).  b9_clock_stamp(ctx, 0) / (ctx, 1) enqueue a small
 * kernel that records, per compute unit, the shader-cycle counter (s_memtime) and the 100 MHz reference counter
 * (s_memrealtime); b9_clock_mhz waits for the stream and returns delta(shader cycles) / delta(reference) x 100 MHz between
 * the two stamps: the median over the compute units both stamps reached (differences are formed per CU: the cycle counters
 * of different CUs are offset against each other) and, when the pointers are not NULL, the extremes and the length of the
 * stretch in seconds of the reference clock.  What a roofline's clock-dependent peak should be read against. */
int b9_clock_stamp(b9_ctx *ctx, int32_t which);
int b9_clock_mhz(b9_ctx *ctx, double *mhz, double *mhz_min, double *mhz_max, double *ref_seconds);

#ifdef __cplusplus
}
#endif
#endif /* BASE9_HIP_H */
