// b9_marg_step.hip.h -- k_marg_step: the marginalised mode's fused sampler step (decision + stars + next step's candidate tables).
// Part of the single translation unit b9_kernels.hip (included there, after b9_mcmc_step.hip.h and b9_star_marg.hip.h); gfx950 only.
#pragma once

// ------------------------------------------------------------------------------------------
// Marginalised mode, ONE launch per MCMC step (round 5; rounds 3-4 ran k_derive_iso, k_marg_table, k_star_marg [,
// k_marg_merge, k_marg_wd_table, k_star_marg_wd] back to back: on a single chain the three small dependent launches were
// 40 % of the step).  The scheme is the given-mass fused step's (StepDev, b9_mcmc_step.hip.h) with "candidate isochrone"
// replaced by "candidate node table":
//
//   K(t) decides step t-1 (every workgroup's first wave, redundantly, from step t-1's partial sums: identical bits),
//        evaluates the stars against the node table of step t's proposal -- ONE of the two candidate tables K(t-1) built --
//        and builds BOTH candidate tables of step t+1 (base = the state if step t is rejected / step t's proposal if it is
//        accepted; step t+1's normals are counter-based and depend on neither).
//
// A table builder does not read a derived isochrone (that would be a derive -> table dependency between workgroups of one
// launch): it derives what it needs itself -- the mass column into LDS, and the two magnitude rows of a primary or
// companion node by interpolating the pack's 4 (8 with a helium axis) corner isochrones on the spot, in interp_corner's
// operation order, so the table holds the bits k_derive_iso + k_marg_table produce.  No isochrone is materialised at all.
//
// Roles by workgroup id: [one writer per walker][table builders: walker x candidate x population x 64-node chunk]
// [WD-table builders (catalogues with WD-stage stars)][pad to 8][star workgroups: star_marg_body][WD-stage stars].
// Small catalogues split a star chunk's window over several workgroups (b9k_marg_split); their per-star shares are merged
// by k_marg_step_merge, a second small launch (a last-arriver merge inside the launch was measured: the agent-scope
// fences serialise at ~70 ns per workgroup, write-through shares cost 4-6 us more than the launch;
// tools/probes/last_probe.hip).
// ------------------------------------------------------------------------------------------
struct MargStep {
    int K, Q;
    int n_chunks_cap;         // node chunks of the longest isochrone = table builders per (walker, candidate, population)
    int n_wd_blocks;          // WD-table builders per (walker, candidate, population): ceil(8 K / 128); 0 = no WD-stage stars
    int wsplit;
    double cut2;
    MargLayout L;
    double *tab;              // [2 parities][2 candidates][W * pops][L.total]
    double *wd_tab;           // [2][2][wd_stride]: one candidate's block has k_marg_wd_table's layout for W * pops rows
    long long wd_stride;
    double *shares;           // split launches: per-star shares (k_star_marg's layout)
};

// value of the isochrone with header h (cell corners c) at common EEP e, magnitude column col (MASS: the mass column):
// interp_corner's arithmetic on corner_rows' registers -- age, then Y, then FeH
template <bool MASS>
__device__ __forceinline__ double corner_interp(const DevPack &pk, const CornerRegs &c, const IsoHdr &h, int ny, int e, int col)
{
    double vf[2];
#pragma unroll
    for (int df = 0; df < 2; ++df) {
        double vy[2] = {0.0, 0.0};
#pragma unroll
        for (int dy = 0; dy < 2; ++dy) {
            if (dy < ny) {
                const long long p0 = c.off[df * 4 + dy * 2] + e, p1 = c.off[df * 4 + dy * 2 + 1] + e;
                const double a = MASS ? pk.mass[p0] : pk.mags[p0 * pk.nfp + col];
                const double b = MASS ? pk.mass[p1] : pk.mags[p1 * pk.nfp + col];
                vy[dy] = lerp(a, b, h.t_age);
            }
        }
        vf[df] = (ny == 2) ? lerp(vy[0], vy[1], h.t_y) : vy[0];
    }
    return lerp(vf[0], vf[1], h.t_feh);
}

// The header of candidate `cand`'s isochrone of population `pop` from its parameter row in LDS: every wave brackets the
// grid cell and reads the corner rows itself (registers; no barrier).  agb_tip is left 0.
__device__ __forceinline__ IsoHdr marg_header(const DevPack &pk, const double *s_par, int pop, int mass_cap, CornerRegs &cr)
{
    AxisRegs ax[3];
    preload_axes3(pk, ax);
    const double log_age = s_par[B9_P_LOGAGE], feh = s_par[B9_P_FEH], y = pop ? s_par[B9_P_Y2] : s_par[B9_P_Y];
    const GridCell cell = grid_cell(pk, ax, log_age, feh, y);
    cr = corner_rows(pk, cell);
    return header_of(pk, cell, cr, log_age, feh, y, mass_cap);
}

// Table builder: chunk c of the node table of (walker w, candidate cand, population pop) of step t+1 -- k_marg_table's
// rows, boxes and nb, bit for bit.  Chunk 0 also publishes the candidate's header (with the AGB-tip mass) and, for
// population 0, its parameter row.
//
// What of the candidate's isochrone a chunk needs is derived into LDS TILES, each value once: the mass column (every
// common EEP: a companion may lie anywhere below its primary), the primaries' rows (the chunk's 64 / K + 1 EEPs), and per
// wave (= mass ratio) the rows its 64 companions bracket -- neighbours in mass, hence a short run of EEPs: up to
// B9_MSTEP_SEC_ROWS rows derived by the wave itself; a longer run (none on the synthetic packs) is interpolated per lane.
// (A first version interpolated every row per (node, mass ratio, filter) thread on the spot: 4096 row-value derivations per
// chunk against 3600 for the whole isochrone, a 27 us chain; deriving the whole isochrone into LDS instead needs 29 KB per
// workgroup -- of EVERY workgroup of the launch: the star role would lose 3 of its 7 workgroups per CU.)
#define B9_MSTEP_SEC_ROWS 24
// doubles of dynamic LDS: [mass column + 8][primary rows: 65 x NFP][4 waves x SEC_ROWS x NFP]
#define B9_MSTEP_LDS_DOUBLES(NFP, mass_cap) ((size_t)(mass_cap) + 8 + (size_t)65 * (NFP) + (size_t)4 * B9_MSTEP_SEC_ROWS * (NFP))

// minimum / maximum towards lane 0 of every row of 16 lanes (DPP), then towards lane 0 of the wave
__device__ __forceinline__ void row_min_max(double &lo, double &hi)
{
    lo = __builtin_fmin(lo, lane_down<8>(lo)); hi = __builtin_fmax(hi, lane_down<8>(hi));
    lo = __builtin_fmin(lo, lane_down<4>(lo)); hi = __builtin_fmax(hi, lane_down<4>(hi));
    lo = __builtin_fmin(lo, lane_down<2>(lo)); hi = __builtin_fmax(hi, lane_down<2>(hi));
    lo = __builtin_fmin(lo, lane_down<1>(lo)); hi = __builtin_fmax(hi, lane_down<1>(hi));
}
__device__ __forceinline__ void rows_min_max(double &lo, double &hi)
{
    lo = __builtin_fmin(lo, lane_down<16>(lo)); hi = __builtin_fmax(hi, lane_down<16>(hi));
    lo = __builtin_fmin(lo, lane_down<32>(lo)); hi = __builtin_fmax(hi, lane_down<32>(hi));
}

template <int NFP, int NPOPS>
__device__ __forceinline__ void marg_build_table(const DevPack &pk, const StepDev &sd, const MargStep &mx, int w, int cand, int pop, int c, double *smem)
{
    constexpr bool box32 = B9_BOX32(NFP, NPOPS);
    if (!sd.derive_next || MSTEP_NO_BUILD) return;
    const int tid = threadIdx.x, lane = tid & 63, jl = tid >> 6, W = sd.n_walkers, n_pops = sd.n_pops, K = mx.K, Q = mx.Q;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), mass_cap = sd.mass_cap;
    __shared__ double s_par[B9_NPARAM], s_z[12];
    __shared__ double s_box[4][2][NFP], s_shift[NFP];
    double *const s_mass = smem, *const s_prim = smem + mass_cap + 8, *const s_sec = s_prim + 65 * NFP + (size_t)wave * B9_MSTEP_SEC_ROWS * NFP;
    HSTAMP(0);
    if (wave == 0) candidate_row_wave0<true>(sd, w, cand, s_par, s_z);
    __syncthreads();
    HSTAMP(1);
    const size_t rows = (size_t)W * n_pops, cset = (size_t)((sd.set ^ 1) * 2 + cand);
    const int wp = w * n_pops + pop;
    if (c == 0 && pop == 0 && tid < B9_NPARAM) sd.cand_par[(cset * W + w) * B9_NPARAM + tid] = s_par[tid];
    CornerRegs cr;
    IsoHdr h = marg_header(pk, s_par, pop, mass_cap, cr);
    HSTAMP(2);
    IsoHdr *hp = sd.cand_hdr + cset * rows + wp;
    if (!h.valid) { if (c == 0 && tid == 0) *hp = h; return; }
    const int n = h.n, n_nodes = (n - 1) * K, ny = pk.n_y > 1 ? 2 : 1;
    if (c * 64 >= n_nodes) return;                                      // (the star kernel stops at the isochrone's last chunk; chunk 0 always exists)
    // tiles: the mass column and the primaries' rows e0 .. e0 + n_prim - 1 (one value per thread and round)
    const int e0 = (c * 64) / K;
    const int e_last = (c * 64 + 63) / K + 1;
    const int n_prim = (e_last < n - 1 ? e_last : n - 1) - e0 + 1;     // <= 65
    for (int e = tid; e < n; e += 256) s_mass[e] = corner_interp<true>(pk, cr, h, ny, e, 0);
    for (int i = tid; i < n_prim * NFP; i += 256) {
        const int r = i / NFP, f = i - r * NFP;
        s_prim[i] = f < pk.nf ? corner_interp<false>(pk, cr, h, ny, e0 + r, f) : 0.0;
    }
    if (tid < 8) s_mass[n + tid] = __builtin_inf();                     // find_bracket's masked over-read
    if (lane < NFP) { s_box[jl][0][lane] = __builtin_inf(); s_box[jl][1][lane] = NEG_INF; }
    // modulus + absorption per filter (compile-time indices into the kernel argument: a run-time index would copy the array to scratch)
#pragma unroll
    for (int f = 0; f < NFP; ++f) if (tid == f) s_shift[f] = s_par[B9_P_MOD] + pk.abs_m1[f] * s_par[B9_P_ABS];
    __syncthreads();
    HSTAMP(3);
    if (c == 0 && tid == 0) { h.agb_tip = s_mass[n - 1]; *hp = h; }
    double *out = mx.tab + (cset * rows + wp) * mx.L.total;
    const MargLayout &L = mx.L;
    const int node = c * 64 + lane, sub = lane >> 4, i16 = lane & 15, u = c * 4 + sub;
    // the primary
    bool ok = node < n_nodes;
    const int e = ok ? node / K : e0, s = node - e * K;
    const double a = s_mass[e], d = s_mass[e + 1] - a;
    ok = ok && d > 0.0;
    const double dM = d / K;
    const double m1 = fma((double)s, dM, a);
    const double t1 = ok ? (m1 - a) / d : 0.0;
    const double *const pr0 = s_prim + (size_t)(e - e0) * NFP;           // the primary's two rows
    for (int j = jl; j < Q; j += 4) {
        double *row = out + L.o_rows + (((size_t)u * Q + j) * 16 + i16) * NFP;
        int lo2 = 0; double t2 = 0.0;
        bool dark2 = false, tiled = true;
        const double *sr0 = s_sec;                                        // the companion's two rows in the wave's tile
        if (j > 0) {
            if (ok) {
                // companion below the isochrone's first point: no flux, magnitude 99.999
                const double m2 = ((double)j / (double)Q) * m1;
                dark2 = m2 < s_mass[0];
                find_bracket(s_mass, n, m2, lo2, t2);
            }
            // the run of rows the wave's companions bracket, derived once into the wave's tile
            const bool need = ok && !dark2;
            int lmin = need ? lo2 : 2147483647, lmax = need ? lo2 : -1;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { const int a2 = __shfl_xor(lmin, o, 64), b2 = __shfl_xor(lmax, o, 64); lmin = a2 < lmin ? a2 : lmin; lmax = b2 > lmax ? b2 : lmax; }
            lmin = __builtin_amdgcn_readfirstlane(lmin); lmax = __builtin_amdgcn_readfirstlane(lmax);
            const int n_sec = lmax >= 0 ? lmax - lmin + 2 : 0;
            tiled = n_sec <= B9_MSTEP_SEC_ROWS;
            if (tiled) {
                __builtin_amdgcn_wave_barrier();                          // (the previous mass ratio's reads of the tile are done)
                for (int i = lane; i < n_sec * NFP; i += 64) {
                    const int r = i / NFP, f = i - r * NFP;
                    s_sec[i] = f < pk.nf ? corner_interp<false>(pk, cr, h, ny, lmin + r, f) : 0.0;
                }
                __builtin_amdgcn_wave_barrier();                          // (one wave: its LDS accesses complete in program order)
                sr0 = s_sec + (size_t)(need ? lo2 - lmin : 0) * NFP;
            }
        }
#pragma unroll 2
        for (int f = 0; f < NFP; ++f) {
            const double shift = s_shift[f];
            double C = 0.0;
            if (ok) {
                const double p1f = lerp(pr0[f], pr0[NFP + f], t1);
                if (j == 0) C = p1f + shift;
                else {
                    double r20, r21;
                    if (tiled) { r20 = sr0[f]; r21 = sr0[NFP + f]; }
                    else {
                        r20 = f < pk.nf ? corner_interp<false>(pk, cr, h, ny, lo2, f) : 0.0;
                        r21 = f < pk.nf ? corner_interp<false>(pk, cr, h, ny, lo2 + 1, f) : 0.0;
                    }
                    const double p2 = dark2 ? B9_MAG_NOFLUX : lerp(r20, r21, t2);
                    const double comb = (-2.5 / LN10) * log_pos(exp_fast((-0.4 * LN10) * p1f) + exp_fast((-0.4 * LN10) * p2));
                    C = comb + shift;
                }
            }
            row[f] = C;
            // boxes: a NaN magnitude stays out of them (fmin / fmax ignore it); its term is dropped by the star loop's X < xcut
            double lo = ok ? C : __builtin_inf(), hi = ok ? C : NEG_INF;
            row_min_max(lo, hi);
            if (i16 == 0) box_store<NFP>(box32, out + L.o_box2 + ((size_t)u * Q + j) * 2 * NFP, out + L.o_box2f + ((size_t)u * Q + j) * NFP, f, lo, hi);
            rows_min_max(lo, hi);
            if (lane == 0) { s_box[jl][0][f] = __builtin_fmin(s_box[jl][0][f], lo); s_box[jl][1][f] = __builtin_fmax(s_box[jl][1][f], hi); }
        }
    }
    HSTAMP(4);
    if (jl == 0) {                // nb = -2 log(prior(m1) dM / Q) of every node, and its minima
        const double nb = ok ? -2.0 * (log_prior_mass_dev(pk.log_mass_norm, m1) + log_pos(dM / Q)) : __builtin_inf();
        out[L.o_nb + node] = nb;
        double mn = nb;
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) mn = __builtin_fmin(mn, __shfl_xor(mn, o, 64));
        if (i16 == 0) out[L.o_nbmin16 + u] = mn;
        mn = __builtin_fmin(mn, __shfl_xor(mn, 16, 64)); mn = __builtin_fmin(mn, __shfl_xor(mn, 32, 64));
        if (lane == 0) out[L.o_nbmin64 + c] = mn;
    }
    __syncthreads();
    if (tid < NFP) {
        double lo = s_box[0][0][tid], hi = s_box[0][1][tid];
        for (int k = 1; k < 4; ++k) { lo = __builtin_fmin(lo, s_box[k][0][tid]); hi = __builtin_fmax(hi, s_box[k][1][tid]); }
        box_store<NFP>(box32, out + L.o_box1 + (size_t)c * 2 * NFP, out + L.o_box1f + (size_t)c * NFP, tid, lo, hi);
    }
    HSTAMP(5);
}

// WD-table builder: k_marg_wd_table's rows for (walker w, candidate cand, population pop), mass steps
// 1 + (blk * 2 + wave / 2) * 64 + lane, atmosphere type = wave & 1.  The WD chain needs the candidate isochrone's header
// only (its AGB-tip mass and grid cell): derived here, nothing is read from another workgroup of the launch.
template <int NFP>
__device__ __forceinline__ void marg_build_wd_table(const DevPack &pk, const StepDev &sd, const MargStep &mx, int w, int cand, int pop, int blk)
{
    if (!sd.derive_next) return;
    const int tid = threadIdx.x, lane = tid & 63, W = sd.n_walkers, n_pops = sd.n_pops, K = mx.K, steps = 8 * K;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), type = wave & 1;
    __shared__ double s_par[B9_NPARAM], s_z[12];
    if (wave == 0) candidate_row_wave0<true>(sd, w, cand, s_par, s_z);
    __syncthreads();
    const size_t cset = (size_t)((sd.set ^ 1) * 2 + cand);
    const int wp = w * n_pops + pop, n_wp = W * n_pops;
    CornerRegs cr;
    const IsoHdr h = marg_header(pk, s_par, pop, sd.mass_cap, cr);
    if (!h.valid) return;
    const int ny = pk.n_y > 1 ? 2 : 1;
    IsoView<NFP> is;
    is.n = h.n; is.tip = corner_interp<true>(pk, cr, h, ny, h.n - 1, 0); is.i_feh = h.i_feh; is.i_y = h.i_y; is.t_feh = h.t_feh; is.t_y = h.t_y;
    is.mass = nullptr; is.mags = nullptr;                               // (every node lies above the tip: the WD branch never reads them)
    const int j = 1 + (blk * 2 + (wave >> 1)) * 64 + lane;
    const double dM = (pk.m_wd_up - is.tip) / steps;
    if (!(dM > 0.0) || j > steps) return;
    WdAxes ax;
    ax.log_age = pk.log_age;
    for (int df = 0; df < 2; ++df) for (int dy = 0; dy < 2; ++dy)
        ax.tips[df * 2 + dy] = pk.tips + (size_t)((is.i_feh + df) * pk.n_y + (is.i_y + (dy < ny ? dy : 0))) * pk.n_age;
    ax.wc_log_age_lds = nullptr; ax.wc_track = pk.wc_track; ax.wc_mass = pk.wc_mass; ax.wc_carb = pk.wc_carb;
    ax.at_log_teff = pk.at_log_teff; ax.at_logg = pk.at_logg;
    const double m1 = is.tip + dM * j, mod = s_par[B9_P_MOD], av = s_par[B9_P_ABS];
    double p[NFP];
    star_mags<NFP>(pk, ax, is, s_par, m1, type, p);
    double *wtab = mx.wd_tab + cset * mx.wd_stride;
    double *row = wtab + (((size_t)wp * 2 + type) * steps + (j - 1)) * NFP;
#pragma unroll
    for (int f = 0; f < NFP; ++f) row[f] = p[f] + (mod + pk.abs_m1[f] * av);
    if (type == 0) wtab[(size_t)n_wp * 2 * steps * NFP + (size_t)wp * steps + (j - 1)] = log_prior_mass_dev(pk.log_mass_norm, m1);
}

// The star roles' choice of candidate: the decision of step t-1, waited for by the workgroup's first wave (wait_decision: the
// walker's writer publishes it) and shared through LDS behind one barrier.  hdr / par / tab / wd: THIS parity's two
// candidates as the launch's own `const __restrict__` kernel arguments -- read through the StepDev / MargStep pointers
// (which the builders of the same launch write through, for the other parity) the compiler cannot prove the tables
// unclobbered and turns every scalar load of a box or an nb word into a vector load + v_readfirstlane: +18 % launch time.
// WARM: L2Warm (b9_star_marg.hip.h) -- both candidates' tables of the walker, read through inside the wait for the decision.
template <bool WARM>
struct MargStepSel {
    const StepDev &sd;
    const MargStep &mx;
    int *s_sel;
    bool wd;                 // the WD-stage stars' role: .tab is the candidate's WD table
    const IsoHdr *__restrict__ hdr;
    const double *__restrict__ par, *__restrict__ tab, *__restrict__ wdt;
    int rank, count;         // WARM: this workgroup among the star workgroups of its XCD
    L2Warm warm;
    __device__ __forceinline__ void issue(int w)
    {
        if constexpr (WARM) {
            const size_t rows = (size_t)sd.n_walkers * sd.n_pops, per = (size_t)sd.n_pops * mx.L.total;      // (doubles per candidate of this walker)
            if (wd) warm.issue(tab, nullptr, 0, 0, 1);
            else warm.issue(tab + (size_t)w * per, tab + rows * mx.L.total + (size_t)w * per, per, rank, count);
        }
    }
    __device__ __forceinline__ MargSel finish(int w)
    {
        if constexpr (WARM) warm.wait();          // (the lines have long arrived when the decision has)
        if (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) == 0) {
            const int s0 = (!MSTEP_NO_DECIDE && wait_decision(sd, w)) ? 1 : 0;
            if (threadIdx.x == 0) *s_sel = s0;
        }
        __syncthreads();
        const size_t rows = (size_t)sd.n_walkers * sd.n_pops;
        const size_t cand = MSTEP_FIXED_CAND ? (size_t)0 : (size_t)__builtin_amdgcn_readfirstlane(*s_sel);
        return MargSel{hdr + cand * rows, par + cand * sd.n_walkers * B9_NPARAM, wd ? wdt + cand * mx.wd_stride : tab + cand * rows * mx.L.total};
    }
};

template <int NFP, int NPOPS, bool SPLIT, int TILE>
__device__ __forceinline__ int marg_step_body(const DevPack &pk, const DevStars &st, const StepDev &sd, const DevPriors &pr, const MargStep &mx,
                                              int front_blocks, int star_blocks, const IsoHdr *__restrict__ hdr_rd, const double *__restrict__ par_rd,
                                              const double *__restrict__ tab_rd, const double *__restrict__ wd_rd, double *smem)
{
    const int W = sd.n_walkers;
    int b = blockIdx.x;
    if (b < front_blocks) {
        if (MSTEP_NO_FRONT) return 3;
        if (b < W) { step_derive(pk, sd, pr, b, 0, 0, 0, 1, nullptr, 1); return 4; }      // the writer of walker b
        b -= W;
        const int n_tab = W * 2 * NPOPS * mx.n_chunks_cap;
        if (b < n_tab) {        // b = ((w * 2 + cand) * NPOPS + pop) * n_chunks_cap + c
            const int c = b % mx.n_chunks_cap; b /= mx.n_chunks_cap;
            const int pop = b % NPOPS; b /= NPOPS;
            marg_build_table<NFP, NPOPS>(pk, sd, mx, b >> 1, b & 1, pop, c, smem);
            return 2;
        }
        b -= n_tab;
        if (b < W * 2 * NPOPS * mx.n_wd_blocks) {
            const int blk = b % mx.n_wd_blocks; b /= mx.n_wd_blocks;
            const int pop = b % NPOPS; b /= NPOPS;
            marg_build_wd_table<NFP>(pk, sd, mx, b >> 1, b & 1, pop, blk);
            return 5;
        }
        return 3;
    }
    b -= front_blocks;
    if (MSTEP_NO_STARS) return 3;
    __shared__ int s_sel;
    MargStepSel<SPLIT> select{sd, mx, &s_sel, b >= star_blocks, hdr_rd, par_rd, tab_rd, wd_rd, b >> 3, (star_blocks + 7) >> 3, {}};
    double *const partial = sd.partial + (size_t)sd.set * (sd.partial_stride / 2);
    const MargSample ms{};
    if (b < star_blocks) {
        star_marg_body<NFP, NPOPS, false, SPLIT, false, TILE>(pk, st, b, smem, nullptr, 0, sd.mass_cap, partial, sd.partial_stride, nullptr, mx.K, mx.Q, ms, mx.L,
                                                        W, mx.cut2, mx.wsplit, mx.shares, select);
        return 0;
    }
    b -= star_blocks;
    const int nb = (st.n_wd + 3) / 4;                        // WD-stage stars: b = w * ceil(n_wd / 4) + group of four
    star_marg_wd_body<NFP, NPOPS, false>(pk, st, b % nb, b / nb, W, nullptr, 0, sd.mass_cap, partial, sd.partial_stride, nullptr, mx.K, ms, select);
    return 1;
}


// (TILE: the star role's rows through the dynamic LDS the builders' tiles occupy in THEIR workgroups -- star_marg_body)
template <int NFP, int NPOPS, bool SPLIT, int TILE = 1>
__global__ __launch_bounds__(256, TILE ? B9_TILE_OCC(NFP, NPOPS, TILE) : B9_MSTEP_WAVES(NFP, NPOPS))
void k_marg_step(DevPack pk, DevStars st, StepDev sd, DevPriors pr, MargStep mx, int front_blocks, int star_blocks,
                 const IsoHdr *__restrict__ hdr_rd, const double *__restrict__ par_rd, const double *__restrict__ tab_rd, const double *__restrict__ wd_rd)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    B9_GANTT_ENTER();
    const int role = marg_step_body<NFP, NPOPS, SPLIT, TILE>(pk, st, sd, pr, mx, front_blocks, star_blocks, hdr_rd, par_rd, tab_rd, wd_rd, smem);
    B9_GANTT_EXIT(sd.step, role);
}

// The per-star shares of a split k_marg_step launch merged (k_marg_merge's sums), against the candidate that launch's
// writer recorded in the state row it wrote (B9_ST_SEL of parity `set`).
template <int NPOPS>
__global__ __launch_bounds__(64) void k_marg_step_merge(DevStars st, StepDev sd, MargStep mx)
{
    const int w = blockIdx.y, W = sd.n_walkers;
    const size_t cset = (size_t)(sd.set * 2 + (sd.state[((size_t)sd.set * W + w) * B9_STATE_STRIDE + B9_ST_SEL] != 0.0 ? 1 : 0));
    marg_merge_body<NPOPS>(st, sd.cand_hdr + cset * W * NPOPS, sd.cand_par + cset * W * B9_NPARAM,
                           sd.partial + (size_t)sd.set * (sd.partial_stride / 2), sd.partial_stride, nullptr, mx.shares);
}
