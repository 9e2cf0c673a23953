// b9_mcmc_step.hip.h -- k_mcmc_step: the fused one-launch sampler step (decision + stars + speculative candidates).
// Part of the single translation unit b9_kernels.hip (included there, in this order); gfx950 only.
#pragma once

// ------------------------------------------------------------------------------------------
// k_mcmc_step: the fused sampler step (see StepDev in b9_device.h).  ONE launch per MCMC step.
//
// Roles by workgroup id:  [heavy-star workgroups][candidate-derivation workgroups][pad to 8][hot].
// Every role starts with step_decide(): the accept/reject decision of the PREVIOUS step, taken
// redundantly by every workgroup of a walker from the same fixed-order sum (identical bits).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ const double *step_state_in(const StepDev &sd, int w)
{
    return sd.state + ((size_t)(sd.set ^ 1) * sd.n_walkers + w) * B9_STATE_STRIDE;
}

// Decision of step t-1 for walker w.  Every WAVE takes it on its own -- lane l adds the partials
// l, l + 64, ... in order, then the shuffle tree -- so there is no LDS traffic and no barrier, all
// waves of all workgroups obtain the same bits, and a wave may use the shortcut below whatever its
// neighbours do.  lp_new = log-posterior of the state after that step (not set on the shortcut).
// SHORTCUT: the caller only needs the 0/1 outcome -- if the walker's writer workgroup (which leads
// the grid) has already published it for this step, take it from there and skip the sum.  Waves
// that start before the writer is done compute it themselves: same bits either way, nobody waits.
#define B9_SHORTCUT true
#define B9_DECIDE_K 6
struct DecideLoads {           // everything one wave's decision reads (registers)
    double lp_cur, lpr, lu, selp, h0, h1, v[B9_DECIDE_K];
    unsigned long long flag;
};

// The loads: EVERYTHING the decision may need is requested here, before any of it is looked at, with clamped
// indices instead of branches: the state words, the writer's flag, the first 64 K hot partials and the
// heavy-star partials of BOTH candidate slots (which one counts is a state word).  One memory round trip;
// the launch's first few microseconds are a chain of such round trips and nothing else.
template <bool SHORTCUT>
__device__ __forceinline__ void decide_issue(const StepDev &sd, int w, DecideLoads &dl)
{
    const int lane = threadIdx.x & 63;
    const double *in = step_state_in(sd, w);
    const double *part = sd.partial + (size_t)w * sd.partial_stride + (size_t)(sd.set ^ 1) * (sd.partial_stride / 2);
    const int n_hot = sd.n_partial - sd.heavy_parts;
    dl.lp_cur = in[B9_ST_LP]; dl.lpr = in[B9_ST_LPRIOR]; dl.lu = in[B9_ST_LOGU]; dl.selp = in[B9_ST_SEL];
    dl.flag = SHORTCUT ? __hip_atomic_load(sd.decided + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
#pragma unroll
    for (int k = 0; k < B9_DECIDE_K; ++k) {
        const int j = lane + 64 * k;
        dl.v[k] = part[j < n_hot ? j : n_hot - 1];
    }
    const int hl = lane < sd.heavy_parts ? lane : sd.heavy_parts - 1;
    dl.h0 = part[n_hot + hl]; dl.h1 = part[n_hot + sd.heavy_parts + hl];
}

template <bool SHORTCUT>
__device__ __forceinline__ bool decide_finish(const StepDev &sd, int w, const DecideLoads &dl, double &lp_new)
{
    const int lane = threadIdx.x & 63;
    if (!sd.has_prev) { lp_new = dl.lp_cur; return false; }
    if (SHORTCUT && (dl.flag >> 1) == sd.step) { lp_new = 0.0; return (dl.flag & 1ull) != 0; }      // wave-uniform: one word per walker
    const int n_hot = sd.n_partial - sd.heavy_parts;
    // lane l: the hot waves' partials l, l + 64, ... in order, then heavy-star partial l of the candidate
    // that step evaluated; then the shuffle tree.  The same bits in every workgroup.
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < B9_DECIDE_K; ++k) acc = (lane + 64 * k < n_hot) ? acc + dl.v[k] : acc;
    if (n_hot > 64 * B9_DECIDE_K) {
        const double *part = sd.partial + (size_t)w * sd.partial_stride + (size_t)(sd.set ^ 1) * (sd.partial_stride / 2);
        for (int j = lane + 64 * B9_DECIDE_K; j < n_hot; j += 64) acc += part[j];
    }
    acc = (lane < sd.heavy_parts) ? acc + (dl.selp != 0.0 ? dl.h1 : dl.h0) : acc;          // (heavy_parts <= 64)
    STAMP(9);
    const double t = __shfl(wave_sum(acc), 0, 64);
    STAMP(10);
    STAMP(11);
    const double lp_prop = (dl.lpr != NEG_INF) ? dl.lpr + t : NEG_INF;       // prior + sum, as k_finalize forms it
    const bool ok = isfinite(lp_prop) && (dl.lu < lp_prop - dl.lp_cur);
    lp_new = ok ? lp_prop : dl.lp_cur;
    return ok;
}

template <bool SHORTCUT>
__device__ __forceinline__ bool step_decide(const StepDev &sd, int w, double &lp_new)
{
    DecideLoads dl;
    decide_issue<SHORTCUT>(sd, w, dl);
    return decide_finish<SHORTCUT>(sd, w, dl, lp_new);
}

// Hot role: k_star_like's body for one walker, with the mass columns and headers of BOTH candidates
// requested before the decision is known (same round trip as the partial sums the decision needs).
template <int NFP, int NPOPS>
__device__ __forceinline__ void step_hot(const DevPack &pk, const DevStars &st, const StepDev &sd, int L,
                                         int group_tiles, int n_groups, int groups_per_block, int n_blocks, double *smem)
{
    const int tid = threadIdx.x, W = sd.n_walkers, mass_cap = sd.mass_cap;
    const int xcd = L & 7, s = L >> 3;
    const int w = s % W;
    const int g = (s / W) * 8 + xcd;                     // this workgroup among the walker's n_blocks * NPOPS
    if (g >= n_blocks * NPOPS) return;
    // its canonical tile groups g, g + n_blocks, ... (TileSeq, b9_star_like.hip.h): group_tiles tiles each, strided over the
    // slot order -- binaries lead that order, so consecutive tiles would give some workgroups only expensive (binary)
    // tiles and others only cheap ones; strided, every workgroup gets its share of both and they finish together
    const TileSeq seq = make_tile_seq<NPOPS>(g, n_blocks, n_groups, group_tiles, groups_per_block, st.n_pad / 256);
    STAMP(0);
    // The decision is taken by the workgroup's first wave alone and travels through LDS behind the barrier the mass
    // columns need anyway: a quarter of the partial-sum traffic at the launch's start.  Its loads leave first.
    __shared__ int s_sel;
    const bool first_wave = __builtin_amdgcn_readfirstlane(tid >> 6) == 0;
    DecideLoads dl = {};
    if (first_wave) decide_issue<B9_SHORTCUT>(sd, w, dl);
    const int i = seq.slot(seq.tile(seq.group(0), 0));       // (a group's first tile always exists)
    const double m1 = st.mass1[i], q = st.q[i], ea = st.ea[i];
    const size_t rows = (size_t)W * NPOPS;
    const size_t cb0 = (size_t)(sd.set * 2) * rows + (size_t)w * NPOPS;      // candidate 0; candidate 1 is `rows` further
    double *const lds_mass = smem;
    // The 2 * NPOPS mass columns are contiguous in LDS, so flat element f of the copy lands at lds2[f].
    // The first FR * 256 elements travel through registers: their loads are issued HERE, before the
    // decision's partial sums are requested, and written to LDS after it -- one memory round trip
    // for everything instead of two.
    constexpr int FR = 4;
    const int half = mass_cap / 2, total2 = 2 * NPOPS * half;
    double2 *const lds2 = reinterpret_cast<double2 *>(lds_mass);
    auto src2 = [&](int f) -> const double2 * {
        const int c = f / half, j = f - c * half;
        return reinterpret_cast<const double2 *>(sd.cand_iso + (cb0 + (size_t)(c / NPOPS) * rows + (c % NPOPS)) * sd.iso_stride) + j;
    };
    static_assert(FR == 4, "the four loads and stores are written out (a loop over an array of them ended in scratch)");
    auto clamp2 = [&](int f) { return f < total2 ? f : total2 - 1; };           // (clamped, not branched: the loads leave back to back)
    const double2 fr0 = *src2(clamp2(tid)), fr1 = *src2(clamp2(tid + 256)), fr2 = *src2(clamp2(tid + 512)), fr3 = *src2(clamp2(tid + 768));
    IsoHdr h[2][NPOPS];
    double pmod[2], pav[2], plam[2];
#pragma unroll
    for (int cand = 0; cand < 2; ++cand) {
#pragma unroll
        for (int kp = 0; kp < NPOPS; ++kp) h[cand][kp] = sd.cand_hdr[cb0 + (size_t)cand * rows + kp];
        const double *par = sd.cand_par + ((size_t)(sd.set * 2 + cand) * W + w) * B9_NPARAM;
        pmod[cand] = par[B9_P_MOD]; pav[cand] = par[B9_P_ABS]; plam[cand] = NPOPS == 2 ? par[B9_P_LAMBDA] : 1.0;
    }
    STAMP(1);
    if (tid < total2) lds2[tid] = fr0;
    if (tid + 256 < total2) lds2[tid + 256] = fr1;
    if (tid + 512 < total2) lds2[tid + 512] = fr2;
    if (tid + 768 < total2) lds2[tid + 768] = fr3;
    for (int f = tid + FR * 256; f < total2; f += 256) lds2[f] = *src2(f);     // very long isochrones only
    if (first_wave) {
        double lp_new;
        const int sel0 = decide_finish<B9_SHORTCUT>(sd, w, dl, lp_new) ? 1 : 0;
        if (tid == 0) s_sel = sel0;
    }
    STAMP(2);
    __syncthreads();                                     // the LDS mass columns, the decision
    const int sel = __builtin_amdgcn_readfirstlane(s_sel);       // (wave-uniform: the selected candidate's view lives in scalar registers)
    STAMP(3);
    IsoView<NFP> iso[NPOPS];
    bool valid = true;
    double tip_min = __builtin_inf();
#pragma unroll
    for (int kp = 0; kp < NPOPS; ++kp) {
        const IsoHdr hh = sel ? h[1][kp] : h[0][kp];
        valid = valid && hh.valid;
        iso[kp].n = hh.n; iso[kp].tip = hh.agb_tip;
        iso[kp].i_feh = hh.i_feh; iso[kp].i_y = hh.i_y; iso[kp].t_feh = hh.t_feh; iso[kp].t_y = hh.t_y;
        iso[kp].mass = lds_mass + (size_t)(sel * NPOPS + kp) * mass_cap;
        iso[kp].mags = sd.cand_iso + (cb0 + (size_t)sel * rows + kp) * sd.iso_stride + mass_cap;
        tip_min = hh.agb_tip < tip_min ? hh.agb_tip : tip_min;
    }
    const double mod = sel ? pmod[1] : pmod[0], av = sel ? pav[1] : pav[0], lam = sel ? plam[1] : plam[0];
    const double log_lam = NPOPS == 2 ? wave_uniform(log(lam)) : 0.0, log_1ml = NPOPS == 2 ? wave_uniform(log1p(-lam)) : 0.0;

    double *const prow = sd.partial + (size_t)w * sd.partial_stride + (size_t)sd.set * (sd.partial_stride / 2);
    hot_groups<NFP, NPOPS, true>(pk, st, seq, iso, valid, tip_min, mod, av, log_lam, log_1ml, i, m1, q, ea, nullptr,
                           [&](int c, int k, double tot) { prow[c * 4 + k] = tot; });
    STAMP(8);
}

// Candidate-derivation role (and, for candidate 0 / population 0 / part 0 of each walker, the
// WRITER of the new state and of the chain record).
__device__ __forceinline__ void step_derive(const DevPack &pk, const StepDev &sd, const DevPriors &pr,
                                            int w, int cand, int pop, int part, int parts, double *s_state_out = nullptr,
                                            int mode = 2 /* 0: derive only  1: writer only  2: (cand 0, pop 0, part 0) is also the writer */)
{
    const int tid = threadIdx.x, d = sd.d, W = sd.n_walkers, n_pops = sd.n_pops;
    __shared__ double s_par[B9_NPARAM], s_z[12], s_cur[B9_NPARAM], s_prop[B9_NPARAM];
    // (only the first wave's threads use the decision; its loads leave first)
    const bool first_wave = __builtin_amdgcn_readfirstlane(tid >> 6) == 0;
    DecideLoads dl = {};
    if (first_wave) decide_issue<false>(sd, w, dl);
    const AxisRegs axr = preload_axis(pk);                 // first round trip, needs no parameter
    // everything the role reads before the isochrone tables is requested now, in one round trip
    const double *in = step_state_in(sd, w);
    const double cur_v = tid < B9_NPARAM ? in[B9_ST_CUR + tid] : 0.0;
    const double prev_prop_v = tid < B9_NPARAM ? in[B9_ST_PROP + tid] : 0.0;
    const size_t rows = (size_t)W * n_pops;
    const double pc0 = tid < B9_NPARAM ? sd.cand_par[((size_t)(sd.set * 2 + 0) * W + w) * B9_NPARAM + tid] : 0.0;
    const double pc1 = tid < B9_NPARAM ? sd.cand_par[((size_t)(sd.set * 2 + 1) * W + w) * B9_NPARAM + tid] : 0.0;
    bool v0 = true, v1 = true;
    for (int k = 0; k < n_pops; ++k) {
        v0 = v0 && sd.cand_hdr[(size_t)(sd.set * 2 + 0) * rows + (size_t)w * n_pops + k].valid;
        v1 = v1 && sd.cand_hdr[(size_t)(sd.set * 2 + 1) * rows + (size_t)w * n_pops + k].valid;
    }
    double crow[11];
#pragma unroll
    for (int j = 0; j < 11; ++j) crow[j] = (tid < d && j < d) ? sd.chol[tid * d + j] : 0.0;
    const int fidx = tid < d ? sd.free_idx[tid] : 0;
    {   // wave 3: the normals of step t+1 (Philox + Box-Muller), independent of every decision
        const int j = tid - 192, n_pairs = (d + 1) >> 1;
        if (j >= 0 && j < n_pairs) {
            unsigned r[4];
            const unsigned long long sn = sd.step + 1;
            philox4x32((unsigned)sn, (unsigned)(sn >> 32), (unsigned)sd.walker_ids[w], (unsigned)j, sd.k0, sd.k1, r);
            const double u1 = u01(r[0], r[1]), u2 = u01(r[2], r[3]);
            const double rad = sqrt(-2.0 * log(u1)), ang = 2.0 * M_PI * u2;
            s_z[2 * j] = rad * cos(ang);
            s_z[2 * j + 1] = rad * sin(ang);
        }
    }
    double lp_new = 0.0;
    bool ok = false;
    if (first_wave) ok = decide_finish<false>(sd, w, dl, lp_new);
    if (tid < B9_NPARAM) {
        s_cur[tid] = ok ? prev_prop_v : cur_v;             // state after step t-1
        s_prop[tid] = ok ? pc1 : pc0;                      // the proposal THIS launch's star workgroups evaluate
    }
    if (s_state_out) {                                     // (k_mcmc_finish: the state after the block, for its summary row)
        if (tid < B9_NPARAM) s_state_out[tid] = ok ? prev_prop_v : cur_v;
        if (tid == 0) { s_state_out[B9_NPARAM] = lp_new; s_state_out[B9_NPARAM + 1] = in[B9_ST_NACC] + ((sd.has_prev && ok) ? 1.0 : 0.0); }
    }
    const bool writer = mode != 0 && (cand == 0 && pop == 0 && part == 0);
    if (writer && tid == 0 && sd.has_prev)                 // publish the outcome for workgroups that start later
        __hip_atomic_store(sd.decided + w, (sd.step << 1) | (ok ? 1ull : 0ull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (writer) {
        double *out = sd.state + ((size_t)sd.set * W + w) * B9_STATE_STRIDE;
        if (tid < B9_NPARAM) { out[B9_ST_CUR + tid] = s_cur[tid]; out[B9_ST_PROP + tid] = s_prop[tid]; }
        if (tid == 0) {
            out[B9_ST_LP] = lp_new;
            const bool pv = ok ? v1 : v0;
            out[B9_ST_LPRIOR] = pv ? log_prior_cluster(pr, s_prop, n_pops) : NEG_INF;
            out[B9_ST_SEL] = ok ? 1.0 : 0.0;
            out[B9_ST_NACC] = in[B9_ST_NACC] + ((sd.has_prev && ok) ? 1.0 : 0.0);
            {   // log u of the accept test of the proposal evaluated by THIS launch (draw index n_pairs of its step)
                unsigned r[4];
                philox4x32((unsigned)sd.step, (unsigned)(sd.step >> 32), (unsigned)sd.walker_ids[w], (unsigned)((d + 1) >> 1), sd.k0, sd.k1, r);
                out[B9_ST_LOGU] = log(u01(r[0], r[1]));
            }
            if (sd.has_prev && sd.lps) sd.lps[(size_t)sd.row * W + w] = lp_new;
        }
        if (sd.has_prev && sd.samples && tid < d) sd.samples[((size_t)sd.row * W + w) * d + tid] = s_cur[fidx];
    }
    if (mode == 1 || !sd.derive_next) return;
    // candidate `cand` of step t+1:  base = state (step t rejected) or step t's proposal (accepted);
    // row[free[i]] += sum_j chol[i][j] z_j  (j ascending, plain multiply-add -- as the host twin does)
    if (tid < B9_NPARAM) s_par[tid] = cand ? s_prop[tid] : s_cur[tid];
    double delta = 0.0;
#pragma unroll
    for (int j = 0; j < 11; ++j) if (j < d) delta = delta + crow[j] * s_z[j];
    __syncthreads();
    if (tid < d) s_par[fidx] += delta;
    __syncthreads();
    const size_t cset = (size_t)((sd.set ^ 1) * 2 + cand);
    if (pop == 0 && part == 0 && tid < B9_NPARAM) sd.cand_par[(cset * W + w) * B9_NPARAM + tid] = s_par[tid];
    derive_iso_block(pk, s_par, pop, w * n_pops + pop, sd.cand_hdr + cset * rows, sd.cand_iso + cset * rows * sd.iso_stride,
                     sd.iso_stride, sd.mass_cap, part, parts, axr);
}

// The decision of step t-1 as walker w's WRITER publishes it (sd.decided[w] = step << 1 | accepted), waited for by one wave.
// The writers are the launch's first workgroups: resident, and deciding, before any workgroup that waits for them is
// dispatched.  Used where re-summing would cost more than waiting: a walker of the marginalised step has one partial per
// 64-star chunk (782 at 50k stars), and 1400 first-round workgroups each reading their walker's 49 freshly rewritten lines
// made every decision of the launch take 7.6 us (the writer's own included) against the 3-4 us of eight writers reading
// alone.  Bounded: a writer that never shows (it cannot, but nothing else depends on that) leaves the decision to the waiter.
__device__ __forceinline__ bool wait_decision(const StepDev &sd, int w)
{
    if (!sd.has_prev) return false;
    for (int spin = 0; spin < (1 << 16); ++spin) {
        const unsigned long long f = __hip_atomic_load(sd.decided + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((f >> 1) == sd.step) return (f & 1ull) != 0;
        __builtin_amdgcn_s_sleep(4);
    }
    double lp_new;
    return step_decide<false>(sd, w, lp_new);
}

// The parameter row of candidate `cand` of step t+1 for walker w, formed by ONE wave (the caller's first) into s_par[12]
// (LDS; s_z[12] is scratch): the decision of step t-1 from its partial sums, the normals of step t+1 (Philox + Box-Muller,
// independent of every decision: while the decision's loads are in flight), and
//     row = base;  row[free[i]] += sum_j chol[i][j] z_j   (j ascending, plain multiply-add -- as the host twin does)
// with base = the state after step t-1 (cand 0: step t rejected) or step t's proposal (cand 1: accepted).  The other waves
// see s_par behind the caller's next workgroup barrier.
// WAIT (the marginalised step, whose walkers have ~800 partial sums each): the decision is not re-summed here but taken from
// the word the walker's writer publishes (wait_decision).
template <bool WAIT = false>
__device__ __forceinline__ void candidate_row_wave0(const StepDev &sd, int w, int cand, double *s_par, double *s_z)
{
    const int tid = threadIdx.x, d = sd.d, W = sd.n_walkers;
    const double *in = step_state_in(sd, w);
    DecideLoads dl = {};
    if (!WAIT) decide_issue<false>(sd, w, dl);
    const double cur_v = tid < B9_NPARAM ? in[B9_ST_CUR + tid] : 0.0;
    const double prev_prop_v = tid < B9_NPARAM ? in[B9_ST_PROP + tid] : 0.0;
    const double pc0 = tid < B9_NPARAM ? sd.cand_par[((size_t)(sd.set * 2 + 0) * W + w) * B9_NPARAM + tid] : 0.0;
    const double pc1 = tid < B9_NPARAM ? sd.cand_par[((size_t)(sd.set * 2 + 1) * W + w) * B9_NPARAM + tid] : 0.0;
    double crow[11];
#pragma unroll
    for (int j = 0; j < 11; ++j) crow[j] = (tid < d && j < d) ? sd.chol[tid * d + j] : 0.0;
    const int fidx = tid < d ? sd.free_idx[tid] : 0;
    {
        const int n_pairs = (d + 1) >> 1;
        if (tid < n_pairs) {
            unsigned r[4];
            const unsigned long long sn = sd.step + 1;
            philox4x32((unsigned)sn, (unsigned)(sn >> 32), (unsigned)sd.walker_ids[w], (unsigned)tid, sd.k0, sd.k1, r);
            const double u1 = u01(r[0], r[1]), u2 = u01(r[2], r[3]);
            const double rad = sqrt(-2.0 * log(u1)), ang = 2.0 * M_PI * u2;
            s_z[2 * tid] = rad * cos(ang);
            s_z[2 * tid + 1] = rad * sin(ang);
        }
    }
    double lp_new = 0.0;
    const bool ok = WAIT ? wait_decision(sd, w) : decide_finish<false>(sd, w, dl, lp_new);
    if (tid < B9_NPARAM) s_par[tid] = cand ? (ok ? pc1 : pc0) : (ok ? prev_prop_v : cur_v);
    __builtin_amdgcn_wave_barrier();                     // (one wave: its LDS accesses complete in program order)
    double delta = 0.0;
#pragma unroll
    for (int j = 0; j < 11; ++j) if (j < d) delta = delta + crow[j] * s_z[j];
    if (tid < d) s_par[fidx] += delta;                   // (each lane owns one sampled parameter)
}

// Derivation role, running AHEAD of the decision (as the tree launch's does, b9_mcmc_tree.hip.h): the role's chain was
// decision -> parameter row -> grid brackets -> corner rows -> table values, five dependent legs.  Only the workgroup's FIRST
// wave takes the decision (its loads leave first), draws step t+1's normals and forms the candidate's parameter row; waves
// 1-3 meanwhile bracket the grid cell of the PREVIOUS state (a candidate is that state plus at most two steps: almost always
// the same cell), read its corner rows and the corner values of the workgroup's share into registers.  When the parameters
// arrive only the interpolation weights, the lerps and the stores remain; a candidate that fell into another cell repeats
// the two trips for its own.  Same draws, same sums, same interpolation as step_derive (the writer's and k_mcmc_finish's
// path) and derive_iso_block: same bits.
#define B9_STEP_KV 3          // output items per thread whose corner values are kept in registers at a time
__device__ __forceinline__ void step_derive_ahead(const DevPack &pk, const StepDev &sd, int w, int cand, int pop, int part, int parts)
{
    if (!sd.derive_next) return;
    const int tid = threadIdx.x, W = sd.n_walkers, n_pops = sd.n_pops;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    __shared__ double s_par[B9_NPARAM], s_z[12];
    const double *in = step_state_in(sd, w);
    const size_t rows = (size_t)W * n_pops;
    const size_t cset = (size_t)((sd.set ^ 1) * 2 + cand);
    if (wave == 0) {
        candidate_row_wave0(sd, w, cand, s_par, s_z);
        __syncthreads();
        if (pop == 0 && part == 0 && tid < B9_NPARAM) sd.cand_par[(cset * W + w) * B9_NPARAM + tid] = s_par[tid];
        return;
    }
    // ---- waves 1..3: the derivation, ahead for the previous state's cell ----
    const int nfp = pk.nfp, mass_cap = sd.mass_cap, wp = w * n_pops + pop;
    const int first = part * 192 + (tid - 64), stride = parts * 192;
    AxisRegs ax[3];
    preload_axes3(pk, ax);
    const double g_age = in[B9_ST_CUR + B9_P_LOGAGE], g_feh = in[B9_ST_CUR + B9_P_FEH];
    const double g_y = in[B9_ST_CUR + (pop ? B9_P_Y2 : B9_P_Y)];
    GridCell cell = grid_cell(pk, ax, g_age, g_feh, g_y);
    CornerRegs cr = corner_rows(pk, cell);
    double v[B9_STEP_KV][8];
    {
        const int total = (cr.n >= 2 && cr.n <= mass_cap) ? cr.n * (nfp + 1) : 0;
        corner_values<B9_STEP_KV>(pk, cr, total, first, stride, v);
    }
    __syncthreads();                                         // the candidate's parameters (first wave)
    const double log_age = s_par[B9_P_LOGAGE], feh = s_par[B9_P_FEH], y = pop ? s_par[B9_P_Y2] : s_par[B9_P_Y];
    const GridCell own = grid_cell(pk, ax, log_age, feh, y);
    const bool same = own.i_age == cell.i_age && own.i_feh == cell.i_feh && own.i_y == cell.i_y;
    if (!same) { cell = own; cr = corner_rows(pk, cell); }
    const IsoHdr h = header_of(pk, cell, cr, log_age, feh, y, mass_cap);
    IsoHdr *hp = sd.cand_hdr + cset * rows + wp;
    if (part == 0 && tid == 64) {
        // (agb_tip of a valid isochrone is stored by the thread that interpolates the last point's mass)
        hp->valid = h.valid; hp->first_eep = h.first_eep; hp->n = h.n; hp->i_feh = h.i_feh; hp->i_y = h.i_y; hp->i_age = h.i_age;
        hp->t_feh = h.t_feh; hp->t_y = h.t_y; hp->t_age = h.t_age;
        if (!h.valid) hp->agb_tip = 0.0;
    }
    if (!h.valid) return;
    const int total = h.n * (nfp + 1);
    double *omass = sd.cand_iso + (cset * rows + wp) * sd.iso_stride;
    double *omags = omass + mass_cap;
    if (!same) corner_values<B9_STEP_KV>(pk, cr, total, first, stride, v);
    store_values<B9_STEP_KV>(pk, h, total, first, stride, v, omass, omags, &hp->agb_tip);
    for (int f = first + B9_STEP_KV * stride; f < total; f += B9_STEP_KV * stride) {        // (few derivation parts: more than KV items per thread)
        corner_values<B9_STEP_KV>(pk, cr, total, f, stride, v);
        store_values<B9_STEP_KV>(pk, h, total, f, stride, v, omass, omags, &hp->agb_tip);
    }
}

// Grid: [one WRITER per walker][derivation workgroups][heavy-star workgroups][pad to 8][hot workgroups]
// (B9_DERIVE_ORDER=0: heavy-star workgroups first; < 0: the derivation workgroups trail the grid instead).
template <int NFP, int NPOPS>
__device__ __forceinline__ int step_body(const DevPack &pk, const DevStars &st, const StepDev &sd, const DevPriors &pr, int group_tiles, int n_groups, int groups_per_block, int n_blocks,
                 int front_blocks, int hot_blocks, int heavy_parts, int derive_parts, int derive_first, double *smem)
{
    const int W = sd.n_walkers, n_heavy = W * heavy_parts, n_derive = W * 2 * NPOPS * derive_parts;
    int b = blockIdx.x;
    // role of this workgroup: 0 hot, 1 heavy, 2 derivation (index b within the role), 3 none (padding)
    int role;
    if (b < front_blocks) {
        // Default order (derive_first == 2): writers and derivation lead, the heavy-star workgroups follow them.  With the
        // heavy-star workgroups in front (B9_DERIVE_ORDER=0) the hot workgroups at grid positions 512.. -- the third
        // workgroups of the compute units the first heavy-star workgroups landed on -- ran 4 us longer than their peers and
        // set the launch's end on the two-population shape (24.1 -> 21.6 us/step); the other shapes do not care.
        if (derive_first == 2) {
            const int n_wd = W + n_derive;
            b = b < n_wd ? b + n_heavy : (b < n_wd + n_heavy ? b - n_wd : b);
        }
        if (b < n_heavy) role = 1;
        else {
            b -= n_heavy;
            if (b < W) role = 4;                                               // the writer of walker b
            else if (derive_first && b - W < n_derive) { role = 2; b -= W; }
            else role = 3;
        }
    } else {
        b -= front_blocks;
        if (b < hot_blocks) role = 0;
        else {                                                                  // trailing derivation workgroups (derive_first == 0)
            b -= hot_blocks;
            role = (!derive_first && b < n_derive && sd.derive_next) ? 2 : 3;
        }
    }
    if (role == 0) {
        B9_MARK("hot-begin");
        step_hot<NFP, NPOPS>(pk, st, sd, b, group_tiles, n_groups, groups_per_block, n_blocks, smem);
        B9_MARK("hot-end");
        return role;
    }
    if (role == 1) {
        const int w = b / heavy_parts, part = b - w * heavy_parts;
        // (raising the role's wave priority -- s_setprio 3 -- was measured: no effect on its chain or on the hot waves)
        HSTAMP(0);
        B9_MARK("heavy-begin");
        const size_t rows = (size_t)W * NPOPS, c0 = (size_t)(sd.set * 2);
        const IsoHdr *const hd[2] = {sd.cand_hdr + c0 * rows, sd.cand_hdr + (c0 + 1) * rows};
        const double *const is[2] = {sd.cand_iso + c0 * rows * sd.iso_stride, sd.cand_iso + (c0 + 1) * rows * sd.iso_stride};
        const double *const pr2[2] = {sd.cand_par + c0 * W * B9_NPARAM, sd.cand_par + (c0 + 1) * W * B9_NPARAM};
        double *const base = sd.partial + (size_t)w * sd.partial_stride + (size_t)sd.set * (sd.partial_stride / 2) + (size_t)n_groups * 4;
        double *const out[2] = {base + part, base + heavy_parts + part};
        // the first wave decides (its loads leave before everything else of the role); the others get it through LDS
        const bool first_wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) == 0;
        DecideLoads dl = {};
        if (first_wave) decide_issue<B9_SHORTCUT>(sd, w, dl);
        auto decide = [&] {
            __shared__ int s_hsel;
            if (first_wave) {
                double lp_new;
                const int s0 = decide_finish<B9_SHORTCUT>(sd, w, dl, lp_new) ? 1 : 0;
                if (threadIdx.x == 0) s_hsel = s0;
            }
            __syncthreads();
            return s_hsel;
        };
        heavy_stars<NFP, NPOPS, 2>(pk, st, hd, is, sd.iso_stride, sd.mass_cap, pr2, decide, w, part, heavy_parts, out, nullptr, smem);
        B9_MARK("heavy-end");
        return role;
    }
    if (role == 2) {       // b = ((w * 2 + cand) * NPOPS + pop) * derive_parts + part
        const int part = b % derive_parts; b /= derive_parts;
        const int pop = b % NPOPS; b /= NPOPS;
        step_derive_ahead(pk, sd, b >> 1, b & 1, pop, part, derive_parts);
    }
    // The WRITER of a walker (decision, new state, chain record) is a workgroup of its own: as part 0 of a derivation
    // it made that workgroup the launch's last (state + prior + log u before its share of the isochrone: 9.1 against
    // 7.1 us on the small shapes, where the derivation is the critical path).
    if (role == 4) step_derive(pk, sd, pr, b, 0, 0, 0, 1, nullptr, 1);
    return role;
}

template <int NFP, int NPOPS>
__global__ __launch_bounds__(256, B9_K1_WAVES(NFP, NPOPS))
void k_mcmc_step(DevPack pk, DevStars st, StepDev sd, DevPriors pr, int group_tiles, int n_groups, int groups_per_block, int n_blocks,
                 int front_blocks, int hot_blocks, int heavy_parts, int derive_parts, int derive_first)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    B9_GANTT_ENTER();
    const int role = step_body<NFP, NPOPS>(pk, st, sd, pr, group_tiles, n_groups, groups_per_block, n_blocks, front_blocks, hot_blocks, heavy_parts, derive_parts, derive_first, smem);
    B9_GANTT_EXIT(sd.step, role);
}

// Summary row of walker w over the block (the multi-GPU driver all-gathers these rows straight from HBM and pools
// them into the adaptive proposal): [lp, position(12), #moves, n, sum x, sum x x^T], x = chain sample - origin.
// Every sum runs over the steps in ascending order with a plain multiply and add (no fma), so a host loop in that
// order (b9h::summary_rows) gives the same bits.  One workgroup; s_last = the walker's state after the block.
#define B9_ROWS_CHUNK 128
__device__ __forceinline__ void block_summary_row(const StepDev &sd, int w, const double *s_last /* LDS: [12] position, [12] lp */)
{
    const int tid = threadIdx.x, d = sd.d, W = sd.n_walkers, S = sd.n_steps;
    __shared__ double s_x[(B9_ROWS_CHUNK + 1) * 11];       // row 0 = the last sample of the previous chunk
    __shared__ int s_moved;
    __shared__ double s_org[11];
    if (tid == 0) s_moved = 0;
    if (tid < d) s_org[tid] = sd.row_origin[tid];
    __syncthreads();
    const int i = tid / d, j = tid - i * d;                // thread (i, j) of the first d*d: sum x_i x_j; threads 128.. : sum x_i
    const bool pair = tid < d * d, single = tid >= 128 && tid < 128 + d;
    double acc = 0.0;
    for (int s0 = 0; s0 < S; s0 += B9_ROWS_CHUNK) {
        const int ns = (S - s0) < B9_ROWS_CHUNK ? (S - s0) : B9_ROWS_CHUNK;
        if (s0 > 0 && tid < d) s_x[tid] = s_x[B9_ROWS_CHUNK * 11 + tid];         // carry the previous chunk's last sample
        __syncthreads();
        for (int e = tid; e < ns * d; e += 256) {
            const int s = e / d, k = e - s * d, step = s0 + s;
            // the block's last sample is this kernel's own output: take it from the state in LDS, not from memory
            const double v = (step == S - 1) ? s_last[sd.free_idx[k]] : sd.samples[((size_t)step * W + w) * d + k];
            s_x[(s + 1) * 11 + k] = v - s_org[k];
        }
        __syncthreads();
        if (pair) for (int s = 1; s <= ns; ++s) acc = acc + s_x[s * 11 + i] * s_x[s * 11 + j];
        if (single) for (int s = 1; s <= ns; ++s) acc = acc + s_x[s * 11 + (tid - 128)];
        int moved = 0;
        for (int s = 1 + tid; s <= ns; s += 256) {
            if (s0 + s - 1 == 0) continue;                 // the block's first step has no predecessor in the block
            bool diff = false;
            for (int k = 0; k < d; ++k) diff = diff || (s_x[s * 11 + k] != s_x[(s - 1) * 11 + k]);
            moved += diff ? 1 : 0;
        }
        if (moved) atomicAdd(&s_moved, moved);
        __syncthreads();
    }
    double *row = sd.rows + (size_t)w * B9_ROW_LEN(d);
    double *hrow = sd.host_rows ? sd.host_rows + (size_t)w * B9_ROW_LEN(d) : nullptr;        // the mapped host mirror
    if (pair) { row[B9_ROW_SUM + d + tid] = acc; if (hrow) hrow[B9_ROW_SUM + d + tid] = acc; }
    if (single) { row[B9_ROW_SUM + (tid - 128)] = acc; if (hrow) hrow[B9_ROW_SUM + (tid - 128)] = acc; }
    if (tid >= 192 && tid < 192 + B9_NPARAM) {
        row[B9_ROW_POS + (tid - 192)] = s_last[tid - 192];
        if (hrow) hrow[B9_ROW_POS + (tid - 192)] = s_last[tid - 192];
    }
    if (tid == 255) {
        row[B9_ROW_LP] = s_last[B9_NPARAM]; row[B9_ROW_MOVED] = (double)s_moved; row[B9_ROW_N] = (double)S;
        if (hrow) { hrow[B9_ROW_LP] = s_last[B9_NPARAM]; hrow[B9_ROW_MOVED] = (double)s_moved; hrow[B9_ROW_N] = (double)S; }
    }
}

// the block's last decision: one workgroup per walker, writer role only (+ the block's summary rows)
__global__ __launch_bounds__(256) void k_mcmc_finish(DevPack pk, StepDev sd, DevPriors pr)
{
    __shared__ double s_last[B9_NPARAM + 2];
    step_derive(pk, sd, pr, blockIdx.x, 0, 0, 0, 1, (sd.rows || sd.host_state) ? s_last : nullptr);
    if (sd.rows) {                                           // (uniform over the grid)
        __syncthreads();
        block_summary_row(sd, blockIdx.x, s_last);
    }
    if (sd.host_state) {
        // what the host reads of the walker's final state row: position, log-posterior, accepted count (from LDS)
        if (!sd.rows) __syncthreads();
        double *h = sd.host_state + (size_t)blockIdx.x * B9_STATE_STRIDE;
        if (threadIdx.x < B9_NPARAM) h[B9_ST_CUR + threadIdx.x] = s_last[threadIdx.x];
        if (threadIdx.x == B9_NPARAM) { h[B9_ST_LP] = s_last[B9_NPARAM]; h[B9_ST_NACC] = s_last[B9_NPARAM + 1]; }
    }
}

// Summary rows of a block whose steps were two launches each (marginalised mode): the chain record is on the device,
// the walker's final state in `cur_fin` / `lp_fin`; same sums, same order, same bits as k_mcmc_finish's rows.
__global__ __launch_bounds__(256) void k_chain_rows(StepDev sd, const double *__restrict__ cur_fin, const double *__restrict__ lp_fin)
{
    __shared__ double s_last[B9_NPARAM + 2];
    const int w = blockIdx.x, tid = threadIdx.x;
    if (tid < B9_NPARAM) s_last[tid] = cur_fin[(size_t)w * B9_NPARAM + tid];
    if (tid == B9_NPARAM) { s_last[B9_NPARAM] = lp_fin[w]; s_last[B9_NPARAM + 1] = 0.0; }
    __syncthreads();
    block_summary_row(sd, w, s_last);
}

// The block's opening: ONE small launch copies the block's upload (starting state, proposal factor, moment origin,
// RNG streams, cleared counters) from the pinned host mirror, mapped into the device, into the device block -- and, for a
// block that continues its predecessor (B9_BLOCK_CONTINUE), takes the starting state from that block's final state
// rows instead.  It replaces a host-to-device copy command and a separate continue kernel (two dispatch boundaries and
// ~10 us of a short block's fixed cost).
__global__ __launch_bounds__(256) void k_mcmc_begin(const double *__restrict__ host_up, double *__restrict__ dev, int up_words,
                                                    const double *__restrict__ prev_final, double *__restrict__ cur0,
                                                    double *__restrict__ lp0, double *__restrict__ state0, int n_walkers)
{
    const int tid = threadIdx.x;
    // (the source is HOST memory: four words per thread are requested together -- one PCIe round trip for up to 1024 words,
    //  not one per loop iteration)
    for (int i0 = 0; i0 < up_words; i0 += 1024) {
        double v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { const int i = i0 + tid + 256 * k; v[k] = host_up[i < up_words ? i : up_words - 1]; }
#pragma unroll
        for (int k = 0; k < 4; ++k) { const int i = i0 + tid + 256 * k; if (i < up_words) dev[i] = v[k]; }
    }
    if (!prev_final) return;
    __syncthreads();
    for (int w = 0; w < n_walkers; ++w) {
        const double *src = prev_final + (size_t)w * B9_STATE_STRIDE;
        double *dst = state0 + (size_t)w * B9_STATE_STRIDE;
        if (tid < B9_NPARAM) { const double v = src[B9_ST_CUR + tid]; cur0[(size_t)w * B9_NPARAM + tid] = v; dst[B9_ST_CUR + tid] = v; }
        if (tid == B9_NPARAM) { const double lp = src[B9_ST_LP]; lp0[w] = lp; dst[B9_ST_LP] = lp; dst[B9_ST_LPRIOR] = NEG_INF; }
    }
}



// B9_BLOCK_CONTINUE: the next block's starting state is the previous block's final state, copied on the device
// (state rows of the final parity -> this block's cur0 / lp0 for D0 and its parity-0 state rows for K(0)).
__global__ void k_mcmc_continue(const double *__restrict__ prev_final, double *__restrict__ cur0, double *__restrict__ lp0,
                                double *__restrict__ state0)
{
    const int w = blockIdx.x, tid = threadIdx.x;
    const double *src = prev_final + (size_t)w * B9_STATE_STRIDE;
    double *dst = state0 + (size_t)w * B9_STATE_STRIDE;
    if (tid < B9_NPARAM) { const double v = src[B9_ST_CUR + tid]; cur0[(size_t)w * B9_NPARAM + tid] = v; dst[B9_ST_CUR + tid] = v; }
    if (tid == B9_NPARAM) { const double lp = src[B9_ST_LP]; lp0[w] = lp; dst[B9_ST_LP] = lp; dst[B9_ST_LPRIOR] = NEG_INF; }
}
