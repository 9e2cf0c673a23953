// b9_mcmc_tree.hip.h -- k_mcmc_tree: the tree-speculative sampler step (TreeDev in b9_device.h): one launch advances
// every chain by `depth` Metropolis steps.  For launches with few walkers, where the one-step fused launch
// (k_mcmc_step) leaves most of the GPU idle and a chain's rate is set by the launch's latency chain, not by its work.
// Part of the single translation unit b9_kernels.hip; gfx950 only.  Included there TWICE (no include guard), inside the
// namespaces tree_kd3 / tree_kd5 with B9_TREE_KD = 3 / 5: the walk reads a node's partial sums in one round trip, KD words
// per lane, so KD bounds the catalogue's canonical tile groups (16 KD).  Three words serve the BASELINE single-chain shapes
// at the registers they always had; five let a 50 000-star catalogue (66 groups) keep the tree launch at 2 - 4 walkers.


__device__ __forceinline__ int tree_level(int n) { return 32 - __clz(n + 1); }          // floor(log2(n + 1)) + 1

// Everything a wave's walk of the previous launch's tree reads, requested before any of it is looked at (clamped
// indices instead of branches: one memory round trip).
struct TreeLoads {
    double lp, logu[B9_TREE_MAX_DEPTH], lpr[B9_TREE_MAX_NODES], hv[B9_TREE_MAX_NODES], v[B9_TREE_MAX_NODES][B9_TREE_KD];
};

struct TreeWalk {
    int outcome;                     // accept bits of the levels walked (level 1 most significant)
    int last;                        // the last accepted node, -1 if none: the state after the walk is its proposal (or the old state)
    int n_acc;
    double lp;                       // log-posterior of the state after the walk
    int node[B9_TREE_MAX_DEPTH];     // per level: the node tested ...
    int last_at[B9_TREE_MAX_DEPTH];  // ... and the last accepted node after that level's test (-1: none yet)
    double lp_at[B9_TREE_MAX_DEPTH]; // ... and the log-posterior of the state after it
};

__device__ __forceinline__ const double *tree_state_in(const TreeDev &td, int w)
{
    return td.state + ((size_t)(td.set ^ 1) * td.n_walkers + w) * B9_TREE_STATE_STRIDE;
}
__device__ __forceinline__ const double *tree_partial_in(const TreeDev &td, int w)
{
    const int NN = (1 << td.depth) - 1;
    return td.partial + ((size_t)(td.set ^ 1) * td.n_walkers + w) * NN * td.part_stride;
}

__device__ __forceinline__ void tree_issue(const TreeDev &td, int w, TreeLoads &tl)
{
    const int lane = threadIdx.x & 63, NN = (1 << td.depth) - 1, n_hot = 4 * td.n_groups;
    const double *in = tree_state_in(td, w), *part = tree_partial_in(td, w);
    tl.lp = in[B9_TS_LP];
#pragma unroll
    for (int j = 0; j < B9_TREE_MAX_DEPTH; ++j) tl.logu[j] = in[B9_TS_LOGU + j];
    const int hl = lane < td.heavy_parts ? lane : td.heavy_parts - 1;
#pragma unroll
    for (int n = 0; n < B9_TREE_MAX_NODES; ++n) {
        tl.lpr[n] = in[B9_TS_LPRIOR + n];
        const double *row = part + (size_t)(n < NN ? n : 0) * td.part_stride;
#pragma unroll
        for (int k = 0; k < B9_TREE_KD; ++k) { const int j = lane + 64 * k; tl.v[n][k] = row[j < n_hot ? j : n_hot - 1]; }
        tl.hv[n] = row[n_hot + hl];
    }
}

// one accept test: prior + sum, as k_finalize forms it
__device__ __forceinline__ bool tree_test(double lpr, double t, double logu, double lp_base, double &lp_prop)
{
    lp_prop = (lpr != NEG_INF) ? lpr + t : NEG_INF;
    return isfinite(lp_prop) && (logu < lp_prop - lp_base);
}

// The walk: per node, lane l adds the hot waves' partials l, l + 64, ... in order, then heavy-star partial l, then the
// shuffle tree -- the same bits in every workgroup; then the sequential algorithm's tests, level by level.
__device__ __forceinline__ TreeWalk tree_walk(const TreeDev &td, const TreeLoads &tl)
{
    const int lane = threadIdx.x & 63, n_hot = 4 * td.n_groups;      // (the launch plan keeps n_hot <= 64 B9_TREE_KD)
    double T[B9_TREE_MAX_NODES];
#pragma unroll
    for (int n = 0; n < B9_TREE_MAX_NODES; ++n) {
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < B9_TREE_KD; ++k) acc = (lane + 64 * k < n_hot) ? acc + tl.v[n][k] : acc;
        acc = (lane < td.heavy_parts) ? acc + tl.hv[n] : acc;
        T[n] = acc;
    }
    {
        static_assert(B9_TREE_MAX_NODES == 7, "wave_sum7");
        double S[B9_TREE_MAX_NODES];
        wave_sum7(T, S);
#pragma unroll
        for (int n = 0; n < B9_TREE_MAX_NODES; ++n) T[n] = S[n];
    }
    TreeWalk tw;
    tw.outcome = 0; tw.last = -1; tw.n_acc = 0; tw.lp = tl.lp;
#pragma unroll
    for (int j = 0; j < B9_TREE_MAX_DEPTH; ++j) { tw.node[j] = 0; tw.last_at[j] = -1; tw.lp_at[j] = tl.lp; }
    // level 1: node 0
    if (td.levels_prev >= 1) {
        double lpp;
        const bool ok = tree_test(tl.lpr[0], T[0], tl.logu[0], tw.lp, lpp);
        tw.node[0] = 0;
        if (ok) { tw.lp = lpp; tw.last = 0; ++tw.n_acc; }
        tw.outcome = ok ? 1 : 0;
        tw.last_at[0] = tw.last; tw.lp_at[0] = tw.lp;
    }
    // level 2: node 1 + b1
    if (td.levels_prev >= 2) {
        const int b1 = tw.outcome & 1, n = 1 + b1;
        double lpp;
        // (unary plus: a conditional on two array ELEMENTS is an lvalue -- a select of addresses, which keeps the whole
        //  array out of registers; on two VALUES it is a register select)
        double l1 = tl.lpr[1], l2 = tl.lpr[2];
        asm volatile("" : "+v"(l1), "+v"(l2));
        const bool ok = tree_test(b1 ? l2 : l1, b1 ? +T[2] : +T[1], tl.logu[1], tw.lp, lpp);
        tw.node[1] = n;
        if (ok) { tw.lp = lpp; tw.last = n; ++tw.n_acc; }
        tw.outcome = tw.outcome * 2 + (ok ? 1 : 0);
        tw.last_at[1] = tw.last; tw.lp_at[1] = tw.lp;
    }
    // level 3: node 3 + (b1 b2)
    if (td.levels_prev >= 3) {
        const int p = tw.outcome & 3, n = 3 + p;
        // (the four candidates pass through an empty asm: left alone, the compiler turns the select chain over array elements
        //  into ONE load at a selected offset -- which pins the whole array in scratch memory)
        double l3 = tl.lpr[3], l4 = tl.lpr[4], l5 = tl.lpr[5], l6 = tl.lpr[6];
        asm volatile("" : "+v"(l3), "+v"(l4), "+v"(l5), "+v"(l6));
        const double lpr = p == 0 ? l3 : (p == 1 ? l4 : (p == 2 ? l5 : l6));
        const double t = p == 0 ? +T[3] : (p == 1 ? +T[4] : (p == 2 ? +T[5] : +T[6]));
        double lpp;
        const bool ok = tree_test(lpr, t, tl.logu[2], tw.lp, lpp);
        tw.node[2] = n;
        if (ok) { tw.lp = lpp; tw.last = n; ++tw.n_acc; }
        tw.outcome = tw.outcome * 2 + (ok ? 1 : 0);
        tw.last_at[2] = tw.last; tw.lp_at[2] = tw.lp;
    }
    return tw;
}

// issue + walk by one wave (the loads of everything else the caller needs first should precede the call in program order)
__device__ __forceinline__ TreeWalk tree_decide(const TreeDev &td, int w)
{
    TreeLoads tl;
    WSTAMP(1);
    tree_issue(td, w, tl);
    WSTAMP_LANDED(2, 3);                 // (diagnostic build: waits for the loads and stamps both sides)
    const TreeWalk tw = tree_walk(td, tl);
    if (tw.outcome >= 0) WSTAMP(4);
    return tw;
}

// index of candidate (walker w, outcome o of the previous launch, node n) in cand_par / cand_hdr / cand_iso of parity `set`
__device__ __forceinline__ size_t tree_cand(const TreeDev &td, int set, int w, int o, int n)
{
    const int NN = (1 << td.depth) - 1, NO = 1 << td.depth;
    return (((size_t)set * td.n_walkers + w) * NO + o) * NN + n;
}

// ---- hot role: one node's star likelihood (k_star_like's hot body on the candidate the walk selects) --------------------
template <int NFP, int NPOPS>
__device__ __forceinline__ void tree_hot(const DevPack &pk, const DevStars &st, const TreeDev &td, int L, int group_tiles, double *smem)
{
    const int tid = threadIdx.x, W = td.n_walkers, mass_cap = td.mass_cap, NN = (1 << td.depth) - 1, V = W * NN;
    const int xcd = L & 7, s = L >> 3;
    const int v = s % V, gb = (s / V) * 8 + xcd;             // every node of every walker re-reads a star tile from one XCD's L2
    if (gb >= td.n_groups * NPOPS) return;
    const int w = v / NN, n = v - w * NN;
    if (tree_level(n) > td.levels) return;                   // (a block's last launch may evaluate fewer levels)
    // one canonical tile group per workgroup (TileSeq, b9_star_like.hip.h): tiles group, group + n_groups, ...
    const TileSeq seq = make_tile_seq<NPOPS>(gb, td.n_groups, td.n_groups, group_tiles, 1, st.n_pad / 256);
    __shared__ int s_o;
    const bool first_wave = __builtin_amdgcn_readfirstlane(tid >> 6) == 0;
    const int i = seq.slot(seq.g);                           // (the group's first tile is tile number `group`)
    const double m1 = st.mass1[i], q = st.q[i], ea = st.ea[i];
    if (first_wave) {
        const int o = tree_decide(td, w).outcome;
        if (tid == 0) s_o = o;
    }
    __syncthreads();
    const size_t cb = tree_cand(td, td.set, w, s_o, n);
    double *const lds_mass = smem;
    {
        const int half = mass_cap / 2;
        double2 *const lds2 = reinterpret_cast<double2 *>(lds_mass);
        for (int f = tid; f < NPOPS * half; f += 256) {
            const int c = f / half, j = f - c * half;
            lds2[f] = reinterpret_cast<const double2 *>(td.cand_iso + (cb * NPOPS + c) * td.iso_stride)[j];
        }
    }
    IsoView<NFP> iso[NPOPS];
    bool valid = true;
    double tip_min = __builtin_inf();
#pragma unroll
    for (int kp = 0; kp < NPOPS; ++kp) {
        const IsoHdr hh = td.cand_hdr[cb * NPOPS + kp];
        valid = valid && hh.valid;
        iso[kp].n = hh.n; iso[kp].tip = hh.agb_tip;
        iso[kp].i_feh = hh.i_feh; iso[kp].i_y = hh.i_y; iso[kp].t_feh = hh.t_feh; iso[kp].t_y = hh.t_y;
        iso[kp].mass = lds_mass + (size_t)kp * mass_cap;
        iso[kp].mags = td.cand_iso + (cb * NPOPS + kp) * td.iso_stride + mass_cap;
        tip_min = hh.agb_tip < tip_min ? hh.agb_tip : tip_min;
    }
    const double *par = td.cand_par + cb * B9_NPARAM;
    const double mod = par[B9_P_MOD], av = par[B9_P_ABS], lam = NPOPS == 2 ? par[B9_P_LAMBDA] : 1.0;
    const double log_lam = NPOPS == 2 ? wave_uniform(log(lam)) : 0.0, log_1ml = NPOPS == 2 ? wave_uniform(log1p(-lam)) : 0.0;
    __syncthreads();                                         // the LDS mass columns
    double *const prow = td.partial + (((size_t)td.set * W + w) * NN + n) * td.part_stride;
    hot_groups<NFP, NPOPS, true>(pk, st, seq, iso, valid, tip_min, mod, av, log_lam, log_1ml, i, m1, q, ea, nullptr,
                           [&](int c, int k, double tot) { prow[c * 4 + k] = tot; });
}

// ---- heavy role: the stars above a node's AGB tip (k_star_like's heavy role on the selected candidate) -----------------
template <int NFP, int NPOPS>
__device__ __forceinline__ void tree_heavy(const DevPack &pk, const DevStars &st, const TreeDev &td, int b, double *smem)
{
    const int W = td.n_walkers, NN = (1 << td.depth) - 1;
    const int part = b % td.heavy_parts, v = b / td.heavy_parts, w = v / NN, n = v - w * NN;
    if (tree_level(n) > td.levels) return;
    __shared__ int s_ho;
    const bool first_wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) == 0;
    if (first_wave) {
        const int o = tree_decide(td, w).outcome;
        if (threadIdx.x == 0) s_ho = o;
    }
    __syncthreads();
    const size_t cb = tree_cand(td, td.set, w, s_ho, n);
    const IsoHdr *const h1[1] = {td.cand_hdr + cb * NPOPS};
    const double *const i1[1] = {td.cand_iso + cb * NPOPS * td.iso_stride}, *const p1[1] = {td.cand_par + cb * B9_NPARAM};
    double *const o1[1] = {td.partial + (((size_t)td.set * W + w) * NN + n) * td.part_stride + (size_t)td.n_groups * 4 + part};
    heavy_stars<NFP, NPOPS, 1>(pk, st, h1, i1, td.iso_stride, td.mass_cap, p1, [] { return 0; }, 0, part, td.heavy_parts, o1, nullptr, smem);
}

// ---- proposal bookkeeping shared by the writer and the derivation role ---------------------------------------------------
// Normals of `n_steps` consecutive steps starting at global step `step0` into s_z[step][12], by the lanes of wave 3
// (Philox + Box-Muller: they depend on (seed, step, walker) only).
__device__ __forceinline__ void tree_draw_z(const TreeDev &td, int w, unsigned long long step0, int n_steps, double (*s_z)[12])
{
    const int j = (int)threadIdx.x - 192, n_pairs = (td.d + 1) >> 1;
    if (j >= 0 && j < n_steps * n_pairs) {
        const int si = j / n_pairs, pj = j - si * n_pairs;
        const unsigned long long sn = step0 + si;
        unsigned r[4];
        philox4x32((unsigned)sn, (unsigned)(sn >> 32), (unsigned)td.walker_ids[w], (unsigned)pj, td.k0, td.k1, r);
        const double u1 = u01(r[0], r[1]), u2 = u01(r[2], r[3]);
        const double rad = sqrt(-2.0 * log(u1)), ang = 2.0 * M_PI * u2;
        s_z[si][2 * pj] = rad * cos(ang);
        s_z[si][2 * pj + 1] = rad * sin(ang);
    }
}

// ---- table role (prologue launch, one workgroup per walker): the block's step table (TreeDev::step_tab) ---------------------
__device__ __forceinline__ void tree_table(const TreeDev &td, int w)
{
    const int tid = threadIdx.x, d = td.d, n_pairs = (d + 1) >> 1;
    __shared__ double s_zc[64][12];
    double *tab = td.step_tab + (size_t)w * td.tab_steps * B9_TREE_TAB_ROW;
    const int wid = td.walker_ids[w];
    for (int s0 = 0; s0 < td.tab_steps; s0 += 64) {
        const int ns = td.tab_steps - s0 < 64 ? td.tab_steps - s0 : 64;
        __syncthreads();
        for (int e = tid; e < ns * (n_pairs + 1); e += 256) {
            const int si = e / (n_pairs + 1), pj = e - si * (n_pairs + 1);
            const unsigned long long sn = td.block_step0 + (unsigned)(s0 + si);
            unsigned r[4];
            philox4x32((unsigned)sn, (unsigned)(sn >> 32), (unsigned)wid, (unsigned)pj, td.k0, td.k1, r);
            if (pj < n_pairs) {
                const double u1 = u01(r[0], r[1]), u2 = u01(r[2], r[3]);
                const double rad = sqrt(-2.0 * log(u1)), ang = 2.0 * M_PI * u2;
                s_zc[si][2 * pj] = rad * cos(ang);
                s_zc[si][2 * pj + 1] = rad * sin(ang);
            } else tab[(size_t)(s0 + si) * B9_TREE_TAB_ROW + 11] = log(u01(r[0], r[1]));     // draw index n_pairs: the accept test's u
        }
        __syncthreads();
        for (int e = tid; e < ns * d; e += 256) {
            const int si = e / d, t = e - si * d;
            double delta = 0.0;
            for (int j = 0; j < d; ++j) delta = delta + td.chol[t * d + j] * s_zc[si][j];      // j ascending, plain multiply-add
            tab[(size_t)(s0 + si) * B9_TREE_TAB_ROW + t] = delta;
        }
    }
}

// ---- derivation role: candidate (outcome o2 of THIS launch's tree, node n2 of the next tree) -------------------------------
// The state x after the previous launch's walk; the end state of outcome o2 = x plus the steps of this launch's tree that
// o2 accepts (each an `s_par[free[i]] += sum_j chol[i][j] z_j`, j ascending, plain multiply-add -- the sequential
// algorithm's own update, so the same bits); node n2's proposal = that plus the next tree's accepted ancestors' steps and
// its own.  Nothing here waits for memory after the first round trip until the isochrone tables.
__device__ __forceinline__ void tree_derive_prologue(const DevPack &pk, const TreeDev &td, int w, int o2, int n2, int pop, int part, int parts)
{
    const int tid = threadIdx.x, d = td.d, n_pops = td.n_pops, depth = td.depth;
    __shared__ double s_par[B9_NPARAM], s_z[2 * B9_TREE_MAX_DEPTH][12], s_delta[2 * B9_TREE_MAX_DEPTH][12];
    __shared__ int s_last;
    const bool first_wave = __builtin_amdgcn_readfirstlane(tid >> 6) == 0;
    const AxisRegs axr = preload_axis(pk);                 // first round trip, needs no parameter
    const double *in = tree_state_in(td, w);
    // the old state and every node's proposal of the previous launch's tree: which of them is x, the walk says
    double cand_v[B9_TREE_MAX_NODES + 1];
    {
        const int NN = (1 << depth) - 1;
        cand_v[0] = tid < B9_NPARAM ? in[B9_TS_CUR + tid] : 0.0;
#pragma unroll
        for (int n = 0; n < B9_TREE_MAX_NODES; ++n) cand_v[n + 1] = (tid < B9_NPARAM && n < NN) ? in[B9_TS_PROP + 12 * n + tid] : 0.0;
    }
    const bool prologue = td.derive_mode == 2;
    double crow[11];
#pragma unroll
    for (int j = 0; j < 11; ++j) crow[j] = (prologue && tid < d && j < d) ? td.chol[tid * d + j] : 0.0;
    const int fidx = tid < d ? td.free_idx[tid] : 0;
    // the steps' increments: table words (K launches; requested with everything else of the first round trip)
    double dtab[2 * B9_TREE_MAX_DEPTH];
    {
        const double *tab = td.step_tab + (size_t)w * td.tab_steps * B9_TREE_TAB_ROW;
        const long long r_cur = (long long)(td.step - td.block_step0), r_next = (long long)(td.next_step - td.block_step0);
#pragma unroll
        for (int i = 0; i < B9_TREE_MAX_DEPTH; ++i) {
            dtab[i] = (!prologue && tid < d && i < depth) ? tab[(size_t)(r_cur + i) * B9_TREE_TAB_ROW + tid] : 0.0;
            dtab[B9_TREE_MAX_DEPTH + i] = (!prologue && tid < d && i < depth) ? tab[(size_t)(r_next + i) * B9_TREE_TAB_ROW + tid] : 0.0;
        }
    }
    // the prologue alone draws its tree's normals itself (wave 3): the block's table is being written by this same launch
    if (prologue) tree_draw_z(td, w, td.next_step, depth, s_z + B9_TREE_MAX_DEPTH);
    if (first_wave) {
        const int last = tree_decide(td, w).last;
        if (tid == 0) s_last = last;
    }
    __syncthreads();
    {
        const int last = s_last;
        double x = cand_v[0];
#pragma unroll
        for (int n = 0; n < B9_TREE_MAX_NODES; ++n) x = (last == n) ? +cand_v[n + 1] : x;
        if (tid < B9_NPARAM) s_par[tid] = x;
    }
    // delta of every step: the table's (K launches), or sum_j chol[i][j] z_j formed here (prologue) -- the same operations
#pragma unroll
    for (int si = 0; si < 2 * B9_TREE_MAX_DEPTH; ++si) {
        double delta = dtab[si];
        if (prologue && si >= B9_TREE_MAX_DEPTH && si - B9_TREE_MAX_DEPTH < depth) {
            delta = 0.0;
#pragma unroll
            for (int j = 0; j < 11; ++j) if (j < d) delta = delta + crow[j] * s_z[si][j];
        }
        if (tid < d) s_delta[si][tid] = delta;
    }
    __syncthreads();
    if (tid < d) {
        double v = s_par[fidx];                              // (each thread owns one sampled parameter: sequential adds, no hazard)
        if (!prologue)
            for (int i = 1; i <= depth; ++i) if ((o2 >> (depth - i)) & 1) v += s_delta[i - 1][tid];
        const int lv = tree_level(n2), p2 = n2 + 1 - (1 << (lv - 1));
        for (int i = 1; i < lv; ++i) if ((p2 >> (lv - 1 - i)) & 1) v += s_delta[B9_TREE_MAX_DEPTH + i - 1][tid];
        v += s_delta[B9_TREE_MAX_DEPTH + lv - 1][tid];
        s_par[fidx] = v;
    }
    __syncthreads();
    const size_t cb = tree_cand(td, td.set ^ 1, w, o2, n2);
    if (pop == 0 && part == 0 && tid < B9_NPARAM) td.cand_par[cb * B9_NPARAM + tid] = s_par[tid];
    derive_iso_block(pk, s_par, pop, (int)(cb * n_pops + pop), td.cand_hdr, td.cand_iso, td.iso_stride, td.mass_cap, part, parts, axr);
}

// ---- writer: the walk's result -- new state, chain rows of the previous launch's steps, and what the NEXT launch's walk
// needs about the tree this launch evaluates (every node's proposal and log-prior, every level's log u) -----------------
__device__ __forceinline__ void tree_writer(const TreeDev &td, const DevPriors &pr, int w, double *s_state_out /* LDS [14] or null */)
{
    const int tid = threadIdx.x, d = td.d, W = td.n_walkers, n_pops = td.n_pops, depth = td.depth, NN = (1 << depth) - 1;
    __shared__ double s_lvl[B9_TREE_MAX_DEPTH + 1][B9_NPARAM];       // state after level j (0: before the walk)
    __shared__ double s_prop[B9_TREE_MAX_NODES][B9_NPARAM];
    __shared__ TreeWalk s_tw;
    const bool first_wave = __builtin_amdgcn_readfirstlane(tid >> 6) == 0;
    const double *in = tree_state_in(td, w);
    double cand_v[B9_TREE_MAX_NODES + 1];
    cand_v[0] = tid < B9_NPARAM ? in[B9_TS_CUR + tid] : 0.0;
#pragma unroll
    for (int n = 0; n < B9_TREE_MAX_NODES; ++n) cand_v[n + 1] = (tid < B9_NPARAM && n < NN) ? in[B9_TS_PROP + 12 * n + tid] : 0.0;
    const double nacc_in = in[B9_TS_NACC];
    const int fidx = tid < d ? td.free_idx[tid] : 0;
    if (first_wave) {
        const TreeWalk tw0 = tree_decide(td, w);
        if (tid == 0) s_tw = tw0;
    }
    __syncthreads();
    const int tw_outcome = s_tw.outcome, tw_n_acc = s_tw.n_acc;
    const double tw_lp = s_tw.lp;
    if (tid < B9_NPARAM) {
        s_lvl[0][tid] = cand_v[0];
#pragma unroll
        for (int j = 0; j < B9_TREE_MAX_DEPTH; ++j) {
            const int la = s_tw.last_at[j];
            double x = cand_v[0];
#pragma unroll
            for (int n = 0; n < B9_TREE_MAX_NODES; ++n) x = (la == n) ? +cand_v[n + 1] : x;
            s_lvl[j + 1][tid] = x;
        }
    }
    // the tree this launch evaluates: candidate set of the walk's outcome (full-depth launches precede every launch but a
    // block's first, whose set is outcome slot 0)
    const int o = td.levels_prev > 0 ? tw_outcome : 0;
    const size_t cb = tree_cand(td, td.set, w, o, 0);
    if (td.derive_mode != 0 || td.levels > 0) {
        for (int e = tid; e < NN * B9_NPARAM; e += 256) s_prop[e / B9_NPARAM][e % B9_NPARAM] = td.cand_par[cb * B9_NPARAM + e];
    }
    __syncthreads();
    const int lv = td.levels_prev;
    double *out = td.state + ((size_t)td.set * W + w) * B9_TREE_STATE_STRIDE;
    if (tid < B9_NPARAM) out[B9_TS_CUR + tid] = s_lvl[lv][tid];
    if (td.levels > 0)
        for (int e = tid; e < NN * B9_NPARAM; e += 256) out[B9_TS_PROP + e] = s_prop[e / B9_NPARAM][e % B9_NPARAM];
    if (tid == 0) { out[B9_TS_LP] = tw_lp; out[B9_TS_NACC] = nacc_in + (double)tw_n_acc; }
    if (td.levels > 0) {
        if (tid >= 64 && tid < 64 + NN) {                    // log-prior of every node's proposal
            const int n = tid - 64;
            bool pv = true;
            for (int k = 0; k < n_pops; ++k) pv = pv && td.cand_hdr[(cb + n) * n_pops + k].valid;
            out[B9_TS_LPRIOR + n] = pv ? log_prior_cluster(pr, s_prop[n], n_pops) : NEG_INF;
        }
        if (tid >= 128 && tid < 128 + depth)                 // log u of level j's accept test: the block's step table
            out[B9_TS_LOGU + (tid - 128)] = td.step_tab[((size_t)w * td.tab_steps + (size_t)(td.step - td.block_step0) + (tid - 128)) * B9_TREE_TAB_ROW + 11];
    }
    // chain rows of the previous launch's steps: the state after each of its levels
    for (int j = 1; j <= lv; ++j) {
        if (td.samples && tid < d) td.samples[((size_t)(td.row + j - 1) * W + w) * d + tid] = s_lvl[j][fidx];
        if (td.lps && tid == 0) td.lps[(size_t)(td.row + j - 1) * W + w] = s_tw.lp_at[j - 1];
    }
    if (s_state_out) {
        if (tid < B9_NPARAM) s_state_out[tid] = s_lvl[lv][tid];
        if (tid == 0) { s_state_out[B9_NPARAM] = tw_lp; s_state_out[B9_NPARAM + 1] = nacc_in + (double)tw_n_acc; }
    }
}

// The K launches' derivation role.  The first wave takes the decision and forms the candidate's parameters (registers and
// one LDS row; no barrier inside); the other three waves meanwhile run the derivation AHEAD for the grid cell of the
// previous state (a candidate is that state plus at most 2 x depth steps: almost always the same cell) -- corner rows,
// then the corner values of the workgroup's share, held in registers (b9_derive.hip.h) -- so that when the parameters
// arrive only the interpolation and the stores remain.  A candidate in another cell repeats the two round trips for
// its own cell: same values either way.  Measured on C1 (tools/gantt_step.py): the role ended 4.8 us after its decision
// (parameters 0.8, corner rows 1.1, values + stores 1.6 + barriers); now the decision's own 5.2 us hide the two trips.
#define B9_TREE_KV 3          // output items per thread whose corner values are kept in registers at a time
__device__ __forceinline__ void tree_derive(const DevPack &pk, const TreeDev &td, int w, int o2, int n2, int pop, int part, int parts)
{
    if (td.derive_mode == 2) { tree_derive_prologue(pk, td, w, o2, n2, pop, part, parts); return; }
    const int tid = threadIdx.x, d = td.d, n_pops = td.n_pops, depth = td.depth;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    __shared__ double s_par[B9_NPARAM];
    const double *in = tree_state_in(td, w);
    const size_t cb = tree_cand(td, td.set ^ 1, w, o2, n2);
    if (wave == 0) {
        // the old state and every node's proposal of the previous launch's tree: which of them is x, the walk says
        double cand_v[B9_TREE_MAX_NODES + 1];
        const int NN = (1 << depth) - 1;
        cand_v[0] = tid < B9_NPARAM ? in[B9_TS_CUR + tid] : 0.0;
#pragma unroll
        for (int n = 0; n < B9_TREE_MAX_NODES; ++n) cand_v[n + 1] = (tid < B9_NPARAM && n < NN) ? in[B9_TS_PROP + 12 * n + tid] : 0.0;
        const int fidx = tid < d ? td.free_idx[tid] : 0;
        // the steps' increments: table words, requested with everything else of the first round trip
        double dtab[2 * B9_TREE_MAX_DEPTH];
        {
            const double *tab = td.step_tab + (size_t)w * td.tab_steps * B9_TREE_TAB_ROW;
            const long long r_cur = (long long)(td.step - td.block_step0), r_next = (long long)(td.next_step - td.block_step0);
#pragma unroll
            for (int i = 0; i < B9_TREE_MAX_DEPTH; ++i) {
                dtab[i] = (tid < d && i < depth) ? tab[(size_t)(r_cur + i) * B9_TREE_TAB_ROW + tid] : 0.0;
                dtab[B9_TREE_MAX_DEPTH + i] = (tid < d && i < depth) ? tab[(size_t)(r_next + i) * B9_TREE_TAB_ROW + tid] : 0.0;
            }
        }
        const int last = tree_decide(td, w).last;
        double x = cand_v[0];
#pragma unroll
        for (int n = 0; n < B9_TREE_MAX_NODES; ++n) x = (last == n) ? +cand_v[n + 1] : x;
        if (tid < B9_NPARAM) s_par[tid] = x;
        __builtin_amdgcn_wave_barrier();                     // (one wave: its LDS accesses complete in program order)
        if (tid < d) {
            double v = s_par[fidx];                          // each lane owns one sampled parameter: sequential adds, no hazard
            const int lv = tree_level(n2), p2 = n2 + 1 - (1 << (lv - 1));
#pragma unroll
            for (int i = 1; i <= B9_TREE_MAX_DEPTH; ++i) if (i <= depth && ((o2 >> (depth - i)) & 1)) v += dtab[i - 1];
#pragma unroll
            for (int i = 1; i < B9_TREE_MAX_DEPTH; ++i) if (i < lv && ((p2 >> (lv - 1 - i)) & 1)) v += dtab[B9_TREE_MAX_DEPTH + i - 1];
#pragma unroll
            for (int i = 0; i < B9_TREE_MAX_DEPTH; ++i) if (i == lv - 1) v += dtab[B9_TREE_MAX_DEPTH + i];
            s_par[fidx] = v;
        }
        __syncthreads();
        if (pop == 0 && part == 0 && tid < B9_NPARAM) td.cand_par[cb * B9_NPARAM + tid] = s_par[tid];
        return;
    }
    // ---- waves 1..3: the derivation, ahead for the previous state's cell ----
    const int nfp = pk.nfp, mass_cap = td.mass_cap, wp = (int)(cb * n_pops + pop);
    const int first = part * 192 + (tid - 64), stride = parts * 192;
    AxisRegs ax[3];
    preload_axes3(pk, ax);
    const double g_age = in[B9_TS_CUR + B9_P_LOGAGE], g_feh = in[B9_TS_CUR + B9_P_FEH];
    const double g_y = in[B9_TS_CUR + (pop ? B9_P_Y2 : B9_P_Y)];
    GridCell cell = grid_cell(pk, ax, g_age, g_feh, g_y);
    CornerRegs cr = corner_rows(pk, cell);
    double v[B9_TREE_KV][8];
    {
        const int total = (cr.n >= 2 && cr.n <= mass_cap) ? cr.n * (nfp + 1) : 0;
        corner_values<B9_TREE_KV>(pk, cr, total, first, stride, v);
    }
    __syncthreads();                                         // the candidate's parameters (first wave)
    const double log_age = s_par[B9_P_LOGAGE], feh = s_par[B9_P_FEH], y = pop ? s_par[B9_P_Y2] : s_par[B9_P_Y];
    const GridCell own = grid_cell(pk, ax, log_age, feh, y);
    const bool same = own.i_age == cell.i_age && own.i_feh == cell.i_feh && own.i_y == cell.i_y;
    if (!same) { cell = own; cr = corner_rows(pk, cell); }
    const IsoHdr h = header_of(pk, cell, cr, log_age, feh, y, mass_cap);
    IsoHdr *hp = td.cand_hdr + wp;
    if (part == 0 && tid == 64) {
        // (agb_tip of a valid isochrone is stored by the thread that interpolates the last point's mass)
        hp->valid = h.valid; hp->first_eep = h.first_eep; hp->n = h.n; hp->i_feh = h.i_feh; hp->i_y = h.i_y; hp->i_age = h.i_age;
        hp->t_feh = h.t_feh; hp->t_y = h.t_y; hp->t_age = h.t_age;
        if (!h.valid) hp->agb_tip = 0.0;
    }
    if (!h.valid) return;
    const int total = h.n * (nfp + 1);
    double *omass = td.cand_iso + (size_t)wp * td.iso_stride;
    double *omags = omass + mass_cap;
    if (!same) corner_values<B9_TREE_KV>(pk, cr, total, first, stride, v);
    store_values<B9_TREE_KV>(pk, h, total, first, stride, v, omass, omags, &hp->agb_tip);
    for (int f = first + B9_TREE_KV * stride; f < total; f += B9_TREE_KV * stride) {      // (few derivation parts: more than KV items per thread)
        corner_values<B9_TREE_KV>(pk, cr, total, f, stride, v);
        store_values<B9_TREE_KV>(pk, h, total, f, stride, v, omass, omags, &hp->agb_tip);
    }
}

// Grid: [writers W][derivation][heavy][pad to 8][hot].  n_front = first hot workgroup id.
template <int NFP, int NPOPS>
__global__ __launch_bounds__(256, B9_K1_WAVES(NFP, NPOPS))
void k_mcmc_tree(DevPack pk, DevStars st, TreeDev td, DevPriors pr, int group_tiles, int n_front, int derive_parts)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    B9_GANTT_ENTER();
    WSTAMP_ON(td.derive_mode == 1 ? 1ull : 0ull);    // (the K launches only)
    WSTAMP(0);
    const int W = td.n_walkers, NN = (1 << td.depth) - 1, NO = td.derive_mode == 2 ? 1 : (1 << td.depth);
    int b = blockIdx.x, role = 3;                            // 0 hot, 1 heavy, 2 derivation, 3 padding, 4 writer (tools/gantt_step.py)
    const int n_writers = W;                                 // (the prologue has no tree to take a decision on: its "writers" write the block's step table)
    const int n_derive = td.derive_mode == 0 ? 0 : W * NO * NN * NPOPS * derive_parts;
    if (b >= n_front) { role = 0; tree_hot<NFP, NPOPS>(pk, st, td, b - n_front, group_tiles, smem); }
    else if (b < n_writers) { role = 4; if (td.derive_mode == 2) tree_table(td, b); else tree_writer(td, pr, b, nullptr); }
    else if (b - n_writers < n_derive) {          // b = (((w * NO + o2) * NN + n2) * NPOPS + pop) * parts + part
        role = 2;
        b -= n_writers;
        const int part = b % derive_parts; b /= derive_parts;
        const int pop = b % NPOPS; b /= NPOPS;
        const int n2 = b % NN; b /= NN;
        const int o2 = b % NO; b /= NO;
        tree_derive(pk, td, b, o2, n2, pop, part, derive_parts);
    } else if (td.levels > 0 && b - n_writers - n_derive < W * NN * td.heavy_parts) {
        role = 1;
        tree_heavy<NFP, NPOPS>(pk, st, td, b - n_writers - n_derive, smem);
    }
    B9_GANTT_EXIT(td.step / (unsigned)td.depth, role);
}

// the block's last walk: one workgroup per walker, writer role only (+ the block's summary rows, the host mirror)
__global__ __launch_bounds__(256) void k_tree_finish(TreeDev td, DevPriors pr)
{
    __shared__ double s_last[B9_NPARAM + 2];
    WSTAMP_ON(0ull);
    tree_writer(td, pr, blockIdx.x, s_last);
    __syncthreads();
    if (td.rows) {
        StepDev sd{};
        sd.d = td.d; sd.n_walkers = td.n_walkers; sd.n_steps = td.n_steps; sd.samples = td.samples; sd.free_idx = td.free_idx;
        sd.row_origin = td.row_origin; sd.rows = td.rows; sd.host_rows = td.host_rows;
        block_summary_row(sd, blockIdx.x, s_last);
    }
    if (td.host_state) {
        double *h = td.host_state + (size_t)blockIdx.x * B9_TREE_STATE_STRIDE;
        if (threadIdx.x < B9_NPARAM) h[B9_TS_CUR + threadIdx.x] = s_last[threadIdx.x];
        if (threadIdx.x == B9_NPARAM) { h[B9_TS_LP] = s_last[B9_NPARAM]; h[B9_TS_NACC] = s_last[B9_NPARAM + 1]; }
    }
}

// The block's opening (as k_mcmc_begin): the upload from the mapped host mirror, and the starting state -- the caller's,
// or the previous block's final state rows -- into BOTH parities' state rows (the prologue launch reads one, K(0) the other).
__global__ __launch_bounds__(256) void k_tree_begin(const double *__restrict__ host_up, double *__restrict__ dev, int up_words,
                                                    const double *__restrict__ prev_final, double *__restrict__ state, int n_walkers)
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
        const double *src = prev_final + (size_t)w * B9_TREE_STATE_STRIDE;
        for (int p = 0; p < 2; ++p) {
            double *dst = state + ((size_t)p * n_walkers + w) * B9_TREE_STATE_STRIDE;
            if (tid < B9_NPARAM) dst[B9_TS_CUR + tid] = src[B9_TS_CUR + tid];
            if (tid == B9_NPARAM) { dst[B9_TS_LP] = src[B9_TS_LP]; dst[B9_TS_NACC] = 0.0; }
        }
    }
}
