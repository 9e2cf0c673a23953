// b9_derive.hip.h -- k_derive_iso: the isochrone of (logAge, FeH, Y) by EEP-wise tri-linear interpolation.
// Part of the single translation unit b9_kernels.hip (included there, in this order); gfx950 only.
#pragma once

// ------------------------------------------------------------------------------------------
// k_derive_iso
// ------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------
struct Corners {
    long long off[8];      // point offset of EEP `lo` in each corner isochrone
    int ny;
    double t_age, t_y, t_feh;
};

template <bool MASS>
__device__ __forceinline__ double interp_corner(const DevPack &pk, const Corners &c, int e, int col)
{
    double vf[2];
#pragma unroll
    for (int df = 0; df < 2; ++df) {
        double vy[2] = {0.0, 0.0};
        for (int dy = 0; dy < c.ny; ++dy) {
            long long p0 = c.off[(df * 2 + dy) * 2 + 0] + e, p1 = c.off[(df * 2 + dy) * 2 + 1] + e;
            double a = MASS ? pk.mass[p0] : pk.mags[p0 * pk.nfp + col];
            double b = MASS ? pk.mass[p1] : pk.mags[p1 * pk.nfp + col];
            vy[dy] = lerp(a, b, c.t_age);
        }
        vf[df] = (c.ny == 2) ? lerp(vy[0], vy[1], c.t_y) : vy[0];
    }
    return lerp(vf[0], vf[1], c.t_feh);
}

// Largest i in [0, n-2] with ax[i] <= x, found by one wave in one step: lane l loads ax[l]
// (axes have <= 64 * B9_AXIS_CHUNKS entries) and the bracket is a popcount of the ballot.
// Equal to the oracle's bracket() for an ascending axis.
__device__ __forceinline__ int bracket_wave(const double *__restrict__ ax, int n, double x, int lane)
{
    int cnt = 0;
    for (int base = 0; base < n; base += 64) {
        const int j = base + lane;
        const bool le = (j < n) && (ax[j] <= x);
        cnt += __popcll(__ballot(le));
    }
    int i = cnt - 1;
    return i < 0 ? 0 : (i > n - 2 ? n - 2 : i);
}

// The three grid axes, one per wave (0: logAge, 1: FeH, 2: Y), preloaded into registers: lane l of
// the wave holds ax[l] and ax[l + 64].  Loading them needs no parameter, so k_derive_iso requests
// them at kernel entry, in the same round trip as everything else it reads first.
struct AxisRegs { double v0, v1; int n; };

__device__ __forceinline__ AxisRegs preload_axis(const DevPack &pk)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const double *ax = wave == 0 ? pk.log_age : (wave == 1 ? pk.feh : pk.y);
    AxisRegs a;
    a.n = wave == 0 ? pk.n_age : (wave == 1 ? pk.n_feh : (wave == 2 ? pk.n_y : 0));
    a.v0 = lane < a.n ? ax[lane] : __builtin_inf();
    a.v1 = lane + 64 < a.n ? ax[lane + 64] : __builtin_inf();
    return a;
}

// bracket of x on a preloaded axis (n <= 128), else on the axis in memory
__device__ __forceinline__ int bracket_regs(const AxisRegs &a, const double *__restrict__ ax, double x, int lane)
{
    if (a.n > 128) return bracket_wave(ax, a.n, x, lane);
    const int cnt = __popcll(__ballot(a.v0 <= x)) + __popcll(__ballot(a.v1 <= x));
    const int i = cnt - 1;
    return i < 0 ? 0 : (i > a.n - 2 ? a.n - 2 : i);
}

// Derives the isochrone of (walker w, population pop) from parameter row `par` (any address
// space).  All threads of the workgroup call it; workgroup `part` of `parts` produces its share of
// the output values (one value per thread and iteration) and part 0 publishes the header.
// Three dependent round trips: {parameters, axes} -> corner index rows -> table values.
__device__ __forceinline__ void derive_iso_block(const DevPack &pk, const double *par, int pop, int wp,
                                                 IsoHdr *__restrict__ hdr, double *__restrict__ iso_data,
                                                 long long iso_stride, int mass_cap, int part, int parts,
                                                 const AxisRegs &axr)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthreads = blockDim.x;
    __shared__ IsoHdr sh;
    __shared__ Corners sc;
    __shared__ int s_br[3];
    const double log_age = par[B9_P_LOGAGE], feh = par[B9_P_FEH];
    const double y = pop ? par[B9_P_Y2] : par[B9_P_Y];
    __syncthreads();                                     // sh / sc / s_br may still be in use (previous call)
    // three waves bracket the three axes concurrently
    if (wave == 0) { int i = bracket_regs(axr, pk.log_age, log_age, lane); if (lane == 0) s_br[0] = i; }
    if (wave == 1) { int i = bracket_regs(axr, pk.feh, feh, lane); if (lane == 0) s_br[1] = i; }
    if (wave == 2) { int i = pk.n_y > 1 ? bracket_regs(axr, pk.y, y, lane) : 0; if (lane == 0) s_br[2] = i; }
    __syncthreads();
    if (wave == 0) {
        // lanes 0..7: one corner isochrone each
        const int ny = pk.n_y > 1 ? 2 : 1;
        const int i_age = s_br[0], i_feh = s_br[1], i_y = s_br[2];
        const int df = (lane >> 2) & 1, dy = (lane >> 1) & 1, da = lane & 1;
        const int dyc = dy < ny ? dy : 0;
        const int kk = ((i_feh + df) * pk.n_y + (i_y + dyc)) * pk.n_age + i_age + da;
        int f0 = -2147483647, f1 = 2147483647;
        long long off = 0;
        double ax0 = 0.0;
        if (lane < 8) { f0 = pk.first[kk]; f1 = f0 + pk.cnt[kk]; off = pk.off[kk]; }
        // lanes 8..13 fetch the axis values the interpolation weights need (same round trip)
        if (lane == 8)  ax0 = pk.log_age[i_age];
        if (lane == 9)  ax0 = pk.log_age[i_age + 1];
        if (lane == 10) ax0 = pk.feh[i_feh];
        if (lane == 11) ax0 = pk.feh[i_feh + 1];
        if (lane == 12) ax0 = pk.y[i_y];
        if (lane == 13) ax0 = pk.y[ny == 2 ? i_y + 1 : i_y];
        int lo = f0, hi = f1;
#pragma unroll
        for (int o = 4; o > 0; o >>= 1) {
            int l2 = __shfl_xor(lo, o, 64), h2 = __shfl_xor(hi, o, 64);
            lo = l2 > lo ? l2 : lo;
            hi = h2 < hi ? h2 : hi;
        }
        lo = __shfl(lo, 0, 64); hi = __shfl(hi, 0, 64);
        const double a_lo = __shfl(ax0, 8, 64), a_hi = __shfl(ax0, 9, 64);
        const double f_lo = __shfl(ax0, 10, 64), f_hi = __shfl(ax0, 11, 64);
        const double y_lo = __shfl(ax0, 12, 64), y_hi = __shfl(ax0, 13, 64);
        const double a_min = pk.log_age[0], a_max = pk.log_age[pk.n_age - 1];
        const double f_min = pk.feh[0], f_max = pk.feh[pk.n_feh - 1];
        const double y_min = pk.y[0], y_max = pk.y[pk.n_y - 1];
        if (lane < 8) sc.off[lane] = off + (lo - f0);
        if (lane == 0) {
            IsoHdr h;
            h.valid = 0; h.first_eep = 0; h.n = 0; h.i_feh = i_feh; h.i_y = i_y; h.i_age = i_age;
            h.agb_tip = 0.0; h.t_feh = h.t_y = h.t_age = 0.0;
            bool ok = (log_age >= a_min && log_age <= a_max) && (feh >= f_min && feh <= f_max) &&
                      pk.n_age >= 2 && pk.n_feh >= 2;
            if (pk.n_y > 1) ok = ok && (y >= y_min && y <= y_max);
            const int n = hi - lo;
            if (ok && n >= 2 && n <= mass_cap) {
                h.t_age = (log_age - a_lo) / (a_hi - a_lo);
                h.t_feh = (feh - f_lo) / (f_hi - f_lo);
                if (ny == 2) h.t_y = (y - y_lo) / (y_hi - y_lo);
                h.first_eep = lo; h.n = n; h.valid = 1;
            }
            sc.ny = ny; sc.t_age = h.t_age; sc.t_y = h.t_y; sc.t_feh = h.t_feh;
            sh = h;
        }
    }
    __syncthreads();
    if (!sh.valid) { if (tid == 0 && part == 0) hdr[wp] = sh; return; }
    const int n = sh.n, nfp = pk.nfp;
    double *omass = iso_data + (size_t)wp * iso_stride;
    double *omags = omass + mass_cap;
    // the thread that starts first also interpolates the last point's mass = the AGB-tip mass
    if (part == 0 && tid == 0) {
        IsoHdr h = sh;
        h.agb_tip = interp_corner<true>(pk, sc, n - 1, 0);
        hdr[wp] = h;
    }
    const int total = n * (nfp + 1);
    for (int idx = part * nthreads + tid; idx < total; idx += parts * nthreads) {
        const int e = idx / (nfp + 1), c = idx - e * (nfp + 1);
        if (c == nfp) omass[e] = interp_corner<true>(pk, sc, e, 0);
        else omags[(size_t)e * nfp + c] = (c < pk.nf) ? interp_corner<false>(pk, sc, e, c) : 0.0;
    }
}

// ------------------------------------------------------------------------------------------
// The same derivation held in ONE WAVE's registers (the tree launch's derivation role, b9_mcmc_tree.hip.h): a wave
// brackets all three axes itself, reads the corner rows itself and keeps the corner values of its share of the output
// in registers, so the whole chain {axes} -> corner rows -> table values needs no workgroup barrier and can be run
// AHEAD, for a guessed grid cell, while another wave still takes the step's decision.  Values and operation order are
// derive_iso_block's (interp_corner): identical bits.
// ------------------------------------------------------------------------------------------
struct GridCell { int i_age, i_feh, i_y; };

__device__ __forceinline__ void preload_axes3(const DevPack &pk, AxisRegs (&a)[3])
{
    const int lane = threadIdx.x & 63;
    a[0].n = pk.n_age; a[1].n = pk.n_feh; a[2].n = pk.n_y;
    a[0].v0 = lane < pk.n_age ? pk.log_age[lane] : __builtin_inf();
    a[0].v1 = lane + 64 < pk.n_age ? pk.log_age[lane + 64] : __builtin_inf();
    a[1].v0 = lane < pk.n_feh ? pk.feh[lane] : __builtin_inf();
    a[1].v1 = lane + 64 < pk.n_feh ? pk.feh[lane + 64] : __builtin_inf();
    a[2].v0 = lane < pk.n_y ? pk.y[lane] : __builtin_inf();
    a[2].v1 = lane + 64 < pk.n_y ? pk.y[lane + 64] : __builtin_inf();
}

__device__ __forceinline__ GridCell grid_cell(const DevPack &pk, const AxisRegs (&a)[3], double log_age, double feh, double y)
{
    const int lane = threadIdx.x & 63;
    GridCell g;
    g.i_age = bracket_regs(a[0], pk.log_age, log_age, lane);
    g.i_feh = bracket_regs(a[1], pk.feh, feh, lane);
    g.i_y = pk.n_y > 1 ? bracket_regs(a[2], pk.y, y, lane) : 0;
    return g;
}

__device__ __forceinline__ double wave_bcast(double v, int src)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src), hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ long long wave_bcast(long long v, int src)
{
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v & 0xffffffffll), src);
    const int hi = __builtin_amdgcn_readlane((int)(v >> 32), src);
    return ((long long)hi << 32) | (long long)lo;
}

// a grid cell's corner rows and axis values, wave-uniform (what derive_iso_block keeps in `sc` / `sh`)
struct CornerRegs {
    long long off[8];       // point offset of EEP `lo` in each corner isochrone, index (dfeh * 2 + dy) * 2 + dage
    int lo, n;              // first common EEP, common points (hi - lo: < 2 or > mass_cap = no isochrone)
    double a_lo, a_hi, f_lo, f_hi, y_lo, y_hi, a_min, a_max, f_min, f_max, y_min, y_max;
};

__device__ __forceinline__ CornerRegs corner_rows(const DevPack &pk, const GridCell &g)
{
    const int lane = threadIdx.x & 63;
    const int ny = pk.n_y > 1 ? 2 : 1;
    const int df = (lane >> 2) & 1, dy = (lane >> 1) & 1, da = lane & 1;
    const int dyc = dy < ny ? dy : 0;
    const int kk = ((g.i_feh + df) * pk.n_y + (g.i_y + dyc)) * pk.n_age + g.i_age + da;
    int f0 = -2147483647, f1 = 2147483647;
    long long off = 0;
    double ax0 = 0.0;
    if (lane < 8) { f0 = pk.first[kk]; f1 = f0 + pk.cnt[kk]; off = pk.off[kk]; }
    // lanes 8..19: the cell's axis values and the axes' ends (same round trip)
    if (lane >= 8 && lane < 20) {
        const int j = lane - 8, ax = (j >> 1) % 3, up = j & 1;
        const double *base = ax == 0 ? pk.log_age : (ax == 1 ? pk.feh : pk.y);
        const int n_ax = ax == 0 ? pk.n_age : (ax == 1 ? pk.n_feh : pk.n_y);
        const int i_ax = ax == 0 ? g.i_age : (ax == 1 ? g.i_feh : g.i_y);
        const int step = (ax == 2 && ny == 1) ? 0 : 1;
        ax0 = base[j < 6 ? i_ax + (up ? step : 0) : (up ? n_ax - 1 : 0)];
    }
    int lo = f0, hi = f1;
#pragma unroll
    for (int o = 4; o > 0; o >>= 1) {
        const int l2 = __shfl_xor(lo, o, 64), h2 = __shfl_xor(hi, o, 64);
        lo = l2 > lo ? l2 : lo;
        hi = h2 < hi ? h2 : hi;
    }
    CornerRegs c;
    c.lo = __builtin_amdgcn_readlane(lo, 0);
    c.n = __builtin_amdgcn_readlane(hi, 0) - c.lo;
    const long long adj = off + (c.lo - f0);
#pragma unroll
    for (int j = 0; j < 8; ++j) c.off[j] = wave_bcast(adj, j);
    c.a_lo = wave_bcast(ax0, 8);  c.a_hi = wave_bcast(ax0, 9);
    c.f_lo = wave_bcast(ax0, 10); c.f_hi = wave_bcast(ax0, 11);
    c.y_lo = wave_bcast(ax0, 12); c.y_hi = wave_bcast(ax0, 13);
    c.a_min = wave_bcast(ax0, 14); c.a_max = wave_bcast(ax0, 15);
    c.f_min = wave_bcast(ax0, 16); c.f_max = wave_bcast(ax0, 17);
    c.y_min = wave_bcast(ax0, 18); c.y_max = wave_bcast(ax0, 19);
    return c;
}

// the isochrone's header for parameters (log_age, feh, y) in cell g (derive_iso_block's lane-0 block; agb_tip is left 0)
__device__ __forceinline__ IsoHdr header_of(const DevPack &pk, const GridCell &g, const CornerRegs &c, double log_age, double feh,
                                            double y, int mass_cap)
{
    IsoHdr h;
    h.valid = 0; h.first_eep = 0; h.n = 0; h.i_feh = g.i_feh; h.i_y = g.i_y; h.i_age = g.i_age;
    h.agb_tip = 0.0; h.t_feh = h.t_y = h.t_age = 0.0;
    bool ok = (log_age >= c.a_min && log_age <= c.a_max) && (feh >= c.f_min && feh <= c.f_max) && pk.n_age >= 2 && pk.n_feh >= 2;
    if (pk.n_y > 1) ok = ok && (y >= c.y_min && y <= c.y_max);
    if (ok && c.n >= 2 && c.n <= mass_cap) {
        h.t_age = (log_age - c.a_lo) / (c.a_hi - c.a_lo);
        h.t_feh = (feh - c.f_lo) / (c.f_hi - c.f_lo);
        if (pk.n_y > 1) h.t_y = (y - c.y_lo) / (c.y_hi - c.y_lo);
        h.first_eep = c.lo; h.n = c.n; h.valid = 1;
    }
    return h;
}

// Corner values of KV output items of this thread: item = first + k * stride (< total), item -> (EEP, column) as in
// derive_iso_block; column nfp = the mass.
template <int KV>
__device__ __forceinline__ void corner_values(const DevPack &pk, const CornerRegs &c, int total, int first, int stride, double (&v)[KV][8])
{
    const int nfp = pk.nfp, ny = pk.n_y > 1 ? 2 : 1;
#pragma unroll
    for (int k = 0; k < KV; ++k) {
        const int idx = first + k * stride;
        const bool live = idx < total;
        const int e = idx / (nfp + 1), col = idx - e * (nfp + 1);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            v[k][j] = 0.0;
            if (((j >> 1) & 1) && ny == 1) continue;
            const long long p = c.off[j] + e;
            if (live && col == nfp) v[k][j] = pk.mass[p];
            else if (live && col < pk.nf) v[k][j] = pk.mags[p * nfp + col];
        }
    }
}

// ... and their interpolation + store (interp_corner's order: age, then Y, then FeH)
template <int KV>
__device__ __forceinline__ void store_values(const DevPack &pk, const IsoHdr &h, int total, int first, int stride, const double (&v)[KV][8],
                                             double *__restrict__ omass, double *__restrict__ omags, double *__restrict__ agb_tip)
{
    const int nfp = pk.nfp, ny = pk.n_y > 1 ? 2 : 1;
#pragma unroll
    for (int k = 0; k < KV; ++k) {
        const int idx = first + k * stride;
        if (idx >= total) continue;
        const int e = idx / (nfp + 1), col = idx - e * (nfp + 1);
        double vf[2];
#pragma unroll
        for (int df = 0; df < 2; ++df) {
            const double v0 = lerp(v[k][df * 4 + 0], v[k][df * 4 + 1], h.t_age);
            const double v1 = lerp(v[k][df * 4 + 2], v[k][df * 4 + 3], h.t_age);
            vf[df] = (ny == 2) ? lerp(v0, v1, h.t_y) : v0;
        }
        const double out = lerp(vf[0], vf[1], h.t_feh);
        if (col == nfp) { omass[e] = out; if (idx == total - 1) *agb_tip = out; }      // the last point's mass = the AGB-tip mass
        else omags[(size_t)e * nfp + col] = (col < pk.nf) ? out : 0.0;
    }
}

// k_derive_iso: grid = (walkers * pops, parts).  Every workgroup of a row re-derives the (cheap)
// header and then produces its share of the values, so the table reads of one isochrone are a
// single round trip spread over ~15 workgroups.
//
// Device-resident sampler (mc.enabled): the launch of step t first finishes step t-1 when
// mc.has_prev -- each workgroup re-sums walker w's partials of the star kernel's previous launch,
// adds the prior of the previous proposal (params_prev) and accepts or rejects it (identical bits
// in every workgroup; workgroup (pop 0, part 0) stores the new state and the chain record) -- and
// then draws step t's proposal from that state, publishes it to `params`, and derives its
// isochrone(s).  One MCMC step = this launch + the star-likelihood launch.
__global__ __launch_bounds__(256) void k_derive_iso(DevPack pk, double *__restrict__ params,
                                                     int n_pops, IsoHdr *__restrict__ hdr,
                                                     double *__restrict__ iso_data, long long iso_stride,
                                                     int mass_cap, McmcDev mc, DevPriors pr,
                                                     const double *__restrict__ partial_prev, int n_partial,
                                                     long long partial_stride,
                                                     const IsoHdr *__restrict__ hdr_prev,
                                                     const double *__restrict__ params_prev)
{
    const int wp = blockIdx.x, w = wp / n_pops, pop = wp % n_pops;
    const double *par = params + (size_t)w * B9_NPARAM;
    __shared__ double s_par[B9_NPARAM], s_z[12], s_cur[B9_NPARAM], s_prop[B9_NPARAM], s_red[4];
    const AxisRegs axr = preload_axis(pk);                 // first round trip, needs no parameter
    if (mc.enabled) {
        // Everything the prologue reads is requested NOW, in one round trip: the walker's current
        // row and log-posterior, the previous proposal, this thread's row of the proposal factor,
        // and (inside finish_logpost) the partials.  Nothing below waits on memory again until the
        // isochrone tables.
        const int tid = threadIdx.x, d = mc.d;
        const bool writer = (blockIdx.y == 0 && pop == 0);
        const size_t st_in = (size_t)mc.pin * mc.n_walkers + w;
        const double cur_v = tid < B9_NPARAM ? mc.cur[st_in * B9_NPARAM + tid] : 0.0;
        const double prop_v = (mc.has_prev && tid < B9_NPARAM) ? params_prev[(size_t)w * B9_NPARAM + tid] : 0.0;
        const double lp_cur = mc.lp_cur[st_in];
        double crow[11];
#pragma unroll
        for (int j = 0; j < 11; ++j) crow[j] = (tid < d && j < d) ? mc.chol[tid * d + j] : 0.0;
        const int fidx = tid < d ? mc.free_idx[tid] : 0;
        draw_z(mc, w, mc.step, 192, s_z);                  // wave 3: this step's normals, independent of the state
        if (tid < B9_NPARAM) { s_prop[tid] = prop_v; s_cur[tid] = cur_v; }
        __syncthreads();
        if (mc.has_prev) {
            const double lp_prop = finish_logpost(hdr_prev, partial_prev + (size_t)w * partial_stride, n_partial,
                                                  s_prop, pr, n_pops, w, s_red);
            // Metropolis accept/reject of step t-1 (u: draw index n_pairs of that step's Philox stream)
            unsigned r[4];
            const unsigned long long sp = mc.step - 1;
            philox4x32((unsigned)sp, (unsigned)(sp >> 32), (unsigned)mc.walker_ids[w], (unsigned)((d + 1) >> 1), mc.k0, mc.k1, r);
            const bool ok = isfinite(lp_prop) && (log(u01(r[0], r[1])) < lp_prop - lp_cur);
            const double lp_new = ok ? lp_prop : lp_cur;
            if (tid < B9_NPARAM && ok) s_cur[tid] = prop_v;      // own slot only: no hazard with the reads above
            __syncthreads();
            if (writer) {
                const size_t st_out = (size_t)(mc.pin ^ 1) * mc.n_walkers + w;
                if (tid < B9_NPARAM) mc.cur[st_out * B9_NPARAM + tid] = s_cur[tid];
                if (tid == 0) {
                    mc.lp_cur[st_out] = lp_new;
                    if (ok) atomicAdd(mc.n_acc, 1ull);
                    if (mc.lps) mc.lps[(size_t)mc.row * mc.n_walkers + w] = lp_new;
                }
                if (mc.samples && tid < d) mc.samples[((size_t)mc.row * mc.n_walkers + w) * d + tid] = s_cur[fidx];
            }
        }
        // proposal of step t:  s_par = state;  s_par[free[i]] += sum_j chol[i][j] z_j  (j ascending, plain multiply-add)
        if (tid < B9_NPARAM) s_par[tid] = s_cur[tid];
        double delta = 0.0;
#pragma unroll
        for (int j = 0; j < 11; ++j) if (j < d) delta = delta + crow[j] * s_z[j];
        __syncthreads();
        if (tid < d) s_par[fidx] += delta;
        __syncthreads();
        if (writer && tid < B9_NPARAM) params[(size_t)w * B9_NPARAM + tid] = s_par[tid];
        par = s_par;
    }
    derive_iso_block(pk, par, pop, wp, hdr, iso_data, iso_stride, mass_cap, blockIdx.y, gridDim.y, axr);
}


// k_derive_iso_rows: the same derivation for up to 8 parameter rows that travel IN THE KERNEL ARGUMENTS
// (a host-driven b9_logpost call then needs no upload of its own: one launch carries the data).  The
// workgroup (pop 0, part 0) of every walker also stores the row in `params` for the launches that follow.
struct HostRows { double v[8][B9_NPARAM]; };

__global__ __launch_bounds__(256) void k_derive_iso_rows(DevPack pk, HostRows rows, double *__restrict__ params, int n_pops,
                                                          IsoHdr *__restrict__ hdr, double *__restrict__ iso_data,
                                                          long long iso_stride, int mass_cap)
{
    const int wp = blockIdx.x, w = wp / n_pops, pop = wp % n_pops, tid = threadIdx.x;
    __shared__ double s_row[B9_NPARAM];
    const AxisRegs axr = preload_axis(pk);
    if (tid < B9_NPARAM) {
        const double v = rows.v[w][tid];
        s_row[tid] = v;
        if (blockIdx.y == 0 && pop == 0) params[(size_t)w * B9_NPARAM + tid] = v;
    }
    __syncthreads();
    derive_iso_block(pk, s_row, pop, wp, hdr, iso_data, iso_stride, mass_cap, blockIdx.y, gridDim.y, axr);
}
