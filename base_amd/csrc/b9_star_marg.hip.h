// b9_star_marg.hip.h -- marginalised mode: k_marg_table (per-call node table), k_star_marg (one LANE per star) and
// k_star_marg_wd (WD-stage stars) (+ the sampleMass draws).
// Part of the single translation unit b9_kernels.hip (included there, in this order); gfx950 only.
#pragma once

// ------------------------------------------------------------------------------------------
// Marginalised mode (SURVEY 8a row a6 "marg.cpp-like", [RECALL] margEvolveWithBinary).
//
// A star's likelihood is integrated over primary mass (iso_increm = K equal sub-steps inside every EEP interval of the
// derived isochrone, left-endpoint rule) and mass ratio (Q nodes j / Q):
//     L = sum over nodes (n, j) of  prior(m1_n) (dM_n / Q)  prod_f N(obs_f | C_f(n, j), sigma_f^2)
// where C(n, j) is the system's combined apparent magnitude.  NOTHING about a node depends on the star: C(n, j) and
// nb_n = -2 log(prior dM / Q) are tabulated once per call and (walker, population) by k_marg_table.  A star's work is
// then  X = nb_n + sum_f w_f (C_f - obs_f)^2  per node and a log-sum-exp of -X / 2.
//
// Round 4 layout: ONE LANE PER STAR.  A wave holds 64 slot-neighbouring stars (the slot order is mass-sorted, so their
// photometry -- hence the nodes that matter for them -- is similar) and walks the node table; a node's words are the same
// for every lane, so they arrive by SCALAR loads (s_load_dwordx16: a row of 8 magnitudes in one instruction, no VGPRs,
// no LDS) and the per-lane state is just the star's observations / weights and its running log-sum-exp.  No cross-lane
// operation on the data path; a workgroup is FOUR waves holding the same 64 stars, each walking a quarter of every
// 64-node chunk (see k_star_marg).  (Round 3's layout -- one WAVE per star, lane = node -- paid per star for what is now
// paid per 64 stars: bound tables in LDS, wave maxima, a chunk list, 8 vector loads per companion; 1720 wave-instructions
// per star-eval against ~300 here.)
//
// Pruning (rigorous; exact to ~1e-13 relative).  Terms more than B9_MARG_CUT e-folds below a LOWER bound `ref` of what the
// star's value finally contains are dropped (< N e^-40 relative).  ref = max(the lane's running maximum, the field floor):
// the star's value is log(e^la + p L) with la the field-star term, so a cluster term more than CUT below (la - c0m) is
// negligible whatever the other terms are -- field stars and outliers, whose chi^2 is huge everywhere, prune at once
// instead of keeping every node alive.  A lane without a floor (membership prior exactly 1; the sampleMass draws, which
// must see every node that could win) gets its first ref from a seed pass over every 16th node.
// Units are skipped by BOXES: for a set of rows with magnitudes in [lo_f, hi_f], chi^2 >= sum_f w_f dist(obs_f, [lo_f, hi_f])^2.
// Level 1: a 64-node chunk with all its mass ratios; level 2: 16 nodes x one mass ratio.  A unit is evaluated if ANY lane
// cannot exclude it; lanes that could have excluded it evaluate it too (more terms is more exact, never wrong).
// ------------------------------------------------------------------------------------------
#define B9_MARG_CUT 40.0             // terms more than this many e-folds below the reference are dropped

// log(x) for any positive normal x: log_ge1's reduction and polynomial are fdlibm's general e_log.c form (k may be
// negative), 1 ulp; the library log / log10 are 98 VALU instructions each
__device__ __forceinline__ double log_pos(double x) { return log_ge1(x); }

__device__ __forceinline__ double log_prior_mass_dev(double lmn, double m)
{
    const double lm = log_pos(m);
    const double z = (lm * (1.0 / LN10) - MF_MU) / MF_SIGMA;
    return lmn - 0.5 * z * z - lm - log(LN10);
}

// SAMPLE (b9_sample_mass, the sampleMass counterpart -- SURVEY 8f row 4): besides the marginal, every
// star draws ONE (primary mass, mass ratio[, population]) node from its conditional posterior over the
// same grid by the Gumbel-max rule: the node that maximises  log-term + G,  G = -log(-log u),
// u = Philox(seed; row, star, node).  The rule is an argmax, hence independent of the order in
// which the nodes are visited -- the CPU oracle picks the same node.  Nodes the pruning drops (> 40 e-folds
// below the lane's running maximum) draw no number: they could only win with probability e^-40.
struct MargSample {
    double *mass, *ratio, *member;   // [rows][n_stars]
    int *pop;                        // [rows][n_stars] or null
    unsigned k0, k1;
    long long row0;                  // global index of row 0 (RNG counter)
    unsigned *cost;                  // COST instance (the catalogue plan's counting pass): [star chunk][4 waves] units evaluated
};

struct Best { double key, mass, ratio; int pop; };

__device__ __forceinline__ double gumbel(unsigned k0, unsigned k1, unsigned long long row, unsigned star, unsigned long long node, unsigned pop)
{
    unsigned r[4];
    philox4x32((unsigned)row, star, (unsigned)node, (unsigned)(node >> 32) * 2u + pop, k0, k1 ^ (unsigned)(row >> 32), r);
    return -log(-log(u01(r[0], r[1])));
}


// v_max_f64 / v_min_f64 / v_fma_f64 with a wave-uniform (SGPR) operand, as single instructions.  Written out because the
// compiler (i) brackets every fmax / fmin of values it cannot prove quiet with two canonicalising v_max x, x -- 8 instructions
// per filter of a box test instead of 5 -- and (ii) evaluates a Horner step whose coefficient lives in a VGPR as
// v_mov_b64 + v_fmac (the coefficient is loop-invariant, the two-address fmac would destroy it): with the coefficient as
// the one SGPR operand a VOP3 instruction may take, a step is one v_fma_f64.  (Operands are never signalling NaNs.)
__device__ __forceinline__ double max_vs(double a, double s_u) { double r; asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "s"(s_u)); return r; }
__device__ __forceinline__ double min_vs(double a, double s_u) { double r; asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "s"(s_u)); return r; }
__device__ __forceinline__ double max_vv(double a, double b) { double r; asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ double fma_vvs(double a, double b, double s_u) { double r; asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(s_u)); return r; }

// The star enters every chi^2 as the SCALED pair so_f = sqrt(w_f) obs_f, sw_f = sqrt(w_f) (staged at load): a row's
// contribution sqrt(w_f) (C_f - obs_f) is ONE fma(sw_f, C_f, -so_f) with the magnitude as the scalar operand, its square
// one more -- two VALU instructions per filter where w (C - obs)^2 took three (subtract, multiply, fma).

// The node table as the star loop reads its wave-uniform words (boxes, nb minima, seed rows): through the CONSTANT address
// space, so that every such read is a scalar load by construction.  (Left to the compiler, a uniform global load becomes an
// s_load only while its "no store in this kernel may clobber it" analysis succeeds -- it does in k_star_marg alone and gives
// up inside the much larger k_marg_step, where the same reads turned into vector loads + v_readfirstlane: +18 % launch time.)
// The tables a launch reads are written by an EARLIER launch (k_marg_table, or the previous k_marg_step's builders).
typedef const __attribute__((address_space(4))) double *b9_ctab;

// lower bound of sum_f w_f (C_f - obs_f)^2 over every row with lo_f <= C_f <= hi_f (box = {lo[NFP], hi[NFP]}, wave-uniform):
// the scaled distance of obs_f to the interval is max(sw lo - so, so - sw hi, 0)
template <int NFP>
__device__ __forceinline__ double box_bound64(b9_ctab box, const double (&so)[NFP], const double (&sw)[NFP])
{
    double lb = 0.0;
#pragma unroll
    for (int f = 0; f < NFP; ++f) {
        double a, b;
        asm("v_fma_f64 %0, %1, %2, -%3" : "=v"(a) : "v"(sw[f]), "s"(box[f]), "v"(so[f]));          // sw lo - so
        asm("v_fma_f64 %0, -%1, %2, %3" : "=v"(b) : "v"(sw[f]), "s"(box[NFP + f]), "v"(so[f]));   // so - sw hi
        double m = max_vv(a, b);
        asm("v_max_f64 %0, %1, 0" : "=v"(m) : "v"(m));
        lb = fma(m, m, lb);
    }
    return lb;
}

// The same bound in PACKED fp32 (the BOX32 instances: B9_BOX32): two filters per v_pk_fma_f32 (box words as the SGPR-pair
// operand), v_max3_f32 for the clamp -- 5 instructions per filter PAIR where the fp64 form takes 5 per filter (62 box tests
// per workgroup at 50k stars x 8 walkers are 27 % of the star role's VALU instructions).  The bound stays rigorous:
//   * the table holds the box a second time as floats rounded OUTWARD (box_store: a looser box, still a box);
//   * the star's fp32 pair (swf, sof) carries a relative error 2^-24 each, the fma rounds once: the computed distance is
//     within d_f = 2^-22 |so_f| (+ 2^-22 relative) of the true one, so  lb32 <= (1 + eta) lb + (1 + 1 / eta) sum_f d_f^2  for
//     any eta > 0 (2 m d <= eta m^2 + d^2 / eta) -- with eta = 2^-11 and the roundings of the fp32 sum:
//     B9_BOX_INV lb32 <= lb + slack,  B9_BOX_INV = 1 - 2^-10,  slack = 2100 x 2^-44 sum_f so_f^2  (0.006 for eight magnitudes of
//     25 at sigma 0.01);
//   * a box is skipped only when  B9_BOX_INV lb32 + nbmin > xcut + slack,  which implies lb + nbmin > xcut: no term within the
//     cut is lost.  What it costs is boxes that pass at 40.04 e-folds instead of 40.
// The star's fp32 words and its slack live in LDS (`sf`, `s_slack`: the four waves of a workgroup hold the SAME 64 stars),
// read where a box is tested: 16 more live VGPRs cost the row loop its occupancy.
typedef float b9_f2 __attribute__((ext_vector_type(2)));
typedef float b9_f4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(4))) unsigned long long *b9_cbox;     // {lo pairs[NFP / 2], hi pairs[NFP / 2]}
#define B9_BOX_INV 0.9990234375f
template <int NFP>
__device__ __forceinline__ double box_bound32(b9_cbox box, const b9_f4 *sf, int lane)
{
    // the lane's words {sw, sw', so, so'} of a filter pair: one ds_read_b128; up to four pairs requested TOGETHER (left to the
    // compiler the reads share one register quad: a box is four dependent LDS round trips -- two populations 163 -> 175 us)
#ifndef B9_BOX_NB
#define B9_BOX_NB 2
#endif
    constexpr int NB = B9_BOX_NB;
    const unsigned a0 = (unsigned)(size_t)(const __attribute__((address_space(3))) b9_f4 *)sf + (unsigned)lane * 16u;
    b9_f2 acc = {0.0f, 0.0f};
#pragma unroll
    for (int p0 = 0; p0 < NFP / 2; p0 += NB) {
        b9_f4 s4[NB];
#pragma unroll
        for (int k = 0; k < NB; ++k) asm volatile("ds_read_b128 %0, %1" : "=v"(s4[k]) : "v"(a0 + (unsigned)(p0 + k) * 1024u));
        if constexpr (NB == 4) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(s4[0]), "+v"(s4[1]), "+v"(s4[2]), "+v"(s4[3]));
        else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(s4[0]), "+v"(s4[1]));
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            const int p = p0 + k;
            const b9_f2 swp = {s4[k].x, s4[k].y}, sop = {s4[k].z, s4[k].w};
            b9_f2 a, b, m;
            asm("v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=v"(a) : "v"(swp), "s"(box[p]), "v"(sop));             // sw lo - so
            asm("v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[1,0,0] neg_hi:[1,0,0]" : "=v"(b) : "v"(swp), "s"(box[NFP / 2 + p]), "v"(sop));   // so - sw hi
            asm("v_max3_f32 %0, %1, %2, 0" : "=v"(m.x) : "v"(a.x), "v"(b.x));
            asm("v_max3_f32 %0, %1, %2, 0" : "=v"(m.y) : "v"(a.y), "v"(b.y));
            acc = __builtin_elementwise_fma(m, m, acc);
        }
    }
    return (double)((acc.x + acc.y) * B9_BOX_INV);
}
// the star's side of the box test: lane `lane`'s fp32 words into sf[NFP / 2][64]; returns its slack (kept in LDS too: the
// row loop has no registers to spare -- two more live doubles doubled the fused step's scratch traffic, C2 127 -> 136 us)
template <int NFP>
__device__ __forceinline__ double box_stage(b9_f4 *sf, int lane, bool write, const double (&so)[NFP], const double (&sw)[NFP])
{
    double s2 = 0.0;
#pragma unroll
    for (int f = 0; f < NFP; ++f) s2 = fma(so[f], so[f], s2);
    if (write) {
#pragma unroll
        for (int p = 0; p < NFP / 2; ++p) sf[p * 64 + lane] = b9_f4{(float)sw[2 * p], (float)sw[2 * p + 1], (float)so[2 * p], (float)so[2 * p + 1]};
    }
    return s2 < 1e30 ? s2 * (2100.0 / 17592186044416.0) : __builtin_inf();         // (2^44; beyond fp32's range of squares: every box passes)
}
// does any lane need the box?  (box: the table's boxes -- box2 / box1, or box2f / box1f for BOX32; k: which; nbm: the smallest nb of its rows;
// xcut: a term counts while X < xcut)
template <int NFP, bool BOX32>
__device__ __forceinline__ bool box_pass(b9_ctab boxes, size_t k, double nbm, double xcut, const double (&so)[NFP], const double (&sw)[NFP],
                                         const b9_f4 *sf, const double *s_slack, int lane)
{
    if constexpr (BOX32) {
        const double lb = box_bound32<NFP>((b9_cbox)boxes + k * NFP, sf, lane);
        asm volatile("" : "+v"(lane));          // (the slack is read HERE, not hoisted over the row loop)
        return __ballot(lb + nbm <= xcut + s_slack[lane]) != 0ull;
    } else {
        return __ballot(box_bound64<NFP>(boxes + k * 2 * NFP, so, sw) + nbm < xcut) != 0ull;
    }
}
// a box into the table (the builders): floats, rounded outward
// (outward: the bound is moved by 2^-22 of itself + 1e-30 before the conversion rounds it by at most 2^-24 -- two instructions;
// the box grows by 6e-6 mag at magnitude 25)
__device__ __forceinline__ float f32_below(double x) { return (float)(fma(-__builtin_fabs(x), 0x1p-22, x) - 1e-30); }
__device__ __forceinline__ float f32_above(double x) { return (float)(fma(__builtin_fabs(x), 0x1p-22, x) + 1e-30); }
// (the fp64 box is always written, the float copy for the instances that test in fp32.  Measured: leaving the fp64 stores out
// of those instances' builders moves the fused step's register allocation -- C2 121 -> 125 us -- for four stores saved)
template <int NFP>
__device__ __forceinline__ void box_store(bool box32, double *box, double *box_f, int f, double lo, double hi)      // (an empty box -- lo > hi: no finite row -- is stored as [0, 0])
{
    const bool any = lo <= hi;
    box[f] = any ? lo : 0.0; box[NFP + f] = any ? hi : 0.0;
    if (box32) { float *b = reinterpret_cast<float *>(box_f); b[f] = any ? f32_below(lo) : 0.0f; b[NFP + f] = any ? f32_above(hi) : 0.0f; }
}

// X = nb + chi^2 of one table row (wave-uniform row, per-lane star)
template <int NFP>
__device__ __forceinline__ double row_x(b9_ctab row, double nb, const double (&so)[NFP], const double (&sw)[NFP])
{
    double x = nb;
#pragma unroll
    for (int f = 0; f < NFP; ++f) { const double dd = fma(sw[f], row[f], -so[f]); x = fma(dd, dd, x); }
    return x;
}

// One table row (NFP magnitudes) + its nb in SCALAR registers, loaded ASYNCHRONOUSLY: load() only issues the s_load
// instructions, wait() is the s_waitcnt before the first use.  Written as inline assembly because the compiler waits for
// a scalar load right where it issues it when the consumer follows (one exposed L2 round trip per table row, 54 % of the
// kernel's wave cycles); with the next row requested BEFORE the current one is evaluated the trip hides behind ~60
// VALU instructions.  (Scalar loads return out of order, so the only wait is lgkmcnt(0): at most one row is in flight.)
typedef double b9_d4 __attribute__((ext_vector_type(4)));
typedef double b9_d8 __attribute__((ext_vector_type(8)));
template <int NFP> struct SRow;
template <> struct SRow<4> {
    b9_d4 m; double nb;
    __device__ __forceinline__ void load(const double *row, const double *nbp)
    {
        asm volatile("s_load_dwordx8 %0, %1, 0x0" : "=s"(m) : "s"(row));
        asm volatile("s_load_dwordx2 %0, %1, 0x0" : "=s"(nb) : "s"(nbp));
    }
    __device__ __forceinline__ void wait() { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(m), "+s"(nb)); }
    __device__ __forceinline__ double mag(int f) const { return m[f]; }
};
template <> struct SRow<8> {
    b9_d8 m; double nb;
    __device__ __forceinline__ void load(const double *row, const double *nbp)
    {
        asm volatile("s_load_dwordx16 %0, %1, 0x0" : "=s"(m) : "s"(row));
        asm volatile("s_load_dwordx2 %0, %1, 0x0" : "=s"(nb) : "s"(nbp));
    }
    __device__ __forceinline__ void wait() { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(m), "+s"(nb)); }
    __device__ __forceinline__ double mag(int f) const { return m[f]; }
};
template <> struct SRow<16> {
    b9_d8 m0, m1; double nb;
    __device__ __forceinline__ void load(const double *row, const double *nbp)
    {
        asm volatile("s_load_dwordx16 %0, %1, 0x0" : "=s"(m0) : "s"(row));
        asm volatile("s_load_dwordx16 %0, %1, 0x40" : "=s"(m1) : "s"(row));
        asm volatile("s_load_dwordx2 %0, %1, 0x0" : "=s"(nb) : "s"(nbp));
    }
    __device__ __forceinline__ void wait() { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(m0), "+s"(m1), "+s"(nb)); }
    __device__ __forceinline__ double mag(int f) const { return f < 8 ? m0[f & 7] : m1[f & 7]; }
};

// The same row as the TILE instances hold it: read back from the wave's LDS tile (every lane the same address: a broadcast)
// into vector registers.
template <int NFP> struct VRow {
    double m[NFP], nb;
    __device__ __forceinline__ double mag(int f) const { return m[f]; }
};

// X = nb + chi^2 of a row held in scalar registers (or, TILE, in vector registers: the same operations)
template <int NFP, class Row>
__device__ __forceinline__ double srow_x(const Row &r, const double (&so)[NFP], const double (&sw)[NFP])
{
    double x = r.nb;
#pragma unroll
    for (int f = 0; f < NFP; ++f) { const double dd = fma(sw[f], r.mag(f), -so[f]); x = fma(dd, dd, x); }
    return x;
}

// exp(x) for |x| <= 700 to 3e-13 relative (the star's sum needs 1e-10): Cody-Waite reduction, degree-10 Horner polynomial
// on [-ln2/2, ln2/2] with its coefficients in scalar registers, v_ldexp_f64.  No clamping: the caller bounds x.
__device__ __forceinline__ double exp_marg(double x)
{
    const double k = rint(x * 1.4426950408889634074);
    double r = fma(-k, 6.93147180369123816490e-01, x);
    r = fma(-k, 1.90821492927058770002e-10, r);
    double p;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(p) : "v"(r), "s"(1.0 / 3628800.0), "v"(1.0 / 362880.0));
    p = fma_vvs(p, r, 1.0 / 40320.0);   p = fma_vvs(p, r, 1.0 / 5040.0);    p = fma_vvs(p, r, 1.0 / 720.0);
    p = fma_vvs(p, r, 1.0 / 120.0);     p = fma_vvs(p, r, 1.0 / 24.0);      p = fma_vvs(p, r, 1.0 / 6.0);
    p = fma(p, r, 0.5);                 p = fma(p, r, 1.0);                 p = fma(p, r, 1.0);
    return ldexp(p, (int)k);
}

// One more term t = -X / 2 into the lane's log-sum-exp:  value = ref + log(sm).  The reference is FIXED while terms stay
// within 600 e-folds above it (sm then holds up to e^600: no overflow, and a term 40 e-folds under the largest still adds
// its full precision), so the common path is one exponential and one add; a term further above (first terms of a lane
// whose reference was a poor seed) moves the reference -- wave-uniform branch, rare.  tmax: the largest term seen (the
// pruning threshold follows it).
__device__ __forceinline__ void lse_term(double t, double &ref, double &sm, double &tmax)
{
    const double d = t - ref;
    if (__ballot(d > 600.0) != 0ull) {
        const bool up = d > 600.0;
        const double e = exp_marg(max_vs(up ? -d : d, -700.0));
        sm = up ? fma(sm, e, 1.0) : sm + e;
        ref = up ? t : ref;
    } else {
        sm += exp_marg(d);
    }
    tmax = max_vv(tmax, t);
}

// Waves per SIMD the instances are built for (tools/kernel_resources.py)
#define B9_MARG_WAVES(NFP, NPOPS, SAMPLE) (((NFP) >= 16 || (SAMPLE)) ? 4 : ((NPOPS) == 2 ? 6 : 7))

// A workgroup = FOUR waves holding the SAME 64 stars; a chunk's units (sub-chunk of 16 nodes x one mass ratio) are dealt over
// them diagonally -- unit (sub, j) is wave (2 sub + j) mod 4's: a quarter of every star's window, wherever it lies -- and the
// four partial log-sum-exps are merged through LDS at the end.  (With one wave
// per 64 stars the launch lasted as long as its heaviest wave -- a chunk of giants whose windows barely overlap walks
// 2000 terms against an average of 240; the mean wave lived a quarter of the launch.)  A wave prunes against the maxima
// the four waves' seed passes merged BEHIND A BARRIER and, from there on, its OWN running maximum only: which terms enter a
// star's sum is a function of the data, never of how the waves' clocks interleave (round 4 read the neighbours' running
// maxima from LDS without a barrier -- a valid reference whenever it was read, but a timing-dependent one, and with it the
// last bits of a star's value).  A wave's sub-chunks sample every region of the window, so its own maximum is within a
// node spacing of the workgroup's.  The price (`tools/marg_stats.py`, bench cluster): 184 rows evaluated per star-eval against
// round 4's 163, at the same launch time.  Measured and not adopted (round 5): exchanging the maxima behind a barrier after
// every chunk of the walk (179 rows; 137.5 -> 153.6 us per step of 50k stars x 8 walkers: a barrier stalls the three waves
// that are ahead); starting the walk at the chunk whose boxes look best for the 64 stars together and re-testing level 1
// at every later chunk (184 rows, 141.0 us); the same plus ONE exchange after that first chunk (179 rows, 155.4 us).
// What a wave evaluates is the union of its stars' own halos, whatever the order it meets them in.  The level-1 boxes are dealt over the waves too (wave k tests chunks k, k + 4, ...)
// and the outcome travels as a bit mask in LDS.
#define B9_MARG_MASK_WORDS 16        // level-1 outcomes of up to 1024 chunks go through the mask; later chunks are tested by every wave

// What a launch evaluates the stars AGAINST: the headers, parameter rows and node tables of its walkers.  The plain
// launch knows them at entry; the sampler's fused step (b9_marg_step.hip.h) picks one of two candidates by the previous
// step's accept / reject decision, taken inside the launch: `select.issue(w)` requests what the decision reads (before
// the star's own words are requested: its chain is the longer one), `select.finish(w)` -- called once per workgroup, by
// all its threads, after the star's words have been requested -- returns the choice.
struct MargSel { const IsoHdr *hdr; const double *params; const double *tab; };

// WARM (split catalogues -- one chain, a few hundred workgroups, every one resident): the star workgroups of an XCD read their
// walker's node tables through once, a slice each (256 lanes x one 128-byte line per round, at most B9_WARM_ROUNDS rounds),
// before any of them walks.  The tables were written by the launch before on whatever XCD its workgroups ran; a lone wave's
// walk otherwise pays a miss in its XCD's L2 for every box and every unit it touches, one after the other (~3 us per unit
// against ~1 of arithmetic: tools/gantt_marg.py) -- the warm-up takes the misses all at once (in k_marg_step: inside the wait
// for the decision, both candidates).  Not for tables that would crowd the XCD's 4 MB (B9_WARM_MAX_BYTES: the 8 x 8 grid's
// two candidates, 4.4 MB, lost 3 us).
#define B9_WARM_ROUNDS 4
#define B9_WARM_MAX_BYTES (3u << 19)
struct L2Warm {
    unsigned r[B9_WARM_ROUNDS];
    // the lines (16 doubles) of [b0, b0 + n) and then [b1, b1 + n) (b1 null: none): workgroup `rank` of `count`
    __device__ __forceinline__ void issue(const double *b0, const double *b1, size_t n, int rank, int count)
    {
#pragma unroll
        for (int k = 0; k < B9_WARM_ROUNDS; ++k) r[k] = 0u;
        const size_t total = b1 ? 2 * n : n;
        if (total * 8 > B9_WARM_MAX_BYTES) return;
#pragma unroll
        for (int k = 0; k < B9_WARM_ROUNDS; ++k) {
            const size_t off = (((size_t)k * count + rank) * 256 + threadIdx.x) * 16;
            if (off < total) {
                const double *a = off < n ? b0 + off : b1 + (off - n);
                asm volatile("global_load_dword %0, %1, off" : "=v"(r[k]) : "v"(a));
            }
        }
    }
    // (the registers are the loads' until here)
    __device__ __forceinline__ void wait() { asm volatile("s_waitcnt vmcnt(0)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3])); }
};
static_assert(B9_WARM_ROUNDS == 4, "L2Warm::wait names four registers");

template <bool WARM>
struct MargSelPlain {
    MargSel s;
    size_t per;              // WARM: doubles of one walker's tables
    int rank, count;         //       this workgroup among the launch's workgroups on its XCD
    L2Warm warm;
    __device__ __forceinline__ void issue(int w) { if constexpr (WARM) warm.issue(s.tab + (size_t)w * per, nullptr, per, rank, count); }
    __device__ __forceinline__ MargSel finish(int) { if constexpr (WARM) warm.wait(); return s; }
};

// TILE: how a unit's 16 rows reach the lanes.  The scalar path (SRow) holds ONE row ahead in scalar registers -- all that
// fits -- so a wave pays a scalar-load round trip per row unless six other waves of its SIMD cover it.  The TILE instances
// fetch a unit's 16 rows + 16 nb words in ONE vector round trip (lane l: 16 bytes of the kilobyte), stage them in the wave's
// LDS tile (`tile_lds`: 4 x (16 NFP + 16) doubles; in k_marg_step the dynamic LDS its builders' tiles occupy in THEIR
// workgroups) and read them back row by row as broadcasts into vector registers.  Same rows, same operations, same order:
// the bits of the scalar path -- which instance runs is the launch's choice (b9_kernels.hip).  Measured, us per sampler
// step, scalar -> tile: one chain on 10k stars 31.6 -> 26.2 (a lone wave is latency-bound), 20k stars 37.2 -> 33.3; 8
// walkers x 50k stars 138.0 -> 133.9; two populations (30k stars x 8 walkers) 207 -> 175; 16 filters 238 -> 215.  Built for
// B9_TILE_OCC waves per SIMD with B9_TILE_UNROLL rows in flight (sweep: 8 filters occupancy 4 / 5 / 6 / 7: 146 / 137 / 134 /
// 138 us; two populations 186 / 175 / 184 / 192; 16 filters at 6: 418, the rows spill).
#define B9_TILE_DOUBLES(NFP) (16 * (NFP) + 16)
// which instances test their boxes in packed fp32 (box_bound32)
#ifndef B9_BOX32_P2
#define B9_BOX32_P2 0
#endif
#define B9_BOX32(NFP, NPOPS) ((NPOPS) == 1 || B9_BOX32_P2)
// TILE = 2, the SPARSE setting (launches that leave the chip nearly empty: one chain on a split catalogue): four rows in flight
// at occupancy 4 -- 20k stars, one chain, 8 x 8 grid: 62.0 -> 55.9 us per step; with 8 walkers on the same catalogue it
// loses (91 -> 99.7), so the launch picks by its own size (b9_kernels.hip: marg_sparse).
#ifndef B9_TILE_OCC_DENSE            // (tools/build_variant.py -DB9_TILE_OCC_DENSE=7 -DB9_TILE_UNROLL_DENSE=1: the sweep's variants)
#define B9_TILE_OCC_DENSE 6
#define B9_TILE_UNROLL_DENSE 2
#endif
#ifndef B9_TILE_OCC_P2
#define B9_TILE_OCC_P2 5
#define B9_TILE_UNROLL_P2 4
#endif
#define B9_TILE_OCC(NFP, NPOPS, TILE) (((NFP) >= 16 || (TILE) == 2) ? 4 : ((NPOPS) == 2 ? B9_TILE_OCC_P2 : B9_TILE_OCC_DENSE))
#define B9_TILE_UNROLL(NFP, NPOPS, TILE) (((NFP) >= 16 || (TILE) == 2) ? 4 : ((NPOPS) == 2 ? B9_TILE_UNROLL_P2 : B9_TILE_UNROLL_DENSE))
template <int NFP, int NPOPS, bool SAMPLE, bool SPLIT, bool COST, int TILE, class Select>
__device__ __forceinline__ void star_marg_body(const DevPack &pk, const DevStars &st, int block_id, double *tile_lds,
                 const double *__restrict__ iso_data, long long iso_stride, int mass_cap,
                 double *__restrict__ partial, long long partial_stride, double *__restrict__ perstar,
                 int K, int Q, const MargSample &ms, const MargLayout &L,
                 int n_walkers, double cut2, int wsplit, double *__restrict__ shares, Select select)
{
    __shared__ double s_tmax[NPOPS][4][64], s_ref[NPOPS][4][64], s_sm[NPOPS][4][64];
    __shared__ unsigned long long s_mask[NPOPS][B9_MARG_MASK_WORDS];
    constexpr bool BOX32 = B9_BOX32(NFP, NPOPS);
    __shared__ b9_f4 s_sf[BOX32 ? (NFP / 2) * 64 : 1];                   // the 64 stars in fp32 for the box tests (box_bound32)
    __shared__ double s_slack[BOX32 ? 64 : 1];
    __shared__ double s_bkey[SAMPLE ? 4 : 1][64], s_bmass[SAMPLE ? 4 : 1][64], s_bratio[SAMPLE ? 4 : 1][64];
    __shared__ int s_bpop[SAMPLE ? 4 : 1][64];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    MLIFE(0, __builtin_amdgcn_s_memrealtime());
    // 1-D grid; ids are dealt round-robin over the 8 XCDs.  An XCD takes one of `wsplit` (1, 2, 4 or 8, dividing the walker
    // count) walker groups and one of 8 / wsplit star-chunk groups: its L2 then holds n_walkers / wsplit node tables and
    // fetches 1 / (8 / wsplit) of the star data per walker of its group -- the split trades table re-fetches against star
    // re-fetches (b9k_star_marg picks it).  Dispatch position p -> star chunk marg_order[p]: most expensive chunks first,
    // neighbours in the order on different XCD groups (a contiguous share per XCD left the XCD holding the giants working
    // alone for the launch's second half), and inside an XCD the walkers of one star chunk are neighbours in dispatch
    // order: the chunk's star data is fetched from HBM once per group.  (Speed only; any placement is correct.)
    const int xcd = block_id & 7, i_x = block_id >> 3;
    const int csplit = 8 / wsplit, wg_n = n_walkers / wsplit;           // chunk groups; walkers per group
    const int a = xcd % wsplit, b = xcd / wsplit;
    // SPLIT (small catalogues; DevStars::mg_piece): the dispatch position names a PIECE -- workgroup `split` of the n_split that
    // share one (star chunk, walker) takes the node chunks c = split, split + n_split, ... of every star's window and leaves
    // its per-star share (ref, sum) in `shares` for k_marg_merge.  (A window is one to three chunks wide, so of two pieces one
    // often walks most of it -- 16 us against 5, tools/probes/gantt_top.py.  Dealing the 16-node SUB-chunks over the pieces
    // instead evens that out but makes every piece test every level-1 box: one chain on 10k stars 24.2 -> 23.4 us per step, on
    // 20k stars 26.6 -> 28.4, 8 x 8 grid 43.0 / 53.5 -> 38.6 / 57.9, 8 walkers 43.6 / 76.9 -> 44.4 / 84.0 -- not adopted.)  (n_split is a compile-time 1 in the unsplit instance: its
    // code is the one-workgroup kernel's.)
    const int p_local = i_x / wg_n, wl = i_x - p_local * wg_n;
    const int w = wl * wsplit + a;
    const int pos = p_local * csplit + b;
    if (SPLIT ? pos >= st.mg_n_pieces : pos * 64 >= st.mg_pad) return;
    const int piece = SPLIT ? st.mg_piece[pos] : 0;
    const int split = SPLIT ? (piece >> 20) & 31 : 0, n_split = SPLIT ? (piece >> 25) & 63 : 1;      // (at most 32 pieces)
    SSTAMP(block_id, 0);
    select.issue(w);
    const int sc = SPLIT ? piece & 0xFFFFF : st.marg_order[pos];
    const int slot = sc * 64 + lane;                                // (slot of the marginalised mode's own copy: DevStars::mg_*)
    // the star's own words are requested before anything that depends on the walker's candidate
    const int orig = st.mg_perm[slot];
    double so[NFP], sw[NFP];                                       // the star, scaled: sqrt(w) obs, sqrt(w)
#pragma unroll
    for (int f = 0; f < NFP; ++f) { so[f] = st.mg_so[B9_SIDX(NFP, f, slot)]; sw[f] = st.mg_sw[B9_SIDX(NFP, f, slot)]; }
    const double c0m = st.mg_c0m[slot], la = st.mg_la[slot];
    if constexpr (BOX32) { const double sl = box_stage<NFP>(s_sf, lane, wave == 0, so, sw); if (wave == 0) s_slack[lane] = sl; }      // (visible behind the barrier that closes the seed pass)
    const MargSel sel = select.finish(w);
    SSTAMP(block_id, 1);
    const IsoHdr *__restrict__ const hdr = sel.hdr;
    const double *__restrict__ const tab = sel.tab;
    const double *par = sel.params + (size_t)w * B9_NPARAM;
    IsoView<NFP> iso_g[NPOPS];
    double tip_min;
    const bool valid = load_iso_views<NFP, NPOPS>(hdr, iso_data, iso_stride, mass_cap, w, iso_g, tip_min);
    if (!valid) {
        if (wave == 0 && split == 0) {
            if (lane == 0) partial[(size_t)w * partial_stride + sc] = 0.0;
            if (perstar && orig >= 0) perstar[(size_t)w * st.n + orig] = NEG_INF;
        }
        return;
    }
    const bool dead = orig < 0;
    // the field floor (in the units of the terms: the star's constant c0m is added at the end)
    const double floor_t = (SAMPLE || !(cut2 < __builtin_inf())) ? NEG_INF : la - c0m;
    if (threadIdx.x < NPOPS * B9_MARG_MASK_WORDS) (&s_mask[0][0])[threadIdx.x] = 0ull;

    // ---- per population: the pruning reference every wave starts from, then the level-1 boxes of this wave's chunks
    double ref[NPOPS], sm[NPOPS], tmax[NPOPS];
#pragma unroll
    for (int kp = 0; kp < NPOPS; ++kp) {
        ref[kp] = floor_t; sm[kp] = 0.0;                                  // this wave's share of the lane's value: ref + log(sm)
        tmax[kp] = dead ? __builtin_inf() : floor_t;                      // a term counts while it is within CUT of tmax
        // seed pass: lanes without a reference take the best single-star term among every 16th node (this wave: its sub-chunks)
        if (__ballot(!dead && tmax[kp] == NEG_INF) != 0ull && cut2 < __builtin_inf()) {
            const b9_ctab t_wp = (b9_ctab)(tab + (size_t)(w * NPOPS + kp) * L.total);
            const int n_units = (((iso_g[kp].n - 1) * K + 63) >> 6) * 4;
            double xmin = __builtin_inf();
            for (int u = wave; u < n_units; u += 4) {
                const double x = row_x<NFP>(t_wp + L.o_rows + (size_t)u * Q * 16 * NFP, t_wp[L.o_nb + u * 16], so, sw);
                xmin = __builtin_fmin(xmin, x);
            }
            if (!dead && tmax[kp] == NEG_INF) { ref[kp] = -0.5 * xmin; tmax[kp] = ref[kp]; }
        }
        s_tmax[kp][wave][lane] = tmax[kp];
    }
    __syncthreads();
#pragma unroll
    for (int kp = 0; kp < NPOPS; ++kp) {
        const b9_ctab t_wp = (b9_ctab)(tab + (size_t)(w * NPOPS + kp) * L.total);
        const int n_chunks = ((iso_g[kp].n - 1) * K + 63) >> 6;
        tmax[kp] = __builtin_fmax(__builtin_fmax(tmax[kp], s_tmax[kp][0][lane]), __builtin_fmax(s_tmax[kp][1][lane], __builtin_fmax(s_tmax[kp][2][lane], s_tmax[kp][3][lane])));
        const double xcut = fma(-2.0, tmax[kp], cut2);
        const int c_end = n_chunks < 64 * B9_MARG_MASK_WORDS ? n_chunks : 64 * B9_MARG_MASK_WORDS;
        for (int c = split + n_split * wave; c < c_end; c += 4 * n_split) {          // (this workgroup's node chunks, dealt over its waves)
            MSTAT(0, 1);
            if (box_pass<NFP, BOX32>(t_wp + (BOX32 ? L.o_box1f : L.o_box1), (size_t)c, t_wp[L.o_nbmin64 + c], xcut, so, sw, s_sf, s_slack, lane) && lane == 0)
                atomicOr(&s_mask[kp][c >> 6], 1ull << (c & 63));
        }
    }
    __syncthreads();

    SSTAMP(block_id, 2);
    Best best; best.key = NEG_INF; best.mass = 0.0; best.ratio = 0.0; best.pop = 0;
    const unsigned long long g_row = SAMPLE ? (unsigned long long)(ms.row0 + w) : 0ull;
    unsigned n_cost = 0;                                 // COST: (16 nodes x one mass ratio) units this wave evaluates
    double lw_pop[2] = {0.0, 0.0};                       // log weight of the population in the key
    if (SAMPLE && NPOPS == 2) { const double lam = par[B9_P_LAMBDA]; lw_pop[0] = log(lam); lw_pop[1] = log1p(-lam); }

#pragma unroll
    for (int kp = 0; kp < NPOPS; ++kp) {
        const double *__restrict__ const t_g = tab + (size_t)(w * NPOPS + kp) * L.total;      // (generic: the asynchronous row loads take addresses)
        const b9_ctab t_wp = (b9_ctab)t_g;
        const double *__restrict__ const t_rows = t_g + L.o_rows, *__restrict__ const t_nb = t_g + L.o_nb;
        const b9_ctab t_box2 = t_wp + (BOX32 ? L.o_box2f : L.o_box2), t_nbmin16 = t_wp + L.o_nbmin16;
        const int n_chunks = ((iso_g[kp].n - 1) * K + 63) >> 6;
        // one chunk that passed level 1: this wave's sub-chunk, mass ratio by mass ratio
        auto chunk = [&](int c) {
            MSTAT(1, 1);
            double xcut = fma(-2.0, tmax[kp], cut2);                      // a term counts while X < xcut
            // The chunk's 4 Q units (sub-chunk of 16 nodes x mass ratio) are dealt over the four waves DIAGONALLY: unit (sub, j)
            // is wave (2 sub + j) mod 4's.  A main-sequence star's window is one or two sub-chunks wide and alive at the first one
            // or two mass ratios (a q = 0.5 companion already costs it its chi^2): with wave k taking sub-chunk k whole, one or
            // two waves walked all of it while the others waited at the closing barrier (tools/gantt_marg.py: walk 13 us against
            // 2) -- dealt this way, two sub-chunks x two mass ratios are four waves' work.
            for (int sub = 0; sub < 4; ++sub) {
            const int u = c * 4 + sub;
            const double nbm = t_nbmin16[u];
            const double *__restrict__ const nbp = t_nb + u * 16;
            for (int j = (wave - 2 * sub) & 3; j < Q; j += 4) {
                // level 2: this wave's 16 nodes x one mass ratio
                MSTAT(2, 1);
                if constexpr (BOX32) {
                    if (!box_pass<NFP, true>(t_box2, (size_t)u * Q + j, nbm, xcut, so, sw, s_sf, s_slack, lane)) continue;
                } else {
                    const double lb2 = box_bound64<NFP>(t_box2 + ((size_t)u * Q + j) * 2 * NFP, so, sw);
                    if (__ballot(lb2 + nbm < xcut) == 0ull) continue;
                }
                MSTAT(3, 1);
                MLIFE_UNIT();
                if (COST) ++n_cost;
                const double *__restrict__ const rowp = t_rows + ((size_t)u * Q + j) * 16 * NFP;
                // one row's term for every lane (the 64 stars), into the lanes that still count it
                auto term = [&](const auto &r, int i) {
                    const double x = srow_x<NFP>(r, so, sw);
                    const bool live = x < xcut;
                    MSTAT_LIVE(4, live);
                    if (live) {
                        const double t = -0.5 * x;
                        if (SAMPLE) {
                            const int node = u * 16 + i;
                            const double key = t + lw_pop[kp] + gumbel(ms.k0, ms.k1, g_row, (unsigned)orig, (unsigned long long)((long long)node * Q + j), (unsigned)kp);
                            if (key > best.key) {
                                const int e = node / K, sb = node - e * K;
                                const double a = iso_g[kp].mass[e];
                                best.key = key; best.mass = fma((double)sb, (iso_g[kp].mass[e + 1] - a) / K, a);
                                best.ratio = (double)j / (double)Q; best.pop = kp;
                            }
                        }
                        lse_term(t, ref[kp], sm[kp], tmax[kp]);
                    }
                };
                if constexpr (TILE) {
                    constexpr int T_UNROLL = B9_TILE_UNROLL(NFP, NPOPS, TILE);
                    // the unit in one vector round trip -> the wave's LDS tile -> rows as broadcasts
                    double *const tile = tile_lds + wave * B9_TILE_DOUBLES(NFP);
                    const double2 *__restrict__ const src = reinterpret_cast<const double2 *>(rowp);
                    double2 v[(NFP + 7) / 8];
#pragma unroll
                    for (int k = 0; k < (NFP + 7) / 8; ++k) v[k] = (NFP >= 8 || lane < 32) ? src[lane + 64 * k] : double2{0.0, 0.0};
                    const double nbv = nbp[lane & 15];
                    __builtin_amdgcn_wave_barrier();             // (the previous unit's reads of the tile are done)
#pragma unroll
                    for (int k = 0; k < (NFP + 7) / 8; ++k) if (NFP >= 8 || lane < 32) reinterpret_cast<double2 *>(tile)[lane + 64 * k] = v[k];
                    if (lane < 16) tile[16 * NFP + lane] = nbv;
                    __builtin_amdgcn_wave_barrier();             // (one wave: its LDS accesses complete in program order)
#pragma unroll T_UNROLL
                    for (int i = 0; i < 16; ++i) {
                        VRow<NFP> r;
#pragma unroll
                        for (int f = 0; f < NFP; ++f) r.m[f] = tile[i * NFP + f];
                        r.nb = tile[16 * NFP + i];
                        term(r, i);
                    }
                    xcut = fma(-2.0, tmax[kp], cut2);
                    continue;
                }
                // the unit's 16 rows, two at a time: row i + 1 is requested before row i is evaluated.  (The request past the
                // unit's last row reads the next unit's first row / the word after nb's sub-chunk: inside the table, unused.)
                SRow<NFP> ra, rb;
                ra.load(rowp, nbp);
                ra.wait();
#pragma unroll 1
                for (int i = 0; i < 16; i += 2) {
                    rb.load(rowp + (i + 1) * NFP, nbp + i + 1);
                    __builtin_amdgcn_sched_barrier(0);          // (the request stays AHEAD of the evaluation it is to hide behind)
                    term(ra, i);
                    rb.wait();
                    ra.load(rowp + (i + 2) * NFP, nbp + i + 2);
                    __builtin_amdgcn_sched_barrier(0);
                    term(rb, i + 1);
                    ra.wait();
                }
                xcut = fma(-2.0, tmax[kp], cut2);
            }
            }
        };
        const int n_words = (n_chunks + 63) >> 6;
        for (int wi = 0; wi < n_words && wi < B9_MARG_MASK_WORDS; ++wi) {
            unsigned long long m = s_mask[kp][wi];
            m = ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(m >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((unsigned)m);
            while (m) { const int c = wi * 64 + __builtin_ctzll(m); m &= m - 1; chunk(c); }
        }
        for (int c = 64 * B9_MARG_MASK_WORDS + split; c < n_chunks; c += n_split) {      // (tables longer than the mask: every wave tests)
            if (box_pass<NFP, BOX32>(t_wp + (BOX32 ? L.o_box1f : L.o_box1), (size_t)c, t_wp[L.o_nbmin64 + c], fma(-2.0, tmax[kp], cut2), so, sw, s_sf, s_slack, lane)) chunk(c);
        }
        s_ref[kp][wave][lane] = ref[kp]; s_sm[kp][wave][lane] = sm[kp];
    }
    if (SAMPLE) { s_bkey[wave][lane] = best.key; s_bmass[wave][lane] = best.mass; s_bratio[wave][lane] = best.ratio; s_bpop[wave][lane] = best.pop; }
    if (COST && lane == 0) ms.cost[(size_t)sc * 4 + wave] = n_cost;
    SSTAMP(block_id, 3);
    __syncthreads();
    SSTAMP(block_id, 4);
    if (wave != 0) return;
    // ---- wave 0: merge the four shares, finish the star, sum the chunk
    double ll[NPOPS];
#pragma unroll
    for (int kp = 0; kp < NPOPS; ++kp) {
        double r = NEG_INF;
#pragma unroll
        for (int k = 0; k < 4; ++k) r = (s_sm[kp][k][lane] > 0.0 && s_ref[kp][k][lane] > r) ? s_ref[kp][k][lane] : r;
        double S = 0.0;
#pragma unroll
        for (int k = 0; k < 4; ++k) S += (s_sm[kp][k][lane] > 0.0) ? s_sm[kp][k][lane] * exp_fast(s_ref[kp][k][lane] - r) : 0.0;
        SSTAMP(block_id, 5);
        if (SPLIT) {                                    // this workgroup's share of the star's sum: merged by k_marg_merge
            double *sh = shares + ((((size_t)w * st.mg_n_pieces + st.mg_share_base[sc] + split) * NPOPS + kp) * 128);
            sh[lane] = r; sh[64 + lane] = S;
        }
        ll[kp] = (S > 0.0) ? c0m + (r + log(S)) : NEG_INF;
    }
    if (SPLIT) return;
    double v = 0.0;
    if (!dead) {
        double l = ll[0];
        if (NPOPS == 2) { const double lam = par[B9_P_LAMBDA]; l = logaddexp(log(lam) + ll[0], log1p(-lam) + ll[NPOPS - 1]); }
        v = logaddexp(la, l);
        if (perstar) perstar[(size_t)w * st.n + orig] = v;
        if (SAMPLE) {
#pragma unroll
            for (int k = 1; k < 4; ++k)
                if (s_bkey[k][lane] > best.key) { best.key = s_bkey[k][lane]; best.mass = s_bmass[k][lane]; best.ratio = s_bratio[k][lane]; best.pop = s_bpop[k][lane]; }
            const size_t o = (size_t)w * st.n + orig;
            const bool any = best.key != NEG_INF;
            ms.mass[o] = any ? best.mass : 0.0;
            ms.ratio[o] = any ? best.ratio : 0.0;
            ms.member[o] = (l == NEG_INF) ? 0.0 : exp(l - v);       // p L_cluster / (p L_cluster + (1 - p) L_field)
            if (ms.pop) ms.pop[o] = any ? best.pop : 0;
        }
    }
    // the chunk's partial sum (fixed order: the wave's shuffle tree; empty slots add 0)
    const double tot = wave_sum(v);
    if (lane == 0) partial[(size_t)w * partial_stride + sc] = tot;
    MLIFE(1, __builtin_amdgcn_s_memrealtime());
}

template <int NFP, int NPOPS, bool SAMPLE, bool SPLIT, bool COST = false, int TILE = 0>
__global__ __launch_bounds__(256, TILE ? B9_TILE_OCC(NFP, NPOPS, TILE) : B9_MARG_WAVES(NFP, NPOPS, SAMPLE))
void k_star_marg(DevPack pk, DevStars st, const IsoHdr *__restrict__ hdr,
                 const double *__restrict__ iso_data, long long iso_stride,
                 int mass_cap, const double *__restrict__ params,
                 double *__restrict__ partial, long long partial_stride, double *__restrict__ perstar,
                 int K, int Q, MargSample ms, const double *__restrict__ tab, MargLayout L,
                 int n_walkers, double cut2, int wsplit, double *__restrict__ shares)
{
    extern __shared__ __attribute__((aligned(16))) double tile_lds[];        // (TILE: 4 x B9_TILE_DOUBLES doubles)
    star_marg_body<NFP, NPOPS, SAMPLE, SPLIT, COST, TILE>(pk, st, (int)blockIdx.x, tile_lds, iso_data, iso_stride, mass_cap, partial, partial_stride, perstar, K, Q, ms, L,
                                                    n_walkers, cut2, wsplit, shares,
                                                    MargSelPlain<SPLIT && !COST>{MargSel{hdr, params, tab}, (size_t)NPOPS * L.total, (int)blockIdx.x >> 3, ((int)gridDim.x + 7) >> 3, {}});
}

// k_marg_merge: the stars of a SPLIT launch -- one wave per (star chunk, walker), lane = star: the chunk's shares (ref, sum) of
// every star merged in the order of the pieces (the wave-0 merge of k_star_marg, continued), the star finished (mass-prior
// constant, populations, field-star mixture) and the chunk's partial sum formed: what k_star_marg's last wave does when one
// workgroup holds the whole window.
template <int NPOPS>
__device__ __forceinline__ void marg_merge_body(const DevStars &st, const IsoHdr *__restrict__ hdr, const double *__restrict__ params,
                                                double *__restrict__ partial, long long partial_stride, double *__restrict__ perstar,
                                                const double *__restrict__ shares)
{
    const int lane = threadIdx.x, sc = blockIdx.x, w = blockIdx.y;
    const int sb = st.mg_share_base[sc], n_split = st.mg_share_base[sc + 1] - sb;
    bool valid = true;
#pragma unroll
    for (int kp = 0; kp < NPOPS; ++kp) valid = valid && hdr[w * NPOPS + kp].valid;
    if (!valid) return;                                  // (k_star_marg's split 0 has written the chunk's 0 / -inf)
    const int slot = sc * 64 + lane, orig = st.mg_perm[slot];
    const double c0m = st.mg_c0m[slot], la = st.mg_la[slot];
    double ll[NPOPS];
#pragma unroll
    for (int kp = 0; kp < NPOPS; ++kp) {
        const double *sh = shares + (((size_t)w * st.mg_n_pieces + sb) * NPOPS + kp) * 128;
        double r = NEG_INF;
        for (int k = 0; k < n_split; ++k) {
            const double rk = sh[(size_t)k * NPOPS * 128 + lane], sk = sh[(size_t)k * NPOPS * 128 + 64 + lane];
            r = (sk > 0.0 && rk > r) ? rk : r;
        }
        double S = 0.0;
        for (int k = 0; k < n_split; ++k) {
            const double rk = sh[(size_t)k * NPOPS * 128 + lane], sk = sh[(size_t)k * NPOPS * 128 + 64 + lane];
            S += (sk > 0.0) ? sk * exp_fast(rk - r) : 0.0;
        }
        ll[kp] = (S > 0.0) ? c0m + (r + log(S)) : NEG_INF;
    }
    double v = 0.0;
    if (orig >= 0) {
        double l = ll[0];
        if (NPOPS == 2) { const double lam = params[(size_t)w * B9_NPARAM + B9_P_LAMBDA]; l = logaddexp(log(lam) + ll[0], log1p(-lam) + ll[NPOPS - 1]); }
        v = logaddexp(la, l);
        if (perstar) perstar[(size_t)w * st.n + orig] = v;
    }
    const double tot = wave_sum(v);
    if (lane == 0) partial[(size_t)w * partial_stride + sc] = tot;
}

template <int NPOPS>
__global__ __launch_bounds__(64) void k_marg_merge(DevStars st, const IsoHdr *__restrict__ hdr, const double *__restrict__ params,
                                                   double *__restrict__ partial, long long partial_stride, double *__restrict__ perstar,
                                                   const double *__restrict__ shares)
{
    marg_merge_body<NPOPS>(st, hdr, params, partial, partial_stride, perstar, shares);
}

// ------------------------------------------------------------------------------------------
// k_marg_table: one call's node table (layout: MargLayout, b9_device.h).  One workgroup per (walker-population, 64-node
// chunk); thread = (node of the chunk, mass-ratio slice): for primary node n (EEP interval n / K, sub-step n % K) and mass
// ratio j / Q the secondary mass m2 = (j / Q) m1, bracket + linear interpolation of the derived isochrone's rows, the
// system's combined magnitude per filter and -- so that the star loop has nothing left to add -- modulus and absorption.
// Then the boxes: minimum / maximum over the 16 rows of every (sub-chunk, mass ratio) by shuffles inside a 16-lane
// row, over the chunk through LDS.  Rows of nodes that do not exist (past the isochrone's end, empty EEP interval) hold
// zeros, carry nb = +inf and are left out of the boxes.
// ------------------------------------------------------------------------------------------
template <int NFP>
__global__ __launch_bounds__(256) void k_marg_table(DevPack pk, const IsoHdr *__restrict__ hdr, const double *__restrict__ iso_data, long long iso_stride,
                                                    int mass_cap, int n_pops, const double *__restrict__ params, int K, int Q,
                                                    double *__restrict__ tab, MargLayout L)
{
    extern __shared__ __attribute__((aligned(16))) double s_mass[];
    const int wp = blockIdx.x, c = blockIdx.y, tid = threadIdx.x, lane = tid & 63, jl = tid >> 6;
    const bool box32 = B9_BOX32(NFP, n_pops);               // (the box format the star kernel's instance reads)
    const IsoHdr h = hdr[wp];
    if (!h.valid) return;
    const int n_nodes = (h.n - 1) * K;
    if (c * 64 >= n_nodes) return;                                      // (the star kernel stops at the isochrone's last chunk)
    const double *g_mass = iso_data + (size_t)wp * iso_stride, *g_mags = g_mass + mass_cap;
    double *s_box = s_mass + mass_cap + 8;                             // [4 waves][lo | hi][NFP]
    for (int j = tid; j < mass_cap; j += 256) s_mass[j] = g_mass[j];
    if (tid < 8) s_mass[mass_cap + tid] = __builtin_inf();              // find_bracket's masked over-read
    __syncthreads();
    const double *par = params + (size_t)(wp / n_pops) * B9_NPARAM;
    const double mod = par[B9_P_MOD], av = par[B9_P_ABS];
    double *out = tab + (size_t)wp * L.total;
    const int node = c * 64 + lane, sub = lane >> 4, i16 = lane & 15, u = c * 4 + sub;
    // the primary
    bool ok = node < n_nodes;
    const int e = ok ? node / K : 0, s = node - e * K;
    const double a = s_mass[e], d = s_mass[e + 1] - a;
    ok = ok && d > 0.0;
    const double dM = d / K;
    const double m1 = fma((double)s, dM, a);
    const double t1 = ok ? (m1 - a) / d : 0.0;
    const double *r0 = g_mags + (size_t)e * NFP;
    double p1[NFP];
#pragma unroll
    for (int f = 0; f < NFP; ++f) p1[f] = lerp(r0[f], r0[NFP + f], t1);
    double lo64[NFP], hi64[NFP];
#pragma unroll
    for (int f = 0; f < NFP; ++f) { lo64[f] = __builtin_inf(); hi64[f] = NEG_INF; }
    for (int j = jl; j < Q; j += 4) {
        double C[NFP];
#pragma unroll
        for (int f = 0; f < NFP; ++f) C[f] = 0.0;
        if (ok) {
            if (j == 0) {
#pragma unroll
                for (int f = 0; f < NFP; ++f) C[f] = p1[f] + (mod + pk.abs_m1[f] * av);
            } else {
                // companion below the isochrone's first point: no flux, magnitude 99.999
                const double m2 = ((double)j / (double)Q) * m1;
                const bool dark2 = m2 < s_mass[0];
                int lo2; double t2;
                find_bracket(s_mass, h.n, m2, lo2, t2);
                const double *r2 = g_mags + (size_t)lo2 * NFP;
#pragma unroll
                for (int f = 0; f < NFP; ++f) {
                    const double p2 = dark2 ? B9_MAG_NOFLUX : lerp(r2[f], r2[NFP + f], t2);
                    const double comb = (-2.5 / LN10) * log_pos(exp_fast((-0.4 * LN10) * p1[f]) + exp_fast((-0.4 * LN10) * p2));
                    C[f] = comb + (mod + pk.abs_m1[f] * av);
                }
            }
        }
        double *row = out + L.o_rows + (((size_t)u * Q + j) * 16 + i16) * NFP;
#pragma unroll
        for (int f = 0; f < NFP; ++f) row[f] = C[f];
        // boxes: a NaN magnitude stays out of them (fmin / fmax ignore it); its term is dropped by the star loop's X < xcut
#pragma unroll
        for (int f = 0; f < NFP; ++f) {
            double lo = ok ? C[f] : __builtin_inf(), hi = ok ? C[f] : NEG_INF;
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) { lo = __builtin_fmin(lo, __shfl_xor(lo, o, 64)); hi = __builtin_fmax(hi, __shfl_xor(hi, o, 64)); }
            if (i16 == 0) box_store<NFP>(box32, out + L.o_box2 + ((size_t)u * Q + j) * 2 * NFP, out + L.o_box2f + ((size_t)u * Q + j) * NFP, f, lo, hi);
            lo64[f] = __builtin_fmin(lo64[f], lo); hi64[f] = __builtin_fmax(hi64[f], hi);
        }
    }
#pragma unroll
    for (int f = 0; f < NFP; ++f) {
        double lo = lo64[f], hi = hi64[f];
        lo = __builtin_fmin(lo, __shfl_xor(lo, 16, 64)); hi = __builtin_fmax(hi, __shfl_xor(hi, 16, 64));
        lo = __builtin_fmin(lo, __shfl_xor(lo, 32, 64)); hi = __builtin_fmax(hi, __shfl_xor(hi, 32, 64));
        if (lane == 0) { s_box[(jl * 2 + 0) * NFP + f] = lo; s_box[(jl * 2 + 1) * NFP + f] = hi; }
    }
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
        double lo = s_box[tid], hi = s_box[NFP + tid];
        for (int k = 1; k < 4; ++k) { lo = __builtin_fmin(lo, s_box[(k * 2) * NFP + tid]); hi = __builtin_fmax(hi, s_box[(k * 2 + 1) * NFP + tid]); }
        box_store<NFP>(box32, out + L.o_box1 + (size_t)c * 2 * NFP, out + L.o_box1f + (size_t)c * NFP, tid, lo, hi);
    }
}

// ------------------------------------------------------------------------------------------
// k_star_marg_wd: the WD-stage stars of the marginalised mode, one wavefront per (walker, star): lanes stride over
// the 8 iso_increm primary-mass steps in (AGB tip, M_wd_up], whose magnitudes k_marg_wd_table has tabulated per
// (walker, population, DA / DB); online log-sum-exp per lane, wavefront-shuffle merge.  Grid: (ceil(n_wd / 4), walkers).
// ------------------------------------------------------------------------------------------
struct Lse { double mx, sm; };      // online log-sum-exp:  value = mx + log(sm)
__device__ __forceinline__ void lse_add(Lse &a, double x)
{
    if (x == NEG_INF) return;
    if (x > a.mx) { a.sm = a.sm * exp_fast(a.mx - x) + 1.0; a.mx = x; }
    else a.sm += exp_fast(x - a.mx);
}

__device__ __forceinline__ Lse lse_merge(Lse a, Lse b)
{
    if (b.mx == NEG_INF) return a;
    if (a.mx == NEG_INF) return b;
    Lse r;
    if (a.mx >= b.mx) { r.mx = a.mx; r.sm = a.sm + b.sm * exp_fast(b.mx - a.mx); }
    else { r.mx = b.mx; r.sm = b.sm + a.sm * exp_fast(a.mx - b.mx); }
    return r;
}

// The WD-stage stars' node table.  Nothing about a node of THEIR integral depends on the star either: the magnitudes of a white
// dwarf of ZAMS mass m1_j = tip + j dM (j = 1 .. 8 K) under a walker's parameters depend on (walker, population, j, DA / DB)
// only -- so the WD chain (IFMR -> cooling tracks -> atmosphere) runs 2 x 8 K times per (walker, population) here instead of
// 8 K times per STAR and walker in k_star_marg_wd (a thousand WD-stage stars: 500 x fewer chains).  Rows hold the apparent
// magnitudes (modulus and absorption added, as the star loop formed them); lpm the mass prior of the node.
//   wtab = rows[wp][type][8 K][NFP] | lpm[wp][8 K]          grid (walkers * pops, ceil(8 K / 64)) x 128 threads (type = wave)
template <int NFP>
__global__ __launch_bounds__(128) void k_marg_wd_table(DevPack pk, const IsoHdr *__restrict__ hdr, const double *__restrict__ iso_data,
                                                       long long iso_stride, int mass_cap, int n_pops, const double *__restrict__ params,
                                                       int K, double *__restrict__ wtab, int n_wp)
{
    const int wp = blockIdx.x, w = wp / n_pops, type = threadIdx.x >> 6, steps = 8 * K;
    const int j = 1 + blockIdx.y * 64 + (threadIdx.x & 63);
    const IsoHdr h = hdr[wp];
    if (!h.valid) return;
    IsoView<NFP> is;
    is.n = h.n; is.tip = h.agb_tip; is.i_feh = h.i_feh; is.i_y = h.i_y; is.t_feh = h.t_feh; is.t_y = h.t_y;
    is.mass = iso_data + (size_t)wp * iso_stride; is.mags = is.mass + mass_cap;
    const double *par = params + (size_t)w * B9_NPARAM;
    const double dM = (pk.m_wd_up - is.tip) / steps;
    if (!(dM > 0.0) || j > steps) return;
    WdAxes ax;
    ax.log_age = pk.log_age;
    const int ny = pk.n_y > 1 ? 2 : 1;
    for (int df = 0; df < 2; ++df) for (int dy = 0; dy < 2; ++dy)
        ax.tips[df * 2 + dy] = pk.tips + (size_t)((is.i_feh + df) * pk.n_y + (is.i_y + (dy < ny ? dy : 0))) * pk.n_age;
    ax.wc_log_age_lds = nullptr; ax.wc_track = pk.wc_track; ax.wc_mass = pk.wc_mass; ax.wc_carb = pk.wc_carb;
    ax.at_log_teff = pk.at_log_teff; ax.at_logg = pk.at_logg;
    const double m1 = is.tip + dM * j, mod = par[B9_P_MOD], av = par[B9_P_ABS];
    double p[NFP];
    star_mags<NFP>(pk, ax, is, par, m1, type, p);
    double *row = wtab + (((size_t)wp * 2 + type) * steps + (j - 1)) * NFP;
#pragma unroll
    for (int f = 0; f < NFP; ++f) row[f] = p[f] + (mod + pk.abs_m1[f] * av);
    if (type == 0) wtab[(size_t)n_wp * 2 * steps * NFP + (size_t)wp * steps + (j - 1)] = log_prior_mass_dev(pk.log_mass_norm, m1);
}

// (block_x, w): the four-star group and the walker -- the plain launch's (blockIdx.x, blockIdx.y); `select` as in
// star_marg_body (its .tab is the WD-stage stars' node table), called by every thread of the workgroup
template <int NFP, int NPOPS, bool SAMPLE, class Select>
__device__ __forceinline__ void star_marg_wd_body(const DevPack &pk, const DevStars &st, int block_x, int w, int n_walkers,
                                                  const double *__restrict__ iso_data, long long iso_stride, int mass_cap,
                                                  double *__restrict__ partial, long long partial_stride, double *__restrict__ perstar,
                                                  int K, const MargSample &ms, Select select)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int k_wd = block_x * 4 + wave, n_wp = n_walkers * NPOPS;
    const bool live = k_wd < st.n_wd;
    select.issue(w);
    const int slot = st.wd_slot[live ? k_wd : 0], orig = st.perm[slot];
    double obs[NFP], wgt[NFP];
#pragma unroll
    for (int f = 0; f < NFP; ++f) { obs[f] = st.obs[B9_SIDX(NFP, f, slot)]; wgt[f] = st.w[B9_SIDX(NFP, f, slot)]; }
    const double c0m = st.c0m[slot], la = st.la[slot];
    const int wd_type = st.flags[slot] & 1;
    const MargSel sel = select.finish(w);
    // the workgroup's four stars leave ONE partial: their values summed in wave order ((v0 + v1) + (v2 + v3); a wave past
    // the last star adds 0) -- a catalogue of a thousand WD-stage stars had a thousand partials for every decision to read
    __shared__ double s_v4[4];
    double v_out = 0.0;
    const int k_part = (st.mg_pad >> 6) + block_x;
    const IsoHdr *__restrict__ const hdr = sel.hdr;
    const double *__restrict__ const wtab = sel.tab;
    const double *par = sel.params + (size_t)w * B9_NPARAM;
    IsoView<NFP> iso[NPOPS];
    double tip_min;
    const bool valid = load_iso_views<NFP, NPOPS>(hdr, iso_data, iso_stride, mass_cap, w, iso, tip_min);
    if (!valid) {          // (uniform over the workgroup: one walker)
        if (lane == 0 && live && perstar) perstar[(size_t)w * st.n + orig] = NEG_INF;
        if (tid == 0) partial[(size_t)w * partial_stride + k_part] = 0.0;
        return;
    }
    if (live) {
    double ll[NPOPS];
    Best best; best.key = NEG_INF; best.mass = 0.0; best.ratio = 0.0; best.pop = 0;
    const unsigned long long g_row = SAMPLE ? (unsigned long long)(ms.row0 + w) : 0ull;
    double lw_pop[2] = {0.0, 0.0};
    if (SAMPLE && NPOPS == 2) { const double lam = par[B9_P_LAMBDA]; lw_pop[0] = log(lam); lw_pop[1] = log1p(-lam); }
#pragma unroll
    for (int kp = 0; kp < NPOPS; ++kp) {
        const IsoView<NFP> &is = iso[kp];
        Lse acc; acc.mx = NEG_INF; acc.sm = 0.0;
        const int steps = 8 * K, wp = w * NPOPS + kp;
        const double dM = (pk.m_wd_up - is.tip) / steps;
        if (dM > 0.0) {
            const double log_w = log(dM);
            const double *__restrict__ const rows = wtab + (((size_t)wp * 2 + wd_type) * steps) * NFP;      // k_marg_wd_table's
            const double *__restrict__ const lpm = wtab + (size_t)n_wp * 2 * steps * NFP + (size_t)wp * steps;
            for (int j = 1 + lane; j <= steps; j += 64) {
                const double *__restrict__ const r = rows + (size_t)(j - 1) * NFP;
                double chi2 = 0.0;
#pragma unroll
                for (int f = 0; f < NFP; ++f) { const double d = r[f] - obs[f]; chi2 = fma(wgt[f] * d, d, chi2); }
                if (isfinite(chi2)) {
                    const double term = (lpm[j - 1] - 0.5 * chi2) + log_w;
                    lse_add(acc, term);
                    if (SAMPLE) {
                        const double m1 = is.tip + dM * j;
                        const double key_ = term + lw_pop[kp] + gumbel(ms.k0, ms.k1, g_row, (unsigned)orig, (unsigned long long)j, (unsigned)kp);
                        if (key_ > best.key) { best.key = key_; best.mass = m1; best.ratio = 0.0; best.pop = kp; }
                    }
                }
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            Lse b; b.mx = __shfl_down(acc.mx, o, 64); b.sm = __shfl_down(acc.sm, o, 64);
            acc = lse_merge(acc, b);
        }
        ll[kp] = (acc.mx == NEG_INF) ? NEG_INF : c0m + (acc.mx + log(acc.sm));
    }
    if (SAMPLE) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            Best b; b.key = __shfl_down(best.key, o, 64); b.mass = __shfl_down(best.mass, o, 64);
            b.ratio = __shfl_down(best.ratio, o, 64); b.pop = __shfl_down(best.pop, o, 64);
            if (b.key > best.key) best = b;
        }
    }
    if (lane == 0) {
        double l = ll[0];
        if (NPOPS == 2) { const double lam = par[B9_P_LAMBDA]; l = logaddexp(log(lam) + ll[0], log1p(-lam) + ll[NPOPS - 1]); }
        const double v = logaddexp(la, l);
        v_out = v;
        if (perstar) perstar[(size_t)w * st.n + orig] = v;
        if (SAMPLE) {
            const size_t o = (size_t)w * st.n + orig;
            const bool any = best.key != NEG_INF;
            ms.mass[o] = any ? best.mass : 0.0;
            ms.ratio[o] = any ? best.ratio : 0.0;
            ms.member[o] = (l == NEG_INF) ? 0.0 : exp(l - v);
            if (ms.pop) ms.pop[o] = any ? best.pop : 0;
        }
    }
    }
    if (lane == 0) s_v4[wave] = v_out;
    __syncthreads();
    if (tid == 0) partial[(size_t)w * partial_stride + k_part] = (s_v4[0] + s_v4[1]) + (s_v4[2] + s_v4[3]);
}

template <int NFP, int NPOPS, bool SAMPLE>
__global__ __launch_bounds__(256) void k_star_marg_wd(DevPack pk, DevStars st, const IsoHdr *__restrict__ hdr,
                                                      const double *__restrict__ iso_data, long long iso_stride,
                                                      int mass_cap, const double *__restrict__ params,
                                                      double *__restrict__ partial, long long partial_stride, double *__restrict__ perstar,
                                                      int K, MargSample ms, const double *__restrict__ wtab)
{
    star_marg_wd_body<NFP, NPOPS, SAMPLE>(pk, st, (int)blockIdx.x, (int)blockIdx.y, (int)gridDim.y, iso_data, iso_stride, mass_cap, partial, partial_stride,
                                          perstar, K, ms, MargSelPlain<false>{MargSel{hdr, params, wtab}, 0, 0, 0, {}});
}
