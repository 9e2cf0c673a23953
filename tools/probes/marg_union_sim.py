#!/usr/bin/env python3
"""CPU simulation (numpy, no GPU): for the bench cluster and one parameter row, which marginalisation nodes matter for
each star (log-term within CUT e-folds of max(its best term, field-star term)), and how large the UNION of those node
sets is over 64 slot-neighbouring stars -- the work of a lane-per-star layout in which a wave walks the union."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from base_amd import abi, synth

K, Q, CUT = 4, 4, 40.0
n_stars = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
pack = synth.make_pack("parsec", 8); truth = synth.default_params(pack)
cl = synth.make_cluster(pack, n_stars, seed=9003, truth=truth)
row = synth.walker_params(truth, 8, seed=42, scale=0.05)[3]
first, imass, imags = synth.derive_isochrone(pack, row[abi.P_LOGAGE], row[abi.P_FEH], row[abi.P_Y])
n_eep = len(imass); nf = 8
# nodes
e = np.repeat(np.arange(n_eep - 1), K); s = np.tile(np.arange(K), n_eep - 1)
a = imass[e]; d = imass[e + 1] - a; dM = d / K; m1 = a + s * dM; t1 = (m1 - a) / d
p1 = imags[e] + t1[:, None] * (imags[e + 1] - imags[e])
mu, sg = -1.02, 0.67729
lp = -0.5 * ((np.log10(m1) - mu) / sg) ** 2 - np.log(m1) - np.log(np.log(10.0))   # (norm constant dropped: same for all)
base = lp + np.log(dM / Q)
comb = np.empty((Q, len(m1), nf)); comb[0] = p1
for j in range(1, Q):
    m2 = j / Q * m1
    p2 = np.stack([np.interp(m2, imass, imags[:, f]) for f in range(nf)], 1)
    p2[m2 < imass[0]] = 99.999
    comb[j] = -2.5 * np.log10(10 ** (-0.4 * p1) + 10 ** (-0.4 * p2))
shift = row[abi.P_MOD] + (pack["abs_coeff"] - 1.0) * row[abi.P_ABS]
comb += shift
N = len(m1); print("nodes", N, "chunks", (N + 63) // 64)
# slot order: binaries by mass, then singles by mass
q = cl["mass_ratio"]; order = np.lexsort((cl["mass1"], ~(q > 0)))
obs, sig, prior = cl["obs"][order], cl["sigma"][order], cl["clust_prior"][order]
w = np.where(sig > 0, 1.0 / np.maximum(sig, 1e-30) ** 2, 0.0)
g = np.where(sig > 0, -0.5 * np.log(2 * np.pi * np.maximum(sig, 1e-30) ** 2), 0.0).sum(1)
log_fs = -np.log(cl["filter_prior_max"] - cl["filter_prior_min"]).sum()
la = np.log1p(-prior) + log_fs; c0m = np.log(prior) + g
lmn = 0.0
wanted_nodes = np.zeros(n_stars, int); wanted_nodes_fs = np.zeros(n_stars, int)
union_nodes, union_nodes_fs, union_chunks_fs, union_pnodes_fs = [], [], [], []
own_chunks_fs = np.zeros(n_stars, int)
for c0 in range(0, n_stars, 64):
    sl = slice(c0, min(c0 + 64, n_stars))
    dd = comb[None] - obs[sl, None, None, :]                  # [star, j, node, f]
    chi = (w[sl, None, None, :] * dd * dd).sum(-1)
    term = base[None, None, :] - 0.5 * chi                    # [star, j, node]
    best = term.reshape(term.shape[0], -1).max(1)
    want = term >= (best - CUT)[:, None, None]
    ref = np.maximum(best, la[sl] - c0m[sl])                   # the field-star term bounds what matters
    want_fs = term >= (ref - CUT)[:, None, None]
    wanted_nodes[sl] = want.sum((1, 2)); wanted_nodes_fs[sl] = want_fs.sum((1, 2))
    union_nodes.append(want.any(0).sum()); union_nodes_fs.append(want_fs.any(0).sum())
    pn = want_fs.any(1)                                        # [star, node]: any j
    union_pnodes_fs.append(pn.any(0).sum())
    ch = np.add.reduceat(pn, np.arange(0, N, 64), axis=1) > 0
    own_chunks_fs[sl] = ch.sum(1); union_chunks_fs.append(ch.any(0).sum())
print("per star: wanted (node, j) terms           mean %.1f  median %.0f  p99 %.0f" % (wanted_nodes.mean(), np.median(wanted_nodes), np.percentile(wanted_nodes, 99)))
print("per star, with the field-star floor:        mean %.1f  median %.0f  p99 %.0f" % (wanted_nodes_fs.mean(), np.median(wanted_nodes_fs), np.percentile(wanted_nodes_fs, 99)))
print("per star own chunks (field floor):          mean %.2f" % own_chunks_fs.mean())
u = np.array(union_nodes); uf = np.array(union_nodes_fs); uc = np.array(union_chunks_fs); up = np.array(union_pnodes_fs)
print("union over a 64-star wave, (node, j) terms: mean %.0f  max %d   (of %d)" % (u.mean(), u.max(), N * Q))
print("  with the field-star floor:                mean %.0f  max %d" % (uf.mean(), uf.max()))
print("  primary nodes in the union (any j):       mean %.0f  max %d" % (up.mean(), up.max()))
print("  64-node chunks in the union:              mean %.2f  max %d" % (uc.mean(), uc.max()))

# ---- unit-granular (16 nodes x 1 mass ratio) unions under different slot orders
def unions(order_idx, label):
    ob, sg, pr = cl["obs"][order_idx], cl["sigma"][order_idx], cl["clust_prior"][order_idx]
    ww = np.where(sg > 0, 1.0 / np.maximum(sg, 1e-30) ** 2, 0.0)
    gg = np.where(sg > 0, -0.5 * np.log(2 * np.pi * np.maximum(sg, 1e-30) ** 2), 0.0).sum(1)
    la_ = np.log1p(-pr) + log_fs; c0 = np.log(pr) + gg
    tot_units, tot_live, n_ch = 0, 0, 0
    per = []
    for c0i in range(0, len(order_idx), 64):
        sl = slice(c0i, min(c0i + 64, len(order_idx)))
        dd = comb[None] - ob[sl, None, None, :]
        chi = (ww[sl, None, None, :] * dd * dd).sum(-1)
        term = base[None, None, :] - 0.5 * chi
        best = term.reshape(term.shape[0], -1).max(1)
        ref = np.maximum(best, la_[sl] - c0[sl])
        want = term >= (ref - CUT)[:, None, None]                # [star, j, node]
        any_ = want.any(0)                                        # [j, node]
        pad = (-N) % 16
        u = np.pad(any_, ((0, 0), (0, pad))).reshape(Q, -1, 16).any(-1)
        per.append(u.sum()); tot_live += want.sum(); n_ch += 1
    per = np.array(per)
    print(f"{label:40s} units/chunk mean {per.mean():6.1f} p90 {np.percentile(per,90):5.0f} max {per.max():4d}; live share {tot_live / (per.sum() * 16 * 64):.3f}")
    return per

isb = q > 0
sing = np.where(~isb)[0]; binr = np.where(isb)[0]
o_s = sing[np.argsort(cl["mass1"][sing])]; o_b = binr[np.argsort(cl["mass1"][binr])]
pb = unions(o_b, "binaries by mass"); ps = unions(o_s, "singles by mass")
jb = np.minimum(np.round(q[binr] * Q).astype(int), Q - 1)
o_b2 = binr[np.lexsort((cl["mass1"][binr], jb))]
unions(o_b2, "binaries by (round(q Q), mass)")
alls = np.argsort(cl["mass1"]); unions(alls, "all stars by mass (no split)")
vmag = cl["obs"][:, 2]; unions(np.argsort(vmag), "all stars by observed magnitude, filter 2")
ob = cl["obs"]; sgm = cl["sigma"]
valid = sgm > 0
meanmag = np.where(valid, ob, 0).sum(1) / np.maximum(valid.sum(1), 1)
unions(np.argsort(meanmag), "by mean magnitude over valid filters")
for f in (0, 4, 7): unions(np.argsort(ob[:, f]), f"by observed magnitude, filter {f}")
X = ob - ob.mean(0); u_, s_, vt = np.linalg.svd(X[valid.all(1)], full_matrices=False)
unions(np.argsort(X @ vt[0]), "by first principal component")
# two-key: coarse magnitude bins, colour inside (snake)
col = ob[:, 0] - ob[:, 7]
mb = np.floor((meanmag - meanmag.min()) / 0.05).astype(int)
unions(np.lexsort((col, mb)), "0.05-mag bins of mean magnitude, colour inside")
