#!/usr/bin/env python3
"""CPU simulation: units (16 nodes x 1 mass ratio) a 64-star chunk evaluates under the kernel's box test, with the pruning
reference (a) the field floor only, (b) the star's final maximum -- against the exact set of units holding a wanted term."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from base_amd import abi, synth
K, Q, CUT = 4, 4, 40.0
n_stars = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
pack = synth.make_pack("parsec", 8); truth = synth.default_params(pack)
cl = synth.make_cluster(pack, n_stars, seed=9003, truth=truth)
row = synth.walker_params(truth, 8, seed=42, scale=0.05)[3]
first, imass, imags = synth.derive_isochrone(pack, row[abi.P_LOGAGE], row[abi.P_FEH], row[abi.P_Y])
n_eep = len(imass); nf = 8
e = np.repeat(np.arange(n_eep - 1), K); s = np.tile(np.arange(K), n_eep - 1)
a = imass[e]; d = imass[e + 1] - a; dM = d / K; m1 = a + s * dM; t1 = (m1 - a) / d
p1 = imags[e] + t1[:, None] * (imags[e + 1] - imags[e])
base = -0.5 * ((np.log10(m1) + 1.02) / 0.67729) ** 2 - np.log(m1) - np.log(np.log(10.0)) + np.log(dM / Q)
comb = np.empty((Q, len(m1), nf)); comb[0] = p1
for j in range(1, Q):
    m2 = j / Q * m1
    p2 = np.stack([np.interp(m2, imass, imags[:, f]) for f in range(nf)], 1); p2[m2 < imass[0]] = 99.999
    comb[j] = -2.5 * np.log10(10 ** (-0.4 * p1) + 10 ** (-0.4 * p2))
comb += row[abi.P_MOD] + (pack["abs_coeff"] - 1.0) * row[abi.P_ABS]
N = len(m1); pad = (-N) % 16; NU = (N + pad) // 16
cpad = np.pad(comb, ((0, 0), (0, pad), (0, 0)), constant_values=np.nan).reshape(Q, NU, 16, nf)
lo2, hi2 = np.nanmin(cpad, 2), np.nanmax(cpad, 2)                       # [Q, NU, nf]
nbmin = np.nanmin(np.pad(-2 * base, (0, pad), constant_values=np.nan).reshape(NU, 16), 1)
ob, sg, pr = cl["obs"], cl["sigma"], cl["clust_prior"]
w = np.where(sg > 0, 1.0 / np.maximum(sg, 1e-30) ** 2, 0.0)
g = np.where(sg > 0, -0.5 * np.log(2 * np.pi * np.maximum(sg, 1e-30) ** 2), 0.0).sum(1)
log_fs = -np.log(cl["filter_prior_max"] - cl["filter_prior_min"]).sum()
floor = np.log1p(-pr) + log_fs - (np.log(pr) + g)
X = ob - ob.mean(0); vt = np.linalg.svd(X, full_matrices=False)[2]
orders = {"mass (binaries, then singles)": np.lexsort((cl["mass1"], ~(cl["mass_ratio"] > 0))), "first principal component": np.argsort(X @ vt[0])}
for name, order in orders.items():
    tot = np.zeros(4); nch = 0
    for c0 in range(0, n_stars, 64):
        idx = order[c0:c0 + 64]
        o, ww, fl = ob[idx], w[idx], floor[idx]
        dd = comb[None] - o[:, None, None, :]
        term = base[None, None, :] - 0.5 * (ww[:, None, None, :] * dd * dd).sum(-1)
        best = term.reshape(len(idx), -1).max(1)
        ref = np.maximum(best, fl)
        want = term >= (ref - CUT)[:, None, None]
        wu = np.pad(want.any(0), ((0, 0), (0, pad))).reshape(Q, NU, 16).any(-1)
        cl_ = np.clip(o[:, None, None, :], lo2[None], hi2[None])
        lb = (ww[:, None, None, :] * (o[:, None, None, :] - cl_) ** 2).sum(-1) + nbmin[None, None, :]      # [star, Q, NU]
        pa = (lb < (-2 * fl + 2 * CUT)[:, None, None]).any(0)
        pb = (lb < (-2 * ref + 2 * CUT)[:, None, None]).any(0)
        # (c) the reference a lane holds when the unit is visited in ascending order: running max over earlier units' terms
        tu = np.pad(term, ((0, 0), (0, 0), (0, pad)), constant_values=-np.inf).reshape(len(idx), Q, NU, 16).max(-1)     # best term per unit
        run = np.maximum.accumulate(np.maximum(tu.max(1), fl[:, None]), axis=1)                                         # after chunk-unit u (all j)
        prev = np.concatenate([fl[:, None], run[:, :-1]], 1)
        pc = (lb < (-2 * prev[:, None, :] + 2 * CUT)).any(0)
        tot += [wu.sum(), pa.sum(), pb.sum(), pc.sum()]; nch += 1
    print(f"{name:32s} units per chunk: wanted {tot[0]/nch:5.1f} | boxes, floor only {tot[1]/nch:5.1f} | boxes, final max {tot[2]/nch:5.1f} | boxes, running max (ascending) {tot[3]/nch:5.1f}")
# the C++ staging's ordering, restated (used filters only, column means, power iteration from ones)
used = (sg > 0) & np.isfinite(ob)
mean = np.where(used, ob, 0).sum(0) / np.maximum(used.sum(0), 1)
Xc = np.where(used, ob - mean, 0.0)
cov = Xc.T @ Xc
pc = np.ones(nf)
for _ in range(200):
    pc = cov @ pc; pc /= np.linalg.norm(pc)
print("pc", np.round(pc, 3), "svd", np.round(vt[0], 3))
key = Xc @ pc
order = np.argsort(key, kind="stable")
tot = np.zeros(2); nch = 0
for c0 in range(0, n_stars, 64):
    idx = order[c0:c0 + 64]
    o, ww, fl = ob[idx], w[idx], floor[idx]
    cl_ = np.clip(o[:, None, None, :], lo2[None], hi2[None])
    lb = (ww[:, None, None, :] * (o[:, None, None, :] - cl_) ** 2).sum(-1) + nbmin[None, None, :]
    pa = (lb < (-2 * fl + 2 * CUT)[:, None, None]).any(0)
    tot += [pa.sum(), 0]; nch += 1
print("C++-style PC1 order: boxes, floor only", tot[0] / nch)
key2 = (Xc @ pc) / np.maximum((used * pc ** 2).sum(1), 1e-300)
order = np.argsort(key2, kind="stable")
tot = 0.0; nch = 0
for c0 in range(0, n_stars, 64):
    idx = order[c0:c0 + 64]
    o, ww, fl = ob[idx], w[idx], floor[idx]
    cl_ = np.clip(o[:, None, None, :], lo2[None], hi2[None])
    lb = (ww[:, None, None, :] * (o[:, None, None, :] - cl_) ** 2).sum(-1) + nbmin[None, None, :]
    tot += (lb < (-2 * fl + 2 * CUT)[:, None, None]).any(0).sum(); nch += 1
print("least-squares coefficient on the used filters: boxes, floor only", tot / nch)
