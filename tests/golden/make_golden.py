#!/usr/bin/env python3
"""Generates tests/golden/*.npz -- small input/output vectors for the hot path.

PROVENANCE: the reference (BASE-9) source is not mounted (/root/reference holds only a redirect
README), so these vectors are NOT reference outputs.  They are produced by this repo's CPU oracle
(oracle/b9_oracle.c) after it has been cross-checked against the independent numpy statement in
tests/numpy_ref.py, and they pin both the oracle and the HIP path against regressions.
"BASE-9 parity unpinned" applies to them as to everything else.

    python tests/golden/make_golden.py        # rewrites the fixtures
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))

import oracle  # noqa: E402
from base_amd import abi, mcmc, synth  # noqa: E402

CASES = {
    # name: (pack, n_filt, n_stars, wd_frac, n_y, n_pops, pack kwargs)
    "c0_girardi_3f_200": ("girardi", 3, 200, 0.0, 1, 1, dict(n_feh=4, n_age=8, n_eep=60)),
    "c1_dsed_8f_300": ("dsed", 8, 300, 0.0, 1, 1, dict(n_feh=4, n_age=8, n_eep=60)),
    "c3_parsec_8f_wd_300": ("parsec", 8, 300, 0.1, 1, 1, dict(n_feh=4, n_age=8, n_eep=60)),
    "c4_parsec_8f_2pop_300": ("parsec", 8, 300, 0.05, 3, 2, dict(n_feh=3, n_age=6, n_eep=50)),
    # ragged WD cooling tracks (b9_pack ABI 2: wc_n_age / wc_offset), a WD-rich cluster
    "c5_parsec_8f_wdragged_300": ("parsec", 8, 300, 0.3, 1, 1, dict(n_feh=4, n_age=8, n_eep=60, wd_ragged=True)),
}
PACK_KEYS = ["feh", "y", "log_age", "iso_first_eep", "iso_n_eep", "iso_offset", "mass", "mags", "abs_coeff",
             "wc_carb", "wc_mass", "wc_log_age", "wc_log_teff", "wc_log_radius", "at_logg", "at_log_teff", "at_mags"]
STAR_KEYS = ["obs", "sigma", "mass1", "mass_ratio", "clust_prior", "stage", "wd_type", "filter_prior_min", "filter_prior_max"]


def main(only=None):
    for name, (pk, nf, ns, wd, ny, npops, kw) in CASES.items():
        if only and name not in only:
            continue
        pack_d = synth.make_pack(pk, n_filt=nf, n_y=ny, **kw)
        truth = synth.default_params(pack_d)
        cl = synth.make_cluster(pack_d, ns, seed=9001, truth=truth, wd_frac=wd, n_pops=npops)
        pack, stars = abi.make_pack(pack_d), abi.make_stars(cl)
        priors = synth.default_priors(pack_d, truth, npops)
        options = abi.make_options(n_pops=npops)
        params = synth.walker_params(truth, 4, n_pops=npops)
        params[3, abi.P_FEH] = pack_d["feh"][-1] + 0.5     # one row outside the grid
        lp, ps = oracle.Oracle(pack, stars, priors, options).logpost(params, perstar=True)
        iso = oracle.derive_isochrone(oracle.load(), pack, params[0])
        # marginalised mode (2 sub-steps x 3 mass ratios), the sampleMass draws on the same grid, and a short
        # Metropolis chain (the host block runner over the oracle: what b9_mcmc_run_block must reproduce)
        orc = oracle.Oracle(pack, stars, priors, options)
        opt_m = abi.make_options(mode=abi.MODE_MARGINALISED, n_pops=npops, marg_iso_increm=2, marg_n_q=3)
        orc_m = oracle.Oracle(pack, stars, priors, opt_m)
        mlp, mps = orc_m.logpost(params, perstar=True)
        sm = orc_m.sample_mass(params[:3], seed=11, row0=5)
        free = np.array([abi.P_LOGAGE, abi.P_FEH, abi.P_MOD, abi.P_ABS] + ([abi.P_Y, abi.P_Y2, abi.P_LAMBDA] if npops == 2 else []))
        chol = np.diag([3e-4, 2e-3, 8e-4, 6e-4] + ([3e-4, 3e-4, 2e-3] if npops == 2 else []))
        start = params[:3].copy()
        chain = mcmc.HostBlockRunner(orc.logpost).run(start, orc.logpost(start), np.array([0, 1, 2]), free, chol, 2024, 40, 20)
        out = {f"pack_{k}": np.asarray(pack_d[k]) for k in PACK_KEYS}
        out.update({f"pack_{k}": np.asarray(pack_d[k]) for k in ("wc_n_age", "wc_offset") if k in pack_d})
        out.update(marg_logpost=mlp, marg_perstar=mps, sm_mass=sm[0], sm_ratio=sm[1], sm_member=sm[2], sm_pop=sm[3], sm_margin=sm[4],
                   chain_free=free, chain_chol=chol, chain_params=chain[0], chain_logpost=chain[1], chain_samples=chain[2],
                   chain_lps=chain[3], chain_accepted=chain[4])
        out.update({f"star_{k}": np.asarray(cl[k]) for k in STAR_KEYS})
        out.update(pack_n_filt=nf, pack_ifmr_id=pack_d["ifmr_id"], pack_m_wd_up=pack_d["m_wd_up"], pack_n_at_type=2,
                   n_pops=npops, prior_mean=np.array(list(priors.mean)), prior_var=np.array(list(priors.var)),
                   prior_age=np.array([priors.log_age_min, priors.log_age_max]),
                   params=params, logpost=lp, perstar=ps,
                   iso_first=iso[0], iso_mass=iso[1], iso_mags=iso[2], iso_tip=iso[3])
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print(name, lp)


if __name__ == "__main__":
    main(set(sys.argv[1:]))        # no arguments: every case; else only the named ones
