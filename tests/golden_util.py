"""Loads a tests/golden/*.npz fixture back into ABI structs."""
import glob
import os

import numpy as np

from base_amd import abi

HERE = os.path.dirname(os.path.abspath(__file__))


def names():
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(HERE, "golden", "*.npz")))


def load(name):
    z = np.load(os.path.join(HERE, "golden", name + ".npz"))
    pack_d = {k[5:]: z[k] for k in z.files if k.startswith("pack_")}
    for k in ("n_filt", "ifmr_id", "n_at_type"):
        pack_d[k] = int(pack_d[k])
    pack_d["m_wd_up"] = float(pack_d["m_wd_up"])
    cl = {k[5:]: z[k] for k in z.files if k.startswith("star_")}
    cl["n_filt"] = pack_d["n_filt"]
    priors = abi.make_priors(z["prior_mean"], z["prior_var"], float(z["prior_age"][0]), float(z["prior_age"][1]))
    options = abi.make_options(n_pops=int(z["n_pops"]))
    return z, pack_d, cl, abi.make_pack(pack_d), abi.make_stars(cl), priors, options
