"""pytest configuration: the `gpu` marker, import path, shared problem builders."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


from base_amd import abi, synth  # noqa: E402


def build_problem(name="parsec", n_filt=8, n_stars=500, seed=9001, n_y=1, n_pops=1, wd_frac=0.0,
                  mode=abi.MODE_GIVEN_MASS, small=True, **pack_kw):
    """(pack_dict, cluster_dict, Pinned pack, Pinned stars, priors, options)"""
    kw = dict(n_feh=4, n_age=8, n_eep=90) if small else {}
    kw.update(pack_kw)
    pack_d = synth.make_pack(name, n_filt=n_filt, n_y=n_y, **kw)
    truth = synth.default_params(pack_d)
    cl = synth.make_cluster(pack_d, n_stars, seed=seed, truth=truth, wd_frac=wd_frac, n_pops=n_pops)
    pack = abi.make_pack(pack_d)
    stars = abi.make_stars(cl)
    priors = synth.default_priors(pack_d, truth, n_pops)
    options = abi.make_options(mode, n_pops, 4, 4)
    return pack_d, cl, pack, stars, priors, options


@pytest.fixture(scope="session")
def oracle_lib():
    import oracle
    return oracle.load()
