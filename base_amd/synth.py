"""Synthetic model packs and synthetic clusters.

No real Girardi / DSED / PARSEC / Bergeron tables exist in this environment, so the packs here
are smooth analytic stand-ins with the *shape* of the real grids (axes, EEP counts, ragged EEP
ranges, filter counts).  The cluster generator carries its own small numpy forward model
(`forward_mags`) -- deliberately independent of both the C oracle and the HIP kernels, so the
tests can use it as a third opinion.

Counterpart of the reference's simCluster/scatterCluster [RECALL], which are out of scope as
products; this is input generation for tests and benchmarks only.
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np

from . import abi

ABS_COEFF_8 = np.array([1.569, 1.321, 1.000, 0.751, 0.479, 0.282, 0.175, 0.112])   # UBVRIJHK-like
FILTERS_8 = ["U", "B", "V", "R", "I", "J", "H", "K"]

#: grid shapes standing in for the three MS/RGB packs named in BASELINE.json
PACK_SHAPES = {
    # name: (n_feh, feh range, n_age, age range, n_eep, age-spacing power)
    "girardi": dict(n_feh=6, feh=(-1.7, 0.3), n_age=50, age=(7.8, 10.2), n_eep=300, age_pow=1.0),
    "dsed":    dict(n_feh=9, feh=(-2.5, 0.5), n_age=52, age=(8.4, 10.18), n_eep=280, age_pow=1.0),
    "parsec":  dict(n_feh=10, feh=(-2.0, 0.5), n_age=60, age=(7.8, 10.2), n_eep=400, age_pow=1.15),
}


def _tip_mass(log_age, feh, y):
    # lifetime ~ M^-2.7, mildly metallicity/helium dependent
    return 10.0 ** ((10.0 + 0.08 * feh - 1.2 * (y - 0.27) - log_age) / 2.7)


def _iso_points(log_age, feh, y, eep_ids, n_eep_ref, n_filt):
    """Analytic isochrone: (mass[n], mags[n, n_filt]) for the given EEP ids."""
    x = eep_ids / float(n_eep_ref - 1)
    tip = _tip_mass(log_age, feh, y)
    xm = np.clip(x / 0.6, 0.0, 1.0)
    ms_mass = 0.1 + (0.92 * tip - 0.1) * xm ** 1.6
    xr = np.clip((x - 0.6) / 0.4, 0.0, None)
    mass = ms_mass + (tip - 0.92 * tip) * (1.0 - (1.0 - np.minimum(xr, 1.0)) ** 2) + 1e-4 * tip * xr
    logm = np.log10(mass)
    log_l = 3.6 * logm + 0.25 * logm ** 2 - 0.12 * feh + 2.6 * xr ** 1.5
    mbol = 4.75 - 2.5 * log_l
    temp = 0.55 * logm - 0.04 * feh + 0.3 * (y - 0.27) - 0.22 * xr ** 1.2      # ~ log Teff - 3.76
    lam = (np.arange(n_filt) - 2.0) / 3.0                                         # 0 at "V"
    mags = mbol[:, None] + temp[:, None] * (2.2 * lam[None, :]) + 0.15 * lam[None, :] ** 2 * temp[:, None] ** 2
    return mass, mags


def make_pack(name: str = "parsec", n_filt: int = 8, n_y: int = 1, ragged: bool = True,
              wd: bool = True, ifmr_id: int = abi.IFMR_WILLIAMS, n_feh: Optional[int] = None,
              n_age: Optional[int] = None, n_eep: Optional[int] = None, wd_ragged: bool = False) -> Dict:
    """Build a synthetic pack as a dict of numpy arrays (keys = b9_pack fields)."""
    sh = dict(PACK_SHAPES[name])
    if n_feh: sh["n_feh"] = n_feh
    if n_age: sh["n_age"] = n_age
    if n_eep: sh["n_eep"] = n_eep
    feh = np.linspace(sh["feh"][0], sh["feh"][1], sh["n_feh"])
    u = np.linspace(0.0, 1.0, sh["n_age"]) ** sh["age_pow"]
    log_age = sh["age"][0] + (sh["age"][1] - sh["age"][0]) * u
    y = np.array([0.27]) if n_y == 1 else np.linspace(0.23, 0.23 + 0.04 * (n_y - 1), n_y)
    n_iso = len(feh) * len(y) * len(log_age)
    first = np.zeros(n_iso, np.int32)
    count = np.zeros(n_iso, np.int32)
    offset = np.zeros(n_iso, np.int64)
    masses, mags = [], []
    off = 0
    k = 0
    for i_f, fe in enumerate(feh):
        for i_y, yy in enumerate(y):
            for i_a, la in enumerate(log_age):
                f0 = (i_f + 2 * i_a + i_y) % 3 if ragged else 0
                n = sh["n_eep"] - f0 - ((3 * i_f + i_a) % 4 if ragged else 0)
                ids = np.arange(f0, f0 + n, dtype=np.float64)
                m, mg = _iso_points(la, fe, yy, ids, sh["n_eep"], n_filt)
                first[k], count[k], offset[k] = f0, n, off
                masses.append(m); mags.append(mg)
                off += n; k += 1
    d = dict(name=name, n_filt=n_filt, feh=feh, y=y, log_age=log_age,
             iso_first_eep=first, iso_n_eep=count, iso_offset=offset,
             mass=np.concatenate(masses), mags=np.concatenate(mags, axis=0),
             abs_coeff=(ABS_COEFF_8[:n_filt] if n_filt <= 8 else np.linspace(1.6, 0.1, n_filt)),
             filters=(FILTERS_8[:n_filt] if n_filt <= 8 else [f"F{i}" for i in range(n_filt)]),
             ifmr_id=ifmr_id, m_wd_up=8.0)
    if wd:
        d.update(make_wd_tables(n_filt, ragged=wd_ragged))
    return d


def wd_cooling_tracks(pack: Dict):
    """The WD cooling model as a list of tracks in (carbonicity, mass) order: (log_age[n], log_teff[n], log_radius[n]) each.
    Accepts the ragged form (wc_n_age / wc_offset) and the rectangular one (one shared age axis)."""
    n_tracks = max(1, len(pack["wc_carb"])) * len(pack["wc_mass"])
    te, ra = np.asarray(pack["wc_log_teff"]).ravel(), np.asarray(pack["wc_log_radius"]).ravel()
    if "wc_n_age" in pack:
        age = np.asarray(pack["wc_log_age"])
        return [(age[o:o + n], te[o:o + n], ra[o:o + n]) for n, o in zip(np.asarray(pack["wc_n_age"]), np.asarray(pack["wc_offset"]))]
    n = len(pack["wc_log_age"])
    return [(np.asarray(pack["wc_log_age"]), te[t * n:(t + 1) * n], ra[t * n:(t + 1) * n]) for t in range(n_tracks)]


def make_wd_tables(n_filt: int, n_carb: int = 3, ragged: bool = False) -> Dict:
    """Synthetic WD cooling (Montgomery-like, carbonicity axis) + Bergeron-like atmospheres.  ragged: every
    (carbonicity, mass) track gets its own cooling-age axis -- different length, range and spacing -- as real cooling
    tracks have (b9_pack: wc_n_age / wc_offset)."""
    carb = np.linspace(0.2, 0.8, n_carb) if n_carb > 1 else np.array([0.38])
    wmass = np.linspace(0.4, 1.2, 9)

    def track(c, m, lage):
        log_teff = 5.05 - 0.27 * (lage - 6.0) - 0.012 * (lage - 6.0) ** 2 + 0.12 * (m - 0.6) + 0.05 * (c - 0.38)
        log_rad = np.log10(8.8e8) - np.log10(m / 0.6) / 3.0 + 0.02 * (log_teff - 4.0)
        return log_teff, log_rad

    if ragged:
        ages, tes, ras, n_age, offset = [], [], [], [], []
        off = 0
        for ic, c in enumerate(carb):
            for im, m in enumerate(wmass):
                n = 50 - 3 * ((2 * ic + im) % 5) + (7 if im == 4 else 0)
                lo, hi = 6.0 + 0.07 * ((ic + 2 * im) % 4), 10.3 - 0.05 * ((3 * ic + im) % 3)
                lage = lo + (hi - lo) * np.linspace(0.0, 1.0, n) ** (1.0 + 0.15 * (im % 3))
                te, ra = track(c, m, lage)
                ages.append(lage); tes.append(te); ras.append(ra); n_age.append(n); offset.append(off)
                off += n
        cool = dict(wc_carb=carb, wc_mass=wmass, wc_n_age=np.array(n_age, np.int32), wc_offset=np.array(offset, np.int64),
                    wc_log_age=np.concatenate(ages), wc_log_teff=np.concatenate(tes), wc_log_radius=np.concatenate(ras))
    else:
        lage = np.linspace(6.0, 10.3, 50)
        cc, mm, aa = np.meshgrid(carb, wmass, lage, indexing="ij")
        log_teff, log_rad = track(cc, mm, aa)
        cool = dict(wc_carb=carb, wc_mass=wmass, wc_log_age=lage, wc_log_teff=log_teff.ravel(), wc_log_radius=log_rad.ravel())
    logg = np.linspace(7.0, 9.5, 6)
    lteff = np.linspace(3.4, 5.1, 60)
    lam = (np.arange(n_filt) - 2.0) / 3.0
    at = np.zeros((2, len(logg), len(lteff), n_filt))
    for t in range(2):
        gg, tt = np.meshgrid(logg, lteff, indexing="ij")
        mbol = 12.6 - 10.0 * (tt - 4.0) + (gg - 8.0) * (2.5 / 1.5) + 0.15 * t
        col = -(tt - 4.0) * (1.4 - 0.2 * t)
        at[t] = mbol[..., None] + col[..., None] * lam[None, None, :] + 0.05 * t * lam[None, None, :] ** 2
    return dict(cool, at_logg=logg, at_log_teff=lteff, at_mags=at.ravel(), n_at_type=2)


# ------------------------------------------------------------------------------------------
# numpy forward model (independent restatement used only to *generate* observations)
# ------------------------------------------------------------------------------------------
def _bracket(ax, x):
    i = int(np.clip(np.searchsorted(ax, x, side="right") - 1, 0, len(ax) - 2))
    return i, (x - ax[i]) / (ax[i + 1] - ax[i])


def derive_isochrone(pack: Dict, log_age: float, feh: float, y: float):
    """numpy isochrone: returns (first_eep, mass[n], mags[n, nf]) or None outside the grid."""
    la, fe, yy = pack["log_age"], pack["feh"], pack["y"]
    if not (la[0] <= log_age <= la[-1] and fe[0] <= feh <= fe[-1]):
        return None
    if len(yy) > 1 and not (yy[0] <= y <= yy[-1]):
        return None
    ia, ta = _bracket(la, log_age)
    i_f, tf = _bracket(fe, feh)
    iy, ty = _bracket(yy, y) if len(yy) > 1 else (0, 0.0)
    ny = 2 if len(yy) > 1 else 1
    nA, nY = len(la), len(yy)
    nf = pack["n_filt"]
    corners = {}
    lo, hi = -10 ** 9, 10 ** 9
    for df in range(2):
        for dy in range(ny):
            for da in range(2):
                k = ((i_f + df) * nY + (iy + dy)) * nA + ia + da
                corners[(df, dy, da)] = k
                lo = max(lo, int(pack["iso_first_eep"][k]))
                hi = min(hi, int(pack["iso_first_eep"][k] + pack["iso_n_eep"][k]))
    if hi - lo < 2:
        return None

    def cols(k):
        s = int(pack["iso_offset"][k]) + lo - int(pack["iso_first_eep"][k])
        e = s + (hi - lo)
        return np.concatenate([pack["mags"].reshape(-1, nf)[s:e], pack["mass"][s:e, None]], axis=1)

    vf = []
    for df in range(2):
        vy = []
        for dy in range(ny):
            a, b = cols(corners[(df, dy, 0)]), cols(corners[(df, dy, 1)])
            vy.append(a + ta * (b - a))
        vf.append(vy[0] + ty * (vy[1] - vy[0]) if ny == 2 else vy[0])
    v = vf[0] + tf * (vf[1] - vf[0])
    return lo, v[:, nf].copy(), v[:, :nf].copy()


def _ifmr(pack, par, m):
    i = pack.get("ifmr_id", abi.IFMR_WILLIAMS)
    if i == abi.IFMR_WEIDEMANN:
        return np.interp(m, [1, 2, 3, 4, 5, 6, 7], [0.55, 0.60, 0.68, 0.79, 0.88, 0.95, 1.02])
    if i == abi.IFMR_WILLIAMS:
        return 0.339 + 0.129 * m
    if i == abi.IFMR_SALARIS_LIN:
        return 0.466 + 0.084 * m
    if i == abi.IFMR_SALARIS_PW:
        return np.where(m < 4.0, 0.134 * m + 0.331, 0.047 * m + 0.679)
    d = m - 3.0
    q = par[abi.P_IFMR_QUAD] if i == abi.IFMR_QUADRATIC else 0.0
    return par[abi.P_IFMR_INTERCEPT] + par[abi.P_IFMR_SLOPE] * d + q * d * d


def _lin(ax, x):
    """bracket with clamped index, extrapolating weights (vectorised)."""
    i = np.clip(np.searchsorted(ax, x, side="right") - 1, 0, len(ax) - 2)
    return i, (x - ax[i]) / (ax[i + 1] - ax[i])


def _wd_mags(pack, par, m, wd_type):
    nf = pack["n_filt"]
    la, fe, yy = pack["log_age"], pack["feh"], pack["y"]
    nA, nY = len(la), len(yy)
    i_f, tf = _bracket(fe, par[abi.P_FEH])
    iy, ty = _bracket(yy, par[abi.P_Y]) if nY > 1 else (0, 0.0)
    tips_all = pack["mass"][pack["iso_offset"] + pack["iso_n_eep"] - 1]

    def corner(ifeh, iyy):
        tips = tips_all[(ifeh * nY + iyy) * nA:(ifeh * nY + iyy) * nA + nA]
        out = np.interp(-m, -tips, la)            # tips descend with age
        heavy = m > tips[0]
        out = np.where(heavy, la[0] - 2.7 * np.log10(np.maximum(m, 1e-30) / tips[0]), out)
        return out

    vf = []
    for df in range(2):
        vy = [corner(i_f + df, iy + dy) for dy in range(2 if nY > 1 else 1)]
        vf.append(vy[0] + ty * (vy[1] - vy[0]) if nY > 1 else vy[0])
    prec = vf[0] + tf * (vf[1] - vf[0])
    log_age = par[abi.P_LOGAGE]
    not_yet = prec >= log_age
    wdm = _ifmr(pack, par, m)
    with np.errstate(invalid="ignore", divide="ignore"):
        cool = np.log10(np.maximum(10.0 ** log_age - 10.0 ** prec, 1e-300))
    nC, nM = len(pack["wc_carb"]), len(pack["wc_mass"])
    tracks = wd_cooling_tracks(pack)
    im, tm = _lin(pack["wc_mass"], wdm)
    if nC > 1:
        ic, tc = _lin(pack["wc_carb"], np.full_like(m, par[abi.P_CARBONICITY]))
    else:
        ic, tc = np.zeros_like(im), np.zeros_like(tm)

    def along(q, t_idx):
        """Quantity q (1 Teff, 2 radius) of every star along its track t_idx[i], at its cooling age (own axis per track)."""
        out = np.empty(len(m))
        for t in np.unique(t_idx):
            sel = t_idx == t
            age, tab = tracks[t][0], tracks[t][q]
            ia, ta = _lin(age, cool[sel])
            out[sel] = tab[ia] + ta * (tab[ia + 1] - tab[ia])
        return out

    def tri(q):
        def at_c(icc):
            a0, a1 = along(q, icc * nM + im), along(q, icc * nM + im + 1)
            return a0 + tm * (a1 - a0)
        if nC > 1:
            c0, c1 = at_c(ic), at_c(ic + 1)
            return c0 + tc * (c1 - c0)
        return at_c(ic)

    lteff, lrad = tri(1), tri(2)
    logg = 26.12302173752 + np.log10(wdm) - 2.0 * lrad
    nG, nTe = len(pack["at_logg"]), len(pack["at_log_teff"])
    at = pack["at_mags"].reshape(-1, nG, nTe, nf)
    tyv = np.where((np.asarray(wd_type) > 0) & (at.shape[0] > 1), 1, 0)
    it, tt = _lin(pack["at_log_teff"], lteff)
    ig, tg = _lin(pack["at_logg"], logg)
    g0 = at[tyv, ig, it] + tt[:, None] * (at[tyv, ig, it + 1] - at[tyv, ig, it])
    g1 = at[tyv, ig + 1, it] + tt[:, None] * (at[tyv, ig + 1, it + 1] - at[tyv, ig + 1, it])
    mags = g0 + tg[:, None] * (g1 - g0)
    mags[not_yet] = -4.0
    return mags


def _single_mags(pack, par, iso, m, wd_type):
    first, imass, imags = iso
    nf = pack["n_filt"]
    out = np.full((len(m), nf), abi.MAG_NOFLUX)
    tip = imass[-1]
    ms = (m >= imass[0]) & (m <= tip)
    for f in range(nf):
        out[ms, f] = np.interp(m[ms], imass, imags[:, f])
    wd = (m > tip) & (m <= pack["m_wd_up"])
    if wd.any() and len(pack.get("wc_mass", [])) >= 2:
        out[wd] = _wd_mags(pack, par, m[wd], np.asarray(wd_type)[wd])
    return out


def forward_mags(pack: Dict, par, mass1, mass_ratio, wd_type=None, pop: int = 0) -> np.ndarray:
    """Predicted apparent magnitudes [n, n_filt] (numpy forward model)."""
    par = np.asarray(par, dtype=np.float64)
    mass1 = np.asarray(mass1, dtype=np.float64)
    mass_ratio = np.asarray(mass_ratio, dtype=np.float64)
    wd_type = np.zeros(len(mass1), np.int32) if wd_type is None else np.asarray(wd_type)
    iso = derive_isochrone(pack, par[abi.P_LOGAGE], par[abi.P_FEH], par[abi.P_Y2 if pop else abi.P_Y])
    if iso is None:
        raise ValueError("parameters outside the model grid")
    par_p = par.copy()
    par_p[abi.P_Y] = par[abi.P_Y2 if pop else abi.P_Y]
    m1 = _single_mags(pack, par_p, iso, mass1, wd_type)
    out = m1.copy()
    b = mass_ratio > 0
    if b.any():
        m2 = _single_mags(pack, par_p, iso, mass1[b] * mass_ratio[b], wd_type[b])
        out[b] = -2.5 * np.log10(10.0 ** (-0.4 * m1[b]) + 10.0 ** (-0.4 * m2))
    out += par[abi.P_MOD] + (pack["abs_coeff"][None, :] - 1.0) * par[abi.P_ABS]
    return out


# ------------------------------------------------------------------------------------------
# clusters
# ------------------------------------------------------------------------------------------
def default_params(pack: Dict, log_age: float = 9.3, feh: float = -0.15, y: Optional[float] = None,
                   mod: float = 10.2, av: float = 0.12) -> np.ndarray:
    p = np.zeros(abi.B9_NPARAM)
    yy = pack["y"]
    p[abi.P_LOGAGE], p[abi.P_FEH], p[abi.P_MOD], p[abi.P_ABS] = log_age, feh, mod, av
    p[abi.P_Y] = float(yy[0] if len(yy) == 1 else (yy[0] + 0.3 * (yy[-1] - yy[0]))) if y is None else y
    p[abi.P_Y2] = float(yy[0] if len(yy) == 1 else (yy[0] + 0.75 * (yy[-1] - yy[0])))
    p[abi.P_LAMBDA] = 0.5
    p[abi.P_CARBONICITY] = 0.38
    p[abi.P_IFMR_INTERCEPT], p[abi.P_IFMR_SLOPE], p[abi.P_IFMR_QUAD] = 0.72, 0.11, 0.004
    return p


def make_cluster(pack: Dict, n_stars: int, seed: int, truth: Optional[np.ndarray] = None,
                 binary_frac: float = 0.3, wd_frac: float = 0.0, field_frac: float = 0.03,
                 n_pops: int = 1, unused_frac: float = 0.01) -> Dict:
    """Synthetic cluster drawn from the pack at `truth` (SURVEY 8d distributions)."""
    rng = np.random.default_rng(seed)
    truth = default_params(pack) if truth is None else np.asarray(truth, dtype=np.float64)
    nf = pack["n_filt"]
    iso = derive_isochrone(pack, truth[abi.P_LOGAGE], truth[abi.P_FEH], truth[abi.P_Y])
    tip = iso[1][-1]
    lo = max(0.15, iso[1][0] * 1.001)
    # Miller-Scalo log-normal IMF truncated to [lo, tip]
    mass1 = np.empty(n_stars)
    filled = 0
    while filled < n_stars:
        cand = 10.0 ** rng.normal(-1.02, 0.677, size=4 * n_stars + 64)
        cand = cand[(cand >= lo) & (cand <= tip * 0.9999)]
        take = min(len(cand), n_stars - filled)
        mass1[filled:filled + take] = cand[:take]
        filled += take
    stage = np.full(n_stars, abi.STAGE_MSRG, np.int32)
    n_wd = int(round(wd_frac * n_stars))
    if n_wd:
        idx = rng.choice(n_stars, n_wd, replace=False)
        mass1[idx] = rng.uniform(tip * 1.02, min(pack["m_wd_up"] * 0.98, tip * 2.5), n_wd)
        stage[idx] = abi.STAGE_WD
    q = np.where(rng.random(n_stars) < binary_frac, rng.uniform(0.05, 1.0, n_stars), 0.0)
    q[stage == abi.STAGE_WD] = 0.0
    wd_type = (rng.random(n_stars) < 0.2).astype(np.int32)
    pop = (rng.random(n_stars) >= truth[abi.P_LAMBDA]).astype(np.int32) if n_pops == 2 else np.zeros(n_stars, np.int32)
    pred = forward_mags(pack, truth, mass1, q, wd_type, pop=0)
    if n_pops == 2 and pop.any():
        pred[pop == 1] = forward_mags(pack, truth, mass1[pop == 1], q[pop == 1], wd_type[pop == 1], pop=1)
    bright = np.clip((pred - pred.min(axis=0)) / (np.ptp(pred, axis=0) + 1e-9), 0, 1)
    sigma = 0.01 + 0.04 * bright ** 2
    obs = pred + rng.normal(size=pred.shape) * sigma
    # field stars: uniform in the observed magnitude box
    is_field = rng.random(n_stars) < field_frac
    lo_f, hi_f = obs.min(axis=0) - 0.5, obs.max(axis=0) + 0.5
    obs[is_field] = rng.uniform(lo_f, hi_f, size=(int(is_field.sum()), nf))
    # a few unused filters
    drop = rng.random(obs.shape) < unused_frac
    drop[:, min(2, nf - 1)] = False
    sigma = np.where(drop, -1.0, sigma)
    prior = np.clip(rng.normal(0.9, 0.05, n_stars), 0.5, 0.999)
    return dict(n_filt=nf, obs=obs, sigma=sigma, mass1=mass1, mass_ratio=q, clust_prior=prior,
                stage=stage, wd_type=wd_type, filter_prior_min=lo_f, filter_prior_max=hi_f,
                truth=truth, is_field=is_field, pop=pop)


def default_priors(pack: Dict, truth: np.ndarray, n_pops: int = 1) -> abi.b9_priors:
    mean = np.array(truth, dtype=np.float64)
    var = np.zeros(abi.B9_NPARAM)
    var[abi.P_FEH], var[abi.P_MOD], var[abi.P_ABS] = 0.3 ** 2, 0.3 ** 2, 0.1 ** 2
    if len(pack["y"]) > 1:
        var[abi.P_Y] = 0.03 ** 2
        if n_pops == 2:
            var[abi.P_Y2] = 0.03 ** 2
    return abi.make_priors(mean, var, pack["log_age"][0], pack["log_age"][-1])


def walker_params(truth: np.ndarray, n_walkers: int, seed: int = 42, n_pops: int = 1,
                  scale: float = 1.0) -> np.ndarray:
    """Gaussian ball of parameter rows around the truth (seed 42)."""
    rng = np.random.default_rng(seed)
    p = np.tile(np.asarray(truth, dtype=np.float64), (n_walkers, 1))
    p[:, abi.P_LOGAGE] += rng.normal(0, 0.02 * scale, n_walkers)
    p[:, abi.P_FEH] += rng.normal(0, 0.05 * scale, n_walkers)
    p[:, abi.P_MOD] += rng.normal(0, 0.03 * scale, n_walkers)
    p[:, abi.P_ABS] = np.abs(p[:, abi.P_ABS] + rng.normal(0, 0.01 * scale, n_walkers))
    p[:, abi.P_CARBONICITY] += rng.normal(0, 0.02 * scale, n_walkers)
    if n_pops == 2:
        p[:, abi.P_LAMBDA] = np.clip(p[:, abi.P_LAMBDA] + rng.normal(0, 0.05 * scale, n_walkers), 0.02, 0.98)
    return np.ascontiguousarray(p)


# ------------------------------------------------------------------------------------------
# on-disk formats (docs/FORMATS.md): what the C++ host loaders in base_amd/host/ parse
# ------------------------------------------------------------------------------------------
def _g(x) -> str:
    return repr(float(x))          # shortest round-trip representation: files reload bit-exactly


def write_models_dir(pack: Dict, root: str, ms_name: Optional[str] = None, wd_name: str = "montgomery") -> str:
    """Write a synthetic pack as <root>/msrgb/<name>.model, absorption.dat, wd/cooling_*.dat, wd/atmos_D?.dat."""
    import os
    ms_name = ms_name or pack.get("name", "parsec")
    nf = pack["n_filt"]
    filters = list(pack["filters"])
    os.makedirs(os.path.join(root, "msrgb"), exist_ok=True)
    os.makedirs(os.path.join(root, "wd"), exist_ok=True)
    nY, nA = len(pack["y"]), len(pack["log_age"])
    mags = pack["mags"].reshape(-1, nf)
    with open(os.path.join(root, "msrgb", ms_name + ".model"), "w") as f:
        f.write("# synthetic MS/RGB grid (base_amd.synth); format: docs/FORMATS.md\n")
        f.write("%f " + " ".join(filters) + "\n")
        for i_f, fe in enumerate(pack["feh"]):
            for i_y, yy in enumerate(pack["y"]):
                f.write(f"%s [Fe/H]={_g(fe)} [alpha/Fe]=0.0 l/Hp=1.938 Y={_g(yy)}\n")
                for i_a, la in enumerate(pack["log_age"]):
                    k = (i_f * nY + i_y) * nA + i_a
                    off, n, e0 = int(pack["iso_offset"][k]), int(pack["iso_n_eep"][k]), int(pack["iso_first_eep"][k])
                    f.write(f"%a logAge={_g(la)}\n")
                    for j in range(n):
                        f.write(f"{e0 + j} {_g(pack['mass'][off + j])} " + " ".join(_g(v) for v in mags[off + j]) + "\n")
    with open(os.path.join(root, "absorption.dat"), "w") as f:
        f.write("# filter  A_filter/A_V\n")
        for name, c in zip(filters, pack["abs_coeff"]):
            f.write(f"{name} {_g(c)}\n")
    if len(pack.get("wc_mass", [])) >= 2:
        nM = len(pack["wc_mass"])
        tracks = wd_cooling_tracks(pack)
        with open(os.path.join(root, "wd", f"cooling_{wd_name}.dat"), "w") as f:
            f.write("# logCoolAge logTeff logRadius\n")
            for ic, c in enumerate(pack["wc_carb"]):
                f.write(f"%c carbonicity={_g(c)}\n")
                for im, m in enumerate(pack["wc_mass"]):
                    f.write(f"%m mass={_g(m)}\n")
                    age, te, ra = tracks[ic * nM + im]
                    for it in range(len(age)):
                        f.write(f"{_g(age[it])} {_g(te[it])} {_g(ra[it])}\n")
        nG, nTe = len(pack["at_logg"]), len(pack["at_log_teff"])
        at = pack["at_mags"].reshape(-1, nG, nTe, nf)
        for t in range(at.shape[0]):
            with open(os.path.join(root, "wd", f"atmos_{'DB' if t else 'DA'}.dat"), "w") as f:
                f.write("%f " + " ".join(filters) + "\n")
                for ig, g in enumerate(pack["at_logg"]):
                    f.write(f"%g logg={_g(g)}\n")
                    for it in range(nTe):
                        f.write(f"{_g(pack['at_log_teff'][it])} " + " ".join(_g(v) for v in at[t, ig, it]) + "\n")
    return root


def write_phot(cluster: Dict, filters, path: str, wd_type_column: bool = True) -> str:
    """[RECALL] .phot: id <filters> sig<filters> mass1 massRatio stage CMprior useDBI (+ wdType)."""
    nf = len(filters)
    obs = np.asarray(cluster["obs"]).reshape(-1, nf)
    sig = np.asarray(cluster["sigma"]).reshape(-1, nf)
    with open(path, "w") as f:
        f.write("id " + " ".join(filters) + " " + " ".join("sig" + x for x in filters) + " mass1 massRatio stage CMprior useDBI\n")
        for i in range(obs.shape[0]):
            row = [str(i + 1)] + [_g(v) for v in obs[i]] + [_g(v) for v in sig[i]] + [
                _g(cluster["mass1"][i]), _g(cluster["mass_ratio"][i]), str(int(cluster["stage"][i])),
                _g(cluster["clust_prior"][i]), "1"]
            if wd_type_column:
                row.append(str(int(cluster["wd_type"][i])))
            f.write(" ".join(row) + "\n")
    return path


def write_yaml(path: str, phot: str, models: str, out_base: str, truth, ms_model: str = "parsec",
               sigmas: Optional[Dict] = None, **extra) -> str:
    """A base9.yaml in the layout the reference uses [RECALL], restricted to what this build reads."""
    sg = dict(Fe_H=0.3, distMod=0.3, Av=0.1, Y=0.0, carbonicity=0.0)
    sg.update(sigmas or {})
    t = np.asarray(truth)
    txt = f"""general:
  files:
    photFile: "{phot}"
    outputFileBase: "{out_base}"
    modelDirectory: "{models}"
  main_sequence:
    msRgbModel: {ms_model}
  white_dwarfs:
    wdModel: montgomery
    ifmr: 1
    M_wd_up: 8.0
  cluster:
    starting:
      Fe_H: {_g(t[abi.P_FEH])}
      Av: {_g(t[abi.P_ABS])}
      Y: {_g(t[abi.P_Y])}
      carbonicity: {_g(t[abi.P_CARBONICITY])}
      logAge: {_g(t[abi.P_LOGAGE])}
      distMod: {_g(t[abi.P_MOD])}
    priors:
      means:
        Fe_H: {_g(t[abi.P_FEH])}
        distMod: {_g(t[abi.P_MOD])}
        Av: {_g(t[abi.P_ABS])}
        Y: {_g(t[abi.P_Y])}
        carbonicity: {_g(t[abi.P_CARBONICITY])}
      sigmas:
        Fe_H: {sg['Fe_H']}
        distMod: {sg['distMod']}
        Av: {sg['Av']}
        Y: {sg['Y']}
        carbonicity: {sg['carbonicity']}
    minMag: -99.0       # magnitude window of the stars to use ...
    maxMag: 99.0
    index: 0            # ... in this filter column
  seed: {extra.get('seed', 73)}
  verbose: 0
singlePopMcmc:
  stage2IterMax: {extra.get('burn', 1000)}
  runIter: {extra.get('run', 2000)}
  thin: {extra.get('thin', 1)}
gpu:
  walkers: {extra.get('walkers', 4)}
  block: 50
"""
    with open(path, "w") as f:
        f.write(txt)
    return path


# ------------------------------------------------------------------------------------------
# the five BASELINE.json configurations at ONE GPU's share (SURVEY 8d shapes; seeds 9001 + index)
# ------------------------------------------------------------------------------------------
BASELINE_CONFIGS = {
    # name: (pack, n_filt, n_stars, wd_frac, n_y, n_pops, walkers on one GPU, note)
    "C0": ("girardi", 3, 200, 0.0, 1, 1, 1, "200-star, Girardi-shaped, 3 filters, 1 chain (the reference's CPU plumbing case)"),
    "C1": ("dsed", 8, 10000, 0.0, 1, 1, 1, "10k-star, DSED-shaped, 8 filters, 1 chain, 1 GPU"),
    "C2": ("parsec", 8, 50000, 0.0, 1, 1, 8, "50k-star, PARSEC-shaped, 8 filters, 64 walkers / 8 GPUs -> 8 per GPU (bench.py workload)"),
    "C3": ("parsec", 8, 20000, 0.05, 1, 1, 1, "20k-star mixed MS+WD (5% WD: Bergeron-like atmospheres + IFMR), 8 filters, 1 GPU"),
    "C4": ("parsec", 8, 30000, 0.0, 3, 2, 8, "two-population 30k-star, 8 filters, 32 walkers / 4 GPUs -> 8 per GPU"),
}


def make_baseline_config(name: str, wd_ragged: bool = False) -> Dict:
    """Synthetic pack + cluster of BASELINE.json configs[int(name[1])], pinned into ABI structs.
    Returns dict(pack_d, cluster, pack, stars, priors, options, truth, walkers, n_pops, free, note)."""
    pk, nf, ns, wd, ny, npops, walkers, note = BASELINE_CONFIGS[name]
    pack_d = make_pack(pk, nf, n_y=ny, wd_ragged=wd_ragged)
    truth = default_params(pack_d)
    cl = make_cluster(pack_d, ns, seed=9001 + int(name[1]), truth=truth, wd_frac=wd, n_pops=npops)
    free = (abi.P_LOGAGE, abi.P_FEH, abi.P_MOD, abi.P_ABS) + ((abi.P_Y, abi.P_Y2, abi.P_LAMBDA) if npops == 2 else ())
    return dict(pack_d=pack_d, cluster=cl, pack=abi.make_pack(pack_d), stars=abi.make_stars(cl),
                priors=default_priors(pack_d, truth, npops), options=abi.make_options(n_pops=npops), truth=truth,
                walkers=walkers, n_pops=npops, free=free, note=note)
