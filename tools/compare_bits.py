"""Bit-for-bit comparison of two builds of libbase9hip.so: sampler chains (tree depths 3 / 2 and the one-step launch, one and two
populations, WD stars) and marginalised per-star values of the shipped library against build/variants/lib_oldsum.so (or any variant
named in the script) -- used when a change must not move a single bit (e.g. the cross-lane sums on DPP / permlane moves).
    python tools/build_variant.py oldsum   (on the tree to compare against);   python tools/compare_bits.py   (GPU box)"""
import os, sys, subprocess, pickle
sys.path.insert(0, os.getcwd())
if len(sys.argv) > 1:
    import numpy as np
    from base_amd import abi, engine, mcmc, synth
    out = {}
    for name, (pk, nf, ns, wd, ny, npops, W) in {"C1": ("dsed", 8, 10000, 0.0, 1, 1, 1), "C2s": ("parsec", 8, 20000, 0.0, 1, 1, 8), "C3": ("parsec", 8, 20000, 0.05, 1, 1, 1), "C4s": ("parsec", 8, 9000, 0.02, 3, 2, 8), "W2": ("parsec", 5, 30000, 0.01, 1, 1, 2)}.items():
        pack_d = synth.make_pack(pk, nf, n_y=ny); truth = synth.default_params(pack_d)
        cl = synth.make_cluster(pack_d, ns, seed=11, truth=truth, wd_frac=wd, n_pops=npops)
        eng = engine.Engine(abi.make_pack(pack_d), abi.make_stars(cl), synth.default_priors(pack_d, truth, npops), abi.make_options(n_pops=npops))
        free = np.array(mcmc.DEFAULT_FREE if npops == 1 else mcmc.DEFAULT_FREE + (abi.P_Y, abi.P_Y2, abi.P_LAMBDA), dtype=np.int32)
        start = synth.walker_params(truth, W, seed=7, n_pops=npops, scale=0.02); lp = eng.logpost(start)
        chol = np.diag([mcmc.DEFAULT_STEP[int(k)] for k in free]) * 0.3
        r = eng.mcmc_run_block(start, lp, np.arange(W, dtype=np.int32), free, chol, 7, 0, 60)
        out[name] = (lp.tobytes(), r[0].tobytes(), r[1].tobytes(), r[2].tobytes(), r[3].tobytes(), r[4], eng.step_depth(W))
        # marginalised logpost too
        engm = engine.Engine(abi.make_pack(pack_d), abi.make_stars({k: (v[:3000] if hasattr(v, "__len__") and len(v) == ns else v) for k, v in cl.items()}), synth.default_priors(pack_d, truth, npops), abi.make_options(abi.MODE_MARGINALISED, npops, 4, 4))
        out[name + "m"] = engm.logpost(start, perstar=True)[1].tobytes() if True else None
    pickle.dump(out, open(sys.argv[1], "wb"))
else:
    env = dict(os.environ)
    subprocess.check_call([sys.executable, __file__, "/tmp/new.pkl"], env=env)
    env["B9_HIP_LIB"] = "build/variants/lib_oldsum.so"
    subprocess.check_call([sys.executable, __file__, "/tmp/old.pkl"], env=env)
    a, b = pickle.load(open("/tmp/new.pkl", "rb")), pickle.load(open("/tmp/old.pkl", "rb"))
    for k in a:
        print(k, "identical" if a[k] == b[k] else "DIFFERENT", a[k][-1] if not k.endswith("m") else "")
