#!/usr/bin/env python3
"""What the engine does on the round-3 fuzz seeds, chain by chain, next to the compiled reference's verdict (tests/golden/fuzz_seed_*.npz)."""
import glob, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import level_errors, load_golden, objects_from, problem_dict
from rslmtoasa_amd import _lib
from rslmtoasa_amd.recursion import Recursion

for f in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "fuzz_seed_*.npz")), key=lambda s: int(s.split("_")[-1][:-4])):
    z = load_golden(os.path.basename(f)[:-4])
    p = problem_dict(z)
    lld = int(z["lld"])
    pairs = z.get("pairs")
    nunit = len(pairs) if pairs is not None else len(z["irec"])
    ok = np.array([z["ok_t%d" % t] for t in z["threads"]])
    for u in range(nunit):
        ham, lat, ctl, en = objects_from(p, z["irec"][u:u + 1] if pairs is None else [1], lld)
        if pairs is not None:
            lat.ijpair = pairs[u:u + 1]
        rec = Recursion(ham, lat, ctl, en, device=0)
        try:
            if pairs is None:
                rec.recur_b(); a, b = rec.a_b[:, :, :, :1], rec.b2_b[:, :, :, :1]; sl = slice(u, u + 1)
            else:
                rec.recur_b_ij(); a, b = rec.a_b, rec.b2_b; sl = slice(4 * u, 4 * u + 4)
            out = "finite=%s" % bool(np.isfinite(a).all() and np.isfinite(b).all())
            if ok[:, u].all():
                ea, eb = level_errors(a, z["a_b_ref"][:, :, :, sl]), level_errors(b, z["b2_b_ref"][:, :, :, sl])
                out += " err a %.1e b2 %.1e | ref spread a %.1e b2 %.1e" % (np.nanmax(ea), np.nanmax(eb), z["a_b_spread"][:, sl].max(), z["b2_b_spread"][:, sl].max())
        except _lib.RsrecError as e:
            out = "RsrecError: %s" % e
        rec.close()
        print("seed %6d unit %d reference ok=%s -> engine %s" % (int(z["seed"]), u, ok[:, u].tolist(), out), flush=True)
