#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY -- copies the DATA of a few of the reference's end-to-end cases into tests/golden/scf/.

For each case: the input data files of the reference's test case directory (input.nml + <label>.nml), the namelist
patch of tests/scf/cases.json, and the expected values of tests/scf/references/<name>/ref.json (or, for the legacy
regression case, etot/ws_r/vmad of Fe.nml.ref).  These are inputs and expected outputs, not source.
Runs only in the build container (needs /root/reference).
"""
import json
import os
import re
import shutil

REF = os.environ.get("RSREC_REFERENCE", "/root/reference")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden", "scf")

SCF = ["Example_bulk_bccFe_nsp2_block", "Example_bulk_bccFe_nsp2_block_hoh", "Example_bulk_bccFe_nsp2_chebyshev",
       "Example_bulk_bccFe_nsp4_block", "Example_impurity_B2FeCo_block_hoh", "Example_surface_fccCu001_block_hoh"]


def fortran_value(v):
    if isinstance(v, bool):
        return ".true." if v else ".false."
    if isinstance(v, str):
        return "'%s'" % v
    return repr(v)


def main():
    os.makedirs(OUT, exist_ok=True)
    cases = {c["name"]: c for c in json.load(open(os.path.join(REF, "tests/scf/cases.json")))["cases"]}
    manifest = {}
    for name in SCF:
        c = cases[name]
        src = os.path.join(REF, "tests/scf/cases", c["case"])
        dst = os.path.join(OUT, "inputs", c["case"].replace("/", "_"))
        os.makedirs(dst, exist_ok=True)
        for fn in os.listdir(src):
            if fn.endswith(".nml"):
                shutil.copy(os.path.join(src, fn), os.path.join(dst, fn))
                os.chmod(os.path.join(dst, fn), 0o644)
        ref = json.load(open(os.path.join(REF, "tests/scf/references", name, "ref.json")))
        patch = {g: {k: fortran_value(v) for k, v in kv.items()} for g, kv in c["namelists"].items()}
        manifest[name] = {"inputs": os.path.relpath(dst, OUT), "patch": patch, "expected": ref,
                          "abs_tol": c.get("abs_tol", 1e-6), "rel_tol": c.get("rel_tol", 1e-6), "source": "tests/scf/cases.json + tests/scf/references/%s/ref.json" % name}
    # legacy regression case: the only end-to-end test of the scalar Lanczos path
    src = os.path.join(REF, "tests/regression/bccFe_lanczos")
    dst = os.path.join(OUT, "inputs", "regression_bccFe_lanczos")
    os.makedirs(dst, exist_ok=True)
    for fn in ("input.nml", "Fe.nml"):
        shutil.copy(os.path.join(src, fn), os.path.join(dst, fn))
        os.chmod(os.path.join(dst, fn), 0o644)
    txt = open(os.path.join(src, "Fe.nml.ref")).read()
    exp = {k: float(re.search(r"(?im)^\s*%s\s*=\s*([-+0-9.eEdD]+)" % k, txt).group(1).replace("D", "E").replace("d", "e")) for k in ("etot", "ws_r", "vmad")}
    manifest["Regression_bccFe_lanczos"] = {"inputs": os.path.relpath(dst, OUT), "patch": {}, "expected": {"nml": {"Fe_out.nml": exp}},
                                            "abs_tol": 1e-6, "rel_tol": 0.0, "source": "tests/regression/bccFe_lanczos/Fe.nml.ref (abs 1e-6, tests/regression/test_comparison.py:77)"}
    json.dump(manifest, open(os.path.join(OUT, "manifest.json"), "w"), indent=1)
    print("wrote", len(manifest), "cases to", OUT)


if __name__ == "__main__":
    main()
