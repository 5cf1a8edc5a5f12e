import sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
from helpers import objects_from, supercell_problem
from rslmtoasa_amd.recursion import Recursion
p = supercell_problem((22, 22, 22))
rec = Recursion(*objects_from(p, np.array([97], np.int32), 50, emin=-3.0, emax=1.8))
for o in (1, 0, 1):
    rec.set_option("side_stream", o)
    for _ in range(4):
        t0 = time.time(); rec.chebyshev_recur(); w = time.time() - t0
    print("single-site chebyshev side_stream=%d: wall %.1f ms device %.1f ms" % (o, w * 1e3, rec.timing()["total_ms"]))
rec.close()
