import numpy as np, sys, os
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,ROOT); sys.path.insert(0,ROOT+'/tests')
from test_gpu_kubo import make_rec, scaled
import rslmtoasa_amd.recursion as R
z=dict(np.load(ROOT+"/build/kubo_full_hoh.npz"))
for k in ("hoh","nsp","cond_ll","acheb","bcheb"): z[k]=z[k].item()
rec,p=make_rec(z)
orig=scaled(rec,z)
for cll in (4,10,50):
    mu=rec.compute_moments_stochastic(z["v_a"], z["v_b"], cll, vo_a=z.get("vo_a"), vo_b=z.get("vo_b"), atlist=z["atlist"])
    ref=z["mu_nm"][:,:,:cll,:cll]
    print(cll,"max|mu| gpu %.3e ref %.3e err %.2e"%(np.abs(mu).max(),np.abs(ref).max(),np.abs(mu-ref).max()/np.abs(ref).max()), flush=True)
    print("  err per n:", ["%.1e"%(np.abs(mu[:,:,n]-ref[:,:,n]).max()) for n in range(min(cll,8))], " per m:", ["%.1e"%(np.abs(mu[:,:,:,m]-ref[:,:,:,m]).max()) for m in range(min(cll,8))], flush=True)
rec.close()
