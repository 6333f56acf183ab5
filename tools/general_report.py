"""General-form driver (new_interior_sparse) over tests/golden/general/*.npz: objective vs the reference's own result
and the Netlib optimum."""
import glob, os, sys
import numpy as np
from scipy import sparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from interiorpointmethod_amd import general_form as G

for f in sorted(glob.glob(os.path.join("tests", "golden", "general", "*.npz"))):
    z = np.load(f)
    nm = os.path.basename(f)[:-4]
    def mat(p):
        return None if p + "_none" in z.files else sparse.csc_matrix((z[p + "_data"], z[p + "_indices"], z[p + "_indptr"]), shape=tuple(int(v) for v in z[p + "_shape"]))
    try:
        obj, info = G.new_interior_sparse(c=z["c"], Aineq=mat("Aineq"), bineq=z["bineq"] if "bineq" in z.files else None, Aeq=mat("Aeq"),
                                          beq=z["beq"] if "beq" in z.files else None, lb=z["lb"], ub=z["ub"], tol=1e-8, return_info=True,
                                          start=os.environ.get("IPM_START", "reference"))
    except Exception as e:
        print("%-10s ERROR %s" % (nm, e)); continue
    o, r = float(z["netlib_optimum"]), float(z["ref_objective"])
    print("%-10s %-9s it=%4d obj % .10e  netlib % .10e (rel %.1e)  reference % .4e  lb!=0:%d ub<inf:%d" % (
        nm, info["status_name"], info["iterations"], obj, o, abs(obj - o) / max(1, abs(o)), r,
        int(np.count_nonzero(z["lb"]) > 0), int(np.isfinite(z["ub"]).any())), flush=True)
