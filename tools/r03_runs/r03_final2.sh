#!/bin/bash
# round 3 evidence, second part: config 5 (16384 x 32768: bench line, kernel trace, PMC passes of adat_syrk_kernel with the
# current kernel sources), the expected-status probe of the general-form files that do not converge from the reference's
# start, and the batched 73-LP suite at 4 / 6 / 8 / 12 LPs in flight.      tools/r03_runs/r03_final2.sh
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; O=$R/gpurun_out; mkdir -p $O
python - <<'PY' 2>&1 | tee $O/g_general_status.txt
import os, numpy as np
from scipy import sparse
from interiorpointmethod_amd import general_form as G
for name in ("STANDATA", "SHELL", "SCAGR25"):
    z = np.load(os.path.join("tests/golden/general", name + ".npz"))
    def mat(p):
        if p + "_none" in z.files: return None
        return sparse.csc_matrix((z[p + "_data"], z[p + "_indices"], z[p + "_indptr"]), shape=tuple(int(v) for v in z[p + "_shape"]))
    obj, info = G.new_interior_sparse(c=z["c"], Aineq=mat("Aineq"), bineq=z["bineq"] if "bineq" in z.files else None, Aeq=mat("Aeq"),
                                      beq=z["beq"] if "beq" in z.files else None, lb=z["lb"], ub=z["ub"], tol=1e-8, return_info=True)
    print(name, "status", info["status"], "iterations", info["iterations"], "factor path", info.get("factor_path"))
PY
for W in 4 6 8 12; do
  timeout -k 10 300 python bench.py --workload netlib --netlib-set all --workers $W --no-cpu-baseline > $O/g_netlib_w$W.json 2> $O/g_netlib_w$W.err || { tail -5 $O/g_netlib_w$W.err; exit 1; }
  python -c "
import json
d=json.loads(open('gpurun_out/g_netlib_w$W.json').read().strip().splitlines()[-1]); s=d.get('summary', d.get('config', {}))
print('workers $W:', round(d['value'],2), 'LPs/s, wall', round(d.get('wall_seconds', 0),3))"
done
timeout -k 10 400 python bench.py --m 16384 --n 32768 --steps 5 --warmup 1 --no-netlib --no-cpu-baseline > $O/g_dense16k.json 2> $O/g_dense16k.err || { tail -5 $O/g_dense16k.err; exit 1; }
cd /tmp && export TMPDIR=/tmp
rm -rf $O/g16k_trace $O/g16k_pmc_*
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/g16k_trace -o p -- python3 $R/bench.py --m 16384 --n 32768 --steps 5 --warmup 1 --no-netlib --no-cpu-baseline > $O/g16k_trace.log 2>&1 || { tail -5 $O/g16k_trace.log; exit 1; }
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  T=$(echo $C | cut -d' ' -f1)
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/g16k_pmc_$T -o p -- python3 $R/bench.py --m 16384 --n 32768 --steps 3 --warmup 1 --no-netlib --no-cpu-baseline > $O/g16k_pmc_$T.log 2>&1 || { tail -5 $O/g16k_pmc_$T.log; exit 1; }
done
cd $R
python tools/prof_db_stats.py $O/g16k_trace 16 > $O/g16k_kernel_stats.txt; head -10 $O/g16k_kernel_stats.txt | cut -c1-150
rm -f $O/g16k_trace/*.db $O/g16k_trace/*/*.db
python tools/pmc_form_kernel.py --out $O/g16k_pmc_form_kernel.json --shape 16384 32768 $O/g16k_pmc_* | tail -14
cp $O/g16k_pmc_form_kernel.json profiles/r03_pmc_form_kernel_16k.json
timeout -k 10 400 python bench.py --m 16384 --n 32768 --steps 5 --warmup 1 --no-netlib --no-cpu-baseline > $O/g_dense16k.json 2> $O/g_dense16k.err || { tail -5 $O/g_dense16k.err; exit 1; }
python -c "
import json
d=json.loads(open('gpurun_out/g_dense16k.json').read().strip().splitlines()[-1]); print('16k', d['value'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline']['traffic_source'], d['objective_check'], d['phases_ms_per_step'])"
