#!/bin/bash
# round 4 evidence run, part B: config 5 (16384 x 32768) with its CPU baseline, kernel trace and PMC passes; 8192 x 16384; the sizes
# sweep fused vs serial; kernel statistics of the 73-LP suite under the lockstep batches.   tools/r04_runs/r04_final_b.sh
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; O=$R/gpurun_out; mkdir -p $O
timeout -k 10 500 python bench.py --m 16384 --n 32768 --steps 5 --warmup 1 --no-netlib > $O/r04_final_dense16k.log 2> $O/r04_final_dense16k.err || { tail -5 $O/r04_final_dense16k.err; exit 1; }
tail -1 $O/r04_final_dense16k.log | cut -c1-900
cd /tmp && export TMPDIR=/tmp
rm -rf $O/g_trace $O/g_pmc_*
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/g_trace -o p -- python3 $R/bench.py --m 16384 --n 32768 --no-netlib --no-cpu-baseline --steps 4 --warmup 1 > $O/g_trace.log 2>&1 || { tail -5 $O/g_trace.log; exit 1; }
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  T=$(echo $C | cut -d' ' -f1)
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/g_pmc_$T -o p -- python3 $R/bench.py --m 16384 --n 32768 --steps 2 --warmup 1 --no-netlib --no-cpu-baseline > $O/g_pmc_$T.log 2>&1 || { tail -5 $O/g_pmc_$T.log; exit 1; }
done
cd $R
python tools/prof_db_stats.py $O/g_trace 14 > $O/r04_dense16k_kernel_stats.txt; head -10 $O/r04_dense16k_kernel_stats.txt | cut -c1-150
python tools/pmc_form_kernel.py --kernel adat_syrk_kernel --out $O/r04_pmc_form_kernel_16k.json --shape 16384 32768 $O/g_pmc_* | tail -12
cp $O/r04_pmc_form_kernel_16k.json profiles/r04_pmc_form_kernel_16k.json
rm -rf $O/g_trace $O/g_pmc_*
# config 5 once more with the PMC file of this source in profiles/ (roofline.traffic)
timeout -k 10 500 python bench.py --m 16384 --n 32768 --steps 5 --warmup 1 --no-netlib > $O/r04_final_dense16k.log 2> $O/r04_final_dense16k.err || { tail -5 $O/r04_final_dense16k.err; exit 1; }
tail -1 $O/r04_final_dense16k.log > $O/r04_final_dense16k.json; cut -c1-900 $O/r04_final_dense16k.json
timeout -k 10 300 python bench.py --m 8192 --n 16384 --steps 10 --warmup 2 --no-netlib --no-cpu-baseline > $O/r04_final_dense8k.log 2> $O/r04_final_dense8k.err || { tail -5 $O/r04_final_dense8k.err; exit 1; }
tail -1 $O/r04_final_dense8k.log > $O/r04_final_dense8k.json; cut -c1-500 $O/r04_final_dense8k.json
bash tools/prof_suite.sh --netlib-set all > $O/r04_netlib_suite_kernel_stats_lockstep.txt 2>&1; head -30 $O/r04_netlib_suite_kernel_stats_lockstep.txt | cut -c1-150
bash tools/ff_sizes.sh > $O/r04_ff_sizes_final.txt 2>&1; cat $O/r04_ff_sizes_final.txt
