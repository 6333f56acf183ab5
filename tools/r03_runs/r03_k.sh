#!/bin/bash
# what the driver runs at round end, in its order: smoke, GPU tests, bench (1 GPU), bench under torch.distributed.run with one rank
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out; mkdir -p $O
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/k_pytest.log 2>&1 || { tail -40 $O/k_pytest.log; exit 1; }
tail -1 $O/k_pytest.log
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/k_bench.json 2> $O/k_bench.err || { tail -20 $O/k_bench.err; exit 1; }
python -c "
import json
d=json.loads(open('gpurun_out/k_bench.json').read().strip().splitlines()[-1]); print('bench', round(d['value'],2), d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['objective_check'], round(d['netlib_all']['value'],2), round(d['netlib']['value'],2), d['cpu_baseline']['value'])"
timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-netlib > $O/k_bench_dist.json 2> $O/k_bench_dist.err || { tail -20 $O/k_bench_dist.err; exit 1; }
tail -1 $O/k_bench_dist.json | cut -c1-300
