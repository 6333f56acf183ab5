"""Host-side logic that needs no GPU: the .mat loader (layout of sparse_interior.py:139-216), problem
validity screening, workload generator pins, and the batch cost model."""
import os

import numpy as np
import pytest
from scipy import sparse
from scipy.io import savemat

from interiorpointmethod_amd import matio, batch
from interiorpointmethod_amd.workloads import synthetic_lp, flops_per_iteration


def _write_mat(path, A, b, f, cTlb, int_dtypes=True):
    A = sparse.coo_matrix(A)
    k = A.data.astype(np.int16) if int_dtypes else A.data
    st = np.zeros((1, 1), dtype=[("i", "O"), ("j", "O"), ("k", "O")])
    st["i"][0, 0] = A.row.reshape(1, -1).astype(np.uint16)
    st["j"][0, 0] = A.col.reshape(1, -1).astype(np.uint16)
    st["k"][0, 0] = k.reshape(1, -1)
    savemat(path, {"A": st, "b": np.asarray(b).reshape(-1, 1), "f": np.asarray(f).reshape(-1, 1),
                   "cTlb": np.array([[cTlb]]), "num_variables": np.array([[A.shape[1]]]),
                   "num_constraints": np.array([[A.shape[0]]])})


def test_mat_loader_casts_and_keeps_shape(tmp_path):
    """Keys f, b, cTlb, A{i,j,k}, num_variables, num_constraints (sparse_interior.py:157-167); integer
    payloads come back as float64; a trailing empty column keeps the declared shape instead of
    the max-index inference of sparse_interior.py:215."""
    A = np.array([[3, 6, 8, 0], [8, 4, 1, 0]])
    _write_mat(os.path.join(tmp_path, "TOY.mat"), A, np.array([30, 44], dtype=np.uint8),
               np.array([-100, -125, -20, 0], dtype=np.int16), 2.5)
    As, b, c, cTlb = matio.create_problem_from_mps("TOY", root=str(tmp_path))
    assert sparse.issparse(As) and As.shape == (2, 4) and As.dtype == np.float64
    assert b.shape == (2, 1) and c.shape == (4, 1) and b.dtype == np.float64 and c.dtype == np.float64
    assert np.array_equal(As.toarray(), A.astype(float)) and cTlb == 2.5
    assert matio.is_valid_problem(As, b, c)
    assert not matio.is_valid_problem(As, np.array([[np.inf], [1.0]]), c)


def test_npz_fixture_roundtrip(golden_dir):
    A, b, c, cTlb, valid = matio.load_npz_problem(os.path.join(golden_dir, "netlib", "AFIRO.npz"))
    assert A.shape == (27, 51) and A.nnz == 102 and valid and b.shape == (27, 1) and c.shape == (51, 1)


def test_synthetic_workload_pins():
    """Sanity pins of SURVEY.md 8(d)."""
    A, b, c = synthetic_lp(64, 128, seed=0)
    assert A[0, 0] == 0.1257302210933933
    assert np.isclose(b[0, 0], 7.898903583954942, rtol=1e-14) and np.isclose(c[0, 0], 13.513107422440132, rtol=1e-14)
    assert flops_per_iteration(4096, 8192) == pytest.approx(1.608e11, rel=1e-3)
    assert flops_per_iteration(16384, 32768) == pytest.approx(1.027e13, rel=1e-3)


def test_record_layout_and_summary():
    assert batch.RECORD_FIELDS[:4] == ("id", "status", "iterations", "objective") and batch.NF == 14
    assert batch.RECORD_FIELDS[9:] == ("timeouts_recovered", "serial_launches", "setup_seconds", "solve_seconds",
                                       "teardown_seconds")
    rec = np.array([[0, 1, 10, 1.0, 0, 0, 0, 0.1, 2, 1, 3, 0.01, 0.05, 0.02], [1, 3, 5, np.nan, 0, 0, 0, 0.2, 0, 0, 0, 0, 0, 0],
                    [2, -6, 0, np.nan, 0, 0, 0, 0.0, 0, 0, 0, 0, 0, 0]])
    s = batch.summarize(rec)
    assert (s["converged"], s["nan"], s["invalid"], s["total_iterations"], s["pivots_fixed"]) == (1, 1, 1, 15, 2)
    assert (s["timeouts_recovered"], s["serial_launches"]) == (1, 3) and abs(s["device_solve_seconds_sum"] - 0.05) < 1e-15
    # a custom solve_fn that reports only the nine basic statistics still yields a full-width record (zeros)
    row = batch._row(4, dict(status=1, iterations=3, objective=2.0, rp=0, rd=0, gap=0, seconds=0.1, pivots_fixed=0))
    assert len(row) == batch.NF and row[9:] == [0.0] * 5


def test_envelope_row_order_recovers_a_staircase():
    """Host side of the tile-envelope Cholesky: a block-bidiagonal A with shuffled rows has a dense-looking tile
    envelope; reverse Cuthill-McKee on the pattern of A A^T brings it back to a narrow band."""
    from interiorpointmethod_amd import solver as S
    nbk, bs, cs = 30, 100, 120
    blocks = [[None] * nbk for _ in range(nbk)]
    for i in range(nbk):
        blocks[i][i] = sparse.random(bs, cs, density=0.05, random_state=np.random.RandomState(i), format="csr") + \
            sparse.eye(bs, cs, format="csr")
        if i + 1 < nbk:
            blocks[i + 1][i] = sparse.random(bs, cs, density=0.03, random_state=np.random.RandomState(99 + i), format="csr")
    A = sparse.csr_matrix(sparse.bmat(blocks))
    shuffled = A[np.random.default_rng(0).permutation(A.shape[0])]
    perm = S.envelope_row_order(shuffled)
    assert perm is not None and sorted(perm.tolist()) == list(range(A.shape[0]))
    P = abs(shuffled) @ abs(shuffled).T
    before, dense = S._tile_envelope_work(P)
    after, _ = S._tile_envelope_work(sparse.csr_matrix(P)[perm][:, perm])
    assert before > 0.8 * dense and after < 0.15 * dense
    # a matrix whose normal matrix is dense gains nothing: no reordering
    assert S.envelope_row_order(sparse.csr_matrix(np.random.default_rng(1).standard_normal((300, 400)))) is None


def test_results_table_layout_matches_reference_log(tmp_path):
    """tools/netlib_report.py writes the table of the reference driver (script.py:139-198).  The header and the AFIRO
    row of the reference's own log (conclusion1.txt:1 and :3, quoted here as data) are reproduced character for
    character from the same numbers; the SciPy column is computed on the host exactly as script.py:155-164 does."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("netlib_report", os.path.join(root, "tools", "netlib_report.py"))
    R = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(R)
    ref_header = "Name                    Interi time           Scipy time              Obj fun               Interi                Scipy"
    ref_afiro = "AFIRO                          0.40                 0.03              -464.75              -464.75              -464.75"
    assert R.header().rstrip("\r\n") == ref_header
    assert R.row("AFIRO", -464.75314286, 0.40, 0.03, -464.7531428, -464.753142).rstrip("\r\n") == ref_afiro
    calls = []

    def stub(c, Aineq, bineq, Aeq, beq, lb, ub):          # stands in for the GPU solve on a GPU-less host
        calls.append(c.shape[0])
        return -464.7531
    rows = R.run_general(["AFIRO"], stub, with_scipy=True)
    assert calls and rows[0][0] == "AFIRO" and abs(rows[0][1] - (-464.75314286)) < 1e-6        # Netlib optimum column
    assert abs(rows[0][5] - (-464.7531)) < 1e-3                                                 # SciPy on the host
    out = os.path.join(tmp_path, "t.txt")
    R.write_table(rows, out)
    lines = open(out, newline="").read().split("\r\n")
    assert lines[0] == ref_header and lines[1].split()[0] == "AFIRO" and lines[1].split()[3:] == ["-464.75"] * 3


def test_bench_netlib_helpers():
    """bench.py's Netlib leg: the parity set loads (26 LPs), the flop model follows the factorization path the solver
    takes (STOCFOR3: the sparse multifrontal factor, a few 1e6 flop per iteration; under IPM_FACTOR=dense the tile
    envelope, ~12 % of m^3/3), the roofline record has the contract keys, and a PMC file collected for another kernel
    source is refused."""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench", os.path.join(root, "bench.py"))
    B = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(B)
    names, probs, flops = B.load_netlib("parity")
    assert len(names) == 26 and names == sorted(B.PARITY_SET)
    i = names.index("STOCFOR3")
    m = probs[i][0].shape[0]
    assert B.PATHS["STOCFOR3"] == "sparse" and B.PATHS["MAROS-R7"] == "dense" and B.PATHS["AFIRO"] == "dense"
    assert 2e6 < flops[i] < 2e7
    from interiorpointmethod_amd.solver import path_flops
    path, f_chol, f_sweeps = path_flops(probs[i][0], factor="dense")
    assert path == "dense" and 0.05 * m ** 3 / 3 < f_chol < 0.25 * m ** 3 / 3 and f_sweeps == 4.0 * m * m
    j = names.index("AFIRO")
    assert flops[j] > 27 ** 3 / 3
    rec = np.zeros((26, batch.NF)); rec[:, 0] = np.arange(26); rec[:, 1] = 1; rec[:, 2] = 30; rec[:, 7] = 0.01
    rf = B.netlib_roofline(names, probs, flops, rec, 1.0, 1)
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(rf) and 0 < rf["frac"] < 1
    assert rf["latency_floor"]["chain_seconds_one_gpu"] > 0
    t, why = B.load_traffic(4096, 8192)
    assert (t is None and ("stale" in why or "no PMC" in why or "shape" in why)) or t > 0
    assert len(B.kernel_source_sha()) == 16


def test_prepare_is_host_only(golden_dir):
    """solver.prepare -- the host analysis IpmSolver does before it touches the device: no device needed; STOCFOR2 goes to the
    sparse factor with a minimum-degree row order, AFIRO (27 rows) stays dense and unpermuted, a dense ndarray passes through."""
    from interiorpointmethod_amd import solver as S
    probs = []
    for nm in ("STOCFOR2", "AFIRO"):
        A, b, c, cTlb, valid = matio.load_npz_problem(os.path.join(golden_dir, "netlib", nm + ".npz"))
        probs.append((A, b, c))
    probs.append(synthetic_lp(64, 128, seed=0))
    P0, P1, P2 = (S.prepare(*p) for p in probs)
    m0 = probs[0][0].shape[0]
    assert P0.factor == "sparse" and sorted(P0.perm.tolist()) == list(range(m0)) and P0.order_info["nnz_factor"] > 0
    assert sparse.issparse(P0.A) and P0.A.shape == probs[0][0].shape
    assert np.array_equal(P0.b, np.asarray(probs[0][1], dtype=float).ravel()[P0.perm])
    assert abs(P0.A - sparse.csr_matrix(sparse.csc_matrix(probs[0][0], dtype=float))[P0.perm]).sum() == 0
    assert P1.factor == "dense" and P1.perm is None and (P1.m, P1.n) == (27, 51)
    assert P2.factor == "dense" and isinstance(P2.A, np.ndarray) and P2.A.shape == (64, 128) and P2.perm is None
    Q = S.prepare(*probs[0])
    assert Q.factor == P0.factor and np.array_equal(Q.perm, P0.perm)         # deterministic


def test_bench_compact_line_fits_the_driver_tail():
    """The driver keeps only a tail of stdout: the last line of bench.py must stay under 4 KB (target 2 KB) whatever the
    per-LP tables hold.  Built from a committed full record of round 3 (25 KB) and from a synthetic worst case."""
    import importlib.util
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    with open(os.path.join(root, "profiles", "r03_final_bench.json")) as fh:
        full = json.load(fh)
    assert len(json.dumps(full)) > 20000
    full["netlib_all"]["projected_makespan_8gpu_s"] = 1.4512345678
    full["netlib_all"]["slowest_lp"] = "80BAU3B"
    for k in ("netlib_all", "netlib"):
        full[k]["wall_seconds_runs"] = [1.4123456, 1.4234567, 1.4345678]
    full["netlib_all"]["emulated_multi_gpu"] = {"world": 8, "wall_seconds": 0.71234567, "value": 49.1234567, "unit": "LPs/s", "converged": 35,
                                                "shard_walls": [0.7] * 8, "slowest_lp_of_slowest_shard": "PILOT87", "how": "h" * 300}
    line = bench.compact_line(full)
    assert "\n" not in line and len(line) < 2048, len(line)
    c = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in c, k
    assert c["value"] == pytest.approx(full["value"], rel=1e-5)
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in c["roofline"]
    assert c["roofline"]["frac"] == pytest.approx(full["roofline"]["frac"], rel=1e-5)
    assert set(("value", "unit", "cores", "kind", "sample")) <= set(c["cpu_baseline"])
    assert c["netlib_all"]["converged"] == full["netlib_all"]["summary"]["converged"]
    assert c["netlib_all"]["projected_makespan_8gpu_s"] == pytest.approx(1.45123, rel=1e-4)
    assert "per_lp" not in c["netlib_all"] and "per_lp" not in c["netlib"]
    assert c["netlib_all"]["emulated_multi_gpu"] == {"world": 8, "wall_seconds": 0.712346, "value": 49.1235, "converged": 35}
    # worst case: very long free-text fields and a thousand LPs
    full["config"]["workload"] = "w" * 5000
    full["cpu_baseline"]["sample"] = "s" * 5000
    full["roofline"]["kernel"] = "k" * 5000
    full["netlib_all"]["per_lp"] = {"LP%04d" % i: {"status": 0, "it": i} for i in range(1000)}
    assert len(bench.compact_line(full)) < 2048
    # the netlib workload's own line
    nl = dict(full["netlib_all"], steps=73, warmup=1, ms_per_step=1.0, higher_is_better=True, scaling="strong",
              vs_baseline=None, dtype="f64", data="netlib fixtures", config={"workload": "Netlib all"})
    c2 = json.loads(bench.compact_line(nl))
    assert c2["summary"]["converged"] == 35 and len(json.dumps(c2)) < 2048


def test_lockstep_rule_is_the_same_on_every_rank():
    """run_batch(lockstep="auto") takes the lockstep batches only where a rank holds enough LPs of more than 128 rows
    (batch.lockstep_wanted; the driver loop being batched: script.py:147-173 of the reference)."""
    import scipy.sparse as sp
    from interiorpointmethod_amd import batch
    mk = lambda m: (sp.eye(m, m + 3, format="csc"), np.ones(m), np.ones(m + 3))     # noqa: E731
    many = [mk(200 + i) for i in range(batch.LOCKSTEP_MIN_LPS)] + [mk(20)] * 5
    few = many[1:]
    assert batch.lockstep_wanted(many, world=1, workers=8, mode="auto")
    assert not batch.lockstep_wanted(few, world=1, workers=8, mode="auto")        # small LPs do not count
    assert not batch.lockstep_wanted(many, world=2, workers=8, mode="auto")       # half of them per rank
    assert not batch.lockstep_wanted(many, world=1, workers=1, mode="auto")       # nothing to overlap
    assert batch.lockstep_wanted(few, world=1, workers=8, mode=True) and not batch.lockstep_wanted(many, world=1, workers=8, mode=False)
