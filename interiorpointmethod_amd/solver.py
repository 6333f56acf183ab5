"""Host-side mirror of the reference's interior-point call surface, running on libipm_hip.so.

Same names and argument meaning as the reference (payakorn/InteriorPointMethod):

    solve(A, b, c)                      -> (x, y, s)      the north-star seam
    interior_sparse(A, b, c, cTlb, tol) -> objective - cTlb          main.py:760-815
    interior(A, b, c, tol)              -> objective                 main.py:707-757 (returns None there)
    direction_predicted_sparse(..., method="normal"|"full") -> (dx, dy, ds)   main.py:197-229
    direction_corrected_sparse(...)     -> (dx, dy, ds)              main.py:247-269
    direction_predicted / direction_corrected (dense-path names)     main.py:185-194, 232-244
    solve_linear(B, rhs)                -> (N, 1)                    main.py:176-182

A is a scipy sparse matrix (any format; the reference passes CSC) or a dense array; b, c are
(len,) or (len, 1) of any numeric dtype (the .mat files hold int16/uint8) and are
cast to float64 here.  Outputs are fresh (len, 1) float64 arrays like the reference's.
PyTorch is used only to own the device workspace and the stream; all arithmetic is HIP.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _lib

try:  # scipy is optional on the host side (dense inputs work without it)
    from scipy import sparse as _sp
except Exception:  # pragma: no cover
    _sp = None

STATUS_NAMES = {0: "running", 1: "converged", 2: "max_iter", 3: "nan"}


def _dptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


REORDER_MIN_ROWS = 2048      # below this the factorization is a latency chain; reordering buys nothing


def _tile_envelope_work(P, nb=128):
    """Sum over 128-column blocks of (envelope height in blocks)^2 for the symmetric pattern P: the work measure the
    blocked Cholesky in libipm_hip pays (it skips the blocks below the tile envelope)."""
    P = P.tocoo()
    nblk = (P.shape[0] + nb - 1) // nb
    last = np.arange(nblk)
    np.maximum.at(last, np.minimum(P.row, P.col) // nb, np.maximum(P.row, P.col) // nb)
    last = np.maximum.accumulate(last)
    return float(np.sum((last - np.arange(nblk)) ** 2)), float(np.sum((nblk - 1 - np.arange(nblk)) ** 2))


def envelope_row_order(A, force=False):
    """Reverse Cuthill-McKee order of the rows of A on the pattern of A A^T, or None when it does not shrink the
    tile envelope of the normal matrix by at least 30 % (force=True: always the RCM order).  The row order of A is the caller's to choose: y and dy are
    permuted back, x and s are untouched, so the solver seam is unchanged (the reference's SuperLU also reorders
    internally, COLAMD, main.py:180)."""
    from scipy.sparse.csgraph import reverse_cuthill_mckee
    P = abs(A) @ abs(A).T
    P = _sp.csr_matrix(P)
    P.data[:] = 1.0
    perm = np.asarray(reverse_cuthill_mckee(P, symmetric_mode=True), dtype=np.int64)
    before, dense = _tile_envelope_work(P)
    after, _ = _tile_envelope_work(P[perm][:, perm])
    if force or after < 0.7 * min(before, dense):
        return perm
    return None


def sparse_factor_order(A, alternative_ms=0.0):
    """Fill-reducing row order for the multifrontal sparse Cholesky (ipm_order_rows: minimum degree + elimination-tree
    postorder, host only) -> (perm, info) with info = dict(nnz_pattern, nnz_factor, flops, height), or (None, None)
    when A A^T is too dense for it.  The reference gets this from SuperLU's COLAMD inside spsolve (main.py:180).
    alternative_ms > 0: the predicted ms per iteration of the dense-tile path; the elimination then gives up (None, None)
    at the first pivot whose degree shows that the sparse factor cannot beat it (factor="auto" passes it, "sparse" does not)."""
    lib = _lib.load()
    A = _sp.csc_matrix(A)
    m, n = A.shape
    perm = np.zeros(m, dtype=np.int32)
    info = np.zeros(8)
    if alternative_ms > 0.0:
        info[0], info[1] = float(alternative_ms), -1.0     # (marker: include/ipm_hip.h)
    ip = np.ascontiguousarray(A.indptr, dtype=np.int32)
    ii = np.ascontiguousarray(A.indices, dtype=np.int32)
    rc = lib.ipm_order_rows(m, n, ip.ctypes.data_as(C.POINTER(C.c_int32)), ii.ctypes.data_as(C.POINTER(C.c_int32)),
                            perm.ctypes.data_as(C.POINTER(C.c_int32)), _dptr(info))
    if rc == _lib.ERR_WORKSPACE:
        return None, None
    _lib.check(None, rc)
    return perm.astype(np.int64), dict(nnz_pattern=int(info[0]), nnz_factor=int(info[1]), flops=float(info[2]),
                                       height=int(info[3]), panel_height=int(info[4]), path_area=float(info[5]),
                                       panels=int(info[6]), widest_front=int(info[7]))


SPARSE_FACTOR_MIN_ROWS = 600          # below five 128-row blocks the dense chain is shorter than one tree sweep set
FUSED_SMALL_MAX_ROWS = 128            # sparse handles up to this many rows run the fused single-workgroup kernel (small_lp.h)


def prefer_sparse_factor(m, info, dense_blocks):
    """The rule of factor="auto", fitted to measurements on MI355X (tools/sparse_factor_check.py, one LP on the GPU, ms per
    iteration sparse / dense): STOCFOR3 1.16 / 14.0, SIERRA 0.64 / 2.2, STOCFOR2 0.49 / 1.47, CZPROB 0.60 / 0.92, SCTAP3
    0.57 / 0.92, SHELL 0.48 / 0.67, GFRD-PNC 0.36 / 0.65, SCTAP2 0.66 / 0.84, 80BAU3B 2.6 / 3.35, GANGES 1.01 / 1.13 -- but
    25FV47 1.34 / 0.70, NESM 2.85 / 1.77, GREENBEA 3.2 / 1.73, BNL2 4.4 / 1.62, D2Q06C 6.4 / 1.56, PILOTNOV 3.3 / 0.91, GROW15
    2.66 / 0.71.  The sparse factor walks the panel tree five times per iteration and every level is a hand-off between
    workgroups: 0.061 ms per level of the panel tree plus 3.5e-6 ms per (front rows)^2 along the critical path (round 2: least
    squares over 23 LPs gave 5.6e-6, worst error 0.3 ms; round 3: the update of the large fronts moved to the matrix cores and
    13 re-measured LPs give 2.6e-6 .. 5.3e-6, BNL2 4.4 -> 2.8, D2Q06C 6.4 -> 4.15, PILOTNOV 3.3 -> 2.3, 25FV47 1.34 -> 0.99 ms; info["panel_height"], info["path_area"] from ipm_order_rows).  The dense-tile
    path walks a chain of m/128 pivot blocks at 0.08 ms each and does its flops on the matrix cores.  A predicted gain of
    10 % switches paths: of the 23 measured LPs only SCFXM3 (0.54 / 0.63, predicted 0.86) is on the slower path."""
    if info is None or m < SPARSE_FACTOR_MIN_ROWS or info.get("panel_height", 0) <= 0:
        return False
    t_sparse = max(0.3, -0.13 + 0.061 * info["panel_height"] + 3.5e-6 * info["path_area"])      # ms per iteration
    return 1.1 * t_sparse < dense_tile_ms(dense_blocks)


def dense_tile_ms(dense_blocks):
    """Predicted ms per iteration of the dense-tile path (0.1 + 0.08 per 128-row block, fitted with the rule above)."""
    return 0.1 + 0.08 * dense_blocks


def _worth_ordering(A):
    """Cheap screen before the minimum-degree ordering: an upper bound on the entries of A A^T (sum over columns of
    c (c - 1) / 2).  Beyond a few million the factor is close to dense and the ordering would only burn host time."""
    c = np.diff(A.indptr).astype(np.float64)
    return float(np.sum(c * (c - 1.0) / 2.0)) <= 4.0e6


def path_flops(A, factor=None, want_info=False):
    """(path, Cholesky flops, flops of the four triangular sweeps) of one iteration AS THE DEVICE RUNS IT for this A under
    IpmSolver's factor rule: the sparse factor costs sum over columns of (entries of the column)^2 and 4 nnz(L); the
    dense-tile path factor_flops(A) and 4 m^2.  bench.py's roofline denominator for the Netlib runs.
    want_info: a fourth value, the ipm_order_rows info of an LP put on the sparse factor (None otherwise)."""
    m = A.shape[0]
    factor = factor or os.environ.get("IPM_FACTOR", "auto")
    if _sp is not None and _sp.issparse(A) and factor != "dense" and m > FUSED_SMALL_MAX_ROWS and \
            (factor == "sparse" or (m >= SPARSE_FACTOR_MIN_ROWS and _worth_ordering(_sp.csc_matrix(A)))):
        perm, info = sparse_factor_order(A, 0.0 if factor == "sparse" else dense_tile_ms((m + 127) // 128))
        if perm is not None and (factor == "sparse" or prefer_sparse_factor(m, info, (m + 127) // 128)):
            out = ("sparse", float(info["flops"]), 4.0 * info["nnz_factor"])
            return out + (info,) if want_info else out
    out = ("dense", factor_flops(A), 4.0 * m * m)
    return out + (None,) if want_info else out


def factor_flops(A, nb=128):
    """Flops of the blocked Cholesky of A A^T AS THE DEVICE RUNS IT for this A: dense handles and sparse handles whose
    tile envelope removes less than 20 % of the work factor the full matrix (m^3/3); otherwise only the blocks inside
    the tile envelope (after the reverse Cuthill-McKee row order where IpmSolver applies it) are touched:
    sum over block columns of nb^3 (h^2 + 2 h + 1/3), h = envelope height in blocks below the diagonal block.
    Used by bench.py for the roofline denominator of the Netlib runs -- STOCFOR3's factor is 11 % of m^3/3."""
    m = A.shape[0]
    if _sp is None or not _sp.issparse(A):
        return m ** 3 / 3.0
    P = _sp.csr_matrix(abs(A) @ abs(A).T)
    P.data[:] = 1.0
    if m >= REORDER_MIN_ROWS:
        perm = envelope_row_order(A)
        if perm is not None:
            P = P[perm][:, perm]
    work, dense = _tile_envelope_work(P, nb)
    if not work < 0.8 * dense:                   # the library's rule (ipm_set_A_csc): the envelope must remove work
        return m ** 3 / 3.0
    Pc = P.tocoo()
    nblk = (m + nb - 1) // nb
    last = np.arange(nblk)
    np.maximum.at(last, np.minimum(Pc.row, Pc.col) // nb, np.maximum(Pc.row, Pc.col) // nb)
    hgt = (np.maximum.accumulate(last) - np.arange(nblk)).astype(np.float64)
    return float(np.sum(nb ** 3 * (hgt * hgt + 2.0 * hgt + 1.0 / 3.0)))


def _col(v, n, name):
    v = np.ascontiguousarray(np.asarray(v, dtype=np.float64).reshape(-1))
    if v.shape[0] != n:
        raise ValueError("%s has length %d, expected %d" % (name, v.shape[0], n))
    return v


class Prepared:
    """The host-side analysis of one LP -- what IpmSolver does before it touches the device: canonical A, the factorization
    path (factor="auto" rule), the fill-reducing or envelope row order and A, b in that order.  `prepare` builds it and
    IpmSolver(..., prepared=P) takes it.  (Computing it AHEAD of the solves on helper threads in the batched mode was tried and
    is slower -- 14.55 -> 13.2 LPs/s on the 73-LP suite: the helpers' SciPy sections hold the interpreter lock the eight
    worker threads need between their library calls.)"""
    __slots__ = ("host", "A", "b", "c", "m", "n", "factor", "order_info", "perm")


def prepare(A, b, c, dense=False, reorder="auto", factor=None):
    """Host-only part of IpmSolver.__init__ (no device is touched) -> Prepared."""
    P = Prepared()
    P.perm = None
    if _sp is not None and _sp.issparse(A):
        A = _sp.csc_matrix(A, dtype=np.float64)
        A.sum_duplicates()
        A.sort_indices()
        m, n = A.shape
        if dense or A.nnz == 0:
            A = np.ascontiguousarray(A.toarray())
    else:
        A = np.ascontiguousarray(np.asarray(A, dtype=np.float64))
        if A.ndim != 2:
            raise ValueError("A must be 2-D")
        m, n = A.shape
    P.m, P.n = int(m), int(n)
    b = _col(b, P.m, "b")
    c = _col(c, P.n, "c")
    P.host = (A, b, c)
    # factor: "dense" = blocked dense-tile Cholesky (tile envelope, RCM row order), "sparse" = multifrontal sparse
    # Cholesky (minimum-degree row order), "auto" (default; environment IPM_FACTOR overrides) = whichever the model
    # of prefer_sparse_factor expects to be faster
    factor = factor or os.environ.get("IPM_FACTOR", "auto")
    if factor not in ("auto", "dense", "sparse"):
        raise ValueError("factor must be 'auto', 'dense' or 'sparse'")
    P.factor, P.order_info = "dense", None
    if _sp is not None and _sp.issparse(A) and factor != "dense" and \
            (factor == "sparse" or (P.m >= SPARSE_FACTOR_MIN_ROWS and _worth_ordering(A))):
        perm, info = sparse_factor_order(A, 0.0 if factor == "sparse" else dense_tile_ms((P.m + 127) // 128))
        if perm is not None and (factor == "sparse" or prefer_sparse_factor(P.m, info, (P.m + 127) // 128)):
            P.factor, P.order_info = "sparse", info
            P.perm = perm
            A = _sp.csc_matrix(_sp.csr_matrix(A)[perm])
            A.sort_indices()
            b = np.ascontiguousarray(b[perm])
        elif factor == "sparse":
            raise ValueError("factor='sparse': A A^T is too dense for the sparse factor (ipm_order_rows)")
    if P.factor == "dense" and _sp is not None and _sp.issparse(A) and reorder and \
            (reorder == "rcm" or P.m >= REORDER_MIN_ROWS):
        perm = envelope_row_order(A, force=(reorder == "rcm"))     # "auto": only when it pays
        if perm is not None:
            P.perm = perm
            A = _sp.csc_matrix(_sp.csr_matrix(A)[perm])
            A.sort_indices()
            b = np.ascontiguousarray(b[perm])
    P.A, P.b, P.c = A, b, c
    return P


class IpmSolver:
    """One LP bound to one GPU: owns a libipm_hip handle whose workspace is a torch tensor."""

    def __init__(self, A, b, c, device=0, eta=0.91, pivot_guard_eps=1e-30, pivot_guard_big=1e64,
                 check_every=4, use_torch=True, dense=False, regularize=0.0, reorder="auto", concurrent=False,
                 auto_regularize=True, factor=None, prepared=None, lockstep=False):
        lib = _lib.load()
        self._lib = lib
        self._h = None
        if prepared is None:
            prepared = prepare(A, b, c, dense=dense, reorder=reorder, factor=factor)
        # device row i = caller's row perm[i] (sparse A whose rows the host analysis reordered: minimum degree or RCM)
        self._perm = prepared.perm
        self.m, self.n = prepared.m, prepared.n
        self._host = prepared.host  # caller's row order: used by start-point heuristics only
        self.factor, self.order_info = prepared.factor, prepared.order_info
        A, b, c = prepared.A, prepared.b, prepared.c
        opts = _lib.Options()
        lib.ipm_default_options(C.byref(opts))
        opts.eta, opts.pivot_guard_eps, opts.pivot_guard_big = eta, pivot_guard_eps, pivot_guard_big
        ce = int(check_every)
        if os.environ.get("IPM_CHECK_EVERY"):
            ce = int(os.environ["IPM_CHECK_EVERY"])
        opts.check_every = ce
        opts.regularize = float(regularize)
        # concurrent=True: this handle shares the GPU with others (batched mode) -- one stream per handle, no look-ahead, no
        # device polling (include/ipm_hip.h: IPM_FLAG_SINGLE_STREAM).  Without it the library still protects itself (it
        # counts the live handles per device and falls back to stream events).
        # lockstep=True: the handle is meant for solve_lockstep (ipm_solve_batch: iteration k of several LPs in the same launches)
        opts.flags = (_lib.FLAG_LOCKSTEP if lockstep else 0) | \
                     ((_lib.FLAG_NO_DEVICE_POLLING | _lib.FLAG_SINGLE_STREAM) if concurrent else 0) | \
                     (0 if auto_regularize else _lib.FLAG_NO_AUTO_REGULARIZE) | \
                     (_lib.FLAG_SPARSE_FACTOR if self.factor == "sparse" else 0)
        nbytes = C.c_size_t(0)
        self.sparse = _sp is not None and _sp.issparse(A)
        if self.sparse:                      # A stays sparse on the device (CSR + CSC, sparse formation of B)
            opts.sparse_nnz = int(A.nnz)
            # (sized from the options: a sparse-factor handle carries no dense m x m normal matrix)
            _lib.check(None, lib.ipm_workspace_bytes_opts(self.m, self.n, C.byref(opts), C.byref(nbytes)))
        else:
            _lib.check(None, lib.ipm_workspace_bytes(self.m, self.n, C.byref(nbytes)))
        self.workspace_bytes = nbytes.value
        ws_ptr, stream = None, None
        self._ws = None
        if use_torch:
            import torch
            if not torch.cuda.is_available():
                raise _lib.IpmLibraryError("no ROCm device visible to torch; the HIP path cannot run")
            dev = torch.device("cuda", device)
            self._ws = torch.empty(self.workspace_bytes, dtype=torch.uint8, device=dev)   # device buffer only
            ws_ptr = C.c_void_p(self._ws.data_ptr())
            stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        h = C.c_void_p()
        _lib.check(None, lib.ipm_create(int(device), self.m, self.n, C.byref(opts), ws_ptr,
                                        self.workspace_bytes if ws_ptr else 0, stream, C.byref(h)))
        self._h = h
        if self.sparse:
            indptr = np.ascontiguousarray(A.indptr, dtype=np.int32)
            indices = np.ascontiguousarray(A.indices, dtype=np.int32)
            data = np.ascontiguousarray(A.data, dtype=np.float64)
            self._check(lib.ipm_set_A_csc(h, indptr.ctypes.data_as(C.POINTER(C.c_int32)),
                                          indices.ctypes.data_as(C.POINTER(C.c_int32)), _dptr(data),
                                          int(data.shape[0])))
            if self.factor == "sparse" and self.schedule()["fused_small"]:
                # m <= 128: the library serves the LP with the fused single-workgroup kernel and builds no sparse factor
                # (ipm_set_A_csc); report the path the handle really takes (the row order stays: it is harmless)
                self.factor, self.order_info = "dense", None
        else:
            self._check(lib.ipm_set_A_dense(h, C.c_void_p(A.ctypes.data), self.n, 0))
        self._check(lib.ipm_set_bc(h, _dptr(b), _dptr(c)))
        self.stats = None

    # -- plumbing
    def _check(self, code):
        _lib.check(self._h, code)

    def close(self):
        if self._h is not None:
            self._lib.ipm_destroy(self._h)
            self._h = None
            self._ws = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- state
    def init_state(self, y0=1.0):
        self._check(self._lib.ipm_init_state(self._h, float(y0)))

    def _rows_in(self, v):          # caller's row order -> device row order
        return v if self._perm is None else np.ascontiguousarray(v[self._perm])

    def _rows_out(self, v):         # device row order -> caller's row order
        if self._perm is None:
            return v
        out = np.empty_like(v)
        out[self._perm] = v
        return out

    def set_state(self, x, y, s):
        x, y, s = _col(x, self.n, "x"), self._rows_in(_col(y, self.m, "y")), _col(s, self.n, "s")
        self._check(self._lib.ipm_set_state(self._h, _dptr(x), _dptr(y), _dptr(s)))

    def get_state(self):
        x, y, s = np.empty(self.n), np.empty(self.m), np.empty(self.n)
        self._check(self._lib.ipm_get_state(self._h, _dptr(x), _dptr(y), _dptr(s)))
        return x.reshape(-1, 1), self._rows_out(y).reshape(-1, 1), s.reshape(-1, 1)

    # -- seams
    def newton_direction(self, corrector=False):
        dx, dy, ds = np.empty(self.n), np.empty(self.m), np.empty(self.n)
        st = _lib.Stats()
        self._check(self._lib.ipm_newton_direction(self._h, 1 if corrector else 0, _dptr(dx), _dptr(dy),
                                                   _dptr(ds), C.byref(st)))
        self.stats = st.as_dict()
        return dx.reshape(-1, 1), self._rows_out(dy).reshape(-1, 1), ds.reshape(-1, 1)

    def iterate(self, n_steps):
        st = _lib.Stats()
        self._check(self._lib.ipm_iterate(self._h, int(n_steps), C.byref(st)))
        self.stats = st.as_dict()
        return self.stats

    def solve(self, tol=1e-8, max_iter=5000, tol_gap=None):
        st = _lib.Stats()
        e3 = tol if tol_gap is None else tol_gap
        self._check(self._lib.ipm_solve(self._h, float(tol), float(tol), float(e3), int(max_iter), C.byref(st)))
        self.stats = st.as_dict()
        return self.stats

    def history(self):
        """Per-iteration records of the last solve()/iterate() (oldest first; the most recent 1024): list of dicts
        with k, objective, rp_norm, rd_norm, gap, mu, sigma, alpha_aff_p/d, alpha_p/d, pivots_fixed -- the line the
        reference prints per iteration (main.py:808-809, :1186)."""
        buf = (_lib.IterRecord * _lib.HISTORY_CAPACITY)()
        n = C.c_int32(0)
        self._check(self._lib.ipm_get_history(self._h, buf, _lib.HISTORY_CAPACITY, C.byref(n)))
        return [{k: getattr(buf[i], k) for k, _ in _lib.IterRecord._fields_} for i in range(n.value)]

    def schedule(self):
        """How the handle runs its factorization (ipm_get_schedule): dict for tests and diagnostics."""
        out = (C.c_int32 * 12)()
        self._check(self._lib.ipm_get_schedule(self._h, out))
        keys = ("blocks", "group_steps", "grouped_trsv", "device_polling", "counter_steps", "event_steps", "envelope",
                "live_handles", "timeouts_recovered", "fused_small", "fused_factor", "sparse_level_mode")
        return dict(zip(keys, (int(v) for v in out)))

    def factor_info(self):
        """Structure of the sparse factor (ipm_get_factor_info) or None for the dense-tile path."""
        if self.factor != "sparse":
            return None
        out = (C.c_int64 * 8)()
        self._check(self._lib.ipm_get_factor_info(self._h, out))
        keys = ("panels", "tasks", "height", "widest_front", "factor_entries", "update_entries", "product_terms",
                "serial_launches")
        return dict(zip(keys, (int(v) for v in out)))

    def set_profiling(self, level=2):
        """0 off, 1 time the A D^2 A^T kernel only, 2 all phases (True == 2 for old callers)."""
        level = 2 if level is True else (0 if level is False else int(level))
        self._check(self._lib.ipm_set_profiling(self._h, level))

    def phase_ms(self):
        out = (C.c_double * 4)()
        self._check(self._lib.ipm_get_phase_ms(self._h, out))
        return dict(form=out[0], factor=out[1], trisolve=out[2], other=out[3])

    # -- kernel-level
    def form_normal_matrix(self, d):
        d = _col(d, self.n, "d")
        B = np.empty((self.m, self.m))
        self._check(self._lib.ipm_form_normal_matrix(self._h, _dptr(d), _dptr(B), self.m))
        if self._perm is not None:
            out = np.empty_like(B)
            out[np.ix_(self._perm, self._perm)] = B
            return out
        return B

    def get_factor(self):
        """Lower Cholesky factor of the current normal matrix in DEVICE row order (rows self._perm of the caller's
        A when a reordering was applied)."""
        L = np.empty((self.m, self.m))
        self._check(self._lib.ipm_get_factor(self._h, _dptr(L), self.m))
        return L

    def normal_solve(self, rhs, d=None, reuse_factor=False):
        """z with (A diag(d) A^T) z = rhs on the device (d = None: ones); reuse_factor keeps the previous factor."""
        rhs = self._rows_in(_col(rhs, self.m, "rhs"))
        z = np.empty(self.m)
        dptr = None if d is None else _dptr(_col(d, self.n, "d"))
        nfix = C.c_int32(0)
        self._check(self._lib.ipm_normal_solve(self._h, dptr, _dptr(rhs), _dptr(z), 1 if reuse_factor else 0, C.byref(nfix)))
        self.last_pivots_fixed = nfix.value          # > 0 with d = 1: A A^T is singular, i.e. A has dependent rows
        return self._rows_out(z)

    def mehrotra_start(self):
        """Mehrotra's starting point (SIAM J. Optim. 2 (1992) 575-601, section 7): least-squares x and (y, s), shifted
        into the positive orthant and balanced.  NOT the reference's start (x = s = 1, sparse_interior.py:193-200): an
        optional mode that changes the trajectory; two solves with A A^T on the device, the rest is
        O(nnz) host arithmetic."""
        A, b, c = self._host
        x = A.T @ self.normal_solve(b)
        y = self.normal_solve(A @ c, reuse_factor=True)
        s = c - A.T @ y
        x = np.asarray(x).ravel(); s = np.asarray(s).ravel()
        x = x + max(-1.5 * x.min(), 0.0)
        s = s + max(-1.5 * s.min(), 0.0)
        xs = 0.5 * float(x @ s)
        if not (np.isfinite(xs) and s.sum() > 0 and x.sum() > 0 and xs > 0):
            return np.ones(self.n), np.ones(self.m), np.ones(self.n)          # degenerate data: the reference's start
        x = x + xs / s.sum()
        s = s + xs / x.sum()
        return x, np.asarray(y).ravel(), s

    def solve_linear(self, B, rhs):
        """B z = rhs for the CALLER's dense SPD matrix (main.py:176-182): the row order this handle keeps its own A in
        plays no part (ipm_solve_linear factors B as given, without the tile envelope or the sparse factor)."""
        B = np.ascontiguousarray(np.asarray(B, dtype=np.float64))
        rhs = _col(rhs, self.m, "rhs")
        z = np.empty(self.m)
        nfix = C.c_int32(0)
        self._check(self._lib.ipm_solve_linear(self._h, _dptr(B), self.m, _dptr(rhs), _dptr(z), C.byref(nfix)))
        return z.reshape(-1, 1), nfix.value


def solve_lockstep(solvers, tol=1e-8, max_iter=5000, tol_gap=None):
    """ipm_solve_batch: solve the LPs of `solvers` (IpmSolver objects created with lockstep=True on one device, a state set) AT ONCE,
    iteration k of all of them in the same launches (csrc/lockstep.h) -> list of statistics dicts, one per solver.  Per-LP semantics
    and arithmetic are those of IpmSolver.solve on each of them alone (bit-identical iterates)."""
    lib = _lib.load()
    n = len(solvers)
    hs = (C.c_void_p * n)(*[sv._h for sv in solvers])
    st = (_lib.Stats * n)()
    _lib.check(solvers[0]._h, lib.ipm_solve_batch(hs, n, tol, tol, tol if tol_gap is None else tol_gap, int(max_iter), st))
    out = []
    for sv, s_ in zip(solvers, st):
        sv.stats = s_.as_dict()
        out.append(sv.stats)
    return out


class LockstepBatch:
    """ipm_batch_*: the lockstep batch, incrementally.  add(solver) lets an IpmSolver (lockstep=True, a state set) JOIN between two
    steps; step() runs opt.check_every iterations of every active LP in the same launches and returns the solvers that finished,
    each with its statistics in solver.stats.  The solvers stay owned by the caller (close them after they are reported finished)."""

    def __init__(self, device=0, tol=1e-8, max_iter=5000, tol_gap=None, stream=None):
        """stream: a torch.cuda.Stream the batch's launches go to (the caller keeps it alive); None: a stream of the batch's own."""
        self._lib = _lib.load()
        self._b = C.c_void_p()
        self._stream = stream
        _lib.check(None, self._lib.ipm_batch_create(int(device), C.c_void_p(stream.cuda_stream) if stream is not None else None, C.byref(self._b)))
        self.tol, self.max_iter, self.tol_gap = float(tol), int(max_iter), float(tol if tol_gap is None else tol_gap)
        self.solvers = []
        self.active = 0

    def _check(self, code):
        if code != _lib.IPM_OK:
            raise _lib.IpmError(code, (self._lib.ipm_batch_last_error(self._b) or b"").decode("utf-8", "replace"))

    def add(self, solver):
        idx = C.c_int32(-1)
        self._check(self._lib.ipm_batch_add(self._b, solver._h, self.tol, self.tol, self.tol_gap, self.max_iter, C.byref(idx)))
        assert idx.value == len(self.solvers)
        self.solvers.append(solver)
        self.active += 1
        return idx.value

    def step(self):
        cap = max(1, len(self.solvers))
        fin = (C.c_int32 * cap)()
        nf, na = C.c_int32(0), C.c_int32(0)
        self._check(self._lib.ipm_batch_step(self._b, fin, cap, C.byref(nf), C.byref(na)))
        self.active = na.value
        out = []
        for k in range(nf.value):
            sv = self.solvers[fin[k]]
            st = _lib.Stats()
            self._check(self._lib.ipm_batch_stats(self._b, fin[k], C.byref(st)))
            sv.stats = st.as_dict()
            out.append(sv)
        return out

    def close(self):
        if self._b:
            self._lib.ipm_batch_destroy(self._b)
            self._b = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def lockstep_eligible(solver):
    """Can this IpmSolver join solve_lockstep?  Sparse A on the dense-tile factor, more than 128 rows (the small LPs have their fused
    single-workgroup kernel, the sparse-factor LPs their tree sweeps)."""
    return bool(solver.sparse and solver.factor != "sparse" and not solver.schedule()["fused_small"])


def _info(solver, cTlb=0.0):
    st = dict(solver.stats)
    st["status_name"] = STATUS_NAMES.get(st["status"], "?")
    st["rp"] = st["rp_norm"] / (1.0 + st["b_norm"])          # reference scaling, main.py:170
    st["rd"] = st["rd_norm"] / (1.0 + st["c_norm"])          # main.py:171
    st["objective_minus_cTlb"] = st["objective"] - cTlb
    return st


_last_info = None


def last_info():
    """Statistics of the most recent solve()/interior*() call in this process."""
    return _last_info


def solve_with_info(A, b, c, tol=1e-8, max_iter=5000, y0=1.0, device=0, tol_gap=None, start="reference",
                    history=False, **opts):
    """solve() plus the statistics record (iterations, status, objective, rp, rd, gap, ...).
    start="reference": x = s = 1, y = y0 as the reference does; start="mehrotra": IpmSolver.mehrotra_start().
    history=True adds info["history"], the per-iteration records (IpmSolver.history()).  An LP whose A has more
    than 5 % dependent rows (the QAP family) is solved with the 1e-14 Tikhonov shift, switched on by the library
    after the first factorization (info["auto_regularized"] == 1; auto_regularize=False keeps it off)."""
    global _last_info
    if start == "mehrotra" and not opts.get("regularize") and os.environ.get("IPM_AUTO_REGULARIZE", "1") != "0":
        # the least-squares start factors A A^T: guarded pivots there are dependent rows of A.  Where they are a
        # sizeable fraction of the rows (the QAP family: 9-16 %; every other Netlib file: at most 2.7 %) the guard alone
        # stalls the loop and the 1e-14 Tikhonov shift is switched on; a handful of dependent rows is left
        # to the guard (the shift breaks 25FV47, BNL1, D6CUBE, WOOD1P, which have 1-11 of them)
        with IpmSolver(A, b, c, device=device, **opts) as probe:
            probe.normal_solve(np.zeros(probe.m))
            if probe.last_pivots_fixed > 0.05 * probe.m:
                opts = dict(opts, regularize=1e-14)
    import time as _time
    t0 = _time.perf_counter()
    with IpmSolver(A, b, c, device=device, **opts) as sv:
        t1 = _time.perf_counter()
        if start == "mehrotra":
            sv.set_state(*sv.mehrotra_start())
        elif start == "reference":
            sv.init_state(y0)
        else:
            raise ValueError('start must be "reference" or "mehrotra"')
        sv.solve(tol=tol, max_iter=max_iter, tol_gap=tol_gap)
        t2 = _time.perf_counter()
        x, y, s = sv.get_state()
        info = _info(sv)
        if history:
            info["history"] = sv.history()
        fac = sv.factor
        # the library's hidden recoveries (batch.RECORD_FIELDS): polls that timed out and were rolled back and repeated,
        # and sparse-factor sweeps that ran as one workgroup after such a time-out
        info["timeouts_recovered"] = sv.schedule()["timeouts_recovered"]
        fi = sv.factor_info()
        info["serial_launches"] = fi["serial_launches"] if fi else 0
        info["factor_path"] = fac
    t3 = _time.perf_counter()
    # host-side phases of the call (seconds): handle creation + upload + symbolic analysis, the solve, read-back + destroy
    info["setup_seconds"], info["solve_seconds"], info["teardown_seconds"] = t1 - t0, t2 - t1, t3 - t2
    if os.environ.get("IPM_LP_TIMING"):
        import sys
        print("[lp-timing] m=%d factor=%s setup %.3f solve %.3f (device %.3f) teardown %.3f" %
              (sv.m, fac, t1 - t0, t2 - t1, info["solve_ms"] * 1e-3, t3 - t2), file=sys.stderr, flush=True)
    _last_info = info
    return x, y, s, info


def solve(A, b, c, tol=1e-8, max_iter=5000, y0=1.0, device=0, **opts):
    """min c^T x s.t. Ax=b, x>=0 by the Mehrotra predictor-corrector loop on the GPU -> (x, y, s)."""
    x, y, s, _ = solve_with_info(A, b, c, tol=tol, max_iter=max_iter, y0=y0, device=device, **opts)
    return x, y, s


def interior_sparse(A, b, c, cTlb=0.0, tol=1e-20, device=0):
    """Drop-in for main.py:760-815: start x=s=y=1, cap 5000, returns sum(x*c) - cTlb."""
    _, _, _, info = solve_with_info(A, b, c, tol=tol, max_iter=5000, y0=1.0, device=device)
    return info["objective"] - float(cTlb)


def interior(A, b, c, tol=1e-20, device=0):
    """Drop-in for main.py:707-757 (dense path: y0=0, cap 50000); returns the objective."""
    _, _, _, info = solve_with_info(np.asarray(A, dtype=np.float64), b, c, tol=tol, max_iter=50000, y0=0.0,
                                    device=device)
    return info["objective"]


_METHODS = ("normal", "full")


def direction_predicted_sparse(A, b, c, x, y, s, method="normal", device=0):
    """main.py:197: predictor direction at (x, y, s).  method="normal" (main.py:221-229) and method="full" (the
    unreduced KKT system of main.py:198-212) define the same direction; the device always solves it through the normal
    equations (the Schur complement of the full system), which agrees with the reference's method="full" LU to 1e-11
    at a well-conditioned point (tests/test_gpu_parity.py::test_direction_kats)."""
    if method not in _METHODS:
        raise ValueError('method must be "normal" or "full" (the "eliminate" variant of the reference uses a wrong '
                         'right-hand side, main.py:270-276, and is not mirrored)')
    with IpmSolver(A, b, c, device=device) as sv:
        sv.set_state(x, y, s)
        return sv.newton_direction(corrector=False)


def direction_corrected_sparse(A, b, c, x, y, s, delta_x_aff=None, delta_y_aff=None, delta_s_aff=None,
                               method="full", device=0):
    """main.py:247: corrector direction (the reference's default method here is "full").  The affine direction is
    recomputed on the device from (x, y, s) (same factor reused), so the delta_*_aff arguments are accepted for
    signature compatibility only."""
    if method not in _METHODS:
        raise ValueError('method must be "normal" or "full"')
    with IpmSolver(A, b, c, device=device) as sv:
        sv.set_state(x, y, s)
        sv.newton_direction(corrector=False)
        return sv.newton_direction(corrector=True)


def direction_predicted(A, b, c, x, y, s, device=0):
    """Dense-path name of the same seam (main.py:185-194)."""
    return direction_predicted_sparse(np.asarray(A, dtype=np.float64), b, c, x, y, s, method="full", device=device)


def direction_corrected(A, b, c, x, y, s, delta_x_aff=None, delta_y_aff=None, delta_s_aff=None, device=0):
    """Dense-path name of the corrector seam (main.py:232-244)."""
    return direction_corrected_sparse(np.asarray(A, dtype=np.float64), b, c, x, y, s, delta_x_aff, delta_y_aff,
                                      delta_s_aff, method="full", device=device)


def solve_linear(A, b, method="hip", device=0):
    """main.py:176-182 for a symmetric positive (semi)definite matrix: guarded Cholesky on the GPU."""
    B = np.asarray(A.todense() if (_sp is not None and _sp.issparse(A)) else A, dtype=np.float64)
    m = B.shape[0]
    rhs = np.asarray(b, dtype=np.float64).reshape(-1)
    with IpmSolver(np.eye(m, 1), np.zeros(m), np.zeros(1), device=device) as sv:
        z, _ = sv.solve_linear(B, rhs)
    return z
