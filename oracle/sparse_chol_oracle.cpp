// sparse_chol_oracle.cpp -- TEST INFRASTRUCTURE ONLY (tests/, never the product path): a sequential CPU restatement of the
// multifrontal sparse Cholesky that csrc/sparse_chol.h runs on the GPU, over the same symbolic structures
// (csrc/sparse_symbolic.h: minimum-degree order, elimination tree, panels, child inverse maps).  It lets the CPU test suite
// check the index structures and the numeric scheme (front assembly, guarded pivots, update matrices, forward / backward
// sweeps with update vectors) against dense LAPACK without a GPU, and gives the GPU tests an independent factor to compare
// with.  Reference being replaced: scipy's spsolve on B = A D^2 A^T (SuperLU; main.py:180, :226); guard semantics as
// oracle/ipm_oracle.py::guarded_cholesky (pivot <= eps max diag B -> big).
//
// Build: g++ -O2 -std=c++17 -shared -fPIC -I interiorpointmethod_amd/csrc oracle/sparse_chol_oracle.cpp -o oracle/libsparse_chol_oracle.so
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "sparse_symbolic.h"

using namespace ipm::sym;

extern "C" {

// A: m x n CSC (rows sorted).  d: n, rhs: m.  Outputs: perm[m] (new -> old), Ldense (m x m row-major, PERMUTED order, lower),
// z[m] = B^{-1} rhs in the CALLER's row order, nfixed, stats[8] = {panels, tree height, widest front, factor entries,
// update entries, flops, fan-in nodes, most children of a panel}.  wcap / lds: panel limits (the GPU uses 32 / 8192).  Returns 0, or a positive code.
int spchol_oracle(int m, int n, const int* cp, const int* ri, const double* cv, const double* d, const double* rhs, double eps,
                  double big, double shift_rel, int wcap, int lds, int* perm, double* Ldense, double* z, int* nfixed,
                  double* stats) {
    std::vector<int> pv;
    OrderInfo oi;
    if (order_rows(m, n, cp, ri, pv, oi)) return 1;
    for (int i = 0; i < m; ++i) perm[i] = pv[i];
    std::vector<int> pos((size_t)m);
    for (int k = 0; k < m; ++k) pos[pv[k]] = k;
    // permuted A: CSC with sorted rows + CSR
    std::vector<int> pcp((size_t)n + 1, 0), pri((size_t)cp[n]);
    std::vector<double> pcv((size_t)cp[n]);
    for (int j = 0; j < n; ++j) {
        std::vector<std::pair<int, double>> col;
        for (int q = cp[j]; q < cp[j + 1]; ++q) col.emplace_back(pos[ri[q]], cv[q]);
        std::sort(col.begin(), col.end());
        for (size_t t = 0; t < col.size(); ++t) { pri[(size_t)cp[j] + t] = col[t].first; pcv[(size_t)cp[j] + t] = col[t].second; }
        pcp[(size_t)j + 1] = cp[j + 1];
    }
    Pattern P;
    if (!normal_pattern(m, n, pcp.data(), pri.data(), (int64_t)1.5e8, P)) return 2;
    Supernodes S;
    const char* rl = getenv("IPM_SP_RELAX");
    if (analyse(P, wcap, lds, S, (int64_t)3e8, rl ? atof(rl) : 1.0)) return 3;
    // B entries straight into the panels
    std::vector<double> L((size_t)S.lptr[S.nsn], 0.0), U((size_t)S.uptr[S.nsn], 0.0);
    {
        std::vector<int> where((size_t)m, -1);
        std::vector<std::vector<std::pair<int, double>>> rowsA((size_t)m);          // CSR of the permuted A
        for (int j = 0; j < n; ++j) for (int q = pcp[j]; q < pcp[j + 1]; ++q) rowsA[pri[q]].emplace_back(j, pcv[q]);
        for (int J = 0; J < S.nsn; ++J) {
            const int64_t r0 = S.rowptr[J];
            const int r = (int)(S.rowptr[(size_t)J + 1] - r0), w = S.w[J], c0 = S.c0[J];
            for (int a = 0; a < r; ++a) where[S.rows[(size_t)(r0 + a)]] = a;
            for (int b = 0; b < w; ++b) {
                const int k = c0 + b;
                for (auto& e : rowsA[k]) {
                    const int j = e.first;
                    for (int q = pcp[j]; q < pcp[j + 1]; ++q) {
                        const int i = pri[q];
                        if (i < k) continue;
                        if (where[i] < 0) return 4;
                        L[(size_t)(S.lptr[J] + (int64_t)where[i] * w + b)] += pcv[q] * e.second * d[j];
                    }
                }
            }
            for (int a = 0; a < r; ++a) where[S.rows[(size_t)(r0 + a)]] = -1;
        }
    }
    double maxdiag = -1.7976931348623157e308;
    for (int i = 0; i < m; ++i) maxdiag = std::max(maxdiag, L[(size_t)S.diagpos[i]]);
    const double thresh = eps * maxdiag, shift = shift_rel * maxdiag;
    int nfix = 0;
    for (int J = 0; J < S.nsn; ++J) {
        const int r = (int)(S.rowptr[(size_t)J + 1] - S.rowptr[J]), w = S.w[J], p = r - w;
        double* Lp = L.data() + S.lptr[J];
        double* Up = U.data() + S.uptr[J];
        for (int64_t e = 0; e < (int64_t)p * p; ++e) Up[e] = 0.0;
        for (int t = S.childptr[J]; t < S.childptr[(size_t)J + 1]; ++t) {           // children in ascending order, child-major
            const int K = S.child[(size_t)t];
            const int wk = S.w[K], pk = (int)(S.rowptr[(size_t)K + 1] - S.rowptr[K]) - wk;
            const double* Uk = U.data() + S.uptr[K];
            const int* rel = S.crel.data() + S.rowptr[K] + wk;
            for (int i = 0; i < pk; ++i)
                for (int j = 0; j <= i; ++j) {
                    const int a = rel[i], b = rel[j];
                    const double v = Uk[(size_t)i * pk + j];
                    if (b < w) Lp[(size_t)a * w + b] += v; else Up[(size_t)(a - w) * p + (b - w)] += v;
                }
        }
        for (int c = 0; c < w; ++c) Lp[(size_t)c * w + c] += shift;
        for (int c = 0; c < w; ++c) {
            double pvt = Lp[(size_t)c * w + c];
            if (!(pvt > thresh)) { pvt = big; ++nfix; }
            const double l = std::sqrt(pvt);
            Lp[(size_t)c * w + c] = l;
            for (int a = c + 1; a < r; ++a) Lp[(size_t)a * w + c] /= l;
            for (int a = c + 1; a < r; ++a)
                for (int b = c + 1; b < w && b <= a; ++b) Lp[(size_t)a * w + b] -= Lp[(size_t)a * w + c] * Lp[(size_t)b * w + c];
        }
        for (int i = 0; i < p; ++i)
            for (int j = 0; j <= i; ++j) {
                double dot = 0.0;
                for (int c = 0; c < w; ++c) dot += Lp[(size_t)(w + i) * w + c] * Lp[(size_t)(w + j) * w + c];
                Up[(size_t)i * p + j] -= dot;
            }
    }
    *nfixed = nfix;
    if (Ldense) {
        memset(Ldense, 0, sizeof(double) * (size_t)m * m);
        for (int J = 0; J < S.nsn; ++J) {
            const int r = (int)(S.rowptr[(size_t)J + 1] - S.rowptr[J]), w = S.w[J];
            for (int a = 0; a < r; ++a)
                for (int b = 0; b < w && b <= a; ++b)
                    Ldense[(size_t)S.rows[(size_t)(S.rowptr[J] + a)] * m + S.c0[J] + b] = L[(size_t)(S.lptr[J] + (int64_t)a * w + b)];
        }
    }
    // forward sweep with update vectors, backward sweep by gathering from the ancestors
    std::vector<double> zz((size_t)m), uvec(S.rows.size(), 0.0), f;
    for (int k = 0; k < m; ++k) zz[k] = rhs[pv[k]];
    for (int J = 0; J < S.nsn; ++J) {
        const int r = (int)(S.rowptr[(size_t)J + 1] - S.rowptr[J]), w = S.w[J], c0 = S.c0[J];
        const double* Lp = L.data() + S.lptr[J];
        f.assign((size_t)r, 0.0);
        for (int a = 0; a < w; ++a) f[a] = zz[c0 + a];
        for (int t = S.childptr[J]; t < S.childptr[(size_t)J + 1]; ++t) {
            const int K = S.child[(size_t)t];
            const int wk = S.w[K], pk = (int)(S.rowptr[(size_t)K + 1] - S.rowptr[K]) - wk;
            const int* rel = S.crel.data() + S.rowptr[K] + wk;
            for (int i = 0; i < pk; ++i) f[rel[i]] += uvec[(size_t)(S.rowptr[K] + wk + i)];
        }
        for (int c = 0; c < w; ++c) {
            f[c] /= Lp[(size_t)c * w + c];
            for (int a = c + 1; a < w; ++a) f[a] -= Lp[(size_t)a * w + c] * f[c];
        }
        for (int a = 0; a < w; ++a) zz[c0 + a] = f[a];
        for (int a = w; a < r; ++a) {
            double dot = 0.0;
            for (int c = 0; c < w; ++c) dot += Lp[(size_t)a * w + c] * f[c];
            uvec[(size_t)(S.rowptr[J] + a)] = f[a] - dot;
        }
    }
    for (int J = S.nsn - 1; J >= 0; --J) {
        const int r = (int)(S.rowptr[(size_t)J + 1] - S.rowptr[J]), w = S.w[J], c0 = S.c0[J];
        const double* Lp = L.data() + S.lptr[J];
        const int* rows = S.rows.data() + S.rowptr[J];
        std::vector<double> g((size_t)w);
        for (int c = 0; c < w; ++c) {
            double s = 0.0;
            for (int a = w; a < r; ++a) s += Lp[(size_t)a * w + c] * zz[rows[a]];
            g[c] = zz[c0 + c] - s;
        }
        for (int c = w - 1; c >= 0; --c) {
            g[c] /= Lp[(size_t)c * w + c];
            for (int a = 0; a < c; ++a) g[a] -= Lp[(size_t)c * w + a] * g[c];
        }
        for (int c = 0; c < w; ++c) zz[c0 + c] = g[c];
    }
    for (int k = 0; k < m; ++k) z[pv[k]] = zz[k];
    if (stats) {
        stats[0] = S.nsn; stats[1] = S.height; stats[2] = S.rmax; stats[3] = (double)S.nnz_factor; stats[4] = (double)S.uptr[S.nsn];
        stats[5] = S.flops; stats[6] = S.nvirtual; stats[7] = S.max_children;
    }
    return 0;
}

// Symbolic structures of the PRODUCT's analysis (csrc/sparse_symbolic.h) laid open for an independent check
// (tests/test_sparse_symbolic.py compares them with a 20-line NumPy restatement that shares no code with that header):
// the fill-reducing order perm[m] (new -> old), the elimination-tree parent array of the permuted pattern, and the
// entries of every column of L (diagonal included) as the panel structures of analyse() imply them with amalgamation
// switched off (relax = 0: fundamental supernodes, no explicit zeros).  Returns 0, or a positive code.
int spsym_structures(int m, int n, const int* cp, const int* ri, int wcap, int lds, int* perm, int* parent, int* colcount) {
    std::vector<int> pv, par;
    OrderInfo oi;
    if (order_rows(m, n, cp, ri, pv, oi)) return 1;
    Pattern P, Q;
    if (!normal_pattern(m, n, cp, ri, (int64_t)1.5e8, P)) return 2;
    permute(P, pv, Q);
    etree(Q, par);
    Supernodes S;
    if (analyse(Q, wcap, lds, S, (int64_t)3e8, 0.0)) return 3;
    for (int i = 0; i < m; ++i) { perm[i] = pv[i]; parent[i] = par[i]; colcount[i] = -1; }
    for (int J = 0; J < S.nsn; ++J) {
        const int r = (int)(S.rowptr[(size_t)J + 1] - S.rowptr[J]), w = S.w[J];
        for (int b = 0; b < w; ++b) colcount[S.c0[J] + b] = r - b;
    }
    return 0;
}

}  // extern "C"
