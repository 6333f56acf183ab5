// sparse_symbolic.h -- host side of the SPARSE Cholesky of B = A D^2 A^T: fill-reducing row order,
// elimination tree, supernodes and the index structures the multifrontal kernels of sparse_chol.h walk.  Pure C++ (no HIP).
//
// What it replaces: the reference hands B to scipy's spsolve (main.py:180, :226), i.e. SuperLU with a COLAMD column order,
// a symbolic factorization and a supernodal numeric factorization.  The pattern of B does not depend on D, so everything here
// is computed ONCE per LP (ipm_order_rows before the upload, the rest in ipm_set_A_csc) and reused by every iteration.
//
//   normal_pattern   pattern of A A^T (strict lower + upper, no diagonal) from the CSC structure of A
//   min_degree       minimum (external) degree order on a quotient graph: eliminated vertices become elements, elements
//                    reachable through the pivot are absorbed, degrees of the pivot's neighbours are recomputed exactly.
//                    Ties break to the lowest index: the order is a function of the pattern only (reproducible)
//   etree / postorder  Liu's elimination tree with path compression; children visited in ascending order
//   analyse          column structures by child merging, fundamental supernodes cut into panels that fit the LDS budget of
//                    the numeric kernel, child lists, child -> parent inverse maps, slot of every entry of B in the panels
#pragma once
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <cstdint>
#include <queue>
#include <utility>
#include <vector>

namespace ipm {
namespace sym {

struct Pattern {                    // symmetric, both triangles, no diagonal, sorted adjacency
    int m = 0;
    std::vector<int64_t> ptr;
    std::vector<int> idx;
};

// pattern of A A^T.  cp/ri: CSC of A (rows need not be sorted).  Returns false when more than `cap` entries would be needed.
inline bool normal_pattern(int m, int n, const int* cp, const int* ri, int64_t cap, Pattern& P) {
    std::vector<int64_t> rp((size_t)m + 1, 0);
    const int64_t nnz = cp[n];
    for (int64_t q = 0; q < nnz; ++q) rp[(size_t)ri[q] + 1]++;
    for (int i = 0; i < m; ++i) rp[(size_t)i + 1] += rp[i];
    std::vector<int> ci((size_t)nnz);
    {
        std::vector<int64_t> nx(rp.begin(), rp.end() - 1);
        for (int j = 0; j < n; ++j)
            for (int q = cp[j]; q < cp[j + 1]; ++q) ci[(size_t)nx[ri[q]]++] = j;
    }
    P.m = m; P.ptr.assign((size_t)m + 1, 0); P.idx.clear();
    std::vector<int> mark((size_t)m, -1), row;
    for (int i = 0; i < m; ++i) {
        row.clear();
        mark[i] = i;
        for (int64_t p = rp[i]; p < rp[(size_t)i + 1]; ++p) {
            const int j = ci[(size_t)p];
            for (int q = cp[j]; q < cp[j + 1]; ++q) {
                const int k = ri[q];
                if (mark[k] != i) { mark[k] = i; row.push_back(k); }
            }
        }
        if ((int64_t)P.idx.size() + (int64_t)row.size() > cap) return false;
        std::sort(row.begin(), row.end());
        P.idx.insert(P.idx.end(), row.begin(), row.end());
        P.ptr[(size_t)i + 1] = (int64_t)P.idx.size();
    }
    return true;
}

// order[k] = vertex eliminated k-th.  `budget` bounds the work (adjacency entries scanned); returns false when exceeded, or
// (degree_cap > 0) as soon as a pivot of more than degree_cap neighbours comes up: every order of the rest then has a chain of
// fronts of about degree_cap, degree_cap - w, ... rows up to the root, which the caller has decided it does not want.
inline bool min_degree(const Pattern& P, std::vector<int>& order, int64_t budget = (int64_t)6e7, int degree_cap = 0, int64_t* work_out = nullptr) {
    const int m = P.m;
    std::vector<std::vector<int>> adjv((size_t)m), adje((size_t)m), elem((size_t)m);
    std::vector<char> state((size_t)m, 0);               // 0 variable, 1 element, 2 absorbed element
    std::vector<int> deg((size_t)m), mark((size_t)m, -1), Lp;
    int stamp = 0;
    int64_t work = 0;
    typedef std::pair<int, int> DI;
    std::priority_queue<DI, std::vector<DI>, std::greater<DI>> pq;
    for (int i = 0; i < m; ++i) {
        adjv[i].assign(P.idx.begin() + P.ptr[i], P.idx.begin() + P.ptr[(size_t)i + 1]);
        deg[i] = (int)adjv[i].size();
        pq.push(DI(deg[i], i));
    }
    order.clear(); order.reserve((size_t)m);
    while ((int)order.size() < m) {
        int p = -1;
        while (!pq.empty()) {
            DI t = pq.top(); pq.pop();
            if (state[t.second] == 0 && deg[t.second] == t.first) { p = t.second; break; }
        }
        if (p < 0) return false;                           // (cannot happen: every live variable has a current entry)
        if (degree_cap > 0 && deg[p] > degree_cap) { if (work_out) *work_out = work; return false; }
        order.push_back(p);
        // reach of p: live variable neighbours + live members of its elements
        ++stamp; Lp.clear(); mark[p] = stamp;
        for (int v : adjv[p]) if (state[v] == 0 && mark[v] != stamp) { mark[v] = stamp; Lp.push_back(v); }
        for (int e : adje[p]) {
            if (state[e] != 1) continue;
            for (int v : elem[e]) if (state[v] == 0 && mark[v] != stamp) { mark[v] = stamp; Lp.push_back(v); }
            work += (int64_t)elem[e].size();
            state[e] = 2; std::vector<int>().swap(elem[e]);                    // absorbed into the new element p
        }
        state[p] = 1;
        std::vector<int>().swap(adjv[p]); std::vector<int>().swap(adje[p]);
        std::sort(Lp.begin(), Lp.end());
        elem[p] = Lp;
        const int lp_stamp = stamp;
        for (int i : Lp) {
            // variable neighbours now covered by element p (and dead ones) leave the list; absorbed elements leave, p joins
            std::vector<int>& av = adjv[i];
            size_t w = 0;
            for (size_t t = 0; t < av.size(); ++t) { const int v = av[t]; if (state[v] == 0 && mark[v] != lp_stamp) av[w++] = v; }
            av.resize(w);
            std::vector<int>& ae = adje[i];
            w = 0;
            for (size_t t = 0; t < ae.size(); ++t) if (state[ae[t]] == 1 && ae[t] != p) ae[w++] = ae[t];
            ae.resize(w);
            ae.push_back(p);
            work += (int64_t)av.size() + (int64_t)ae.size();
        }
        for (int i : Lp) {                                                     // exact external degree
            ++stamp; mark[i] = stamp;
            int d = 0;
            for (int v : adjv[i]) if (mark[v] != stamp) { mark[v] = stamp; ++d; }
            for (int e : adje[i]) {
                std::vector<int>& me = elem[e];
                size_t w = 0;
                for (size_t t = 0; t < me.size(); ++t) {
                    const int v = me[t];
                    if (state[v] != 0) continue;                               // compact stale members on the way
                    me[w++] = v;
                    if (mark[v] != stamp) { mark[v] = stamp; ++d; }
                }
                work += (int64_t)me.size();
                me.resize(w);
            }
            deg[i] = d;
            pq.push(DI(d, i));
        }
        // the marks of Lp were overwritten by the degree passes: nothing below relies on them
        if (work > budget) { if (work_out) *work_out = work; return false; }
    }
    if (work_out) *work_out = work;
    return true;
}

// permuted pattern: Q = P(order, order); pos[old] = new
inline void permute(const Pattern& P, const std::vector<int>& order, Pattern& Q) {
    const int m = P.m;
    std::vector<int> pos((size_t)m);
    for (int k = 0; k < m; ++k) pos[order[k]] = k;
    Q.m = m; Q.ptr.assign((size_t)m + 1, 0); Q.idx.resize(P.idx.size());
    for (int k = 0; k < m; ++k) Q.ptr[(size_t)k + 1] = Q.ptr[k] + (P.ptr[(size_t)order[k] + 1] - P.ptr[order[k]]);
    for (int k = 0; k < m; ++k) {
        int64_t o = Q.ptr[k];
        for (int64_t p = P.ptr[order[k]]; p < P.ptr[(size_t)order[k] + 1]; ++p) Q.idx[(size_t)o++] = pos[P.idx[(size_t)p]];
        std::sort(Q.idx.begin() + Q.ptr[k], Q.idx.begin() + o);
    }
}

// Liu's elimination tree of a symmetric pattern
inline void etree(const Pattern& P, std::vector<int>& parent) {
    const int m = P.m;
    parent.assign((size_t)m, -1);
    std::vector<int> anc((size_t)m, -1);
    for (int i = 0; i < m; ++i) {
        for (int64_t p = P.ptr[i]; p < P.ptr[(size_t)i + 1]; ++p) {
            int k = P.idx[(size_t)p];
            if (k >= i) break;
            while (k != -1 && k < i) {                   // climb to the root of k's subtree, compressing the path to i
                const int nx = anc[k];
                anc[k] = i;
                if (nx == -1) parent[k] = i;
                k = nx;
            }
        }
    }
}

// postorder of a forest (children in ascending order): post[k] = vertex visited k-th
inline void postorder(const std::vector<int>& parent, std::vector<int>& post) {
    const int m = (int)parent.size();
    std::vector<int> head((size_t)m, -1), next((size_t)m, -1), stack;
    for (int v = m - 1; v >= 0; --v) if (parent[v] >= 0) { next[v] = head[parent[v]]; head[parent[v]] = v; }
    post.clear(); post.reserve((size_t)m);
    for (int r = 0; r < m; ++r) {
        if (parent[r] >= 0) continue;
        stack.push_back(r);
        while (!stack.empty()) {
            const int v = stack.back();
            const int c = head[v];
            if (c >= 0) { head[v] = next[c]; stack.push_back(c); }
            else { post.push_back(v); stack.pop_back(); }
        }
    }
}

struct OrderInfo {
    int64_t nnz_pattern = 0;      // entries of the strict lower triangle of A A^T
    int64_t nnz_factor = 0;       // entries of L (with the diagonal)
    double flops = 0.0;           // sum over columns of (entries of the column)^2: the sparse factorization's multiply-adds
    int height = 0;               // elimination-tree height in columns
};

// Fill-reducing row order of A for the Cholesky of A D^2 A^T: minimum degree, then the elimination-tree postorder (subtrees
// contiguous, parent after child).  perm[new] = old.  Returns 0, or 1 when the pattern or the work exceeds the caps
// (the caller then keeps its dense/envelope path).
// keep (optional): receives the pattern of A A^T IN THE FINAL ORDER, so that a caller who analyses next does not form it again.
inline int order_rows(int m, int n, const int* cp, const int* ri, std::vector<int>& perm, OrderInfo& info,
                      int64_t pattern_cap = (int64_t)6e7, Pattern* keep = nullptr, int degree_cap = 0, int64_t work_budget = (int64_t)6e7) {
    Pattern P, Q;
    if (!normal_pattern(m, n, cp, ri, pattern_cap, P)) return 1;
    std::vector<int> order, parent, post;
    int64_t md_work = 0;
    const bool md_ok = min_degree(P, order, work_budget, degree_cap, &md_work);
    if (getenv("IPM_ORDER_TRACE")) fprintf(stderr, "[order] m %d pattern %lld work %lld eliminated %zu %s\n", m, (long long)P.idx.size(), (long long)md_work, order.size(), md_ok ? "ok" : "gave up");
    if (!md_ok) return 1;
    permute(P, order, Q);
    etree(Q, parent);
    postorder(parent, post);
    perm.resize((size_t)m);
    for (int k = 0; k < m; ++k) perm[k] = order[post[k]];
    // statistics of the final order
    permute(P, perm, Q);
    etree(Q, parent);
    std::vector<std::vector<int>> st((size_t)m);
    std::vector<int> mark((size_t)m, -1), hgt((size_t)m, 1);
    std::vector<std::vector<int>> kids((size_t)m);
    for (int k = 0; k < m; ++k) if (parent[k] >= 0) kids[parent[k]].push_back(k);
    info = OrderInfo();
    info.nnz_pattern = (int64_t)P.idx.size() / 2;
    for (int k = 0; k < m; ++k) {
        std::vector<int>& s = st[k];
        mark[k] = k;
        for (int64_t p = Q.ptr[k]; p < Q.ptr[(size_t)k + 1]; ++p) { const int i = Q.idx[(size_t)p]; if (i > k && mark[i] != k) { mark[i] = k; s.push_back(i); } }
        for (int c : kids[k]) {
            for (int i : st[c]) if (i > k && mark[i] != k) { mark[i] = k; s.push_back(i); }
            std::vector<int>().swap(st[c]);
            hgt[k] = std::max(hgt[k], hgt[c] + 1);
        }
        const double cnt = (double)s.size() + 1.0;
        info.nnz_factor += (int64_t)s.size() + 1;
        info.flops += cnt * cnt;
        info.height = std::max(info.height, hgt[k]);
        if (info.nnz_factor > (int64_t)2e8) return 1;
    }
    if (keep) *keep = std::move(Q);
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// Supernodal structure of the factor in the GIVEN row order (expected: the output of order_rows; any order is valid).
struct Supernodes {
    int m = 0, nsn = 0;                    // nsn panels, fan-in nodes included
    std::vector<int> c0, w;                // [nsn] first column and width of panel J (a panel = up to `wcap` consecutive columns of a
                                           // supernode; w = 0: a FAN-IN node, which only sums the update matrices of its children)
    std::vector<int64_t> rowptr;           // [nsn+1] into rows
    std::vector<int> rows;                 // panel J: its own columns first, then the rows below, ascending
    std::vector<int> crel;                 // aligned with rows: for a >= w_J, position of rows[a] in the PARENT's front
    std::vector<int64_t> lptr;             // [nsn+1] panel values: r x w row-major at lptr[J]
    std::vector<int64_t> uptr;             // [nsn+1] update matrix of J: p x p row-major (lower part used), p = r - w
    std::vector<int> parent;               // [nsn] panel tree (parent > child)
    std::vector<int> childptr, child;      // CSR children, ascending
    std::vector<int64_t> diagpos;          // [m] slot of B(i,i) in the panel values
    int height = 0, rmax = 0, wmax = 0, pcmax = 0, nvirtual = 0, max_children = 0;
    int64_t panel_max = 0;                 // largest r x w
    int64_t nnz_factor = 0;
    double flops = 0.0;
};

// lds_doubles: the numeric kernel keeps a panel (r x w) in LDS, so w <= lds_doubles / r (at least 1).
inline int analyse(const Pattern& P, int wcap, int lds_doubles, Supernodes& S, int64_t value_cap = (int64_t)3e8, double relax = 1.0) {
    const int m = P.m;
    std::vector<int> parent;
    etree(P, parent);
    // column structures (below-diagonal rows, ascending) by child merging; counts decide the fundamental supernodes
    std::vector<std::vector<int>> st((size_t)m), kids((size_t)m);
    for (int k = 0; k < m; ++k) if (parent[k] >= 0) kids[parent[k]].push_back(k);
    std::vector<int> mark((size_t)m, -1), cnt((size_t)m, 0);
    // the structures of all columns are kept until their supernode is formed: free a child's only when it is not the
    // first column of a panel that still needs it -- simpler: keep the structure of the LAST column of every chain
    std::vector<char> start((size_t)m, 1);
    int64_t total = 0;
    for (int k = 0; k < m; ++k) {
        std::vector<int>& s = st[k];
        mark[k] = k;
        for (int64_t p = P.ptr[k]; p < P.ptr[(size_t)k + 1]; ++p) { const int i = P.idx[(size_t)p]; if (i > k && mark[i] != k) { mark[i] = k; s.push_back(i); } }
        for (int c : kids[k]) for (int i : st[c]) if (i > k && mark[i] != k) { mark[i] = k; s.push_back(i); }
        std::sort(s.begin(), s.end());
        cnt[k] = (int)s.size();
        total += cnt[k] + 1;
        if (total > value_cap) return 1;
        if (k > 0 && parent[k - 1] == k && cnt[k - 1] == cnt[k] + 1 && kids[k].size() == 1) start[k] = 0;
        // children structures are no longer needed once merged, except as the structure of their own panel (first column)
        for (int c : kids[k]) if (!start[c]) std::vector<int>().swap(st[c]);
    }
    // Relaxed amalgamation: a supernode whose columns end right before its parent's first column (the parent's LAST child in
    // the postorder) is merged into the parent when the explicit zeros this stores stay a small part of the merged trapezoid.
    // Every level of the panel tree is a dependent step of each sweep (and a front to assemble), so a chain of tall, narrow
    // supernodes -- the rule in these LP factors -- costs far more as levels than as a few hundred stored zeros.
    if (relax > 0) {
        std::vector<int> cfirst;
        for (int k = 0; k < m; ++k) if (start[k]) cfirst.push_back(k);
        const int nch = (int)cfirst.size();
        cfirst.push_back(m);
        int64_t acc_true = 0;                             // true entries of the (merged) chain that ends with chain ch
        int head = 0;
        for (int ch = 0; ch < nch; ++ch) {
            const int a0 = cfirst[ch], b = cfirst[ch + 1];
            int64_t tnz = 0;
            for (int c = a0; c < b; ++c) tnz += cnt[c] + 1;
            if (ch == 0 || head < 0) { head = a0; acc_true = tnz; }
            if (b >= m || parent[b - 1] != b) { head = -1; continue; }
            // parent chain = ch + 1, columns [b, e); its rows below e
            const int e = cfirst[ch + 2];
            int64_t tp = 0;
            for (int c = b; c < e; ++c) tp += cnt[c] + 1;
            int64_t nb = 0;
            for (int i : st[b]) if (i >= e) ++nb;
            const int64_t W = e - head;
            const int64_t stored = W * (W + 1) / 2 + W * nb;
            const int64_t zeros = stored - (acc_true + tp);
            const double frac = (double)zeros / (double)stored;
            const bool ok = W <= 4 || (W <= 16 && frac < 0.5 * relax) || (W <= 48 && frac < 0.2 * relax) || frac < 0.05 * relax;
            if (!ok) { head = -1; continue; }
            std::vector<int> merged;
            merged.reserve((size_t)(W - 1 + nb));
            for (int c = head + 1; c < e; ++c) merged.push_back(c);
            for (int i : st[b]) if (i >= e) merged.push_back(i);
            st[head].swap(merged);
            for (int c = head; c < b; ++c) cnt[c] = (int)((e - 1 - c) + nb);
            start[b] = 0;
            acc_true += tp;                               // head stays: the merged chain now ends with chain ch + 1
            total += zeros;
            if (total > value_cap) return 1;
        }
    }
    // panels: cut every supernode [a, b) into pieces of at most min(wcap, lds_doubles / r) columns
    std::vector<int> pc0;                                   // first column of real panel q
    for (int a = 0; a < m;) {
        int b = a + 1;
        while (b < m && !start[b]) ++b;
        int c = a;
        while (c < b) {
            const int r = cnt[c] + 1;                       // rows of a panel starting at column c: c itself + everything below
            int w = std::min(wcap, std::max(1, lds_doubles / std::max(1, r)));
            w = std::min(w, b - c);
            pc0.push_back(c);
            c += w;
        }
        a = b;
    }
    const int nreal = (int)pc0.size();
    pc0.push_back(m);
    std::vector<int> sn_of((size_t)m, 0);
    for (int q = 0; q < nreal; ++q) for (int c = pc0[q]; c < pc0[q + 1]; ++c) sn_of[c] = q;
    // rows of real panel q = its columns, then the structure of its chain's first column beyond the panel (within a chain
    // struct(c) = {c+1, ...} + struct(c+1), amalgamated chains store the full trapezoid)
    std::vector<std::vector<int>> prow((size_t)nreal);
    std::vector<int> ppar((size_t)nreal, -1);
    {
        int chain_first = 0;
        for (int q = 0; q < nreal; ++q) {
            const int c0 = pc0[q], c1 = pc0[q + 1];
            if (start[c0]) chain_first = c0;
            std::vector<int>& rw = prow[q];
            for (int c = c0; c < c1; ++c) rw.push_back(c);
            for (int i : st[chain_first]) if (i >= c1) rw.push_back(i);
            if ((int)rw.size() > c1 - c0) ppar[q] = sn_of[rw[(size_t)(c1 - c0)]];
        }
    }
    // FAN-IN nodes: a panel with more than `fan_limit` children gets intermediate nodes (w = 0, the parent's rows) that sum
    // groups of `fan` consecutive children, recursively: the extend-add of a front is sequential over its children, so a
    // star of several hundred leaves (the rule in these LP factors) would otherwise be one workgroup's serial loop
    const int fan = 8, fan_limit = 12;
    struct Virt { int owner; std::vector<int> kids; };       // kids: real panel q as q, fan-in node v as -(v + 1)
    std::vector<Virt> virt;
    std::vector<std::vector<int>> attach((size_t)nreal);     // fan-in nodes emitted right after real panel q
    std::vector<int> vpar_real((size_t)nreal, -1);           // fan-in node a real panel reports to (else its real parent)
    std::vector<int> vpar_virt;
    {
        std::vector<std::vector<int>> kids((size_t)nreal);
        for (int q = 0; q < nreal; ++q) if (ppar[q] >= 0) kids[ppar[q]].push_back(q);
        for (int jp = 0; jp < nreal; ++jp) {
            if ((int)kids[jp].size() <= fan_limit) continue;
            std::vector<int> level(kids[jp].begin(), kids[jp].end()), lastreal(kids[jp].begin(), kids[jp].end());
            while ((int)level.size() > fan_limit) {
                std::vector<int> nlevel, nlast;
                for (size_t g = 0; g < level.size(); g += (size_t)fan) {
                    const size_t ge = std::min(level.size(), g + (size_t)fan);
                    if (ge - g == 1) { nlevel.push_back(level[g]); nlast.push_back(lastreal[g]); continue; }
                    const int v = (int)virt.size();
                    virt.push_back(Virt{jp, std::vector<int>(level.begin() + g, level.begin() + ge)});
                    vpar_virt.push_back(-1);
                    for (size_t t = g; t < ge; ++t) { if (level[t] >= 0) vpar_real[level[t]] = v; else vpar_virt[(size_t)(-level[t] - 1)] = v; }
                    attach[(size_t)lastreal[ge - 1]].push_back(v);
                    nlevel.push_back(-(v + 1)); nlast.push_back(lastreal[ge - 1]);
                }
                level.swap(nlevel); lastreal.swap(nlast);
            }
        }
    }
    const int nvirt = (int)virt.size();
    std::vector<int> fid_real((size_t)nreal), fid_virt((size_t)nvirt);
    {
        int id = 0;
        for (int q = 0; q < nreal; ++q) { fid_real[q] = id++; for (int v : attach[q]) fid_virt[v] = id++; }
    }
    S = Supernodes();
    S.m = m; S.nsn = nreal + nvirt; S.nvirtual = nvirt;
    const int nsn = S.nsn;
    S.c0.assign((size_t)nsn, 0); S.w.assign((size_t)nsn, 0); S.parent.assign((size_t)nsn, -1);
    std::vector<int> src((size_t)nsn);                      // rows of final node = rows of this real panel
    for (int q = 0; q < nreal; ++q) {
        const int J = fid_real[q];
        S.c0[J] = pc0[q]; S.w[J] = pc0[q + 1] - pc0[q]; src[J] = q;
        S.parent[J] = vpar_real[q] >= 0 ? fid_virt[vpar_real[q]] : (ppar[q] >= 0 ? fid_real[ppar[q]] : -1);
    }
    for (int v = 0; v < nvirt; ++v) {
        const int J = fid_virt[v];
        S.c0[J] = pc0[virt[v].owner]; S.w[J] = 0; src[J] = virt[v].owner;
        S.parent[J] = vpar_virt[v] >= 0 ? fid_virt[vpar_virt[v]] : fid_real[virt[v].owner];
    }
    S.rowptr.assign((size_t)nsn + 1, 0); S.lptr.assign((size_t)nsn + 1, 0); S.uptr.assign((size_t)nsn + 1, 0);
    for (int J = 0; J < nsn; ++J) {
        if (S.parent[J] >= 0 && S.parent[J] <= J) return 2;                       // (cannot happen)
        const std::vector<int>& rw = prow[src[J]];
        S.rows.insert(S.rows.end(), rw.begin(), rw.end());
        S.rowptr[(size_t)J + 1] = (int64_t)S.rows.size();
        const int64_t r = (int64_t)rw.size(), w = S.w[J], p = r - w;
        S.lptr[(size_t)J + 1] = S.lptr[J] + (r * w + 15) / 16 * 16;                // 128-byte aligned panels and update matrices
        S.uptr[(size_t)J + 1] = S.uptr[J] + (p * p + 15) / 16 * 16;
        if (S.lptr[(size_t)J + 1] + S.uptr[(size_t)J + 1] > value_cap) return 1;
        S.rmax = std::max(S.rmax, (int)r); S.wmax = std::max(S.wmax, (int)w);
        S.panel_max = std::max(S.panel_max, r * w);
        if (S.parent[J] >= 0) S.pcmax = std::max(S.pcmax, (int)p);
        for (int c = 0; c < w; ++c) { const double q = (double)(r - c); S.flops += q * q; S.nnz_factor += (int64_t)q; }
    }
    // children (ascending) and the panel-tree height
    S.childptr.assign((size_t)nsn + 1, 0);
    for (int J = 0; J < nsn; ++J) if (S.parent[J] >= 0) S.childptr[(size_t)S.parent[J] + 1]++;
    for (int J = 0; J < nsn; ++J) { S.max_children = std::max(S.max_children, S.childptr[(size_t)J + 1]); S.childptr[(size_t)J + 1] += S.childptr[J]; }
    S.child.resize((size_t)S.childptr[nsn]);
    {
        std::vector<int> nx(S.childptr.begin(), S.childptr.end() - 1), hgt((size_t)nsn, 1);
        for (int J = 0; J < nsn; ++J) {
            const int pj = S.parent[J];
            if (pj >= 0) { S.child[(size_t)nx[pj]++] = J; hgt[pj] = std::max(hgt[pj], hgt[J] + 1); }
            S.height = std::max(S.height, hgt[J]);
        }
    }
    // crel: where each row below a panel sits in its parent's front
    S.crel.assign(S.rows.size(), -1);
    {
        std::vector<int> where((size_t)m, -1);
        for (int J = 0; J < nsn; ++J) {
            if (S.childptr[(size_t)J + 1] == S.childptr[J]) continue;
            const int64_t r0 = S.rowptr[J], r = S.rowptr[(size_t)J + 1] - r0;
            for (int64_t a = 0; a < r; ++a) where[S.rows[(size_t)(r0 + a)]] = (int)a;
            for (int t = S.childptr[J]; t < S.childptr[(size_t)J + 1]; ++t) {
                const int K = S.child[(size_t)t];
                const int64_t k0 = S.rowptr[K], rk = S.rowptr[(size_t)K + 1] - k0;
                for (int64_t i = S.w[K]; i < rk; ++i) {
                    const int a = where[S.rows[(size_t)(k0 + i)]];
                    if (a < 0) return 2;                         // (cannot happen: the child's rows are a subset of the parent's)
                    S.crel[(size_t)(k0 + i)] = a;
                }
            }
            for (int64_t a = 0; a < r; ++a) where[S.rows[(size_t)(r0 + a)]] = -1;
        }
    }
    S.diagpos.resize((size_t)m);
    for (int J = 0; J < nsn; ++J)
        for (int c = 0; c < S.w[J]; ++c) S.diagpos[(size_t)S.c0[J] + c] = S.lptr[J] + (int64_t)c * S.w[J] + c;
    return 0;
}

// Critical path of the panel tree for a cost model: the largest, over root-to-leaf paths, of the sum of (rows of the front)^2,
// and the number of panels on that path.
inline void critical_path(const Supernodes& S, double& area, int& levels) {
    std::vector<double> acc((size_t)S.nsn, 0.0);
    std::vector<int> lev((size_t)S.nsn, 0);
    area = 0.0; levels = 0;
    for (int J = 0; J < S.nsn; ++J) {                        // children precede parents
        const double r = (double)(S.rowptr[(size_t)J + 1] - S.rowptr[J]);
        acc[J] += r * r; lev[J] += 1;
        if (acc[J] > area) { area = acc[J]; levels = lev[J]; }
        const int pj = S.parent[J];
        if (pj >= 0 && acc[J] > acc[pj]) { acc[pj] = acc[J]; lev[pj] = lev[J]; }
    }
}

}  // namespace sym
}  // namespace ipm
