// ff_schedule.h -- host side of the fused formation + factorization (form_factor.h): the ORDERED WORK LIST the
// persistent workers draw from.  Plain C++ (no HIP): built into libipm_hip.so, and checked on the CPU through
// ipm_debug_ff_schedule (tests/test_ff_schedule.py).
//
// What is being scheduled (dense handles, 128 x 128 tiles (i, c), i >= c, of B = A D^2 A^T and of its Cholesky factor;
// replaces main.py:224 + the factorization inside main.py:180/:226 of the reference, fused):
//   F(i,c,q)          one of Q K-chunks of the formation of the tile PAIR (i,c), (i+1,c), i even: raw partial tiles into the
//                     slabs (tile, q) of the two tiles (a half above the diagonal or below the matrix is dropped)
//   T(i,c,[j0,j1))    tile (i,c) -= sum_{j0<=j<j1} L(i,j) L(c,j)^T, optionally + the Q formation slabs (ADD_BASE, once per
//                     tile, any time after its F chunks), optionally followed by the panel solve L(i,c) = tile inv(L(c,c))^T
//                     (PANEL, once, after everything else of the tile and after the diagonal block c is factored)
// and, outside the list, the PIVOT CHAIN on its own stream and its own CUs: potrf(k) of diagonal block k, the panel solve
// of tile (k+1,k) and the update of tile (k+1,k+1) by column k (the existing kernels of potrf_f64.h / gemm_nt_f64.h).
// The chain's tiles get everything else from the workers: tile (k+1,k) columns [0,k), tile (k,k) columns [0,k-1).
//
// Why an ordered list and one ticket counter: a worker takes the next item of the list, waits (bounded spin) for what the
// item needs, runs it.  Everything an item needs is produced by items EARLIER in the list (or by the chain, which itself
// only needs earlier items), and earlier items are held by workgroups that are running, so the launch cannot deadlock
// whatever the residency or timing; a bad order only costs waiting.  The order is the start order of a list scheduling of
// the item DAG by bottom level on durations calibrated with an item trace of the launch itself (ff_build_schedule below;
// tools/ff_replay.py replays a list under the same model and reproduces the measured chain to ~1 %).
// The list is a pure function of (nblk, Q, workers, model): the arithmetic order of every tile is fixed, results are
// bitwise reproducible.
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <math.h>

#include <algorithm>
#include <limits>
#include <queue>
#include <set>
#include <string>
#include <vector>

namespace ipm {

struct FFItem {
    unsigned char type;        // FF_F / FF_T
    unsigned char i, c;        // tile
    unsigned char q;           // F: K-chunk (slab index)
    union {
        struct { unsigned char j0, j1, flags, seq; } t;   // T: column range of L applied; FF_INIT | FF_ADD_BASE | FF_PANEL | FF_SIG_DIAG;
                                                          //    1-based sequence number among the tile's T items (tprog hand-off)
        struct { unsigned short s0, s1; } f;              // F: stage range [s0, s1) of the K loop (BK = 32 stages)
    };
};
enum { FF_F = 0, FF_T = 1, FF_D = 2 };     // FF_D(i): max diag(B) over the rows of block i straight from A and d (chain_mode 1)
enum { FF_INIT = 1, FF_ADD_BASE = 2, FF_PANEL = 4, FF_SIG_DIAG = 8 };     // SIG_DIAG: the finished diagonal tile is handed to the chain (dready[i] += 10)
constexpr int FF_MAX_NBLK = 96;

#if defined(__HIPCC__)
#define FF_HD __host__ __device__
#else
#define FF_HD
#endif
FF_HD inline int ff_tile(int i, int c) { return i * (i + 1) / 2 + c; }

struct FFModel {                       // durations in microseconds, calibrated on an item trace of the launch itself (MI355X, 512-thread
                                       // workgroups, one per CU; tools/ff_trace.py -> profiles/r04_ff_item_trace_baseline.txt, tools/ff_replay.py)
    double f_over = 17.9, f_stage = 3.91;      // formation chunk: fixed + per BK = 16 stage of a 256 x 128 tile pair
    double t_over = 4.1, t_col = 15.8;         // update item: fixed + per 128-column block of L applied
    double t_rmw = 2.0, t_panel = 21.0;        // reading the tile back (all but its first item); the product with inv(L_cc)
    double t_base = 13.0;                      // adding ONE formation slab (53 us for four before the register-major slab layout)
    double d_item = 140.0;                     // one FF_D item (diag(B) of 128 rows straight from A and d)
    double gap = 0.8, handoff = 1.5;           // end of an item -> next ticket; counter bump -> visible to a spinning consumer
    // the pivot chain: potrf_diag of one block, and -- chain_mode 0 only -- its two small GEMM launches with the launch gaps
    double potrf = 36.0, cpanel = 8.0, cupdate = 5.0, g_potrf_panel = 3.5, g_panel_update = 3.5, g_update_potrf = 4.0;
    double chain_start = 220.0;                // chain_mode 0: ff_maxdiag_kernel in front of the first potrf_diag
    // The chain = per step k: potrf_diag of block k, the panel solve of tile (k+1,k), the update of tile (k+1,k+1) by column k.
    // chain_mode 0: THREE LAUNCHES per step on a second stream, on CUs the worker launch leaves free (one per shader engine), with
    //               ff_maxdiag_kernel in front;
    // chain_mode 1: TWO PERSISTENT launches enqueued before the workers (ff_chain_kernel: one workgroup, the diagonal blocks;
    //               ff_crit_kernel: four workgroups, the two small products of every step in 32-row strips) -- no launch gaps, and
    //               every other CU works; FF_D items (max diag(B), the pivot guard's scale) head the list.
    int chain_mode = 0;
    int batch = 4;                     // columns of L per bulk update item
    int tail = 2;                      // newest columns of a tile applied one at a time (a batch that ends at column j cannot start before
                                       // L(.,j) exists, i.e. one pipeline step before the tile's final item is due)
    double stagger = 0.3;              // formation chunk lengths spread over (1 -+ stagger) of the mean, phase per pair: the workers do not
                                       // finish their chunks in lockstep
    int nstages = 512;                 // K / 16 of the formation
    int q_last = 0;                    // > 0: chunks per pair for the pairs of the last two block rows (shorter chunks pack the end)
    // calibration of chain_mode 1 (profiles/r04_ff_item_trace_roles_kernel.txt: the replay of the list under these values ends
    // at 3347 us, the launch it models at 3355 us)
    void roles_calibration() {
        chain_mode = 1;
        f_over = 6.1; f_stage = 4.02; t_over = 7.0; t_col = 15.9; t_rmw = 1.0; t_panel = 18.3; t_base = 6.75; d_item = 130.0;
        potrf = 36.8; cpanel = 7.3; cupdate = 7.3; handoff = 1.0; chain_start = 0.0;
    }
};

struct FFSchedule {
    std::vector<FFItem> items;
    std::vector<int> tile_items;       // [ntile] T items per tile (what a consumer of the finished tile waits for)
    std::vector<int> tile_q;           // [ntile] formation chunks (slabs) of the tile
    double makespan_us = 0.0;          // simulated end of the factorization
    double form_end_us = 0.0;          // simulated end of the last formation chunk
};

// columns of L the WORKERS apply to tile (i,c): the chain applies column c-1 to its diagonal tile itself
inline int ff_limit(int i, int c) { return (i == c) ? (c > 0 ? c - 1 : 0) : c; }
inline bool ff_needs_panel(int i, int c) { return i > c + 1; }

// The work list = the start order of a LIST SCHEDULING of the item DAG on W workers by bottom level (longest path to the end of the
// factorization, on the calibrated durations above): a worker that comes free takes the open update item with the highest bottom level
// whose inputs are complete, or -- when a formation chunk has a higher one, or no update is ready -- the next formation chunk; with
// the formation exhausted it takes the item that becomes ready first and waits inside it.  Everything an item waits for is produced
// by items EARLIER in the list or by the chain, which itself only waits for earlier items: no deadlock at any timing.
// Q: formation chunks per pair; Qmax: slab capacity per tile.
inline void ff_build_schedule(int nblk, int Q, int W, const FFModel& M_in, FFSchedule& out, int Qmax = 0) {
    if (Qmax < Q) Qmax = Q;
    const double INF = std::numeric_limits<double>::infinity();
    const int ntile = nblk * (nblk + 1) / 2;
    FFModel Mx = M_in;
    if (const char* e = getenv("IPM_FF_BATCH")) Mx.batch = std::max(1, atoi(e));
    if (const char* e = getenv("IPM_FF_TAIL")) Mx.tail = std::max(1, atoi(e));
    if (const char* e = getenv("IPM_FF_STAGGER")) Mx.stagger = std::min(0.9, std::max(0.0, atof(e)));
    if (const char* e = getenv("IPM_FF_Q_LAST")) Mx.q_last = std::max(0, atoi(e));
    if (const char* e = getenv("IPM_FF_MODEL")) {             // tuning aid: "name=value,name=value" over the durations above
        std::string spec(e);
        size_t p0 = 0;
        while (p0 < spec.size()) {
            size_t p1 = spec.find(',', p0); if (p1 == std::string::npos) p1 = spec.size();
            const std::string kv = spec.substr(p0, p1 - p0);
            const size_t eq = kv.find('=');
            if (eq != std::string::npos) {
                const std::string k = kv.substr(0, eq); const double v = atof(kv.c_str() + eq + 1);
                struct { const char* n; double* p; } tab[] = {{"f_over", &Mx.f_over}, {"f_stage", &Mx.f_stage}, {"t_over", &Mx.t_over}, {"t_col", &Mx.t_col},
                    {"t_rmw", &Mx.t_rmw}, {"t_panel", &Mx.t_panel}, {"t_base", &Mx.t_base}, {"d_item", &Mx.d_item}, {"potrf", &Mx.potrf},
                    {"cpanel", &Mx.cpanel}, {"cupdate", &Mx.cupdate}, {"handoff", &Mx.handoff}, {"gap", &Mx.gap}};
                for (auto& t : tab) if (k == t.n) *t.p = v;
            }
            p0 = p1 + 1;
        }
    }
    const FFModel& M = Mx;
    const int mode = M.chain_mode;
    enum Kind { K_F, K_T, K_D, K_POTRF, K_CPANEL, K_CUPDATE };
    struct Node {
        Kind kind; int i = 0, c = 0, q = 0, j0 = 0, j1 = 0, flags = 0, seq = 0, s0 = 0, s1 = 0;
        double dur = 0.0, bl = 0.0, est = 0.0, fin = 0.0;
        int npred = 0, left = 0;
        std::vector<int> succ;
    };
    std::vector<Node> nd;
    nd.reserve((size_t)ntile * 6 + 4 * (size_t)nblk);
    auto add = [&](Kind k, double dur) { Node x; x.kind = k; x.dur = dur; x.fin = INF; nd.push_back(x); return (int)nd.size() - 1; };
    auto edge = [&](int a, int b) { nd[(size_t)a].succ.push_back(b); nd[(size_t)b].npred++; };
    // ---- chain
    std::vector<int> potrf((size_t)nblk), cpan, cupd;
    const double g1 = mode ? M.handoff : M.g_potrf_panel, g2 = mode ? M.handoff : M.g_panel_update, g3 = mode ? M.handoff : M.g_update_potrf;
    for (int k = 0; k < nblk; ++k) { potrf[(size_t)k] = add(K_POTRF, M.potrf + g1); nd.back().i = k; }
    cpan.resize((size_t)nblk - 1); cupd.resize((size_t)nblk - 1);
    for (int k = 0; k + 1 < nblk; ++k) { cpan[(size_t)k] = add(K_CPANEL, M.cpanel + g2); nd.back().i = k; }
    for (int k = 0; k + 1 < nblk; ++k) { cupd[(size_t)k] = add(K_CUPDATE, M.cupdate + g3); nd.back().i = k; }
    for (int k = 0; k + 1 < nblk; ++k) { edge(potrf[(size_t)k], cpan[(size_t)k]); edge(cpan[(size_t)k], cupd[(size_t)k]); edge(cupd[(size_t)k], potrf[(size_t)k + 1]); }
    // ---- update items per tile
    std::vector<std::vector<int>> titems((size_t)ntile);
    std::vector<int> lprod((size_t)ntile, -1);            // node that makes L(r,c) final (c < r)
    for (int i = 0; i < nblk; ++i)
        for (int c = 0; c <= i; ++c) {
            const int lim = ff_limit(i, c);
            const bool pan = ff_needs_panel(i, c);
            std::vector<std::pair<int, int>> cuts;
            const int nb_end = std::max(0, lim - M.tail);
            int j = 0;
            while (j + M.batch <= nb_end) { cuts.emplace_back(j, j + M.batch); j += M.batch; }
            if (j < nb_end) { cuts.emplace_back(j, nb_end); j = nb_end; }
            while (j < lim) { cuts.emplace_back(j, j + 1); ++j; }
            if (cuts.empty()) cuts.emplace_back(0, 0);
            std::vector<int>& ids = titems[(size_t)ff_tile(i, c)];
            for (size_t s = 0; s < cuts.size(); ++s) {
                const bool last = s + 1 == cuts.size();
                const int id = add(K_T, M.t_over + M.t_col * (cuts[s].second - cuts[s].first) + (s ? M.t_rmw : 0.0) + ((last && pan) ? M.t_panel : 0.0) + M.handoff + M.gap);
                Node& x = nd[(size_t)id];
                x.i = i; x.c = c; x.j0 = cuts[s].first; x.j1 = cuts[s].second; x.seq = (int)s + 1;
                x.flags = (s == 0 ? FF_INIT : 0) | ((last && pan) ? FF_PANEL : 0) | ((last && i == 0 && c == 0) ? FF_SIG_DIAG : 0);
                if (s) edge(ids.back(), id);
                ids.push_back(id);
            }
            if (pan) lprod[(size_t)ff_tile(i, c)] = ids.back();
        }
    for (int k = 0; k + 1 < nblk; ++k) {
        lprod[(size_t)ff_tile(k + 1, k)] = cpan[(size_t)k];
        edge(titems[(size_t)ff_tile(k + 1, k)].back(), cpan[(size_t)k]);
        edge(titems[(size_t)ff_tile(k + 1, k + 1)].back(), cupd[(size_t)k]);
    }
    edge(titems[0].back(), potrf[0]);
    for (int t = 0; t < ntile; ++t)
        for (int id : titems[(size_t)t]) {
            const int i = nd[(size_t)id].i, c = nd[(size_t)id].c, j1 = nd[(size_t)id].j1;
            if (j1 > nd[(size_t)id].j0) {
                edge(lprod[(size_t)ff_tile(i, j1 - 1)], id);
                if (c != i) edge(lprod[(size_t)ff_tile(c, j1 - 1)], id);
            }
            if (nd[(size_t)id].flags & FF_PANEL) edge(potrf[(size_t)c], id);
        }
    // ---- which update item of a tile adds the formation slabs: the one before the final item (the formation is then needed as late
    //      as possible without sitting on the tile's last, urgent step)
    std::vector<int> base_item((size_t)ntile);
    for (int t = 0; t < ntile; ++t) { const std::vector<int>& ids = titems[(size_t)t]; base_item[(size_t)t] = ids.size() == 1 ? ids[0] : ids[ids.size() - 2]; }
    // ---- formation chunks of the tile pairs (2r, c), (2r + 1, c)
    std::vector<int> fch;
    out.tile_q.assign((size_t)ntile, 0);
    {
        int npairs = 0;
        for (int i = 0; i < nblk; i += 2)
            for (int c = 0; c <= std::min(i + 1, nblk - 1); ++c) {
                int q = (M.q_last > 0 && i + 2 >= nblk) ? M.q_last : Q;
                q = std::max(1, std::min(std::min(q, Qmax), M.nstages));
                const double phi = std::fmod(npairs * 0.381966, 1.0);
                ++npairs;
                std::vector<int> cut((size_t)q + 1, 0);
                double tot = 0.0, acc = 0.0;
                std::vector<double> len((size_t)q);
                for (int k = 0; k < q; ++k) { len[(size_t)k] = 1.0 + M.stagger * (2.0 * std::fmod((double)k / q + phi, 1.0) - 1.0); tot += len[(size_t)k]; }
                for (int k = 0; k < q; ++k) { acc += len[(size_t)k]; cut[(size_t)k + 1] = (int)(M.nstages * acc / tot + 0.5); }
                cut[(size_t)q] = M.nstages;
                for (int k = 0; k < q; ++k) {
                    const int id = add(K_F, M.f_over + M.f_stage * (cut[(size_t)k + 1] - cut[(size_t)k]) + M.gap);
                    Node& x = nd[(size_t)id];
                    x.i = i; x.c = c; x.q = k; x.s0 = cut[(size_t)k]; x.s1 = cut[(size_t)k + 1];
                    fch.push_back(id);
                    for (int r = i; r <= i + 1; ++r)
                        if (r >= c && r < nblk) { edge(id, base_item[(size_t)ff_tile(r, c)]); out.tile_q[(size_t)ff_tile(r, c)]++; }
                }
            }
    }
    for (int t = 0; t < ntile; ++t) { Node& x = nd[(size_t)base_item[(size_t)t]]; x.flags |= FF_ADD_BASE; x.dur += M.t_base * out.tile_q[(size_t)t]; }
    // ---- FF_D items (chain_mode 1): the pivot guard's scale before the first diagonal block is factored
    std::vector<int> ditems;
    if (mode) for (int i = 0; i < nblk; ++i) { const int id = add(K_D, M.d_item + M.gap); nd[(size_t)id].i = i; edge(id, potrf[0]); ditems.push_back(id); }
    // ---- bottom levels
    const int N = (int)nd.size();
    {
        std::vector<int> order, indeg((size_t)N), stack;
        order.reserve((size_t)N);
        for (int u = 0; u < N; ++u) { indeg[(size_t)u] = nd[(size_t)u].npred; if (!indeg[(size_t)u]) stack.push_back(u); }
        while (!stack.empty()) {
            const int u = stack.back(); stack.pop_back(); order.push_back(u);
            for (int v : nd[(size_t)u].succ) if (--indeg[(size_t)v] == 0) stack.push_back(v);
        }
        if ((int)order.size() != N) { out.items.clear(); return; }          // (cannot happen: the DAG is acyclic by construction)
        for (size_t k = order.size(); k-- > 0;) {
            Node& x = nd[(size_t)order[k]];
            double b = 0.0;
            for (int v : x.succ) b = std::max(b, nd[(size_t)v].bl);
            x.bl = x.dur + b;
        }
    }
    // ---- list scheduling
    typedef std::pair<double, int> PI;                         // (-bottom level, node): min-heap = highest bottom level first, ties to the lower id
    std::priority_queue<PI, std::vector<PI>, std::greater<PI>> avail_F;
    std::set<PI> avail_T;
    for (int id : fch) avail_F.push(PI(-nd[(size_t)id].bl, id));
    for (int u = 0; u < N; ++u) nd[(size_t)u].left = nd[(size_t)u].npred;
    std::vector<int> rel;                                      // release worklist (iterative: the chain releases recursively)
    auto release = [&](int u0) {
        rel.clear(); rel.push_back(u0);
        while (!rel.empty()) {
            const int u = rel.back(); rel.pop_back();
            for (int v : nd[(size_t)u].succ) {
                Node& y = nd[(size_t)v];
                y.est = std::max(y.est, nd[(size_t)u].fin);
                if (--y.left == 0) {
                    if (y.kind == K_T) avail_T.insert(PI(-y.bl, v));
                    else if (y.kind != K_F && y.kind != K_D) {     // chain nodes run by themselves as soon as their inputs are there
                        const double st = (y.kind == K_POTRF && y.i == 0 && !mode) ? std::max(y.est, M.chain_start) : y.est;
                        y.fin = st + y.dur;
                        rel.push_back(v);
                    }
                }
            }
        }
    };
    for (int t = 0; t < ntile; ++t) { const int id = titems[(size_t)t][0]; if (nd[(size_t)id].left == 0) avail_T.insert(PI(-nd[(size_t)id].bl, id)); }
    typedef std::pair<double, int> Ev;
    std::priority_queue<Ev, std::vector<Ev>, std::greater<Ev>> free_at;
    for (int w = 0; w < W; ++w) free_at.push(Ev(0.0, w));
    out.items.clear();
    size_t nT = 0, doneT = 0, nextD = 0;
    for (int t = 0; t < ntile; ++t) nT += titems[(size_t)t].size();
    std::vector<int> order_out;
    int guard = 0;
    while (doneT < nT || !avail_F.empty() || nextD < ditems.size()) {
        const Ev ev = free_at.top(); free_at.pop();
        const double t = ev.first;
        if (nextD < ditems.size()) {
            const int v = ditems[nextD++];
            nd[(size_t)v].fin = t + nd[(size_t)v].dur;
            order_out.push_back(v); release(v);
            free_at.push(Ev(nd[(size_t)v].fin, ev.second));
            continue;
        }
        int pick = -1;
        for (const PI& p : avail_T) if (nd[(size_t)p.second].est <= t) { pick = p.second; break; }
        if (pick >= 0 && !avail_F.empty() && -avail_F.top().first > nd[(size_t)pick].bl) pick = -1;     // a formation chunk is more urgent
        if (pick < 0 && !avail_F.empty()) {
            const int v = avail_F.top().second; avail_F.pop();
            nd[(size_t)v].fin = t + nd[(size_t)v].dur;
            out.form_end_us = std::max(out.form_end_us, nd[(size_t)v].fin);
            order_out.push_back(v); release(v);
            free_at.push(Ev(nd[(size_t)v].fin, ev.second));
            continue;
        }
        if (pick < 0) {
            if (avail_T.empty()) {                                     // nothing schedulable yet (an item in flight will release the next ones)
                if (++guard > 64 * W * nblk) break;
                free_at.push(Ev(t + 5.0, ev.second));
                continue;
            }
            double best = INF;                                         // formation exhausted: the item that becomes ready first; wait inside it
            for (const PI& p : avail_T) if (nd[(size_t)p.second].est < best) { best = nd[(size_t)p.second].est; pick = p.second; }
        }
        Node& x = nd[(size_t)pick];
        avail_T.erase(PI(-x.bl, pick));
        x.fin = std::max(t, x.est) + x.dur;
        order_out.push_back(pick); ++doneT;
        release(pick);
        free_at.push(Ev(x.fin, ev.second));
    }
    // ---- encode
    out.tile_items.assign((size_t)ntile, 0);
    for (int t = 0; t < ntile; ++t) out.tile_items[(size_t)t] = (int)titems[(size_t)t].size();
    out.makespan_us = 0.0;
    for (int v : order_out) {
        const Node& x = nd[(size_t)v];
        FFItem it{};
        it.i = (unsigned char)x.i; it.c = (unsigned char)x.c;
        if (x.kind == K_F) { it.type = FF_F; it.q = (unsigned char)x.q; it.f.s0 = (unsigned short)x.s0; it.f.s1 = (unsigned short)x.s1; }
        else if (x.kind == K_D) { it.type = FF_D; }
        else { it.type = FF_T; it.t.j0 = (unsigned char)x.j0; it.t.j1 = (unsigned char)x.j1; it.t.flags = (unsigned char)x.flags; it.t.seq = (unsigned char)x.seq; }
        out.items.push_back(it);
    }
    for (int k = 0; k < nblk; ++k) out.makespan_us = std::max(out.makespan_us, nd[(size_t)potrf[(size_t)k]].fin);
}

}  // namespace ipm
