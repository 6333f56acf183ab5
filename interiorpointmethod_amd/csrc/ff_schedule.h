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
// whatever the residency or timing; a bad order only costs waiting.  The order is the start order of a discrete-event
// simulation of the machine (workers, chain, durations measured on MI355X) under a priority rule:
//   (0) tiles in the chain's window (their column is at most 2 steps ahead) as soon as anything can be applied to them,
//   (1) final batches (everything up to the tile's last column is available),
//   (2) batches of >= FF_BATCH columns for tiles further away (few read-modify-write passes per tile),
//   (3) the next formation chunk (column-major, so that columns become complete in the order the chain needs them),
//   (4) anything left once the formation is exhausted.
// The list is a pure function of (nblk, Q, workers): the arithmetic order of every tile is fixed, results are bitwise
// reproducible.
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <limits>
#include <queue>
#include <set>
#include <vector>

namespace ipm {

struct FFItem {
    unsigned char type;        // FF_F / FF_T
    unsigned char i, c;        // tile
    unsigned char q;           // F: K-chunk (slab index)
    union {
        struct { unsigned char j0, j1, flags, seq; } t;   // T: column range of L applied; FF_INIT | FF_ADD_BASE | FF_PANEL | FF_SIG_DIAG0;
                                                          //    1-based sequence number among the tile's T items (tprog hand-off)
        struct { unsigned short s0, s1; } f;              // F: stage range [s0, s1) of the K loop (BK = 32 stages)
    };
};
enum { FF_F = 0, FF_T = 1 };
enum { FF_INIT = 1, FF_ADD_BASE = 2, FF_PANEL = 4, FF_SIG_DIAG0 = 8 };
constexpr int FF_BATCH = 4;            // columns of L per deferred batch (K = 512)
constexpr int FF_WINDOW = 2;           // chain look-ahead: columns within this many steps are served at once
constexpr int FF_MAX_NBLK = 96;

#if defined(__HIPCC__)
#define FF_HD __host__ __device__
#else
#define FF_HD
#endif
FF_HD inline int ff_tile(int i, int c) { return i * (i + 1) / 2 + c; }

struct FFModel {                       // durations in microseconds (MI355X, one 512-thread workgroup per CU)
    double stage = 4.4;                // one BK = 32 stage of a 128 x 128 tile (update / panel items; 4.2-4.4 standalone)
    double pstage = 3.95;              // one BK = 16 stage of a 256 x 128 tile PAIR (formation): 3.91 us standalone; inside the fused
                                       // launch 9280 cycles at the 2.1 GHz the chip holds there = 4.4 us (in-kernel cycle profile)
    // The durations below are DELIBERATELY on the optimistic side of what the cycle profile shows (T overhead 10.5 + 3 us,
    // panel product 19 us, formation stage 4.4 us): the list is drawn in order with blocking waits, and an order that
    // expects the chain's inputs early puts a worker in front of each of them before it is ready, so that the hand-off
    // costs no pick-up time.  Measured at 4096 x 8192, 224 workers: 3.91 ms with these values, 4.19-4.26 ms with the
    // profile's own; drawing ready items without waiting (IPM_FF_CLAIM=1) 4.88 ms -- the launch is bound by the chain's
    // dependencies, not by the order of the list.
    double f_overhead = 6.0;           // F chunk: prologue + slab stores + drain
    double t_overhead = 7.0;           // T item: ticket, wait + acquire, tile load, combine, store + drain
    double t_base = 2.0;               // reading Q slabs
    double t_panel = 16.0;             // second product with inv(L_cc) (4 stages + staging through LDS)
    int batch = 4, window = 2;         // columns per deferred batch; chain look-ahead (FF_BATCH / FF_WINDOW)
    int q_first = 8, q_second = 4;     // formation chunks per tile for the block rows 0-3 / 4-7 (capped by the slab capacity): the chain
                                       // cannot start before the first diagonal tile is formed -- at 0.98 ms of a 3.96 ms launch with
                                       // four staggered chunks (profiles/r03_ff_timeline_iter.txt).  Measured, worker launch in ms for
                                       // (q_first, q_second) = (4,4) / (8,4) / (16,8) / (16,16): 3.99 / 3.91 / 4.25 / 4.22 -- an even
                                       // earlier start only brings the read-modify-write passes of the updates forward
    double potrf = 38.0, crit_panel = 8.0, crit_update = 6.0, boundary = 3.0;
    int f_stages = 128;                // stages per F chunk (set by the caller: ceil(K / 16 / Q))
    int nstages = 512;                 // K / 16 of the formation
    int stagger = 1;                   // spread the chunk lengths of the first band (see ff_build_schedule)
    int row_weight = 10, col_weight = 40;   // formation order key = row_weight (i - 1) + col_weight c (see ff_build_schedule)
};

struct FFSchedule {
    std::vector<FFItem> items;
    std::vector<int> tile_items;       // [ntile] T items per tile (what the chain waits for on its tiles)
    std::vector<int> tile_q;           // [ntile] formation chunks (slabs) of the tile: more for the tiles the chain needs first
    double makespan_us = 0.0;          // simulated end of the factorization
    double form_end_us = 0.0;          // simulated end of the last formation chunk
};

// columns of L the WORKERS apply to tile (i,c): the chain applies column c-1 to its diagonal tile itself
inline int ff_limit(int i, int c) { return (i == c) ? (c > 0 ? c - 1 : 0) : c; }
inline bool ff_needs_panel(int i, int c) { return i > c + 1; }

// Q: formation chunks per tile (ordinary tiles); Qmax: slab capacity per tile (>= Q; the first block rows use up to it).
inline void ff_build_schedule(int nblk, int Q, int W, const FFModel& M_in, FFSchedule& out, int Qmax = 0) {
    if (Qmax < Q) Qmax = Q;
    const double INF = std::numeric_limits<double>::infinity();
    const int ntile = nblk * (nblk + 1) / 2;
    const bool trace = getenv("IPM_FF_TRACE") != nullptr;
    FFModel Mx = M_in;
    if (const char* e = getenv("IPM_FF_ROW_WEIGHT")) Mx.row_weight = atoi(e);
    if (const char* e = getenv("IPM_FF_COL_WEIGHT")) Mx.col_weight = atoi(e);
    if (const char* e = getenv("IPM_FF_STAGGER")) Mx.stagger = atoi(e);
    if (const char* e = getenv("IPM_FF_BATCH")) Mx.batch = std::max(1, atoi(e));
    if (const char* e = getenv("IPM_FF_WINDOW")) Mx.window = std::max(1, atoi(e));
    if (const char* e = getenv("IPM_FF_Q_FIRST")) Mx.q_first = std::max(1, atoi(e));
    if (const char* e = getenv("IPM_FF_Q_SECOND")) Mx.q_second = std::max(1, atoi(e));
    const FFModel& M = Mx;
    struct Tile {
        int i, c, limit; bool panel;
        int f_sched = 0; double f_time = 0.0;          // F chunks scheduled, latest finish
        int qn = 0;                                    // F chunks of the tile
        bool base_in = false, paneled = false;
        int applied = 0, nitems = 0;
        double ready = 0.0;                            // finish time of the last T item scheduled on the tile
        bool complete() const { return base_in && applied == limit && (!panel || paneled); }
    };
    std::vector<Tile> T((size_t)ntile);
    for (int i = 0; i < nblk; ++i)
        for (int c = 0; c <= i; ++c) { Tile& t = T[(size_t)ff_tile(i, c)]; t.i = i; t.c = c; t.limit = ff_limit(i, c); t.panel = ff_needs_panel(i, c); }
    // rowfin[r][j]: time tile (r, j) became L (j < r)
    std::vector<std::vector<double>> rowfin((size_t)nblk);
    for (int r = 0; r < nblk; ++r) rowfin[(size_t)r].assign((size_t)r + 1, INF);
    std::vector<int> rf((size_t)nblk, 0);              // leading tiles of row r final at the current time
    std::vector<double> potrf_done((size_t)nblk, INF), potrf_start((size_t)nblk, INF);
    // Formation order = the order in which the factorization needs the tiles.  The pivot chain reaches column c at about
    // t0 + c * step; row i does not have to be current before the chain gets to step i - 1, and its own pipeline (update +
    // panel solve of tile (i,c), needing L(i,c-1)) runs at about half a chain step per column, entirely BEHIND the chain if
    // it starts at t0 + (i - 1) * step / 2.  Tile (i,c) is therefore needed around (i - 1) / 2 + c / 2 chain steps: the
    // demand grows linearly in time like the formation's supply does (half of the tiles have i + c <= nblk - 1), whereas a
    // column-major order needs 3/4 of all tiles by half time and leaves the chain waiting for the formation.
    // Inside a band of W / Q consecutive tiles the chunks go q-major, so the W concurrent items are W / Q tiles x Q chunks.
    std::vector<FFItem> forder;
    {
        // formation items are tile PAIRS (2r, c), (2r + 1, c) (form_factor.h: 256 x 128 blocks); a pair is needed when its
        // more urgent tile is
        struct Pair { int key, i, c; };
        std::vector<Pair> order;
        for (int i = 0; i < nblk; i += 2)
            for (int c = 0; c <= std::min(i + 1, nblk - 1); ++c) {
                int key = 1 << 30;
                if (c <= i) key = std::min(key, M.row_weight * std::max(i - 1, 0) + M.col_weight * c);
                if (i + 1 < nblk) key = std::min(key, M.row_weight * i + M.col_weight * c);
                order.push_back(Pair{key, i, c});
            }
        std::stable_sort(order.begin(), order.end(), [&](const Pair& a, const Pair& b) { return a.key != b.key ? a.key < b.key : a.c < b.c; });
        // Chunks per pair: Q, but more for the block rows the chain needs first (FFModel::q_first / q_second), so that the
        // first diagonal tiles are complete after ~150 us instead of a whole long chunk.
        // Chunk boundaries in (BK = 16) stages.  Uniform, except for the first band of ordinary pairs (the items most workers
        // start with): there the chunk lengths are spread over 0.4 .. 1.6 of the mean, so that the workers do not finish their
        // formation chunks in lockstep -- with equal chunks all of them are deaf for one whole chunk (hundreds of microseconds)
        // at the same time, and everything the chain waits for waits with them; staggered, one worker comes free every
        // microsecond or so.
        const int ns = M.nstages;
        auto pair_q = [&](const Pair& p) { const int q = p.i < 4 ? M.q_first : (p.i < 8 ? M.q_second : Q); return std::max(1, std::min(std::min(q, Qmax), ns)); };
        size_t b0 = 0;
        bool first_band = true;
        while (b0 < order.size()) {
            const int qb = pair_q(order[b0]);
            const size_t band = (size_t)std::max(1, W / qb);
            size_t b1 = b0;
            while (b1 < order.size() && b1 < b0 + band && pair_q(order[b1]) == qb) ++b1;
            std::vector<std::vector<int>> cut(b1 - b0, std::vector<int>((size_t)qb + 1, 0));
            const bool stag = qb == Q && first_band && Q > 1 && M.stagger;
            if (qb == Q) first_band = false;
            for (size_t t = b0; t < b1; ++t) {
                std::vector<double> len((size_t)qb, 1.0);
                if (stag) {
                    const double phi = (double)(t - b0) / (double)(b1 - b0) / qb;
                    for (int q = 0; q < qb; ++q) { double u = (double)q / qb + phi; u -= (double)(int)u; len[(size_t)q] = 0.4 + 1.2 * u; }
                }
                double tot = 0.0; for (double v : len) tot += v;
                double acc = 0.0;
                for (int q = 0; q < qb; ++q) { acc += len[(size_t)q]; cut[t - b0][(size_t)q + 1] = (int)(ns * acc / tot + 0.5); }
                cut[t - b0][(size_t)qb] = ns;
                if (order[t].c <= order[t].i) T[(size_t)ff_tile(order[t].i, order[t].c)].qn = qb;
                if (order[t].i + 1 < nblk) T[(size_t)ff_tile(order[t].i + 1, order[t].c)].qn = qb;
            }
            for (int q = 0; q < qb; ++q)
                for (size_t t = b0; t < b1; ++t) {
                    FFItem it{}; it.type = FF_F; it.i = (unsigned char)order[t].i; it.c = (unsigned char)order[t].c; it.q = (unsigned char)q;
                    it.f.s0 = (unsigned short)cut[t - b0][(size_t)q]; it.f.s1 = (unsigned short)cut[t - b0][(size_t)q + 1];
                    forder.push_back(it);
                }
            b0 = b1;
        }
    }
    size_t fnext = 0;
    std::multiset<double> events;                       // future times at which the state changes
    // ---- chain state machine
    int ck = 0, cphase = 0;                             // step, 0: potrf pending, 1: crit panel pending, 2: crit update pending
    double chain_free = 0.0, diag_ready = INF, panel_end = 0.0;
    auto tile_done_time = [&](int i, int c) -> double { const Tile& t = T[(size_t)ff_tile(i, c)]; return t.complete() ? t.ready : INF; };
    auto advance_chain = [&]() {
        for (;;) {
            if (ck >= nblk) return;
            if (cphase == 0) {
                if (ck == 0) diag_ready = tile_done_time(0, 0);
                if (diag_ready == INF) return;
                potrf_start[(size_t)ck] = std::max(chain_free + M.boundary, diag_ready);
                if (trace) fprintf(stderr, "[ff] step %2d potrf start %7.1f (chain free %7.1f, diag ready %7.1f)\n", ck, potrf_start[(size_t)ck], chain_free, diag_ready);
                potrf_done[(size_t)ck] = potrf_start[(size_t)ck] + M.potrf;
                events.insert(potrf_start[(size_t)ck]); events.insert(potrf_done[(size_t)ck]);
                chain_free = potrf_done[(size_t)ck];
                if (ck + 1 >= nblk) { out.makespan_us = chain_free; ck = nblk; return; }
                cphase = 1;
            }
            if (cphase == 1) {
                const double tw = tile_done_time(ck + 1, ck);
                if (tw == INF) return;
                panel_end = std::max(chain_free + M.boundary, tw) + M.crit_panel;
                if (trace) fprintf(stderr, "[ff]         panel (%d,%d) input %7.1f chain %7.1f\n", ck + 1, ck, tw, chain_free);
                rowfin[(size_t)ck + 1][(size_t)ck] = panel_end;
                events.insert(panel_end);
                chain_free = panel_end;
                cphase = 2;
            }
            if (cphase == 2) {
                const double tw = tile_done_time(ck + 1, ck + 1);
                if (tw == INF) return;
                if (trace) fprintf(stderr, "[ff]         update (%d,%d) input %7.1f chain %7.1f\n", ck + 1, ck + 1, tw, chain_free);
                diag_ready = std::max(chain_free + M.boundary, tw) + M.crit_update;
                events.insert(diag_ready);
                chain_free = diag_ready;
                ++ck; cphase = 0;
            }
        }
    };
    typedef std::pair<double, int> Ev;                  // (free time, worker)
    std::priority_queue<Ev, std::vector<Ev>, std::greater<Ev>> free_at;
    for (int w = 0; w < W; ++w) free_at.push(Ev(0.0, w));
    out.items.clear();
    size_t remaining = (size_t)ntile;                   // tiles not complete
    int guard = 0;
    while (!free_at.empty() && remaining > 0) {
        const Ev ev = free_at.top(); free_at.pop();
        const double t = ev.first;
        advance_chain();
        for (int r = 0; r < nblk; ++r) while (rf[(size_t)r] < r && rowfin[(size_t)r][(size_t)rf[(size_t)r]] <= t) ++rf[(size_t)r];
        int kc = 0;                                     // chain position: diagonal blocks whose factorization has begun
        while (kc < nblk && potrf_start[(size_t)kc] <= t) ++kc;
        // ---- candidates
        int best = -1, best_class = 99;
        for (int id = 0; id < ntile; ++id) {
            const Tile& x = T[(size_t)id];
            if (x.complete() || x.ready > t) continue;             // done, or an item of the tile is in flight
            const bool fc = x.f_sched == x.qn && x.f_time <= t;
            const int a = std::min(std::min(rf[(size_t)x.i], rf[(size_t)x.c]), x.limit);
            const int pend = a - x.applied;
            const bool can_base = fc && !x.base_in;
            const bool base_ok = x.base_in || can_base;
            const bool panel_ready = x.panel && base_ok && a == x.limit && potrf_done[(size_t)x.c] <= t + 8.0;
            const bool panel_only = panel_ready && x.base_in && pend == 0;
            if (!(pend > 0 || can_base || panel_only)) continue;
            // a tile whose LAST column is all that is missing waits for its diagonal block, so that update and panel solve
            // are one pass over the tile -- while there is formation work to do instead
            if (x.panel && base_ok && a == x.limit && pend <= 1 && !panel_ready && fnext < forder.size()) continue;
            int cls;
            if (x.c <= kc + M.window) cls = 0;
            else if (base_ok && a == x.limit) cls = 1;
            else if (pend >= M.batch) cls = 2;
            else if (can_base && pend == 0 && x.applied == 0 && fnext < forder.size()) continue;   // nothing but the base yet: wait for columns
            else cls = 4;
            if (cls == 4 && fnext < forder.size()) continue;       // small deferred batches only once the formation is exhausted
            if (cls < best_class) { best_class = cls; best = id; }  // ties: lowest tile id = lowest row, then column... see below
            else if (cls == best_class && best >= 0) {
                const Tile& y = T[(size_t)best];
                if (x.c < y.c || (x.c == y.c && x.i < y.i)) best = id;
            }
        }
        if (best < 0 && fnext < forder.size()) {
            // ---- a formation chunk
            const FFItem it = forder[fnext++];
            const double fin = t + M.f_overhead + M.pstage * (it.f.s1 - it.f.s0);
            if (it.c <= it.i) { Tile& x = T[(size_t)ff_tile(it.i, it.c)]; x.f_sched++; x.f_time = std::max(x.f_time, fin); }
            if (it.i + 1 < nblk) { Tile& x = T[(size_t)ff_tile(it.i + 1, it.c)]; x.f_sched++; x.f_time = std::max(x.f_time, fin); }
            out.form_end_us = std::max(out.form_end_us, fin);
            events.insert(fin);
            out.items.push_back(it);
            free_at.push(Ev(fin, ev.second));
            continue;
        }
        if (best < 0) {
            // nothing to do right now: sleep until the state changes (at run time: the next ticket's wait)
            auto nx = events.upper_bound(t);
            if (nx == events.end()) { if (++guard > 4 * W) break; continue; }       // this worker retires
            free_at.push(Ev(*nx, ev.second));
            continue;
        }
        // ---- a T item on tile `best`
        Tile& x = T[(size_t)best];
        const bool fc = x.f_sched == x.qn && x.f_time <= t;
        const int a = std::min(std::min(rf[(size_t)x.i], rf[(size_t)x.c]), x.limit);
        FFItem it{};
        it.type = FF_T; it.i = (unsigned char)x.i; it.c = (unsigned char)x.c;
        it.t.j0 = (unsigned char)x.applied; it.t.j1 = (unsigned char)a;
        if (x.nitems == 0) it.t.flags |= FF_INIT;
        if (fc && !x.base_in) it.t.flags |= FF_ADD_BASE;
        const bool base_after = x.base_in || (it.t.flags & FF_ADD_BASE);
        double dur = M.t_overhead + M.stage * 4.0 * (a - x.applied) + ((it.t.flags & FF_ADD_BASE) ? M.t_base : 0.0);
        double fin = t + dur;
        if (x.panel && base_after && a == x.limit && potrf_done[(size_t)x.c] <= t + dur + 8.0) {
            it.t.flags |= FF_PANEL;
            fin = std::max(fin, potrf_done[(size_t)x.c]) + M.t_panel;
        }
        if (x.i == 0 && x.c == 0) it.t.flags |= FF_SIG_DIAG0;
        x.applied = a; x.base_in = base_after; x.nitems++; x.ready = fin;
        it.t.seq = (unsigned char)x.nitems;
        if (it.t.flags & FF_PANEL) { x.paneled = true; rowfin[(size_t)x.i][(size_t)x.c] = fin; }
        if (x.complete()) --remaining;
        if (trace && getenv("IPM_FF_TRACE")[0] == '2')
            fprintf(stderr, "[ff] t %7.1f T(%d,%d)[%d,%d) flags %d cls %d -> %7.1f\n", t, x.i, x.c, it.t.j0, it.t.j1, it.t.flags, best_class, fin);
        events.insert(fin);
        out.items.push_back(it);
        free_at.push(Ev(fin, ev.second));
    }
    advance_chain();
    out.tile_items.assign((size_t)ntile, 0);
    out.tile_q.assign((size_t)ntile, 0);
    for (int id = 0; id < ntile; ++id) { out.tile_items[(size_t)id] = T[(size_t)id].nitems; out.tile_q[(size_t)id] = T[(size_t)id].qn; }
    if (out.makespan_us == 0.0) out.makespan_us = chain_free;
}

}  // namespace ipm
