// lockstep_merge.h -- the schedule of a lockstep batch (lockstep.h): the launch programs of the LPs of a batch merged into GLOBAL
// STEPS of one kernel type each, every LP's own order preserved.  Host code only (no HIP): tested on the CPU through
// ipm_debug_ls_merge (tests/test_lockstep_merge.py).
//
// The number of steps is the length of the batch's chain of dependent launches, so this is a shortest-common-supersequence
// problem over the programs' type sequences.  The programs follow one template (residuals, formation, nb x {potrf, panel, update},
// substitutions, vector kernels; reference loop main.py:780-807) with different block counts and a few variant sections, so a
// progressive pairwise alignment is close to the longest program:
//   S = the longest program's types;  for every other program P, longest first:  S = SCS(S, P)  by the textbook O(|S||P|) table;
//   then every program takes the EARLIEST embedding into S (an LP is done with its iteration as early as possible) and the steps
//   nobody uses are dropped.
// The first version of the batch let the LP with the most launches left pick the type of the next step (ls_merge_leader below):
// two LPs of similar length that are out of phase then take turns as the leader and the schedule is the SUM of their programs
// (53 LPs up to 2200 rows: 190 steps for a longest program of 94).
#pragma once
#include <stdint.h>

#include <algorithm>
#include <numeric>
#include <utility>
#include <vector>

namespace ipm {

struct LsPlanStep {
    int type;
    std::vector<std::pair<int, int>> members;      // (program, position in that program), at most max_group of them
};

inline void ls_split_groups(std::vector<LsPlanStep>& plan, int max_group) {
    std::vector<LsPlanStep> out;
    for (LsPlanStep& st : plan) {
        if (st.members.empty()) continue;
        for (size_t o = 0; o < st.members.size(); o += (size_t)max_group) {
            LsPlanStep part;
            part.type = st.type;
            part.members.assign(st.members.begin() + (long)o, st.members.begin() + (long)std::min(st.members.size(), o + (size_t)max_group));
            out.push_back(std::move(part));
        }
    }
    plan.swap(out);
}

// the first version (kept for the A/B in tests and IPM_LS_MERGE=leader): the LP with the most launches left sets the next type
inline void ls_merge_leader(const std::vector<std::vector<int>>& types, int max_group, std::vector<LsPlanStep>& plan) {
    plan.clear();
    const size_t n = types.size();
    std::vector<size_t> pos(n, 0);
    for (;;) {
        size_t lead = n, left = 0;
        for (size_t i = 0; i < n; ++i) { const size_t l = types[i].size() - pos[i]; if (l > left) { left = l; lead = i; } }
        if (lead == n) break;
        LsPlanStep st;
        st.type = types[lead][pos[lead]];
        for (size_t i = 0; i < n; ++i)
            if (pos[i] < types[i].size() && types[i][pos[i]] == st.type) { st.members.emplace_back((int)i, (int)pos[i]); ++pos[i]; }
        plan.push_back(std::move(st));
    }
    ls_split_groups(plan, max_group);
}

// shortest common supersequence of S and P (both kept in order); ties keep S's element first
inline std::vector<int> ls_scs(const std::vector<int>& S, const std::vector<int>& P) {
    const size_t a = S.size(), b = P.size(), W = b + 1;
    std::vector<uint32_t> dp((a + 1) * W);
    for (size_t j = 0; j <= b; ++j) dp[a * W + j] = (uint32_t)(b - j);
    for (size_t i = a; i-- > 0;) {
        dp[i * W + b] = (uint32_t)(a - i);
        for (size_t j = b; j-- > 0;)
            dp[i * W + j] = 1u + (S[i] == P[j] ? dp[(i + 1) * W + j + 1] : std::min(dp[(i + 1) * W + j], dp[i * W + j + 1]));
    }
    std::vector<int> out;
    out.reserve(dp[0]);
    size_t i = 0, j = 0;
    while (i < a && j < b) {
        if (S[i] == P[j]) { out.push_back(S[i]); ++i; ++j; }
        else if (dp[(i + 1) * W + j] <= dp[i * W + j + 1]) out.push_back(S[i++]);
        else out.push_back(P[j++]);
    }
    while (i < a) out.push_back(S[i++]);
    while (j < b) out.push_back(P[j++]);
    return out;
}

inline void ls_merge_aligned(const std::vector<std::vector<int>>& types, int max_group, std::vector<LsPlanStep>& plan) {
    plan.clear();
    const size_t n = types.size();
    if (n == 0) return;
    std::vector<size_t> order(n);
    std::iota(order.begin(), order.end(), (size_t)0);
    std::stable_sort(order.begin(), order.end(), [&](size_t x, size_t y) { return types[x].size() > types[y].size(); });
    std::vector<int> S = types[order[0]];
    for (size_t k = 1; k < n; ++k) {
        const std::vector<int>& P = types[order[k]];
        // (a program that is already a subsequence of S -- the common case: the same template with fewer blocks -- changes nothing)
        size_t j = 0;
        for (size_t i = 0; i < S.size() && j < P.size(); ++i) if (S[i] == P[j]) ++j;
        if (j < P.size()) S = ls_scs(S, P);
    }
    plan.resize(S.size());
    for (size_t s = 0; s < S.size(); ++s) plan[s].type = S[s];
    for (size_t i = 0; i < n; ++i) {               // earliest embedding, LPs in their batch order inside a step
        size_t j = 0;
        for (size_t s = 0; s < S.size() && j < types[i].size(); ++s)
            if (S[s] == types[i][j]) { plan[s].members.emplace_back((int)i, (int)j); ++j; }
    }
    ls_split_groups(plan, max_group);              // (drops the steps nobody uses, too)
}

}  // namespace ipm
