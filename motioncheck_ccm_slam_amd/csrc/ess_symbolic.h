// ess_symbolic.h -- host-only symbolic phase of the essential graph's block-sparse Cholesky (see ess_host.cpp, ess_kernels.hip).
// Plain C++ (no HIP): tests/support/ess_symbolic_check.cpp compiles it with g++ and replays the kernels' index walks on the CPU.
#pragma once
#include <algorithm>
#include <cstdint>
#include <iterator>
#include <utility>
#include <vector>

// ---- symbolic phase (host): ordering, fill, and the index lists the kernels walk
// Ordering: rounds of INDEPENDENT low-degree eliminations.  In each round the alive vertices whose degree is within one of the
// minimum (or 20 %) are scanned in (degree, index) order and every one not adjacent to a vertex already taken this round is
// eliminated; its neighbours become a clique.  Vertices of a round do not touch each other, so their columns of the factor
// depend only on earlier rounds: a round is one kernel launch.  A keyframe chain (the spanning tree of an essential graph is
// mostly the agents' trajectories) is halved per round, where strict minimum degree would peel it from its ends one at a time.
struct EssSymbolic {
    int ncol = 0, nnz = 0;
    std::vector<int> perm, iperm;                  // column -> free vertex, free vertex -> column
    std::vector<int> colptr, rowidx;               // strictly-lower blocks per column, rows ascending
    std::vector<int> round_ptr, cols;              // columns grouped by round
    std::vector<int> aptr, alist;                  // per target: edge * 4 + code contributions (assembly)
    std::vector<int> tptr, tpa, tpb;               // per target: factor products (left-looking update)
    std::vector<int> rptr, rslot, rcol;            // per row: its blocks, ascending column
};

static int g_ess_slack_mode = 1;      // candidate rule of a round, see ess_symbolic (test knob: tests/support/ess_symbolic_check.cpp)

static int ess_slot(const EssSymbolic& S, int row, int col)
{
    const int* b = S.rowidx.data() + S.colptr[col]; const int* e = S.rowidx.data() + S.colptr[col + 1];
    const int* p = std::lower_bound(b, e, row);
    return (p != e && *p == row) ? (int)(p - S.rowidx.data()) : -1;
}

static void ess_symbolic(int nf, int ne, const int32_t* ei, const int32_t* ej, const std::vector<int>& fidx, EssSymbolic& S)
{
    std::vector<std::vector<int>> adj(nf);
    for (int k = 0; k < ne; k++) {
        const int a = fidx[ei[k]], b = fidx[ej[k]];
        if (a >= 0 && b >= 0 && a != b) { adj[a].push_back(b); adj[b].push_back(a); }
    }
    for (auto& v : adj) { std::sort(v.begin(), v.end()); v.erase(std::unique(v.begin(), v.end()), v.end()); }
    std::vector<char> alive(nf, 1), blocked(nf, 0);
    std::vector<std::vector<int>> colrows(nf);     // by elimination position, in free-vertex ids for now
    S.perm.clear(); S.iperm.assign(nf, -1); S.round_ptr.assign(1, 0);
    int remaining = nf;
    std::vector<int> cand, merged;
    while (remaining > 0) {
        int dmin = 1 << 30;
        for (int v = 0; v < nf; v++) if (alive[v]) dmin = std::min(dmin, (int)adj[v].size());
        const int lim = g_ess_slack_mode == 0 ? std::max(dmin + 1, dmin + dmin / 5) : g_ess_slack_mode == 1 ? 2 * dmin + 2 : (1 << 30);
        cand.clear();
        for (int v = 0; v < nf; v++) if (alive[v] && (int)adj[v].size() <= lim) cand.push_back(v);
        std::stable_sort(cand.begin(), cand.end(), [&](int a, int b) { return adj[a].size() < adj[b].size(); });
        std::fill(blocked.begin(), blocked.end(), 0);
        std::vector<int> picked;
        for (int v : cand) {
            if (blocked[v]) continue;
            picked.push_back(v); blocked[v] = 1;
            for (int u : adj[v]) blocked[u] = 1;
        }
        for (int v : picked) {
            const std::vector<int> N = adj[v];                       // alive neighbours (invariant: adjacency lists hold alive vertices only)
            const int pos = (int)S.perm.size();
            S.perm.push_back(v); S.iperm[v] = pos; colrows[pos] = N;
            alive[v] = 0; remaining--;
            for (int u : N) {
                merged.clear();
                std::set_union(adj[u].begin(), adj[u].end(), N.begin(), N.end(), std::back_inserter(merged));
                merged.erase(std::remove_if(merged.begin(), merged.end(), [&](int w) { return w == u || w == v; }), merged.end());
                adj[u].swap(merged);
            }
            adj[v].clear();
        }
        S.round_ptr.push_back((int)S.perm.size());
    }
    S.ncol = nf;
    S.colptr.assign(nf + 1, 0);
    for (int c = 0; c < nf; c++) S.colptr[c + 1] = S.colptr[c] + (int)colrows[c].size();
    S.nnz = S.colptr[nf];
    S.rowidx.resize(std::max(S.nnz, 1));
    for (int c = 0; c < nf; c++) {
        int* r = S.rowidx.data() + S.colptr[c];
        for (size_t a = 0; a < colrows[c].size(); a++) r[a] = S.iperm[colrows[c][a]];
        std::sort(r, r + colrows[c].size());
    }
    S.cols.resize(nf);
    for (int c = 0; c < nf; c++) S.cols[c] = c;                        // columns are numbered in elimination order: a round is a contiguous range
    // products of the left-looking update: column k contributes L_ak L_bk^T to target (a, b) for every pair a >= b of its rows
    const int ntargets = nf + S.nnz;
    std::vector<std::vector<std::pair<int, int>>> prod(ntargets);
    for (int k = 0; k < nf; k++) {
        const int c0 = S.colptr[k], c1 = S.colptr[k + 1];
        for (int a = c0; a < c1; a++) {
            const int ra = S.rowidx[a];
            prod[ra].push_back({a, a});                                // diagonal of row ra
            for (int b = c0; b < a; b++) {
                const int t = ess_slot(S, ra, S.rowidx[b]);            // rows ascend: rowidx[b] < ra
                prod[nf + t].push_back({a, b});
            }
        }
    }
    S.tptr.assign(ntargets + 1, 0);
    for (int t = 0; t < ntargets; t++) S.tptr[t + 1] = S.tptr[t] + (int)prod[t].size();
    S.tpa.resize(std::max(S.tptr[ntargets], 1)); S.tpb.resize(std::max(S.tptr[ntargets], 1));
    for (int t = 0; t < ntargets; t++)
        for (size_t q = 0; q < prod[t].size(); q++) { S.tpa[S.tptr[t] + q] = prod[t][q].first; S.tpb[S.tptr[t] + q] = prod[t][q].second; }
    // row lists for the forward solve
    std::vector<std::vector<std::pair<int, int>>> rows(nf);
    for (int k = 0; k < nf; k++)
        for (int a = S.colptr[k]; a < S.colptr[k + 1]; a++) rows[S.rowidx[a]].push_back({a, k});
    S.rptr.assign(nf + 1, 0);
    for (int r = 0; r < nf; r++) S.rptr[r + 1] = S.rptr[r] + (int)rows[r].size();
    S.rslot.resize(std::max(S.rptr[nf], 1)); S.rcol.resize(std::max(S.rptr[nf], 1));
    for (int r = 0; r < nf; r++)
        for (size_t q = 0; q < rows[r].size(); q++) { S.rslot[S.rptr[r] + q] = rows[r][q].first; S.rcol[S.rptr[r] + q] = rows[r][q].second; }
    // assembly lists, in edge order: diagonal targets take Ji^T Ji / Jj^T Jj of every incident edge, the lower block of a pair of
    // free keyframes takes Ji^T Jj (or its transpose, whichever lands below the diagonal in the permuted order)
    std::vector<std::vector<int>> asm_l(ntargets);
    for (int k = 0; k < ne; k++) {
        const int a = fidx[ei[k]], b = fidx[ej[k]];
        if (a >= 0) asm_l[S.iperm[a]].push_back(4 * k + 0);
        if (b >= 0) asm_l[S.iperm[b]].push_back(4 * k + 1);
        if (a >= 0 && b >= 0 && a != b) {
            const int pa = S.iperm[a], pb = S.iperm[b];
            if (pa > pb) asm_l[nf + ess_slot(S, pa, pb)].push_back(4 * k + 2);       // block (i, j) = Ji^T Jj
            else asm_l[nf + ess_slot(S, pb, pa)].push_back(4 * k + 3);               // block (j, i) = (Ji^T Jj)^T
        }
    }
    S.aptr.assign(ntargets + 1, 0);
    for (int t = 0; t < ntargets; t++) S.aptr[t + 1] = S.aptr[t] + (int)asm_l[t].size();
    S.alist.resize(std::max(S.aptr[ntargets], 1));
    for (int t = 0; t < ntargets; t++) std::copy(asm_l[t].begin(), asm_l[t].end(), S.alist.begin() + S.aptr[t]);
}

