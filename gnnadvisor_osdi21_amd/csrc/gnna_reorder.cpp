// gnna_reorder.cpp -- locality renumbering of the nodes of a graph (host, multi-threaded).
//
// Role of the reference's rabbit.reorder (rabbit_module/src/reorder.cpp:235-295, a wrapper around the
// third-party Rabbit Order): given the edge list, return a relabelling that puts nodes with common
// neighbours close together, so that the rows a destination gathers sit in a small moving window of the
// feature matrix (cache hits on one GPU, small halos between GPUs).  Rabbit Order does that by
// incremental community aggregation with a lock-free merge protocol (needs boost / numa / tcmalloc, is
// not reproducible under OpenMP).  This is a different, deterministic algorithm with the same contract:
//
//   1. communities: size-capped label propagation on the symmetrised graph.  Every sweep computes, for
//      all nodes in parallel, the label that carries most of the node's edges (ties -> smallest label);
//      sweeps alternate between "may only move to a smaller label" and "to a larger one", which rules
//      out the two-node label swaps of synchronous propagation; moves are then applied in node order
//      against the size cap (the only sequential O(N) part).
//   2. order of the communities: the community graph (summed edge weights) is walked greedily -- start
//      at the heaviest community, always append the unplaced community most strongly tied to the tail
//      of the chain (falling back to the most strongly tied to anything placed) -- which lays adjacent
//      regions of a spatial / band-like graph next to each other.
//   3. order inside and across the community borders: a few barycentre sweeps (position <- mean
//      position of the neighbours, parallel Jacobi), started from the community ranks, pull every node
//      to the middle of its own neighbourhood; the final id is the rank of the refined position.
//
// Threads: std::thread over contiguous node ranges, results independent of the thread count.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <numeric>
#include <queue>
#include <thread>
#include <utility>
#include <vector>

#include "gnna.h"
#include "gnna_internal.h"

namespace {

template <typename F>
void parallel_nodes(int64_t n, int threads, F &&fn)
{
    const int64_t nt = std::max<int64_t>(1, std::min<int64_t>(threads, n / 2048 + 1));
    if (nt == 1) { fn(0, n); return; }
    std::vector<std::thread> th;
    const int64_t step = (n + nt - 1) / nt;
    for (int64_t t = 0; t < nt; t++) {
        const int64_t lo = t * step, hi = std::min(n, lo + step);
        if (lo >= hi) break;
        th.emplace_back([&fn, lo, hi] { fn(lo, hi); });
    }
    for (auto &t : th) t.join();
}

int host_threads()
{
    unsigned hw = std::thread::hardware_concurrency();
    return (int)std::max(1u, std::min(hw ? hw : 1u, 64u));
}

}  // namespace

extern "C" {
#pragma GCC visibility push(default)

int gnna_reorder_community_i32(const int32_t *src, const int32_t *dst, int64_t num_edges, int64_t num_nodes,
                               int32_t *new_id)
{
    using gnna::fail;
    if (num_edges < 0 || num_nodes < 0 || (num_nodes > 0 && !new_id) || (num_edges > 0 && (!src || !dst)))
        return fail(GNNA_ERR_INVALID_ARGUMENT, "bad reorder arguments");
    if (num_nodes > 0x7fffffffLL || 2 * num_edges > 0x7fffffffLL)
        return fail(GNNA_ERR_UNSUPPORTED, "graph too large for int32 reorder");
    const int64_t n = num_nodes;
    if (n == 0) return GNNA_OK;
    const int threads = host_threads();

    // symmetrised, duplicate-free adjacency (reorder.cpp:31-97 does the same before aggregating)
    std::vector<int32_t> rp((size_t)n + 1), ci;
    {
        std::vector<int32_t> s2((size_t)2 * num_edges), d2((size_t)2 * num_edges);
        std::copy_n(src, num_edges, s2.begin()); std::copy_n(dst, num_edges, s2.begin() + num_edges);
        std::copy_n(dst, num_edges, d2.begin()); std::copy_n(src, num_edges, d2.begin() + num_edges);
        ci.resize((size_t)2 * num_edges);
        const int64_t nnz = gnna_csr_from_edges_i32(s2.data(), d2.data(), 2 * num_edges, n, rp.data(), ci.data());
        if (nnz < 0) return (int)nnz;
        ci.resize((size_t)nnz);
    }

    // ---- 1. size-capped label propagation --------------------------------------------------------------
    // cap: communities of at most ~N/64 nodes (and at least 256), i.e. the chain of step 2 has >= 64 links
    const int64_t cap = std::max<int64_t>(256, n / 64);
    std::vector<int32_t> label((size_t)n), want((size_t)n), csize((size_t)n, 1);
    std::iota(label.begin(), label.end(), 0);
    for (int sweep = 0; sweep < 10; sweep++) {
        const bool to_smaller = (sweep % 2) == 0;
        parallel_nodes(n, threads, [&](int64_t lo, int64_t hi) {
            std::vector<int32_t> nb;
            for (int64_t v = lo; v < hi; v++) {
                const int32_t b = rp[(size_t)v], e = rp[(size_t)v + 1];
                want[(size_t)v] = label[(size_t)v];
                if (e == b) continue;
                nb.resize((size_t)(e - b));
                for (int32_t k = b; k < e; k++) nb[(size_t)(k - b)] = label[(size_t)ci[(size_t)k]];
                std::sort(nb.begin(), nb.end());
                // most frequent neighbour label; ties -> smallest label; the node's own label wins ties with it
                const int32_t own = label[(size_t)v];
                int32_t best = own, best_cnt = 0, own_cnt = 0;
                for (size_t i = 0; i < nb.size();) {
                    size_t j = i;
                    while (j < nb.size() && nb[j] == nb[i]) j++;
                    const int32_t cnt = (int32_t)(j - i);
                    if (nb[i] == own) own_cnt = cnt;
                    if (cnt > best_cnt) { best_cnt = cnt; best = nb[i]; }
                    i = j;
                }
                if (best != own && best_cnt > own_cnt && (to_smaller ? best < own : best > own)) want[(size_t)v] = best;
            }
        });
        int64_t moved = 0;
        for (int64_t v = 0; v < n; v++) {
            const int32_t w = want[(size_t)v], l = label[(size_t)v];
            if (w != l && csize[(size_t)w] < cap) {
                csize[(size_t)w]++; csize[(size_t)l]--;
                label[(size_t)v] = w;
                moved++;
            }
        }
        if (moved * 200 < n && sweep >= 3) break;   // < 0.5 % of the nodes still moving
    }

    // compact community ids
    std::vector<int32_t> comm_of((size_t)n, -1);
    int32_t nc = 0;
    for (int64_t v = 0; v < n; v++) {
        int32_t &c = comm_of[(size_t)label[(size_t)v]];
        if (c < 0) c = nc++;
    }
    std::vector<int32_t> comm((size_t)n);
    std::vector<int64_t> cnodes((size_t)nc, 0);
    for (int64_t v = 0; v < n; v++) { comm[(size_t)v] = comm_of[(size_t)label[(size_t)v]]; cnodes[(size_t)comm[(size_t)v]]++; }

    // ---- 2. chain of communities ------------------------------------------------------------------------
    // community graph as sorted (a, b) -> weight triples, built per thread and merged
    std::vector<std::pair<int64_t, int64_t>> cedges;   // (a * nc + b, weight), a != b
    {
        std::vector<std::vector<int64_t>> keys((size_t)threads);
        const int64_t step = (n + threads - 1) / threads;
        std::vector<std::thread> th;
        for (int t = 0; t < threads; t++) {
            th.emplace_back([&, t] {
                auto &out = keys[(size_t)t];
                for (int64_t v = t * step; v < std::min(n, (t + 1) * step); v++) {
                    const int64_t a = comm[(size_t)v];
                    for (int32_t k = rp[(size_t)v]; k < rp[(size_t)v + 1]; k++) {
                        const int64_t b = comm[(size_t)ci[(size_t)k]];
                        if (a != b) out.push_back(a * nc + b);
                    }
                }
                std::sort(out.begin(), out.end());
            });
        }
        for (auto &t : th) t.join();
        std::vector<int64_t> all;
        size_t total = 0;
        for (auto &k : keys) total += k.size();
        all.reserve(total);
        for (auto &k : keys) { all.insert(all.end(), k.begin(), k.end()); std::vector<int64_t>().swap(k); }
        std::sort(all.begin(), all.end());
        for (size_t i = 0; i < all.size();) {
            size_t j = i;
            while (j < all.size() && all[j] == all[i]) j++;
            cedges.emplace_back(all[i], (int64_t)(j - i));
            i = j;
        }
    }
    std::vector<int64_t> cstart((size_t)nc + 1, 0);
    for (auto &e : cedges) cstart[(size_t)(e.first / nc) + 1]++;
    for (int32_t c = 0; c < nc; c++) cstart[(size_t)c + 1] += cstart[(size_t)c];
    std::vector<int32_t> order_c;
    order_c.reserve((size_t)nc);
    {
        std::vector<char> placed((size_t)nc, 0);
        std::vector<double> tie((size_t)nc, 0.0);   // weight between an unplaced community and the placed set
        std::priority_queue<std::pair<double, int32_t>> heap;   // (tie, community), stale entries skipped
        auto place = [&](int32_t c) {
            placed[(size_t)c] = 1;
            order_c.push_back(c);
            for (int64_t k = cstart[(size_t)c]; k < cstart[(size_t)c + 1]; k++) {
                const int32_t b = (int32_t)(cedges[(size_t)k].first % nc);
                if (placed[(size_t)b]) continue;
                tie[(size_t)b] += (double)cedges[(size_t)k].second;
                heap.emplace(tie[(size_t)b], b);
            }
        };
        int32_t seed = 0;
        for (int32_t c = 1; c < nc; c++)
            if (cnodes[(size_t)c] > cnodes[(size_t)seed]) seed = c;
        place(seed);
        int32_t scan = 0;   // next candidate among communities tied to nothing placed (isolated ones: id order)
        while ((int32_t)order_c.size() < nc) {
            const int32_t tail = order_c.back();
            int32_t next = -1;
            double best = 0.0;
            for (int64_t k = cstart[(size_t)tail]; k < cstart[(size_t)tail + 1]; k++) {   // strongest tie to the tail
                const int32_t b = (int32_t)(cedges[(size_t)k].first % nc);
                const double w = (double)cedges[(size_t)k].second / (double)cnodes[(size_t)b];
                if (!placed[(size_t)b] && (w > best || (w == best && next >= 0 && b < next))) { best = w; next = b; }
            }
            while (next < 0 && !heap.empty()) {                                          // else: to anything placed
                const auto top = heap.top();
                heap.pop();
                if (!placed[(size_t)top.second] && top.first == tie[(size_t)top.second]) next = top.second;
            }
            if (next < 0) {
                while (placed[(size_t)scan]) scan++;
                next = scan;
            }
            place(next);
        }
    }

    // ---- 3. barycentre refinement ----------------------------------------------------------------------------
    std::vector<double> pos((size_t)n), nxt((size_t)n);
    {
        std::vector<int64_t> cbase((size_t)nc, 0);
        int64_t at = 0;
        for (int32_t c : order_c) { cbase[(size_t)c] = at; at += cnodes[(size_t)c]; }
        std::vector<int64_t> fill((size_t)nc, 0);
        for (int64_t v = 0; v < n; v++) {
            const int32_t c = comm[(size_t)v];
            pos[(size_t)v] = (double)(cbase[(size_t)c] + fill[(size_t)c]++);
        }
    }
    std::vector<int32_t> perm((size_t)n);
    for (int it = 0; it < 8; it++) {
        parallel_nodes(n, threads, [&](int64_t lo, int64_t hi) {
            for (int64_t v = lo; v < hi; v++) {
                const int32_t b = rp[(size_t)v], e = rp[(size_t)v + 1];
                if (e == b) { nxt[(size_t)v] = pos[(size_t)v]; continue; }
                double s = 0.0;
                for (int32_t k = b; k < e; k++) s += pos[(size_t)ci[(size_t)k]];
                nxt[(size_t)v] = 0.5 * pos[(size_t)v] + 0.5 * s / (double)(e - b);
            }
        });
        // re-spread to ranks so that the arrangement does not contract (ties: by node id)
        std::iota(perm.begin(), perm.end(), 0);
        std::sort(perm.begin(), perm.end(), [&](int32_t a, int32_t b2) {
            return nxt[(size_t)a] < nxt[(size_t)b2] || (nxt[(size_t)a] == nxt[(size_t)b2] && a < b2);
        });
        for (int64_t r = 0; r < n; r++) pos[(size_t)perm[(size_t)r]] = (double)r;
    }
    for (int64_t v = 0; v < n; v++) new_id[(size_t)v] = (int32_t)pos[(size_t)v];
    return GNNA_OK;
}

#pragma GCC visibility pop
}  // extern "C"
