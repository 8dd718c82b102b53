// gnna_reorder.cpp -- locality renumbering of the nodes of a graph (host, multi-threaded).
//
// Role of the reference's rabbit.reorder (rabbit_module/src/reorder.cpp:235-295, a wrapper around the
// third-party Rabbit Order): given the edge list, return a relabelling that puts nodes with common
// neighbours close together, so that the rows a destination gathers sit in a small moving window of the
// feature matrix (cache hits on one GPU, small halos between GPUs).  Rabbit Order does that by
// incremental community aggregation with a lock-free merge protocol (needs boost / numa / tcmalloc, is
// not reproducible under OpenMP).  This is a different, deterministic algorithm with the same contract:
//
//   1. backbone: an edge (u, v) is kept when u and v have at least T common neighbours (merge of the two
//      sorted adjacency lists, all edges in parallel, early exit at T).  Edges inside a community / a
//      spatial neighbourhood are embedded in many triangles, the long-range edges that make a graph a
//      small world -- and defeat any breadth-first ordering -- are in none.  Hubs (degree > 16 x average)
//      are left out of the backbone; they are adjacent to everything.
//   2. coarse order: breadth-first discovery order over the backbone, component by component (largest
//      first), started at a peripheral node of the component (the last node of a first sweep); a walk that
//      advances on two fronts (a ring of neighbourhoods) is unfolded into arm - core - arm.
//   3. fine order: median sweeps over the backbone until the arrangement stops settling (4 .. 16) (position <- median position of the neighbours,
//      parallel Jacobi, re-spread to ranks after every sweep) pull every node to the middle of its own
//      neighbourhood without being dragged by a stray long-range neighbour, one closing pass by the mode (the shortest
//      interval holding half of the neighbours) for nodes whose neighbours are spread over many communities; nodes without backbone edges
//      are then put at the median position of all their neighbours.  The final id is the rank of the refined position.
//
// Threads: std::thread over contiguous node ranges; the result does not depend on the thread count.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <numeric>
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

// a few coarse tasks (chunks of a sort, pairs of a merge round), one thread each up to `threads`
template <typename F>
void parallel_tasks(int64_t count, int threads, F &&fn)
{
    const int64_t nt = std::max<int64_t>(1, std::min<int64_t>(threads, count));
    if (nt == 1) { for (int64_t i = 0; i < count; i++) fn(i); return; }
    std::vector<std::thread> th;
    for (int64_t t = 0; t < nt; t++)
        th.emplace_back([&fn, t, nt, count] { for (int64_t i = t; i < count; i += nt) fn(i); });
    for (auto &t : th) t.join();
}

int host_threads() { return gnna::host_thread_budget(64); }      // (the CPUs the container is granted, not the ones it sees)

// A symmetric, duplicate-free adjacency with sorted rows, wherever it lives (built here from an edge list, or the caller's CSR)
struct IdList {
    const int32_t *p = nullptr;
    size_t len = 0;
    size_t size() const { return len; }
    const int32_t *data() const { return p; }
    int32_t operator[](size_t i) const { return p[i]; }
};

struct Lap {
    bool on;
    std::chrono::steady_clock::time_point t;
    explicit Lap(bool debug) : on(debug), t(std::chrono::steady_clock::now()) {}
    void operator()(const char *what)
    {
        if (!on) return;
        auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[reorder] %-28s %.2f s\n", what, std::chrono::duration<double>(now - t).count());
        t = now;
    }
};

// How many ids of list[0 .. len) are marked in the bitmap -- counted up to `need` (the caller only asks "at least need?", so
// any count >= need is as good as another): eight probes between two looks at the exit condition.  The backbone stage is this
// loop (4e10 probes on the Reddit-like graph: the end points of an edge of a power-law graph have ~2,000 neighbours each).
// Round 6 tried two things against it on the GPU box's EPYC 9575F, both recorded in profiles/r6/reorder_stages.md and neither
// kept: prefetching the lists ahead (2.11 -> 2.29 s: the scan is not waiting for memory) and an AVX-512 form -- sixteen ids per
// step through a gather of the bitmap words and a mask popcount -- (1.67 s against 1.53 s for this loop: Zen 5's gather is
// no faster than sixteen scalar probes of an L1-resident bitmap).  What did pay: handing the nodes out in blocks (below).
inline int32_t count_marked(const int32_t *list, int64_t len, const uint64_t *mk, int32_t need)
{
    int32_t common = 0;
    int64_t j = 0;
    for (; j + 8 <= len && common < need; j += 8) {
        int32_t c8 = 0;
        for (int q = 0; q < 8; q++) {
            const uint32_t x = (uint32_t)list[j + q];
            c8 += (int32_t)((mk[x >> 6] >> (x & 63u)) & 1ull);
        }
        common += c8;
    }
    for (; j < len && common < need; j++) {
        const uint32_t x = (uint32_t)list[j];
        common += (int32_t)((mk[x >> 6] >> (x & 63u)) & 1ull);
    }
    return common;
}

// Steps 1-3 of the header comment on the adjacency (rp, ci) of n nodes -> new_id[old id].
int community_order(const int64_t n, const std::vector<int64_t> &rp, const IdList ci, int32_t *new_id, const int threads,
                    const bool debug, Lap &lap)
{
    // ---- 1. backbone: edges with >= T common neighbours ------------------------------------------------------
    const double avg_deg = (double)ci.size() / (double)n;
    const int T = std::getenv("GNNA_REORDER_SUPPORT") ? std::atoi(std::getenv("GNNA_REORDER_SUPPORT"))
                                                       : (avg_deg < 32 ? 1 : (avg_deg < 128 ? 2 : 3));
    const int64_t hub = (int64_t)std::max(64.0, 16.0 * avg_deg);
    // two unrelated nodes of degrees du, dv still share about du * dv * sum_x d_x^2 / (2m)^2 neighbours (the
    // popular nodes); an edge has to beat that expectation clearly to count as embedded
    double sum_d2 = 0.0;
    for (int64_t v = 0; v < n; v++) {
        const double d = (double)(rp[(size_t)v + 1] - rp[(size_t)v]);
        if (d <= (double)hub) sum_d2 += d * d;
    }
    const double chance = sum_d2 / ((double)ci.size() * (double)ci.size());
    std::vector<int64_t> brp((size_t)n + 1, 0);
    std::vector<int32_t> bci;
    {
        std::vector<uint8_t> keep(ci.size(), 0);
        // Every undirected pair once, from the side of its end point u of HIGHER degree (ties: higher id): u's neighbours are
        // marked in a per-thread bitmap (n bits: cache resident), the SHORTER list -- v's -- is then a sequential scan that
        // stops at `need` marked entries, and the answer is written to both directions (the reverse entry by binary search in
        // v's sorted list).  Same result as merging the two sorted lists per directed edge -- the support and `need` are
        // symmetric -- at a fraction of the time on graphs with rows of hundreds of edges (round 5: the merge made the
        // Reddit-like graph's renumbering the longest leg of bench.py; scanning the shorter list of a pair instead of the
        // list of the higher id halves what is left: a pair that is NOT kept costs min(du, dv) instead of dv).
        // (nodes are handed out in blocks of 512 from a shared counter: the work of a node is the sum over its neighbours of the
        // shorter list of the pair -- far from uniform along the id range of a locality-ordered or degree-sorted graph; the flags
        // written do not depend on who counts a pair; Reddit-like on 16 threads: 2.2 -> 1.55 s)
        std::atomic<int64_t> next_block{0};
        constexpr int64_t kNodeBlock = 512;
        parallel_nodes(n, threads, [&](int64_t, int64_t) {
            std::vector<uint64_t> mark((size_t)((n + 63) / 64) + 1, 0);
            for (;;) {
            const int64_t lo = next_block.fetch_add(kNodeBlock), hi = std::min(n, lo + kNodeBlock);
            if (lo >= n) break;
            for (int64_t u = lo; u < hi; u++) {
                const int64_t ub = rp[(size_t)u], ue = rp[(size_t)u + 1];
                if (ue - ub > hub) continue;
                for (int64_t k = ub; k < ue; k++) mark[(size_t)ci[(size_t)k] >> 6] |= 1ull << (ci[(size_t)k] & 63);
                for (int64_t k = ub; k < ue; k++) {
                    const int32_t v = ci[(size_t)k];
                    const int64_t vb = rp[(size_t)v], ve = rp[(size_t)v + 1];
                    if (ve - vb > ue - ub || (ve - vb == ue - ub && v >= u)) continue;     // (the pair is v's to count)
                    const int32_t need = std::max<int32_t>(T, (int32_t)std::ceil(3.0 * chance * (double)(ue - ub) * (double)(ve - vb)));
                    const int32_t common = count_marked(ci.data() + vb, ve - vb, mark.data(), need);
                    if (common >= need) {
                        keep[(size_t)k] = 1;
                        const int32_t *vl = ci.data() + vb, *vend = ci.data() + ve;
                        const int64_t back = std::lower_bound(vl, vend, (int32_t)u) - vl;    // (u is in v's list: the adjacency is symmetric)
                        if (back < ve - vb && vl[back] == (int32_t)u) keep[(size_t)(vb + back)] = 1;
                    }
                }
                for (int64_t k = ub; k < ue; k++) mark[(size_t)ci[(size_t)k] >> 6] = 0;
            }
            }
        });
        parallel_nodes(n, threads, [&](int64_t lo, int64_t hi) {
            for (int64_t u = lo; u < hi; u++) {
                int64_t c = 0;
                for (int64_t k = rp[(size_t)u]; k < rp[(size_t)u + 1]; k++) c += keep[(size_t)k];
                brp[(size_t)u + 1] = c;
            }
        });
        for (int64_t u = 0; u < n; u++) brp[(size_t)u + 1] += brp[(size_t)u];
        bci.resize((size_t)brp[(size_t)n]);
        parallel_nodes(n, threads, [&](int64_t lo, int64_t hi) {
            for (int64_t u = lo; u < hi; u++) {
                int64_t at = brp[(size_t)u];
                for (int64_t k = rp[(size_t)u]; k < rp[(size_t)u + 1]; k++)
                    if (keep[(size_t)k]) bci[(size_t)at++] = ci[(size_t)k];
            }
        });
    }
    if (std::getenv("GNNA_REORDER_DEBUG"))
        std::fprintf(stderr, "[reorder] %lld nodes, %zu adjacency entries, backbone (>= %d common neighbours) keeps %zu\n",
                     (long long)n, ci.size(), T, bci.size());

    lap("backbone");
    // ---- 2. breadth-first discovery order over the backbone ---------------------------------------------------
    std::vector<double> pos((size_t)n, -1.0), nxt((size_t)n);
    std::vector<char> in_backbone((size_t)n, 0);
    {
        std::vector<int32_t> comp((size_t)n, -1), queue;
        queue.reserve((size_t)n);
        // A node is discovered once `theta` of its backbone neighbours have been walked (the seed's whole
        // neighbourhood starts the walk): the front advances through a neighbourhood only when it is adjacent to
        // it in many places, so the odd long-range edge that survived step 1 does not open a second front far away.
        const int theta = avg_deg < 32 ? 1 : 3;
        // walk tag and hit counter of a node are separate words: packed into one int32 (tag * 65536 + count) the tag
        // overflowed from 32768 walks on -- common once unsupported edges and hubs are dropped on a large graph
        std::vector<int32_t> hit_tag((size_t)n, -1), hits((size_t)n, 0);
        std::vector<int32_t> depth((size_t)n, 0);
        // `only`: the walk may take nodes whose component tag is `only` and nothing else -- -1 (unassigned) in the first
        // sweep, the component's own first-sweep tag in the second.  (Without the restriction a walk re-absorbed the
        // nodes of components walked before it: with thousands of small components next to a giant one -- a graph of
        // average degree ~50 -- the second sweep walked the giant component once per component, 65 s for 1e5 nodes.)
        auto bfs = [&](int32_t seed, int32_t tag, int32_t only, std::vector<int32_t> &out) {   // nodes of comp `tag` in discovery order
            out.clear();
            out.push_back(seed);
            comp[(size_t)seed] = tag;
            depth[(size_t)seed] = 0;
            for (int64_t k = brp[(size_t)seed]; k < brp[(size_t)seed + 1]; k++) {
                const int32_t u = bci[(size_t)k];
                if (comp[(size_t)u] == only) { comp[(size_t)u] = tag; depth[(size_t)u] = 0; out.push_back(u); }
            }
            for (size_t head = 0; head < out.size(); head++) {
                const int32_t v = out[head];
                for (int64_t k = brp[(size_t)v]; k < brp[(size_t)v + 1]; k++) {
                    const int32_t u = bci[(size_t)k];
                    if (comp[(size_t)u] != only) continue;
                    if (hit_tag[(size_t)u] != tag) { hit_tag[(size_t)u] = tag; hits[(size_t)u] = 0; }   // counter of this walk
                    if (++hits[(size_t)u] >= theta) {
                        comp[(size_t)u] = tag;
                        depth[(size_t)u] = depth[(size_t)v] + 1;
                        out.push_back(u);
                    }
                }
            }
        };
        // A walk that starts inside a ring (or in the middle of a long strip) advances on two fronts, and its
        // discovery order interleaves the two arms -- the ring comes out folded.  If, a few levels in, the level
        // set falls into two backbone-connected parts of comparable size, the nodes beyond are told apart by
        // which part they descend from and the order becomes: arm B reversed, the core, arm A.
        // (round 6: the levels are examined side by side -- each level's connected parts touch only that level's nodes, so the
        // threads share `arm` -- and the FIRST level that splits decides, as in the one-thread walk through the levels)
        std::vector<int32_t> arm((size_t)n, 0);
        auto unfold = [&](std::vector<int32_t> &w) {
            if (w.size() < 4096) return;
            const int32_t maxd = depth[(size_t)w.back()];
            const int Lmax = std::min(maxd / 2, 16);
            if (Lmax < 1) return;
            std::vector<size_t> level_at((size_t)Lmax + 2, w.size());          // level L = w[level_at[L] .. level_at[L + 1])
            {
                size_t at = 0;
                for (int L = 1; L <= Lmax + 1; L++) {
                    while (at < w.size() && depth[(size_t)w[at]] < L) at++;
                    level_at[(size_t)L] = at;
                }
            }
            struct Verdict { bool split = false; int32_t first = 0, second = 0; };
            std::vector<Verdict> verdict((size_t)Lmax + 1);
            parallel_tasks(Lmax, threads, [&](int64_t t) {
                const int32_t L = (int32_t)t + 1;
                const size_t lo = level_at[(size_t)L], hi = level_at[(size_t)L + 1];
                if (hi - lo < 64) return;
                // connected parts of level L (arm = -1 - part while exploring)
                for (size_t i = lo; i < hi; i++) arm[(size_t)w[i]] = -1;
                std::vector<std::pair<int64_t, int32_t>> parts;   // (size, id)
                std::vector<int32_t> stack;
                int32_t np = 0;
                for (size_t i = lo; i < hi; i++) {
                    if (arm[(size_t)w[i]] != -1) continue;
                    const int32_t id = -2 - np;
                    int64_t sz = 0;
                    stack.assign(1, w[i]);
                    arm[(size_t)w[i]] = id;
                    while (!stack.empty()) {
                        const int32_t v = stack.back();
                        stack.pop_back();
                        sz++;
                        for (int64_t k = brp[(size_t)v]; k < brp[(size_t)v + 1]; k++) {
                            const int32_t u = bci[(size_t)k];
                            if (depth[(size_t)u] == L && arm[(size_t)u] == -1) { arm[(size_t)u] = id; stack.push_back(u); }   // (depth first: `arm` of another level's node is another thread's)
                        }
                    }
                    parts.emplace_back(sz, id);
                    np++;
                }
                std::sort(parts.rbegin(), parts.rend());
                const bool split = parts.size() >= 2 && 4 * parts[1].first >= parts[0].first &&
                                   5 * (parts[0].first + parts[1].first) >= 4 * (int64_t)(hi - lo);
                if (!split) {
                    for (size_t i = lo; i < hi; i++) arm[(size_t)w[i]] = 0;
                    return;
                }
                verdict[(size_t)L].split = true;
                verdict[(size_t)L].first = parts[0].second;
                verdict[(size_t)L].second = parts[1].second;
            });
            for (int32_t L = 1; L <= Lmax; L++) {
                if (!verdict[(size_t)L].split) continue;
                const size_t lo = level_at[(size_t)L], hi = level_at[(size_t)L + 1];
                // arms: 1 = descends from the largest part, 2 = from the second; earlier levels are the core (0)
                for (size_t i = 0; i < lo; i++) arm[(size_t)w[i]] = 0;
                for (size_t i = lo; i < w.size(); i++) {
                    const int32_t v = w[i];
                    int32_t a = 0;
                    if (i < hi) a = arm[(size_t)v] == verdict[(size_t)L].first ? 1 : (arm[(size_t)v] == verdict[(size_t)L].second ? 2 : 0);
                    if (a == 0) {
                        int64_t c1 = 0, c2 = 0;
                        for (int64_t k = brp[(size_t)v]; k < brp[(size_t)v + 1]; k++) {
                            const int32_t u = bci[(size_t)k];
                            if (depth[(size_t)u] <= depth[(size_t)v] && comp[(size_t)u] == comp[(size_t)v]) {
                                c1 += arm[(size_t)u] == 1;
                                c2 += arm[(size_t)u] == 2;
                            }
                        }
                        a = c2 > c1 ? 2 : 1;
                    }
                    arm[(size_t)v] = a;
                }
                std::vector<int32_t> out;
                out.reserve(w.size());
                for (size_t i = w.size(); i-- > lo;)
                    if (arm[(size_t)w[i]] == 2) out.push_back(w[i]);
                for (size_t i = 0; i < lo; i++) out.push_back(w[i]);
                for (size_t i = lo; i < w.size(); i++)
                    if (arm[(size_t)w[i]] == 1) out.push_back(w[i]);
                w.swap(out);
                return;
            }
        };
        // components by a first sweep (tags 0, 2, 4, ...), then re-walked from their last-discovered node
        // (tags 1, 3, 5, ...): a peripheral start keeps the levels thin
        std::vector<std::pair<int64_t, int32_t>> comps;   // (-size, index of the component)
        std::vector<int32_t> walk, more;
        std::vector<int32_t> first_order;                 // the first sweep's walks, one after the other
        std::vector<int64_t> first_begin;                 // component i = first_order[first_begin[i] .. first_begin[i + 1])
        first_order.reserve((size_t)n);
        int32_t tag = 0;
        for (int64_t v = 0; v < n; v++) {
            if (comp[(size_t)v] >= 0 || brp[(size_t)v + 1] == brp[(size_t)v]) continue;
            bfs((int32_t)v, tag, -1, walk);
            comps.emplace_back(-(int64_t)walk.size(), (int32_t)first_begin.size());
            first_begin.push_back((int64_t)first_order.size());
            first_order.insert(first_order.end(), walk.begin(), walk.end());
            tag += 2;
        }
        first_begin.push_back((int64_t)first_order.size());
        lap("first sweep (components)");
        if (debug) std::fprintf(stderr, "[reorder] %zu backbone components\n", comps.size());
        std::sort(comps.begin(), comps.end());
        int64_t at = 0;
        // Leftovers: with theta > 1 a node with fewer than theta backbone neighbours in a walk is never discovered and
        // seeds a tiny walk of its own later.  Laid out as components of their own they would land far from the
        // neighbourhood they belong to; they are placed afterwards, at the median position of their placed neighbours.
        const int64_t leftover_below = theta > 1 ? 16 : 1;
        std::vector<int32_t> leftovers;
        for (auto &c : comps) {
            const int64_t fb = first_begin[(size_t)c.second], fe = first_begin[(size_t)c.second + 1];
            const int32_t seed = first_order[(size_t)(fe - 1)];       // the first sweep's last node: a peripheral start
            const int32_t old_tag = comp[(size_t)seed];
            bfs(seed, old_tag + 1, old_tag, walk);
            // a walk from another start may stall before it has covered the component (a node needs theta walked
            // neighbours): go on from the nodes it has not reached, in the first sweep's order
            if ((int64_t)walk.size() < fe - fb) {
                for (int64_t i = fb; i < fe; i++) {
                    const int32_t v = first_order[(size_t)i];
                    if (comp[(size_t)v] != old_tag) continue;
                    const int32_t base_depth = walk.empty() ? 0 : depth[(size_t)walk.back()] + 1;
                    bfs(v, old_tag + 1, old_tag, more);
                    for (int32_t u : more) depth[(size_t)u] += base_depth;
                    walk.insert(walk.end(), more.begin(), more.end());
                }
            }
            if ((int64_t)walk.size() < leftover_below && -comps.front().first >= 16 * leftover_below) {
                leftovers.insert(leftovers.end(), walk.begin(), walk.end());
                continue;
            }
            if (debug && &c == &comps.front()) lap("  second sweep: largest component walked");
            unfold(walk);
            if (debug && &c == &comps.front()) lap("  second sweep: largest component unfolded");
            for (int32_t v : walk) { pos[(size_t)v] = (double)at++; in_backbone[(size_t)v] = 1; }
        }
        lap("  second sweep: the other components");
        if (debug) std::fprintf(stderr, "[reorder] %zu leftover nodes of tiny walks\n", leftovers.size());
        // a few rounds: a leftover whose neighbours are leftovers too gets its place once they have theirs
        for (int round = 0; round < 3 && !leftovers.empty(); round++) {
            std::vector<int32_t> still;
            std::vector<double> nbp;
            std::vector<std::pair<int32_t, double>> placed_now;
            for (int32_t v : leftovers) {
                nbp.clear();
                for (int64_t k = rp[(size_t)v]; k < rp[(size_t)v + 1]; k++)
                    if (pos[(size_t)ci[(size_t)k]] >= 0) nbp.push_back(pos[(size_t)ci[(size_t)k]]);
                if (nbp.empty()) { still.push_back(v); continue; }
                std::nth_element(nbp.begin(), nbp.begin() + nbp.size() / 2, nbp.end());
                placed_now.emplace_back(v, nbp[nbp.size() / 2] + 0.5);       // next to the median neighbour
            }
            for (auto &pv : placed_now) { pos[(size_t)pv.first] = pv.second; in_backbone[(size_t)pv.first] = 1; }
            leftovers.swap(still);
        }
        for (int64_t v = 0; v < n; v++)
            if (pos[(size_t)v] < 0) pos[(size_t)v] = (double)at++;        // no backbone edge: placed in step 3
        // (fractional positions of the leftovers become ranks at the first re-spread of step 3)
    }

    lap("second sweep + unfold");
    // ---- 3. barycentre refinement ----------------------------------------------------------------------------
    std::vector<int32_t> perm((size_t)n);
    // positions -> ranks, so that the arrangement does not contract (ties: by node id); returns how far a node moved on average
    // (the order is total -- position, then node id -- so the result does not depend on how the sort is split: the chunks are
    // sorted by the threads and merged pairwise; up to 17 of these per call since the sweeps run until settled)
    std::vector<std::pair<double, int32_t>> keyed((size_t)n), merged((size_t)n);
    auto respread = [&]() -> double {
        parallel_nodes(n, threads, [&](int64_t lo, int64_t hi) {
            for (int64_t v = lo; v < hi; v++) keyed[(size_t)v] = std::make_pair(nxt[(size_t)v], (int32_t)v);
        });
        const int64_t chunks = std::max<int64_t>(1, std::min<int64_t>(threads, n / 65536 + 1));
        const int64_t step = (n + chunks - 1) / chunks;
        parallel_tasks(chunks, threads, [&](int64_t c) {
            std::sort(keyed.begin() + std::min(n, c * step), keyed.begin() + std::min(n, (c + 1) * step));
        });
        auto *from = &keyed, *to = &merged;
        for (int64_t width = step; width < n; width *= 2) {
            const int64_t pairs = (n + 2 * width - 1) / (2 * width);
            parallel_tasks(pairs, threads, [&](int64_t p) {
                const int64_t a = p * 2 * width, m = std::min(n, a + width), z = std::min(n, a + 2 * width);
                std::merge(from->begin() + a, from->begin() + m, from->begin() + m, from->begin() + z, to->begin() + a);
            });
            std::swap(from, to);
        }
        for (int64_t r = 0; r < n; r++) perm[(size_t)r] = (*from)[(size_t)r].second;
        double moved = 0.0;
        for (int64_t r = 0; r < n; r++) {
            double &p = pos[(size_t)perm[(size_t)r]];
            moved += std::fabs((double)r - p);
            p = (double)r;
        }
        return moved / (double)n;
    };
    // Sweeps until the arrangement stands still: four were the fixed count until the Rabbit Order yardstick of round 5 showed
    // block-structured graphs only half way there (500 planted blocks, share of the edges within 4,096 ids: 0.632 after 4
    // sweeps, 0.678 after 8, 0.680 after 16 = converged; planted 0.698, Rabbit Order 0.690; products-like with hidden locality
    // 0.537 / 0.571 / 0.603, still moving at 32) while the window-structured Reddit-like graph is done after 4-8.  A sweep costs
    // 0.2-0.7 s at 0.2-2.4 M nodes.  The parallel (Jacobi) update never comes to rest -- neighbours keep trading places at
    // an average displacement of 150-1,000 ranks -- so the stop is relative: after at least four sweeps, the first sweep that
    // moves the nodes by more than 0.9 x what the sweep before it did (Reddit-like: 5 sweeps, blocks: 11, products-like: 10),
    // at most 16.
    const int forced_sweeps = std::getenv("GNNA_REORDER_SWEEPS") ? std::atoi(std::getenv("GNNA_REORDER_SWEEPS")) : -1;
    const int bsweeps = forced_sweeps >= 0 ? forced_sweeps : 16;
    double moved_before = -1.0;
    // One closing pass by the MODE instead of the median: a popular node whose neighbours are spread over many communities
    // (40 % in its own, the rest everywhere) has its median between communities.  Such a node goes to the middle of the SHORTEST
    // interval that holds half of its neighbours -- when that interval is clearly shorter (x 0.6) than the central half of them;
    // a node in the middle of a uniform neighbourhood (both about equal) keeps its median.  Measured (share of the edges within
    // 4,096 ids): 500 planted blocks 0.680 -> 0.687 (Rabbit Order 0.690, planted 0.698), products-like with hidden locality
    // 0.582 -> 0.59, Reddit-like with hidden locality 0.776 -> 0.78; a second pass adds nothing.  GNNA_REORDER_MODE=0 skips it.
    const int mode_passes = std::getenv("GNNA_REORDER_MODE") ? std::atoi(std::getenv("GNNA_REORDER_MODE")) : 1;
    const double mode_ratio = std::getenv("GNNA_REORDER_MODE_RATIO") ? std::atof(std::getenv("GNNA_REORDER_MODE_RATIO")) : 0.6;
    auto mode_sweep = [&] {
        parallel_nodes(n, threads, [&](int64_t lo, int64_t hi) {
            std::vector<double> x;
            for (int64_t v = lo; v < hi; v++) {
                const int64_t b = brp[(size_t)v], e = brp[(size_t)v + 1];
                nxt[(size_t)v] = pos[(size_t)v];
                const int64_t d = e - b;
                if (d < 8) continue;
                x.resize((size_t)d);
                for (int64_t k = b; k < e; k++) x[(size_t)(k - b)] = pos[(size_t)bci[(size_t)k]];
                std::sort(x.begin(), x.end());
                const int64_t h = (d + 1) / 2;
                double best = x[(size_t)(h - 1)] - x[0];
                int64_t at = 0;
                for (int64_t i = 1; i + h <= d; i++) {
                    const double w = x[(size_t)(i + h - 1)] - x[(size_t)i];
                    if (w < best) { best = w; at = i; }
                }
                const double central = x[(size_t)((3 * d) / 4)] - x[(size_t)(d / 4)];
                nxt[(size_t)v] = best < mode_ratio * central ? x[(size_t)(at + h / 2)] : x[(size_t)(d / 2)];
            }
        });
    };
    for (int it = 0; it < bsweeps; it++) {
        parallel_nodes(n, threads, [&](int64_t lo, int64_t hi) {
            std::vector<double> nbp;
            for (int64_t v = lo; v < hi; v++) {
                const int64_t b = brp[(size_t)v], e = brp[(size_t)v + 1];
                if (e == b) { nxt[(size_t)v] = pos[(size_t)v]; continue; }
                // the MEDIAN position of the neighbours: a stray long-range neighbour must not drag the node away
                nbp.resize((size_t)(e - b));
                for (int64_t k = b; k < e; k++) nbp[(size_t)(k - b)] = pos[(size_t)bci[(size_t)k]];
                std::nth_element(nbp.begin(), nbp.begin() + (e - b) / 2, nbp.end());
                nxt[(size_t)v] = nbp[(size_t)((e - b) / 2)];
            }
        });
        const double moved = respread();
        if (debug) std::fprintf(stderr, "[reorder] sweep %d: average displacement %.2f ranks\n", it + 1, moved);
        if (forced_sweeps < 0 && it + 1 >= 4 && moved_before >= 0.0 && moved > 0.9 * moved_before) break;
        moved_before = moved;
    }
    for (int mp = 0; mp < mode_passes; mp++) {
        mode_sweep();
        const double moved = respread();
        if (debug) std::fprintf(stderr, "[reorder] mode pass %d: average displacement %.2f ranks\n", mp + 1, moved);
    }
    // nodes outside the backbone (hubs, nodes whose edges are all unsupported): mean position of all neighbours
    // that are inside it
    parallel_nodes(n, threads, [&](int64_t lo, int64_t hi) {
        for (int64_t v = lo; v < hi; v++) {
            nxt[(size_t)v] = pos[(size_t)v];
            if (in_backbone[(size_t)v]) continue;
            std::vector<double> nbp;
            for (int64_t k = rp[(size_t)v]; k < rp[(size_t)v + 1]; k++)
                if (in_backbone[(size_t)ci[(size_t)k]]) nbp.push_back(pos[(size_t)ci[(size_t)k]]);
            if (!nbp.empty()) {
                std::nth_element(nbp.begin(), nbp.begin() + nbp.size() / 2, nbp.end());
                nxt[(size_t)v] = nbp[nbp.size() / 2];
            }
        }
    });
    respread();
    lap("median sweeps");
    for (int64_t v = 0; v < n; v++) new_id[(size_t)v] = (int32_t)pos[(size_t)v];
    return GNNA_OK;
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
    if (num_nodes > 0x7fffffffLL)
        return fail(GNNA_ERR_UNSUPPORTED, "graph too large for int32 node ids");
    const int64_t n = num_nodes;
    if (n == 0) return GNNA_OK;
    const int threads = host_threads();
    const bool debug = std::getenv("GNNA_REORDER_DEBUG") != nullptr;
    Lap lap(debug);

    // symmetrised, duplicate-free adjacency (reorder.cpp:31-97 does the same before aggregating).  64-bit row offsets:
    // the list may hold more than 2^31 entries once both directions are in (papers100M symmetrised: 3.2e9); built in
    // parallel over slabs of the edge list (atomic degree counts and cursors), then sorted and de-duplicated per row.
    std::vector<int64_t> rp((size_t)n + 1, 0);
    std::vector<int32_t> ci;
    {
        std::vector<int64_t> cursor((size_t)n + 1, 0);
        auto over_edges = [&](auto &&fn) {
            const int64_t nt = std::max<int64_t>(1, std::min<int64_t>(threads, num_edges / (1 << 20) + 1));
            std::vector<std::thread> th;
            const int64_t step = (num_edges + nt - 1) / nt;
            for (int64_t t = 0; t < nt; t++) {
                const int64_t lo = t * step, hi = std::min(num_edges, lo + step);
                if (lo >= hi) break;
                th.emplace_back([&fn, lo, hi] { fn(lo, hi); });
            }
            for (auto &t : th) t.join();
        };
        int64_t bad_edge = -1;
        over_edges([&](int64_t lo, int64_t hi) {
            for (int64_t e = lo; e < hi; e++)
                if (src[e] < 0 || src[e] >= n || dst[e] < 0 || dst[e] >= n) { __atomic_store_n(&bad_edge, e, __ATOMIC_RELAXED); return; }
        });
        if (bad_edge >= 0)
            return fail(GNNA_ERR_INVALID_ARGUMENT, "edge %lld (%d -> %d) outside [0, %lld)", (long long)bad_edge, src[bad_edge],
                        dst[bad_edge], (long long)n);
        over_edges([&](int64_t lo, int64_t hi) {
            for (int64_t e = lo; e < hi; e++) {
                __atomic_fetch_add(&cursor[(size_t)src[e] + 1], 1, __ATOMIC_RELAXED);
                __atomic_fetch_add(&cursor[(size_t)dst[e] + 1], 1, __ATOMIC_RELAXED);
            }
        });
        lap("  adjacency: counted");
        for (int64_t v = 0; v < n; v++) cursor[(size_t)v + 1] += cursor[(size_t)v];
        std::vector<int64_t> start(cursor.begin(), cursor.end());
        std::vector<int32_t> bucket((size_t)(2 * num_edges));
        over_edges([&](int64_t lo, int64_t hi) {
            for (int64_t e = lo; e < hi; e++) {
                bucket[(size_t)__atomic_fetch_add(&cursor[(size_t)src[e]], 1, __ATOMIC_RELAXED)] = dst[e];
                bucket[(size_t)__atomic_fetch_add(&cursor[(size_t)dst[e]], 1, __ATOMIC_RELAXED)] = src[e];
            }
        });
        lap("  adjacency: scattered");
        std::vector<int32_t> uniq((size_t)n, 0);
        parallel_nodes(n, threads, [&](int64_t lo, int64_t hi) {
            for (int64_t v = lo; v < hi; v++) {
                int32_t *b = bucket.data() + start[(size_t)v], *e = bucket.data() + start[(size_t)v + 1];
                std::sort(b, e);                                   // (the scatter order depends on the threads, the sorted row does not)
                uniq[(size_t)v] = (int32_t)(std::unique(b, e) - b);
            }
        });
        lap("  adjacency: rows sorted");
        for (int64_t v = 0; v < n; v++) rp[(size_t)v + 1] = rp[(size_t)v] + uniq[(size_t)v];
        ci.resize((size_t)rp[(size_t)n]);
        parallel_nodes(n, threads, [&](int64_t lo, int64_t hi) {
            for (int64_t v = lo; v < hi; v++)
                std::copy_n(bucket.data() + start[(size_t)v], uniq[(size_t)v], ci.data() + rp[(size_t)v]);
        });
    }

    lap("symmetrised adjacency");
    return community_order(n, rp, IdList{ci.data(), ci.size()}, new_id, threads, debug, lap);
}

// The same renumbering for a caller that already holds the graph as a CSR with sorted, duplicate-free rows (what the loader
// builds, dataset.py:108-118): a symmetric CSR IS the adjacency the algorithm works on -- nothing is counted, scattered or sorted
// again (1.2 of 4.6 s at 1.1e8 edges).  Symmetry is decided by two 64-bit sums over all entries, of a hash of (row, column) and
// of (column, row): equal for a symmetric matrix, different otherwise up to 2^-128.  A directed CSR goes through the edge-list path.
int gnna_reorder_community_csr_i32(const int32_t *row_pointers, const int32_t *column_index, int64_t num_nodes, int32_t *new_id)
{
    using gnna::fail;
    if (num_nodes < 0 || (num_nodes > 0 && (!new_id || !row_pointers)))
        return fail(GNNA_ERR_INVALID_ARGUMENT, "bad reorder arguments");
    const int64_t n = num_nodes;
    if (n == 0) return GNNA_OK;
    const int64_t nnz = row_pointers[n];
    if (row_pointers[0] != 0 || nnz < 0 || (nnz > 0 && !column_index)) return fail(GNNA_ERR_INVALID_ARGUMENT, "bad CSR");
    const int threads = host_threads();
    const bool debug = std::getenv("GNNA_REORDER_DEBUG") != nullptr;
    Lap lap(debug);
    auto mix = [](uint64_t x) { x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ull; x ^= x >> 27; x *= 0x94d049bb133111ebull; return x ^ (x >> 31); };
    std::vector<uint64_t> sums((size_t)threads * 4, 0);
    std::vector<int64_t> bad((size_t)threads, -1);
    {
        std::vector<std::thread> th;
        const int64_t step = (n + threads - 1) / threads;
        for (int t = 0; t < threads; t++)
            th.emplace_back([&, t] {
                uint64_t a0 = 0, a1 = 0, b0 = 0, b1 = 0;
                for (int64_t u = t * step; u < std::min(n, (t + 1) * step); u++) {
                    const int64_t b = row_pointers[u], e = row_pointers[u + 1];
                    if (e < b || e > nnz) { bad[(size_t)t] = u; return; }
                    for (int64_t k = b; k < e; k++) {
                        const int64_t v = column_index[k];
                        if (v < 0 || v >= n || (k > b && column_index[k - 1] >= v)) { bad[(size_t)t] = u; return; }
                        const uint64_t uv = ((uint64_t)u << 32) | (uint64_t)v, vu = ((uint64_t)v << 32) | (uint64_t)u;
                        a0 += mix(uv); a1 += mix(uv ^ 0x9E3779B97F4A7C15ull);
                        b0 += mix(vu); b1 += mix(vu ^ 0x9E3779B97F4A7C15ull);
                    }
                }
                sums[(size_t)t * 4] = a0; sums[(size_t)t * 4 + 1] = a1; sums[(size_t)t * 4 + 2] = b0; sums[(size_t)t * 4 + 3] = b1;
            });
        for (auto &t : th) t.join();
    }
    for (int64_t u : bad)
        if (u >= 0) return fail(GNNA_ERR_INVALID_ARGUMENT, "row %lld of the CSR: ids must lie in [0, %lld) in strictly increasing order",
                                (long long)u, (long long)n);
    uint64_t tot[4] = {0, 0, 0, 0};
    for (int t = 0; t < threads; t++) for (int q = 0; q < 4; q++) tot[q] += sums[(size_t)t * 4 + q];
    if (tot[0] != tot[2] || tot[1] != tot[3]) {
        if (debug) std::fprintf(stderr, "[reorder] the CSR is not symmetric: edge-list path\n");
        std::vector<int32_t> rows((size_t)nnz);
        parallel_nodes(n, threads, [&](int64_t lo, int64_t hi) {
            for (int64_t u = lo; u < hi; u++) std::fill(rows.begin() + row_pointers[u], rows.begin() + row_pointers[u + 1], (int32_t)u);
        });
        return gnna_reorder_community_i32(rows.data(), column_index, nnz, n, new_id);
    }
    std::vector<int64_t> rp((size_t)n + 1);
    for (int64_t u = 0; u <= n; u++) rp[(size_t)u] = row_pointers[u];
    lap("CSR checked (symmetric)");
    return community_order(n, rp, IdList{column_index, (size_t)nnz}, new_id, threads, debug, lap);
}

// Applies a renumbering to what the loader holds (dataset.py:147-172 relabels the edge list and rebuilds everything from it):
//   gnna_relabel_edges_i32: src[e] <- new_id[src[e]], dst[e] <- new_id[dst[e]] in place, and the new mean |src - dst|;
//   gnna_relabel_csr_i32:   the CSR of the relabelled graph from the CSR of the old one -- row new_id[u] is row u with its ids
//     mapped and sorted again; no global sort, no de-duplication (a permutation keeps distinct ids distinct).
int gnna_relabel_edges_i32(int32_t *src, int32_t *dst, int64_t num_edges, const int32_t *new_id, int64_t num_nodes, double *avg_edge_span)
{
    using gnna::fail;
    if (num_edges < 0 || num_nodes < 0 || (num_edges > 0 && (!src || !dst || !new_id)))
        return fail(GNNA_ERR_INVALID_ARGUMENT, "bad relabel arguments");
    const int threads = host_threads();
    const int64_t nt = std::max<int64_t>(1, std::min<int64_t>(threads, num_edges / (1 << 20) + 1));
    std::vector<double> span((size_t)nt, 0.0);
    std::vector<int64_t> bad((size_t)nt, -1);
    std::vector<std::thread> th;
    const int64_t step = (num_edges + nt - 1) / nt;
    for (int64_t t = 0; t < nt; t++)
        th.emplace_back([&, t] {
            int64_t acc = 0;
            for (int64_t e = t * step; e < std::min(num_edges, (t + 1) * step); e++) {
                if (src[e] < 0 || src[e] >= num_nodes || dst[e] < 0 || dst[e] >= num_nodes) { bad[(size_t)t] = e; return; }
                const int32_t a = new_id[src[e]], b = new_id[dst[e]];
                src[e] = a; dst[e] = b;
                acc += a > b ? (int64_t)a - b : (int64_t)b - a;
            }
            span[(size_t)t] = (double)acc;
        });
    for (auto &t : th) t.join();
    for (int64_t e : bad) if (e >= 0) return fail(GNNA_ERR_INVALID_ARGUMENT, "edge %lld outside [0, %lld)", (long long)e, (long long)num_nodes);
    if (avg_edge_span) {
        double total = 0.0;
        for (double v : span) total += v;
        *avg_edge_span = num_edges ? total / (double)num_edges : 0.0;
    }
    return GNNA_OK;
}

int gnna_relabel_csr_i32(const int32_t *row_pointers, const int32_t *column_index, int64_t num_nodes, const int32_t *new_id,
                         int32_t *out_row_pointers, int32_t *out_column_index)
{
    using gnna::fail;
    const int64_t n = num_nodes;
    if (n < 0 || (n > 0 && (!row_pointers || !new_id || !out_row_pointers)))
        return fail(GNNA_ERR_INVALID_ARGUMENT, "bad relabel arguments");
    if (n == 0) { if (out_row_pointers) out_row_pointers[0] = 0; return GNNA_OK; }
    const int threads = host_threads();
    std::vector<int32_t> old_of((size_t)n, -1);
    for (int64_t u = 0; u < n; u++) {
        const int32_t w = new_id[u];
        if (w < 0 || w >= n || old_of[(size_t)w] >= 0) return fail(GNNA_ERR_INVALID_ARGUMENT, "new_id is not a permutation (entry %lld)", (long long)u);
        old_of[(size_t)w] = (int32_t)u;
    }
    out_row_pointers[0] = 0;
    for (int64_t w = 0; w < n; w++) {
        const int32_t u = old_of[(size_t)w];
        const int64_t d = (int64_t)row_pointers[u + 1] - row_pointers[u];
        if (d < 0) return fail(GNNA_ERR_INVALID_ARGUMENT, "row_pointers decrease at row %d", u);
        out_row_pointers[w + 1] = (int32_t)(out_row_pointers[w] + d);
    }
    std::vector<int64_t> bad((size_t)threads + 1, -1);
    parallel_nodes(n, threads, [&](int64_t lo, int64_t hi) {
        for (int64_t w = lo; w < hi; w++) {
            const int32_t u = old_of[(size_t)w];
            int32_t *o = out_column_index + out_row_pointers[w];
            const int64_t b = row_pointers[u], e = row_pointers[u + 1];
            for (int64_t k = b; k < e; k++) {
                const int32_t v = column_index[k];
                if (v < 0 || v >= n) { __atomic_store_n(&bad[0], (int64_t)u, __ATOMIC_RELAXED); return; }
                o[k - b] = new_id[v];
            }
            std::sort(o, o + (e - b));
        }
    });
    if (bad[0] >= 0) return fail(GNNA_ERR_INVALID_ARGUMENT, "row %lld holds an id outside [0, %lld)", (long long)bad[0], (long long)n);
    return GNNA_OK;
}

#pragma GCC visibility pop
}  // extern "C"
