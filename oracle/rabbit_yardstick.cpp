// rabbit_yardstick.cpp -- TEST INFRASTRUCTURE ONLY: a one-thread restatement of Rabbit Order, the renumbering the
// reference calls from custom_dataset.rabbit_reorder() (GNNAdvisor/dataset.py:138-172 -> rabbit.reorder,
// rabbit_module/src/reorder.cpp:235-295).  It is the YARDSTICK the product's own renumbering
// (gnna_reorder_community_i32, csrc/gnna_reorder.cpp -- a different algorithm) is held to in tests/ and in
// tools/probe_reorder_quality.py: same graph, both orders, the reference's locality measure (mean |src - dst|,
// dataset.py:99-100) and the aggregation time side by side.  Only tests/, tools/ and bench.py's optional yardstick leg
// may load it; the product never does.
//
// What is restated (algorithm of J. Arai et al., "Rabbit Order: Just-in-time Parallel Reordering for Fast Graph
// Analysis", IPDPS 2016, as the vendored rabbit_module implements it):
//   * the adjacency it works on: edge list symmetrised, self-loops dropped, duplicate (s, t) summed -- every directed
//     entry weighs 1 in both directions (reorder.cpp make_adj_list, :28-90; rabbit_reorder pushes weight 1.0f, :254);
//   * incremental aggregation (rabbit_order.hpp aggregate, :554-631): vertices in ascending order of degree; a vertex v
//     first unites the edges of itself and of the vertices merged into it since (unite, :391-441: targets replaced by
//     their current community, self-loops dropped, duplicates summed, sorted by target), then is merged into the
//     neighbour community u with the largest modularity gain  w(v,u) - str(v) str(u) / total  if that is positive
//     (find_best, :447-459: first maximum in target order), else stays a top-level community (merge, :470-522);
//   * the dendrogram: u.child = v, v.sibling = u's previous child (merge, :497-512);
//   * the permutation (compute_perm, :633-673): per top-level community a depth-first walk -- a vertex, then its chain
//     of `child` links, a popped vertex pushes the chain of its sibling -- ids numbered in pop order, communities
//     concatenated in the order they became top-level.
// What is NOT the reference: its threads.  The reference runs the vertex loop under OpenMP with lock-free merges
// (schedule(static, 1), retry queues) and sorts the merge order with a parallel, unstable sort, so its permutation differs
// from run to run; this file is the np = 1 execution with ties of equal degree broken by vertex id.  PARITY UNPINNED: the
// vendored source needs boost, libnuma and tcmalloc, none of which is in the image, so no output of the reference itself
// exists to compare with.  What pins this file instead (tests/test_rabbit_yardstick.py): a case walked by hand, the invariants
// of the reference's own check_result (:700-740), and a SECOND restatement written independently from the reference's text in
// plain Python (dictionaries, no shared code) that must give the identical permutation on 60 random multigraphs.
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdlib>
#include <utility>
#include <vector>

namespace {

typedef uint32_t vint;
constexpr vint kNone = 0xffffffffu;                      // the reference's vmax
typedef std::pair<vint, float> Edge;

double now_sec()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

struct Graph {
    vint n = 0;
    // the symmetrised, aggregated input adjacency (never modified: compute_modularity reads it at the end)
    std::vector<int64_t> off;
    std::vector<vint> tgt;
    std::vector<float> wgt;
    // state of the aggregation
    std::vector<std::vector<Edge>> es;                   // united edge list of a processed vertex
    std::vector<char> has_es;                            // 0: the vertex' edges are still its input slice
    std::vector<float> str;                              // total weighted degree of the community's members; < 0: merged
    std::vector<vint> child, sibling, united_child, coms;
    double tot_wgt = 0.0;
    std::vector<vint> tops;
};

// rabbit_order.hpp:343-375 (one step of path compression on the way)
vint trace_com(Graph &g, vint v)
{
    vint com = v;
    for (;;) {
        const vint c = g.coms[com];
        if (c == com) break;
        com = c;
    }
    if (v != com && g.coms[v] != com) g.coms[v] = com;
    return com;
}

// rabbit_order.hpp:381-389: sort by target, sum the weights of equal targets
void compact(std::vector<Edge> &v, size_t from)
{
    if (v.size() - from < 2) return;
    std::sort(v.begin() + from, v.end(), [](const Edge &a, const Edge &b) { return a.first < b.first; });
    size_t o = from;
    for (size_t i = from + 1; i < v.size(); i++) {
        if (v[i].first == v[o].first) v[o].second += v[i].second;
        else v[++o] = v[i];
    }
    v.resize(o + 1);
}

// rabbit_order.hpp:391-441
void unite(Graph &g, vint v, std::vector<Edge> &nbrs)
{
    nbrs.clear();
    size_t icmb = 0;
    auto push_edges = [&](vint u) {
        if (g.has_es[u]) {
            for (const Edge &e : g.es[u]) {
                const vint c = trace_com(g, e.first);
                if (c != v) nbrs.push_back(Edge(c, e.second));
            }
        } else {
            for (int64_t i = g.off[u]; i < g.off[u + 1]; i++) {
                const vint c = trace_com(g, g.tgt[i]);
                if (c != v) nbrs.push_back(Edge(c, g.wgt[i]));
            }
        }
        if (nbrs.size() - icmb >= 2048) {                 // (:419-423: keeps the buffer small; the result does not depend on it)
            compact(nbrs, icmb);
            icmb = nbrs.size();
        }
    };
    push_edges(v);
    while (g.united_child[v] != g.child[v]) {
        const vint c = g.child[v];
        for (vint w = c; w != kNone && w != g.united_child[v]; w = g.sibling[w]) {
            push_edges(w);
            if (g.has_es[w]) std::vector<Edge>().swap(g.es[w]);   // (never read again: w's edges now live in v's list)
        }
        g.united_child[v] = c;
    }
    compact(nbrs, 0);
    g.es[v].assign(nbrs.begin(), nbrs.end());
    g.has_es[v] = 1;
}

// rabbit_order.hpp:447-459
vint find_best(const Graph &g, vint v, double vstr)
{
    double dmax = 0.0;
    vint best = v;
    for (const Edge &e : g.es[v]) {
        const double d = (double)e.second - vstr * (double)g.str[e.first] / g.tot_wgt;
        if (dmax < d) { dmax = d; best = e.first; }
    }
    return best;
}

// rabbit_order.hpp:470-522 without the contention paths (one thread: a lock is never found taken, a CAS never fails)
vint merge(Graph &g, vint v, std::vector<Edge> &nbrs)
{
    unite(g, v, nbrs);
    const float vstr = g.str[v];
    g.str[v] = -1.f;
    const vint u = find_best(g, v, (double)vstr);
    if (u == v) {
        g.str[v] = vstr;
    } else {
        g.sibling[v] = g.child[u];
        g.str[u] = g.str[u] + vstr;
        g.child[u] = v;
        g.coms[v] = u;
    }
    return u;
}

}  // namespace

// new_id[v] = position of vertex v in the Rabbit order.  src / dst: any directed edge list over ids 0 .. num_nodes - 1
// (the reference passes edge_index as it stands).  stats (optional, 8 doubles): 0 top-level communities, 1 modularity of
// the top-level communities on the input adjacency (reorder.cpp compute_modularity, :138-181), 2 seconds adjacency,
// 3 seconds aggregation, 4 seconds permutation, 5 vertices merged, 6 adjacency entries after symmetrisation, 7 unused.
// Returns 0, or -1 on a bad argument.
extern "C" __attribute__((visibility("default")))
int rabbit_yardstick_i32(const int32_t *src, const int32_t *dst, int64_t num_edges, int32_t num_nodes, int32_t *new_id,
                         double *stats)
{
    if (num_nodes < 0 || num_edges < 0 || (num_edges > 0 && (!src || !dst)) || (num_nodes > 0 && !new_id)) return -1;
    for (int64_t i = 0; i < num_edges; i++)
        if (src[i] < 0 || src[i] >= num_nodes || dst[i] < 0 || dst[i] >= num_nodes) return -1;
    Graph g;
    const vint n = (vint)num_nodes;
    g.n = n;
    const double t0 = now_sec();

    // ---- adjacency: both directions of every entry, loops dropped, duplicates summed (reorder.cpp:28-90) ----
    {
        std::vector<int64_t> cnt((size_t)n + 1, 0);
        for (int64_t i = 0; i < num_edges; i++)
            if (src[i] != dst[i]) { cnt[(size_t)src[i] + 1]++; cnt[(size_t)dst[i] + 1]++; }
        for (vint v = 0; v < n; v++) cnt[v + 1] += cnt[v];
        std::vector<vint> raw((size_t)cnt[n]);
        {
            std::vector<int64_t> cur(cnt.begin(), cnt.end() - 1);
            for (int64_t i = 0; i < num_edges; i++)
                if (src[i] != dst[i]) {
                    raw[(size_t)cur[src[i]]++] = (vint)dst[i];
                    raw[(size_t)cur[dst[i]]++] = (vint)src[i];
                }
        }
        g.off.assign((size_t)n + 1, 0);
        std::vector<int64_t> uniq((size_t)n, 0);
#pragma omp parallel for schedule(dynamic, 256)
        for (int64_t v = 0; v < (int64_t)n; v++) {
            std::sort(raw.begin() + cnt[v], raw.begin() + cnt[v + 1]);
            int64_t k = 0;
            for (int64_t i = cnt[v]; i < cnt[v + 1]; i++)
                if (i == cnt[v] || raw[(size_t)i] != raw[(size_t)i - 1]) k++;
            uniq[(size_t)v] = k;
        }
        for (vint v = 0; v < n; v++) g.off[v + 1] = g.off[v] + uniq[v];
        g.tgt.resize((size_t)g.off[n]);
        g.wgt.resize((size_t)g.off[n]);
#pragma omp parallel for schedule(dynamic, 256)
        for (int64_t v = 0; v < (int64_t)n; v++) {
            int64_t o = g.off[(size_t)v] - 1;
            for (int64_t i = cnt[v]; i < cnt[v + 1]; i++) {
                if (i == cnt[v] || raw[(size_t)i] != raw[(size_t)i - 1]) { o++; g.tgt[(size_t)o] = raw[(size_t)i]; g.wgt[(size_t)o] = 1.f; }
                else g.wgt[(size_t)o] += 1.f;
            }
        }
    }
    // ---- graph state (rabbit_order.hpp graph::graph, :296-315) ----
    g.es.resize(n);
    g.has_es.assign(n, 0);
    g.str.resize(n);
    g.child.assign(n, kNone);
    g.sibling.assign(n, kNone);
    g.united_child.assign(n, kNone);
    g.coms.resize(n);
    for (vint v = 0; v < n; v++) {
        float s = 0.f;
        for (int64_t i = g.off[v]; i < g.off[v + 1]; i++) s += g.wgt[(size_t)i];
        g.str[v] = s;
        g.tot_wgt += (double)s;
        g.coms[v] = v;
    }
    const double t1 = now_sec();

    // ---- merge order: ascending degree (merge_order, :527-538), ties by id ----
    std::vector<vint> ord(n);
    for (vint v = 0; v < n; v++) ord[v] = v;
    std::stable_sort(ord.begin(), ord.end(), [&](vint a, vint b) { return g.off[a + 1] - g.off[a] < g.off[b + 1] - g.off[b]; });

    // ---- incremental aggregation (aggregate, :554-631; one thread: nothing is ever left pending) ----
    std::vector<Edge> nbrs;
    int64_t merged = 0;
    for (vint i = 0; i < n; i++) {
        const vint v = ord[i];
        const vint u = merge(g, v, nbrs);
        if (u == v) g.tops.push_back(v);
        else merged++;
    }
    const double t2 = now_sec();

    // ---- permutation (compute_perm, :633-673) ----
    std::vector<vint> com_of(n), local(n);
    std::vector<int64_t> offsets(g.tops.size() + 1, 0);
    {
        std::vector<vint> stack;
        auto descendants = [&](vint v) {                      // :548-552: v, then the chain of `child` links below it
            stack.push_back(v);
            while ((v = g.child[v]) != kNone) stack.push_back(v);
        };
        for (size_t comid = 0; comid < g.tops.size(); comid++) {
            vint newid = 0;
            descendants(g.tops[comid]);
            while (!stack.empty()) {
                const vint v = stack.back();
                stack.pop_back();
                com_of[v] = (vint)comid;
                local[v] = newid++;
                if (g.sibling[v] != kNone) descendants(g.sibling[v]);
            }
            offsets[comid + 1] = newid;
        }
        for (size_t c = 0; c < g.tops.size(); c++) offsets[c + 1] += offsets[c];
    }
    if (offsets.back() != (int64_t)n) return -2;              // (the reference asserts this, :665)
    for (vint v = 0; v < n; v++) new_id[v] = (int32_t)(local[v] + offsets[com_of[v]]);
    const double t3 = now_sec();

    if (stats) {
        // modularity of the top-level communities on the input adjacency (reorder.cpp:138-181)
        std::vector<double> all(g.tops.size(), 0.0), loop(g.tops.size(), 0.0);
        double m2 = 0.0;
        for (vint v = 0; v < n; v++) {
            const vint c = com_of[v];
            for (int64_t i = g.off[v]; i < g.off[v + 1]; i++) {
                m2 += g.wgt[(size_t)i];
                all[c] += g.wgt[(size_t)i];
                if (com_of[g.tgt[(size_t)i]] == c) loop[c] += g.wgt[(size_t)i];
            }
        }
        double q = 0.0;
        if (m2 > 0.0)
            for (size_t c = 0; c < g.tops.size(); c++) q += loop[c] / m2 - (all[c] / m2) * (all[c] / m2);
        stats[0] = (double)g.tops.size();
        stats[1] = q;
        stats[2] = t1 - t0;
        stats[3] = t2 - t1;
        stats[4] = t3 - t2;
        stats[5] = (double)merged;
        stats[6] = (double)g.off[n];
        stats[7] = 0.0;
    }
    return 0;
}
