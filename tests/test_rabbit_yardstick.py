"""The Rabbit Order restatement (oracle/rabbit_yardstick.cpp: the reference's renumbering, rabbit_module/src/
rabbit_order.hpp:554-673 behind rabbit.reorder, reorder.cpp:235-295, executed on one thread) and the product's own
renumbering held to it.  The vendored Rabbit Order cannot be built here (boost / numa / tcmalloc), so the restatement
is pinned by hand-checkable cases and by the invariants the reference asserts on its own result (check_result,
rabbit_order.hpp:700-740; compute_perm's asserts, :665-671), not by outputs of the reference's binary."""
import numpy as np
import torch

import oracle
from gnnadvisor_osdi21_amd import _lib, graph


def _sym(pairs):
    e = np.array(list(pairs) + [(b, a) for a, b in pairs], dtype=np.int32)
    return e[:, 0].copy(), e[:, 1].copy()


def test_two_cliques_and_a_bridge_by_hand():
    """Two 4-cliques {0..3}, {4..7} joined by the edge 3 - 4.  By hand: every undirected pair weighs 2 after the
    symmetrisation of a list that holds both directions (total 52; str = 6 for the six degree-3 vertices, 8 for 3 and 4).
    Merge order = ascending degree, ties by id: 0 1 2 5 6 7 3 4.
      0: gains 1: 2 - 6*6/52, 2: the same, 3: 2 - 6*8/52 -> first maximum in target order: 0 -> 1 (str 12)
      1: united edges {2: 4, 3: 4} -> 2 (4 - 12*6/52 beats 4 - 12*8/52), str 18;   2: {3: 6} -> 3, str 26
      5: gains 4: 2 - 6*8/52, 6 and 7: 2 - 6*6/52 -> 6;   6: {4: 4, 7: 4} -> 7;   7: {4: 6} -> 4
      3: {4: 2}: 2 - 26*26/52 < 0 -> top-level; 4 likewise.
    Numbering (compute_perm pushes a vertex and its chain of `child` links, then pops from the BACK): community of 3:
    stack 3 2 1 0 -> 0, 1, 2, 3; community of 4: stack 4 7 6 5 -> 5, 6, 7, 4."""
    cl = [(i, j) for i in range(4) for j in range(i + 1, 4)]
    s, d = _sym(cl + [(i + 4, j + 4) for i, j in cl] + [(3, 4)])
    new, st = oracle.rabbit_yardstick(s, d, 8)
    assert st["communities"] == 2 and st["merged"] == 6 and st["adjacency_entries"] == 26
    # modularity of {0..3}, {4..7}: each community holds 24 of the 52 weight inside and 26 in total
    assert abs(st["modularity"] - 2 * (24 / 52 - (26 / 52) ** 2)) < 1e-12
    order = np.argsort(new).tolist()                       # order[k] = the vertex numbered k
    assert order == [0, 1, 2, 3, 5, 6, 7, 4], order


def test_edge_list_conventions_of_the_reference():
    """reorder.cpp:28-90: self-loops are dropped, a one-directional entry counts in both directions, duplicates add up;
    isolated vertices are communities of their own, numbered where the merge order (degree 0 first) puts them."""
    # path 0 - 1 - 2 given once per pair, one direction only, plus a loop and a duplicate; vertices 3, 4 isolated
    s = np.array([0, 1, 1, 2, 2], dtype=np.int32)
    d = np.array([1, 2, 2, 2, 1], dtype=np.int32)           # (1,2) twice and (2,1): weight 3; (2,2) dropped
    new, st = oracle.rabbit_yardstick(s, d, 5)
    assert st["adjacency_entries"] == 4 and sorted(new.tolist()) == list(range(5))
    assert st["communities"] == 3                           # {3}, {4}, {0, 1, 2}
    assert new[3] == 0 and new[4] == 1                      # degree-0 vertices come first in the merge order
    assert sorted(new[:3].tolist()) == [2, 3, 4]
    # empty graph, no vertices, bad ids
    new, st = oracle.rabbit_yardstick(np.zeros(0, np.int32), np.zeros(0, np.int32), 3)
    assert new.tolist() == [0, 1, 2] and st["communities"] == 3
    new, st = oracle.rabbit_yardstick(np.zeros(0, np.int32), np.zeros(0, np.int32), 0)
    assert new.size == 0
    try:
        oracle.rabbit_yardstick(np.array([0], np.int32), np.array([7], np.int32), 3)
        assert False, "an id outside the vertex range must be refused"
    except ValueError:
        pass


def _py_rabbit(src, dst, n):
    """A SECOND, independent restatement of the same algorithm in plain Python (dictionaries instead of sorted vectors,
    recursion-free), written from the reference's text -- adjacency reorder.cpp:28-90; merge order rabbit_order.hpp:527-538;
    unite :391-441; find_best :447-459; merge :470-522; compute_perm :633-673 -- not from the C++ restatement.  Small graphs
    only.  All weights are small integers, so float32 / float64 make no difference to any comparison."""
    adj = [dict() for _ in range(n)]
    for s, t in zip(src.tolist(), dst.tolist()):
        if s != t:
            adj[s][t] = adj[s].get(t, 0.0) + 1.0
            adj[t][s] = adj[t].get(s, 0.0) + 1.0
    es = [sorted(a.items()) for a in adj]
    strength = [float(sum(w for _, w in e)) for e in es]
    total = float(sum(strength))
    coms = list(range(n))
    child, sibling, united = [None] * n, [None] * n, [None] * n

    def trace(v):
        c = v
        while coms[c] != c:
            c = coms[c]
        return c

    tops = []
    for v in sorted(range(n), key=lambda u: (len(es[u]), u)):
        nb = {}

        def push(u):
            for t, w in es[u]:
                c = trace(t)
                if c != v:
                    nb[c] = nb.get(c, 0.0) + w
        push(v)
        while united[v] != child[v]:
            c = child[v]
            w_ = c
            while w_ is not None and w_ != united[v]:
                push(w_)
                w_ = sibling[w_]
            united[v] = c
        es[v] = sorted(nb.items())
        vstr = strength[v]
        strength[v] = -1.0
        best, dmax = v, 0.0
        for t, w in es[v]:
            d = w - vstr * strength[t] / total
            if dmax < d:
                dmax, best = d, t
        if best == v:
            strength[v] = vstr
            tops.append(v)
        else:
            sibling[v] = child[best]
            child[best] = v
            strength[best] += vstr
            coms[v] = best
    new_id = [0] * n
    base = 0
    for top in tops:
        stack, k = [], 0

        def descendants(u):
            while u is not None:
                stack.append(u)
                u = child[u]
        descendants(top)
        while stack:
            u = stack.pop()
            new_id[u] = base + k
            k += 1
            if sibling[u] is not None:
                descendants(sibling[u])
        base += k
    return np.array(new_id, dtype=np.int32), len(tops)


def test_two_independent_restatements_agree_on_random_small_graphs():
    """The C++ restatement (oracle/rabbit_yardstick.cpp) and the plain-Python one above produce the SAME permutation and the
    same number of communities on 60 random multigraphs with loops, duplicates, one-directional entries and isolated
    vertices -- what pins the yardstick in the absence of the reference's binary."""
    rng = np.random.default_rng(2024)
    for case in range(60):
        n = int(rng.integers(1, 120))
        m = int(rng.integers(0, 6 * n + 1))
        if case % 3 == 0:                                        # planted blocks: merges that matter
            k = max(1, n // 12)
            s = rng.integers(0, n, m)
            d = np.where(rng.random(m) < 0.8, (s // k) * k + rng.integers(0, k, m), rng.integers(0, n, m)).clip(0, n - 1)
        else:
            s, d = rng.integers(0, n, m), rng.integers(0, n, m)
        s, d = s.astype(np.int32), d.astype(np.int32)
        got, st = oracle.rabbit_yardstick(s, d, n)
        want, ntops = _py_rabbit(s, d, n)
        assert st["communities"] == ntops, (case, n, m)
        assert np.array_equal(got, want), (case, n, m)


def test_invariants_on_random_graphs_and_reproducibility():
    """What the reference asserts about its own result: the permutation is a bijection (compute_perm, :668-671); the
    members of a top-level community are numbered contiguously (offsets, :655-662); and a merge only ever happens for a
    positive modularity gain, so the communities' modularity is positive on graphs with community structure."""
    for seed, (n, e) in enumerate(((3000, 30000), (2000, 40000), (5000, 30000))):
        g = graph.powerlaw_graph(n, e, max(8, n // 10), locality=0.8, window=50, seed=seed)
        rows = torch.repeat_interleave(torch.arange(n), (g.row_pointers[1:] - g.row_pointers[:-1]).long()).numpy()
        cols = g.column_index.numpy()
        new, st = oracle.rabbit_yardstick(rows, cols, n)
        assert np.array_equal(np.sort(new), np.arange(n))
        again, _ = oracle.rabbit_yardstick(rows, cols, n)
        assert np.array_equal(new, again)                   # one thread, ties by id: reproducible
        assert st["merged"] + st["communities"] == n
        assert st["modularity"] > 0.3, st
        assert st["adjacency_entries"] == cols.size         # the generator's lists are symmetric, loop-free, deduplicated
        # contiguity of communities: the connected components of "merged into" are exactly st["communities"] id ranges --
        # checked through the edges: an order that keeps communities together has most edges inside a community-sized span
        span = np.abs(new[rows].astype(np.int64) - new[cols].astype(np.int64)).mean()
        scr = np.random.default_rng(seed).permutation(n)
        span_scr = np.abs(scr[rows].astype(np.int64) - scr[cols].astype(np.int64)).mean()
        assert span < 0.5 * span_scr, (span, span_scr)


def test_native_renumbering_against_the_rabbit_yardstick():
    """Row f-3's quality bar made explicit: on graphs with hidden structure and scrambled ids the product's renumbering
    (gnna_reorder_community_i32) leaves a mean edge span (the reference's own locality measure, dataset.py:99-100) no
    worse than 1.15 x what Rabbit Order -- the algorithm the reference runs -- reaches on the same edge list."""
    cases = [
        ("window 400", lambda: graph.powerlaw_graph(20000, 1_200_000, 1500, locality=0.9, window=400, seed=3)),
        ("window 1000", lambda: graph.powerlaw_graph(50000, 2_000_000, 1500, locality=0.9, window=1000, seed=5)),
        # families neither algorithm was shaped on: planted blocks (Rabbit Order's home ground) and R-MAT (no communities
        # to speak of: Rabbit finds thousands) -- measured span ratios product / Rabbit: 0.86, 0.80, 0.62
        ("100 blocks", lambda: graph.community_graph(50000, 3_000_000, 100, p_in=0.9, seed=7)),
        ("300 blocks, 80 % inside", lambda: graph.community_graph(30000, 1_500_000, 300, p_in=0.8, seed=8)),
        ("R-MAT 2^15", lambda: graph.rmat_graph(1 << 15, 2_000_000, seed=3)),
    ]
    for name, make in cases:
        g = make()
        n = g.num_nodes
        rows = torch.repeat_interleave(torch.arange(n), (g.row_pointers[1:] - g.row_pointers[:-1]).long())
        cols = g.column_index.long()
        perm = torch.randperm(n, generator=torch.Generator().manual_seed(1))
        src, dst = perm[rows], perm[cols]
        ours = _lib.reorder_community(src, dst, n).long()
        rb, st = oracle.rabbit_yardstick(src.numpy(), dst.numpy(), n)
        rb = torch.from_numpy(rb).long()
        span_ours = _lib.edge_span(ours[src], ours[dst])
        span_rabbit = _lib.edge_span(rb[src], rb[dst])
        span_scrambled = _lib.edge_span(src, dst)
        assert span_rabbit < (0.8 if name.startswith("R-MAT") else 0.5) * span_scrambled, (name, span_rabbit, span_scrambled, st)   # the yardstick itself works here
        assert span_ours <= 1.15 * span_rabbit, (name, span_ours, span_rabbit, span_scrambled)
