"""Graph input formats, native host CSR builder, renumbering hook (CPU only)."""
import os

import numpy as np
import pytest
import torch

import oracle
from gnnadvisor_osdi21_amd import _lib, graph
from gnnadvisor_osdi21_amd.loader import custom_dataset


def _edges(seed, n, e):
    rng = np.random.default_rng(seed)
    return rng.integers(0, n, e), rng.integers(0, n, e)


@pytest.mark.parametrize("n,e", [(1, 0), (5, 1), (50, 600), (400, 3000), (3000, 200000)])
def test_host_csr_builder_matches_scipy_semantics(n, e):
    src, dst = _edges(n + e, n, e)
    rp, ci = _lib.csr_from_edges(src, dst, n)
    orp, oci = oracle.np_csr_from_edges(src, dst, n)
    assert rp.dtype == torch.int32 and ci.dtype == torch.int32
    assert np.array_equal(rp.numpy(), orp) and np.array_equal(ci.numpy(), oci)
    np.testing.assert_array_equal(_lib.degrees(rp).numpy(), oracle.np_degrees(orp))
    if e:
        assert abs(_lib.edge_span(src, dst) - np.mean(np.abs(src - dst))) < 1e-9


def test_host_csr_builder_rejects_bad_edges():
    with pytest.raises(_lib.GnnaError, match="outside"):
        _lib.csr_from_edges(np.array([0, 7]), np.array([1, 1]), 5)
    with pytest.raises(_lib.GnnaError):
        _lib.csr_from_edges(np.array([0, -1]), np.array([1, 1]), 5)


def test_survey_probe_multigraph():
    # src=[0,0,0,2,2,1], dst=[3,1,3,0,0,2] -> indptr [0,2,3,4,4], indices [1,3,2,0]
    rp, ci = _lib.csr_from_edges([0, 0, 0, 2, 2, 1], [3, 1, 3, 0, 0, 2], 4)
    assert rp.tolist() == [0, 2, 3, 4, 4] and ci.tolist() == [1, 3, 2, 0]


def test_reorder_is_a_permutation_and_shrinks_span():
    rng = np.random.default_rng(2)
    n = 5000
    u = np.arange(n).repeat(5)
    v = (u + rng.integers(1, 25, u.size)) % n          # banded graph ...
    perm = rng.permutation(n)
    su, sv = perm[u], perm[v]                           # ... with scrambled ids
    new = _lib.reorder_rcm(su, sv, n).numpy()
    assert sorted(new.tolist()) == list(range(n))
    before, after = np.mean(np.abs(su - sv)), np.mean(np.abs(new[su] - new[sv]))
    assert after < 0.05 * before
    # deterministic
    assert np.array_equal(new, _lib.reorder_rcm(su, sv, n).numpy())
    # isolated nodes and several components are all numbered
    new2 = _lib.reorder_rcm(np.array([0, 5]), np.array([1, 6]), 10).numpy()
    assert sorted(new2.tolist()) == list(range(10))


def test_community_reorder_recovers_locality_where_a_bfs_sweep_cannot():
    """gnna_reorder_community_i32 on a power-law graph whose edges are 90 % local (within a window of ids) and
    whose ids were then scrambled: a valid, reproducible permutation that brings the average edge span back to
    within a small factor of the hidden ordering's.  The 10 % long-range edges make the graph a small world, which
    is exactly where the BFS-based reverse Cuthill-McKee sweep gains nothing."""
    from gnnadvisor_osdi21_amd import graph
    n = 20000
    g = graph.powerlaw_graph(n, 1_200_000, 1500, locality=0.9, window=400, seed=3)
    rows = torch.repeat_interleave(torch.arange(n), (g.row_pointers[1:] - g.row_pointers[:-1]).long())
    cols = g.column_index.long()
    hidden = _lib.edge_span(rows, cols)
    perm = torch.randperm(n, generator=torch.Generator().manual_seed(1))
    src, dst = perm[rows], perm[cols]
    scrambled = _lib.edge_span(src, dst)
    new = _lib.reorder_community(src, dst, n)
    assert sorted(new.tolist()) == list(range(n))
    assert torch.equal(new, _lib.reorder_community(src, dst, n))          # deterministic (any thread count)
    after = _lib.edge_span(new.long()[src], new.long()[dst])
    rcm = _lib.reorder_rcm(src, dst, n).long()
    after_rcm = _lib.edge_span(rcm[src], rcm[dst])
    assert scrambled > 5 * hidden
    assert after < 0.4 * scrambled and after < 3.0 * hidden, (hidden, scrambled, after)
    assert after < 0.6 * after_rcm, (after, after_rcm)
    # isolated nodes, several components, an empty graph
    new2 = _lib.reorder_community(np.array([0, 5]), np.array([1, 6]), 10).numpy()
    assert sorted(new2.tolist()) == list(range(10))
    assert _lib.reorder_community(np.zeros(0, dtype=np.int32), np.zeros(0, dtype=np.int32), 3).tolist() == [0, 1, 2]


def test_community_reorder_groups_the_blocks_of_a_block_model():
    """Planted communities (stochastic block model: 90 % of the edges inside blocks of ~500 nodes), ids scrambled:
    after gnna_reorder_community_i32 the nodes of a block are contiguous -- most edges end within one block size."""
    rng = np.random.default_rng(4)
    n, blocks, m = 16000, 32, 400000
    block = rng.permutation(n) % blocks                      # scrambled ids: block membership is arbitrary in id space
    members = [np.flatnonzero(block == b) for b in range(blocks)]
    u = rng.integers(0, n, m)
    inside = rng.random(m) < 0.9
    v = np.where(inside, [members[block[x]][rng.integers(0, len(members[block[x]]))] for x in u], rng.integers(0, n, m))
    src, dst = np.concatenate([u, v]), np.concatenate([v, u])
    new = _lib.reorder_community(src, dst, n).numpy().astype(np.int64)
    assert sorted(new.tolist()) == list(range(n))
    size = n // blocks
    near_before = np.mean(np.abs(src - dst) <= size)
    near_after = np.mean(np.abs(new[src] - new[dst]) <= size)
    assert near_before < 0.1 and near_after > 0.8, (near_before, near_after)
    # every block occupies (almost) one contiguous range of new ids
    spans = [np.percentile(new[mem], 95) - np.percentile(new[mem], 5) for mem in members]
    assert np.median(spans) < 1.5 * size, np.median(spans)


def test_dataset_formats_and_fields(tmp_path):
    src, dst = _edges(5, 60, 500)
    np.savez(tmp_path / "g.npz", src_li=src, dst_li=dst, num_nodes=60)
    with open(tmp_path / "g", "w") as f:
        for a, b in zip(src, dst):
            f.write(f"{a} {b}\n")
    a = custom_dataset(str(tmp_path / "g.npz"), 12, 4, load_from_txt=False, device="cpu")
    b = custom_dataset(str(tmp_path / "g"), 12, 4, load_from_txt=True, device="cpu")
    orp, oci = oracle.np_csr_from_edges(src, dst, 60)
    for ds in (a, b):
        assert ds.num_edges == 500 and ds.num_features == 12 and ds.num_classes == 4
        assert np.array_equal(ds.row_pointers.numpy(), orp) and np.array_equal(ds.column_index.numpy(), oci)
        assert ds.avg_degree == 500 / ds.num_nodes
        assert abs(ds.avg_edgeSpan - np.mean(np.abs(src - dst))) < 1e-9
        assert ds.x.shape == (ds.num_nodes, 12) and ds.y.dtype == torch.long and bool((ds.y == 1).all())
        assert ds.edge_index.shape == (2, 500) and len(ds.val) == 500
        assert int(ds.train_mask.sum()) == ds.num_nodes and int(ds.val_mask.sum()) == int(ds.num_nodes * 0.3)
        np.testing.assert_array_equal(ds.degrees.numpy(), oracle.np_degrees(orp))
    assert a.num_nodes == 60 and b.num_nodes == int(max(src.max(), dst.max())) + 1
    with pytest.raises(ValueError):
        custom_dataset(str(tmp_path / "g"), 12, 4, load_from_txt=False, device="cpu")


def test_reorder_hook_rebuilds_csr_and_degrees():
    src, dst = _edges(9, 300, 2000)
    ds = custom_dataset.from_edges(src, dst, 300, 8, 3, device="cpu")
    rp0 = ds.row_pointers.clone()
    ds.rabbit_reorder()                                  # flag not set: no-op (dataset.py:146-148)
    assert torch.equal(ds.row_pointers, rp0)
    ds.reorder_flag = True
    ds.rabbit_reorder()
    assert ds.edge_index.shape == (2, 2000)
    orp, oci = oracle.np_csr_from_edges(ds.edge_index[0], ds.edge_index[1], 300)
    assert np.array_equal(ds.row_pointers.numpy(), orp) and np.array_equal(ds.column_index.numpy(), oci)
    np.testing.assert_array_equal(ds.degrees.numpy(), oracle.np_degrees(orp))
    # relabelling preserves the degree multiset
    assert sorted(np.diff(orp).tolist()) == sorted(np.diff(rp0.numpy()).tolist())


# ---- sharded ingestion (every rank builds only its destination range; 64-bit edge counts) ------------------------
def _edge_list_with_duplicates(n=700, e=12000, seed=5):
    g = graph.powerlaw_graph(n, e, 80, seed=seed)
    rows = torch.repeat_interleave(torch.arange(n), (g.row_pointers[1:] - g.row_pointers[:-1]).long()).numpy()
    cols = g.column_index.numpy()
    rng = np.random.default_rng(seed)
    dup = rng.integers(0, len(rows), size=500)                       # duplicates: merged per shard, as by the loader
    order = rng.permutation(len(rows) + len(dup))
    return g, np.concatenate([rows, rows[dup]])[order], np.concatenate([cols, cols[dup]])[order]


@pytest.mark.parametrize("world", [1, 2, 5])
@pytest.mark.parametrize("fmt", ["npz", "txt"])
def test_sharded_loader_builds_exactly_each_ranks_rows(tmp_path, world, fmt):
    from gnnadvisor_osdi21_amd import loader
    g, src, dst = _edge_list_with_duplicates()
    if fmt == "npz":
        path = tmp_path / "g.npz"
        np.savez(path, src_li=src, dst_li=dst, num_nodes=g.num_nodes)
    else:
        path = tmp_path / "g.txt"
        np.savetxt(path, np.stack([src, dst], 1), fmt="%d")
    total = 0
    for rank in range(world):
        sh = loader.load_graph_shard(str(path), rank, world, load_from_txt=(fmt == "txt"))
        lo, hi = sh.row_range
        assert sh.bounds[0] == 0 and sh.bounds[-1] == g.num_nodes and sh.bounds == sorted(sh.bounds)
        want_rp = (g.row_pointers[lo:hi + 1] - g.row_pointers[lo]).int()
        assert torch.equal(sh.row_pointers, want_rp)
        assert torch.equal(sh.column_index, g.column_index[int(g.row_pointers[lo]):int(g.row_pointers[hi])])
        assert torch.allclose(sh.degrees, g.degrees[lo:hi])
        assert sh.num_edges == len(src) and abs(sh.avg_degree - len(src) / g.num_nodes) < 1e-9
        total += sh.column_index.numel()
    assert total == g.nnz
    # blocks are balanced by edge count, not by row count
    if world > 1:
        per = [int(g.row_pointers[sh.bounds[r + 1]] - g.row_pointers[sh.bounds[r]]) for r in range(world)]
        assert max(per) - min(per) <= 2 * int((g.row_pointers[1:] - g.row_pointers[:-1]).max()) + 600


def test_row_counts_and_splits_are_64_bit():
    """papers100M symmetrised has 3.2e9 edges: counts, global row pointers and the nnz-balanced bounds use int64; a
    single shard beyond 2^31 - 1 edges is refused."""
    counts = torch.full((1000,), 3_000_000, dtype=torch.int64)
    counts[::7] += 1_234_567
    bounds, rp = _lib.row_splits(counts, 8, want_row_pointers=True)
    total = int(counts.sum())
    assert total > 2 ** 31 and int(rp[-1]) == total and rp.dtype == torch.int64
    assert bounds[0] == 0 and bounds[-1] == 1000
    per = [int(rp[bounds[i + 1]] - rp[bounds[i]]) for i in range(8)]
    assert max(per) - min(per) <= 2 * int(counts.max())
    # accumulate in pieces (how a > 2^31-entry list is fed)
    rows = torch.randint(0, 50, (10000,), dtype=torch.int32, generator=torch.Generator().manual_seed(1))
    c = _lib.row_counts(rows[:4000], 50)
    _lib.row_counts(rows[4000:], 50, c)
    assert torch.equal(c, torch.bincount(rows.long(), minlength=50))
    with pytest.raises(_lib.GnnaError):
        _lib.row_counts(torch.tensor([0, 51], dtype=torch.int32), 50)
    with pytest.raises(_lib.GnnaError):
        _lib.csr_from_edges_range(torch.tensor([0, 1], dtype=torch.int32), torch.tensor([1, 0], dtype=torch.int32), 2, 0, 2,
                                  capacity=1)                        # capacity too small


def test_sharded_loader_with_renumbering_is_a_relabelled_graph():
    from gnnadvisor_osdi21_amd import loader
    g, src, dst = _edge_list_with_duplicates(n=400, e=6000, seed=8)
    shards = [loader.load_graph_shard(None, r, 3, reorder=True, _edges=(src, dst, 400)) for r in range(3)]
    new_id = shards[0].new_id.numpy()
    assert sorted(new_id.tolist()) == list(range(400))
    assert all(np.array_equal(s.new_id.numpy(), new_id) for s in shards)
    # the union of the shards is the CSR of the relabelled edge list
    rp_all, ci_all = _lib.csr_from_edges(new_id[src], new_id[dst], 400)
    for s in shards:
        lo, hi = s.row_range
        assert torch.equal(s.column_index, ci_all[int(rp_all[lo]):int(rp_all[hi])])


def test_rmat_and_community_generators_are_seeded_symmetric_and_shaped_as_named():
    """SURVEY 8(d) names R-MAT (a, b, c = 0.57, 0.19, 0.19) beside the Chung-Lu generator; the community generator is the
    structured counterpart the selection rules are also checked on.  Both: symmetric, no self loops, sorted rows,
    reproducible from the seed; R-MAT is heavy-tailed, the community graph keeps most edges inside a block and hides that
    when scrambled."""
    import scipy.sparse as sp
    for make in (lambda s: graph.rmat_graph(3000, 80000, seed=s), lambda s: graph.community_graph(3000, 60000, 10, seed=s)):
        g, g_again, g_other = make(5), make(5), make(6)
        assert torch.equal(g.column_index, g_again.column_index) and torch.equal(g.row_pointers, g_again.row_pointers)
        assert not torch.equal(g.column_index, g_other.column_index[:g.nnz]) or g.nnz != g_other.nnz
        rp, ci = g.row_pointers.numpy(), g.column_index.numpy()
        A = sp.csr_matrix((np.ones(len(ci)), ci, rp), shape=(g.num_nodes, g.num_nodes))
        assert (A != A.T).nnz == 0 and A.diagonal().sum() == 0
        for i in (0, 17, g.num_nodes - 1):
            row = ci[rp[i]:rp[i + 1]]
            assert (np.diff(row) > 0).all()
        assert np.allclose(g.degrees.numpy(), np.sqrt(np.maximum(np.diff(rp), 1)))
    r = graph.rmat_graph(1 << 12, 120000, seed=1)
    deg = np.diff(r.row_pointers.numpy())
    assert deg.max() > 20 * deg.mean() and np.median(deg) < deg.mean()          # heavy tail
    c = graph.community_graph(6000, 120000, 10, seed=2, p_in=0.9)
    rows = np.repeat(np.arange(6000), np.diff(c.row_pointers.numpy()))
    inside = (rows // 600 == c.column_index.numpy() // 600).mean()
    assert inside > 0.8
    cs = graph.community_graph(6000, 120000, 10, seed=2, p_in=0.9, scramble=True)
    assert cs.nnz == c.nnz and cs.avg_edgeSpan > 2.5 * c.avg_edgeSpan


def test_renumbering_and_csr_do_not_depend_on_the_thread_count():
    """The host passes split their work by the threads the container grants (GNNA_HOST_THREADS overrides the count, read
    once per process): the community renumbering -- chunked sort + pairwise merge of the re-spread, per-thread bitmaps of
    the backbone, slabs of the adjacency scatter -- and the native CSR builder must return the same arrays for 1, 3 and 8
    threads.  A graph of 150,000 nodes, so that the re-spread really runs in several chunks (one per 65,536 nodes)."""
    import hashlib
    import subprocess
    import sys
    code = r"""
import hashlib, sys, torch
sys.path.insert(0, %r)
from gnnadvisor_osdi21_amd import _lib, graph
g = graph.powerlaw_graph(150000, 3000000, 600, locality=0.8, window=700, seed=21)
n = g.num_nodes
rows = torch.repeat_interleave(torch.arange(n), (g.row_pointers[1:] - g.row_pointers[:-1]).long())
perm = torch.randperm(n, generator=torch.Generator().manual_seed(2))
src, dst = perm[rows].to(torch.int32), perm[g.column_index.long()].to(torch.int32)
p = _lib.reorder_community(src, dst, n)
rp, ci = graph.csr_from_edges(src.long(), dst.long(), n)
h = hashlib.sha256()
for t in (p, rp, ci):
    h.update(t.numpy().tobytes())
print("DIGEST", h.hexdigest())
""" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    digests = {}
    for threads in ("1", "3", "8"):
        env = dict(os.environ, GNNA_HOST_THREADS=threads)
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        digests[threads] = [ln for ln in out.stdout.splitlines() if ln.startswith("DIGEST")][-1]
    assert len(set(digests.values())) == 1, digests


# ---- round 6: renumbering from the CSR, native relabelling, node data that follows its node ----------------------
def _scrambled_band(n=6000, seed=4):
    rng = np.random.default_rng(seed)
    u = np.arange(n).repeat(6)
    v = (u + rng.integers(1, 30, u.size)) % n
    perm = rng.permutation(n)
    return perm[u], perm[v]


def test_renumbering_from_the_csr_is_the_renumbering_from_the_edge_list():
    """gnna_reorder_community_csr_i32: a symmetric CSR is used as the adjacency as it stands; a directed one takes the
    edge-list path -- the permutation is gnna_reorder_community_i32's either way."""
    n = 6000
    su, sv = _scrambled_band(n)
    for src, dst in ((np.concatenate([su, sv]), np.concatenate([sv, su])), (su, sv)):      # symmetric, directed
        rp, ci = _lib.csr_from_edges(src, dst, n)
        from_csr = _lib.reorder_community_csr(rp, ci, n).numpy()
        from_edges = _lib.reorder_community(src, dst, n).numpy()
        assert sorted(from_csr.tolist()) == list(range(n))
        assert np.array_equal(from_csr, from_edges)
    bad = ci.clone(); bad[1], bad[0] = ci[0], ci[1]                                        # a row that is not sorted
    if rp[1] >= 2:
        with pytest.raises(_lib.GnnaError, match="increasing"):
            _lib.reorder_community_csr(rp, bad, n)


def test_native_relabelling_is_the_rebuild_from_the_relabelled_edge_list():
    n = 3000
    src, dst = _edges(21, n, 40000)
    rp, ci = _lib.csr_from_edges(src, dst, n)
    new_id = torch.from_numpy(np.random.default_rng(3).permutation(n).astype(np.int32))
    rp2, ci2 = _lib.relabel_csr(rp, ci, new_id, n)
    s2, d2 = new_id.numpy()[src], new_id.numpy()[dst]
    orp, oci = oracle.np_csr_from_edges(s2, d2, n)
    assert np.array_equal(rp2.numpy(), orp) and np.array_equal(ci2.numpy(), oci)
    es, ed = torch.from_numpy(src.astype(np.int32)), torch.from_numpy(dst.astype(np.int32))
    span = _lib.relabel_edges_(es, ed, new_id, n)
    assert np.array_equal(es.numpy(), s2) and np.array_equal(ed.numpy(), d2)
    assert abs(span - np.mean(np.abs(s2.astype(np.int64) - d2))) < 1e-9
    with pytest.raises(_lib.GnnaError, match="permutation"):
        _lib.relabel_csr(rp, ci, torch.zeros(n, dtype=torch.int32), n)


def test_node_data_follows_its_node_when_asked():
    """permute_node_data (set by the mi355x Decider): x / y / masks move with the ids; the reference leaves them (dataset.py:138-172)."""
    n = 2000
    su, sv = _scrambled_band(n)
    src, dst = np.concatenate([su, sv]), np.concatenate([sv, su])
    for permute in (False, True):
        ds = custom_dataset.from_edges(src, dst, n, 5, 3, device="cpu")
        assert ds.edge_index.dtype == np.int32 and len(ds.val) == len(src)
        x0, deg0 = ds.x.clone(), ds.degrees.clone()
        ds.y = torch.arange(n)
        ds.reorder_flag, ds.permute_node_data = True, permute
        ds.rabbit_reorder()
        new_id = torch.from_numpy(ds.new_id)
        assert torch.equal(ds.degrees[new_id], deg0)                       # degrees are rebuilt either way (dataset.py:172)
        if permute:
            assert torch.equal(ds.x[new_id], x0) and torch.equal(ds.y[new_id], torch.arange(n))
        else:
            assert torch.equal(ds.x, x0) and torch.equal(ds.y, torch.arange(n))
        assert ds.avg_edgeSpan_after < 0.2 * ds.avg_edgeSpan and ds.reorder_seconds > 0


def test_loader_refuses_ids_the_int32_csr_cannot_hold():
    with pytest.raises(ValueError, match="num_nodes"):
        custom_dataset.from_edges(np.array([0, 9]), np.array([1, 2]), 5, 4, 2, device="cpu")
