"""Host-side logic and the C-ABI surface, without a GPU: the library loads and exports
every symbol include/gnna.h declares, the partitioner is bit-exact with the oracle and
the reference goldens, the Decider's compat policy reproduces the reference's param.py,
and the extension rejects CPU tensors with the reference's error text."""
import ctypes
import json
import os
import re

import numpy as np
import pytest
import torch

import oracle
from gnnadvisor_osdi21_amd import _lib, decider, graph, load_extension

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "gnna.h")).read()
    declared = set(re.findall(r"GNNA_API\s+[\w\s\*]+?\b(gnna_\w+)\s*\(", header))
    assert declared, "no GNNA_API declarations found"
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in gnna.h but not exported by libgnna.so"
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    assert lib.gnna_version() == 600


def test_binaries_carry_the_hash_of_the_sources_beside_them():
    """gnna_build_id() / GNNAdvisor.build_id(): the loaded binaries were compiled from this tree's sources."""
    from gnnadvisor_osdi21_amd import build as gbuild
    want = gbuild.source_hash()
    assert re.fullmatch(r"[0-9a-f]{16}", want)
    assert _lib.build_id() == "0.6.0+" + want
    assert load_extension().build_id() == f"module {want}, library 0.6.0+{want}"


def test_set_tuning_refuses_another_struct_layout():
    """gnna_tuning.struct_size: a caller built against a header with more / fewer knobs is refused instead of
    having its fields silently shifted."""
    import ctypes
    lib = _lib.load()
    t = _lib.Tuning(*([ctypes.sizeof(_lib.Tuning) - 4] + [-1] * (len(_lib.Tuning._fields_) - 1)))
    before = _lib.get_tuning()
    assert lib.gnna_set_tuning(ctypes.byref(t)) == -1 and b"struct_size" in lib.gnna_last_error()
    assert _lib.get_tuning() == before
    _lib.set_tuning(column_phases=5)
    assert _lib.get_tuning()["column_phases"] == 5
    _lib.reset_tuning()
    assert _lib.get_tuning() == before


def test_build_part_c_abi_bit_exact_vs_oracle_and_golden(golden_dir):
    cases = json.load(open(os.path.join(golden_dir, "build_part.json")))["cases"]
    for c in cases:
        indptr = torch.tensor(c["indptr"], dtype=torch.int32)
        pp, p2n = _lib.build_part(c["partSize"], indptr)
        opp, op2n = oracle.build_part(c["partSize"], indptr.numpy())
        assert pp.dtype == torch.int32 and p2n.dtype == torch.int32
        assert np.array_equal(pp.numpy(), opp) and np.array_equal(p2n.numpy(), op2n), c["name"]
        assert p2n.tolist() == [int(v) for v in c["part2Node"]], c["name"]   # reference golden
    rng = np.random.default_rng(1)
    for _ in range(10):
        deg = rng.integers(0, 200, size=int(rng.integers(1, 400)))
        indptr = torch.tensor(np.concatenate([[0], np.cumsum(deg)]), dtype=torch.int32)
        ps = int(rng.integers(1, 70))
        pp, p2n = _lib.build_part(ps, indptr)
        opp, op2n = oracle.build_part(ps, indptr.numpy())
        assert np.array_equal(pp.numpy(), opp) and np.array_equal(p2n.numpy(), op2n)
        # every group is non-empty, <= ps edges, and groups tile each row exactly
        sizes = np.diff(pp.numpy())
        assert (sizes > 0).all() and (sizes <= ps).all()
        assert np.array_equal(np.bincount(p2n.numpy(), weights=sizes, minlength=len(deg)), deg)


def test_build_part_errors():
    with pytest.raises(_lib.GnnaError):
        _lib.build_part(0, torch.tensor([0, 1], dtype=torch.int32))
    with pytest.raises(_lib.GnnaError):
        _lib.build_part(4, torch.tensor([0, 5, 3], dtype=torch.int32))   # decreasing indptr


def test_extension_module_surface_and_errors():
    GNNA = load_extension()
    for fn in ("SAG", "forward", "backward", "forward_gin", "backward_gin", "build_part"):
        assert callable(getattr(GNNA, fn))
    rp = torch.tensor([0, 3, 3, 8, 9, 9], dtype=torch.int32)
    pp, p2n = GNNA.build_part(2, rp)
    assert pp.tolist() == [0, 2, 3, 5, 7, 8, 9] and p2n.tolist() == [0, 0, 2, 2, 2, 3]
    assert pp.int() is pp or torch.equal(pp.int(), pp)        # caller's .int() (GNNA_main.py:109) is a no-op
    X = torch.ones(5, 4)
    ci = torch.zeros(9, dtype=torch.int32)
    deg = torch.ones(5)
    with pytest.raises(RuntimeError, match="input must be a CUDA tensor"):
        GNNA.SAG(X, rp, ci, deg, pp, p2n, 2, 32, 4)
    with pytest.raises(RuntimeError, match="input must be a CUDA tensor"):
        GNNA.forward(X, torch.ones(4, 2), rp, ci, deg, pp, p2n, 2, 32, 4)
    with pytest.raises(RuntimeError, match="d_output must be a CUDA tensor"):
        GNNA.backward(X, X, torch.ones(4, 2), rp, ci, deg, pp, p2n, 2, 32, 4)
    with pytest.raises(RuntimeError, match="input must be a CUDA tensor"):
        GNNA.forward_gin(X, torch.ones(4, 2), rp, ci, 0.5, pp, p2n, 2, 32, 4)
    with pytest.raises(RuntimeError):
        GNNA.build_part(2, rp.long())                            # wrong dtype
    with pytest.raises(TypeError):
        GNNA.SAG(X, rp, ci, deg, pp, p2n, 2, 32)                 # positional arity as in the reference


def test_build_part_float_compat_mode():
    """build_part(..., float_compat=True): the reference's float32 return dtype (GNNAdvisor.cpp:229-230) for callers
    that depend on it -- same values as the int32 result (sentinel always written: bug A fixed), refused when
    float32 cannot hold the offsets (bug B flagged instead of silently misplacing groups)."""
    GNNA = load_extension()
    rp = torch.tensor([0, 3, 3, 8, 9, 9], dtype=torch.int32)          # last node isolated: the reference drops the sentinel
    pp_i, p2n_i = GNNA.build_part(2, rp)
    pp_f, p2n_f = GNNA.build_part(2, rp, float_compat=True)
    assert pp_f.dtype == torch.float32 and p2n_f.dtype == torch.float32
    assert torch.equal(pp_f.int(), pp_i) and torch.equal(p2n_f.int(), p2n_i)
    assert pp_f.tolist() == [0, 2, 3, 5, 7, 8, 9]                      # reference: [..., 8, 0]
    big = torch.tensor([0, 20000001, 40000003], dtype=torch.int32)    # SURVEY a-6 bug B probe
    with pytest.raises(RuntimeError, match="float32"):
        GNNA.build_part(32, big, float_compat=True)
    assert GNNA.build_part(32, big)[0][-1].item() == 40000003


def test_product_has_no_cpu_path():
    g = graph.uniform_graph(10, 40, seed=1)
    pp, p2n = _lib.build_part(4, g.row_pointers)
    with pytest.raises(_lib.GnnaError, match="no CPU path"):
        _lib.sag(torch.ones(10, 8), g.row_pointers, g.column_index, g.degrees, pp, p2n, 4, 32, 4)


def test_tuning_roundtrip():
    try:
        _lib.set_tuning(8, 4, 2, 0, 1)
        assert _lib.get_tuning() == dict(groups_per_chunk=8, loads_in_flight=4, blocks_per_cu=2,
                                         xcd_remap=0, trust_canonical=1, column_phases=0, avg_degree=0,
                                         nonlocal_ids=0, gcn_prescale=0, pad_rows=0, zero_fill=0,
                                         sweep=0, sweep_slack=0, deterministic=0, pack_ids=0, ids_check_every=64, wide_blocks=0)
        _lib.set_tuning(column_phases=8)
        assert _lib.get_tuning()["column_phases"] == 8
        _lib.set_tuning(groups_per_chunk=32)      # others keep their values
        assert _lib.get_tuning()["groups_per_chunk"] == 32 and _lib.get_tuning()["loads_in_flight"] == 4
    finally:
        _lib.reset_tuning()
    assert _lib.get_tuning()["groups_per_chunk"] == 16


class _DS:
    def __init__(self, c):
        self.num_nodes = c["num_nodes"]; self.avg_degree = c["num_edges"] / c["num_nodes"]
        self.avg_edgeSpan = c["avg_edgeSpan"]; self.num_features = c["input_dim"]
        self.reorder_flag = False; self.reorder_calls = 0
        self.row_pointers = "rp_after_reorder"; self.column_index = "ci_after_reorder"

    def rabbit_reorder(self):
        self.reorder_calls += 1


def test_decider_compat_policy_reproduces_reference_goldens(golden_dir):
    for c in json.load(open(os.path.join(golden_dir, "decider.json")))["cases"]:
        ds = _DS(c)
        ip = decider.inputProperty("rp", "ci", "deg", 32, 32, 4, c["sharedMem"], hiddenDim=c["hidden"],
                                   dataset_obj=ds, enable_rabbit=c.get("enable_rabbit", True),
                                   manual_mode=(c["mode"] == "manual"), policy="compat")
        ip.decider()
        e = c["expect"]
        assert ip.partSize == e["partSize"], c["name"]
        assert bool(ip.reorder_status) == e["reorder"], c["name"]
        for k in ("dimWorker_input", "warpPerBlock_input", "dimWorker_hidden", "warpPerBlock_hidden",
                  "row_pointers", "column_index", "dimWorker", "warpPerBlock"):
            if k in e:
                assert getattr(ip, k) == e[k], (c["name"], k)
        for k in ("reorder_flag", "reorder_calls"):
            if k in e:
                assert getattr(ds, k) == e[k], (c["name"], k)
        if "after_set_input" in e:
            r = ip.set_input()
            assert r is ip and [ip.dimWorker, ip.warpPerBlock, ip.state_set_input] == e["after_set_input"]
            r = ip.set_hidden()
            assert r is ip and [ip.dimWorker, ip.warpPerBlock, ip.state_set_input] == e["after_set_hidden"]


def test_decider_mi355x_policy_is_sane():
    with pytest.raises(ValueError):
        decider.inputProperty(dataset_obj=None)
    for n, e, f, h in [(2708, 10556, 1433, 16), (232965, 114615892, 602, 64), (2449029, 123718280, 100, 64),
                       (111059956, 1615685872, 128, 128), (1000, 100, 8, 3)]:
        c = dict(num_nodes=n, num_edges=e, avg_edgeSpan=n / 3, input_dim=f)
        ip = decider.inputProperty(None, None, None, 32, 32, 4, 100, hiddenDim=h, dataset_obj=_DS(c),
                                   manual_mode=False)
        ip.decider()
        assert 16 <= ip.partSize <= 128 and ip.partSize & (ip.partSize - 1) == 0
        assert ip.dimWorker_hidden == decider.lanes_per_row(h) and 4 <= ip.dimWorker_hidden <= 64
        assert ip.warpPerBlock_hidden == 4 and 1 <= ip.groups_per_chunk <= 32
        assert ip.avg_degree_hint == max(1, int(e / n)) and ip.nonlocal_ids_hint == 1   # span n/3: scattered ids
    local = dict(num_nodes=100000, num_edges=5000000, avg_edgeSpan=300.0, input_dim=64)
    ip = decider.inputProperty(None, None, None, 32, 32, 4, 100, hiddenDim=64, dataset_obj=_DS(local), manual_mode=False)
    ip.decider()
    assert ip.nonlocal_ids_hint == 0            # community-ordered ids: never phase
    assert decider.lanes_per_row(64) == 16 and decider.lanes_per_row(16) == 4
    assert decider.lanes_per_row(41) == 64 and decider.lanes_per_row(100) == 32
    assert decider.choose_part_size(492, 64) == 128 and decider.choose_part_size(50.5, 64) == 64 and decider.choose_part_size(3.9, 16) == 16


def test_graph_builders_follow_the_reference_loader():
    # SURVEY 8c probe: src=[0,0,0,2,2,1], dst=[3,1,3,0,0,2] -> indptr [0,2,3,4,4], indices [1,3,2,0]
    g = graph.graph_from_edges(torch.tensor([0, 0, 0, 2, 2, 1]), torch.tensor([3, 1, 3, 0, 0, 2]), 4)
    assert g.row_pointers.tolist() == [0, 2, 3, 4, 4] and g.column_index.tolist() == [1, 3, 2, 0]
    assert g.num_edges_raw == 6 and g.avg_degree == 1.5
    np.testing.assert_allclose(g.degrees.numpy(), np.sqrt([2, 1, 1, 1]).astype(np.float32))
    rng = np.random.default_rng(3)
    src, dst = rng.integers(0, 50, 600), rng.integers(0, 50, 600)
    g = graph.graph_from_edges(torch.tensor(src), torch.tensor(dst), 50)
    rp, ci = oracle.np_csr_from_edges(src, dst, 50)
    assert np.array_equal(g.row_pointers.numpy(), rp) and np.array_equal(g.column_index.numpy(), ci)
    np.testing.assert_array_equal(g.degrees.numpy(), oracle.np_degrees(rp))
    assert abs(g.avg_edgeSpan - np.mean(np.abs(src - dst))) < 1e-9
    # seeded generators are deterministic, symmetric and loop-free
    a = graph.powerlaw_graph(500, 8000, 120, seed=9)
    b = graph.powerlaw_graph(500, 8000, 120, seed=9)
    assert torch.equal(a.column_index, b.column_index) and torch.equal(a.row_pointers, b.row_pointers)
    rows = torch.repeat_interleave(torch.arange(500), (a.row_pointers[1:] - a.row_pointers[:-1]).long())
    assert not bool((rows == a.column_index).any())
    fwd = set(zip(rows.tolist(), a.column_index.tolist()))
    assert all((c, r) in fwd for r, c in fwd)
    assert int((a.row_pointers[1:] - a.row_pointers[:-1]).max()) <= 2 * 120 + 20


def test_c_abi_example_compiles_and_links(tmp_path):
    """examples/sag_c_abi.cpp builds against include/gnna.h + libgnna.so with hipcc (no GPU needed to link)."""
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    libdir = os.path.join(root, "gnnadvisor_osdi21_amd", "csrc")
    exe = str(tmp_path / "sag_c_abi")
    subprocess.run([hipcc, "-O1", "-I", os.path.join(root, "include"), os.path.join(root, "examples", "sag_c_abi.cpp"),
                    "-L", libdir, "-lgnna", "-Wl,-rpath," + libdir, "-o", exe], check=True, timeout=300)
    assert os.path.getsize(exe) > 0


# ---- round 6: the mi355x policy fixes the reference's renumbering quirks, compat keeps them ------------------
class _ReorderDS:
    """A dataset double whose rabbit_reorder() 'rebuilds' CSR and degrees the way loader.custom_dataset does."""

    def __init__(self, num_nodes, num_edges, span):
        self.num_nodes, self.num_edges, self.num_features = num_nodes, num_edges, 64
        self.avg_degree, self.avg_edgeSpan = num_edges / num_nodes, span
        self.row_pointers, self.column_index, self.degrees = "rp-old", "ci-old", "deg-old"
        self.reorder_flag, self.permute_node_data, self.reorder_calls, self.renumbered = False, False, 0, False

    def rabbit_reorder(self):
        self.reorder_calls += 1
        if self.reorder_flag:
            self.renumbered = True
            self.row_pointers, self.column_index, self.degrees = "rp-new", "ci-new", "deg-new"
            self.avg_edgeSpan_after = self.avg_edgeSpan / 10


def _ip(ds, policy, manual, **kw):
    ip = decider.inputProperty(ds.row_pointers, ds.column_index, ds.degrees, 32, 32, 4, 100, hiddenDim=64, dataset_obj=ds,
                               enable_rabbit=True, manual_mode=manual, policy=policy)
    for k, v in kw.items():
        setattr(ip, k, v)
    ip.decider()
    return ip


def test_mi355x_policy_adopts_the_renumbered_graph_and_compat_keeps_the_reference_quirks(monkeypatch):
    monkeypatch.setenv("GNNA_HOST_THREADS", "16")           # the gate prices the renumbering for the host it runs on
    big = dict(num_nodes=2449029, num_edges=123718280, span=2449029 / 3)
    # manual mode: the reference adopts the CSR and keeps the OLD degrees (param.py:59-64, GNNA_main.py:70,75)
    ds = _ReorderDS(**big); ip = _ip(ds, "compat", True)
    assert (ip.row_pointers, ip.column_index, ip.degrees) == ("rp-new", "ci-new", "deg-old") and not ds.permute_node_data
    ds = _ReorderDS(**big); ip = _ip(ds, "mi355x", True)
    assert (ip.row_pointers, ip.column_index, ip.degrees) == ("rp-new", "ci-new", "deg-new") and ds.permute_node_data
    assert ip.avgEdgeSpan == ds.avg_edgeSpan_after and ip.avgEdgeSpan_before == big["span"]
    # auto mode: the reference renumbers and then runs on the ORIGINAL CSR (param.py:108-117)
    ds = _ReorderDS(**big); ip = _ip(ds, "compat", False)
    assert ds.renumbered and (ip.row_pointers, ip.column_index, ip.degrees) == ("rp-old", "ci-old", "deg-old")
    # mi355x: a long run on an HBM-resident, scattered graph pays for the renumbering -> adopted, hints refreshed
    ds = _ReorderDS(**big)
    ip = _ip(ds, "mi355x", False, expected_aggregations=decider.expected_aggregations("gin", 100, 64, 47, 210))
    gate = ip.renumbering_decision
    assert gate["go"] and gate["saving_s"] > gate["reorder_s"] > 0 and gate["aggregations"] == 2100 - 210
    assert ds.renumbered and ds.permute_node_data and ip.reorder_status
    assert (ip.row_pointers, ip.column_index, ip.degrees) == ("rp-new", "ci-new", "deg-new") and ip.nonlocal_ids_hint == 0
    # ... ten single aggregations do not -> the ids stay, nothing is renumbered, unless the caller insists
    ds = _ReorderDS(**big); ip = _ip(ds, "mi355x", False, expected_aggregations=[(64, 10)])
    assert not ip.renumbering_decision["go"] and not ds.renumbered and not ip.reorder_status and ip.row_pointers == "rp-old"
    ds = _ReorderDS(**big); ip = _ip(ds, "mi355x", False, expected_aggregations=[(64, 10)], force_renumbering=True)
    assert ds.renumbered and ip.row_pointers == "rp-new"
    # a graph whose ids are local already fails the reference's own precondition (param.py:108): no gate, no renumbering
    ds = _ReorderDS(num_nodes=1000000, num_edges=50000000, span=50.0); ip = _ip(ds, "mi355x", False)
    assert ip.renumbering_decision is None and not ds.renumbered


def test_renumbering_gate_numbers():
    # Reddit-like (matrix fits the Infinity Cache): 0.12 ms saved per aggregation at D = 64 -- a 210-epoch 2-layer GCN does not
    # win 2+ s of host time back; products-like (HBM resident): ~2 ms per aggregation -- it does
    aggs = decider.expected_aggregations("gcn", 602, 64, 41, 210)
    assert aggs == [(64, 420), (41, 420)]
    small = decider.renumbering_gate(232965, 114615892, 232965 / 3, aggs, threads=16)
    assert not small["go"] and 0.05e-3 < small["saving_per_aggregation_s"] < 0.3e-3
    big = decider.renumbering_gate(2449029, 123718280, 2449029 / 3, decider.expected_aggregations("gin", 100, 64, 47, 210), threads=16)
    assert big["go"] and 1e-3 < big["saving_per_aggregation_s"] < 4e-3 and big["reorder_s"] < 8
    half = decider.renumbering_gate(2449029, 123718280, 2449029 / 6, [(64, 100)], threads=16)
    assert abs(half["scatter"] - 0.5) < 1e-9
    # GIN's first layer aggregates once per epoch at the narrower of (input, update-first output) widths
    assert decider.expected_aggregations("gin", 602, 64, 41, 10)[0] == (64, 20)
    assert decider.expected_aggregations("gin", 100, 64, 47, 10)[0] == (100, 10)


def test_planner_inputs_come_from_the_library():
    assert _lib.device_cus() == 0 or _lib.device_cus() >= 8          # no device here: 0, never an error
    assert decider.num_cus() in (256, _lib.device_cus())
    os.environ["GNNA_HOST_THREADS"] = "3"
    try:
        assert _lib.host_threads() == 3                                # read at every call
    finally:
        del os.environ["GNNA_HOST_THREADS"]
    assert 1 <= _lib.host_threads() <= 64
