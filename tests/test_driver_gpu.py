"""End-to-end driver (counterpart of GNNA_main.py) on the GPU: flags, printed lines, training."""
import os
import re

import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(capsys, argv):
    from gnnadvisor_osdi21_amd import main as driver
    rc = driver.main(argv)
    out = capsys.readouterr().out
    assert rc == 0
    return out


@pytest.mark.parametrize("model,hidden", [("gcn", 16), ("gin", 64)])
def test_training_prints_the_reference_metric_line(capsys, model, hidden):
    out = _run(capsys, ["--synthetic", "cora-like", "--dim", "96", "--hidden", str(hidden), "--classes", "7",
                        "--model", model, "--num_epoches", "5", "--manual_mode", "False"])
    assert re.search(r"Namespace\(.*dataset='amazon0601',", out)          # 1_log2csv.py:13-16 token
    m = re.search(r"Time \(ms\): (\d+\.\d{3})", out)
    assert m and float(m.group(1)) > 0


def test_verify_and_single_spmm_modes(capsys):
    out = _run(capsys, ["--synthetic", "citeseer-like", "--hidden", "16", "--verify_spmm", "True"])
    assert "# Compute result on GPU" in out and "# Verification PASSED" in out
    out = _run(capsys, ["--synthetic", "citeseer-like", "--hidden", "16", "--single_spmm", "True",
                        "--num_epoches", "20"])
    assert "SpMM profiling size: N: 3327, N: 3327, K: 16" in out
    assert re.search(r"=> SpMM profiling avg \(ms\): \d+\.\d{3}", out)


def test_loss_decreases_with_rabbit_and_auto_decider(capsys):
    """config 1 counterpart (Cora-like GCN, 2 layers, hidden 16): the model actually learns."""
    import torch.nn.functional as F
    from gnnadvisor_osdi21_amd import load_extension
    from gnnadvisor_osdi21_amd.decider import inputProperty
    from gnnadvisor_osdi21_amd.loader import custom_dataset
    from gnnadvisor_osdi21_amd.ops import GCNConv
    GNNA = load_extension()
    torch.manual_seed(0)
    ds = custom_dataset.from_synthetic("cora-like", dim=64, num_class=7, device="cuda")
    ds.y = torch.randint(0, 7, (ds.num_nodes,), device="cuda")
    info = inputProperty(ds.row_pointers, ds.column_index, ds.degrees, 32, 32, 4, 100, hiddenDim=16,
                         dataset_obj=ds, enable_rabbit=True, manual_mode=True)
    x_before, deg_before = ds.x.clone(), ds.degrees.clone()
    info.decider()                                        # manual + rabbit: the renumbered CSR is adopted ...
    assert info.reorder_status and torch.equal(info.row_pointers, ds.row_pointers)
    # ... and under the mi355x policy so are the REBUILT degrees, and the node data moved with the ids (the reference keeps
    # the old ids' degrees on the new rows, GNNA_main.py:70,75 -- `policy="compat"` still does)
    new_id = torch.from_numpy(ds.new_id).cuda()
    assert info.degrees is ds.degrees and torch.equal(ds.degrees[new_id], deg_before) and torch.equal(ds.x[new_id], x_before)
    assert not torch.equal(ds.degrees, deg_before)
    pp, p2n = GNNA.build_part(info.partSize, info.row_pointers)
    rp_host, ci_host = info.row_pointers.clone(), info.column_index.clone()
    info.row_pointers, info.column_index = info.row_pointers.cuda(), info.column_index.cuda()
    info.partPtr, info.part2Node = pp.int().cuda(), p2n.int().cuda()
    c1, c2 = GCNConv(64, 16).cuda(), GCNConv(16, 7).cuda()
    # one GCN layer on the renumbered graph against the oracle on the renumbered graph -- nothing patched by hand
    import numpy as np
    import oracle
    y = c1(ds.x, info.set_input()).detach().cpu().numpy()
    W = c1.weights.detach().cpu().numpy()
    want = oracle.np_forward(ds.x.cpu().numpy(), W, ci_host.numpy(), ds.degrees.cpu().numpy(), pp.numpy(), p2n.numpy())
    scale = oracle.csr_f64(1, np.abs(ds.x.cpu().double().numpy() @ W.astype(np.float64)).astype(np.float32), rp_host.numpy(),
                           ci_host.numpy(), ds.degrees.cpu().numpy())
    assert np.all(np.abs(y - want) <= 1e-4 * np.maximum(1.0, scale)), float(np.max(np.abs(y - want) / np.maximum(1.0, scale)))
    # the reference's coefficient is deg_i*deg_j (a product): scale inputs down to keep it stable
    x = ds.x / (ds.degrees.max() ** 2)
    opt = torch.optim.Adam(list(c1.parameters()) + list(c2.parameters()), lr=0.01)
    losses = []
    for _ in range(40):
        opt.zero_grad()
        h = F.relu(c1(x, info.set_input()))
        out = F.log_softmax(c2(h, info.set_hidden()) / (ds.degrees.max() ** 2), dim=1)
        loss = F.nll_loss(out, ds.y)
        loss.backward()
        opt.step()
        losses.append(float(loss))
    assert losses[-1] < losses[0] and all(torch.isfinite(torch.tensor(losses)))


@pytest.mark.parametrize("model,hidden", [("gcn", 16), ("gin", 64)])
def test_hip_graph_epochs_train_like_eager_epochs(capsys, model, hidden):
    """--hip_graph True records one epoch (forward, backward, Adam) and replays it: same final loss
    as the eager loop from the same seed, same metric line."""
    argv = ["--synthetic", "cora-like", "--dim", "96", "--hidden", str(hidden), "--classes", "7", "--model", model,
            "--num_epoches", "12", "--manual_mode", "False", "--verbose_mode", "True"]
    finals = []
    for graph_flag in ("False", "True"):
        torch.manual_seed(1234)
        out = _run(capsys, argv + ["--hip_graph", graph_flag])
        assert re.search(r"Time \(ms\): (\d+\.\d{3})", out)
        finals.append(float(re.search(r"# final loss: (-?\d+\.\d+|nan|inf)", out).group(1)))
    assert finals[0] == finals[0] and abs(finals[0] - finals[1]) <= 1e-3 * max(1.0, abs(finals[0])), finals


def test_c_abi_consumer_without_python(tmp_path):
    """examples/sag_c_abi.cpp: a host that owns its device memory through the HIP runtime and calls only
    include/gnna.h -- built with hipcc against libgnna.so, run as a separate process."""
    import os
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = str(tmp_path / "sag_c_abi")
    libdir = os.path.join(root, "gnnadvisor_osdi21_amd", "csrc")
    subprocess.run([hipcc, "-O2", "-I", os.path.join(root, "include"), os.path.join(root, "examples", "sag_c_abi.cpp"),
                    "-L", libdir, "-lgnna", "-Wl,-rpath," + libdir, "-o", exe], check=True, timeout=300)
    res = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0 and res.stdout.strip().endswith("OK"), res.stdout + res.stderr


def test_bench_launches_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` from a bare shell (no launcher environment) becomes two ranks by itself and
    prints one SHORT JSON line whose world size is the process group's (gloo + one shared GPU on this box; the
    driver's multi-GPU runs take the same path with RCCL); per-rank and per-leg detail goes to the detail file."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    detail = str(tmp_path / "detail_n2.json")
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--share-gpu",
                          "--steps", "3", "--warmup", "1", "--scale", "0.05", "--config5-leg", "--config5-scale", "0.01",
                          "--detail-file", detail],
                         cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, res.stdout
    assert len(lines[0]) < 8000, len(lines[0])                       # what the driver's 8 KB stdout tail can hold
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["config"]["world_size"] == 2 and rec["verified"] is True
    assert rec["value"] > 0 and rec["config"]["bytes_received_per_rank_per_step"] > 0
    # the three legs of an N-rank run: weak scaling (the headline), strong scaling of the single-GPU graph, config 5
    values = rec["config"]["values"]
    assert set(values) == {"weak", "strong", "config5"}
    assert values["weak"]["value"] == rec["value"] and rec["scaling"] == "weak"
    assert values["strong"]["verified"] and values["config5"]["verified"]
    assert rec["config"]["communicator_ranks_counted"] == 2          # the run checks itself: the communicator counted its ranks
    assert "stream_kernel" in rec["roofline"]["kernel"]
    assert abs(rec["roofline"]["frac"] - rec["roofline"]["achieved"] / rec["roofline"]["peak"]) < 1e-5
    assert "legs" not in rec["config"] and "ranks" not in rec["config"]      # detail, not line
    full = json.load(open(detail))
    legs = full["config"]["legs"]
    assert set(legs) == {"strong", "config5"} and legs["strong"]["verified"] and legs["config5"]["verified"]
    assert legs["strong"]["row_bounds"][0] == 0 and legs["strong"]["row_bounds"][-1] == legs["strong"]["graph_nodes"]
    assert legs["config5"]["dim"] == 128 and legs["config5"]["bytes_received_per_rank_per_step"] > 0
    assert len(full["config"]["ranks"]) == 2
    assert len(full["config"]["exchange_only_ms_per_rank"]) == 2 and len(full["config"]["aggregate_only_ms_per_rank"]) == 2
    assert set(full["roofline"]["per_leg_kernels"]) == {"weak", "strong", "config5"}


def test_drop_in_call_sequence_is_as_fast_as_the_tuned_path():
    """The reference's own call sequence -- build_part(32) and SAG(..., 32, 32, 4), no Decider, no hints, no
    calibration (GNNA_main.py:75-110 in manual mode) -- gets the sliced schedule from the library's own
    statistics and reaches >= 90 % of the Decider-tuned, calibrated configuration bench.py times."""
    if os.environ.get("GNNA_TUNE"):
        pytest.skip("a timing comparison of the default schedules; GNNA_TUNE forces the knobs process-wide")
    import gnnadvisor_osdi21_amd as pkg
    from gnnadvisor_osdi21_amd import _lib, graph
    from gnnadvisor_osdi21_amd.decider import calibrate_phases
    pkg.install_reference_aliases()
    import GNNAdvisor as GNNA
    g = graph.make_config_graph("reddit-like", device="cuda")
    X = torch.randn(g.num_nodes, 64, device="cuda", generator=torch.Generator(device="cuda").manual_seed(2))

    def timed(fn, steps=20):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        _lib.profile_begin(steps)
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        return _lib.profile_end()["main_ms"]

    _lib.reset_tuning()
    _lib.set_graph_hints(None, 0, False)
    pp, p2n = GNNA.build_part(32, g.row_pointers.cpu())
    ppd, p2nd = pp.int().cuda(), p2n.int().cuda()
    t_drop = timed(lambda: GNNA.SAG(X, g.row_pointers, g.column_index, g.degrees, ppd, p2nd, 32, 32, 4))
    assert _lib.last_num_phases() >= 4
    try:
        pp64, p2n64 = _lib.build_part(64, g.row_pointers.cpu())
        pp64, p2n64 = pp64.cuda(), p2n64.cuda()
        _lib.set_tuning(groups_per_chunk=16, loads_in_flight=4)
        calibrate_phases(g.column_index, pp64, p2n64, g.num_nodes, 64, [64])
        out = torch.empty_like(X)
        t_tuned = timed(lambda: _lib.sag(X, g.row_pointers, g.column_index, g.degrees, pp64, p2n64, 64, 32, 4, out=out))
    finally:
        _lib.reset_tuning()
        _lib.set_graph_hints(None, 0, False)
    assert t_tuned / t_drop >= 0.9, (t_drop, t_tuned)


def test_community_renumbering_speeds_up_the_aggregation():
    """f-3 with evidence: a Reddit-sized graph whose edges are 90 % local in a hidden order, ids scrambled (what a
    dataset with community structure and arbitrary ids looks like).  After the loader's rabbit_reorder() hook
    (native community renumbering) the aggregation is clearly faster than on the scrambled ids -- against the
    library's best schedule there (sliced) and, by a wide margin, against the single pass the reference's manual
    mode would run -- and equals the permuted result of the scrambled graph."""
    if os.environ.get("GNNA_TUNE"):
        pytest.skip("a timing comparison of the default schedules; GNNA_TUNE forces the knobs process-wide")
    from gnnadvisor_osdi21_amd import _lib, graph
    dev = torch.device("cuda")
    g = graph.make_config_graph("reddit-like", device=dev, locality=0.9)   # a ring of neighbourhoods (generator's default)
    n, D = g.num_nodes, 64
    rows = torch.repeat_interleave(torch.arange(n, device=dev), (g.row_pointers[1:] - g.row_pointers[:-1]).long())
    perm = torch.randperm(n, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
    src, dst = perm[rows], perm[g.column_index.long()]
    del g, rows

    def run(rp, ci, X, **tune):
        pp, p2n = _lib.build_part(64, rp.cpu())
        ppd, p2nd = pp.to(dev), p2n.to(dev)
        out = torch.empty_like(X)
        _lib.reset_tuning()
        _lib.set_tuning(**tune)
        fn = lambda: _lib.sag(X, rp, ci, None, ppd, p2nd, 64, 32, 4, out=out)
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        _lib.profile_begin(10)
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        ms = _lib.profile_end()["main_ms"]
        _lib.reset_tuning()
        return ms, out

    X = torch.randn(n, D, device=dev, generator=torch.Generator(device=dev).manual_seed(2))
    rp_s, ci_s = graph.csr_from_edges(src, dst, n)
    t_auto, y_s = run(rp_s, ci_s, X)
    t_single, _ = run(rp_s, ci_s, X, column_phases=1)
    new_id = _lib.reorder_community(src.cpu(), dst.cpu(), n).to(dev).long()
    assert torch.equal(torch.sort(new_id).values, torch.arange(n, device=dev))
    rp_r, ci_r = graph.csr_from_edges(new_id[src], new_id[dst], n)
    Xr = torch.empty_like(X)
    Xr[new_id] = X                                           # features follow their nodes
    t_re, y_r = run(rp_r, ci_r, Xr)
    scale = torch.empty_like(X)
    err = (y_r[new_id] - y_s).abs()
    scale = run(rp_s, ci_s, X.abs())[1].clamp(min=1.0)
    assert bool((err <= 1e-4 * scale).all()), float((err / scale).max())
    print(f"# scrambled: {t_auto:.3f} ms (library schedule), {t_single:.3f} ms (single pass); renumbered: {t_re:.3f} ms")
    # (round 3: the sliced schedule on scrambled ids got 15 % faster -- plain id loads, the sweep kernel at this width --
    # so the margin over the library's best there is 1.17 x now, 1.4 x at the start of the round)
    assert t_auto / t_re >= 1.1 and t_single / t_re >= 2.0, (t_auto, t_single, t_re)



def test_default_flow_trains_on_the_renumbered_graph(capsys):
    """SURVEY 8 f-3 through the driver's DEFAULT flow (auto mode, mi355x policy, --enable_rabbit True): products-like with hidden
    locality and scrambled ids, 5-layer GIN, 300 epochs (at the reference's 200 the predicted saving, 4.4 s, is within 5 % of
    the renumbering's 4.2 host seconds: the gate's answer there depends on the box).  The cost gate lets the renumbering run, the kernels
    run on the renumbered CSR, every layer equals the un-renumbered run's under the permutation within 1e-4 x sum|terms|, and
    the epochs are faster than on the ids as they came."""
    if os.environ.get("GNNA_TUNE"):
        pytest.skip("a timing comparison of the default schedules; GNNA_TUNE forces the knobs process-wide")
    from gnnadvisor_osdi21_amd import load_extension
    from gnnadvisor_osdi21_amd import main as driver
    GNNA = load_extension()
    argv = ["--synthetic", "products-like", "--locality", "0.9", "--scramble", "True", "--dim", "100", "--hidden", "64",
            "--classes", "47", "--model", "gin", "--num_epoches", "300", "--manual_mode", "False", "--verbose_mode", "True"]
    runs, times = {}, {}
    for rabbit in ("False", "True"):
        torch.manual_seed(7)
        runs[rabbit] = {}
        assert driver.main(argv + ["--enable_rabbit", rabbit], capture=runs[rabbit]) == 0
        out = capsys.readouterr().out
        times[rabbit] = float(re.search(r"Time \(ms\): (\d+\.\d{3})", out).group(1))
        if rabbit == "True":
            from gnnadvisor_osdi21_amd import _lib
            gate_line = [l for l in out.splitlines() if l.startswith("# renumbering gate:")]
            assert gate_line and "-> renumber" in gate_line[0], (gate_line, runs[rabbit]["inputInfo"].renumbering_decision, _lib.host_threads())
            assert "# renumbered: avg edge span" in out
    a, b = runs["False"], runs["True"]
    ia, ib, da, db = a["inputInfo"], b["inputInfo"], a["dataset"], b["dataset"]
    # (b) the kernels of run B ran on the renumbered CSR: the profile holds it, its statistics and its degrees
    gate = ib.renumbering_decision
    assert gate["go"] and ib.reorder_status and not ia.reorder_status
    assert ib.avgEdgeSpan < 0.2 * ib.avgEdgeSpan_before and abs(ib.avgEdgeSpan_before - ia.avgEdgeSpan) < 1e-6 * ia.avgEdgeSpan
    rows = torch.repeat_interleave(torch.arange(db.num_nodes, device="cuda"), (ib.row_pointers[1:] - ib.row_pointers[:-1]).long())
    span_b = float((rows - ib.column_index.long()).abs().double().mean())
    assert span_b < 0.2 * ia.avgEdgeSpan
    new_id = torch.from_numpy(db.new_id).cuda()
    assert torch.equal(db.x[new_id], da.x) and torch.equal(ib.degrees[new_id], ia.degrees)
    # (a) layer by layer, same weights, same input (A's activations, moved to B's ids): B's output is A's under the permutation
    with torch.no_grad():
        b["model"].load_state_dict(a["model"].state_dict())
        h = da.x
        for i, (ca, cb) in enumerate(zip(a["model"].convs, b["model"].convs)):
            info_a = ia.set_input() if i == 0 else ia.set_hidden()
            info_b = ib.set_input() if i == 0 else ib.set_hidden()
            hb = torch.empty_like(h)
            hb[new_id] = h
            ya, yb = ca(h, info_a, relu=i < 4), cb(hb, info_b, relu=i < 4)
            terms = 0.5 * GNNA.SAG(torch.mm(h.abs(), ca.weights.abs()).contiguous(), ia.row_pointers, ia.column_index, ia.degrees,
                                   ia.partPtr, ia.part2Node, ia.partSize, 32, 4)
            err = (yb[new_id] - ya).abs() / terms.clamp(min=1.0)
            assert float(err.max()) <= 1e-4, (i, float(err.max()))
            h = ya
    # the point of it all: the epochs of the default flow are faster on the renumbered graph
    print(f"# GIN-5 epoch on products-like (hidden locality, scrambled): {times['False']:.2f} ms as loaded, {times['True']:.2f} ms renumbered; "
          f"renumbering took {db.reorder_seconds:.2f} s (gate predicted {gate['reorder_s']:.2f} s, saving {gate['saving_s']:.2f} s)")
    assert times["True"] < 0.85 * times["False"], times


@pytest.mark.parametrize("name,dim,classes", [("cora-like", 1433, 7), ("citeseer-like", 3703, 6)])
def test_configs_1_and_2_at_their_true_widths(capsys, name, dim, classes):
    """BASELINE configs 1-2 through the driver at the datasets' own widths (GNNA_main.py:15-39: Cora F = 1433 / 7 classes,
    Citeseer F = 3703 / 6 classes, hidden 16): the second layer aggregates at D = 7 / 6 -- the widths that are not a multiple of
    four (SURVEY a-2) -- the loss falls, and every layer's output and weight gradient equals the dense fp64 autograd
    formulation within 1e-4 x sum|terms| (the same expression evaluated on |X|, |W|)."""
    import numpy as np
    from util import assert_close_f64
    from gnnadvisor_osdi21_amd import main as driver
    argv = ["--synthetic", name, "--dim", str(dim), "--hidden", "16", "--classes", str(classes), "--model", "gcn",
            "--manual_mode", "False", "--verbose_mode", "True"]
    finals, run = [], {}
    for epochs in (1, 60):
        torch.manual_seed(11)
        run = {}
        assert driver.main(argv + ["--num_epoches", str(epochs)], capture=run) == 0
        out = capsys.readouterr().out
        assert re.search(r"Time \(ms\): (\d+\.\d{3})", out)
        finals.append(float(re.search(r"# final loss: (-?\d+\.\d+|nan|inf)", out).group(1)))
    assert np.isfinite(finals).all() and finals[1] < finals[0], finals
    ds, info, model = run["dataset"], run["inputInfo"], run["model"]
    assert ds.num_features == dim and ds.num_classes == classes and model.conv2.weights.shape == (16, classes)
    n = ds.num_nodes
    rp, ci = info.row_pointers.cpu(), info.column_index.cpu()
    A = torch.zeros(n, n, dtype=torch.float64)
    rows = torch.repeat_interleave(torch.arange(n), (rp[1:] - rp[:-1]).long())
    A[rows, ci.long()] = 1.0
    deg = info.degrees.double().cpu()
    Ahat = A * torch.outer(deg, deg)                                   # the reference's coefficient is the PRODUCT (.cu:355,389)
    for conv, width_in, x in ((model.conv1, dim, ds.x), (model.conv2, 16, torch.randn(n, 16, device="cuda"))):
        layer_info = info.set_input() if conv is model.conv1 else info.set_hidden()
        W = conv.weights.detach()
        conv.weights.grad = None
        xg = x.detach().clone().requires_grad_(True)
        y = conv(xg, layer_info)
        y.square().sum().backward()
        res = {}
        for tag, f in (("ref", lambda t: t), ("abs", torch.abs)):
            Xr = f(x.detach().double().cpu()).clone().requires_grad_(True)
            Wr = f(W.double().cpu()).clone().requires_grad_(True)
            o = Ahat @ (Xr @ Wr)
            o.square().sum().backward()
            res[tag] = (o.detach().numpy(), Xr.grad.numpy(), Wr.grad.numpy())
        for got, k, what in ((y, 0, "out"), (xg.grad, 1, "dX"), (conv.weights.grad, 2, "dW")):
            assert_close_f64(got.detach().cpu().numpy(), res["ref"][k], what=f"{name} layer {width_in}->{W.shape[1]} {what}",
                             scale=res["abs"][k])
