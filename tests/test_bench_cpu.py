"""bench.py host logic that needs no GPU: the driver's contract (defaults, the self-launch command for
`--gpus N`), the byte models of SURVEY.md 8(d), and the bookkeeping that turns a rocprofv3 counter CSV into
per-step traffic (dispatch order, calibration copies)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_defaults_follow_the_driver_contract():
    a = bench.parse_args([])
    assert a.gpus == 1 and a.steps > 0 and a.warmup > 0 and a.dim == 64 and a.config == "reddit-like"
    a = bench.parse_args(["--gpus", "8", "--steps", "5", "--warmup", "2"])
    assert (a.gpus, a.steps, a.warmup) == (8, 5, 2) and a.backend == "nccl" and a.exchange == "auto"
    # N-rank line: the weak-scaling headline plus the strong-scaling and (at 8 ranks) config-5 legs by default
    assert a.scaling == "strong,config5" and not a.config5_leg and a.config5_scale == 1.0


def test_kernel_labels_name_what_ran():
    class W:
        launches, phases = 1, 8
    assert "sliced schedule" in bench.kernel_label(W)
    W.phases = 1
    assert bench.kernel_label(W) == "stream_kernel"
    W.calls = 5
    assert "5 library calls" in bench.kernel_label(W)
    W.calls, W.launches = 1, 4
    assert "4 launches" in bench.kernel_label(W)


def test_gpus_n_launches_n_ranks_on_localhost(monkeypatch):
    seen = {}

    def fake_execv(path, argv):
        seen["path"], seen["argv"] = path, list(argv)
        raise SystemExit(0)
    monkeypatch.setattr(os, "execv", fake_execv)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "7", "--warmup", "3"])
    with pytest.raises(SystemExit):
        bench.self_launch(4)
    argv = seen["argv"]
    assert seen["path"] == sys.executable and argv[1:3] == ["-m", "torch.distributed.run"]
    assert "--nnodes=1" in argv and "--nproc-per-node=4" in argv
    assert argv[argv.index("--master-addr") + 1] == "127.0.0.1"
    assert 1024 < int(argv[argv.index("--master-port") + 1]) < 65536
    tail = argv[argv.index(os.path.join(ROOT, "bench.py")) + 1:]
    assert tail == ["--gpus", "4", "--steps", "7", "--warmup", "3"]      # the ranks get the caller's flags unchanged


def test_byte_models():
    nnz, n, parts, dim = 114623790, 232965, 1901647, 64
    assert bench.gather_model_bytes(nnz, n, parts, dim) == nnz * 260 + n * 260 + parts * 8 == 29877969476
    assert bench.compulsory_bytes(nnz, n, n, dim) == nnz * 4 + (n + 1) * 4 + 2 * n * dim * 4


def test_counter_rows_are_split_by_workload_and_calibrated_on_the_last_copies():
    manifest = {"workloads": [{"warmup": 1, "steps": 3, "launches_per_step": 1, "phases": 8},
                              {"warmup": 1, "steps": 3, "launches_per_step": 2, "phases": 2}],
                "calib_copies": 3}
    rows = []

    def add(kernel, value):
        rows.append({"Dispatch_Id": str(len(rows) + 1), "Kernel_Name": kernel, "Counter_Name": "FETCH_SIZE",
                     "Counter_Value": str(value)})
    add("__amd_rocclr_copyBuffer", 7.0)                       # host-to-device transfers of the set-up: not calibration
    add("slice_count_kernel", 999.0)
    for v in (50.0, 100.0, 100.0, 100.0):                     # workload 0: warm-up + 3 steps, one launch each
        add("stream_kernel", v)
    add("__amd_rocclr_copyBuffer", 9.0)
    for v in (1.0, 1.0, 10.0, 20.0, 10.0, 20.0, 10.0, 20.0):  # workload 1: two launches per step
        add("sweep_kernel", v)
    for _ in range(3):
        add("__amd_rocclr_copyBuffer", 524288.0)              # the 1 GiB calibration copies (KiB, half-counted)
    per_step, copies = bench.split_counters(manifest, rows, "FETCH_SIZE")
    assert per_step == [100.0, 30.0]
    assert copies == [524288.0] * 3
    assert bench.CALIB_BYTES / (1024.0 * sum(copies) / len(copies)) == 2.0
    # a child whose dispatch count does not match the schedule is not trusted
    short, _ = bench.split_counters(manifest, rows[:-6], "FETCH_SIZE")
    assert short[1] is None


def test_roofline_frac_is_the_share_of_the_binding_ceiling():
    """`roofline.frac` = max(measured fabric bytes / t / 8 TB/s, L2 requests x 128 B / t / 34.5 TB/s).  It must rise when
    the same schedule runs faster, must not fall because a schedule moves fewer fabric bytes for the same requests, and
    round 3's record (sweep kernel: 1.3858 ms, 3.89 GB, 229.2 M requests) re-derives to 0.61, round 2's (1.586 ms, 7.9 GB,
    241.5 M requests) to 0.62 with the fabric binding."""
    r3 = bench.binding_shares(3890224405.36, 229.2e6, 1.3858e-3)
    assert r3["binding"] == "l2" and abs(r3["frac"] - 0.6136) < 1e-3 and abs(r3["frac_hbm_measured"] - 0.3509) < 1e-3
    assert r3["frac"] == r3["achieved_binding"] / r3["peak_binding"] == r3["frac_l2"]
    r2 = bench.binding_shares(7.9e9, 241.5e6, 1.586e-3)
    assert r2["binding"] == "fabric" and abs(r2["frac"] - 0.6226) < 1e-3
    # faster on the same schedule (same counters) -> larger frac
    assert bench.binding_shares(3890224405.36, 229.2e6, 1.25e-3)["frac"] > r3["frac"]
    # fewer fabric bytes for the same requests and time -> not smaller
    assert bench.binding_shares(2.0e9, 229.2e6, 1.3858e-3)["frac"] == r3["frac"]
    # no request counter: the fabric share alone
    only = bench.binding_shares(25.4e9, None, 3.45e-3)
    assert only["binding"] == "fabric" and only["frac_l2"] is None and abs(only["frac"] - 25.4e9 / 3.45e-3 / 8e12) < 1e-9


def test_roofline_record_carries_flat_keys_for_the_drivers_parser():
    class G:
        nnz, num_nodes = 114623790, 232965

    class W:
        g, P, dim, launches, phases, swept = G, 1901647, 64, 1, 16, True
    traffic = {"bytes_per_step": 3890224405.36, "l2_requests_per_step": 229.2e6, "l2_hit_rate": 0.8646}
    rec = bench.roofline_record(W, 1.3858, 0.042, traffic, "l2-fabric")
    assert rec["bound"] == "hbm" and rec["peak"] == 8000.0 and rec["unit"] == "GB/s"
    assert abs(rec["frac"] - 0.6136) < 1e-3 and rec["binding"] == "l2"
    for key in ("frac_hbm_measured", "frac_l2", "frac_gather_model_of_hbm", "achieved_binding", "peak_binding",
                "l2_requests_per_edge", "traffic", "achieved"):
        assert isinstance(rec[key], float), key
    assert abs(rec["frac_gather_model_of_hbm"] - 29877969476 / 1.3858e-3 / 8e12) < 1e-6      # 2.69: not a fraction of HBM
    assert abs(rec["achieved"] - 3890224405.36 / 1.3858e-3 / 1e9) < 1e-6                      # the measured fabric rate stays
    assert abs(rec["l2_requests_per_edge"] - 2.0) < 0.01
    # without counters nothing is invented
    rec0 = bench.roofline_record(W, 1.3858, 0.042, {"error": "rocprofv3 not found"}, "l2-fabric")
    assert rec0["traffic"] is None and rec0["frac_l2"] is None and "compulsory" in rec0["achieved_source"]
