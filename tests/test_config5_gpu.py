"""BASELINE config 5 (papers100M GCN hidden = 128, destination-partitioned over 8 MI355X) in its TRUE per-rank shape, on
one GPU: rank 0 of 8 aggregates a rectangular [13.9 M x 111 M] shard -- from the 56.9 GB all-gather buffer (64-bit row
offsets into the sources, 7.1 GB of output) and from the compact halo buffer of the automatic exchange.  One process
plays the rank (`ShardedAggregator(emulate=(rank, world))`, CPU-tested against the whole-graph result in
tests/test_dist.py); the receive buffer is filled from the global features instead of by RCCL.

Checks (bench.RankOf8Workload.verify, the same code the bench line's `verified` comes from): X = ones on every rank
-> exact row nnz; 200 sampled rows of the randn run against an fp64 gather-sum over the GLOBAL features and the shard's
GLOBAL column ids, bound 1e-4 * sum |terms|; 64 rows of the degree-weighted form likewise.
"""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import bench  # noqa: E402

pytestmark = pytest.mark.gpu


def _run(form, scale, samples=200):
    dev = torch.device("cuda", 0)
    w = bench.RankOf8Workload(dev, 128, form=form, scale=scale)
    try:
        chk = w.verify(samples)
        desc = w.describe()
    finally:
        w.release()
        torch.cuda.empty_cache()
    assert chk["ones_exact"], (form, chk)
    assert chk["sampled_rows_ok"] and chk["gcn_weighted_rows_ok"], (form, chk)
    return chk, desc


def test_config5_rank_of_8_from_the_all_gather_buffer_true_shape():
    chk, d = _run("allgather-one-call", 1.0)
    assert d["rows_per_rank"] == 111059956 // 8 and d["source_rows_all_ranks"] == 8 * d["rows_per_rank"]
    assert d["wide_offsets"] and d["source_buffer_GB"] > 56.0          # 64-bit offsets into the sources ...
    assert d["rows_per_rank"] * 128 * 4 > 2 ** 32                        # ... and a > 4 GiB output
    assert d["nnz"] > 1.9e8


def test_config5_rank_of_8_halo_exchange_true_shape():
    chk, d = _run("halo", 1.0)
    assert d["exchange"] == "halo" and d["pieces"] >= 1
    assert sum(d["halo_rows_per_peer"]) == d["halo_rows"] and d["halo_rows_per_peer"][0] == 0
    assert d["wide_offsets"]                                             # the compact buffer is > 4 GiB too
    assert d["bytes_received_per_step"] < 0.7 * d["allgather_bytes_received_per_step"]


def test_config5_rank_of_8_pipelined_all_gather_layout_small():
    """The sub-block-major all-gather layout (K pieces) at 1/20 of the size."""
    chk, d = _run("allgather-pieces", 0.05, samples=64)
    assert d["exchange"] == "allgather"
