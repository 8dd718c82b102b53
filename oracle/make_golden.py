"""Generate tests/golden/*.json from the REFERENCE itself (run in the build container,
where /root/reference exists; the fixtures are committed, the reference never travels).

* build_part.json  -- outputs of the reference's own C++ ``build_part``
                      (GNNAdvisor/GNNConv/GNNAdvisor.cpp:210-251, compiled by
                      oracle/build_ref.py) on fixed and seeded inputs.
* decider.json     -- outputs of the reference's ``inputProperty.decider()``
                      (GNNAdvisor/param.py:51-120), imported from /root/reference with a
                      duck-typed dataset object, plus set_input()/set_hidden() swaps.
* kat_ones.json    -- the reference's known-answer test inputs (unitest.py:27: X = ones;
                      expected = per-row nnz, unitest.py:33-40,54-63) on seeded graphs whose
                      partition comes from the reference build_part.

Fixtures are data only (inputs + expected outputs).
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)

from oracle import build_ref, np_csr_from_edges  # noqa: E402


def _rand_indptr(rng, n, max_deg, p_zero):
    deg = rng.integers(0, max_deg + 1, size=n)
    deg[rng.random(n) < p_zero] = 0
    return np.concatenate([[0], np.cumsum(deg)]).astype(np.int32)


def gen_build_part(ref):
    cases = []

    def add(name, ps, indptr):
        pp, p2n = ref.build_part(int(ps), torch.IntTensor(np.asarray(indptr, dtype=np.int32)))
        assert pp.dtype == torch.float32 and p2n.dtype == torch.float32
        cases.append(dict(name=name, partSize=int(ps), indptr=[int(v) for v in indptr],
                          partPtr=[float(v) for v in pp.tolist()],
                          part2Node=[float(v) for v in p2n.tolist()]))

    add("survey_a", 2, [0, 3, 3, 8, 9])
    for ps in (1, 2, 3, 32):
        add(f"survey_bugA_ps{ps}", ps, [0, 3, 3, 8, 9, 9])
    add("appendixA", 2, [0, 3, 4, 6, 7])
    add("all_zero_degree", 4, [0, 0, 0, 0])
    add("single_node_empty", 4, [0, 0])
    add("single_node", 4, [0, 9])
    add("exact_multiple", 4, [0, 4, 12, 12, 16])
    add("ps_larger_than_any", 1000, [0, 5, 7, 30, 31])
    add("float32_inexact_bugB", 20000000, [0, 20000001, 40000003])
    rng = np.random.default_rng(20210714)
    for k in range(8):
        n = int(rng.integers(1, 60))
        ps = int(rng.choice([1, 2, 3, 4, 7, 8, 16, 32, 64]))
        add(f"seeded_{k}", ps, _rand_indptr(rng, n, 50, 0.3))
    return cases


def gen_decider():
    sys.path.insert(0, "/root/reference/GNNAdvisor")
    import param as ref_param  # the reference's param.py (only imports math)

    class DS:  # duck-typed dataset_obj (param.py:17-36 reads these fields)
        def __init__(self, n, e, f, span):
            self.num_nodes, self.avg_degree, self.avg_edgeSpan = n, e / n, span
            self.num_features = f
            self.reorder_flag = False
            self.row_pointers = "rp_after_reorder"
            self.column_index = "ci_after_reorder"
            self.reorder_calls = 0

        def rabbit_reorder(self):
            self.reorder_calls += 1

    graphs = dict(cora=(2708, 10556, 1433, 16), citeseer=(3327, 9104, 3703, 16),
                  reddit=(232965, 114615892, 602, 64), products=(2449029, 123718280, 100, 64),
                  papers100M=(111059956, 1615685872, 128, 128), amazon0505=(410236, 4878874, 96, 16),
                  tiny_dims=(1000, 20000, 8, 4), wpb_zero=(1000, 30000000, 3000, 3000))
    cases = []
    for name, (n, e, f, h) in graphs.items():
        for smem in (100, 64):
            for span in (1.0, float(n) / 3):
                ds = DS(n, e, f, span)
                ip = ref_param.inputProperty("rp", "ci", "deg", 32, 32, 4, smem, hiddenDim=h,
                                             dataset_obj=ds, enable_rabbit=True, manual_mode=False)
                ip.decider()
                out = dict(partSize=ip.partSize, dimWorker_input=ip.dimWorker_input,
                           warpPerBlock_input=ip.warpPerBlock_input,
                           dimWorker_hidden=ip.dimWorker_hidden,
                           warpPerBlock_hidden=ip.warpPerBlock_hidden,
                           reorder=bool(ip.reorder_status), reorder_flag=bool(ds.reorder_flag),
                           reorder_calls=ds.reorder_calls,
                           row_pointers=ip.row_pointers, column_index=ip.column_index)
                ip.set_input()
                out["after_set_input"] = [ip.dimWorker, ip.warpPerBlock, ip.state_set_input]
                ip.set_hidden()
                out["after_set_hidden"] = [ip.dimWorker, ip.warpPerBlock, ip.state_set_input]
                cases.append(dict(name=name, mode="auto", num_nodes=n, num_edges=e, input_dim=f,
                                  hidden=h, sharedMem=smem, avg_edgeSpan=span, expect=out))
    # reorder rule probes (param.py:108-117)
    for n, span in ((10**4, 1.0), (10**4, 1.1), (10**6, 99.0), (10**6, 101.0)):
        ds = DS(n, 10 * n, 16, span)
        ip = ref_param.inputProperty("rp", "ci", "deg", 32, 32, 4, 100, hiddenDim=16,
                                     dataset_obj=ds, enable_rabbit=True, manual_mode=False)
        ip.decider()
        cases.append(dict(name=f"reorder_rule_{n}_{span}", mode="auto", num_nodes=n,
                          num_edges=10 * n, input_dim=16, hidden=16, sharedMem=100,
                          avg_edgeSpan=span,
                          expect=dict(reorder=bool(ip.reorder_status), partSize=ip.partSize)))
    # manual mode (param.py:58-70)
    for rabbit in (False, True):
        ds = DS(1000, 5000, 16, 50.0)
        ip = ref_param.inputProperty("rp", "ci", "deg", 32, 32, 4, 100, hiddenDim=16,
                                     dataset_obj=ds, enable_rabbit=rabbit, manual_mode=True)
        ip.decider()
        cases.append(dict(name=f"manual_rabbit_{rabbit}", mode="manual", num_nodes=1000,
                          num_edges=5000, input_dim=16, hidden=16, sharedMem=100,
                          avg_edgeSpan=50.0, enable_rabbit=rabbit,
                          expect=dict(partSize=ip.partSize, dimWorker=ip.dimWorker,
                                      warpPerBlock=ip.warpPerBlock,
                                      reorder=bool(ip.reorder_status),
                                      reorder_flag=bool(ds.reorder_flag),
                                      reorder_calls=ds.reorder_calls,
                                      row_pointers=ip.row_pointers,
                                      column_index=ip.column_index)))
    return cases


def gen_kat(ref):
    """Seeded multigraph edge lists -> reference-style CSR -> reference partition ->
    expected SAG(ones) = row-nnz (closed form of unitest.py's check)."""
    rng = np.random.default_rng(42)
    cases = []
    for k, (n, e, ps, dim) in enumerate([(4, 7, 2, 2), (50, 400, 3, 16), (64, 900, 32, 16),
                                         (97, 300, 1, 5), (30, 0, 4, 8)]):
        if k == 0:
            src = np.array([0, 0, 0, 1, 2, 2, 3]); dst = np.array([1, 2, 3, 0, 0, 3, 2])
        else:
            src = rng.integers(0, n, size=e); dst = rng.integers(0, n, size=e)
        rp, ci = np_csr_from_edges(src, dst, n)
        pp, p2n = ref.build_part(ps, torch.IntTensor(rp))
        cases.append(dict(name=f"kat_{k}", num_nodes=n, dim=dim, partSize=ps,
                          src=src.tolist(), dst=dst.tolist(),
                          row_pointers=rp.tolist(), column_index=ci.tolist(),
                          partPtr_ref=[float(v) for v in pp.tolist()],
                          part2Node_ref=[float(v) for v in p2n.tolist()],
                          expected_row_value=[int(v) for v in (rp[1:] - rp[:-1])]))
    return cases


def main():
    ref = build_ref.load()
    assert ref is not None, "reference not available: run in the build container"
    os.makedirs(GOLD, exist_ok=True)
    for name, data in (("build_part", gen_build_part(ref)), ("decider", gen_decider()),
                       ("kat_ones", gen_kat(ref))):
        with open(os.path.join(GOLD, name + ".json"), "w") as f:
            json.dump(dict(generated_by="oracle/make_golden.py", cases=data), f, indent=0)
        print(name, len(data), "cases")


if __name__ == "__main__":
    main()
