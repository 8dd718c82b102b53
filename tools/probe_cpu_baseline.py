"""Why do the passes of bench.py's CPU baseline alternate between ~100 and ~195 ms?  Runs the oracle's
OpenMP SpMM (checker code, timed as the CPU baseline only) on the bench graph under a few thread / placement settings, 14
passes each, and prints the pass times.  usage: python tools/probe_cpu_baseline.py   (needs a GPU only to generate the graph)"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import os, sys, time, json
sys.path.insert(0, %r)
import numpy as np, torch
import oracle
d = np.load(sys.argv[1])
rp, ci, X = d["rp"], d["ci"], d["X"]
Xf = oracle.first_touch_copy(X)
out = oracle.first_touch_copy(None, X.shape)
for _ in range(2):
    oracle.csr_sag_omp(Xf, rp, ci, out=out)
ts = []
for _ in range(14):
    t0 = time.perf_counter(); oracle.csr_sag_omp(Xf, rp, ci, out=out); ts.append(round((time.perf_counter() - t0) * 1e3, 1))
print(json.dumps(dict(threads=oracle.num_threads(), ms=ts)))
""" % ROOT

if __name__ == "__main__":
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch
    from gnnadvisor_osdi21_amd import graph
    g = graph.make_config_graph("reddit-like", device="cuda").to("cpu")
    X = torch.randn(g.num_nodes, 64, generator=torch.Generator().manual_seed(1234)).numpy()
    path = "/tmp/gnna_cpu_probe.npz"
    np.savez(path, rp=g.row_pointers.numpy(), ci=g.column_index.numpy(), X=X)
    ncpu = os.cpu_count()
    variants = [("default (bench.py): OMP_PROC_BIND=close OMP_PLACES=cores", dict(OMP_PROC_BIND="close", OMP_PLACES="cores")),
                ("spread over cores", dict(OMP_PROC_BIND="spread", OMP_PLACES="cores")),
                ("no binding", dict(OMP_PROC_BIND="false")),
                ("close, active wait", dict(OMP_PROC_BIND="close", OMP_PLACES="cores", OMP_WAIT_POLICY="active")),
                ("close, half the threads", dict(OMP_PROC_BIND="close", OMP_PLACES="cores", OMP_NUM_THREADS=str(max(1, ncpu // 4)))),
                ("close, one thread per hardware thread", dict(OMP_PROC_BIND="close", OMP_PLACES="threads", OMP_NUM_THREADS=str(ncpu)))]
    print(json.dumps(dict(host_cpus=ncpu)))
    for name, env in variants:
        e = {k: v for k, v in os.environ.items() if not k.startswith("OMP_")}
        e.update(env)
        r = subprocess.run([sys.executable, "-c", CHILD, path], env=e, capture_output=True, text=True, timeout=600)
        print(name, "->", r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:], flush=True)
