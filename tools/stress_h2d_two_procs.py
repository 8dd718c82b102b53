"""Is a kernel that follows a host-to-device copy into a REUSED buffer always served the new bytes when two processes
time-share the GPU?  (Hypothesis for the transient of tests/test_dist_gpu.py: stale lines of the halo buffer.)
Each of W processes loops: new values into a pageable host tensor -> buf.copy_(host) (blocking) -> the library's gather over
buf (every destination row sums two rows of buf) and a plain torch op over buf -> compare with the host values.
usage: python tools/stress_h2d_two_procs.py [processes=2] [iterations=3000] [rows=5000] [dim=8]"""
import os
import sys
import time

import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def worker(rank, iters, rows, dim, q):
    sys.path.insert(0, ROOT)
    from gnnadvisor_osdi21_amd import _lib
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    # destination row i gathers rows i and (i * 7 + 3) % rows of buf
    a = torch.arange(rows)
    col = torch.stack([a, (a * 7 + 3) % rows], 1).reshape(-1).int()
    rp = (torch.arange(rows + 1) * 2).int()
    pp, p2n = _lib.build_part(32, rp)
    cold, ppd, p2nd = col.to(dev), pp.to(dev), p2n.to(dev)
    buf = torch.zeros(rows, dim, device=dev)
    out = torch.empty(rows, dim, device=dev)
    bad_lib = bad_torch = 0
    first = None
    gen = torch.Generator().manual_seed(rank)
    t0 = time.time()
    for it in range(iters):
        host = torch.randn(rows, dim, generator=gen)
        buf.copy_(host)
        _lib.agg_rect(0, buf, cold, ppd, p2nd, rows, 32, out=out)
        got = out.cpu()
        want = host + host[(a * 7 + 3) % rows]
        nb = int(((got - want).abs() > 1e-5).any(dim=1).sum())
        tsum = (buf * 2.0).cpu()
        nt = int(((tsum - host * 2.0).abs() > 1e-6).any(dim=1).sum())
        if (nb or nt) and first is None:
            first = (it, nb, nt)
        bad_lib += nb > 0
        bad_torch += nt > 0
    q.put(dict(rank=rank, iterations=iters, off_library_gather=bad_lib, off_torch_op=bad_torch, first=first, seconds=round(time.time() - t0, 1)))


if __name__ == "__main__":
    W = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
    rows = int(sys.argv[3]) if len(sys.argv) > 3 else 5000
    dim = int(sys.argv[4]) if len(sys.argv) > 4 else 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=worker, args=(r, iters, rows, dim, q)) for r in range(W)]
    for p in ps:
        p.start()
    for _ in ps:
        print(q.get(timeout=3000), flush=True)
    for p in ps:
        p.join()
