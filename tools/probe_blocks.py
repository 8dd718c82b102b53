import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from gnnadvisor_osdi21_amd import _lib, graph
dev = torch.device("cuda:0")
g = graph.make_config_graph("reddit-like", device=dev)
ps = 128
pp, p2n = _lib.build_part(ps, g.row_pointers.cpu())
ppd, p2nd = pp.to(dev), p2n.to(dev)
def kernel_ms(X, steps=10):
    out = torch.empty_like(X)
    call = lambda: _lib.sag(X, g.row_pointers, g.column_index, g.degrees, ppd, p2nd, ps, 32, 4, out=out)
    for _ in range(4): call()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps): call()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / steps, _lib.last_num_phases(), _lib.last_num_launches()
for D in (96, 100, 128, 160, 192, 256):
    X = torch.randn(g.num_nodes, D, device=dev)
    row = []
    for blocks in (2, 1):
        _lib.reset_tuning(); _lib.set_tuning(wide_blocks=blocks)
        _lib.prepare_graph(g.column_index, ppd, p2nd, g.num_nodes, g.num_nodes, ps, [D])
        w, ph, la = kernel_ms(X)
        row.append("blocks=%d: %.3f ms (%d ph, %d launches)" % (blocks, w, ph, la))
        _lib.release_graph(g.column_index)
    print("reddit-like D=%3d prepared, wall per call | " % D + " | ".join(row), flush=True)
    del X
