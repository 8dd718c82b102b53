"""Seeded randomised sweeps over the less-travelled entry points (rectangular / accumulate,
SDDMM, the torch module's forward/backward).  GNNA_TEST_CASES / GNNA_TEST_SEED scale them up for
soak runs; the defaults keep the suite fast."""
import os

import numpy as np
import pytest
import torch

import oracle
from gnnadvisor_osdi21_amd import _lib, graph, load_extension
from util import assert_close_f64

pytestmark = pytest.mark.gpu
CASES = int(os.environ.get("GNNA_TEST_CASES", "12"))
SEED = int(os.environ.get("GNNA_TEST_SEED", "7"))
DIMS = [1, 3, 4, 6, 16, 31, 64, 100, 129, 256, 257, 300]


def _rand_tuning(rng):
    _lib.set_tuning(int(rng.integers(1, 65)), int(rng.choice([4, 8, 16])), int(rng.choice([0, 0, 2])),
                    int(rng.integers(0, 2)), 0, int(rng.choice([1, 1, 2, 3, 7, 16, 32])), gcn_prescale=int(rng.choice([0, 1, 2])),
                    pad_rows=int(rng.choice([0, 1, 2])), zero_fill=int(rng.choice([0, 1, 1, 2])), sweep=int(rng.choice([0, 0, 1])),
                    sweep_slack=int(rng.choice([0, 1, 1000])), deterministic=int(rng.choice([0, 0, 1])))


def test_rect_accumulate_random():
    """out = A_loc X_loc (overwrite) then += A_rem X_all (accumulate), random splits, phases, widths."""
    from gnnadvisor_osdi21_amd.dist import shard_csr, split_local_remote
    rng = np.random.default_rng(SEED)
    try:
        for k in range(CASES):
            n = int(rng.integers(2, 900)); e = int(rng.integers(0, 30 * n)); D = int(rng.choice(DIMS))
            ps = int(rng.choice([1, 3, 8, 32, 64])); mode = int(rng.choice([0, 1, 2])); eps = float(rng.uniform(-1, 2))
            g = graph.uniform_graph(n, e, seed=SEED * 1000 + k)
            lo = int(rng.integers(0, n)); hi = int(rng.integers(lo + 1, n + 1))
            X = torch.randn(n, D, generator=torch.Generator().manual_seed(k))
            ref = oracle.csr_f64(mode, X.numpy(), g.row_pointers.numpy(), g.column_index.numpy(), g.degrees.numpy(), eps)[lo:hi]
            scale = oracle.csr_f64(mode, np.abs(X.numpy()), g.row_pointers.numpy(), g.column_index.numpy(),
                                   g.degrees.numpy(), abs(eps))[lo:hi]
            rp, ci = shard_csr(g.row_pointers, g.column_index, lo, hi)
            rp_l, ci_l, rp_r, ci_r = split_local_remote(rp, ci, lo, hi)
            pp_l, p2n_l = _lib.build_part(ps, rp_l); pp_r, p2n_r = _lib.build_part(ps, rp_r)
            Xd = X.cuda(); degd = g.degrees.cuda()
            out = torch.full((hi - lo, D), float("nan"), device="cuda")
            cil, ppl, p2nl = ci_l.cuda(), pp_l.cuda(), p2n_l.cuda()
            cir, ppr, p2nr = ci_r.cuda(), pp_r.cuda(), p2n_r.cuda()
            _rand_tuning(rng)
            # half of the cases go through the graph lifecycle (pinned plan, packed column ids where the schedule is sliced)
            prepared = bool(rng.integers(0, 2))
            if prepared:
                _lib.set_tuning(pack_ids=int(rng.choice([0, 1, 2])))
                if p2nl.numel():
                    _lib.prepare_graph(cil, ppl, p2nl, hi - lo, hi - lo, ps, [D])
                if p2nr.numel():
                    _lib.prepare_graph(cir, ppr, p2nr, n, hi - lo, ps, [D] if rng.integers(0, 2) else [])
            _lib.agg_rect(mode, Xd[lo:hi].contiguous(), cil, ppl, p2nl, hi - lo, ps,
                          degrees_out=degd[lo:hi].contiguous(), degrees_in=degd[lo:hi].contiguous(), epsilon=eps, out=out)
            tune_keep = dict(pack_ids=_lib.get_tuning()["pack_ids"])
            _rand_tuning(rng)
            _lib.set_tuning(**tune_keep)
            K = int(rng.choice([0, 0, 1, 2, 3, 5, 16]))       # 0: one call; else one call per source window
            for w in range(max(1, K)):
                _lib.agg_rect(mode, Xd, cir, ppr, p2nr, hi - lo, ps,
                              degrees_out=degd[lo:hi].contiguous(), degrees_in=degd, epsilon=eps, out=out,
                              accumulate=True, windows=(K, w, w + 1) if K else None)
            assert_close_f64(out.cpu().numpy(), ref, scale=scale,
                             what=f"case {k}: n={n} e={e} D={D} ps={ps} mode={mode} K={K} [{lo},{hi}) prepared={prepared} {_lib.get_tuning()}")
            if prepared:
                _lib.release_graph(cil)
                _lib.release_graph(cir)
    finally:
        _lib.reset_tuning()


def test_sddmm_random():
    rng = np.random.default_rng(SEED + 1)
    try:
        for k in range(CASES):
            n = int(rng.integers(1, 700)); e = int(rng.integers(0, 30 * n)); D = int(rng.choice([d for d in DIMS if d >= 4]))
            ps = int(rng.choice([1, 4, 32, 100]))
            g = graph.uniform_graph(n, e, seed=SEED * 2000 + k)
            gen = torch.Generator().manual_seed(k)
            A = torch.randn(n, D, generator=gen); B = torch.randn(n, D, generator=gen)
            pp, p2n = _lib.build_part(ps, g.row_pointers)
            _lib.set_tuning(groups_per_chunk=int(rng.integers(1, 64)))
            # either side may be a column block of a wider (poisoned) buffer: gnna_sddmm_ld_f32
            la = int(rng.choice([D, D, D + 1, 2 * D + 4])); lb = int(rng.choice([D, D, D + 3, 128, 2 * D]))
            lb = max(lb, D)
            oa = int(rng.integers(0, la - D + 1)); ob = int(rng.integers(0, lb - D + 1))
            Aw = torch.full((n, la), float("nan")); Aw[:, oa:oa + D] = A
            Bw = torch.full((n, lb), float("nan")); Bw[:, ob:ob + D] = B
            _lib.set_tuning(column_phases=int(rng.choice([0, 1, 1, 2, 5, 16])))
            out = _lib.sddmm(Aw.cuda()[:, oa:oa + D], Bw.cuda()[:, ob:ob + D], g.column_index.cuda(), pp.cuda(), p2n.cuda(), ps)
            ref = oracle.np_sddmm(A.numpy(), B.numpy(), g.row_pointers.numpy(), g.column_index.numpy())
            rows = np.repeat(np.arange(n), np.diff(g.row_pointers.numpy()))
            scale = np.einsum("ed,ed->e", np.abs(A.numpy().astype(np.float64))[rows],
                              np.abs(B.numpy().astype(np.float64))[g.column_index.numpy()])
            assert_close_f64(out.cpu().numpy(), ref, scale=scale, rtol=1e-5, what=f"sddmm case {k}: n={n} e={e} D={D} ps={ps} la={la}+{oa} lb={lb}+{ob} {_lib.get_tuning()}")
    finally:
        _lib.reset_tuning()


def test_module_forward_backward_random():
    GNNA = load_extension()
    rng = np.random.default_rng(SEED + 2)
    try:
        for k in range(max(4, CASES // 2)):
            n = int(rng.integers(2, 600)); e = int(rng.integers(0, 25 * n))
            fin = int(rng.choice([3, 16, 50, 129])); fout = int(rng.choice([1, 7, 16, 41, 64]))
            ps = int(rng.choice([2, 16, 32]))
            g = graph.uniform_graph(n, e, seed=SEED * 3000 + k)
            gen = torch.Generator().manual_seed(k)
            X = torch.randn(n, fin, generator=gen); W = torch.randn(fin, fout, generator=gen) * 0.3
            dY = torch.randn(n, fout, generator=gen)
            pp, p2n = GNNA.build_part(ps, g.row_pointers)
            a = [t.cuda() for t in (g.row_pointers, g.column_index, g.degrees, pp, p2n)]
            ci, deg, ppn, p2nn = g.column_index.numpy(), g.degrees.numpy(), pp.numpy(), p2n.numpy()
            _rand_tuning(rng)
            # bound for every output: 1e-4 * max(1, sum of |terms|), everything evaluated in fp64 from the formulas
            rp = g.row_pointers.numpy()
            X64, W64, dY64 = X.double().numpy(), W.double().numpy(), dY.double().numpy()
            f32 = lambda a_: np.ascontiguousarray(a_, dtype=np.float32)
            gcn = lambda M: oracle.csr_f64(1, f32(M), rp, ci, deg)
            gin = lambda M: oracle.csr_f64(2, f32(M), rp, ci, None, 0.5)
            tag = f"case {k}: n={n} e={e} fin={fin} fout={fout} ps={ps} {_lib.get_tuning()}"
            y = GNNA.forward(X.cuda(), W.cuda(), *a, ps, 32, 4)[0].cpu().numpy()
            assert_close_f64(y, gcn(X64 @ W64), scale=gcn(np.abs(X64) @ np.abs(W64)), rtol=2e-4, what="forward " + tag)
            dX, dW = GNNA.backward(dY.cuda(), X.cuda(), W.cuda(), *a, ps, 32, 4)
            G64, Ga = gcn(dY64), gcn(np.abs(dY64))
            assert_close_f64(dX.cpu().numpy(), G64 @ W64.T, scale=Ga @ np.abs(W64).T, what="d_input " + tag)
            assert_close_f64(dW.cpu().numpy(), X64.T @ G64, scale=np.abs(X64).T @ Ga, what="d_weight " + tag)
            ag = [a[0], a[1], 0.5, a[3], a[4]]
            yo, t = GNNA.forward_gin(X.cuda(), W.cuda(), *ag, ps, 32, 4)
            T64, Ta = gin(X64), gin(np.abs(X64))
            assert_close_f64(t.cpu().numpy(), T64, scale=Ta, what="gin aggregated " + tag)
            assert_close_f64(yo.cpu().numpy(), T64 @ W64, scale=Ta @ np.abs(W64), what="gin output " + tag)
            dXg, dWg = GNNA.backward_gin(dY.cuda(), t, W.cuda(), *ag, ps, 32, 4)
            t64 = t.double().cpu().numpy()
            assert_close_f64(dXg.cpu().numpy(), gin(dY64 @ W64.T), scale=gin(np.abs(dY64) @ np.abs(W64).T), rtol=2e-4,
                             what="gin d_input " + tag)
            assert_close_f64(dWg.cpu().numpy(), t64.T @ dY64, scale=np.abs(t64).T @ np.abs(dY64), what="gin d_weight " + tag)
    finally:
        _lib.reset_tuning()


def test_weight_gradient_random():
    rng = np.random.default_rng(SEED + 3)
    for k in range(CASES):
        M = int(rng.choice([1, 5, 63, 64, 65, 1000, 4099, 70000])); K = int(rng.integers(1, 200)); N = int(rng.integers(1, 150))
        gen = torch.Generator().manual_seed(k)
        X = torch.randn(M, K, generator=gen); G = torch.randn(M, N, generator=gen)
        got = _lib.xtg(X.cuda(), G.cuda()).cpu().numpy()
        ref = (X.double().t() @ G.double()).numpy()
        scale = (X.double().abs().t() @ G.double().abs()).numpy()
        assert_close_f64(got, ref, rtol=1e-5, scale=scale, what=f"xtg case {k}: M={M} K={K} N={N}")


def test_leading_dimensions_and_flags_random():
    """gnna_agg_ld_f32: random row strides on both sides (a column block of a wider matrix, a padded buffer, the
    library's preferred stride), ACCUMULATE and the ReLU epilogue in every combination, column blocks on / off /
    automatic, with and without the prepared graph.  The gaps between the rows of `out` must come back untouched."""
    rng = np.random.default_rng(SEED + 4)
    try:
        for k in range(CASES):
            n = int(rng.integers(2, 900)); e = int(rng.integers(0, 40 * n)); D = int(rng.choice(DIMS))
            ps = int(rng.choice([1, 3, 8, 32, 64, 128])); mode = int(rng.choice([0, 1, 2])); eps = float(rng.uniform(-1, 2))
            g = graph.uniform_graph(n, e, seed=SEED * 5000 + k)
            rp, ci = g.row_pointers.numpy(), g.column_index.numpy()
            gen = torch.Generator().manual_seed(k)
            X = torch.randn(n, D, generator=gen)
            acc = bool(rng.integers(0, 2)); relu = bool(rng.integers(0, 2))
            ld_in = int(rng.choice([D, D + 1, D + 4, 2 * D + 3, _lib.preferred_ld(D, n, e), 4 * ((D + 3) // 4) + 12]))
            ld_out = int(rng.choice([D, D + 1, D + 7, 2 * D]))
            off_in = int(rng.integers(0, ld_in - D + 1)); off_out = int(rng.integers(0, ld_out - D + 1))
            Xw = torch.full((n, ld_in), 1e30)                 # a poisoned wide buffer; the view is columns [off, off + D)
            Xw[:, off_in:off_in + D] = X
            Ow = torch.randn(n, ld_out, generator=gen)
            ref = oracle.csr_f64(mode, X.numpy(), rp, ci, g.degrees.numpy(), eps)
            scale = oracle.csr_f64(mode, np.abs(X.numpy()), rp, ci, g.degrees.numpy(), abs(eps))
            prior = Ow[:, off_out:off_out + D].double().numpy()
            if acc:
                ref = ref + prior
                scale = scale + np.abs(prior)
            if relu:
                ref = np.maximum(ref, 0.0)
            pp, p2n = _lib.build_part(ps, g.row_pointers)
            cid, ppd, p2nd, degd = g.column_index.cuda(), pp.cuda(), p2n.cuda(), g.degrees.cuda()
            Xd, Od = Xw.cuda(), Ow.cuda()
            _rand_tuning(rng)
            _lib.set_tuning(wide_blocks=int(rng.choice([0, 1, 2])))
            prepared = bool(rng.integers(0, 2)) and p2nd.numel() > 0
            if prepared:
                _lib.set_tuning(pack_ids=int(rng.choice([0, 1, 2])))
                _lib.prepare_graph(cid, ppd, p2nd, n, n, ps, [D] if rng.integers(0, 2) else [])
            tag = (f"case {k}: n={n} e={e} D={D} ps={ps} mode={mode} ld_in={ld_in}+{off_in} ld_out={ld_out}+{off_out} "
                   f"acc={acc} relu={relu} prepared={prepared} {_lib.get_tuning()}")
            for rep in range(2):                               # the second call runs on the cached plan
                Od.copy_(Ow)
                _lib.agg_ld(mode, Xd[:, off_in:off_in + D], cid, ppd, p2nd, n, ps, degrees_out=degd, degrees_in=degd,
                            epsilon=eps, out=Od[:, off_out:off_out + D], accumulate=acc, relu=relu)
                got = Od.cpu()
                assert_close_f64(got[:, off_out:off_out + D].numpy(), ref, scale=scale, what=f"rep {rep} " + tag)
                keep = torch.ones(ld_out, dtype=torch.bool); keep[off_out:off_out + D] = False
                assert torch.equal(got[:, keep], Ow[:, keep]), "columns outside the view were written: " + tag
            if prepared:
                _lib.release_graph(cid)
    finally:
        _lib.reset_tuning()


def test_module_repeated_calls_medium_graphs_random(monkeypatch):
    """The reference caller's path on graphs large enough for the library's own schedule choices (sliced schedule, the sweep
    kernel, packed ids): GNNA.SAG / aggregate_gin three times on the same tensors -- the first call counts the partition, the
    second makes the module prepare the graph by itself, the third reads what was prepared -- every call against the fp64
    oracle, no knob forced.  Then the column ids are rewritten in place (version counter bumps): the module must forget the
    plan and the next call must follow the new ids.  (GNNA_AUTO_PREPARE=2, the eager rule: a stream of fresh graphs like this
    one is what the default rule takes for sampled training and does not prepare.)"""
    if os.environ.get("GNNA_AUTO_PREPARE", "1") != "0":
        monkeypatch.setenv("GNNA_AUTO_PREPARE", "2")
    GNNA = load_extension()
    rng = np.random.default_rng(SEED + 5)
    for k in range(max(3, CASES // 3)):
        n = int(rng.integers(2000, 40000)); deg = float(rng.choice([4, 20, 80, 350, 600]))
        e = int(min(n * deg, 0.2 * n * n)); D = int(rng.choice([8, 16, 32, 41, 64, 100]))
        ps = int(rng.choice([8, 32, 64, 128])); kind = int(rng.integers(0, 2))
        loc = float(rng.choice([0.0, 0.0, 0.9]))
        g = graph.powerlaw_graph(n, e, int(min(n - 1, max(8, 40 * deg))), seed=SEED * 7000 + k, locality=loc, device="cuda")
        rp, ci, degs = g.row_pointers, g.column_index, g.degrees
        pp, p2n = [t.cuda() for t in GNNA.build_part(ps, rp.cpu())]
        X = torch.randn(n, D, device="cuda", generator=torch.Generator(device="cuda").manual_seed(k))
        rp_h, X_h = rp.cpu().numpy(), X.cpu().numpy()

        def check(what):
            ci_h = ci.cpu().numpy()
            mode, eps = (0, 1.0) if kind == 0 else (2, 0.5)
            ref = oracle.csr_f64(mode, X_h, rp_h, ci_h, None, eps)
            scale = oracle.csr_f64(mode, np.abs(X_h), rp_h, ci_h, None, abs(eps))
            before = GNNA.auto_prepared_graphs()
            for call in range(3):
                y = (GNNA.SAG(X, rp, ci, degs, pp, p2n, ps, 32, 4) if kind == 0
                     else GNNA.aggregate_gin(X, rp, ci, 0.5, pp, p2n, ps, 32, 4))
                assert_close_f64(y.cpu().numpy(), ref, scale=scale,
                                 what=f"case {k} {what} call {call}: n={n} nnz={g.nnz} D={D} ps={ps} kind={kind} locality={loc} "
                                      f"phases={_lib.last_num_phases()}")
            if os.environ.get("GNNA_AUTO_PREPARE", "1") != "0" and not os.environ.get("GNNA_TUNE"):
                assert GNNA.auto_prepared_graphs() == before + 1, f"case {k} {what}: prepared once, at the second sighting"
        check("as built")
        # the same tensors, other ids: every row's ids reversed in place (still a valid CSR of another graph)
        ci.copy_((n - 1) - ci)
        check("ids rewritten in place")
        del X, pp, p2n, g, rp, ci, degs
