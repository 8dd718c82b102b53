"""Input profile + Decider (parameter auto-tuner).

Counterpart of the reference's ``inputProperty`` (GNNAdvisor/param.py:4-164): the same
constructor arguments, public fields (``row_pointers, column_index, degrees, partSize,
dimWorker, warpPerBlock, partPtr, part2Node`` -- read by gnn_conv.py and GNNA_main.py) and
methods (``decider / set_input / set_hidden / print_param``).

Two policies:

* ``policy="compat"`` -- the reference's formulas verbatim (param.py:72-117, including the
  ``hiddenDim`` not-times-4 slip at :82), kept so that its golden outputs reproduce bit
  for bit (tests/golden/decider.json).  Those formulas size a 32-lane warp's shared
  memory on an NVIDIA SM and mean nothing on CDNA4.
* ``policy="mi355x"`` (default) -- re-derived for 64-lane wavefronts, 256 CUs and
  register accumulation (DESIGN.md "Decider"): the kernel needs no shared-memory budget,
  so the knobs that matter are the neighbor-group size (balance vs. metadata) and the
  scheduler knobs of libgnna (groups per wavefront work item, loads in flight).
  ``dimWorker``/``warpPerBlock`` keep their places in the API and are reported as the
  lane layout actually used (lanes per feature row, wavefronts per 256-thread block).

Renumbering (``enable_rabbit``).  The reference has two quirks (SURVEY.md 8a "quirks"): in auto mode the
reordered CSR is *not* copied back into the profile (param.py:108-117: the renumbering is paid for and the
kernels run on the ORIGINAL ids), in manual mode it is (param.py:59-64) but ``degrees`` is never refreshed
(GNNA_main.py:70,75: GCN then weights the NEW rows with the OLD ids' degrees) and the node features / labels
stay in the old order.  ``compat`` keeps both bug for bug.  ``mi355x`` fixes them: both modes adopt
the renumbered CSR **and** the rebuilt ``degrees``, the dataset permutes ``x`` / ``y`` / masks with ``new_id``, and
in auto mode the renumbering runs only when it pays for itself: predicted saving per aggregation x expected
aggregations > predicted host seconds (``renumbering_gate``).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Optional

WAVE = 64                # CDNA4 wavefront
WAVES_PER_BLOCK = 4      # 256-thread workgroups
INFINITY_CACHE_BYTES = 256 << 20


def num_cus() -> int:
    """Compute units of the device libgnna runs on (256 on MI355X); 256 when no device is visible (CPU-side planning)."""
    try:
        from . import _lib
        n = int(_lib.device_cus())
        return n if n > 0 else 256
    except Exception:
        return 256


NUM_CUS = 256            # MI355X (planning default; `num_cus()` asks the library)


def _pow2_at_least(x: int) -> int:
    p = 1
    while p < x:
        p <<= 1
    return p


def lanes_per_row(dim: int) -> int:
    """Lanes of a wavefront that cover one feature row (float4 per lane when dim % 4 == 0)."""
    vec = 4 if dim % 4 == 0 else (2 if dim % 2 == 0 else 1)
    return max(4, min(WAVE, _pow2_at_least((dim + vec - 1) // vec)))


def choose_part_size(avg_degree: float, dim: int) -> int:
    """Neighbor-group size for MI355X.

    The kernel merges consecutive groups of one row in registers, so the group size no
    longer bounds shared memory; it sets (a) the metadata volume (8 B per group plus one
    chunk descriptor load per 16 groups) and (b) the granularity of load balancing (a
    wavefront work item is ``groups_per_chunk`` groups, i.e. at most 16 x partSize edges).
    Measured on the Reddit-like graph at D = 64 (round 1, single pass of the first kernel):
    partSize 8 / 16 / 32 / 64 / 128 -> 4.04 / 2.80 / 2.64 / 2.51 / 2.62 ms.  Rule: the
    power of two nearest the average degree, clamped to [16, 64] -- and 128 for rows of several hundred edges:
    in the sliced schedule a work item is 64 groups of ONE source slice, i.e. partSize x 64 / phases edges, and at 16
    phases a 64-edge group leaves 256 edges per item, little against the item's fixed round trips (descriptors -> ids ->
    first rows).  Round 4, Reddit-like D = 64 (average degree 492), prepared graph, kernel ms at partSize 64 / 96 / 128 /
    160 / 192: 1.386 / 1.357 / 1.352 / 1.346 / 1.363 with the sweep kernel; the streaming kernel (D = 16, 128) does not
    care (0.725 / 0.730, 3.118 / 3.115).  ``dim`` is accepted for future per-width rules and unused today.
    """
    del dim
    target = max(1.0, float(avg_degree))
    ps = 1 << max(0, int(round(math.log2(target))))
    if target >= 256.0:
        return 128
    return int(min(max(ps, 16), 64))


def reference_formulas(avg_degree, input_dim, hidden_dim, smem_budget_kb, max_wpb=8, gap=100):
    """The reference Decider's arithmetic (param.py:73-106) as a pure function: it sizes
    per-warp shared memory (ids + one partial row, in KB) for a 32-lane-warp kernel.
    ``est_hidden`` really omits the x4 on hidden_dim (param.py:82); kept for parity."""
    ps = int(avg_degree)
    row_bytes = lambda dim: ps * 4 + dim * 4
    est_in = max_wpb * (row_bytes(input_dim) + gap * 4) / 1e3
    est_hid = max_wpb * (ps * 4 + hidden_dim + 4 * gap) / 1e3
    sm_in, sm_hid = min(est_in, smem_budget_kb), min(est_hid, smem_budget_kb)
    return dict(partSize=ps, est_input_kb=est_in, est_hidden_kb=est_hid,
                smem_input_kb=sm_in, smem_hidden_kb=sm_hid,
                wpb_input=min(int(sm_in * 1e3 / row_bytes(input_dim)), max_wpb),
                wpb_hidden=min(int(sm_hid * 1e3 / row_bytes(hidden_dim)), max_wpb),
                dw_input=min(input_dim, 32), dw_hidden=min(hidden_dim, 32))


def calibrate_phases(column_index, part_pointers, part2Node, num_out_rows, partSize, dims,
                     num_in_rows=None, reps=3, verbose=False):
    """Measure instead of guess: time the aggregation of THIS graph (device tensors) at each feature
    width in ``dims`` with the rule-based column-phase count and its neighbours, and register the
    fastest per width with libgnna (``gnna_set_graph_phases``, keyed by ``column_index``).

    The rule (size of X, average degree, "ids are scattered") is tuned on randomly labelled
    synthetic graphs; a real graph with partial locality is where it can be wrong, and a few
    milliseconds of measurement at set-up time settle it.  Returns {dim: phases}.  The hints for
    the graph (``set_graph_hints``) should already be registered so that the rule's own choice is
    among the candidates."""
    import torch
    from . import _lib
    assert column_index.is_cuda, "calibration runs on the GPU"
    n_in = int(num_in_rows if num_in_rows is not None else num_out_rows)
    chosen = {}
    for D in sorted({int(d) for d in dims if int(d) > 0}):
        X = torch.randn(n_in, D, device=column_index.device)
        out = torch.empty(int(num_out_rows), D, device=column_index.device)

        def run():
            _lib.agg_rect(_lib.MODE_SAG, X, column_index, part_pointers, part2Node, num_out_rows, partSize, out=out)

        _lib.set_graph_phases(column_index, D, 0)
        run()
        rule = _lib.last_num_phases()
        cands = sorted({1, rule, max(1, rule // 2), max(1, rule - 1), min(16, rule + 1), min(16, rule * 2)})
        timing = {}
        for b in cands:
            _lib.set_graph_phases(column_index, D, b)
            run(); run()
            torch.cuda.synchronize()
            _lib.profile_begin(reps)
            for _ in range(reps):
                run()
            torch.cuda.synchronize()
            prof = _lib.profile_end()
            timing[b] = prof["main_ms"] + prof["prologue_ms"]   # (a single pass also gets the cheaper, sparse prologue)
        # keep the rule's choice unless something else is clearly (> 2 %) faster
        best = min(timing, key=timing.get)
        if timing[best] > 0.98 * timing[rule]:
            best = rule
        _lib.set_graph_phases(column_index, D, best)
        chosen[D] = best
        if verbose:
            print("# calibrated dim {}: rule {} phase(s), measured {} -> {} phase(s)".format(
                D, rule, {b: round(t, 3) for b, t in timing.items()}, best))
        del X, out
    return chosen


# ---------------------------------------------------------------------------------------------- renumbering gate
# Gather-model rates (SURVEY 8d bytes / kernel time) the aggregation reaches, measured on MI355X (profiles/r5_bench.json,
# profiles/r6/): ids scattered over a source matrix that fits the Infinity Cache 21.6-22.5 TB/s (Reddit-like, D = 64),
# the same after renumbering 23.6 TB/s; scattered over an HBM-resident matrix 7.7 TB/s (products-like D = 64; config 5's
# rank shape 6.6), after renumbering 17.0 TB/s.
RATE_SCATTERED_CACHED, RATE_LOCAL_CACHED = 21.6e12, 23.6e12
RATE_SCATTERED_HBM, RATE_LOCAL_HBM = 7.7e12, 17.0e12
# Host seconds of the renumbering as loader.rabbit_reorder() runs it (gnna_reorder_community_csr_i32 + the relabelling of the
# CSR and the edge list), per adjacency entry and per node, at `threads` host threads -- a fit of the round-6 measurements on the
# GPU box's 16 granted CPUs (profiles/r6/reorder_stages.md: Reddit-like 1.09e8 entries / 0.23 M nodes 3.1 s, products-like
# 1.19e8 / 2.45 M 4.3 s); the walks over the backbone are one thread's work and do not shrink with the thread count
REORDER_S_PER_EDGE_SERIAL, REORDER_S_PER_EDGE_PARALLEL = 0.5e-8, 36e-8
REORDER_S_PER_NODE = 3.9e-7
REFERENCE_EPOCHS = 200 + 10          # GNNA_main.py:25 (--num_epoches) + the 10 dry runs (:188-189)


def expected_aggregations(model, in_dim, hidden, classes, epochs=REFERENCE_EPOCHS):
    """[(feature width, aggregations at that width)] of a whole training run of the reference's models (GNNA_main.py:143-171):
    GCN aggregates X W, i.e. at each layer's OUTPUT width, forward and backward; GIN aggregates at each layer's input width
    (forward; backward too unless it is the first layer) or, evaluated update-first (ops.GINConv), at the output width."""
    units = lambda w: (int(w) + 63) // 64
    out = []
    if model == "gin":
        dims = [in_dim] + [hidden] * 4 + [classes]
        for i, (a, b) in enumerate(zip(dims[:-1], dims[1:])):
            needs_dx = i > 0
            if 2 * units(b) < (2 * units(a) if needs_dx else units(a)):
                out.append((b, 2 * epochs))
            else:
                out.append((a, (2 if needs_dx else 1) * epochs))
    else:
        out = [(hidden, 2 * epochs), (classes, 2 * epochs)]
    return out


def predicted_aggregation_seconds(num_nodes, nnz, dim, scattered):
    """Gather-model estimate of one aggregation at width `dim`: bytes / the measured rate of the regime."""
    cached = num_nodes * dim * 4 < INFINITY_CACHE_BYTES
    rate = (RATE_SCATTERED_CACHED if scattered else RATE_LOCAL_CACHED) if cached else \
           (RATE_SCATTERED_HBM if scattered else RATE_LOCAL_HBM)
    return nnz * (4.0 * dim + 4.0) / rate


def predicted_reorder_seconds(num_nodes, num_edges, threads):
    t = max(1, int(threads))
    return num_edges * (REORDER_S_PER_EDGE_SERIAL + REORDER_S_PER_EDGE_PARALLEL / t) + num_nodes * REORDER_S_PER_NODE


def renumbering_gate(num_nodes, num_edges, avg_edge_span, aggregations, threads=None):
    """Does renumbering pay for itself?  -> dict(go, saving_per_epoch_set_s, aggregations, saving_s, reorder_s).
    saving = sum over widths of count x (t_scattered - t_local) x how scattered the ids are (span / (N/3), capped at 1:
    a random labelling has mean |src - dst| = N/3); go = saving_s > reorder_s.  The reference's own rule
    (sqrt(span) > sqrt(N)/100, param.py:108) stays a precondition in the caller."""
    if threads is None:
        try:
            from . import _lib
            threads = _lib.host_threads()
        except Exception:
            import os
            threads = os.cpu_count() or 1
    scatter = min(1.0, max(0.0, float(avg_edge_span) / max(1.0, num_nodes / 3.0)))
    total, count = 0.0, 0
    for dim, k in aggregations:
        gain = predicted_aggregation_seconds(num_nodes, num_edges, dim, True) - \
               predicted_aggregation_seconds(num_nodes, num_edges, dim, False)
        total += k * gain * scatter
        count += int(k)
    cost = predicted_reorder_seconds(num_nodes, num_edges, threads)
    return {"go": bool(total > cost), "saving_s": total, "aggregations": count,
            "saving_per_aggregation_s": total / count if count else 0.0, "reorder_s": cost,
            "scatter": scatter, "threads": int(threads)}


@dataclass
class LaunchKnobs:
    """The (dimWorker, warpPerBlock) pair one layer kind is launched with (param.py:25-33)."""
    dimWorker: Optional[int] = None
    warpPerBlock: Optional[int] = None


def _adopt_renumbered(ip: "inputProperty") -> None:
    """mi355x policy: the profile follows the dataset after rabbit_reorder() -- CSR, the REBUILT degrees, the statistics
    the hints are derived from (the reference adopts the CSR in manual mode only and never the degrees)."""
    ds = ip.dataset_obj
    before = ip.avgEdgeSpan
    ip.row_pointers, ip.column_index = ds.row_pointers, ds.column_index
    if getattr(ds, "degrees", None) is not None:
        ip.degrees = ds.degrees
    after = getattr(ds, "avg_edgeSpan_after", None)
    if after is not None:
        ip.avgEdgeSpan_before, ip.avgEdgeSpan = before, float(after)
        if ip.nonlocal_ids_hint is not None:
            ip.nonlocal_ids_hint = 1 if ip.avgEdgeSpan > 0.28 * ip.num_nodes else 0
        ip._say("# renumbered: avg edge span {:.0f} -> {:.0f}".format(before, after))


class _Manual:
    """manual_mode=True (param.py:58-70): the caller's knobs stay; the only decision is whether the graph
    is renumbered, and in this mode the renumbered CSR IS adopted by the profile (compat: CSR only, stale degrees kept;
    mi355x: CSR + rebuilt degrees, node data permuted by the dataset)."""

    @staticmethod
    def run(ip: "inputProperty") -> None:
        ds = ip.dataset_obj
        ds.reorder_flag = bool(ip.enable_rabbit)
        ip.reorder_status = bool(ip.enable_rabbit)
        if ip.enable_rabbit:
            if ip.policy == "mi355x":
                ds.permute_node_data = True
                ds.rabbit_reorder()
                _adopt_renumbered(ip)
            else:
                ds.rabbit_reorder()
                ip.row_pointers, ip.column_index = ds.row_pointers, ds.column_index
        ip._say("\n=> MANUAL Config Complete !!!\n")


class _Auto:
    """manual_mode=False (param.py:72-120): knobs from the policy, then the reorder rule.  compat keeps the reference's
    quirk -- the renumbered CSR is NOT copied back into the profile (param.py:108-117).  mi355x: the reference's rule is
    the precondition, the cost gate decides, and a renumbered graph is adopted."""

    @staticmethod
    def run(ip: "inputProperty") -> None:
        (ip._decide_compat if ip.policy == "compat" else ip._decide_mi355x)()
        if ip.enable_rabbit:
            ds = ip.dataset_obj
            wants = math.sqrt(ip.avgEdgeSpan) > math.sqrt(ip.num_nodes) / 100     # param.py:108
            if ip.policy == "mi355x":
                if wants:
                    aggs = ip.expected_aggregations or expected_aggregations("gcn", ip.inputDim, ip.hiddenDim, ip.hiddenDim)
                    edges = getattr(ds, "num_edges", None) or int(ip.avgNodeDegree * ip.num_nodes)
                    gate = ip.renumbering_decision = renumbering_gate(ip.num_nodes, edges, ip.avgEdgeSpan, aggs)
                    wants = gate["go"] or ip.force_renumbering
                    ip._say("# renumbering gate: predicted saving {:.3f} ms/aggregation x {} aggregations = {:.2f} s vs {:.2f} s "
                            "of renumbering on {} host threads -> {}".format(
                                gate["saving_per_aggregation_s"] * 1e3, gate["aggregations"], gate["saving_s"], gate["reorder_s"],
                                gate["threads"], "renumber" if wants else "keep the ids"))
                ds.reorder_flag = ip.reorder_status = bool(wants)
                ds.permute_node_data = True
                ds.rabbit_reorder()
                if wants:
                    _adopt_renumbered(ip)
            else:
                ds.reorder_flag = ip.reorder_status = bool(wants)
                ds.rabbit_reorder()
        ip._say("\n=> AUTO Decider Complete !!!\n")


class inputProperty(object):
    """Graph profile + launch knobs handed to every layer (reference: param.py:4-164; read by
    gnn_conv.py:17-25,44,67 and GNNA_main.py:101-110).  Public surface kept: constructor arguments, the
    attributes ``row_pointers column_index degrees partSize dimWorker warpPerBlock partPtr part2Node`` (+ the
    ``*_input`` / ``*_hidden`` pairs and the statistics), ``decider() set_input() set_hidden() print_param()``.
    Inside, the per-layer knobs are two ``LaunchKnobs`` records and ``dimWorker`` / ``warpPerBlock`` are
    views of the active one."""

    def __init__(self, row_pointers=None, column_index=None, degrees=None,
                 partSize=None, dimWorker=None, warpPerBlock=None,
                 sharedMem=None,
                 hiddenDim=None,
                 dataset_obj=None,
                 enable_rabbit=False,
                 manual_mode=True,
                 verbose=False,
                 policy="mi355x"):
        if dataset_obj is None:
            raise ValueError("Dataset object MUST SET !!!")
        if policy not in ("mi355x", "compat"):
            raise ValueError("policy must be 'mi355x' or 'compat'")
        self.dataset_obj, self.policy = dataset_obj, policy
        # the graph as the kernels see it (the caller moves these to the GPU, GNNA_main.py:107-110)
        self.row_pointers, self.column_index, self.degrees = row_pointers, column_index, degrees
        self.partPtr = self.part2Node = None
        # statistics the rules read
        self.num_nodes = dataset_obj.num_nodes
        self.avgNodeDegree = dataset_obj.avg_degree
        self.avgEdgeSpan = dataset_obj.avg_edgeSpan
        self.inputDim, self.hiddenDim = dataset_obj.num_features, hiddenDim
        # knobs: one neighbor-group size for the whole model, one LaunchKnobs per layer kind
        self.partSize = partSize
        self._knobs = {"input": LaunchKnobs(dimWorker, warpPerBlock), "hidden": LaunchKnobs(dimWorker, warpPerBlock)}
        self._active = LaunchKnobs(dimWorker, warpPerBlock)
        self.state_set_input = False
        # switches
        self.manual_mode, self.enable_rabbit, self.verbose_flag = manual_mode, enable_rabbit, verbose
        self.reorder_status = False
        # mi355x renumbering gate: [(width, count)] of the run ahead (the driver fills it in; None = the reference's
        # protocol, 2-layer GCN x 210 epochs), the gate's record, and an override for callers who know better
        self.expected_aggregations, self.renumbering_decision, self.force_renumbering = None, None, False
        # constants of the reference's shared-memory model (compat policy only)
        self.MAX_warpPerBlock, self.gap_smem = 8, 100
        self.share_memory = 0.4 * (sharedMem if sharedMem is not None else 0)
        # libgnna scheduler knobs / graph hints chosen by the mi355x policy (None = library default)
        self.groups_per_chunk = self.loads_in_flight = None
        self.avg_degree_hint = self.nonlocal_ids_hint = None

    # ---- views ----------------------------------------------------------------------------------------
    def _say(self, text):
        if self.verbose_flag:
            print(text)

    def _view(kind, field):                                   # noqa: N805  (class-body helper)
        def get(self):
            return getattr(self._knobs[kind], field)

        def put(self, value):
            setattr(self._knobs[kind], field, value)
        return property(get, put)

    dimWorker_input = _view("input", "dimWorker")
    dimWorker_hidden = _view("hidden", "dimWorker")
    warpPerBlock_input = _view("input", "warpPerBlock")
    warpPerBlock_hidden = _view("hidden", "warpPerBlock")
    del _view

    @property
    def dimWorker(self):
        return self._active.dimWorker

    @dimWorker.setter
    def dimWorker(self, value):
        self._active.dimWorker = value

    @property
    def warpPerBlock(self):
        return self._active.warpPerBlock

    @warpPerBlock.setter
    def warpPerBlock(self, value):
        self._active.warpPerBlock = value

    def _activate(self, kind):
        src = self._knobs[kind]
        self._active = LaunchKnobs(src.dimWorker, src.warpPerBlock)
        self.state_set_input = kind == "input"
        return self

    # ------------------------------------------------------------------ decider
    def decider(self):
        """Chooses the knobs (auto mode) or keeps the caller's (manual mode); param.py:51-120."""
        (_Manual if self.manual_mode else _Auto).run(self)

    def _decide_compat(self):
        k = reference_formulas(self.avgNodeDegree, self.inputDim, self.hiddenDim, self.share_memory,
                               self.MAX_warpPerBlock, self.gap_smem)
        if self.verbose_flag:
            print("input-layer shared memory (KB): {:.3f} ".format(k["est_input_kb"]))
            print("input-layer updated (KB): {:.3f}".format(k["smem_input_kb"]))
            print("hidden-layer shared memory (KB): {:.3f}".format(k["est_hidden_kb"]))
            print("hidden-layer updated (KB): {:.3f}".format(k["smem_hidden_kb"]))
        self.partSize = k["partSize"]
        self.warpPerBlock_input, self.warpPerBlock_hidden = k["wpb_input"], k["wpb_hidden"]
        self.dimWorker_input, self.dimWorker_hidden = k["dw_input"], k["dw_hidden"]

    def _decide_mi355x(self):
        # one partition is shared by all layers (GNNA_main.py:102): size it for the hidden
        # width, which is what every aggregation but GIN's first layer sees
        self.partSize = choose_part_size(self.avgNodeDegree, self.hiddenDim)
        self.dimWorker_input = lanes_per_row(self.inputDim)
        self.dimWorker_hidden = lanes_per_row(self.hiddenDim)
        self.warpPerBlock_input = WAVES_PER_BLOCK
        self.warpPerBlock_hidden = WAVES_PER_BLOCK
        # work item = groups_per_chunk consecutive groups per wavefront: keep >= ~8 work
        # items per wavefront slot for balance, and ~16-32 groups so that most rows are
        # wholly owned by one wavefront (plain stores instead of atomics)
        est_parts = self.num_nodes * max(1.0, self.avgNodeDegree / self.partSize)
        slots = num_cus() * 32
        g = 32 if self.avgNodeDegree >= 128 else 16
        while g > 1 and est_parts / g < slots * 8:
            g //= 2
        self.groups_per_chunk = g
        self.loads_in_flight = 4
        # graph hints for libgnna's column-phased schedule (gnna_tuning.avg_degree / nonlocal_ids):
        # it pays only when a row's source ids are scattered over the whole id range.  A random
        # labelling has avgEdgeSpan ~ N/3.  Measured on Reddit-like graphs with a fraction f of
        # window-local edges (span ~ (1-f) N/3): f = 0 -> 4 phases +49 %, f = 0.25 -> +13 % (2 phases) /
        # 0 % (4), f = 0.5 -> -11 % / -42 %, f = 0.75 -> -39 % / -66 %; so only (nearly) random
        # labellings qualify: span > 0.28 N.
        self.avg_degree_hint = max(1, int(self.avgNodeDegree))
        self.nonlocal_ids_hint = 1 if self.avgEdgeSpan > 0.28 * self.num_nodes else 0

    # ------------------------------------------------------------------ per-layer switches
    def set_input(self):
        """Layers that aggregate at the input width take the input-layer knobs (param.py:122-131)."""
        return self._activate("input")

    def set_hidden(self):
        """... and every other layer the hidden-layer knobs (param.py:133-141)."""
        return self._activate("hidden")

    def apply_tuning(self):
        """Push the scheduler knobs and graph hints chosen by the mi355x policy into libgnna.
        The hints are registered for this graph (keyed by its device column_index array) when the
        CSR already lives on the GPU, otherwise as process-wide defaults."""
        if self.groups_per_chunk is None and self.loads_in_flight is None and self.avg_degree_hint is None:
            return
        from . import _lib
        nonlocal_ids = self.nonlocal_ids_hint
        if nonlocal_ids is not None and self.policy == "compat" and self.reorder_status and \
                self.row_pointers is getattr(self.dataset_obj, "row_pointers", None):
            # compat, manual mode: the CSR these kernels will run on IS the renumbered one (auto mode keeps the original,
            # reference quirk param.py:108-117 -- there the ids are as scattered as they were); mi355x refreshed the hint
            # from the renumbered graph's own span when it adopted it
            nonlocal_ids = 0
        _lib.set_tuning(groups_per_chunk=self.groups_per_chunk or -1,
                        loads_in_flight=self.loads_in_flight or -1)
        if self.avg_degree_hint is None:
            return
        ci = self.column_index
        if ci is not None and getattr(ci, "is_cuda", False):
            _lib.set_graph_hints(ci, self.avg_degree_hint, bool(nonlocal_ids))
        else:
            _lib.set_tuning(avg_degree=self.avg_degree_hint, nonlocal_ids=-1 if nonlocal_ids is None else nonlocal_ids)

    def calibrate(self, dims, verbose=None):
        """Measured column-phase schedule for this graph at the given feature widths (auto mode,
        mi355x policy; needs the CSR and the partition on the GPU).  See ``calibrate_phases``."""
        if self.manual_mode or getattr(self, "policy", "mi355x") != "mi355x":
            return {}
        n = int(self.row_pointers.numel()) - 1
        self.measured_phases = calibrate_phases(self.column_index, self.partPtr, self.part2Node, n, self.partSize,
                                                dims, verbose=self.verbose_flag if verbose is None else verbose)
        return self.measured_phases

    def print_param(self):
        if self.verbose_flag:
            layer = "INPUT" if self.state_set_input else "HIDDEN"
            mode = "manual" if self.manual_mode else "auto"
            print("# {} {} partSize: {}".format(mode, layer, self.partSize))
            print("# {} {} dimWorker: {}".format(mode, layer, self.dimWorker))
            print("# {} {} warpPerBlock: {}".format(mode, layer, self.warpPerBlock))
            if not self.manual_mode:
                print("# {} {} reorder_flag: {}".format(mode, layer, self.reorder_status))
