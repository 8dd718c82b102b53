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

Reference quirks preserved in both policies (SURVEY.md 8a "quirks"): in auto mode the
reordered CSR is *not* copied back into this object (param.py:108-117), in manual mode it
is (param.py:59-64); ``degrees`` is never refreshed after reordering.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Optional

WAVE = 64                # CDNA4 wavefront
NUM_CUS = 256            # MI355X
WAVES_PER_BLOCK = 4      # 256-thread workgroups


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


@dataclass
class LaunchKnobs:
    """The (dimWorker, warpPerBlock) pair one layer kind is launched with (param.py:25-33)."""
    dimWorker: Optional[int] = None
    warpPerBlock: Optional[int] = None


class _Manual:
    """manual_mode=True (param.py:58-70): the caller's knobs stay; the only decision is whether the graph
    is renumbered, and in this mode the renumbered CSR IS adopted by the profile."""

    @staticmethod
    def run(ip: "inputProperty") -> None:
        ds = ip.dataset_obj
        ds.reorder_flag = bool(ip.enable_rabbit)
        ip.reorder_status = bool(ip.enable_rabbit)
        if ip.enable_rabbit:
            ds.rabbit_reorder()
            ip.row_pointers, ip.column_index = ds.row_pointers, ds.column_index
        ip._say("\n=> MANUAL Config Complete !!!\n")


class _Auto:
    """manual_mode=False (param.py:72-120): knobs from the policy, then the reorder rule.  Quirk kept on
    purpose: the renumbered CSR is NOT copied back into the profile here (param.py:108-117)."""

    @staticmethod
    def run(ip: "inputProperty") -> None:
        (ip._decide_compat if ip.policy == "compat" else ip._decide_mi355x)()
        if ip.enable_rabbit:
            wants = math.sqrt(ip.avgEdgeSpan) > math.sqrt(ip.num_nodes) / 100     # param.py:108
            ip.dataset_obj.reorder_flag = ip.reorder_status = bool(wants)
            ip.dataset_obj.rabbit_reorder()
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
        slots = NUM_CUS * 32
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
        if nonlocal_ids is not None and self.reorder_status and \
                self.row_pointers is getattr(self.dataset_obj, "row_pointers", None):
            # the CSR these kernels will run on IS the renumbered one (manual mode adopts it; auto mode keeps the
            # original, reference quirk param.py:108-117 -- there the ids are as scattered as they were)
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
