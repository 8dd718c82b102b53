#!/usr/bin/env python3
"""Training / profiling driver: ``python -m gnnadvisor_osdi21_amd.main [flags]``.

Counterpart of the reference's GNNAdvisor/GNNA_main.py: the same command-line flags
(:15-39, booleans are the strings 'True'/'False'), the same stages (load -> inputProperty ->
decider -> build_part -> verify | single-SpMM profile | train, :59-202), the same models
(2-layer GCN :143-153, 5-layer GIN :155-171), optimiser (Adam lr=0.01 :178), loss
(nll_loss of log_softmax vs all-ones labels :185) and the same printed lines that the
reference's log scrapers key on (``print(args)`` with ``dataset='x',`` and
``Time (ms): %.3f``, 1_log2csv.py:13-20).  Extra flags: ``--synthetic NAME`` (seeded stand-in
graph instead of a file, since no dataset ships) and ``--policy`` (Decider policy).
"""
import argparse
import os.path as osp
import sys
import time

import torch
import torch.nn.functional as F


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument("--dataDir", type=str, default="../osdi-ae-graphs", help="the path to graphs")
    p.add_argument("--dataset", type=str, default='amazon0601', help="dataset")
    p.add_argument("--dim", type=int, default=96, help="input embedding dimension size")
    p.add_argument("--hidden", type=int, default=16, help="hidden dimension size")
    p.add_argument("--classes", type=int, default=22, help="output classes size")
    p.add_argument('--model', type=str, default='gcn', choices=['gcn', 'gin'], help="GCN or GIN")
    p.add_argument("--num_epoches", type=int, default=200, help="number of epoches for training, default=200")
    p.add_argument("--partSize", type=int, default=32, help="neighbor-group size")
    p.add_argument("--dimWorker", type=int, default=32, help="number of worker threads (hint on MI355X)")
    p.add_argument("--warpPerBlock", type=int, default=4, help="wavefronts per block (hint on MI355X)")
    p.add_argument("--sharedMem", type=int, default=100, help="shared memory (KB) assumed by the compat Decider policy")
    tf = dict(type=str, choices=['True', 'False'])
    p.add_argument('--manual_mode', default='True', help="True: manual config, False: auto config", **tf)
    p.add_argument('--verbose_mode', default='False', help="True: verbose mode", **tf)
    p.add_argument('--enable_rabbit', default='False', help="True: enable locality reordering", **tf)
    p.add_argument('--loadFromTxt', default='False', help="True: load a TXT edge list, False: .npz", **tf)
    p.add_argument('--single_spmm', default='False', help="True: profile one SpMM for num_epoches rounds", **tf)
    p.add_argument('--verify_spmm', default='False', help="True: verify one SpMM against the CPU reference", **tf)
    p.add_argument('--synthetic', type=str, default=None, help="use a seeded synthetic graph (graph.CONFIGS name)")
    p.add_argument('--scale', type=float, default=1.0, help="shrink the synthetic graph")
    p.add_argument('--locality', type=float, default=0.0, help="synthetic graph: share of the edges within +-4096 ids of a hidden order")
    p.add_argument('--scramble', default='False', **tf, help="synthetic graph: relabel the nodes at random (hides the locality)")
    p.add_argument('--hip_graph', default='False', **tf,
                   help="True: capture one training epoch (forward, backward, Adam) into a HIP graph and replay it "
                        "(MI355X addition; pays on small, launch-bound graphs)")
    p.add_argument('--tune_gemm', default='False', **tf,
                   help="True: let PyTorch's TunableOp time the rocBLAS / hipBLASLt solutions for the layer GEMMs at "
                        "first use (X W and G W^T run 1.3-2x faster; costs 1-2 minutes of set-up, MI355X addition)")
    p.add_argument('--policy', type=str, default='mi355x', choices=['mi355x', 'compat'], help="Decider policy")
    p.add_argument('--force_rabbit', default='False', **tf,
                   help="True: with --enable_rabbit True in auto mode, renumber even when the mi355x cost gate says the run is "
                        "too short to win the host seconds back (MI355X addition)")
    return p


def main(argv=None, capture=None):
    """capture: a dict that receives the run's objects (dataset, inputInfo, model) -- for tests and notebooks."""
    args = build_parser().parse_args(argv)
    print(args)
    flag = lambda s: s == 'True'
    partSize, dimWorker, warpPerBlock, sharedMem = args.partSize, args.dimWorker, args.warpPerBlock, args.sharedMem
    manual_mode, verbose_mode = flag(args.manual_mode), flag(args.verbose_mode)
    enable_rabbit, loadFromTxt = flag(args.enable_rabbit), flag(args.loadFromTxt)
    single_spmm, verify_spmm = flag(args.single_spmm), flag(args.verify_spmm)

    assert torch.cuda.is_available(), "requires an MI355X GPU: there is no CPU path"
    device = torch.device('cuda')
    if flag(args.tune_gemm):
        import tempfile
        import torch.cuda.tunable as tunable
        tunable.enable(True)
        tunable.tuning_enable(True)
        tunable.set_max_tuning_duration(50)
        tunable.set_max_tuning_iterations(10)
        tunable.set_filename(osp.join(tempfile.gettempdir(), "gnna_tunableop.csv"))   # keep the results out of the cwd

    from . import load_extension
    from .decider import inputProperty
    from .loader import custom_dataset
    from .ops import GCNConv, GINConv
    GNNA = load_extension()

    # ---- loading data --------------------------------------------------------------------
    if args.synthetic:
        dataset = custom_dataset.from_synthetic(args.synthetic, args.dim, args.classes, args.scale,
                                                verbose=verbose_mode, device=device, locality=args.locality,
                                                scramble=flag(args.scramble))
    elif loadFromTxt:
        dataset = custom_dataset(osp.join(args.dataDir, args.dataset), args.dim, args.classes,
                                 load_from_txt=True, verbose=verbose_mode, device=device)
    else:
        dataset = custom_dataset(osp.join(args.dataDir, args.dataset + ".npz"), args.dim, args.classes,
                                 load_from_txt=False, verbose=verbose_mode, device=device)

    # ---- input property profile + Decider ---------------------------------------------------
    inputInfo = inputProperty(dataset.row_pointers, dataset.column_index, dataset.degrees,
                              partSize, dimWorker, warpPerBlock, sharedMem,
                              hiddenDim=args.hidden, dataset_obj=dataset, enable_rabbit=enable_rabbit,
                              manual_mode=manual_mode, verbose=verbose_mode, policy=args.policy)
    # what the run ahead will aggregate (the mi355x renumbering gate weighs the host seconds of a renumbering against it)
    from .decider import expected_aggregations
    inputInfo.expected_aggregations = [(args.hidden, args.num_epoches)] if (single_spmm or verify_spmm) else \
        expected_aggregations(args.model, dataset.num_features, args.hidden, dataset.num_classes, args.num_epoches + 10)
    inputInfo.force_renumbering = flag(args.force_rabbit)
    inputInfo.decider()
    inputInfo = inputInfo.set_input()
    if verbose_mode:
        print('----------------------------')
        inputInfo.print_param()
        print()
    inputInfo = inputInfo.set_hidden()
    if verbose_mode:
        inputInfo.print_param()
        print()
        print('----------------------------')

    # ---- neighbor partitioning -----------------------------------------------------------------
    start = time.perf_counter()
    partPtr, part2Node = GNNA.build_part(inputInfo.partSize, inputInfo.row_pointers)
    if verbose_mode:
        print("# Build nb_part (s): {:.3f}".format(time.perf_counter() - start))
    inputInfo.row_pointers = inputInfo.row_pointers.to(device)
    inputInfo.column_index = inputInfo.column_index.to(device)
    inputInfo.partPtr = partPtr.int().to(device)
    inputInfo.part2Node = part2Node.int().to(device)
    inputInfo.apply_tuning()      # scheduler knobs + this graph's hints (keyed by the device column_index)
    # graph lifecycle: the counting pass, its one synchronisation and the scratch sizing happen here, next to
    # build_part, instead of inside the first aggregation -- no epoch (and no captured epoch) synchronises or allocates
    from . import _lib as _gnna_lib
    _prep_widths = sorted({args.hidden, dataset.num_classes, dataset.num_features})
    _gnna_lib.prepare_graph(inputInfo.column_index, inputInfo.partPtr, inputInfo.part2Node, dataset.num_nodes,
                            dataset.num_nodes, inputInfo.partSize, _prep_widths)
    if not manual_mode and not (verify_spmm or single_spmm):
        # measured schedule for the widths the layers aggregate at (hidden, classes; GIN layer 1 aggregates
        # at the input width unless it is evaluated update-first)
        widths = {args.hidden, dataset.num_classes}
        if args.model == 'gin' and dataset.num_features <= 2 * args.hidden:
            widths.add(dataset.num_features)
        inputInfo.calibrate(widths)
        # the measured schedule may use other phase counts than the rule's: make their packed id copies now, not in an epoch
        _gnna_lib.prepare_graph(inputInfo.column_index, inputInfo.partPtr, inputInfo.part2Node, dataset.num_nodes,
                                dataset.num_nodes, inputInfo.partSize, _prep_widths)
    degrees = inputInfo.degrees
    if capture is not None:
        capture.update(dataset=dataset, inputInfo=inputInfo, args=args)

    # ---- single-SpMM verification / profiling (GNNA_main.py:116-137) -------------------------------
    if verify_spmm or single_spmm:
        from .verify import Verification
        # like the reference, the CLI knobs (not the Decider's) are passed here (GNNA_main.py:119-122)
        valid = Verification(args.hidden, inputInfo.row_pointers, inputInfo.column_index, degrees,
                             inputInfo.partPtr, inputInfo.part2Node,
                             inputInfo.partSize, dimWorker, warpPerBlock)
        if verify_spmm:
            valid.compute()
            valid.reference(dataset.edge_index, dataset.val, dataset.num_nodes)
            valid.compare()
        else:
            valid.profile_spmm(round=args.num_epoches)
        return 0

    # ---- model ------------------------------------------------------------------------------------
    if args.model == 'gcn':
        class Net(torch.nn.Module):
            def __init__(self):
                super().__init__()
                self.conv1 = GCNConv(dataset.num_features, args.hidden)
                self.conv2 = GCNConv(args.hidden, dataset.num_classes)

            def forward(self):
                x = self.conv1(dataset.x, inputInfo.set_input(), relu=True)   # F.relu(conv1(...)), fused (GNNA_main.py:151)
                x = self.conv2(x, inputInfo.set_hidden())
                return F.log_softmax(x, dim=1)
    else:
        class Net(torch.nn.Module):
            def __init__(self):
                super().__init__()
                dims = [dataset.num_features] + [args.hidden] * 4 + [dataset.num_classes]
                self.convs = torch.nn.ModuleList(GINConv(a, b) for a, b in zip(dims[:-1], dims[1:]))

            def forward(self):
                x = dataset.x
                for i, conv in enumerate(self.convs):
                    # F.relu after every layer but the last (GNNA_main.py:166-169), fused where the aggregation ends the layer
                    x = conv(x, inputInfo.set_input() if i == 0 else inputInfo.set_hidden(), relu=i + 1 < len(self.convs))
                return F.log_softmax(x, dim=1)

    model = Net().to(device)
    if capture is not None:
        capture.update(dataset=dataset, inputInfo=inputInfo, model=model, args=args)
    if verbose_mode:
        print(model)
    use_graph = flag(args.hip_graph)
    optimizer = torch.optim.Adam(model.parameters(), lr=0.01, capturable=use_graph)

    def nll_loss(log_prob, target):
        """F.nll_loss(log_prob, target) (mean over nodes, GNNA_main.py:186) written as gather + mean:
        torch's nll_loss reduces all N rows in ONE workgroup (0.5 ms forward + 0.3 ms backward at
        N = 410 K, 3.8 + 3.0 ms at N = 2.4 M -- a quarter of a GCN epoch); gather/mean are grid-wide."""
        return -log_prob.gather(1, target.view(-1, 1)).mean()

    def train():
        model.train()
        optimizer.zero_grad()
        loss = nll_loss(model(), dataset.y)
        loss.backward()
        optimizer.step()
        return loss

    if not use_graph:
        for _ in range(10):   # dry run
            train()
        torch.cuda.synchronize()
        start_train = time.perf_counter()
        for _ in range(1, args.num_epoches + 1):
            loss = train()
        torch.cuda.synchronize()
        train_time = time.perf_counter() - start_train
    else:
        # libgnna never synchronises and allocates its per-stream scratch on first use, so the dry runs
        # are made on the capture stream; after them one epoch is recorded and replayed per epoch
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(10):   # dry run
                train()
        side.synchronize()
        epoch_graph = torch.cuda.CUDAGraph()
        optimizer.zero_grad(set_to_none=True)
        with torch.cuda.graph(epoch_graph, stream=side):
            loss = train()
        torch.cuda.synchronize()
        start_train = time.perf_counter()
        for _ in range(1, args.num_epoches + 1):
            epoch_graph.replay()
        torch.cuda.synchronize()
        train_time = time.perf_counter() - start_train
    if verbose_mode:
        print("# final loss: {:.6f}".format(float(loss)))
    print('Time (ms): {:.3f}'.format(train_time * 1e3 / args.num_epoches))
    print()
    return 0


if __name__ == '__main__':
    sys.exit(main())
