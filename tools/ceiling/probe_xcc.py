"""Which XCD does workgroup b run on?  Reads the XCC_ID hardware register in every block of grids of several sizes and
reports how often it equals blockIdx % 8 (the mapping the sliced schedule's per-XCD chunk ranges assume for locality).
usage: probe_xcc.py   (build first: tools/ceiling/build.sh)"""
import ctypes
import json
import os

import torch

here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, "libceiling.so"))
lib.xcc_of_block_launch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
dev = torch.device("cuda:0")
for blocks, threads in ((8, 256), (256, 256), (2048, 256), (57344, 256), (256, 1024), (100000, 64)):
    out = torch.full((blocks,), -1, dtype=torch.int32, device=dev)
    assert lib.xcc_of_block_launch(out.data_ptr(), blocks, threads) == 0
    torch.cuda.synchronize()
    x = out.cpu()
    want = torch.arange(blocks, dtype=torch.int32) % 8
    print(json.dumps(dict(blocks=blocks, threads=threads, xcds_seen=sorted(set(x.tolist())),
                          share_on_block_mod_8=round(float((x == want).float().mean()), 6),
                          first_16=x[:16].tolist())), flush=True)
