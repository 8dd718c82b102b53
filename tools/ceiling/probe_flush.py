"""Flush rate of the chip: 256-byte rows added with float atomics / stored / read-add-written at pseudo-random rows of
a [rows, 64] fp32 matrix, for matrices from L2-sized to HBM-sized.  (The sliced schedule flushes one row per (row,
slice) piece with float atomics: 2.1 M flushes per step at 8 phases on the Reddit-like headline.)
usage: probe_flush.py   (build first: tools/ceiling/build.sh)"""
import ctypes
import json
import os

import torch

here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, "libceiling.so"))
lib.flush_rate_launch.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.c_int]
dev = torch.device("cuda:0")
for rows in (8192, 232965, 2449029):
    Y = torch.zeros(rows, 64, device=dev)
    for waves, per_wave in ((8192, 256), (65536, 32), (262144, 8), (2097152, 1)):
        rec = dict(rows=rows, MB=round(rows * 256 / 1e6, 1), waves=waves, flushes_per_wave=per_wave)
        for name, how in (("atomic", 0), ("store", 1), ("read_add_write", 2)):
            for _ in range(2):
                lib.flush_rate_launch(Y.data_ptr(), rows, waves, per_wave, how)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(5):
                lib.flush_rate_launch(Y.data_ptr(), rows, waves, per_wave, how)
            b.record()
            torch.cuda.synchronize()
            ms = a.elapsed_time(b) / 5
            rec[name] = dict(ms=round(ms, 4), G_flushes_per_s=round(waves * per_wave / ms / 1e6, 2))
        print(json.dumps(rec), flush=True)
