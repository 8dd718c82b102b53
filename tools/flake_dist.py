"""The loop that looks for intermittent failures of the two-ranks-on-one-GPU step (tests/test_dist_gpu.py), one attempt per
session, no retries.  A session = one two-process run of `_two_ranks_once` (aggregations + one strict layer step), or one
two-rank `dist_main` driver run with --driver.  Several loops may run side by side (more processes sharing the GPU).

    python tools/flake_dist.py --sessions 120 --tag a [--driver] [--variants 1:halo,3:allgather]

Appends one JSON line per session to gpurun_out/flake/<tag>.jsonl: variant, ok, exit codes, seconds, and the checker's
ambiguity statistics (elements of H1 inside the bound of zero, the smallest |H1| / scale, how many of them the fp32 path
computed with the other sign than fp64).  Failing sessions leave their dumps under gpurun_out/dist_dump/."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sessions", type=int, default=40)
    ap.add_argument("--tag", default="loop")
    ap.add_argument("--driver", action="store_true")
    ap.add_argument("--variants", default="1:allgather,3:allgather,1:halo,3:halo")
    a = ap.parse_args()
    os.environ.setdefault("GNNA_DEBUG_POISON", "1")
    os.makedirs(os.path.join(ROOT, "gpurun_out", "flake"), exist_ok=True)
    out = open(os.path.join(ROOT, "gpurun_out", "flake", a.tag + ".jsonl"), "a")
    variants = [(int(v.split(":")[0]), v.split(":")[1]) for v in a.variants.split(",")]
    import test_dist_gpu as T
    bad = 0
    for i in range(a.sessions):
        t0 = time.time()
        if a.driver:
            model = ("gcn", "gin")[i % 2]
            try:
                T.test_sharded_training_driver_two_ranks(model)
                rec = dict(kind="driver", model=model, ok=True)
            except BaseException as exc:                       # noqa: BLE001 -- the loop records, it does not judge
                rec = dict(kind="driver", model=model, ok=False, error=repr(exc)[-1500:])
        else:
            chunks, exchange = variants[i % len(variants)]
            try:
                res, codes = T._two_ranks_once(chunks, exchange)
                ok = codes == [0, 0] and all(r[1] for r in res)
                rec = dict(kind="pair", chunks=chunks, exchange=exchange, ok=bool(ok), codes=codes,
                           results=[(r[0], r[1], r[2] if isinstance(r[2], str) else list(r[2])) for r in res])
            except BaseException as exc:                       # noqa: BLE001
                rec = dict(kind="pair", chunks=chunks, exchange=exchange, ok=False, error=repr(exc)[-1500:])
        rec["seconds"] = round(time.time() - t0, 2)
        rec["i"] = i
        bad += not rec["ok"]
        out.write(json.dumps(rec) + "\n")
        out.flush()
    print(f"{a.tag}: {a.sessions} sessions, {bad} failed")


if __name__ == "__main__":
    main()
