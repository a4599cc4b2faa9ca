"""Randomised soak of the overlapped fused mixed step against the one-stream loop (not collected by pytest; run as a script
on a GPU box, `PYTHONPATH=.:tests python tests/soak_mixed.py [seconds]`): shares of random sizes (ragged last waves and
tiles), both LinDS pads, the three auto-reset modes, ring periods 2 .. 32, calls of random lengths (whole cycles overlapped,
remainders and short calls on the ordinary path) — rings, states, step counters, flags: equal bit for bit.  That the
one-stream fused step equals the families' own kernels and the oracle is tests/test_gpu_mixed.py's and the families' soaks'."""
import sys
import time

import numpy as np
import torch

from xenoverse_amd.mixed_shard import MixedShare

MODES = ("disabled", "next_step", "same_step")
REPLAYED = [0]


def run(tot, T, mode, ns, seed, plan, acts, overlap):
    sh = MixedShare(0, 1, *tot, T=T, seed=seed, linds_ns=ns, autoreset_mode=mode)
    sh.set_actions(acts["a"], acts["l"], acts["c"])
    if overlap:
        sh.set_overlap(True)
    sh.reset()
    rec, took = [], 0
    for n in plan:
        sh.step_many(n)
        torch.cuda.synchronize()
        took += int(overlap and sh.overlap_state in (1, -2))
        REPLAYED[0] += int(overlap and sh.overlap_state == -2)      # a hand-off expired and the call was replayed (round 6)
        rec.append({k: v.clone() for k, v in sh.ring.items()})
        st = {}
        for f, e in (("a", sh.ea), ("l", sh.el), ("c", sh.ec)):
            for name, v in zip(("state", "steps", "need_reset"), e.get_state()):
                st[f + "_" + name] = torch.as_tensor(v).clone()
        rec.append(st)
    flags = sh.check_errors()
    if overlap:
        sh.set_overlap(False)
    sh.close()
    return rec, flags, took


def soak(rng, seed):
    tot = (64 * int(rng.randint(1, 65)), 64 * int(rng.randint(1, 33)), 8 * int(rng.randint(1, 257)))
    T = int(rng.choice([2, 4, 6, 8, 16, 32]))
    mode = str(rng.choice(MODES))
    ns = int(rng.choice([16, 32]))
    plan = [int(rng.choice([rng.randint(1, 64), rng.randint(64, 200), T * rng.randint(1, 12)])) for _ in range(int(rng.randint(1, 4)))]
    acts = dict(a=rng.randint(0, 8, (T, tot[0])).astype(np.int32), l=rng.uniform(-1.2, 1.2, (T, tot[1], 8)).astype(np.float32),
                c=rng.randint(0, 2, (T, tot[2])).astype(np.int32))
    ref, f0, _ = run(tot, T, mode, ns, seed, plan, acts, False)
    got, f1, took = run(tot, T, mode, ns, seed, plan, acts, True)
    what = "seed=%d envs=%s T=%d mode=%s ns=%d plan=%s overlapped_calls=%d flags=%d" % (seed, tot, T, mode, ns, plan, took, f1)
    if f0 != f1 or (f1 & 8) or (mode != "disabled" and f1):
        return False, what + " FLAGS %d vs %d" % (f0, f1)
    for i, (p, q) in enumerate(zip(ref, got)):
        for k in p:
            if not torch.equal(p[k], q[k]):
                return False, what + " MISMATCH at snapshot %d key %s" % (i, k)
    return True, what


if __name__ == "__main__":
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    master = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())
    rng = np.random.RandomState(master % (2 ** 31))
    t0, n, bad, ov = time.time(), 0, 0, 0
    print("master seed", master, flush=True)
    while time.time() - t0 < seconds:
        seed = int(rng.randint(1, 2 ** 30))
        ok, what = soak(rng, seed)
        n += 1
        bad += 0 if ok else 1
        ov += 1 if "overlapped_calls=0" not in what else 0
        if not ok or n % 20 == 1:
            print(("ok " if ok else "BAD ") + what, flush=True)
    print("TOTAL %d configurations (%d with overlapped calls, %d calls replayed after an expired hand-off), %d mismatches"
          % (n, ov, REPLAYED[0], bad), flush=True)
    sys.exit(1 if bad else 0)
