"""The opt-in fp32 filter of the ray caster at 64 x 64 and 256 x 256 (16,384 frames).  GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import argparse, bench_families as bf
for res in (64, 256):
    for prec in ("f32", "exact"):
        r = bf.bench_maze(argparse.Namespace(steps=200, warmup=20), res, precision=prec)
        print(prec, res, {k: round(v, 1) for k, v in r["us_per_step"].items()}, flush=True)
