import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import argparse, bench_families as bf
for mk in ("nine_lanes", "nine_lanes_compact", "auto"):
    r = bf.bench_maze(argparse.Namespace(steps=200, warmup=20), 64, move_kernel=mk)
    print(mk, {k: round(v, 1) for k, v in r["us_per_step"].items()}, flush=True)
