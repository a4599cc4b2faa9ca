"""The ray caster at several square resolutions (16,384 frames): XV_MAZE_FILT / XV_LIB_PATH choose the variant.  GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import argparse, bench_families as bf
for res in [int(r) for r in (sys.argv[1] if len(sys.argv) > 1 else "32,64,96,128").split(",")]:
    r = bf.bench_maze(argparse.Namespace(steps=200, warmup=20), res)
    print(os.environ.get("TAG", "?"), res, {k: round(v, 1) for k, v in r["us_per_step"].items()}, flush=True)
