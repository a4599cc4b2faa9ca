"""Where the wall time of a short timed batch goes (the driver's `--steps 20`): host-side stamps around the pieces of
bench.py's timed region, for graph replay and for plain launches.  GPU box only."""
import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from xenoverse_amd import _lib
from xenoverse_amd.anymdp import AnyMDPVecEnv

K = int(os.environ.get("K", "20"))
n_env = int(os.environ.get("ENVS", "65536"))
n_task = int(os.environ.get("TASKS", str(n_env)))
S, A, P = 64, 8, K
env = AnyMDPVecEnv(n_env, device="cuda:0", seed=1234, autoreset_mode="same_step")
tab = bench.make_tables(env.engine, torch, _lib, n_task, 0, 1235, S, A)
env_task = (torch.arange(n_env, device=env.device, dtype=torch.int32) // (n_env // n_task)).contiguous()
env.set_task(tab, env_task_index=env_task)
sys.argv = [sys.argv[0]]
args = bench.parse()
bench.choose_search(env, torch, args, n_task, S, A)
actions = torch.randint(0, A, (P, n_env), device=env.device, dtype=torch.int32)
env.reset()
ring = env.step_many(1, actions)
med = bench.median

for mode in ("on", "off"):
    env.set_step_many_graph(mode)
    for _ in range(5):
        env.step_many(K, actions, out=ring)
    torch.cuda.synchronize()
    rows = []
    for rep in range(41):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        t1 = time.perf_counter()
        env.step_many(K, actions, out=ring)
        t2 = time.perf_counter()
        e1.record()
        t3 = time.perf_counter()
        while not e1.query():
            pass
        t4 = time.perf_counter()
        torch.cuda.synchronize()
        t5 = time.perf_counter()
        rows.append((t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t5 - t0, e0.elapsed_time(e1) * 1e-3))
    names = ["e0.record", "step_many call", "e1.record", "spin on e1", "synchronize", "WALL", "event e0->e1"]
    print("graph", mode, "K", K, " ".join("%s=%.1fus" % (n, med([r[i] for r in rows]) * 1e6) for i, n in enumerate(names)), flush=True)
    # the same region with no event records (wall only)
    walls = []
    for rep in range(41):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        env.step_many(K, actions, out=ring)
        torch.cuda.synchronize()
        walls.append(time.perf_counter() - t0)
    print("graph", mode, "no events, blocking synchronize: WALL=%.1fus" % (med(walls) * 1e6), flush=True)

# raw HIP events through ctypes (what a C-side record inside step_many would cost)
import ctypes as C
hip = C.CDLL("libamdhip64.so")
stream = C.c_void_p(env.engine.torch_stream.cuda_stream)
evs = [C.c_void_p() for _ in range(2)]
for e in evs:
    assert hip.hipEventCreate(C.byref(e)) == 0
for mode in ("on", "off"):
    env.set_step_many_graph(mode)
    for KK in (K,):
        rows = []
        for rep in range(41):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            hip.hipEventRecord(evs[0], stream)
            t1 = time.perf_counter()
            env.step_many(KK, actions, out=ring)
            t2 = time.perf_counter()
            hip.hipEventRecord(evs[1], stream)
            t3 = time.perf_counter()
            while hip.hipEventQuery(evs[1]) != 0:
                pass
            t4 = time.perf_counter()
            torch.cuda.synchronize()
            t5 = time.perf_counter()
            ms = C.c_float()
            hip.hipEventElapsedTime(C.byref(ms), evs[0], evs[1])
            rows.append((t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t5 - t0, ms.value * 1e-3))
        print("raw hip events, graph", mode, "K", KK, " ".join("%s=%.1fus" % (n, med([r[i] for r in rows]) * 1e6) for i, n in enumerate(names)), flush=True)
        # only the closing event (start stamp = host clock)
        rows = []
        for rep in range(41):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            env.step_many(KK, actions, out=ring)
            hip.hipEventRecord(evs[1], stream)
            while hip.hipEventQuery(evs[1]) != 0:
                pass
            t4 = time.perf_counter()
            torch.cuda.synchronize()
            t5 = time.perf_counter()
            rows.append((t4 - t0, t5 - t0))
        print("closing event only, graph", mode, "spin end=%.1fus WALL=%.1fus" % (med([r[0] for r in rows]) * 1e6, med([r[1] for r in rows]) * 1e6), flush=True)
