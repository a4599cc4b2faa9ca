#!/usr/bin/env python3
"""bench.py — env-steps/sec of the AnyMDP hot path (BASELINE.json metric) on N MI355X GPUs of one node.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...; started WITHOUT a launcher,
   `python bench.py --gpus N` launches exactly that as a fresh child process before anything touches the GPU: self_launch)

Workload = BASELINE.json configs[1]: anymdp |S|=64, |A|=8, 65,536 envs per GPU, synthetic tasks generated on
the device.  Default task sharing is "distinct" (2a: one task per env, 44 GiB of
tables per GPU, every table read misses every cache); --tasks 1024 gives "shared" (2b).
A "step" is one vector step of all envs of a rank = one launch of the step kernel; K steps are K back-to-back
launches (xv_anymdp_step_many), auto-reset SAME_STEP, actions pre-generated on the device.

Timing: W warm-up steps, then the K-step batch is timed R times (`--repeats`, default 25); every repetition is
bracketed by a barrier + device synchronisation on both sides and by HIP events on the launch stream, the MAX over
ranks is taken per repetition and the MEDIAN repetition is reported (`steps` stays K; `repeats` = R).  With the
driver's K = 20 one repetition is a 0.15 ms region, which a single sample cannot resolve.

One JSON line on rank 0: metric/value/unit/... as the driver's contract says, plus
  roofline      HBM bound for the step kernel.  `traffic` = bytes the kernel really moved per launch (PMC counters of the
                committed profile of THIS kernel source, else null).  `achieved` / `frac` price those bytes when they are below
                SURVEY.md §8(d)'s algorithmic 562 B per env-step (the search reads one 128-byte line of the 512-byte row) — a
                fraction above 1 is never printed; the survey's figure stays as `achieved_survey_bytes` / `frac_survey_bytes`
  long_call     SURVEY.md §8(d)'s own definition (>= 2,000 back-to-back steps after >= 100 warm-up, median of 5, both clocks)
                for the three issue modes on the same tables: one stream, overlapped, fused roll-out
  cpu_baseline  SURVEY.md §8(d): the reference's execution style (one env per Python object, one step() per call,
                NumPy global RNG; oracle/py_ref_style.py) in P = usable CPUs processes; `c_oracle` = the
                vectorised C oracle (OpenMP) on a working set that does not fit the last-level cache
  rccl / rccl_ranks   N > 1: whether the RCCL process group came up and what all_reduce(ones) returned
"""
import argparse
import gc
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGO_BYTES_PER_ENV_STEP = {8: 8 * 64 + 50}   # w*S + 50, w = 8 (fp64 CDF), S = 64  -> 562 B
# keys of the `long_call` block and of its rows (tests/test_host_round6.py pins them on CPU, tests/test_gpu_bench_line.py on the
# line a GPU run prints: a renamed key fails a test, not a reader)
LONG_CALL_MODES = ("one_stream", "overlapped", "fused_rollout")
LONG_CALL_ROW_KEYS = ("steps", "warmup", "repeats", "launches_per_call", "us_per_step", "env_steps_per_s_events", "wall_us_per_step",
                      "env_steps_per_s", "overlap_state", "graph_state", "device_error_flags", "issue")
ROOFLINE_KEYS = ("bound", "achieved", "peak", "unit", "frac", "traffic", "basis", "achieved_survey_bytes", "frac_survey_bytes")
HBM_PEAK_GBS = 8000.0
EXIT_WATCHDOG = 3


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--workload", default="anymdp", choices=["anymdp", "mixed"],
                    help="anymdp: BASELINE.json's metric (configs[1]); mixed: configs[4], the mixed task batch (anymdp + linds + "
                         "metacontrol) sharded over the GPUs with the all-gather of its rollout chunks (scripts/bench_mixed.py)")
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--repeats", type=int, default=25, help="the K-step batch is timed this many times; median reported")
    ap.add_argument("--envs", type=int, default=65536, help="envs per GPU")
    ap.add_argument("--tasks", type=int, default=0, help="tasks per GPU (0 = one per env, config 2a)")
    ap.add_argument("--period", type=int, default=32, help="rollout-chunk ring length T")
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--cpu-table-gib", type=float, default=16.0, help="working set of the C-oracle CPU line")
    ap.add_argument("--gather-timeout", type=int, default=120,
                    help="N>1: seconds allowed for process-group set-up and for the all-gather pass")
    ap.add_argument("--no-allgather", action="store_true",
                    help="N>1: skip the RCCL all-gather of rollout chunks (pure replicas)")
    ap.add_argument("--search", default="auto", choices=["auto", "binary", "fence", "bucket"],
                    help="auto (what the library's own AUTO does, xv_anymdp_set_search): the bucket search (one table line per "
                         "env-step, --buckets lines per row) when the census of its lines expects few enough unanswerable draws per "
                         "launch and they fit the free HBM, else the fence search (two dependent lines)")
    ap.add_argument("--buckets", type=int, default=16, choices=[16, 32, 64])
    ap.add_argument("--fused", action="store_true", help="also time the fused T-step rollout kernel")
    ap.add_argument("--graph", default="auto", choices=["auto", "on", "off"],
                    help="xv_anymdp_step_many: replay ring cycles from a hipGraph (auto: only for small batches)")
    ap.add_argument("--overlap", default="auto", choices=["auto", "on", "off"],
                    help="xv_anymdp_step_many overlap mode (xv_anymdp_set_step_many_overlap): consecutive vector steps on two HIP "
                         "streams, each wave taking its envs over from the same wave of the step before through the env records; "
                         "auto = on (the library uses it for calls of >= 64 steps; same results)")
    ap.add_argument("--sweep-envs", default=None,
                    help="comma list of envs/GPU (e.g. 16384,32768,65536,131072,262144): time the 2a step at each size "
                         "and write --sweep-out instead of the bench line")
    ap.add_argument("--sweep-out", default=os.path.join(ROOT, "gpurun_out", "anymdp_envs_sweep.json"))
    ap.add_argument("--no-variants", action="store_true",
                    help="skip the `search_variants` object (the fence and the bucket search each timed on the same workload)")
    ap.add_argument("--sustain-seconds", type=float, default=6.0,
                    help="after the timed passes: an UNTIMED leg of back-to-back stepping of this length (excluded from `value`; "
                         "`sustain_s` in the line) so that a monitor sampling every few seconds sees the GPU busy; 0 = off")
    ap.add_argument("--long-steps", type=int, default=2016,
                    help="the `long_call` block (SURVEY.md 8(d): >= 1,000 back-to-back step launches after >= 100 warm-up steps): "
                         "steps per call, rounded to whole ring cycles of --period; 0 = no block")
    ap.add_argument("--long-repeats", type=int, default=5)
    ap.add_argument("--no-live-pmc", action="store_true",
                    help="N = 1: do not measure `roofline.traffic` live (two child runs of this script under rocprofv3 --pmc, "
                         "FETCH_SIZE and WRITE_SIZE in separate passes, before this process touches the GPU; ~20 s); the committed "
                         "profile of this kernel source is used instead")
    ap.add_argument("--no-families", action="store_true",
                    help="N = 1: skip the `families` object (configs 3, 4 and the per-GPU share of 5, a few seconds)")
    ap.add_argument("--transport", default="auto", choices=["auto", "torch", "rccl"],
                    help="N > 1: how rollout chunks are all-gathered — torch.distributed (backend nccl = RCCL) or the C-ABI's "
                         "xv_rollout_allgather over librccl directly; auto = rccl when xv_rccl_unique_id succeeds")
    ap.add_argument("--exchange-selftest", action="store_true",
                    help="N>1 control flow (process group, watchdog, pack -> all-gather -> unpack, MAX over ranks) on "
                         "fabricated CPU records, no GPU and no stepping: a functional test, not a measurement")
    return ap.parse_args()


# -----------------------------------------------------------------------------------------------------------
# CPU baseline.  Runs BEFORE anything touches the GPU: line A forks worker processes.
# -----------------------------------------------------------------------------------------------------------
def _mem_available_gib():
    try:
        for ln in open("/proc/meminfo"):
            if ln.startswith("MemAvailable:"):
                return int(ln.split()[1]) / 2**20
    except Exception:
        pass
    return 8.0


def usable_cpus():
    """CPUs this process may really use: the affinity mask, capped by the cgroup CPU quota (os.cpu_count() reports the
    machine's, which oversubscribes a quota-limited container and collapses OpenMP's spinning barriers)"""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(float(txt[0]) / float(txt[1]) + 0.5)))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, int(q / per + 0.5)))
            break
        except Exception:
            pass
    return n


def _py_worker(args):
    idx, n_obj, seconds, seed = args
    import oracle
    from oracle import py_ref_style
    tab = oracle.anymdp_synth(seed=seed, task_index_base=idx * n_obj, n_task=n_obj, S=64, A=8, s0_max=4)
    v, n = py_ref_style.time_python_loop(tab, seconds, seed=idx)
    return v, n


def cpu_line_a(seconds, seed):
    """the reference's execution style in P = os.cpu_count() processes, each stepping its own env objects"""
    import multiprocessing as mp
    P = usable_cpus()
    n_obj = 32 if _mem_available_gib() > 0.2 * P else 8          # ~0.1 GiB per worker at 32 objects
    ctx = mp.get_context("fork")          # no exec; nothing in this process has touched the GPU yet
    t0 = time.perf_counter()
    with ctx.Pool(P) as pool:
        res = pool.map(_py_worker, [(i, n_obj, seconds, seed) for i in range(P)], chunksize=1)
    return {"value": float(sum(v for v, _ in res)), "unit": "env-steps/s", "cores": P, "kind": "port",
            "sample": "oracle/py_ref_style.py: the reference's style (one env per Python object, one step() per call, "
                      "NumPy global RNG) in P = %d processes (usable CPUs; os.cpu_count() = %s) x %d env objects (S=64, A=8, one "
                      "synthetic task per object), %.0f s each, aggregate (%.1f s wall incl. set-up)"
                      % (P, os.cpu_count(), n_obj, seconds, time.perf_counter() - t0)}


def cpu_line_b(seconds, seed, table_gib):
    """the vectorised C oracle (OpenMP, all host threads) on a table set that does not fit the last-level cache"""
    import numpy as np
    import oracle
    cores = max(1, min(usable_cpus(), oracle.lib().xo_max_threads()))
    gib = min(table_gib, 0.4 * _mem_available_gib())
    n_task = max(64, int(gib * 2**30) // (64 * 8 * 64 * 16) // 64 * 64)      # 512 KiB of flat tables per task
    per = max(1, 65536 // n_task)
    n_env = n_task * per
    t0 = time.perf_counter()
    tab = oracle.anymdp_synth(seed=seed, task_index_base=0, n_task=n_task, S=64, A=8, s0_max=4)
    t_synth = time.perf_counter() - t0
    env_task = (np.arange(n_env, dtype=np.int32) % n_task).astype(np.int32)   # neighbours use different tasks
    ora = oracle.AnyMDPOracle(tab, env_task)
    ora.reset(seed, 0, 0)
    rng = np.random.RandomState(0)
    acts = rng.randint(0, 8, (64, n_env)).astype(np.int32)
    for k in range(5):
        ora.step(seed, 0, 1 + k, acts[k % 64], 2, n_threads=cores)
    t0 = time.perf_counter()
    k = 0
    while True:
        for _ in range(10):
            ora.step(seed, 0, 100 + k, acts[k % 64], 2, n_threads=cores)
            k += 1
        if time.perf_counter() - t0 >= seconds:
            break
    dt = time.perf_counter() - t0
    return {"value": n_env * k / dt, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": "oracle/xeno_oracle.c step (OpenMP, %d threads), %d envs over %d distinct S=64,A=8 tasks "
                      "(%.1f GiB of tables, built in %.1f s), %d vector steps in %.1f s"
                      % (cores, n_env, n_task, n_task * 512 / 2**20, t_synth, k, dt)}


def cpu_baseline(seconds, seed, table_gib):
    try:
        out = cpu_line_a(seconds, seed)           # forks: before the OpenMP runtime of line B starts threads here
    except Exception as ex:                       # never lose the bench line to the baseline
        out = {"value": None, "unit": "env-steps/s", "cores": os.cpu_count(), "kind": "port", "sample": "failed: %r" % (ex,)}
    try:
        out["c_oracle"] = cpu_line_b(seconds, seed, table_gib)
    except Exception as ex:
        out["c_oracle"] = {"error": repr(ex)}
    return out


def under_profiler():
    e = os.environ
    return "rocprof" in e.get("LD_PRELOAD", "") or any(k.startswith(("ROCPROF", "ROCP_")) for k in e)


# -----------------------------------------------------------------------------------------------------------
def kernel_source_hash():
    from xenoverse_amd.build import source_hash
    return source_hash(("anymdp.hip", "philox.h", "xv_common.h"))


def _kernel_kind(name):
    """'plain' | 'hand' | 'rollout' from the step kernel's template arguments <INJECT, G, ROLLOUT, TICKDEV, BK, HAND>"""
    try:
        args = [x.strip() for x in name[name.index("<") + 1:name.rindex(">")].split(",")]
    except ValueError:
        return "plain"
    if len(args) > 2 and args[2] == "true":
        return "rollout"
    if len(args) > 5 and args[5] == "true":
        return "hand"
    return "plain"


def pmc_traffic(n_env, n_task, search, overlap=False, kind=None):
    """HBM bytes per launch of the step kernel from the committed PMC run (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in
    separate passes, gfx950 x2 read correction; scripts/gpu_pmc.sh -> profiles/*pmc_traffic*.json).  Counters cannot
    be read from inside this process, so the figure is the profiled one for the same workload AND the same kernel
    source (hash of csrc/anymdp.hip + headers recorded with the profile) — a stale profile yields null.
    kind: 'plain' (one launch per step on one stream), 'hand' (the overlapped launches: the HAND instantiation; the plain
    kernel's figure when the profile has none — the same table line and streams per env-step, the polls are L2 hits) or
    'rollout' (the fused roll-out: one launch per ring cycle — bytes per LAUNCH, i.e. of `period` steps).  Default: by `overlap`."""
    import glob
    want_src = kernel_source_hash()
    kind = kind or ("hand" if overlap else "plain")
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_traffic*.json")), reverse=True):
        try:
            d = json.load(open(f))
            k = d.get("bench_key", {})
            want = "2a" if n_task == n_env else "2b"
            if k.get("workload") == want and k.get("search") == search and k.get("envs_per_gpu") == n_env \
                    and k.get("kernel_source_sha16") == want_src:
                cand = {}
                for name, v in d["kernels"].items():
                    if "step_kernel" in name and "traffic_bytes_per_launch_corrected" in v:
                        cand.setdefault(_kernel_kind(name), v["traffic_bytes_per_launch_corrected"])
                pick = cand.get(kind)
                if pick is None and kind == "hand":
                    pick = cand.get("plain")
                if pick is None and kind == "plain":
                    pick = cand.get("hand")
                if pick is not None:
                    return pick, os.path.basename(f)
        except Exception:
            pass
    return None, None


def live_pmc_traffic(args):
    """HBM bytes per launch of the step kernels MEASURED BY THIS RUN: two child processes `rocprofv3 --pmc <counter>
    --kernel-trace -- python3 bench.py <the same workload, 128 steps, --fused>` — FETCH_SIZE and WRITE_SIZE in separate passes, only
    --kernel-trace beside the counters, the program itself behind `--` — started BEFORE this process touches the GPU (a child of
    a process that holds the device must not exec), outputs under /tmp.  Units and the gfx950 correction as MI355X_MICROARCH.md's
    HBM section prescribes (counter values are KB; read bytes = 2 x FETCH_SIZE; WRITE_SIZE exact), the same arithmetic as
    scripts/pmc_to_json.py.  -> {"plain" | "hand" | "rollout": bytes per launch, "source": ...} or {"error": ...}; never raises."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return {"error": "rocprofv3 not found"}
    t0 = time.perf_counter()
    agg = {}
    try:
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            d = tempfile.mkdtemp(prefix="xv_pmc_%s_" % c, dir="/tmp")
            cmd = [exe, "--pmc", c, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "pmc", "--", sys.executable,
                   os.path.abspath(__file__), "--gpus", "1", "--steps", "128", "--warmup", "32", "--repeats", "2", "--envs", str(args.envs),
                   "--tasks", str(args.tasks), "--period", str(args.period), "--seed", str(args.seed), "--search", args.search,
                   "--buckets", str(args.buckets), "--graph", args.graph, "--overlap", args.overlap, "--no-cpu-baseline",
                   "--no-families", "--no-live-pmc", "--fused", "--sustain-seconds", "0", "--long-steps", "0"]
            env = dict(os.environ, TMPDIR="/tmp")
            for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK"):
                env.pop(k, None)
            r = subprocess.run(cmd, env=env, cwd="/tmp", capture_output=True, text=True, timeout=150)      # (a pass takes 6-10 s)
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                shutil.rmtree(d, ignore_errors=True)
                return {"error": "rocprofv3 --pmc %s: rc %d, %s" % (c, r.returncode, (r.stderr or "")[-300:])}
            for row in csv.DictReader(open(files[0])):
                if row["Counter_Name"] == c and "anymdp_step_kernel" in row["Kernel_Name"]:
                    agg.setdefault(row["Kernel_Name"].split("(")[0], {}).setdefault(c, []).append(float(row["Counter_Value"]))
            shutil.rmtree(d, ignore_errors=True)
    except Exception as ex:
        return {"error": repr(ex)}
    out = {"source": "live: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate child runs of this command, 128 steps, before the timed "
                     "passes; KB counters, read bytes = 2 x FETCH_SIZE on gfx950)", "kernels": {}}
    for name, cs in agg.items():
        if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
            f, w = sum(cs["FETCH_SIZE"]) / len(cs["FETCH_SIZE"]), sum(cs["WRITE_SIZE"]) / len(cs["WRITE_SIZE"])
            b = (2.0 * f + w) * 1024.0
            out.setdefault(_kernel_kind(name), b)
            out["kernels"][name] = {"bytes_per_launch": b, "dispatches": len(cs["FETCH_SIZE"])}
    out["seconds"] = round(time.perf_counter() - t0, 1)
    if not any(k in out for k in ("plain", "hand", "rollout")):
        return {"error": "no step kernel in the counter files"}
    return out


def roofline_basis(algo, traffic, kern_us, distinct_tasks):
    """-> {bound, achieved, frac, basis}: which bytes `achieved` / `frac` price (never a fraction above 1).
    algo: SURVEY 8(d)'s algorithmic bytes per launch; traffic: PMC bytes per launch or None; kern_us: time per launch."""
    t = kern_us * 1e-6
    survey = algo / t / 1e9
    if not distinct_tasks:
        return {"bound": "cache", "achieved": survey, "frac": None,
                "basis": "shared tasks: the rows are Infinity-Cache / L2 resident; `achieved` prices SURVEY 8(d)'s algorithmic bytes "
                         "as if they were HBM reads, so no fraction of the HBM peak is claimed"}
    if traffic is not None and traffic < algo:
        return {"bound": "hbm", "achieved": traffic / t / 1e9, "frac": traffic / t / 1e9 / HBM_PEAK_GBS,
                "basis": "traffic: HBM bytes the kernel moves per launch (PMC), below SURVEY 8(d)'s algorithmic bytes because the "
                         "search reads one or two 128-byte lines of the 512-byte row; the survey's figure: *_survey_bytes"}
    f = survey / HBM_PEAK_GBS
    return {"bound": "hbm", "achieved": survey, "frac": f if f <= 1.0 else None,
            "basis": "SURVEY 8(d) algorithmic bytes" + ("" if f <= 1.0 else " (above the peak: the search moves fewer bytes than "
                                                        "the survey prices and no PMC profile of this kernel source is committed; frac null)")}


def choose_search(env, torch, args, n_task, S, A):
    """-> (search actually used, GiB of bucket lines, census or None).  auto = the library's AUTO: its census decides
    (AnyMDPVecEnv.set_search("auto", n_bucket=...): probe without allocating, build when AUTO would use the lines and they fit)"""
    want = args.search
    if want == "auto":
        env.set_search("auto", n_bucket=args.buckets)
    elif want == "bucket":
        env.set_search("bucket", n_bucket=args.buckets)
    else:
        env.set_search(want)
    cen = env.bucket_census()
    built = bool(cen["built"])
    return env.effective_search, (cen["bytes"] / 2**30 if built else 0.0), (cen if built else None)


def floor_probe():
    """latency and line-rate floors of a 65,536-lane step measured on this box class (scripts/devtools/floor_probe.py, the
    latest committed profiles/*floor_probe*.json); without one, the round-1 measurements recorded in HISTORY.md 4.1"""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*floor_probe*.json")), reverse=True):
        try:
            d = json.load(open(f))
            return {"empty_launch_us": d["empty_launch_us"], "coop_lines_us": {int(k): v for k, v in d["coop_lines_us"].items()},
                    "random_lines_per_s": d["random_lines_per_s"], "source": os.path.basename(f),
                    "kernel_source_sha16": d.get("kernel_source_sha16")}
        except Exception:
            pass
    return {"empty_launch_us": 2.9, "coop_lines_us": {1: 4.3, 2: 6.75, 3: 9.1}, "random_lines_per_s": 5.0e10,
            "source": "HISTORY.md 4.1 (round-1 box)", "kernel_source_sha16": None}


def make_tables(eng, torch, _lib, n_task, task_base, seed, S=64, A=8, s0_max=4):
    d = eng.device
    words = (S + 63) // 64
    from xenoverse_amd.anymdp import row_lines
    t = dict(S=S, A=A, s0_max=s0_max,
             rows=torch.empty((n_task, S, A, row_lines(S), 16), dtype=torch.float64, device=d),
             state_map=torch.empty((n_task, S), dtype=torch.int32, device=d),
             term_mask=torch.empty((n_task, words), dtype=torch.int64, device=d),
             s0_cdf=torch.empty((n_task, s0_max), dtype=torch.float64, device=d),
             s0_ids=torch.empty((n_task, s0_max), dtype=torch.int32, device=d),
             max_steps=torch.empty(n_task, dtype=torch.int32, device=d))
    _lib.check(eng.lib.xv_anymdp_synth_tasks(
        eng.handle, seed, task_base, n_task, S, A, s0_max,
        *[_lib.ptr(t[k]) for k in ("rows", "state_map", "term_mask", "s0_cdf", "s0_ids", "max_steps")]))
    eng.sync()
    return t


class Watchdog(object):
    """N > 1: a stalled collective (set-up, probe or the all-gather pass) must not hang the job.  When the deadline
    passes, rank 0 prints whatever `emit` has been armed with (the pass-1 measurement, flagged) and EVERY rank leaves
    with a non-zero code — never a re-exec, never a retry in-process."""

    def __init__(self, seconds, rank):
        self.seconds, self.rank = seconds, rank
        self.lock = threading.Lock()
        self.emit = None
        self.timer = None
        self.stage = "start"

    def arm(self, stage):
        self.cancel()
        self.stage = stage
        self.timer = threading.Timer(self.seconds, self._fire)
        self.timer.daemon = True
        self.timer.start()

    def cancel(self):
        if self.timer is not None:
            self.timer.cancel()
            self.timer = None

    def _fire(self):
        msg = "%s did not finish within %d s" % (self.stage, self.seconds)
        try:
            if self.rank == 0 and self.emit is not None:
                self.emit(timeout_note=msg)
            else:
                sys.stderr.write("bench.py rank %d: %s\n" % (self.rank, msg))
                sys.stderr.flush()
        finally:
            os._exit(EXIT_WATCHDOG)


def init_distributed(args, torch, local, wd, cpu_only):
    """-> (dist or None, info dict with rccl / rccl_ranks / note)"""
    import torch.distributed as dist
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    backend = "gloo" if cpu_only else os.environ.get("XV_BENCH_BACKEND", "nccl")   # "nccl" is RCCL on ROCm
    info = {"rccl": False, "rccl_ranks": None, "note": None, "backend": backend}
    wd.arm("process-group set-up (%s)" % backend)
    try:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
            probe = torch.ones(1, device="cuda:%d" % local)
            dist.all_reduce(probe)                          # fail here, not inside the timed region
            torch.cuda.synchronize()
            info["rccl"], info["rccl_ranks"] = True, int(probe.item())
        else:
            dist.init_process_group(backend)
            probe = torch.ones(1)
            dist.all_reduce(probe)
            info["ranks_seen"] = int(probe.item())
    except Exception as ex:   # keep the scaling measurement alive: barrier and MAX over ranks through gloo
        info["note"] = "RCCL unavailable (%r): gloo used for the barrier and the MAX over ranks, no all-gather" % (ex,)
        info["backend"] = "gloo"
        try:
            if dist.is_initialized():
                dist.destroy_process_group()
            dist.init_process_group("gloo")
        except Exception as ex2:
            sys.exit("bench.py: no usable torch.distributed backend: %r / %r" % (ex, ex2))
    wd.cancel()
    return dist, info


def spin_sync(torch, ev):
    """device synchronisation without the wake-up latency of a blocking wait: poll the last event, then synchronise"""
    if ev is not None:
        while not ev.query():
            pass
    if torch.cuda.is_available():
        torch.cuda.synchronize()


def median(xs):
    s = sorted(xs)
    n = len(s)
    return s[n // 2] if n % 2 else 0.5 * (s[n // 2 - 1] + s[n // 2])


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: this process — which has imported neither torch nor anything else
    that initialises the GPU — starts ONE fresh child `python -m torch.distributed.run --nproc-per-node N bench.py <same
    arguments>` (never an exec: a child process, whose output streams through the inherited stdout / stderr) and leaves
    with the child's return code.  The ranks then find WORLD_SIZE in their environment and take the normal path."""
    import socket
    import subprocess
    port = os.environ.get("MASTER_PORT")
    if not port:
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = str(s.getsockname()[1])
        s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    for k in ("RANK", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "GROUP_RANK"):    # a stale single-rank environment must not leak in
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
    sys.stderr.write("bench.py: --gpus %d without a launcher: starting %s\n" % (args.gpus, " ".join(cmd[1:9])))
    sys.stderr.flush()
    sys.stdout.flush()
    proc = subprocess.Popen(cmd, env=env, cwd=ROOT)
    try:
        rc = proc.wait()
    except KeyboardInterrupt:
        proc.terminate()          # the exact child this process started
        rc = proc.wait()
    sys.exit(rc)


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return self_launch(args)
    selftest = args.exchange_selftest
    cpu = None
    if world == 1 and rank == 0 and not args.no_cpu_baseline and not selftest and not args.sweep_envs \
            and not under_profiler():
        if args.workload == "mixed":
            sys.path.insert(0, os.path.join(ROOT, "scripts"))
            import bench_mixed
            try:
                cpu = bench_mixed.cpu_baseline_mixed(args.cpu_seconds, args.seed, usable_cpus())
            except Exception as ex:
                cpu = {"value": None, "unit": "env-steps/s", "cores": usable_cpus(), "kind": "port", "sample": "failed: %r" % (ex,)}
        else:
            cpu = cpu_baseline(args.cpu_seconds, args.seed, args.cpu_table_gib)      # before the GPU is touched

    live = None
    if world == 1 and rank == 0 and not selftest and not args.sweep_envs and not under_profiler() and not args.no_live_pmc \
            and args.workload == "anymdp" and not os.environ.get("XV_BENCH_SHARE_GPU"):
        live = live_pmc_traffic(args)      # child processes; this one has not touched the GPU yet

    import torch
    if os.environ.get("XV_BENCH_SHARE_GPU"):   # functional test of the N>1 path on a 1-GPU box (not a measurement)
        local = 0
        args.overlap = "off"      # one overlapped handle per DEVICE: two processes on one GPU must not both overlap
    if not selftest:
        torch.cuda.set_device(local)
    wd = Watchdog(args.gather_timeout, rank)
    dist, dinfo = None, {"rccl": None, "rccl_ranks": None, "note": None, "backend": None}
    if world > 1:
        dist, dinfo = init_distributed(args, torch, local, wd, cpu_only=selftest)

    if args.sweep_envs:
        return sweep(args, torch, local)

    if args.workload == "mixed":      # BASELINE configs[4] end to end on `world` ranks: its own line
        sys.path.insert(0, os.path.join(ROOT, "scripts"))
        import bench_mixed
        out = bench_mixed.run_mixed(args, torch, dist, dinfo, rank, world, local, wd, selftest=selftest)
        out["cpu_baseline"] = cpu if world == 1 else None
        if selftest:
            out["ranks_seen"] = dinfo.get("ranks_seen")
        if rank == 0:
            print(json.dumps(out), flush=True)
        if dist is not None:
            wd.arm("final barrier")
            dist.barrier()
            dist.destroy_process_group()
            wd.cancel()
        if selftest and rank == 0 and out.get("selftest") != "ok":
            sys.exit(1)
        return

    from xenoverse_amd.distributed import REC_BYTES, RolloutGather, pack_records, unpack_records
    n_env = args.envs
    n_task = args.tasks if args.tasks > 0 else n_env
    S, A, P = 64, 8, min(args.period, max(args.steps, 1))   # a ring no longer than the timed batch: whole cycles can replay
    search = {"auto": "fence"}.get(args.search, args.search)
    bucket_gib, census = 0.0, None
    env = None
    if not selftest:
        from xenoverse_amd import _lib
        from xenoverse_amd.anymdp import AnyMDPVecEnv
        env = AnyMDPVecEnv(n_env, device="cuda:%d" % local, seed=args.seed, env_id_base=rank * n_env,
                           autoreset_mode="same_step")
        tab = make_tables(env.engine, torch, _lib, n_task, rank * n_task, args.seed + 1, S, A)
        per = n_env // n_task
        env_task = (torch.arange(n_env, device=env.device, dtype=torch.int32) // per).contiguous()
        env.set_task(tab, env_task_index=env_task)
        search, bucket_gib, census = choose_search(env, torch, args, n_task, S, A)
        # ring cycles replay from a hipGraph (one submission per `--period` steps): a short timed batch is then one
        # submission (20 steps: 6.4-6.5 vs 7.4 us per step with plain launches), and on long runs the result does not
        # depend on the host's launch rate, which sits close to the 5-us kernel (2,000 steps, graph vs plain launches:
        # 5.19 vs 4.98 us per step on one box of the pool, 5.15 vs 5.35 on another).  `--graph off` issues plain launches
        graph_mode = args.graph if args.graph != "auto" else "on"
        env.set_step_many_graph(graph_mode)
        overlap_requested = args.overlap != "off" and graph_mode == "on"
        env.set_step_many_overlap(overlap_requested)
        device = env.device
        g = torch.Generator(device=device)
        g.manual_seed(args.seed + 17 * rank)
        actions = torch.randint(0, A, (P, n_env), generator=g, device=device, dtype=torch.int32)
        env.reset()
        ring = env.step_many(1, actions)   # allocates the [P, N] output ring

        def step_many(n):
            env.step_many(n, actions, out=ring)
    else:      # fabricated records: a function of (global env id, step) so that the gathered batch can be checked
        device = torch.device("cpu")
        n_env = min(n_env, 4096)
        gid = torch.arange(rank * n_env, (rank + 1) * n_env, dtype=torch.int32)
        tt = torch.arange(P, dtype=torch.int32)[:, None]
        actions = (gid[None, :] + tt) % A

        def fabricate(gid):
            return dict(obs=(gid[None, :] * 3 + tt) % S, reward=gid[None, :].float() * 0.5 + tt.float(),
                        terminated=((gid[None, :] + tt) % 5 == 0).to(torch.uint8),
                        truncated=((gid[None, :] + tt) % 7 == 0).to(torch.uint8))
        ring = fabricate(gid)

        def step_many(n):
            pass

    # exchange step: all-gather of each finished T-step rollout chunk (8-byte records) on a side
    # stream so that it overlaps the next chunk's stepping.  Stepping itself needs no collective.  On GPUs the
    # all-gather runs over RCCL only (gloo would stage 16 MB per rank through the host).
    do_gather = world > 1 and not args.no_allgather and dinfo["note"] is None and \
        (selftest or dist.get_backend() == "nccl" or bool(os.environ.get("XV_BENCH_FORCE_GATHER")))
    gather = None
    gather_note = dinfo["note"] or "none"
    transport = {"used": None, "requested": args.transport, "note": None, "rccl_comm_count": None}
    if do_gather:
        # --transport: "rccl" = the C-ABI's own collective (xv_rccl_* / xv_rollout_allgather: ncclAllGather over librccl,
        # no torch.distributed in the data path), "torch" = all_gather_into_tensor of the process group; auto prefers rccl
        want = args.transport
        if want in ("auto", "rccl"):
            if selftest or device.type != "cuda":
                transport["note"] = "the rccl transport moves device buffers: torch used on CPU"
            else:
                wd.arm("RCCL communicator set-up (xv_rccl_comm_create)")
                try:
                    gather = RolloutGather((P, n_env, REC_BYTES), device=device, transport="rccl", rank=rank, world=world)
                    transport["used"] = "rccl"
                    transport["rccl_comm_count"] = gather.comm.count()
                    gather_note = "all_gather of %d-step rollout chunks, %d B/record (xv_rollout_allgather: ncclAllGather over " \
                                  "librccl through the C-ABI, side stream)" % (P, REC_BYTES)
                except Exception as ex:
                    gather = None
                    transport["note"] = "rccl transport unavailable (%r): torch used" % (ex,)
                wd.cancel()
        if gather is None:
            try:
                gather = RolloutGather((P, n_env, REC_BYTES), device=device)
                transport["used"] = "torch"
                gather_note = "all_gather of %d-step rollout chunks, %d B/record (torch.distributed %s, side stream)" \
                              % (P, REC_BYTES, "nccl = RCCL" if dinfo["rccl"] else dinfo["backend"])
            except Exception as ex:   # never lose the measurement to a collective set-up problem
                gather = None
                gather_note = "all_gather unavailable: %r" % (ex,)

    def run(k_steps, with_gather=False):
        if not with_gather:      # ONE call: the library replays whole ring cycles (and overlaps them when the call is long enough)
            if k_steps > 0:
                step_many(k_steps)
            return
        done = 0
        while done < k_steps:
            n = min(P, k_steps - done)
            step_many(n)
            done += n
            if with_gather and n == P:
                gather.wait()        # the previous chunk must have left before its buffer is repacked
                pack_records(ring["obs"], actions, ring["reward"], ring["terminated"], ring["truncated"],
                             out=gather.local)
                gather.launch()
        if with_gather:
            gather.wait()

    gpu = not selftest

    def barrier(ev=None):
        if gpu:
            spin_sync(torch, ev)
        if dist is not None:
            dist.barrier()
        if gpu:
            torch.cuda.synchronize()

    class _StopEvent:        # the engine's stop event behind the interface spin_sync polls
        def query(self):
            return env.engine.event_done(1)

    def timed_pass(with_gather, repeats, fn=None, warm=None):
        """-> (median wall seconds of a K-step batch, median event ms) after MAX over ranks per repetition.
        The HIP events are the engine's own (xv_engine_event_*: hipEventRecord on the stream the step kernels are
        launched on, ~2 us of host time each instead of ~5 for a torch Event).
        fn / warm: another batch to time / its warm-up (the `long_call` block), bracketed the same way."""
        gc.collect()       # a full collection of this heap takes tens of milliseconds: not inside a 0.1-ms timed region,
        gc.disable()       # and not right in front of it either (the idle GPU clocks down) — before the warm-up
        try:
            run(args.warmup, with_gather) if warm is None else warm()
            walls, evs = [], []
            stop = _StopEvent() if gpu else None
            for _ in range(repeats):
                barrier()
                t0 = time.perf_counter()
                if gpu:
                    env.engine.event_record(0)
                run(args.steps, with_gather) if fn is None else fn()
                if gpu:
                    env.engine.event_record(1)
                barrier(stop)
                walls.append(time.perf_counter() - t0)
                evs.append(env.engine.event_elapsed_ms() if gpu else walls[-1] * 1e3)   # HIP events on the launch stream
        finally:
            gc.enable()    # also when a launch or a collective raised: the rest of the process keeps its collector
        tt = torch.tensor([walls, evs], dtype=torch.float64)
        if dist is not None:              # MAX over ranks, per repetition
            if dist.get_backend() == "nccl":
                tt = tt.to(device)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            tt = tt.cpu()
        return median(tt[0].tolist()), median(tt[1].tolist()), tt[0].tolist()

    R = max(1, args.repeats)
    state = {"done": False, "wall_g": None, "fused": None, "errs": None, "selftest": None}
    report_lock = threading.Lock()

    # pass 1 (the reported value): sharded stepping, no data-path collective — envs are independent
    wall, ev_ms, walls = timed_pass(False, R)
    state["errs"] = env.check_errors() if env is not None else 0
    state["graph"] = 0
    state["overlap"] = 0
    if env is not None and graph_mode == "on" and args.steps >= P and int(env.lib.xv_anymdp_step_many_graph_state(env._h)) >= 0:
        state["graph"] = 1     # whole ring cycles were replayed (a remainder shorter than the ring is plain launches)
        state["overlap"] = 1 if env.step_many_overlap_state == 1 else 0

    # the other searches on the same workload (`value` above is the AUTO choice): each timed with min(R, 5) repetitions
    variants = None
    if env is not None and world == 1 and not args.no_variants and not under_profiler():   # (N = 1: no collective in here)
        def vrow(w, e, is_value):      # both clocks side by side: events (us_per_step, env_steps_per_s) and host wall (wall_*)
            return {"us_per_step": e * 1e3 / args.steps, "env_steps_per_s": n_env * args.steps / (e * 1e-3),
                    "wall_us_per_step": w * 1e6 / args.steps, "wall_env_steps_per_s": n_env * args.steps / w, "is_value": is_value}
        variants = {"auto_choice": search, "clock": "us_per_step / env_steps_per_s: HIP events; wall_*: host wall (what `value` uses)",
                    search: vrow(wall, ev_ms, True)}
        for name in ("fence", "bucket"):
            if name == search:
                continue
            try:
                if name == "bucket":
                    need = n_task * S * A * args.buckets * 128
                    free, _ = torch.cuda.mem_get_info()
                    if not env.bucket_census()["built"] and need + (8 << 30) > free:
                        variants[name] = {"skipped": "the bucket lines (%.0f GiB) do not fit the free HBM" % (need / 2**30)}
                        continue
                    env.set_search("bucket", n_bucket=args.buckets)
                else:
                    env.set_search("fence")
                w, e, _ = timed_pass(False, max(1, min(R, 5)))
                variants[name] = vrow(w, e, False)
            except Exception as ex:
                variants[name] = {"error": repr(ex)}
        env.set_search("auto") if args.search == "auto" else env.set_search(args.search, n_bucket=args.buckets) \
            if args.search == "bucket" else env.set_search(args.search)
        if census is None and env.bucket_census()["built"]:
            census = env.bucket_census()
        state["errs"] |= env.check_errors()
        # the same search issued the other way: one stream (each launch waits for the one before) vs overlapped
        if state.get("overlap") == 1 or args.overlap != "off":
            try:
                lv = {"overlapped" if state.get("overlap") == 1 else "one stream": vrow(wall, ev_ms, True)}
                other = state.get("overlap") != 1
                env.set_step_many_overlap(other)
                w, e, _ = timed_pass(False, max(1, min(R, 5)))
                took = env.step_many_overlap_state == 1
                lv["overlapped" if took else ("one stream" if not other else "one stream (overlap not taken: short call)")] = vrow(w, e, False)
                env.set_step_many_overlap(overlap_requested)      # what was REQUESTED (round 5 restored `not other`: off)
                variants["launch"] = lv
            except Exception as ex:
                variants["launch"] = {"error": repr(ex)}
            state["errs"] |= env.check_errors()

    # `long_call`: SURVEY.md 8(d)'s own definition of the metric inside this line — >= 1,000 (here >= 2,000) back-to-back step
    # launches after >= 100 warm-up steps — for the three ways the library issues open-loop steps of the SAME envs and tables:
    # one stream (every launch behind the one before), overlapped (consecutive launches on up to three streams, per-wave
    # hand-off) and the fused roll-out (one launch per ring cycle).  All three take their actions from a pre-filled ring and are
    # bit-equal (tests/test_gpu_chains.py, test_fused_rollout_equals_stepwise); both clocks, the library's own state words and
    # the device error flags read back after each.  `value` / `ms_per_step` stay the `--steps` burst.
    long_call = None
    PL = max(2, args.period - args.period % 2)
    KL = (max(args.long_steps, PL) + PL - 1) // PL * PL if args.long_steps > 0 else 0
    ring_l = actions_l = None
    if env is not None and KL > 0 and not under_profiler():
        try:
            g2 = torch.Generator(device=device)
            g2.manual_seed(args.seed + 17 * rank + 5)
            actions_l = actions if P == PL else torch.randint(0, A, (PL, n_env), generator=g2, device=device, dtype=torch.int32)
            ring_l = ring if P == PL else env.step_many(1, actions_l)
            WL = (max(100, args.warmup) + PL - 1) // PL * PL
            RL = max(1, args.long_repeats)

            def lrow(w, e, launches, note):      # keys: LONG_CALL_ROW_KEYS
                return {"steps": KL, "warmup": WL, "repeats": RL, "launches_per_call": launches,
                        "us_per_step": e * 1e3 / KL, "env_steps_per_s_events": world * n_env * KL / (e * 1e-3),
                        "wall_us_per_step": w * 1e6 / KL, "env_steps_per_s": world * n_env * KL / w,
                        "overlap_state": env.step_many_overlap_state,
                        "graph_state": int(env.lib.xv_anymdp_step_many_graph_state(env._h)),
                        "device_error_flags": env.check_errors(), "issue": note}
            long_call = {"definition": "SURVEY.md 8(d): hipEvent / host wall around ONE call of `steps` back-to-back vector steps "
                                       "after `warmup` warm-up steps, median of `repeats`, MAX over ranks; ring period %d; same envs, "
                                       "tables, search (%s) and actions ring for the three issue modes" % (PL, search),
                         "clock": "us_per_step / env_steps_per_s_events: HIP events on the launch stream; wall_us_per_step / "
                                  "env_steps_per_s: host wall incl. the closing synchronise"}
            for name, ov in (("one_stream", False), ("overlapped", True)):
                if ov and args.overlap == "off":
                    long_call[name] = {"skipped": "--overlap off"}
                    continue
                env.set_step_many_overlap(ov)
                w, e, _ = timed_pass(False, RL, fn=lambda: env.step_many(KL, actions_l, out=ring_l),
                                     warm=lambda: env.step_many(WL, actions_l, out=ring_l))
                long_call[name] = lrow(w, e, KL, "xv_anymdp_step_many, one step kernel per vector step, ring cycles replayed from "
                                       + ("cycle graphs on up to three HIP streams" if ov else "a hipGraph on the engine's stream"))
            env.set_step_many_overlap(False)

            def fused():
                for _ in range(KL // PL):
                    env.rollout(actions_l, out=ring_l)
            w, e, _ = timed_pass(False, RL, fn=fused, warm=lambda: [env.rollout(actions_l, out=ring_l) for _ in range(WL // PL)])
            long_call["fused_rollout"] = lrow(w, e, KL // PL, "xv_anymdp_rollout: ONE launch per ring cycle of %d steps (open loop, "
                                              "the same input contract as step_many)" % PL)
            env.set_step_many_overlap(overlap_requested)
        except Exception as ex:
            long_call = dict(long_call or {}, error=repr(ex))
            try:
                env.set_step_many_overlap(overlap_requested)
            except Exception:
                pass
        state["errs"] |= env.check_errors()

    # untimed sustained stepping: a monitor that samples the GPU every few seconds sees it busy (the timed passes are
    # milliseconds).  Excluded from `value`; `sustain_s` says how long it ran.  Issued the way `long_call.overlapped` (or, with
    # --overlap off, `.one_stream`) is: calls of `long_call` length on the long ring, so that the two figures can be compared.
    sustain = None
    if env is not None and args.sustain_seconds > 0 and not under_profiler():
        ks, al, rl = (KL, actions_l, ring_l) if (KL > 0 and ring_l is not None) else (8 * P, actions, ring)
        t0 = time.perf_counter()
        done = calls = 0
        while time.perf_counter() - t0 < args.sustain_seconds:
            env.step_many(ks, al, out=rl)
            done += ks
            calls += 1
            if calls % 8 == 0:
                torch.cuda.synchronize()       # bounds the launch queue
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        sustain = {"sustain_s": dt, "steps": done, "steps_per_call": ks, "env_steps_per_s_rank0": n_env * done / dt,
                   "us_per_step": dt * 1e6 / done, "overlap_state": env.step_many_overlap_state,
                   "overlap_requested": bool(overlap_requested)}
        state["errs"] |= env.check_errors()

    def report(timeout_note=None):
        with report_lock:
            if rank != 0 or state["done"]:
                return
            state["done"] = True
            total_steps = world * n_env * args.steps
            kern_us = ev_ms * 1e3 / args.steps
            algo = ALGO_BYTES_PER_ENV_STEP[8] * n_env
            achieved = algo / (kern_us * 1e-6) / 1e9
            traffic_c, traffic_src_c = (None, None) if selftest else pmc_traffic(n_env, n_task, search, bool(state.get("overlap")))
            traffic, traffic_src = traffic_c, traffic_src_c

            def live_bytes(kind):      # this run's own counters, where the child runs delivered them
                if not live or "error" in live:
                    return None
                v = live.get(kind)
                if v is None and kind in ("hand", "plain"):
                    v = live.get("plain" if kind == "hand" else "hand")
                return v
            lb = live_bytes("hand" if state.get("overlap") else "plain")
            if lb is not None:
                traffic, traffic_src = lb, live["source"]
            floor = floor_probe()
            lines = {"bucket": 1, "fence": 2}.get(search)
            # the bare chain was measured on tables that miss every cache: it is the floor of config 2a (one task per env),
            # not of shared tasks whose lines are cache hits (2b runs under it)
            floor_us = floor["coop_lines_us"].get(lines) if (lines and n_task == n_env) else None
            exchange = gather_note if timeout_note is None else gather_note + "; " + timeout_note
            roof = roofline_basis(algo, traffic, kern_us, n_task == n_env)
            out = {
                "metric": "env-steps/sec (whole node), anymdp |S|=64 |A|=8, 65k envs/GPU",
                "value": total_steps / wall, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps,
                "warmup": args.warmup, "repeats": R, "ms_per_step": wall * 1e3 / args.steps, "higher_is_better": True,
                "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                "timing": {"statistic": "median over `repeats` repetitions of the `steps`-step batch, each bracketed by "
                                        "barrier + device sync, MAX over ranks per repetition",
                           "wall_ms_min": min(walls) * 1e3, "wall_ms_median": wall * 1e3, "wall_ms_max": max(walls) * 1e3},
                "config": {"workload": "anymdp S=64 A=8, %d envs/GPU, %d tasks/GPU (%s), fp64 CDF rows, "
                                       "SAME_STEP auto-reset, random actions"
                                       % (n_env, n_task, "2a distinct: one task per env" if n_task == n_env
                                          else "2b shared"),
                           "envs_per_gpu": n_env, "tasks_per_gpu": n_task, "S": S, "A": A,
                           "table_gib_per_gpu": round(n_task * S * A * (1 + (S + 6) // 7) * 128 / 2**30, 2),
                           "bucket_lines_gib_per_gpu": round(bucket_gib, 2),
                           "launch": "one step kernel per vector step (xv_anymdp_step_many%s)"
                                     % ((", ring cycles of %d steps replayed from cycle graphs on up to three HIP streams (three where "
                                         "three launches fit on the device together: steps k, k + 1, k + 2 in flight): consecutive "
                                         "steps overlap, every wave takes its 64 envs over from the same wave of the step before "
                                         "through the env records (xv_anymdp_set_step_many_overlap)" % P)
                                        if state.get("overlap") == 1 else
                                        (", ring cycles of %d steps replayed from a hipGraph" % P if state.get("graph") == 1 else
                                         ", plain launches")),
                           "overlap": bool(state.get("overlap")), "overlap_requested": args.overlap,
                           "search": search, "search_requested": args.search,
                           "bucket_census": None if census is None else {k: census[k] for k in (
                               "n_bucket", "cuts_per_line", "lines_dirty", "p_fallback", "fallbacks_per_launch", "auto_limit", "auto_uses_bucket")},
                           "exchange": exchange, "device_error_flags": state["errs"]},
                # `frac` is never printed above 1 (a fraction above 1 is a wrong bound, not a fast kernel): SURVEY 8(d) prices a
                # step at the 512-byte CDF row + 50 B; the bucket / fence search reads one or two 128-byte lines of it, so when the
                # PMC traffic of this kernel source is below the algorithmic bytes, `achieved` / `frac` are the bytes really moved
                # (HBM busy) and the survey's figure stays beside them as `achieved_survey_bytes` / `frac_survey_bytes`; shared
                # tasks (2b) are served by the Infinity Cache: no HBM fraction is claimed for them (`frac` null)
                "roofline": dict(roof, **{"peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "achieved_survey_bytes": achieved, "frac_survey_bytes": achieved / HBM_PEAK_GBS,
                             # the survey fraction on the line's own wall clock: algorithmic bytes / ms_per_step / peak (host wall
                             # around barrier + sync; the other fields use the HIP-event time `avg_launch_us`)
                             "frac_wall_survey_bytes": algo / (wall / args.steps) / 1e9 / HBM_PEAK_GBS,
                             "frac_wall": None if roof["frac"] is None else roof["frac"] * kern_us / (wall * 1e6 / args.steps),
                             "clock": "HIP events on the launch stream (avg_launch_us); frac_wall*: host wall (ms_per_step)",
                             "traffic": traffic, "traffic_source": traffic_src,
                             # the committed profile of the same kernel source (profiles/*pmc_traffic*.json) beside this run's own
                             "traffic_committed_profile": traffic_c, "traffic_committed_source": traffic_src_c,
                             "traffic_live": (None if not live else ({"error": live["error"]} if "error" in live else
                                                                     {k: live.get(k) for k in ("plain", "hand", "rollout", "seconds")})),
                             "frac_traffic": None if traffic is None else traffic / (kern_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                             "traffic_over_algorithmic": None if traffic is None else traffic / algo,
                             # the step is a latency chain of `dependent_lines` random lines: `floor_us` is that chain measured
                             # bare on this box class (scripts/devtools/floor_probe.py), `lines_per_s` against the measured
                             # random-line rate
                             "primary": "frac" if roof["frac"] is not None else ("frac_of_floor" if floor_us is not None else None),
                             "dependent_lines": lines, "floor_us": floor_us, "frac_of_floor": None if floor_us is None else floor_us / kern_us,
                             "empty_launch_us": floor["empty_launch_us"], "floor_source": floor["source"],
                             # the floors are a microbenchmark of the box class, not of the kernel; the probe records the kernel
                             # source it was taken beside, so a reader sees whether the two belong to one tree
                             "floor_kernel_source_sha16": floor.get("kernel_source_sha16"),
                             "floor_kernel_source_current": (None if selftest else floor.get("kernel_source_sha16") == kernel_source_hash()),
                             "lines_per_s": None if lines is None else lines * n_env / (kern_us * 1e-6),
                             "random_line_rate": floor["random_lines_per_s"],
                             "frac_of_line_rate": None if lines is None else lines * n_env / (kern_us * 1e-6) / floor["random_lines_per_s"],
                             "kernel": "anymdp_step_kernel<false, %d, false, %s, %d%s>  (INJECT, blocks per fence entry | "
                                       "0 = binary search, ROLLOUT, TICKDEV, BK = 0 | bucket-line packing 1, 2, HAND = overlapped launches)"
                                       % (0 if search == "binary" else 1, "true" if state.get("graph") == 1 else "false",
                                          (census or {}).get("format", 1) if search == "bucket" else 0,
                                          ", true" if state.get("overlap") == 1 else ""),
                             "avg_launch_us_note": ("overlapped launches: time per launch in steady state = timed region / launches; two or "
                                                    "three launches are in flight, so a profiler's per-kernel duration (start of dispatch to "
                                                    "end, the wait for the step before included) is a multiple of this") if state.get("overlap") == 1 else None,
                             "kernel_source_sha16": None if selftest else kernel_source_hash(),
                             "avg_launch_us": kern_us, "algorithmic_bytes_per_launch": algo}),
                "rccl": dinfo["rccl"],
                "rccl_ranks": transport["rccl_comm_count"] if transport["rccl_comm_count"] is not None else dinfo["rccl_ranks"],
                "transport": transport["used"], "transport_requested": transport["requested"],
                "transport_note": transport["note"], "rccl_ranks_all_reduce": dinfo["rccl_ranks"],
                "allgather_timeout": timeout_note is not None,
            }
            if selftest:
                out["mode"] = "exchange-selftest: fabricated CPU records, no stepping — NOT a measurement"
                out["value"] = out["ms_per_step"] = None
                out["roofline"] = None
                out["selftest"] = state["selftest"]
                out["ranks_seen"] = dinfo.get("ranks_seen")
            if state["fused"] is not None:
                out["fused_rollout_env_steps_per_s_rank0"] = state["fused"]
            if state["wall_g"] is not None:
                chunks = args.steps // P
                out["with_allgather"] = {"value": total_steps / state["wall_g"], "unit": "env-steps/s",
                                         "gathered_GB_per_s_per_rank":
                                             chunks * P * n_env * REC_BYTES * (world - 1) / state["wall_g"] / 1e9}
            if variants is not None:
                out["search_variants"] = variants
            if long_call is not None:
                for name, ov in (("one_stream", False), ("overlapped", True), ("fused_rollout", None)):
                    row = long_call.get(name)
                    if isinstance(row, dict) and "us_per_step" in row:
                        if ov is None:      # the fused roll-out: bytes per LAUNCH of `period` steps (the child runs use --period too)
                            tr, src = pmc_traffic(n_env, n_task, search, kind="rollout")
                            if live_bytes("rollout") is not None:
                                tr, src = live_bytes("rollout"), live["source"]
                            tr = None if tr is None else tr / PL
                        else:
                            tr, src = pmc_traffic(n_env, n_task, search, ov)
                            if live_bytes("hand" if ov else "plain") is not None:
                                tr, src = live_bytes("hand" if ov else "plain"), live["source"]
                        rb = roofline_basis(algo, tr, row["us_per_step"], n_task == n_env)
                        row["roofline"] = dict(rb, peak=HBM_PEAK_GBS, unit="GB/s", traffic=tr, traffic_source=src,
                                               frac_survey_bytes=algo / (row["us_per_step"] * 1e-6) / 1e9 / HBM_PEAK_GBS)
                if sustain is not None:
                    ref = long_call.get("overlapped" if sustain.get("overlap_state") == 1 else "one_stream")
                    if isinstance(ref, dict) and ref.get("env_steps_per_s"):
                        long_call["sustain_over_long_call"] = sustain["env_steps_per_s_rank0"] * world / ref["env_steps_per_s"]
                out["long_call"] = long_call
            if sustain is not None:
                out.update(sustain_s=sustain["sustain_s"], sustain=sustain)
            out["cpu_baseline"] = cpu if world == 1 else None      # the CPU lines are measured at N = 1 only
            if state.get("families") is not None:
                out["families"] = state["families"]
            print(json.dumps(out), flush=True)

    if args.fused and env is not None:
        T = P
        env.rollout(actions)
        torch.cuda.synchronize()
        f0, f1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = max(1, args.steps // T)
        f0.record()
        for _ in range(reps):
            env.rollout(actions, out=ring)
        f1.record()
        torch.cuda.synchronize()
        state["fused"] = n_env * T * reps / (f0.elapsed_time(f1) * 1e-3)

    # N = 1: configs 3, 4 and the per-GPU share of config 5 beside the headline, in the same JSON line (`families`)
    if world == 1 and env is not None and not args.no_families and not args.sweep_envs:
        try:
            env.close()
            env = None
            del tab, ring, actions
            torch.cuda.empty_cache()
            sys.path.insert(0, os.path.join(ROOT, "scripts"))
            import bench_families
            state["families"] = bench_families.quick_families()
        except Exception as ex:
            state["families"] = {"error": repr(ex)}

    if world > 1 and env is not None and not args.no_families and not selftest:
        # N > 1: BASELINE configs[4] on the same ranks (the N = 1 line carries one GPU's share as `families.mixed_share`)
        try:
            sys.path.insert(0, os.path.join(ROOT, "scripts"))
            import bench_mixed
            margs = argparse.Namespace(**dict(vars(args), steps=max(args.steps, 64), repeats=max(1, min(R, 5))))
            # one handle per device may hold the overlap switch: `env` hands it to the mixed share for the time of its line
            env.set_step_many_overlap(False)
            state["families"] = {"mixed": bench_mixed.run_mixed(margs, torch, dist, dinfo, rank, world, local, wd)}
        except Exception as ex:
            state["families"] = {"mixed": {"error": repr(ex)}}
        try:
            env.set_step_many_overlap(overlap_requested)
        except Exception:
            pass

    if gather is not None:      # pass 2 (N > 1): every finished rollout chunk all-gathered to all ranks, overlapped
        wd.emit = report
        wd.arm("the all-gather pass")
        try:
            if os.environ.get("XV_BENCH_TEST_STALL") and rank == int(os.environ.get("XV_BENCH_TEST_STALL_RANK", "1")):     # watchdog test: one rank never joins
                time.sleep(10 * args.gather_timeout)
            state["wall_g"] = timed_pass(True, max(1, min(R, 5)))[0]
            if selftest:      # every rank sees every shard, rank order == env order
                run(P, True)
                full = torch.cat([gather.out[r] for r in range(world)], dim=1)
                o2, a2, r2, te2, tr2 = unpack_records(full)
                ref = fabricate(torch.arange(0, world * n_env, dtype=torch.int32))
                ok = torch.equal(o2, ref["obs"]) and torch.equal(r2, ref["reward"]) and \
                    torch.equal(te2, ref["terminated"]) and torch.equal(tr2, ref["truncated"]) and \
                    torch.equal(a2, (torch.arange(0, world * n_env, dtype=torch.int32)[None, :] + tt) % A)
                flag = torch.tensor([1.0 if ok else 0.0])
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                state["selftest"] = "ok" if flag.item() == 1.0 else "MISMATCH"
        except Exception as ex:
            gather_note += "; failed: %r" % (ex,)
        wd.cancel()
    report()
    if env is not None:
        env.close()
    if dist is not None:
        wd.arm("final barrier")
        dist.barrier()
        dist.destroy_process_group()
        wd.cancel()
    if selftest and rank == 0 and state["selftest"] != "ok":
        sys.exit(1)


def sweep(args, torch, local):
    """2a step time against the batch size: where the latency floor of the three dependent levels stops mattering"""
    from xenoverse_amd import _lib
    from xenoverse_amd.anymdp import AnyMDPVecEnv
    S, A, P = 64, 8, args.period
    rows = []
    for n_env in [int(x) for x in args.sweep_envs.split(",")]:
        need = n_env * S * A * (1 + (S + 6) // 7) * 128
        free, _ = torch.cuda.mem_get_info()
        if need + (4 << 30) > free:
            rows.append({"envs": n_env, "skipped": "needs %.0f GiB of rows, %.0f GiB free" % (need / 2**30, free / 2**30)})
            continue
        env = AnyMDPVecEnv(n_env, device="cuda:%d" % local, seed=args.seed, autoreset_mode="same_step")
        tab = make_tables(env.engine, torch, _lib, n_env, 0, args.seed + 1, S, A)
        env.set_task(tab, env_task_index=torch.arange(n_env, device=env.device, dtype=torch.int32))
        used, _, _ = choose_search(env, torch, args, n_env, S, A)
        # ring cycles replayed from a hipGraph, as the headline run: plain launches follow the HOST's launch rate (3-5 us per
        # launch depending on the box), which is all a sweep below 65,536 envs would then show
        env.set_step_many_graph(args.graph if args.graph != "auto" else "on")
        env.set_step_many_overlap(args.overlap != "off")
        g = torch.Generator(device=env.device)
        g.manual_seed(args.seed)
        actions = torch.randint(0, A, (P, n_env), generator=g, device=env.device, dtype=torch.int32)
        env.reset()
        ring = env.step_many(args.warmup, actions)
        torch.cuda.synchronize()
        us = []
        for _ in range(max(3, args.repeats // 5)):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            env.step_many(args.steps, actions, out=ring)
            e1.record()
            torch.cuda.synchronize()
            us.append(e0.elapsed_time(e1) * 1e3 / args.steps)
        t = median(us)
        fus = None
        try:
            env.rollout(actions, out=ring)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(8):
                env.rollout(actions, out=ring)
            e1.record()
            torch.cuda.synchronize()
            fus = e0.elapsed_time(e1) * 1e3 / (8 * P)
        except Exception:
            pass
        algo = ALGO_BYTES_PER_ENV_STEP[8] * n_env
        floor = floor_probe()
        lines = 1 if used == "bucket" else 2
        # bytes the search really moves per env-step (PMC, profiles/r04_z_pmc_traffic_*): one 128-byte line + 88 B of streams
        # (bucket) or two lines (fence).  SURVEY 8(d)'s `frac` prices the 512-byte row: beyond ~100k envs a search that reads
        # one line of it runs "faster than reading the rows would allow" (frac > 1) — the fractions to read are the line rate's
        # and, up to 65,536 envs, the latency floor's
        moved = (216.0 if used == "bucket" else 344.5) * n_env
        frac = algo / (t * 1e-6) / 1e9 / HBM_PEAK_GBS
        rows.append({"envs": n_env, "table_gib": round(need / 2**30, 1), "search": used, "us_per_step": t,
                     "env_steps_per_s": n_env / (t * 1e-6),
                     "random_128B_lines_per_s": lines * n_env / (t * 1e-6),
                     "frac_of_line_rate": lines * n_env / (t * 1e-6) / floor["random_lines_per_s"],
                     "floor_us_at_65536": floor["coop_lines_us"].get(lines), "frac_of_floor": (floor["coop_lines_us"].get(lines) / t) if n_env == 65536 else None,   # the chain was measured at 65,536 lanes
                     "algorithmic_GBs": algo / (t * 1e-6) / 1e9, "frac": frac,
                     "frac_traffic": moved / (t * 1e-6) / 1e9 / HBM_PEAK_GBS,
                     "primary": "frac_traffic",
                     "frac_note": ("frac > 1: the search reads %d line(s) of the row, not the 512 bytes SURVEY 8(d) prices" % lines) if frac > 1 else None,
                     "fused_rollout_us_per_step": fus, "overlapped": env.step_many_overlap_state == 1,
                     "device_error_flags": env.check_errors()})
        env.close()
        del tab, env, ring, actions
        torch.cuda.empty_cache()
    out = {"what": "anymdp 2a (one task per env, S=64, A=8, the search AUTO picks, ring cycles replayed from a hipGraph unless --graph off): step time against envs/GPU",
           "steps": args.steps, "warmup": args.warmup, "kernel_source_sha16": kernel_source_hash(), "rows": rows}
    os.makedirs(os.path.dirname(args.sweep_out), exist_ok=True)
    json.dump(out, open(args.sweep_out, "w"), indent=1)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
