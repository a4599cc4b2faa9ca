"""Sub-batch views (xv_anymdp_view) and K interleaved chains (xv_anymdp_step_many_chains) — GPU tests.

The reference steps one env object per call and its batched loop iterates independent envs
(anymdp/anymdp_env.py:92-132, anymdp/test_utils.py:42-60): the envs of a vector step may be stepped as K sub-batches
in any interleaving.  The contract checked here is bit-equality with the one-chain path (outputs of every ring slot,
the env records, the launch tick) — the oracle parity of the one-chain path is tests/test_gpu_anymdp.py's business.
"""
import ctypes as C

import numpy as np
import pytest
import torch

import oracle
from xenoverse_amd import _lib
from xenoverse_amd.anymdp import AnyMDPVecEnv, to_blocked

pytestmark = pytest.mark.gpu


def _np(t):
    return t.detach().cpu().numpy() if torch.is_tensor(t) else np.asarray(t)


def _dev_tables(tab, dev="cuda:0"):
    out = dict(S=tab["S"], A=tab["A"], s0_max=tab["s0_max"])
    tab = dict(tab, rows=to_blocked(tab["cdf"], tab["rs"]))
    for k in ("rows", "state_map", "term_mask", "s0_cdf", "s0_ids", "max_steps"):
        v = np.ascontiguousarray(tab[k])
        if v.dtype == np.uint64:
            v = v.view(np.int64)
        out[k] = torch.from_numpy(v).to(dev)
    return out


def _run_many(tab, n, P, acts_np, plan, search, chains, how, graph, seed=77, overlap=False, mode="same_step", expect_state=None,
              states=None):
    """-> list of snapshots: the ring after every step_many of `plan`, then (state, steps, need_reset, tick)"""
    env = AnyMDPVecEnv(n, seed=seed, autoreset_mode=mode, bucket_lines="off")
    env.set_task(_dev_tables(tab))
    if search == "bucket":
        env.set_search("bucket", n_bucket=16)
    else:
        env.set_search(search)
    env.set_step_many_graph(graph)
    env.set_step_many_overlap(overlap)
    env.reset()
    acts = torch.as_tensor(acts_np, device=env.device)
    rec, ring = [], None
    for n_steps in plan:
        ring = env.step_many(n_steps, acts, out=ring, chains=chains, how=how)
        torch.cuda.synchronize()
        rec.append({k: _np(v).copy() for k, v in ring.items()})
    s, st, nr = env.get_state()
    rec.append({"state": _np(s), "steps": _np(st), "need_reset": _np(nr), "tick": np.asarray(env.engine.tick)})
    assert env.check_errors() == 0
    if states is not None:
        states.append(env.step_many_overlap_state)
    if overlap:
        want = expect_state if expect_state is not None else (1 if (search != "binary" and P % 2 == 0 and plan[-1] >= 64) else 0)
        assert env.step_many_overlap_state == want
    env.close()
    return rec


def _same(a, b):
    assert len(a) == len(b)
    for x, y in zip(a, b):
        for k in x:
            assert np.array_equal(x[k], y[k]), k


@pytest.mark.parametrize("search", ["fence", "bucket", "binary"])
@pytest.mark.parametrize("how", ["streams", "graph"])
@pytest.mark.parametrize("chains", [2, 4, 8])
def test_k_chains_equal_one_chain_small(search, how, chains):
    """whole cycles, a remainder, a second call on the same rings (cached graphs, ticks re-synchronised)"""
    tab = oracle.anymdp_synth(seed=12, task_index_base=0, n_task=16, S=64, A=8, s0_max=4)
    n, P = 2048, 8
    acts = np.random.RandomState(5).randint(0, 8, (P, n)).astype(np.int32)
    plan = [3 * P + 5, 2 * P, 3, P]
    ref = _run_many(tab, n, P, acts, plan, search, 1, "streams", True)
    got = _run_many(tab, n, P, acts, plan, search, chains, how, True)
    _same(ref, got)
    assert ref[0]["terminated"].sum() > 50


@pytest.mark.parametrize("search", ["fence", "bucket", "binary"])
@pytest.mark.parametrize("mode", ["same_step", "next_step"])
def test_overlapped_step_many_equals_plain_launches_small(search, mode):
    """overlap mode (two streams, per-wave hand-off words): whole cycles, remainders, repeated calls on the same rings, a
    call on other rings (graphs rebuilt), ordinary steps in between (hand-off words re-initialised) — against plain
    launches; the per-lane search and odd periods take the ordinary path"""
    tab = oracle.anymdp_synth(seed=12, task_index_base=0, n_task=16, S=64, A=8, s0_max=4)
    n, P = 2000, 8                       # a ragged last workgroup
    acts = np.random.RandomState(5).randint(0, 8, (P, n)).astype(np.int32)
    plan = [3 * P + 5, 9 * P + 3, 3, 2 * P, 8 * P]      # calls of >= 64 steps are overlapped (whole cycles), the rest ordinary
    ref = _run_many(tab, n, P, acts, plan, search, 1, "streams", False, mode=mode)
    got = _run_many(tab, n, P, acts, plan, search, 1, "streams", True, overlap=True, mode=mode)
    _same(ref, got)
    assert ref[0]["terminated"].sum() > 50


def test_overlapped_step_many_odd_period_and_other_rings():
    tab = oracle.anymdp_synth(seed=14, task_index_base=0, n_task=8, S=64, A=8, s0_max=4)
    n = 1024
    for P in (7, 6):
        acts = np.random.RandomState(P).randint(0, 8, (P, n)).astype(np.int32)
        res = []
        for overlap in (False, True):
            env = AnyMDPVecEnv(n, seed=3, autoreset_mode="same_step")
            env.set_task(_dev_tables(tab))
            env.set_step_many_overlap(overlap)
            env.reset()
            a = torch.as_tensor(acts, device=env.device)
            r1 = env.step_many(11 * P + 1, a)
            r2 = env.step_many(12 * P, a)            # other rings: the cycle graphs are rebuilt
            o = env.step(acts[0])                    # an ordinary step in between
            r3 = env.step_many(2 * P, a, out=r1)
            torch.cuda.synchronize()
            res.append([_np(v).copy() for r in (r1, r2, r3) for v in r.values()] + [_np(o[0]), _np(env.get_state()[0])])
            assert env.check_errors() == 0
            if overlap:      # the last call is a short one: ordinary path
                assert env.step_many_overlap_state == 0
            env.close()
        for x, y in zip(*res):
            assert np.array_equal(x, y)


@pytest.mark.parametrize("search", ["bucket", "fence"])
def test_overlapped_step_many_at_65536_envs(search):
    """the headline batch size: 5 cycles of 32 + 9 steps, twice, every output, the env records and the tick"""
    tab = oracle.anymdp_synth(seed=21, task_index_base=0, n_task=1024, S=64, A=8, s0_max=4)
    n, P = 65536, 32
    acts = np.random.RandomState(7).randint(0, 8, (P, n)).astype(np.int32)
    plan = [5 * P + 9, 4 * P]
    ref = _run_many(tab, n, P, acts, plan, search, 1, "streams", True)
    got = _run_many(tab, n, P, acts, plan, search, 1, "streams", True, overlap=True)
    _same(ref, got)


@pytest.mark.parametrize("how", ["streams", "graph"])
def test_k_chains_with_plain_launches(how):
    """step_many graph mode off: the chains issue plain launches on their streams (how = streams) / the parent's own
    launches (how = graph falls back)"""
    tab = oracle.anymdp_synth(seed=13, task_index_base=0, n_task=8, S=64, A=8, s0_max=4)
    n, P = 1024, 4
    acts = np.random.RandomState(6).randint(0, 8, (P, n)).astype(np.int32)
    plan = [2 * P + 1, P]
    ref = _run_many(tab, n, P, acts, plan, "fence", 1, "streams", False)
    got = _run_many(tab, n, P, acts, plan, "fence", 4, how, False)
    _same(ref, got)


@pytest.mark.parametrize("search", ["bucket", "fence"])
@pytest.mark.parametrize("how,chains", [("streams", 2), ("streams", 4), ("streams", 8), ("graph", 4)])
def test_k_chains_equal_one_chain_at_65536_envs(search, how, chains):
    """the headline batch size (65,536 envs; 1,024 shared tasks so that the tables fit any box): every output of a
    32-slot ring over 3 cycles + 7 steps, the env records and the tick, bit for bit"""
    tab = oracle.anymdp_synth(seed=21, task_index_base=0, n_task=1024, S=64, A=8, s0_max=4)
    n, P = 65536, 32
    acts = np.random.RandomState(7).randint(0, 8, (P, n)).astype(np.int32)
    plan = [3 * P + 7]
    ref = _run_many(tab, n, P, acts, plan, search, 1, "streams", True)
    got = _run_many(tab, n, P, acts, plan, search, chains, how, True)
    _same(ref, got)
    assert ref[0]["terminated"].sum() > 1000 and ref[0]["truncated"].sum() >= 0


def test_view_steps_equal_the_parents_slice():
    """sub-batches made by split(): step(), reset(), get_state() of view k == the parent's slice (same seed, same tick),
    in either order of the sub-batches; the views share the env records with the parent"""
    tab = oracle.anymdp_synth(seed=31, task_index_base=0, n_task=8, S=64, A=8, s0_max=4)
    n, K, T = 1024, 4, 24
    acts = np.random.RandomState(8).randint(0, 8, (T, n)).astype(np.int32)

    whole = AnyMDPVecEnv(n, seed=5, autoreset_mode="same_step")
    whole.set_task(_dev_tables(tab))
    obs0, _ = whole.reset()
    ref = [whole.step(acts[t]) for t in range(T)]
    ref = [tuple(_np(x) for x in r[:4]) + (_np(r[4]["reward_gt"]), _np(r[4]["final_obs"]), _np(r[4]["steps"])) for r in ref]
    s_ref = [_np(x) for x in whole.get_state()]
    whole.close()

    env = AnyMDPVecEnv(n, seed=5, autoreset_mode="same_step")
    env.set_task(_dev_tables(tab))
    env.reset()
    subs = env.split(K)
    assert env.split(K) is subs and len(subs) == K and all(s.num_envs == n // K for s in subs)
    per = n // K
    for t in range(T):
        order = range(K) if t % 2 == 0 else reversed(range(K))      # sub-batches are independent: any order
        for c in order:
            sub = subs[c]
            with torch.cuda.stream(sub.stream):      # actions are produced and results read on the sub-batch's stream
                o = sub.step(torch.as_tensor(acts[t, c * per:(c + 1) * per], device=env.device))
                got = tuple(_np(x) for x in o[:4]) + (_np(o[4]["reward_gt"]), _np(o[4]["final_obs"]), _np(o[4]["steps"]))
            for x, y in zip(got, ref[t]):
                assert np.array_equal(x, y[c * per:(c + 1) * per])
    torch.cuda.synchronize()
    s_got = [_np(x) for x in env.get_state()]           # the PARENT's records: the views wrote them
    for x, y in zip(s_got, s_ref):
        assert np.array_equal(x, y)
    assert np.array_equal(_np(subs[1].state), _np(env.state)[per:2 * per])
    assert env.check_errors() == 0
    # the parent's tables are pinned while views exist
    with pytest.raises(_lib.XenoError):
        _lib.check(env.lib.xv_anymdp_build_buckets(env._h, 16))
    assert env.lib.xv_anymdp_destroy(env._h) != 0
    env.set_search("bucket", n_bucket=32)                # other lines: the wrapper drops the views first
    assert env._views == []
    subs = env.split(2)
    assert subs[0].effective_search == "bucket"
    env.close()


def test_alternating_an_env_and_its_sub_batches_never_reuses_a_tick():
    """the pipelined-trainer pattern of quickstart section 9 mixed with whole-env steps: sub-batches in lockstep, then the env,
    then the sub-batches again ... — each side starts past every tick the other has used (AnyMDPVecEnv._sync_tick), so the
    trajectory equals ONE handle stepping the same actions, bit for bit (round 5 let the env redraw the sub-batches' ticks)"""
    tab = oracle.anymdp_synth(seed=31, task_index_base=0, n_task=8, S=64, A=8, s0_max=4)
    n, K = 1024, 4
    per = n // K
    plan = ["s", "s", "s", "p", "p", "s", "s", "p", "s", "p", "p"]
    acts = np.random.RandomState(9).randint(0, 8, (len(plan), n)).astype(np.int32)

    whole = AnyMDPVecEnv(n, seed=6, autoreset_mode="same_step")
    whole.set_task(_dev_tables(tab))
    whole.reset()
    ref = []
    for t in range(len(plan)):
        r = whole.step(acts[t])
        ref.append(tuple(_np(x) for x in r[:4]) + (_np(r[4]["reward_gt"]),))
    s_ref = [_np(x) for x in whole.get_state()]
    t_ref = whole.engine.tick
    whole.close()

    env = AnyMDPVecEnv(n, seed=6, autoreset_mode="same_step")
    env.set_task(_dev_tables(tab))
    env.reset()
    for t, who in enumerate(plan):
        if who == "p":
            torch.cuda.synchronize()
            r = env.step(acts[t])
            got = tuple(_np(x) for x in r[:4]) + (_np(r[4]["reward_gt"]),)
            for x, y in zip(got, ref[t]):
                assert np.array_equal(x, y), (t, who)
        else:
            torch.cuda.synchronize()
            for c, sub in enumerate(env.split(K)):          # (the cached views, re-synchronised)
                with torch.cuda.stream(sub.stream):
                    o = sub.step(torch.as_tensor(acts[t, c * per:(c + 1) * per], device=env.device))
                    got = tuple(_np(x) for x in o[:4]) + (_np(o[4]["reward_gt"]),)
                for x, y in zip(got, ref[t]):
                    assert np.array_equal(x, y[c * per:(c + 1) * per]), (t, who, c)
    torch.cuda.synchronize()
    for x, y in zip([_np(x) for x in env.get_state()], s_ref):
        assert np.array_equal(x, y)
    env._sync_tick()
    assert env.engine.tick == t_ref
    assert env.check_errors() == 0
    env.close()


def test_view_argument_checks():
    tab = oracle.anymdp_synth(seed=31, task_index_base=0, n_task=4, S=16, A=4, s0_max=3)
    env = AnyMDPVecEnv(256, seed=9, env_id_base=1000)
    env.set_task(_dev_tables(tab))
    from xenoverse_amd.engine import Engine
    lib = env.lib
    h = C.c_void_p()
    bad_base = Engine(env.device, seed=9, env_id_base=1000)          # must be 1000 + env_lo
    assert lib.xv_anymdp_view(env._h, bad_base.handle, 64, 64, C.byref(h)) != 0
    bad_seed = Engine(env.device, seed=10, env_id_base=1064)
    assert lib.xv_anymdp_view(env._h, bad_seed.handle, 64, 64, C.byref(h)) != 0
    ok = Engine(env.device, seed=9, env_id_base=1064)
    assert lib.xv_anymdp_view(env._h, ok.handle, 64, 256, C.byref(h)) != 0      # beyond the parent's envs
    assert lib.xv_anymdp_view(env._h, ok.handle, 64, 64, C.byref(h)) == 0
    v = C.c_void_p(h.value)
    h2 = C.c_void_p()
    assert lib.xv_anymdp_view(v, ok.handle, 0, 8, C.byref(h2)) != 0             # no views of views
    # chains must tile the parent
    arr = (C.c_void_p * 1)(v)
    z = torch.zeros((2, 256), dtype=torch.int32, device=env.device)
    f = torch.zeros((2, 256), dtype=torch.float32, device=env.device)
    b = torch.zeros((2, 256), dtype=torch.uint8, device=env.device)
    env.reset()
    rc = lib.xv_anymdp_step_many_chains(env._h, arr, 1, 0, 2, 2, z.data_ptr(), z.data_ptr(), f.data_ptr(), f.data_ptr(),
                                        b.data_ptr(), b.data_ptr(), None, 2)
    assert rc != 0
    assert lib.xv_anymdp_destroy(v) == 0
    env.close()
    for e in (bad_base, bad_seed, ok):
        e.close()


@pytest.mark.parametrize("kind", ["pomdp", "mtpomdp"])
def test_overlapped_token_steps_equal_plain_launches(kind):
    """xv_anymdp_step_tokens_many with the overlap on (cooperative token kernel, HAND): the env record is handed on before the
    observation stage — every output of the rings, the env records and the tick equal the plain launches', on the
    reference's own golden POMDP / multi-token tasks; short calls, an odd period and the per-lane kernel take the plain path"""
    from util import golden_files, load_anymdp_tok_golden
    tasks = [load_anymdp_tok_golden(p)[1] for p in golden_files("anymdptok_") if ("mtpomdp" in p) == (kind == "mtpomdp")]
    n = 1536
    env_task = (np.arange(n) % len(tasks)).astype(np.int32)
    res = []
    for overlap in (False, True):
        env = AnyMDPVecEnv(n, seed=13, autoreset_mode="same_step")
        env.set_task(tasks, env_task_index=env_task)
        env.set_search("bucket", n_bucket=16)
        assert env.token_kernel == "cooperative"
        env.set_step_many_overlap(overlap)
        env.reset()
        d_act = env._tok[1]
        rng = np.random.RandomState(3)
        rec = []
        for P, n_steps in ((8, 75), (8, 64), (7, 70), (8, 20)):
            acts = torch.as_tensor(rng.randint(0, int(env.na), (P, n, d_act)).astype(np.int32), device=env.device)
            out = env.step_tokens_many(n_steps, acts)
            torch.cuda.synchronize()
            rec.append({k: _np(v).copy() for k, v in out.items()})
            if overlap:
                assert env.step_many_overlap_state == (1 if (P % 2 == 0 and n_steps >= 64) else 0)
        s, st, nr = env.get_state()
        rec.append({"state": _np(s), "steps": _np(st), "nr": _np(nr), "tick": np.asarray(env.engine.tick)})
        assert env.check_errors() == 0
        res.append(rec)
        env.close()
    _same(res[0], res[1])
    assert res[0][0]["obs"].std() > 0 and (res[0][0]["terminated"].sum() + res[0][0]["truncated"].sum() + res[0][1]["reward"].std()) > 0


def test_one_overlapped_handle_per_device_and_never_a_view():
    """two overlapped calls in flight can block each other on the hardware queues their streams share: the second handle and
    any view are refused; closing (or switching off) the first frees the slot"""
    tab = oracle.anymdp_synth(seed=5, task_index_base=0, n_task=4, S=64, A=8, s0_max=4)
    a = AnyMDPVecEnv(512, seed=1)
    a.set_task(_dev_tables(tab))
    b = AnyMDPVecEnv(512, seed=2)
    b.set_task(_dev_tables(tab))
    a.set_step_many_overlap(True)
    with pytest.raises(_lib.XenoError):
        b.set_step_many_overlap(True)
    a.reset()
    with pytest.raises(_lib.XenoError):
        a.split(2)[0].set_step_many_overlap(True)
    a.set_step_many_overlap(False)
    b.set_step_many_overlap(True)
    b.close()                                    # a closed owner gives the slot back
    a.set_step_many_overlap(True)
    a.close()


def test_a_held_up_host_between_the_two_launches_of_a_cycle_is_harmless(monkeypatch):
    """The even half of a cycle waits for the odd half's steps for a bounded time (2^20 polls and 2 s).  With the host held up
    for 3.5 s between the two graph launches (test hook XV_PIPE_TEST_STALL_MS) that wait would expire — unless the even half starts behind the
    cycle gate, which passes only once both halves are enqueued (xv_pipe.h): no flag, same results"""
    tab = oracle.anymdp_synth(seed=12, task_index_base=0, n_task=16, S=64, A=8, s0_max=4)
    n, P = 4096, 8
    acts = np.random.RandomState(6).randint(0, 8, (P, n)).astype(np.int32)
    plan = [12 * P + 1]
    ref = _run_many(tab, n, P, acts, plan, "fence", 1, "streams", False)
    monkeypatch.setenv("XV_PIPE_TEST_STALL_MS", "3500")
    got = _run_many(tab, n, P, acts, plan, "fence", 1, "streams", True, overlap=True)      # asserts check_errors() == 0
    _same(ref, got)
    from xenoverse_amd.engine import Engine
    eng = Engine("cuda:0")
    probe = eng.probe_side_streams()
    eng.close()
    assert probe and probe[-1]["accepted"] and probe[-1]["two_stream_us"] > 0, probe


def test_overlap_is_not_taken_when_two_launches_cannot_be_resident_together():
    """A workgroup of step k + 1 spins in its slot until the same workgroup of step k has run; with grids beyond half of what
    the device holds, step k's workgroups could be left without slots.  Such calls take the one-stream path (xv_pipe.h)."""
    tab = oracle.anymdp_synth(seed=3, task_index_base=0, n_task=16, S=64, A=8, s0_max=4)
    n, P = 1 << 21, 2                                   # 8,192 workgroups per launch
    env = AnyMDPVecEnv(n, seed=5, bucket_lines="off")
    env.set_task(_dev_tables(tab), env_task_index=(torch.arange(n, device="cuda", dtype=torch.int32) % 16).contiguous())
    env.set_search("fence")
    env.set_step_many_overlap(True)
    env.reset()
    acts = torch.randint(0, 8, (P, n), device=env.device, dtype=torch.int32)
    env.step_many(64, acts)
    torch.cuda.synchronize()
    assert env.step_many_overlap_state == 0 and env.check_errors() == 0
    env.close()


def test_side_stream_is_chosen_by_measurement_with_a_communicator_created_first():
    """A high-priority side stream created AFTER an RCCL communicator shares a command-processor pipe with the engine's
    stream: the same two-graph schedule ran five times slower than one stream (profiles/r05_t_*).  The side stream is now
    picked by a timed ping-pong; with the communicator made first the overlapped call must not be slower than the
    one-stream call (it is 20-25 % faster; 1.3 x is the alarm threshold, far below the 5 x of the bug)."""
    from xenoverse_amd.distributed import RolloutGather
    try:
        g = RolloutGather((1 << 20,), device="cuda", transport="rccl", rank=0, world=1)
    except Exception as ex:      # no librccl on this box
        pytest.skip("RCCL transport unavailable: %r" % (ex,))
    g.launch(); g.wait()
    torch.cuda.synchronize()
    tab = oracle.anymdp_synth(seed=12, task_index_base=0, n_task=64, S=64, A=8, s0_max=4)
    n, P, steps = 16384, 32, 1024
    env = AnyMDPVecEnv(n, seed=9, bucket_lines="off")
    env.set_task(_dev_tables(tab), env_task_index=(torch.arange(n, device="cuda", dtype=torch.int32) % 64).contiguous())
    env.set_search("fence")
    env.reset()
    acts = torch.randint(0, 8, (P, n), device=env.device, dtype=torch.int32)
    us = {}
    for overlap in (False, True):
        env.set_step_many_overlap(overlap)
        ring = env.step_many(steps, acts)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            env.step_many(steps, acts, out=ring)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / steps)
        us[overlap] = best
        assert env.step_many_overlap_state == (1 if overlap else 0)
    assert env.check_errors() == 0
    env.set_step_many_overlap(False)
    env.close()
    g.close()
    print("us per step: one stream %.2f, overlapped %.2f" % (us[False], us[True]))
    assert us[True] < 1.3 * us[False], us


@pytest.mark.parametrize("P", [2, 4, 6])
def test_overlapped_step_many_on_short_rings(P):
    """Up to three steps are in flight: they must write distinct ring slots, so a ring of two keeps two streams; rings of
    four and six take three (graphs of several ring cycles, steps per stream not a multiple of the period)"""
    tab = oracle.anymdp_synth(seed=12, task_index_base=0, n_task=16, S=64, A=8, s0_max=4)
    n = 8192
    acts = np.random.RandomState(8).randint(0, 8, (P, n)).astype(np.int32)
    plan = [200, 66, 7, 131]
    ref = _run_many(tab, n, P, acts, plan, "fence", 1, "streams", False)
    got = _run_many(tab, n, P, acts, plan, "fence", 1, "streams", True, overlap=True)
    _same(ref, got)


# ---- an expired hand-off is repaired (ABI 12) ------------------------------------------------------------------------------
@pytest.mark.parametrize("search", ["fence", "bucket"])
@pytest.mark.parametrize("mode", ["same_step", "next_step"])
def test_an_expired_hand_off_is_repaired_by_the_replay(monkeypatch, search, mode):
    """XV_PIPE_TEST_FAIL=1 leaves behind what an expired hand-off does — XV_DEVERR_HANDOFF (and a spurious error bit), wrong env
    records, wrong ring contents — between the join of an overlapped call and its replay kernel.  The replay restores the
    records the call's opening kernel kept, re-runs the call on one stream and publishes entry | replay error bits: rings,
    records, tick and flags equal the one-stream path's, and the state word says -2."""
    tab = oracle.anymdp_synth(seed=12, task_index_base=0, n_task=16, S=64, A=8, s0_max=4)
    n, P = 2000, 8
    acts = np.random.RandomState(5).randint(0, 8, (P, n)).astype(np.int32)
    plan = [3 * P + 5, 9 * P + 3, 3, 12 * P]
    ref = _run_many(tab, n, P, acts, plan, search, 1, "streams", False, mode=mode)
    monkeypatch.setenv("XV_PIPE_TEST_FAIL", "1")
    monkeypatch.setenv("XV_PIPE_NO_BACKOFF", "1")          # every long call of the plan is overlapped, fails and is replayed
    got = _run_many(tab, n, P, acts, plan, search, 1, "streams", True, overlap=True, mode=mode, expect_state=-2)   # asserts flags == 0
    _same(ref, got)
    monkeypatch.delenv("XV_PIPE_NO_BACKOFF")
    if search == "fence" and mode == "same_step":
        # with the back-off (the default): the call after a replayed one takes the one-stream path — state 0, same results
        got = _run_many(tab, n, P, acts, plan, search, 1, "streams", True, overlap=True, mode=mode, expect_state=0)
        _same(ref, got)
    monkeypatch.delenv("XV_PIPE_TEST_FAIL")
    again = _run_many(tab, n, P, acts, plan, search, 1, "streams", True, overlap=True, mode=mode, expect_state=1)   # and no replay without a failure
    _same(ref, again)


@pytest.mark.parametrize("kind", ["pomdp", "mtpomdp"])
def test_an_expired_hand_off_of_the_token_steps_is_repaired_by_the_replay(monkeypatch, kind):
    """xv_anymdp_step_tokens_many with the overlap on and XV_PIPE_TEST_FAIL=1 (the flag, wrong records and wrong observation
    rings behind the join): the token replay kernel restores the records and re-runs the call — rings, records, tick and flags
    equal the plain launches' on the reference's golden POMDP / multi-token tasks; the state word says -2"""
    from util import golden_files, load_anymdp_tok_golden
    tasks = [load_anymdp_tok_golden(p)[1] for p in golden_files("anymdptok_") if ("mtpomdp" in p) == (kind == "mtpomdp")]
    n = 1536
    env_task = (np.arange(n) % len(tasks)).astype(np.int32)
    res = []
    for overlap in (False, True):
        if overlap:
            monkeypatch.setenv("XV_PIPE_TEST_FAIL", "1")
            monkeypatch.setenv("XV_PIPE_NO_BACKOFF", "1")
        env = AnyMDPVecEnv(n, seed=13, autoreset_mode="same_step")
        env.set_task(tasks, env_task_index=env_task)
        env.set_search("bucket", n_bucket=16)
        env.set_step_many_overlap(overlap)
        env.reset()
        d_act = env._tok[1]
        rng = np.random.RandomState(3)
        rec = []
        for P, n_steps in ((8, 75), (8, 64), (8, 20)):
            acts = torch.as_tensor(rng.randint(0, int(env.na), (P, n, d_act)).astype(np.int32), device=env.device)
            out = env.step_tokens_many(n_steps, acts)
            torch.cuda.synchronize()
            rec.append({k: _np(v).copy() for k, v in out.items()})
            if overlap:
                assert env.step_many_overlap_state == (-2 if n_steps >= 64 else 0), (n_steps, env.step_many_overlap_state)
        s, st, nr = env.get_state()
        rec.append({"state": _np(s), "steps": _np(st), "nr": _np(nr), "tick": np.asarray(env.engine.tick)})
        assert env.check_errors() == 0
        res.append(rec)
        env.close()
    _same(res[0], res[1])


def test_a_hand_off_that_really_expires_is_repaired():
    """XV_PIPE_TEST_BAD_TAG=1: the call's opening kernel gives env 0 a tag no step waits for, so the first step's first wave polls
    2^19 times and 0.5 s, gives up and sets XV_DEVERR_HANDOFF — the real expiry path.  The call is replayed: same results, no flag."""
    import os
    tab = oracle.anymdp_synth(seed=12, task_index_base=0, n_task=16, S=64, A=8, s0_max=4)
    n, P = 4096, 8
    acts = np.random.RandomState(6).randint(0, 8, (P, n)).astype(np.int32)
    plan = [10 * P]
    ref = _run_many(tab, n, P, acts, plan, "fence", 1, "streams", False)
    os.environ["XV_PIPE_TEST_BAD_TAG"] = "1"
    try:
        import time
        t0 = time.time()
        got = _run_many(tab, n, P, acts, plan, "fence", 1, "streams", True, overlap=True, expect_state=-2)
        assert time.time() - t0 > 0.4          # the wave really waited for its bound (2^19 polls and 0.5 s)
    finally:
        del os.environ["XV_PIPE_TEST_BAD_TAG"]
    _same(ref, got)


def test_a_held_up_host_between_the_second_and_the_third_launch_of_a_set_is_harmless(monkeypatch):
    """Three steps in flight: with the host held up for 3.5 s between the second and the third graph launch of a set, the second
    graph's waves would spin (2-s bound) on a first graph that is still gated — unless every graph of a set starts behind the
    gate (xv_pipe.h).  No flag, no replay, same results."""
    tab = oracle.anymdp_synth(seed=12, task_index_base=0, n_task=16, S=64, A=8, s0_max=4)
    n, P = 4096, 6
    acts = np.random.RandomState(6).randint(0, 8, (P, n)).astype(np.int32)
    plan = [20 * P + 1]
    ref = _run_many(tab, n, P, acts, plan, "fence", 1, "streams", False)
    monkeypatch.setenv("XV_PIPE_TEST_STALL_MS", "3500")
    monkeypatch.setenv("XV_PIPE_TEST_STALL_AT", "1")
    got = _run_many(tab, n, P, acts, plan, "fence", 1, "streams", True, overlap=True, expect_state=1)
    _same(ref, got)


@pytest.mark.parametrize("neighbour", ["matmul", "allgather"])
def test_overlapped_step_many_beside_a_busy_neighbour_of_the_same_process(neighbour):
    """65,536 envs, overlapped launches, while the SAME process keeps another stream busy — a torch matmul loop (a policy
    network's kernels take CU slots) or xv_rollout_allgather of 29-MB chunks (RCCL's kernels).  Results must equal the
    one-stream path bit for bit with no flag left: either the hand-offs simply took longer (state 1), or an expiry was
    replayed (-2), or the library declined (0)."""
    import threading
    tab = oracle.anymdp_synth(seed=21, task_index_base=0, n_task=1024, S=64, A=8, s0_max=4)
    n, P = 65536, 32
    acts = np.random.RandomState(7).randint(0, 8, (P, n)).astype(np.int32)
    plan = [20 * P, 20 * P + 3, 40 * P]
    ref = _run_many(tab, n, P, acts, plan, "bucket", 1, "streams", True)
    stop = threading.Event()
    side = torch.cuda.Stream()
    gather = None
    if neighbour == "allgather":
        from xenoverse_amd.distributed import RolloutGather
        try:
            gather = RolloutGather((32, 65536, 14), device="cuda", transport="rccl", rank=0, world=1)     # 29 MB per rank
        except Exception as ex:
            pytest.skip("RCCL transport unavailable: %r" % (ex,))
    count = [0]

    def busy():
        torch.cuda.set_device(0)
        if neighbour == "matmul":
            with torch.cuda.stream(side):
                a = torch.randn(4096, 4096, device="cuda", dtype=torch.bfloat16)
                b = torch.randn(4096, 4096, device="cuda", dtype=torch.bfloat16)
                while not stop.is_set():
                    for _ in range(8):
                        a = torch.tanh(a @ b) * 0.5
                    side.synchronize()
                    count[0] += 8
        else:
            with torch.cuda.stream(side):      # (this thread's current stream: not the stepping's)
                while not stop.is_set():
                    gather.launch()
                    gather.wait()
                    side.synchronize()
                    count[0] += 1
    th = threading.Thread(target=busy, daemon=True)
    th.start()
    states = []
    try:
        import time
        while count[0] == 0 and th.is_alive():
            time.sleep(0.01)
        for _ in range(3):
            env_states = []
            got = _run_many_any_state(tab, n, P, acts, plan, "bucket", env_states)      # asserts flags == 0
            _same(ref, got)
            assert all(v in (1, -2, 0) for v in env_states), env_states
            states.append(env_states)
    finally:
        stop.set()
        th.join(timeout=30)
        if gather is not None:
            gather.close()
    assert count[0] > 0
    print("neighbour %s: overlap states %s, neighbour iterations %d" % (neighbour, states, count[0]))


def _run_many_any_state(tab, n, P, acts_np, plan, search, states):
    env = AnyMDPVecEnv(n, seed=77, autoreset_mode="same_step", bucket_lines="off")
    env.set_task(_dev_tables(tab))
    env.set_search("bucket", n_bucket=16) if search == "bucket" else env.set_search(search)
    env.set_step_many_graph(True)
    env.set_step_many_overlap(True)
    env.reset()
    acts = torch.as_tensor(acts_np, device=env.device)
    rec, ring = [], None
    for n_steps in plan:
        ring = env.step_many(n_steps, acts, out=ring)
        torch.cuda.synchronize()
        states.append(env.step_many_overlap_state)
        rec.append({k: _np(v).copy() for k, v in ring.items()})
    s, st, nr = env.get_state()
    rec.append({"state": _np(s), "steps": _np(st), "need_reset": _np(nr), "tick": np.asarray(env.engine.tick)})
    assert env.check_errors() == 0
    env.set_step_many_overlap(False)
    env.close()
    return rec
