"""MixedBatch — several environment families stepped concurrently on one GPU (BASELINE.json config 5: anymdp +
linds + metacontrol, 262,144 envs sharded over 8 GPUs = per GPU 16,384 anymdp + 8,192 linds + 8,192 cartpole).

Each family keeps its own engine.  `streams="shared"` (default) launches all families on the caller's stream;
`streams="separate"` gives every family a private HIP stream so that their step kernels can overlap, with
`sync()` making the caller's stream wait for all of them.  Measured on one MI355X at the config-5 share
(`scripts/bench_families.py --families mixed`): the three step kernels take 16.4 us back to back on one stream and
47.6 us on three streams — they are 4-8 us each, so the cross-stream event waits a dependent step loop needs
(actions in, observations out) cost more than the overlap saves; separate streams pay off only when a family's
step is long (MazeWorld frames) or the consumer does not need all families at every step.  A family's trajectory
is identical either way and identical to what it would be stepped alone: engines share nothing (no global RNG).
Across GPUs each rank owns a contiguous slice of every family (`distributed.shard_range`).

Round 3: `step_fused` advances one AnyMDP, one LinDS and one CartPole family with ONE kernel launch (`xv_mixed_step`:
the families' own step bodies share a grid) — same results as `step`, one launch latency instead of three.
"""
import ctypes as C

import torch

from . import _lib
from .engine import AUTORESET, Engine


class _MixedIO(C.Structure):   # xv_mixed_io (include/xeno.h)
    _fields_ = [(k, C.c_void_p) for k in (
        "a_action", "a_obs", "a_reward", "a_reward_gt", "a_terminated", "a_truncated", "a_final_obs",
        "l_action", "l_obs", "l_reward", "l_terminated", "l_truncated", "l_cmd", "l_error", "l_final_obs",
        "c_action", "c_obs", "c_reward", "c_terminated", "c_truncated", "c_final_obs",
        "a_steps", "a_done", "l_steps", "l_done", "c_done")]      # ABI 12: steps / done masks from the fused launch (nullable)


class MixedBatch(object):
    def __init__(self, device="cuda:0", seed=0, streams="shared"):
        assert streams in ("shared", "separate")
        self.device = torch.device(device)
        self.seed = int(seed)
        self.separate = streams == "separate"
        self.envs = {}
        self.streams = {}
        self._trio = None          # (_fused_trio result, cached until add())
        self._fast = None          # copy=False: the marshalled xv_mixed_io structs and the returned views of step_fused

    def add(self, name, env_cls, num_envs, env_id_base=0, **kwargs):
        """Create `env_cls(num_envs, engine=<engine on a private stream>, **kwargs)` under `name`."""
        st = torch.cuda.Stream(device=self.device) if self.separate else torch.cuda.current_stream(self.device)
        eng = Engine(self.device, seed=self.seed, env_id_base=env_id_base, stream=st)
        env = env_cls(num_envs, engine=eng, **kwargs)
        env._own_engine = True   # closed with the env
        self.envs[name] = env
        self.streams[name] = st
        self._trio = self._fast = None
        return env

    def _on(self, name):
        st = self.streams[name]
        if self.separate:
            st.wait_stream(torch.cuda.current_stream(self.device))   # inputs produced on the caller's stream
        return torch.cuda.stream(st)

    def set_task(self, tasks):
        for name, t in tasks.items():
            with self._on(name):
                if isinstance(t, tuple):
                    self.envs[name].set_task(t[0], env_task_index=t[1])
                else:
                    self.envs[name].set_task(t)

    def reset(self):
        out = {}
        for name, env in self.envs.items():
            with self._on(name):
                out[name] = env.reset()
        self.sync()
        return out

    def step(self, actions):
        """actions: dict name -> batched action.  Launches every family's step on its own stream, then makes the
        caller's stream wait for all of them.  -> dict name -> (obs, reward, terminated, truncated, infos)"""
        out = {}
        for name, env in self.envs.items():
            with self._on(name):
                out[name] = env.step(actions[name])
        self.sync()
        return out

    def _fused_trio(self):
        if self._trio is not None:
            return self._trio
        self._trio = self._fused_trio_uncached()
        return self._trio

    def _fused_trio_uncached(self):
        from .anymdp import AnyMDPVecEnv
        from .linds import LinDSVecEnv
        from .metacontrol import CartPoleVecEnv
        pick = {}
        for name, env in self.envs.items():
            for key, cls in (("a", AnyMDPVecEnv), ("l", LinDSVecEnv), ("c", CartPoleVecEnv)):
                if isinstance(env, cls):
                    if key in pick:
                        raise ValueError("step_fused takes one env of each family (anymdp, linds, cartpole)")
                    pick[key] = (name, env)
        if len(pick) != 3 or len(self.envs) != 3:
            raise ValueError("step_fused needs exactly one AnyMDPVecEnv, one LinDSVecEnv and one CartPoleVecEnv")
        if self.separate:
            raise ValueError("step_fused launches one kernel: build the batch with streams='shared'")
        modes = {e.autoreset_mode for _, e in pick.values()}
        if len(modes) != 1:
            raise ValueError("the three families must use one autoreset_mode")
        return pick, modes.pop()

    def step_fused(self, actions):
        """One launch for the whole mixed batch (xv_mixed_step).  actions: dict name -> batched action, as for step().
        -> dict name -> (obs, reward, terminated, truncated, infos), bit for bit what step() returns.  Handles without a
        fused instantiation (AnyMDP on the per-lane search or S > 112, LinDS on the scalar path or other pads) are stepped
        by step() — three launches, same results."""
        pick, mode = self._fused_trio()
        (na, ea), (nl, el), (nc, ec) = pick["a"], pick["l"], pick["c"]
        if ea._tok is not None:
            raise ValueError("step_fused serves MDP tasks")
        ea._check_step(); el._check_step()
        if (not ec.task_set) or ec.need_reset:
            raise Exception("Must \"set_task\" and \"reset\" before doing any actions")
        if not ea.lib.xv_mixed_supported(ea._h, el._h, ec._h):      # asked BEFORE any output buffer is renewed
            return self.step(actions)
        aa = ea._dev(actions[na], torch.int32)
        al = el._action(actions[nl])
        ac = ec._dev(actions[nc], torch.int32)
        if aa.shape != (ea.num_envs,) or ac.shape != (ec.num_envs,):
            raise AssertionError("action batch shapes do not match the env counts")
        if ea._ring is not None and not (el.copy or el.to_numpy or ec.copy or ec.to_numpy):
            return self._step_fused_persistent(na, ea, nl, el, nc, ec, aa, al, ac, mode)
        if all(e.copy and not e.to_numpy for e in (ea, el, ec)) and not ea.with_transition_gt and el._path_name != "scalar":
            # copy=True: AnyMDP and LinDS outputs come from their 64-step slabs (no allocation, no new tensor objects: vector.
            # OutputSlabs), CartPole's five from fresh buffers; ONE launch writes everything incl. steps and the done masks
            ta, pa = ea._slab_next()
            tl, pl = el._slab_next()
            ec._renew("_obs", "_reward", "_term", "_trunc", "_fobs", "_done")
            P = _lib.ptr
            io = _MixedIO(P(aa), pa[0], pa[1], pa[2], pa[3], pa[4], pa[5],
                          P(al), pl[0], pl[1], pl[2], pl[3], pl[4], pl[5], pl[6],
                          P(ac), P(ec._obs), P(ec._reward), P(ec._term), P(ec._trunc), P(ec._fobs),
                          pa[6], pa[7], pl[7], pl[8], P(ec._done))
            _lib.check(ea.lib.xv_mixed_step(ea._h, el._h, ec._h, C.byref(io), AUTORESET[mode]))
            ea._obs = ta["obs"]
            el._obs, el._cmd, el._error = tl["obs"], tl["cmd"], tl["error"]
            ia = {"steps": ta["steps"], "reward_gt": ta["reward_gt"]}
            il = {"steps": tl["steps"], "command": tl["cmd_u"], "error": tl["error"]}
            if mode == "same_step":
                ia["final_obs"], ia["_final_obs"] = ta["final_obs"], ta["done"]
                il["final_obs"], il["_final_obs"] = tl["fobs_u"], tl["done"]
            return {na: (ta["obs"], ta["reward"], ta["term"], ta["trunc"], ia),
                    nl: (tl["obs_u"], tl["reward"], tl["term"], tl["trunc"], il), nc: ec._ret(from_launch=True)}
        ring = None
        if ea._ring is not None:      # copy=False: the AnyMDP family writes one of its two engine-owned output sets
            ring = ea._ring[ea._ring_pos]
            if not ea._holding:
                ea._ring_pos ^= 1
            a_out = [ring[k] for k in ("obs", "reward", "reward_gt", "term", "trunc", "final_obs")]
        else:
            ea._renew("_obs", "_reward", "_reward_gt", "_term", "_trunc", "_final_obs")
            a_out = [ea._obs, ea._reward, ea._reward_gt, ea._term, ea._trunc, ea._final_obs]
        el._renew(*el._STEP_OUTPUTS); el._fresh_final_obs()
        ec._renew("_obs", "_reward", "_term", "_trunc", "_fobs")
        io = _MixedIO(*[_lib.ptr(t) for t in (
            [aa] + a_out + [al, el._obs, el._reward, el._term, el._trunc, el._cmd, el._error, el._fobs,
                            ac, ec._obs, ec._reward, ec._term, ec._trunc, ec._fobs])])
        _lib.check(ea.lib.xv_mixed_step(ea._h, el._h, ec._h, C.byref(io), AUTORESET[mode]))
        if ring is not None:
            infos = {"reward_gt": ring["reward_gt"]}
            if mode == "same_step":
                infos["final_obs"] = ring["final_obs"]
            if not ea.lean_infos:
                _lib.check(ea.lib.xv_anymdp_get_state(ea._h, None, ring["steps_p"], None))
                infos["steps"] = ring["steps"]
                if mode == "same_step":
                    torch.bitwise_or(ring["term"], ring["trunc"], out=ring["done"])
                    infos["_final_obs"] = ring["done_b"]
            ra = (ring["obs"], ring["reward"], ring["term_b"], ring["trunc_b"], infos)
        else:
            ra = (ea._of(ea._obs), ea._of(ea._reward), ea._obf(ea._term), ea._obf(ea._trunc), ea._infos(aa))
        return {na: ra, nl: el._ret(), nc: ec._ret()}

    def _step_fused_persistent(self, na, ea, nl, el, nc, ec, aa, al, ac, mode):
        """copy=False on all three families: every output lives in engine-owned buffers (AnyMDP: two sets used alternately), so
        the xv_mixed_io structs, their pointers and the views handed out are made once; a step sets three action pointers and
        makes ONE launch — the fused kernel writes info["steps"] and the done masks itself (ABI 12)"""
        key = (tuple(b["obs"].data_ptr() for b in ea._ring), el._obs.data_ptr(), el._fobs.data_ptr(), el._steps.data_ptr(),
               ec._obs.data_ptr(), ec._fobs.data_ptr(), ec._done.data_ptr(), mode)
        f = self._fast
        if f is None or f["key"] != key:
            P = _lib.ptr
            sets = []
            lc = el._make_step_cache()
            for b in ea._ring:
                io = _MixedIO(None, P(b["obs"]), P(b["reward"]), P(b["reward_gt"]), P(b["term"]), P(b["trunc"]), P(b["final_obs"]),
                              None, P(el._obs), P(el._reward), P(el._term), P(el._trunc), P(el._cmd), P(el._error), P(el._fobs),
                              None, P(ec._obs), P(ec._reward), P(ec._term), P(ec._trunc), P(ec._fobs),
                              P(b["steps"]), P(b["done"]), P(el._steps), P(el._done), P(ec._done))
                ia = {"reward_gt": b["reward_gt"], "steps": b["steps"]}
                il = {"steps": el._steps, "command": lc["cmd"], "error": el._error}
                ic = {}
                if mode == "same_step":
                    ia["final_obs"], ia["_final_obs"] = b["final_obs"], b["done_b"]
                    il["final_obs"], il["_final_obs"] = lc["fobs"], lc["done_b"]
                    ic["final_obs"], ic["_final_obs"] = ec._fobs, ec._done.view(torch.bool)
                sets.append((io, C.byref(io), {na: (b["obs"], b["reward"], b["term_b"], b["trunc_b"], ia),
                                               nl: (lc["obs"], el._reward, lc["term_b"], lc["trunc_b"], il),
                                               nc: (ec._obs, ec._reward, ec._term.view(torch.bool), ec._trunc.view(torch.bool), ic)}))
            f = self._fast = {"key": key, "sets": sets, "mode": AUTORESET[mode]}
        pos = ea._ring_pos
        if not ea._holding:
            ea._ring_pos ^= 1
        io, ref, out = f["sets"][pos]
        io.a_action, io.l_action, io.c_action = aa.data_ptr(), al.data_ptr(), ac.data_ptr()
        _lib.check(ea.lib.xv_mixed_step(ea._h, el._h, ec._h, ref, f["mode"]))
        return out

    def capture(self, policy_fn, obs, unroll=1, warmup=1, lean=True):
        """[policy_fn(obs dict) -> actions dict; step_fused(actions)] captured in a torch.cuda.graph (capture.py):
        one graph launch per `unroll` vector steps of the whole mixed batch.  obs: dict name -> the observation the family
        returned last.  The envs must be built with copy=False."""
        from .capture import CapturedLoop
        pick, _ = self._fused_trio()
        envs = [e for _, e in pick.values()]
        if any(e.copy or e.to_numpy for e in envs):
            raise ValueError("capture() needs the three envs built with copy=False and to_numpy=False")
        for e in envs:
            e._lean_saved, e._lean_wanted = e.lean_infos, bool(lean) or e.lean_infos

        def hold(on):
            for e in envs:
                e._capture_hold(on)

        def step(actions):
            out = self.step_fused(actions)
            return ({k: v[0] for k, v in out.items()}, out)

        return CapturedLoop(step, [e.engine for e in envs], policy_fn, obs, unroll=unroll, warmup=warmup, device=self.device,
                            hold=hold)

    def sync(self):
        if not self.separate:
            return
        cur = torch.cuda.current_stream(self.device)
        for st in self.streams.values():
            cur.wait_stream(st)

    @property
    def num_envs(self):
        return sum(e.num_envs for e in self.envs.values())

    def close(self):
        for env in self.envs.values():
            env.close()
