"""Generate tests/golden/*.npz by running the REFERENCE itself (build container only).

usage:  python oracle/gen_golden.py [anymdp] [linds] [maze] [cartpole] ...   (default: all available)

Each fixture holds inputs and the reference's outputs for one family.  The
reference is stochastic through numpy's *global* legacy RandomState and reseeds it from OS entropy at
every reset (anymdp_env.py:87), so every recorded call is preceded by `numpy.random.seed(seed_j)`; the
uniform / normal numbers the call consumed are then replayed from `RandomState(seed_j)` and stored next to
the outputs.  A restatement is correct iff, given those numbers, it reproduces the outputs.

Nothing from /root/reference is copied: fixtures are arrays (task tensors sampled by the reference's
sampler, actions, random numbers, results).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import _refimport  # noqa: E402
import sample_ref_tasks  # noqa: E402

GOLD = os.path.join(os.path.dirname(HERE), "tests", "golden")


def _task_arrays(task):
    out = dict(ns=np.int64(task["ns"]), na=np.int64(task["na"]), max_steps=np.float64(task["max_steps"]),
               state_mapping=np.asarray(task["state_mapping"], np.int64),
               s_0=np.atleast_1d(np.asarray(task["s_0"], np.int64)),
               s_0_prob=np.atleast_1d(np.asarray(task["s_0_prob"], np.float64)),
               s_e=np.asarray(task["s_e"], np.int64).reshape(-1),
               transition=np.asarray(task["transition"], np.float64),
               reward=np.asarray(task["reward"], np.float64),
               reward_noise=np.asarray(task["reward_noise"], np.float64))
    return out


def gen_anymdp_one(name, task, n_tuples=4096, n_traj=1536, seed0=1000):
    """G-A for one task: single_step tuples, reset draws, a full step()/reset() trajectory that crosses
    the max_steps truncation boundary, and transition_gt rows."""
    AnyMDPEnv, _ = _refimport.anymdp()
    import xenoverse.anymdp.anymdp_env as envmod

    env = AnyMDPEnv(max_steps=5000)
    env.set_task(task)
    n = len(task["state_mapping"])
    na = int(task["na"])
    s_e = set(int(x) for x in np.asarray(task["s_e"]).reshape(-1))
    non_term = np.array([s for s in range(n) if s not in s_e], dtype=np.int64)
    rng = np.random.RandomState(seed0)

    # the reference reseeds from OS entropy inside reset(): inject the seed at that very call site
    inject = {"seed": 0}
    envmod.pseudo_random_seed = lambda *a, **k: inject["seed"]

    def ref_reset(seed):
        inject["seed"] = int(seed)
        obs, info = env.reset()
        return obs, info

    # ---- reset draws -------------------------------------------------------------------------
    n_reset = 256
    reset_seed = np.arange(seed0 + 500000, seed0 + 500000 + n_reset, dtype=np.int64)
    reset_u = np.zeros(n_reset)
    reset_state = np.zeros(n_reset, np.int64)
    reset_obs = np.zeros(n_reset, np.int64)
    for j in range(n_reset):
        obs, info = ref_reset(reset_seed[j])
        assert info["steps"] == 0
        reset_u[j] = np.random.RandomState(int(reset_seed[j])).random_sample()
        reset_state[j] = env._state
        reset_obs[j] = obs

    # ---- single_step tuples ------------------------------------------------------------------
    ref_reset(1)
    ss_s = non_term[rng.randint(0, len(non_term), n_tuples)]
    ss_a = rng.randint(0, na, n_tuples).astype(np.int64)
    ss_seed = np.arange(seed0, seed0 + n_tuples, dtype=np.int64)
    ss_u = np.zeros(n_tuples)
    ss_z = np.zeros(n_tuples)
    ss_next = np.zeros(n_tuples, np.int64)
    ss_rgt = np.zeros(n_tuples)
    ss_r = np.zeros(n_tuples)
    ss_term = np.zeros(n_tuples, np.uint8)
    for j in range(n_tuples):
        env._state = int(ss_s[j])
        np.random.seed(int(ss_seed[j]))
        rgt, r, term = env.single_step(int(ss_a[j]))
        rs = np.random.RandomState(int(ss_seed[j]))
        ss_u[j] = rs.random_sample()
        ss_z[j] = rs.standard_normal()
        ss_next[j] = env._state
        ss_rgt[j] = rgt
        ss_r[j] = r
        ss_term[j] = term

    # ---- full trajectory through step()/reset(), manual reset on done (how the reference is driven,
    #      anymdp/test_utils.py:42-60), forced across the truncation boundary once ---------------
    tr_a = rng.randint(0, na, n_traj).astype(np.int64)
    tr_seed = np.arange(seed0 + 100000, seed0 + 100000 + n_traj, dtype=np.int64)
    tr_reset_seed = np.arange(seed0 + 200000, seed0 + 200000 + n_traj, dtype=np.int64)
    tr_u = np.zeros(n_traj); tr_z = np.zeros(n_traj); tr_ur = np.zeros(n_traj)
    tr_obs = np.zeros(n_traj, np.int64); tr_r = np.zeros(n_traj); tr_rgt = np.zeros(n_traj)
    tr_term = np.zeros(n_traj, np.uint8); tr_trunc = np.zeros(n_traj, np.uint8)
    tr_steps = np.zeros(n_traj, np.int64); tr_state = np.zeros(n_traj, np.int64)
    tr_reset_obs = np.full(n_traj, -1, np.int64)
    tr_tgt = np.zeros((n_traj, int(task["ns"])))
    tr_set_steps = np.full(n_traj, -1, np.int64)  # steps counter forced BEFORE step t (-1: untouched)
    obs0, _ = ref_reset(seed0 + 300000)
    init_u = np.random.RandomState(seed0 + 300000).random_sample()
    init_state = int(env._state)
    jump_at = n_traj // 3
    for t in range(n_traj):
        if t == jump_at:  # place the episode 3 steps before truncation
            env.steps = int(np.ceil(float(task["max_steps"]))) - 3
            tr_set_steps[t] = env.steps
        np.random.seed(int(tr_seed[t]))
        obs, r, term, trunc, info = env.step(int(tr_a[t]))
        rs = np.random.RandomState(int(tr_seed[t]))
        tr_u[t] = rs.random_sample(); tr_z[t] = rs.standard_normal()
        tr_obs[t] = obs; tr_r[t] = r; tr_rgt[t] = info["reward_gt"]
        tr_term[t] = term; tr_trunc[t] = trunc; tr_steps[t] = info["steps"]; tr_state[t] = env._state
        tr_tgt[t] = info["transition_gt"]
        tr_ur[t] = np.random.RandomState(int(tr_reset_seed[t])).random_sample()
        if term or trunc:
            o, _ = ref_reset(tr_reset_seed[t])
            tr_reset_obs[t] = o

    out = _task_arrays(task)
    out.update(reset_u=reset_u, reset_state=reset_state, reset_obs=reset_obs,
               ss_s=ss_s, ss_a=ss_a, ss_u=ss_u, ss_z=ss_z, ss_next=ss_next, ss_rgt=ss_rgt, ss_r=ss_r,
               ss_term=ss_term,
               tr_a=tr_a, tr_u=tr_u, tr_z=tr_z, tr_ur=tr_ur, tr_obs=tr_obs, tr_r=tr_r, tr_rgt=tr_rgt,
               tr_term=tr_term, tr_trunc=tr_trunc, tr_steps=tr_steps, tr_state=tr_state,
               tr_reset_obs=tr_reset_obs, tr_tgt=tr_tgt[:256], tr_set_steps=tr_set_steps,
               init_u=np.float64(init_u), init_state=np.int64(init_state), init_obs=np.int64(obs0))
    path = os.path.join(GOLD, name + ".npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB;",
          "episodes ended:", int((tr_term | tr_trunc).sum()), "truncations:", int(tr_trunc.sum()))


def gen_anymdp_tok_one(name, task, T=768, seed0=31000):
    """POMDP / MTPOMDP: a step()/reset() trajectory with the draws replayed in the reference's call order
    (per action token: choice, normal; then the observation choices; anymdp_env.py:116-128,148-157)."""
    AnyMDPEnv, _ = _refimport.anymdp()
    import xenoverse.anymdp.anymdp_env as envmod
    env = AnyMDPEnv(max_steps=5000)
    env.set_task(task)
    mt = task["task_type"] == "MTPOMDP"
    da = int(task["da"]) if mt else 1
    do = int(task["do"]) if mt else 1
    na, n = int(task["na"]), len(task["state_mapping"])
    s_e = set(int(x) for x in np.asarray(task["s_e"]).reshape(-1))
    cdfT = np.cumsum(task["transition"], -1)
    cdfT = cdfT / np.where(cdfT[..., -1:] == 0, 1, cdfT[..., -1:])
    inject = {"seed": 0}
    envmod.pseudo_random_seed = lambda *a, **k: inject["seed"]
    rng = np.random.RandomState(seed0)

    def ref_reset(seed):
        inject["seed"] = int(seed)
        obs, info = env.reset()
        rs = np.random.RandomState(int(seed))
        u_r = rs.random_sample()
        u_o = np.array([rs.random_sample() for _ in range(do)])
        return np.atleast_1d(np.asarray(obs, np.int64)), u_r, u_o

    acts = rng.randint(0, na, (T, da)).astype(np.int64)
    U = np.full((T, da), 0.5); Z = np.zeros((T, da)); UO = np.zeros((T, do)); UR = np.zeros(T); UOR = np.zeros((T, do))
    obs = np.zeros((T, do), np.int64); rew = np.zeros(T); rgt = np.zeros(T)
    term = np.zeros(T, np.uint8); trunc = np.zeros(T, np.uint8); steps = np.zeros(T, np.int64)
    state = np.zeros(T, np.int64); reset_obs = np.full((T, do), -1, np.int64); set_steps = np.full(T, -1, np.int64)
    init_obs, init_ur, init_uo = ref_reset(seed0 + 7)
    init_state = int(env._state)
    for t in range(T):
        if t == T // 3:
            env.steps = int(np.ceil(float(task["max_steps"]))) - 3
            set_steps[t] = env.steps
        s_cur = int(env._state)
        np.random.seed(seed0 + 100 + t)
        o, r, te, tr, info = env.step(acts[t] if mt else int(acts[t, 0]))
        rs = np.random.RandomState(seed0 + 100 + t)
        for k in range(da):   # replay in the reference's order, stopping where its token loop stopped
            U[t, k] = rs.random_sample(); Z[t, k] = rs.standard_normal()
            s_cur = int(np.searchsorted(cdfT[s_cur, acts[t, k]], U[t, k], side="right"))
            if s_cur in s_e:
                break
        assert s_cur == int(env._state)
        UO[t] = [rs.random_sample() for _ in range(do)]
        obs[t] = np.atleast_1d(np.asarray(o, np.int64)); rew[t] = r; rgt[t] = info["reward_gt"]
        term[t] = te; trunc[t] = tr; steps[t] = info["steps"]; state[t] = env._state
        if te or tr:
            reset_obs[t], UR[t], UOR[t] = ref_reset(seed0 + 50000 + t)
    out = _task_arrays(task)
    obsT = np.stack([np.asarray(m, np.float64) for m in task["observation_transition"]]) if mt else \
        np.asarray(task["observation_transition"], np.float64)[None]
    out.update(no=np.int64(task["no"]), do=np.int64(do), da=np.int64(da), is_mt=np.int64(mt),
               observation_transition=obsT, tr_a=acts, tr_u=U, tr_z=Z, tr_uo=UO, tr_ur=UR, tr_uor=UOR, tr_obs=obs,
               tr_r=rew, tr_rgt=rgt, tr_term=term, tr_trunc=trunc, tr_steps=steps, tr_state=state,
               tr_reset_obs=reset_obs, tr_set_steps=set_steps, init_obs=init_obs, init_ur=np.float64(init_ur),
               init_uo=init_uo, init_state=np.int64(init_state))
    path = os.path.join(GOLD, name + ".npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB; type", task["task_type"], "da", da, "do", do,
          "ended:", int((term | trunc).sum()), "trunc:", int(trunc.sum()))


def gen_anymdp():
    gen_anymdp_tok_one("anymdptok_pomdp_16x4_seed5", sample_ref_tasks.get(16, 4, 5, "pomdp"))
    gen_anymdp_tok_one("anymdptok_mtpomdp_16x4_seed6", sample_ref_tasks.get(16, 4, 6, "mtpomdp"), seed0=47000)
    # seed 0's MDP has terminal states (s_e = [4, 5, 11]): exercises the early break of the token loop
    gen_anymdp_tok_one("anymdptok_mtpomdp_16x4_seed0", sample_ref_tasks.get(16, 4, 0, "mtpomdp"), seed0=53000)
    for seed in range(4):
        task = sample_ref_tasks.get(16, 4, seed)
        gen_anymdp_one("anymdp_16x4_seed%d" % seed, task, seed0=1000 + 7919 * seed)
    p = sample_ref_tasks.cache_path(64, 8, 1)
    if os.path.exists(p):
        gen_anymdp_one("anymdp_64x8_seed1", sample_ref_tasks.get(64, 8, 1), seed0=77000)
    else:
        print("skip 64x8: task cache not ready (run oracle/sample_ref_tasks.py 64 8 1; ~9 min)")


def _linds_task_arrays(task):
    dyn = str(task["target_type"]) == "dynamic_target"
    out = dict(state_dim=np.int64(task["state_dim"]), observation_dim=np.int64(task["observation_dim"]),
               action_dim=np.int64(task["action_dim"]), max_steps=np.int64(task["max_steps"]),
               ld_A=np.asarray(task["ld_A"], np.float64), ld_B=np.asarray(task["ld_B"], np.float64),
               ld_C=np.asarray(task["ld_C"], np.float64), ld_X=np.asarray(task["ld_X"], np.float64),
               ld_Y=np.asarray(task["ld_Y"], np.float64), action_cost=np.float64(task["action_cost"]),
               reward_base=np.float64(task["reward_base"]), terminate_punish=np.float64(task["terminate_punish"]),
               reward_factor=np.float64(task["reward_factor"]),
               target_valid=np.asarray(task["target_valid"], np.int64), is_dynamic=np.int64(dyn),
               initial_states=np.stack([np.asarray(x, np.float64) for x in task["initial_states"]]),
               noise_drift=np.float64(task["noise_drift"]), target_delay=np.int64(task["target_delay"]))
    if dyn:
        c = task["command"]
        out["four_orders"] = np.array([float(o) for o, _ in c.coeffs], np.float64)
        out["four_coeffs"] = np.stack([np.asarray(f, np.float64) for _, f in c.coeffs])
        out["four_period"] = np.float64(c.max_steps)
    else:
        out["command"] = np.asarray(task["command"], np.float64)
    return out


def gen_linds_one(name, task, T=512, seed0=5000, act_scale=1.5):
    """G-L for one task: a T-step trajectory through the reference LinearDSEnv.step()/reset(), manual reset on
    done.  numpy's global RNG is seeded before every step (process noise) and the stdlib RNG before every reset
    (linds_env.py:117 uses random.choice); both draws are replayed and stored."""
    import random as pyrandom
    LinearDSEnv, _ = _refimport.linds()
    import xenoverse.linds.linds_env as envmod
    envmod.pseudo_random_seed = lambda *a, **k: 12345   # reset()'s reseeding of numpy: irrelevant, made inert

    env = LinearDSEnv()   # dt=0.1, pads 16/16/8 (linds_env.py:16-19)
    env.set_task(task)
    ns = int(task["state_dim"])
    n_init = len(task["initial_states"])
    rng = np.random.RandomState(seed0)

    def ref_reset(seed):
        pyrandom.seed(int(seed))
        idx = pyrandom.Random(int(seed)).randrange(n_init)   # what random.choice(list) draws
        obs, info = env.reset()
        assert np.array_equal(env._state, np.asarray(task["initial_states"][idx]))
        return idx, np.asarray(obs, np.float64), np.asarray(info["command"], np.float64), float(info["error"])

    acts = rng.uniform(-act_scale, act_scale, size=(T, env.pad_action_dim))
    acts[rng.random_sample(T) < 0.3] *= 0.5
    z = np.zeros((T, ns)); x_after = np.zeros((T, ns)); obs = np.zeros((T, 16)); cmd = np.zeros((T, 16))
    reward = np.zeros(T); error = np.zeros(T); term = np.zeros(T, np.uint8); trunc = np.zeros(T, np.uint8)
    steps = np.zeros(T, np.int64)
    reset_idx = np.full(T, -1, np.int64); reset_obs = np.zeros((T, 16)); reset_cmd = np.zeros((T, 16))
    reset_err = np.zeros(T)
    init_idx, init_obs, init_cmd, init_err = ref_reset(seed0 + 1)
    for t in range(T):
        np.random.seed(seed0 + 100 + t)
        o, r, te, tr, info = env.step(acts[t])
        z[t] = np.random.RandomState(seed0 + 100 + t).standard_normal(ns)
        x_after[t] = env._state
        obs[t] = o; cmd[t] = info["command"]; reward[t] = r; error[t] = info["error"]
        term[t] = te; trunc[t] = tr; steps[t] = info["steps"]
        if te or tr:
            reset_idx[t], reset_obs[t], reset_cmd[t], reset_err[t] = ref_reset(seed0 + 10000 + t)
    out = _linds_task_arrays(task)
    out.update(ref_phi=env.ld_phi, ref_gamma=env.ld_gamma, ref_xt=env.ld_Xt, dt=np.float64(env.dt),
               tr_action=acts, tr_z=z, tr_x=x_after, tr_obs=obs, tr_cmd=cmd, tr_reward=reward, tr_error=error,
               tr_term=term, tr_trunc=trunc, tr_steps=steps, tr_reset_idx=reset_idx, tr_reset_obs=reset_obs,
               tr_reset_cmd=reset_cmd, tr_reset_err=reset_err,
               init_idx=np.int64(init_idx), init_obs=init_obs, init_cmd=init_cmd, init_err=np.float64(init_err))
    path = os.path.join(GOLD, name + ".npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB; type", str(task["target_type"]), "delay",
          int(task["target_delay"]), "terminated:", int(term.sum()), "truncated:", int(trunc.sum()))


def gen_linds():
    _, ts = _refimport.linds()
    import random as pyrandom
    got = {"dynamic_target": 0, "static_target": 0}
    k = 0
    tasks = []
    while len(tasks) < 4:
        k += 1
        task = ts.LinearDSSampler(16, 8, 8, seed=k)   # not reproducible even with a seed: arrays are stored
        ty = str(task["target_type"])
        want = "dynamic_target" if len(tasks) in (0, 2, 3) else "static_target"
        if ty != want:
            continue
        if want == "static_target" and not np.any(task["command"]):
            continue
        tasks.append(task)
    tasks[0]["noise_drift"] = 0.0
    tasks[1]["noise_drift"] = 0.013
    tasks[2]["noise_drift"] = 0.02
    tasks[2]["max_steps"] = 60                       # truncation at steps >= max_steps - 1
    tasks[3]["initial_states"] = list(tasks[3]["initial_states"]) + [6.0 * np.random.RandomState(3).randn(16)]
    for i, task in enumerate(tasks):
        gen_linds_one("linds_16x8x8_%d" % i, task, seed0=5000 + 1000 * i)
    # one (32, 8, 8) task: the reference sampler's initial-state rejection loop does not terminate at ns = 32
    # (SURVEY.md §7), so its own building blocks are used with hand-built initial states
    np.random.seed(77)
    A, B, Cm, X, Y = ts.sample_variants_(32, 8, 8)
    task = dict(state_dim=32, observation_dim=8, action_dim=8, max_steps=400, ld_A=A, ld_B=B, ld_C=Cm, ld_X=X,
                ld_Y=Y, action_cost=0.01, reward_base=0.1, terminate_punish=3.0, reward_factor=0.5,
                target_valid=ts.sample_target_spaces_(8), target_type="dynamic_target",
                initial_states=[0.3 * np.random.randn(32) for _ in range(3)], noise_drift=0.01,
                command=ts.RandomFourier(8), target_delay=7)
    gen_linds_one("linds_32x8x8_0", task, T=384, seed0=9000)


def _maze_task_arrays(task):
    return dict(start=np.asarray(task["start"], np.int64), cell_walls=np.asarray(task["cell_walls"], np.int8),
                cell_texts=np.asarray(task["cell_texts"], np.int64), cell_size=np.float64(task["cell_size"]),
                ground_text=np.int64(task["ground_text"]), ceiling_text=np.int64(task["ceiling_text"]),
                step_reward=np.float64(task["step_reward"]), goal_reward=np.float64(task["goal_reward"]),
                collision_reward=np.float64(task["collision_reward"]), wall_height=np.float64(task["wall_height"]),
                agent_height=np.float64(task["agent_height"]), fol_angle=np.float64(task["fol_angle"]),
                commands_sequence=np.asarray(task["commands_sequence"], np.int64),
                landmarks_coordinates=np.asarray(task["landmarks_coordinates"], np.int64),
                cell_landmarks=np.asarray(task["cell_landmarks"], np.int8))


def gen_maze_one(name, task, T=256, max_steps=5000, seed0=0, res=32):
    """G-M for one task: a scripted trajectory through the reference MazeWorldContinuous3D.step(); pose, rules
    and frames recorded per step.  A few `inject` events place the agent next to its goal / advance the
    command age, so that goal reaching, the 500-step command limit and truncation all occur."""
    Maze, mts, dyn, rc = _refimport.mazeworld()
    env = Maze(enable_render=False, resolution=(res, res), max_steps=max_steps, visibility_3D=12.0,
               command_in_observation=False, action_space_type="Discrete16")
    env.set_task(task)
    core = env.maze_core
    obs0, info0 = env.reset()
    rng = np.random.RandomState(seed0)
    acts = rng.choice(16, size=T, p=np.array([2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 6, 3, 3, 3, 3]) / 30.0)
    rec = {k: [] for k in ("pos", "ori", "grid", "reward", "cmd_idx", "cmd_age", "term", "trunc", "steps",
                           "collision", "cmd_rgb")}
    inj_pose = np.full((T, 3), np.nan)     # pose forced BEFORE step t
    inj_age = np.full(T, -1, np.int64)     # _commands_exists forced BEFORE step t
    frames = np.zeros((T, res, res, 3), np.uint8)
    cs = float(task["cell_size"])
    for t in range(T):
        if t in (60, 130):   # stand in a free cell next to the current goal, facing it
            g = core._landmarks_coordinates[core._command]
            for d, ang in (((-1, 0), 0.0), ((1, 0), 3.1415926), ((0, -1), 1.5707963), ((0, 1), -1.5707963)):
                c = (g[0] + d[0], g[1] + d[1])
                if core._cell_walls[c] == 0:
                    core._agent_loc = [c[0] * cs + 0.5 * cs, c[1] * cs + 0.5 * cs]
                    core._agent_ori = ang
                    inj_pose[t] = [core._agent_loc[0], core._agent_loc[1], ang]
                    acts[t:t + 6] = 11          # walk forward at full speed
                    break
        if t == 200:
            core._commands_exists = 498
            inj_age[t] = 498
        obs, r, term, trunc, info = env.step(int(acts[t]))
        frames[t] = obs
        rec["pos"].append(np.array(core._agent_loc, np.float64)); rec["ori"].append(float(core._agent_ori))
        rec["grid"].append(np.array(core._agent_grid, np.int64)); rec["reward"].append(float(r))
        rec["cmd_idx"].append(int(core._commands_sequence_idx)); rec["cmd_age"].append(int(core._commands_exists))
        rec["term"].append(int(term)); rec["trunc"].append(int(trunc)); rec["steps"].append(int(info["steps"]))
        rec["collision"].append(float(core._collision_punish / task["collision_reward"]))
        rec["cmd_rgb"].append(np.asarray(info["command"], np.float32))
        env.need_reset = False      # keep stepping after truncation, as the core allows
    # a few continuous actions (python floats, as the Discrete tables hand them over)
    cont = np.array([[0.37, -0.8], [-1.7, 0.9], [0.0, 1.0], [0.013, 0.61], [-0.22, -1.4], [0.5, 0.5]])
    cont_pose = []
    for a in cont:
        core.do_action([float(a[0]), float(a[1])])
        cont_pose.append([core._agent_loc[0], core._agent_loc[1], float(core._agent_ori)])
    # frames at 64x64 for a subset of the recorded poses, straight from maze_view
    idx64 = np.arange(0, T, 32)
    f64 = np.zeros((len(idx64), 64, 64, 3), np.uint8)
    M = mts.MAZE_TASK_MANAGER
    for q, t in enumerate(idx64):
        img, _ = rc.maze_view(np.array(rec["pos"][t], dtype=np.float32), rec["ori"][t], core._agent_height,
                              core._cell_walls, core._cell_landmarks, core._cell_texts, core._cell_size,
                              M.textlib_walls, M.textlib_grounds[core._ground_text],
                              M.textlib_ceilings[core._ceiling_text], core._wall_height, 1.0, 12.0, 0.20,
                              core._fol_angle, 64, 64, rc.landmarks_rgb_arr)
        f64[q] = img.astype("uint8")
    out = _maze_task_arrays(task)
    out.update(max_steps=np.int64(max_steps), res=np.int64(res), actions=acts.astype(np.int64), inj_pose=inj_pose,
               inj_age=inj_age, frame0=np.asarray(obs0, np.uint8), frames=frames[::8], frame_steps=np.arange(0, T, 8),
               frames64=f64, frames64_steps=idx64, cont_actions=cont, cont_pose=np.array(cont_pose),
               **{"tr_" + k: np.array(v) for k, v in rec.items()})
    path = os.path.join(GOLD, name + ".npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB; goals reached:",
          int(np.sum(np.diff(np.array(rec["cmd_idx"])) > 0)), "truncated:", int(np.sum(rec["trunc"])),
          "contact steps:", int(np.sum(np.array(rec["collision"]) > 0)))


def gen_maze():
    Maze, mts, dyn, rc = _refimport.mazeworld()
    from xenoverse_amd.mazeworld.textures import make_texture_library
    lib = make_texture_library(8, 4, 4, seed=0)    # the reference's JPG assets are not used (nor copied)
    M = mts.MAZE_TASK_MANAGER
    M.textlib_walls, M.textlib_grounds, M.textlib_ceilings = lib["walls"], lib["grounds"], lib["ceilings"]
    for k in range(3):
        task = mts.MazeTaskSampler(n_range=(15, 16), seed=k, verbose=False)
        gen_maze_one("maze_15_seed%d" % k, task, max_steps=(5000 if k else 230), seed0=100 + k)


def gen_maze_agent_one(name, task, T, action_space, agent_kind, res=32, seed0=0, expose_all=False):
    """The reference's SmartSLAMAgent / OracleAgent driving its own MazeWorldContinuous3D: per agent.step() the env
    state it read (pose, cell, command, maze_core._cell_exposed) and what it produced (action, _mask_info, _cost_map,
    first two path cells).  expose_all: `random.random` of the ray caster returns 0.0, so _cell_exposed holds every cell
    DDA_2D lists (pins the cell lists; the 5 % sampling itself is an unseeded stream in the reference)."""
    import random
    Maze, mts, dyn, rc = _refimport.mazeworld()
    from xenoverse.mazeworld.agents.smart_slam_agent import SmartSLAMAgent
    from xenoverse.mazeworld.agents.oracle_agent import OracleAgent
    env = Maze(enable_render=False, resolution=(res, res), max_steps=5000, visibility_3D=12.0,
               command_in_observation=False, action_space_type=action_space)
    env.set_task(task)
    core = env.maze_core
    random.seed(seed0)
    real_random = rc.random.random
    if expose_all:
        rc.random.random = lambda: 0.0
    try:
        obs, info = env.reset()
        agent = (OracleAgent if agent_kind == "oracle" else SmartSLAMAgent)(maze_env=env, render=False)
        n = core._cell_walls.shape[0]
        rec = {k: [] for k in ("pos", "ori", "grid", "command", "cmd_idx", "steps", "exposed", "action", "mask", "cost",
                               "path_len", "path01")}
        r = 0
        for t in range(T):
            rec["pos"].append(np.array(core._agent_loc, np.float64)); rec["ori"].append(float(core._agent_ori))
            rec["grid"].append(np.array(core._agent_grid, np.int64)); rec["command"].append(int(core._command))
            rec["cmd_idx"].append(int(core._commands_sequence_idx)); rec["steps"].append(int(core.steps))
            rec["exposed"].append(np.array(core._cell_exposed, np.uint8))
            a = agent.step(obs, r)
            rec["action"].append(int(a)); rec["mask"].append(np.array(agent._mask_info, np.uint8))
            rec["cost"].append(np.array(agent._cost_map, np.float64)); rec["path_len"].append(len(agent._path))
            p = [tuple(int(v) for v in q) for q in agent._path[:2]] + [(-1, -1)]
            rec["path01"].append(np.array(p[:2], np.int64))
            obs, r, term, trunc, info = env.step(a)
            if term or trunc:
                break
    finally:
        rc.random.random = real_random
    out = _maze_task_arrays(task)
    out.update(res=np.int64(res), n_actions=np.int64(len(env.list_actions)), agent_kind=np.int64(agent_kind == "oracle"),
               expose_all=np.int64(expose_all), **{k: np.asarray(v) for k, v in rec.items()})
    path = os.path.join(GOLD, name + ".npz")
    np.savez_compressed(path, **out)
    print(name, len(rec["action"]), "steps", os.path.getsize(path) // 1024, "KiB")


def gen_maze_agent():
    Maze, mts, dyn, rc = _refimport.mazeworld()
    from xenoverse_amd.mazeworld.textures import make_texture_library
    lib = make_texture_library(8, 4, 4, seed=0)
    M = mts.MAZE_TASK_MANAGER
    M.textlib_walls, M.textlib_grounds, M.textlib_ceilings = lib["walls"], lib["grounds"], lib["ceilings"]
    t11 = mts.MazeTaskSampler(n_range=(11, 12), seed=3, verbose=False)
    t15 = mts.MazeTaskSampler(n_range=(15, 16), seed=5, verbose=False)
    t21 = mts.MazeTaskSampler(n_range=(21, 22), seed=7, verbose=False)
    gen_maze_agent_one("agent_slam_11_d16", t11, 260, "Discrete16", "slam", seed0=1)
    gen_maze_agent_one("agent_slam_15_d32", t15, 320, "Discrete32", "slam", seed0=2)
    gen_maze_agent_one("agent_slam_21_d16", t21, 200, "Discrete16", "slam", seed0=3)
    gen_maze_agent_one("agent_oracle_15_d16", t15, 200, "Discrete16", "oracle", seed0=4)
    gen_maze_agent_one("agent_slam_11_all", t11, 80, "Discrete16", "slam", seed0=5, expose_all=True)


def gen_maze_sampled(n_tasks=24, seed0=10):
    """Tasks of the reference's MazeTaskSampler over its default size range (and a few without loops): topology,
    textures, landmarks, start, commands — the build's sampler consumes the same stream and must return the same task."""
    Maze, mts, dyn, rc = _refimport.mazeworld()
    from xenoverse_amd.mazeworld.textures import make_texture_library
    lib = make_texture_library(8, 4, 4, seed=0)
    M = mts.MAZE_TASK_MANAGER
    M.textlib_walls, M.textlib_grounds, M.textlib_ceilings = lib["walls"], lib["grounds"], lib["ceilings"]
    out = {k: [] for k in ("seed", "allow_loops", "n", "cell_walls", "cell_texts", "cell_landmarks", "start", "n_landmarks",
                           "landmarks", "commands", "scalars")}
    for k in range(n_tasks):
        loops = (k % 4) != 3
        t = mts.MazeTaskSampler(seed=seed0 + k, allow_loops=loops, commands_sequence=32, verbose=False)
        n = t["cell_walls"].shape[0]
        pad = lambda a, fill: np.pad(np.asarray(a), ((0, 25 - n), (0, 25 - n)), constant_values=fill)
        out["seed"].append(seed0 + k); out["allow_loops"].append(loops); out["n"].append(n)
        out["cell_walls"].append(pad(t["cell_walls"], 1).astype(np.int8)); out["cell_texts"].append(pad(t["cell_texts"], 0).astype(np.int64))
        out["cell_landmarks"].append(pad(t["cell_landmarks"], -1).astype(np.int8)); out["start"].append(np.asarray(t["start"], np.int64))
        lm = np.full((15, 2), -1, np.int64); lm[:len(t["landmarks_coordinates"])] = np.asarray(t["landmarks_coordinates"], np.int64)
        out["landmarks"].append(lm); out["n_landmarks"].append(len(t["landmarks_coordinates"]))
        out["commands"].append(np.asarray(t["commands_sequence"], np.int64))
        out["scalars"].append([t["cell_size"], t["wall_height"], t["agent_height"], t["fol_angle"], t["goal_reward"],
                               t["ground_text"], t["ceiling_text"]])
    path = os.path.join(GOLD, "sampler_refmazes.npz")
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in out.items()})
    print("sampler_refmazes.npz", os.path.getsize(path) // 1024, "KiB")


def gen_maze_refcounts(seeds=(0, 1, 2, 3, 40, 41)):
    """Tasks of the reference's MazeTaskSampler with ITS OWN texture folder behind it (37 / 29 / 21 images; only the
    folder's sizes enter a task — texture ids are drawn below them): what `MazeTaskSampler(seed=k)` must return with
    its default library sizes.  Arrays only; no image is read into the fixture."""
    Maze, mts, dyn, rc = _refimport.mazeworld()
    M = mts.MAZE_TASK_MANAGER
    counts = [int(len(M.textlib_walls)), int(len(M.textlib_grounds)), int(len(M.textlib_ceilings))]
    out = {k: [] for k in ("seed", "n", "cell_walls", "cell_texts", "cell_landmarks", "start", "commands", "landmarks",
                           "n_landmarks", "scalars")}
    for seed in seeds:
        t = mts.MazeTaskSampler(seed=seed, commands_sequence=32, verbose=False)
        n = t["cell_walls"].shape[0]
        pad = lambda a: np.pad(np.asarray(a), ((0, 25 - n), (0, 25 - n)))
        lm = np.zeros((15, 2), np.int64); lm[:len(t["landmarks_coordinates"])] = np.asarray(t["landmarks_coordinates"])
        out["seed"].append(seed); out["n"].append(n)
        out["cell_walls"].append(pad(t["cell_walls"])); out["cell_texts"].append(pad(t["cell_texts"]))
        out["cell_landmarks"].append(pad(t["cell_landmarks"])); out["start"].append(np.asarray(t["start"]))
        out["commands"].append(np.asarray(t["commands_sequence"])); out["landmarks"].append(lm)
        out["n_landmarks"].append(len(t["landmarks_coordinates"]))
        out["scalars"].append([t["cell_size"], t["wall_height"], t["agent_height"], t["fol_angle"], t["goal_reward"],
                               t["ground_text"], t["ceiling_text"]])
    path = os.path.join(GOLD, "sampler_refmazes_refcounts.npz")
    np.savez_compressed(path, counts=np.asarray(counts), **{k: np.asarray(v) for k, v in out.items()})
    print("sampler_refmazes_refcounts.npz", os.path.getsize(path) // 1024, "KiB", counts)


def gen_linds_sampled():
    """Tasks of the reference's LinearDSSampler with its seeding function pinned: the reference seeds NumPy with
    timestamp + system random + seed (utils/random_nn.py:9-16, non-reproducible), so for the fixture pseudo_random_seed
    is replaced by `numpy.random.seed(seed)` — the sampler's own code then runs on a known stream."""
    _, ts = _refimport.linds()
    ts.pseudo_random_seed = lambda seed=0: np.random.seed(seed)
    out = {k: [] for k in ("seed", "dims", "max_steps", "A", "B", "C", "X", "Y", "scal", "valid", "is_dyn", "n_init", "init",
                           "noise_drift", "delay", "cmd", "four_n", "four_orders", "four_coeffs")}
    cases = [(16, 8, 8, s) for s in range(10)] + [(8, 4, 4, 20), (8, 4, 4, 21), (4, 2, 3, 30), (16, 8, 8, 77)]
    for ns, na, no, seed in cases:
        t = ts.LinearDSSampler(ns, na, no, seed=seed)
        pad2 = lambda a, r, c: np.pad(np.asarray(a, np.float64), ((0, r - np.shape(a)[0]), (0, c - np.shape(a)[1])))
        pad1 = lambda a, n: np.pad(np.asarray(a, np.float64), (0, n - len(a)))
        out["seed"].append(seed); out["dims"].append([ns, na, no]); out["max_steps"].append(t["max_steps"])
        out["A"].append(pad2(t["ld_A"], 16, 16)); out["B"].append(pad2(t["ld_B"], 16, 8)); out["C"].append(pad2(t["ld_C"], 16, 16))
        out["X"].append(pad1(t["ld_X"], 16)); out["Y"].append(pad1(t["ld_Y"], 16))
        out["scal"].append([t["action_cost"], t["reward_base"], t["terminate_punish"], t["reward_factor"]])
        out["valid"].append(pad1(t["target_valid"], 16)); dyn = t["target_type"] == "dynamic_target"
        out["is_dyn"].append(dyn); out["n_init"].append(len(t["initial_states"]))
        init = np.zeros((8, 16)); init[:len(t["initial_states"]), :ns] = np.asarray(t["initial_states"])[:8]
        out["init"].append(init); out["noise_drift"].append(t["noise_drift"]); out["delay"].append(t["target_delay"])
        cmd = np.zeros(16); orders = np.zeros(6); coeffs = np.zeros((6, 16, 2)); n_terms = 0
        if dyn:
            n_terms = len(t["command"].coeffs)
            for k, (o, c) in enumerate(t["command"].coeffs):
                orders[k] = o; coeffs[k, :no] = c
        else:
            cmd[:no] = t["command"]
        out["cmd"].append(cmd); out["four_n"].append(n_terms); out["four_orders"].append(orders); out["four_coeffs"].append(coeffs)
    path = os.path.join(GOLD, "sampler_reflinds.npz")
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in out.items()})
    print("sampler_reflinds.npz", os.path.getsize(path) // 1024, "KiB")


def gen_acrobot():
    """The reference's OWN Acrobot code: RandomAcrobotEnv._dsdt and ._terminal (random_acrobot.py:58-101), called on
    random inputs with tasks drawn over sample_acrobot's ranges, and the reset-state formula (:123-125).  The
    integrator around them (AcrobotEnv.step / rk4 / wrap / bound) is gymnasium's and is not installed: unpinned."""
    import importlib.util
    _refimport.setup()
    path = os.path.join(_refimport.REF_ROOT, "xenoverse", "metacontrol", "random_acrobot.py")
    spec = importlib.util.spec_from_file_location("_ref_random_acrobot", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    rng = np.random.RandomState(20260901)
    n = 2000
    keys = ("link_length_1", "link_length_2", "link_mass_1", "link_mass_2", "link_com_1", "link_com_2", "gravity")
    prm = np.zeros((n, 7)); y = np.zeros((n, 5)); out = np.zeros((n, 5)); term = np.zeros(n, np.uint8)
    for i in range(n):
        env = object.__new__(mod.RandomAcrobotEnv)
        l1, l2 = float(rng.uniform(0.5, 3.0)), float(rng.uniform(0.5, 3.0))
        task = dict(link_length_1=l1, link_length_2=l2, link_mass_1=float(rng.uniform(0.5, 3.0)),
                    link_mass_2=float(rng.uniform(0.5, 3.0)), link_com_1=float(rng.uniform(0.25, 0.75)) * l1,
                    link_com_2=float(rng.uniform(0.25, 0.75)) * l2, gravity=float(rng.uniform(1.0, 15.0)))
        for k, v in task.items():         # what set_task does (:103-106), without its print
            setattr(env, k, v)
        prm[i] = [task[k] for k in keys]
        s = np.array([rng.uniform(-np.pi, np.pi), rng.uniform(-np.pi, np.pi), rng.uniform(-4 * np.pi, 4 * np.pi),
                      rng.uniform(-9 * np.pi, 9 * np.pi), float(rng.randint(0, 3) - 1)], np.float64)
        if i % 5 == 0:
            s[:4] *= 0.05                  # near the hanging rest pose, where episodes start
        y[i] = s
        out[i] = np.asarray(env._dsdt(s), np.float64)
        env.state = s[:4]
        term[i] = env._terminal()
    # reset(): uniform(-1, 1, 4).astype(float32) * reset_bounds_scale for the scalar (registered) and list forms
    u = rng.random_sample((64, 4))
    r32 = (-1.0 + 2.0 * u).astype(np.float32)
    reset_scalar = (r32 * 0.10).astype(np.float64)                         # float32 array * python float
    scale_vec = np.array([0.3, 0.2, 0.1, 0.05])
    reset_vector = r32 * scale_vec                                         # float32 array * float64 array
    assert (r32 * 0.10).dtype == np.float32 and reset_vector.dtype == np.float64
    np.savez_compressed(os.path.join(GOLD, "acrobot_dsdt.npz"), params=prm, y=y, dsdt=out, terminal=term,
                        reset_u=u, reset_scalar=reset_scalar, reset_vector=reset_vector, scale_vec=scale_vec)
    print("acrobot_dsdt.npz: %d derivative vectors, %d terminal" % (n, int(term.sum())))


def gen_garnet():
    """One seeded task of the reference's GarnetTaskSampler (task_sampler.py:120-160): the build's sampler follows the
    same draw order on RandomState(seed) and must reproduce it bit for bit."""
    _refimport.setup()
    from xenoverse.anymdp.task_sampler import GarnetTaskSampler
    g = GarnetTaskSampler(8, 2, b=2, sigma=0.1, seed=3)
    np.savez_compressed(os.path.join(GOLD, "garnet_8x2_seed3.npz"), max_steps=np.float64(g["max_steps"]),
                        state_mapping=np.asarray(g["state_mapping"], np.int64), transition=g["transition"],
                        reward=g["reward"])
    print("garnet_8x2_seed3.npz")


def gen_anymdp_vi():
    """The reference's own update_value_matrix (solver.py:57-82) on reference-sampled tasks: greedy and uniform-policy
    values at the repair loop's gamma (0.99) and at check_valuefunction's gamma (2^(-1/ns)), plus a warm-started chain
    with shifted terminal rewards (how sample_mdp's repair loop calls it).  Pins oracle xo_update_value_matrix."""
    _refimport.setup()
    from xenoverse.anymdp.solver import update_value_matrix
    out = {}
    for seed in (0, 2):
        task = sample_ref_tasks.get(16, 4, seed)
        T, R = np.asarray(task["transition"], np.float64), np.asarray(task["reward"], np.float64)
        ns, na, _ = T.shape
        out["T%d" % seed], out["R%d" % seed] = T, R
        for gi, gamma in enumerate((0.99, float(np.power(2, -1.0 / ns)))):
            for greedy in (1, 0):
                out["vm_s%d_g%d_%d" % (seed, gi, greedy)] = update_value_matrix(T, R, gamma, np.zeros((ns, na)),
                                                                               is_greedy=bool(greedy))
        out["gamma%d" % seed] = np.array([0.99, float(np.power(2, -1.0 / ns))])
        vm = np.zeros((ns, na))
        bonus = np.zeros(ns)
        rng = np.random.RandomState(seed)
        for k in range(3):        # warm starts, as the repair loop does
            vm = update_value_matrix(T, R + bonus[None, None, :], 0.99, vm)
            out["chain_s%d_%d" % (seed, k)] = np.copy(vm)
            out["chain_bonus_s%d_%d" % (seed, k)] = np.copy(bonus)
            bonus[-1] += rng.uniform(1.0, 10.0)
            bonus[rng.randint(0, ns - 1)] -= rng.uniform(1.0, 10.0)
    np.savez_compressed(os.path.join(GOLD, "sampler_vi_ref.npz"), **out)
    print("sampler_vi_ref.npz", os.path.getsize(os.path.join(GOLD, "sampler_vi_ref.npz")) // 1024, "KiB")


def gen_anymdp_sampled(n=int(os.environ.get("XV_REFPOP_N", "128")), seed0=100):
    """n tasks of the reference's AnyMDPTaskSampler(16, 4, seed = seed0 + k) with the sampler's own bookkeeping: how
    many candidates sample_mdp produced, how many of them could not be repaired (None) and how many
    check_valuefunction rejected.  Used (a) for the seed-compatibility test of the build's sampler and (b) as the
    reference population for the distribution tests of the device sampler."""
    _, ts = _refimport.anymdp()
    import numpy.random as npr
    counts = {"cand": 0, "none": 0, "rej": 0}
    real_sample, real_check = ts.sample_mdp, ts.check_valuefunction

    def sample_mdp(*a, **k):
        r = real_sample(*a, **k)
        counts["cand"] += 1
        counts["none"] += r is None
        return r

    def check(*a, **k):
        ok = real_check(*a, **k)
        counts["rej"] += not ok
        return ok
    ts.sample_mdp, ts.check_valuefunction = sample_mdp, check
    keys = ("transition", "reward", "reward_noise")
    out = {k: [] for k in keys}
    meta = {k: [] for k in ("max_steps", "state_mapping", "s_e_mask", "s_0_mask", "s_0_prob", "goal", "n_cand", "n_none",
                            "n_rej", "seed")}
    try:
        for k in range(n):
            for c in counts:
                counts[c] = 0
            task = ts.AnyMDPTaskSampler(16, 4, seed=seed0 + k)
            for kk in keys:
                out[kk].append(np.asarray(task[kk], np.float64))
            se = np.zeros(16, np.uint8); se[np.asarray(task["s_e"], np.int64).reshape(-1)] = 1
            s0 = np.zeros(16, np.uint8); s0[np.asarray(task["s_0"], np.int64)] = 1
            p0 = np.zeros(16); p0[np.asarray(task["s_0"], np.int64)] = task["s_0_prob"]
            meta["max_steps"].append(task["max_steps"]); meta["state_mapping"].append(task["state_mapping"])
            meta["s_e_mask"].append(se); meta["s_0_mask"].append(s0); meta["s_0_prob"].append(p0)
            meta["goal"].append(bool(task["final_goal_terminate"]))
            meta["n_cand"].append(counts["cand"]); meta["n_none"].append(counts["none"]); meta["n_rej"].append(counts["rej"])
            meta["seed"].append(seed0 + k)
            print("seed", seed0 + k, dict(counts), flush=True)
    finally:
        ts.sample_mdp, ts.check_valuefunction = real_sample, real_check
    path = os.path.join(GOLD, "sampler_refpop_16x4.npz")
    np.savez_compressed(path, **{k: np.stack(v) for k, v in out.items()}, **{k: np.asarray(v) for k, v in meta.items()})
    print("sampler_refpop_16x4.npz", os.path.getsize(path) // 1024, "KiB")


FAMILIES = {"anymdp": gen_anymdp, "linds": gen_linds, "maze": gen_maze, "acrobot": gen_acrobot, "garnet": gen_garnet,
            "anymdp_vi": gen_anymdp_vi, "anymdp_sampled": gen_anymdp_sampled, "maze_sampled": gen_maze_sampled,
            "linds_sampled": gen_linds_sampled, "maze_agent": gen_maze_agent, "maze_refcounts": gen_maze_refcounts}

if __name__ == "__main__":
    os.makedirs(GOLD, exist_ok=True)
    which = sys.argv[1:] or list(FAMILIES)
    for f in which:
        FAMILIES[f]()
