"""The device maze teachers (csrc/maze_agent.hip through xv_maze_agent_*) against the reference agent's recorded
decisions (tests/golden/agent_*.npz) and against the oracle restatement on free-running batches."""
import numpy as np
import pytest
import torch

import oracle
from xenoverse_amd.mazeworld import (DEFAULT_ACTION_SPACE_16, DEFAULT_ACTION_SPACE_32, MazeWorldVecEnv, OracleAgent,
                                     SmartSLAMAgent, build_tables, make_texture_library, teacher_rollout)
from util import golden_files, load_maze_golden

pytestmark = pytest.mark.gpu
FILES = golden_files("agent_")
TEX = None


def tex():
    global TEX
    if TEX is None:
        TEX = make_texture_library(8, 4, 4, seed=0)
    return TEX


def _np(t):
    return t.detach().cpu().numpy()


def _pad(a, NG, reps):
    out = np.zeros((reps, NG, NG), np.uint8)
    out[:, :a.shape[0], :a.shape[1]] = a
    return out


@pytest.mark.parametrize("path", FILES)
def test_reference_decisions_replayed_on_the_device(path):
    """every agent.step() of the reference run, the device agent fed with the env state and _cell_exposed the reference
    read: same memory, same cost map (1e-12), same path head, same action — in each of 3 replicas of the env"""
    g, task = load_maze_golden(path)
    res, na, reps = int(g["res"]), int(g["n_actions"]), 3
    env = MazeWorldVecEnv(reps, resolution=(res, res), textures=tex(), autoreset_mode="disabled", max_steps=5000,
                          action_space_type="Discrete%d" % na)
    env.set_task(task)
    env.reset()
    agent = (OracleAgent if int(g["agent_kind"]) else SmartSLAMAgent)(maze_env=env, keep_cost_map=True)
    n, NG = task["cell_walls"].shape[0], int(env._tab["walls"].shape[-1])
    T = len(g["action"])
    wrong = 0
    for t in range(T):
        env.set_state(pos=np.repeat(g["pos"][t].reshape(2, 1), reps, 1), ori=np.full(reps, g["ori"][t]),
                      steps=np.full(reps, g["steps"][t], np.int32), cmd_idx=np.full(reps, g["cmd_idx"][t], np.int32))
        a = _np(agent.step(None, None, exposed=_pad(g["exposed"][t], NG, reps)))
        s = {k: _np(v) for k, v in agent.inspect(cost=True).items()}
        for r in range(reps):
            assert np.array_equal(s["mask"][r, :n, :n], g["mask"][t]), t
            assert np.allclose(s["cost"][r, :n, :n], g["cost"][t], rtol=1e-12, atol=1e-12), t
            assert int(s["path"][r, 0]) == int(g["path_len"][t]) and np.array_equal(s["path"][r, 1:3], g["path01"][t][0]), t
            if g["path_len"][t] > 1:
                assert np.array_equal(s["path"][r, 3:5], g["path01"][t][1]), t
        assert (a == a[0]).all()
        wrong += int(a[0] != g["action"][t])
    assert wrong == 0, (wrong, T)
    agent.close(); env.close()


def _keep_uniforms(seed, gid_base, tick, n_env, G2):
    """the device's memory-keep draws (xv_env_draw_sub, purpose 6, sub = cell >> 2, word cell & 3) as doubles"""
    out = np.zeros((n_env, G2))
    for e in range(n_env):
        w = np.concatenate([oracle.env_draw_sub(seed, gid_base + e, tick, 6, sub) for sub in range((G2 + 3) // 4)])[:G2]
        out[e] = w.astype(np.float64) / 4294967296.0
    return out


@pytest.mark.parametrize("kind,stm,keep", [("slam", 3, 1.0), ("oracle", 3, 1.0), ("slam", 2, 0.6), ("slam", 0, 1.0)])
def test_free_running_batch_vs_oracle(kind, stm, keep):
    """3 mazes x 6 envs, the device agent drives the device envs (exposure from the ray caster's walk and the device
    draws) through episode ends; the oracle gets the same state every step: exposure maps, memory, cost maps, path heads
    bit-equal / 1e-12, actions equal"""
    tasks = [load_maze_golden(p)[1] for p in FILES[:3]]
    tab = build_tables(tasks)
    env_task = np.repeat(np.arange(len(tasks), dtype=np.int32), 6)
    n, NG = len(env_task), int(tab["NG"])
    seed, base = 2024, 77
    env = MazeWorldVecEnv(n, resolution=(32, 32), textures=tex(), autoreset_mode="same_step", max_steps=45,
                          action_space_type="Discrete16", seed=seed, env_id_base=base)
    env.set_task(tasks, env_task_index=env_task)
    ora = oracle.MazeOracle(tab, tex(), env_task, resolution=(32, 32), max_steps=45)
    table = np.array(DEFAULT_ACTION_SPACE_16, np.float64)
    env.reset(); ora.reset()
    cls = OracleAgent if kind == "oracle" else SmartSLAMAgent
    agent = cls(maze_env=env, keep_cost_map=True, short_term_memory_size=stm, memory_keep_ratio=keep)
    oag = oracle.MazeAgentOracle(ora, table, short_term_memory_size=stm, memory_keep_ratio=keep,
                                 oracle_agent=(kind == "oracle"))
    wrong = 0
    for t in range(110):
        tick = env.engine.tick
        a = _np(agent.step())
        s = {k: _np(v) for k, v in agent.inspect(cost=True).items()}
        ex = ora.expose(seed, base, tick)
        assert np.array_equal(s["exposed"], ex), t
        uk = _keep_uniforms(seed, base, tick, n, NG * NG) if keep < 1.0 else None
        ao = oag.act(ex, uk)
        assert np.array_equal(s["mask"], oag.mask), t
        for e in range(n):
            m = tasks[env_task[e]]["cell_walls"].shape[0]
            assert np.allclose(s["cost"][e, :m, :m], oag.cost[e, :m, :m], rtol=1e-12, atol=1e-12), (t, e)
        assert np.array_equal(s["path"], oag.path), t
        wrong += int((a != ao).sum())
        frames, r, term, trunc, info = env.step(a)
        ora.step(table[a], 2)
        st = env.get_state()
        assert np.max(np.abs(_np(st["pos"]) - ora.pos)) < 1e-9 and np.array_equal(_np(st["steps"]), ora.steps)
        ora.pos[:] = _np(st["pos"]); ora.ori[:] = _np(st["ori"])
    assert wrong == 0, wrong
    assert int(_np(env.get_state()["steps"]).max()) < 45        # episodes ended and restarted with fresh agents
    agent.close(); env.close()


def test_teachers_reach_their_goals():
    """an OracleAgent batch walks to its commanded landmarks (goal rewards arrive); the SLAM agent explores (its memory
    grows) and collects goals too; nothing leaves the device inside the roll-out"""
    tasks = [load_maze_golden(p)[1] for p in FILES[:3]]
    env_task = np.repeat(np.arange(3, dtype=np.int32), 16)
    for cls in (OracleAgent, SmartSLAMAgent):
        env = MazeWorldVecEnv(len(env_task), resolution=(32, 32), textures=tex(), autoreset_mode="same_step",
                              max_steps=400, action_space_type="Discrete16", seed=5)
        env.set_task(tasks, env_task_index=env_task)
        env.reset()
        agent = cls(maze_env=env)
        out = teacher_rollout(env, agent, 300)
        goal = float(tasks[0]["goal_reward"])
        goals = (out["reward"] > 0.5 * goal).sum().item()
        assert out["action"].shape == (300, len(env_task)) and out["action"].is_cuda
        assert goals >= (len(env_task) if cls is OracleAgent else len(env_task) // 3), (cls.__name__, goals)
        if cls is SmartSLAMAgent:
            known = agent.inspect()["mask"].sum(dim=(1, 2)).float().mean().item()
            assert known > 30
        agent.close(); env.close()


def test_large_mazes_take_the_multi_wave_path():
    """35x35 and 41x41 mazes (more than 1,024 cells: four waves per env with workgroup barriers instead of one wave),
    Discrete32, against the oracle agent through restarts"""
    from xenoverse_amd.mazeworld import MazeTaskSampler
    tasks = [MazeTaskSampler(n_range=(35, 36), seed=1, n_wall_textures=8, n_ground_textures=4, n_ceiling_textures=4),
             MazeTaskSampler(n_range=(41, 42), seed=2, n_wall_textures=8, n_ground_textures=4, n_ceiling_textures=4)]
    tab = build_tables(tasks)
    env_task = np.repeat(np.arange(2, dtype=np.int32), 5)
    n = len(env_task)
    seed, base = 5, 9
    env = MazeWorldVecEnv(n, resolution=(48, 32), textures=tex(), autoreset_mode="same_step", max_steps=60,
                          action_space_type="Discrete32", seed=seed, env_id_base=base)
    env.set_task(tasks, env_task_index=env_task)
    ora = oracle.MazeOracle(tab, tex(), env_task, resolution=(48, 32), max_steps=60)
    table = np.array(DEFAULT_ACTION_SPACE_32, np.float64)
    env.reset(); ora.reset()
    agent = SmartSLAMAgent(maze_env=env, keep_cost_map=True)
    oag = oracle.MazeAgentOracle(ora, table)
    wrong = 0
    for t in range(90):
        tick = env.engine.tick
        a = _np(agent.step())
        s = {k: _np(v) for k, v in agent.inspect(cost=True).items()}
        ex = ora.expose(seed, base, tick)
        assert np.array_equal(s["exposed"], ex), t
        ao = oag.act(ex)
        assert np.array_equal(s["mask"], oag.mask) and np.array_equal(s["path"], oag.path), t
        for e in range(n):
            m = tasks[env_task[e]]["cell_walls"].shape[0]
            assert np.allclose(s["cost"][e, :m, :m], oag.cost[e, :m, :m], rtol=1e-12, atol=1e-12), (t, e)
        wrong += int((a != ao).sum())
        frames = env.step(a)[0]; ora.step(table[a], 2)
        st = env.get_state()
        assert np.max(np.abs(_np(st["pos"]) - ora.pos)) < 1e-9
        ora.pos[:] = _np(st["pos"]); ora.ori[:] = _np(st["ori"])
        assert np.array_equal(_np(st["steps"]), ora.steps)
        if t % 30 == 0:      # grids beyond 32 x 32 through the move and ray-cast kernels as well
            from util import frame_mismatch
            frac, worst = frame_mismatch(_np(frames), ora.render(n_threads=4)[0])
            assert frac <= 0.005 and worst <= 1, (t, frac, worst)
    assert wrong == 0, wrong
    agent.close(); env.close()


def test_agent_lifetime_and_misuse():
    """agents need a Discrete action table; an env that takes a new task (or closes) releases its agents first, and a
    stale agent refuses to step"""
    tasks = [load_maze_golden(p)[1] for p in FILES[:2]]
    envc = MazeWorldVecEnv(4, resolution=(16, 16), textures=tex(), action_space_type="Continuous")
    envc.set_task(tasks[0]); envc.reset()
    with pytest.raises(Exception, match="Discrete16 or Discrete32"):
        SmartSLAMAgent(maze_env=envc)
    with pytest.raises(Exception, match="Must use maze_env"):
        SmartSLAMAgent()
    envc.close()
    env = MazeWorldVecEnv(4, resolution=(16, 16), textures=tex(), action_space_type="Discrete16")
    env.set_task(tasks[0]); env.reset()
    a1 = SmartSLAMAgent(maze_env=env)
    a1.step()
    env.set_task(tasks[1]); env.reset()          # new handle: a1 was closed with the old one
    assert a1._h is None and env._agents == []
    with pytest.raises(Exception, match="create a new agent"):
        a1.step()
    a2 = OracleAgent(maze_env=env)
    assert a2.step().shape == (4,)
    env.close()
    assert a2._h is None
