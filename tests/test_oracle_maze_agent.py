"""Pin oracle/xeno_oracle_agent.c (SmartSLAMAgent / OracleAgent restatement) and xo_maze_expose to the reference's own
agent (tests/golden/agent_*.npz, made by oracle/gen_golden.py `maze_agent` from
/root/reference/xenoverse/mazeworld/agents driving the reference env)."""
import numpy as np
import pytest

import oracle
from xenoverse_amd.mazeworld.tables import DEFAULT_ACTION_SPACE_16, DEFAULT_ACTION_SPACE_32, build_tables
from xenoverse_amd.mazeworld.textures import make_texture_library
from util import golden_files, load_maze_golden

FILES = golden_files("agent_")
TEX = None


def textures():
    global TEX
    if TEX is None:
        TEX = make_texture_library(8, 4, 4, seed=0)
    return TEX


def test_golden_present():
    assert len(FILES) >= 5


def _setup(g, task):
    tab = build_tables([task])
    res = int(g["res"])
    o = oracle.MazeOracle(tab, textures(), [0], resolution=(res, res), max_steps=5000)
    o.reset()
    acts = DEFAULT_ACTION_SPACE_16 if int(g["n_actions"]) == 16 else DEFAULT_ACTION_SPACE_32
    ag = oracle.MazeAgentOracle(o, acts, oracle_agent=bool(g["agent_kind"]))
    return tab, o, ag


def _put(o, g, t):
    o.pos[:, 0] = g["pos"][t]; o.ori[0] = g["ori"][t]; o.grid[:, 0] = g["grid"][t]
    o.cmd_idx[0] = g["cmd_idx"][t]; o.steps[0] = g["steps"][t]


def _pad(a, NG):
    out = np.zeros((1, NG, NG), np.uint8)
    out[0, :a.shape[0], :a.shape[1]] = a
    return out


@pytest.mark.parametrize("path", FILES)
def test_agent_replay_matches_the_reference(path):
    """every agent.step() of the reference run, fed with the env state and _cell_exposed it read: same memory, same
    cost map (1e-12: numpy's cos/sin against libm's in the start cells), same path head, same action"""
    g, task = load_maze_golden(path)
    tab, o, ag = _setup(g, task)
    n, NG, T = task["cell_walls"].shape[0], int(tab["NG"]), len(g["action"])
    bad_actions = 0
    for t in range(T):
        _put(o, g, t)
        assert int(tab["commands"][0, min(int(g["cmd_idx"][t]), tab["commands"].shape[1] - 1)]) == int(g["command"][t])
        a = ag.act(_pad(g["exposed"][t], NG))
        assert np.array_equal(ag.mask[0, :n, :n], g["mask"][t]), t
        c = ag.cost[0, :n, :n]
        assert np.allclose(c, g["cost"][t], rtol=1e-12, atol=1e-12), (t, np.abs(c - g["cost"][t]).max())
        assert int(ag.path[0, 0]) == int(g["path_len"][t]), t
        assert np.array_equal(ag.path[0, 1:3], g["path01"][t][0]), t
        if g["path_len"][t] > 1:
            assert np.array_equal(ag.path[0, 3:5], g["path01"][t][1]), t
        bad_actions += int(a[0] != g["action"][t])
    assert bad_actions == 0, (bad_actions, T)


def test_expose_lists_the_cells_the_reference_lists():
    """with the 5 % draw forced to succeed, _cell_exposed is the union of DDA_2D's exposed_cell lists: bit-equal"""
    path = [f for f in FILES if f.endswith("_all.npz")][0]
    g, task = load_maze_golden(path)
    tab, o, ag = _setup(g, task)
    n = task["cell_walls"].shape[0]
    for t in range(len(g["action"])):
        _put(o, g, t)
        ex = o.expose(1, 0, t, prob=1.0)
        assert np.array_equal(ex[0, :n, :n], g["exposed"][t]), t


def test_expose_rate_matches_the_reference():
    """the sampled runs: the fraction of listed cells that got marked is 5 % per (ray, cell) in both"""
    path = [f for f in FILES if "slam_15" in f][0]
    g, task = load_maze_golden(path)
    tab, o, ag = _setup(g, task)
    n = task["cell_walls"].shape[0]
    ref_marks = own_marks = 0
    for t in range(len(g["action"])):
        _put(o, g, t)
        full = o.expose(1, 0, t, prob=1.0)[0, :n, :n]
        own = o.expose(7, 3, t, prob=0.05)[0, :n, :n]
        assert not np.any(own & ~full) and not np.any(g["exposed"][t] & ~full)     # marks only where a ray passes
        ref_marks += int(g["exposed"][t].sum()); own_marks += int(own.sum())
    assert abs(own_marks - ref_marks) < 0.15 * ref_marks, (own_marks, ref_marks)


def test_search_action_matches_numpy_restatement():
    rng = np.random.RandomState(0)
    acts = np.array(DEFAULT_ACTION_SPACE_32, np.float64)
    for k in range(200):
        ori = rng.uniform(-3.1, 3.1)
        t1 = rng.uniform(-2, 2, 2)
        t2 = rng.uniform(-2, 2, 2) if k % 3 else None
        a = oracle.maze_search_action(ori, t1, t2, acts)
        assert 0 <= a < 32
