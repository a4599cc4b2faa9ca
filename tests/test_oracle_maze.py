"""Pin oracle/xeno_oracle.c (MazeWorld) to the reference's own outputs (tests/golden/maze_*.npz, made by
oracle/gen_golden.py from /root/reference/xenoverse/mazeworld with procedural textures injected)."""
import numpy as np
import pytest

import oracle
from xenoverse_amd.mazeworld.tables import DEFAULT_ACTION_SPACE_16, build_tables
from xenoverse_amd.mazeworld.textures import make_texture_library
from util import frame_mismatch, golden_files, load_maze_golden

FILES = golden_files("maze_")
TEX = None


def textures():
    global TEX
    if TEX is None:
        TEX = make_texture_library(8, 4, 4, seed=0)
    return TEX


def test_golden_present():
    assert len(FILES) >= 3


def replay(g, task, res, stepper):
    """drive `stepper` (reset/step/set hooks) through the golden action script with its inject events"""
    T = len(g["actions"])
    table = np.array(DEFAULT_ACTION_SPACE_16, np.float64)
    out = []
    for t in range(T):
        if not np.isnan(g["inj_pose"][t, 0]):
            stepper.set_pose(g["inj_pose"][t])
        if g["inj_age"][t] >= 0:
            stepper.set_age(int(g["inj_age"][t]))
        out.append(stepper.step(table[g["actions"][t]]))
    return out


class OracleStepper(object):
    def __init__(self, g, task, res):
        tab = build_tables([task])
        self.o = oracle.MazeOracle(tab, textures(), [0], resolution=(res, res), max_steps=int(g["max_steps"]))
        self.o.reset()

    def set_pose(self, p):
        self.o.pos[:, 0] = p[:2]; self.o.ori[0] = p[2]

    def set_age(self, a):
        self.o.cmd_age[0] = a

    def step(self, a):
        r, te, tr = self.o.step(a, 0)
        o = self.o
        return dict(pos=o.pos[:, 0].copy(), ori=float(o.ori[0]), grid=o.grid[:, 0].copy(), reward=float(r[0]),
                    cmd_idx=int(o.cmd_idx[0]), cmd_age=int(o.cmd_age[0]), term=int(te[0]), trunc=int(tr[0]),
                    steps=int(o.steps[0]), collision=float(o.collision[0]))


@pytest.mark.parametrize("path", FILES)
def test_pose_rules_rewards_trajectory(path):
    g, task = load_maze_golden(path)
    st = OracleStepper(g, task, int(g["res"]))
    out = replay(g, task, int(g["res"]), st)
    pos = np.array([o["pos"] for o in out]); ori = np.array([o["ori"] for o in out])
    assert np.max(np.abs(pos - g["tr_pos"])) < 1e-9            # fp64 pose (libm sin/cos last-bit differences only)
    assert np.max(np.abs(ori - g["tr_ori"])) < 1e-9
    assert np.array_equal(np.array([o["grid"] for o in out]), g["tr_grid"])      # integer paths: exact
    assert np.array_equal(np.array([o["cmd_idx"] for o in out]), g["tr_cmd_idx"])
    assert np.array_equal(np.array([o["cmd_age"] for o in out]), g["tr_cmd_age"])
    # reference quirk: info["steps"] is read BEFORE do_action (maze_env.py:57), i.e. it lags the counter by one
    assert np.array_equal(np.array([o["steps"] for o in out]) - 1, g["tr_steps"])
    assert np.array_equal(np.array([o["term"] for o in out]), g["tr_term"])
    assert np.array_equal(np.array([o["trunc"] for o in out]), g["tr_trunc"])
    assert np.array_equal(np.array([o["reward"] for o in out], np.float32), g["tr_reward"].astype(np.float32))
    assert np.max(np.abs(np.array([o["collision"] for o in out]) - g["tr_collision"])) < 1e-9
    assert np.sum(np.diff(g["tr_cmd_idx"]) > 0) >= 3 and (g["tr_collision"] > 0).sum() > 50
    # continuous actions after the script
    for a, ref in zip(g["cont_actions"], g["cont_pose"]):
        st.step(a)
        assert np.max(np.abs(np.r_[st.o.pos[:, 0], st.o.ori[0]] - ref)) < 1e-9


@pytest.mark.parametrize("path", FILES)
def test_frames_at_reference_poses(path):
    """frames rendered at the reference's recorded poses: the bar is +-1 LSB on <= 0.5 % of the values"""
    g, task = load_maze_golden(path)
    tab = build_tables([task])
    for res, frames, steps in ((int(g["res"]), g["frames"], g["frame_steps"]), (64, g["frames64"], g["frames64_steps"])):
        n = len(steps)
        o = oracle.MazeOracle(tab, textures(), np.zeros(n, np.int32), resolution=(res, res))
        o.reset()
        o.pos[:] = g["tr_pos"][steps].T
        o.ori[:] = g["tr_ori"][steps]
        o.cmd_idx[:] = g["tr_cmd_idx"][steps]
        f, c = o.render(n_threads=4)
        frac, worst = frame_mismatch(f, frames)
        assert frac <= 0.005 and worst <= 1, (res, frac, worst)
        assert np.array_equal(c, g["tr_cmd_rgb"][steps])


@pytest.mark.parametrize("path", FILES[:1])
def test_reset_frame(path):
    g, task = load_maze_golden(path)
    tab = build_tables([task])
    o = oracle.MazeOracle(tab, textures(), [0], resolution=(int(g["res"]),) * 2)
    o.reset()
    f, _ = o.render()
    frac, worst = frame_mismatch(f[0], g["frame0"])
    assert frac <= 0.005 and worst <= 1


def test_distance_between_the_two_typings_of_the_ray_caster():
    """The golden frames follow the reference's source run as plain Python under NumPy 2 (float32 DDA).  numba, which
    the reference's users have, types the same source with a float64 DDA / wall-column geometry (oracle typing="numba",
    unpinned: numba is not installed here).  How far apart are they?  Measured on the golden trajectories at 64x64:
    0.06 % of the frame values differ on average (SURVEY.md M5 budgets 0.5 %), 2 % in the worst single frame; almost
    all by one level, 1e-5 of the values by more (a wall texel column or a wall-top row moves by one)."""
    fracs, big, worst = [], [], 0.0
    for f in FILES:
        g, task = load_maze_golden(f)
        steps = np.arange(0, len(g["tr_pos"]), 8)
        n = len(steps)
        o = oracle.MazeOracle(build_tables([task]), textures(), np.zeros(n, np.int32), resolution=(64, 64))
        o.reset()
        o.pos[:] = g["tr_pos"][steps].T; o.ori[:] = g["tr_ori"][steps]; o.cmd_idx[:] = g["tr_cmd_idx"][steps]
        a, _ = o.render(n_threads=4)
        b, _ = o.render(n_threads=4, typing="numba")
        d = np.abs(a.astype(np.int32) - b.astype(np.int32))
        fracs.append((d > 0).mean()); big.append((d > 1).mean())
        worst = max(worst, (d > 0).reshape(n, -1).mean(1).max())
    assert 0 < np.mean(fracs) < 0.002 and np.mean(big) < 1e-4 and worst < 0.05, (fracs, big, worst)


def test_numba_typing_variant_is_pinned_to_the_mechanical_rule():
    """tests/golden/raycast_numba_typing_frames.npz (oracle/gen_numba_typing.py): the reference's own maze_view source with
    numba's scalar typing applied mechanically — every literal, scalar argument, int() result and range variable a strong
    float64 / int64 — executed under NumPy on the golden poses.  The oracle's typing="numba" wall stage (float64 DDA and
    wall-column geometry) reproduces those frames bit for bit; the default typing does not (it follows the reference as it
    runs without numba)."""
    import os
    from util import GOLD
    g = np.load(os.path.join(GOLD, "raycast_numba_typing_frames.npz"))
    differs_from_default = 0
    for name in sorted(set(g["fixture"])):
        sel = g["fixture"] == name
        gg, task = load_maze_golden(os.path.join(GOLD, str(name)))
        steps = g["step"][sel]
        n = len(steps)
        o = oracle.MazeOracle(build_tables([task]), textures(), np.zeros(n, np.int32), resolution=(64, 64))
        o.reset()
        o.pos[:] = gg["tr_pos"][steps].T; o.ori[:] = gg["tr_ori"][steps]; o.cmd_idx[:] = gg["tr_cmd_idx"][steps]
        b, _ = o.render(n_threads=4, typing="numba")
        assert np.array_equal(b, g["frames64"][sel]), name
        a, _ = o.render(n_threads=4)
        differs_from_default += int((a != g["frames64"][sel]).sum())
    assert differs_from_default > 0
