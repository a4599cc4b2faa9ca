"""GPU parity: HIP MazeWorld (move/collision + rules + ray-caster, through the C-ABI) vs the reference's golden
trajectories and frames, and vs the CPU oracle on seeded batches.  Pose is fp64 (device sin/cos differ from
libm in the last bits: 1e-9 abs); integer paths (grid cell, command index/age, flags) are exact; frames are
within +-1 LSB on <= 0.5 % of the values."""
import numpy as np
import pytest
import torch

import oracle
from xenoverse_amd.mazeworld import MazeWorldVecEnv, build_tables, make_texture_library
from util import frame_mismatch, golden_files, load_maze_golden

pytestmark = pytest.mark.gpu
FILES = golden_files("maze_")
_TEX = None


def tex():
    global _TEX
    if _TEX is None:
        _TEX = make_texture_library(8, 4, 4, seed=0)
    return _TEX


def _np(t):
    return t.detach().cpu().numpy()


@pytest.mark.parametrize("path", FILES)
def test_golden_trajectory(path):
    g, task = load_maze_golden(path)
    res = int(g["res"])
    env = MazeWorldVecEnv(1, resolution=(res, res), max_steps=int(g["max_steps"]), textures=tex(),
                          autoreset_mode="disabled", action_space_type="Discrete16")
    env.set_task(task)
    f0, info0 = env.reset()
    frac, worst = frame_mismatch(_np(f0)[0], g["frame0"])
    assert frac <= 0.005 and worst <= 1
    T = len(g["actions"])
    fsteps = {int(s): k for k, s in enumerate(g["frame_steps"])}
    bad_frames = []
    for t in range(T):
        if not np.isnan(g["inj_pose"][t, 0]):
            env.set_state(pos=g["inj_pose"][t, :2].reshape(2, 1), ori=g["inj_pose"][t, 2:3])
        if g["inj_age"][t] >= 0:
            env.set_state(cmd_age=[int(g["inj_age"][t])])
        frames, r, term, trunc, info = env.step([int(g["actions"][t])])
        st = env.get_state()
        assert np.max(np.abs(_np(st["pos"])[:, 0] - g["tr_pos"][t])) < 1e-9
        assert abs(float(st["ori"][0]) - g["tr_ori"][t]) < 1e-9
        assert np.array_equal(_np(st["grid"])[:, 0], g["tr_grid"][t])
        assert int(st["cmd_idx"][0]) == g["tr_cmd_idx"][t] and int(st["cmd_age"][0]) == g["tr_cmd_age"][t]
        assert int(info["steps"][0]) == g["tr_steps"][t]
        assert bool(term[0]) == bool(g["tr_term"][t]) and bool(trunc[0]) == bool(g["tr_trunc"][t])
        assert np.float32(r[0].item()) == np.float32(g["tr_reward"][t])
        assert abs(float(st["collision"][0]) - g["tr_collision"][t]) < 1e-9
        assert np.array_equal(_np(info["command"])[0], g["tr_cmd_rgb"][t])
        if t in fsteps:
            frac, worst = frame_mismatch(_np(frames)[0], g["frames"][fsteps[t]])
            bad_frames.append((frac, worst))
    assert max(w for _, w in bad_frames) <= 1 and np.mean([f for f, _ in bad_frames]) <= 0.005
    # continuous actions after the script
    envc = MazeWorldVecEnv(1, resolution=(res, res), textures=tex(), autoreset_mode="disabled",
                           action_space_type="Continuous")
    envc.set_task(task)
    envc.reset()
    envc.set_state(pos=g["tr_pos"][T - 1].reshape(2, 1), ori=g["tr_ori"][T - 1:T])
    for a, ref in zip(g["cont_actions"], g["cont_pose"]):
        envc.step(a.reshape(1, 2))
        st = envc.get_state()
        assert np.max(np.abs(np.r_[_np(st["pos"])[:, 0], float(st["ori"][0])] - ref)) < 1e-9
    env.close(); envc.close()


@pytest.mark.parametrize("path", FILES)
@pytest.mark.parametrize("res", [64])
def test_golden_frames_64(path, res):
    g, task = load_maze_golden(path)
    steps = g["frames64_steps"]
    n = len(steps)
    env = MazeWorldVecEnv(n, resolution=(res, res), textures=tex(), autoreset_mode="disabled")
    env.set_task(task)
    env.reset()
    env.set_state(pos=g["tr_pos"][steps].T.copy(), ori=g["tr_ori"][steps].copy(), cmd_idx=g["tr_cmd_idx"][steps])
    f = _np(env.render_frames())
    frac, worst = frame_mismatch(f, g["frames64"])
    assert frac <= 0.005 and worst <= 1, (frac, worst)
    env.close()


@pytest.mark.parametrize("res,stage", [((32, 32), True), ((48, 40), True), ((256, 256), False)])
def test_batch_vs_oracle_frames_and_state(res, stage):
    """3 mazes x 8 envs each, random discrete actions: state vs the oracle every step, frames on some steps
    (256x256, the registered resolution, takes the build-in-global path of the ray-cast kernel)"""
    tasks = [load_maze_golden(p)[1] for p in FILES]
    tab = build_tables(tasks)
    per = 8 if res[0] <= 64 else 2
    env_task = np.repeat(np.arange(len(tasks), dtype=np.int32), per)
    n = len(env_task)
    env = MazeWorldVecEnv(n, resolution=res, textures=tex(), autoreset_mode="same_step", max_steps=25,
                          action_space_type="Discrete32")
    env.set_task(tasks, env_task_index=env_task)
    ora = oracle.MazeOracle(tab, tex(), env_task, resolution=res, max_steps=25)
    from xenoverse_amd.mazeworld import DEFAULT_ACTION_SPACE_32
    table = np.array(DEFAULT_ACTION_SPACE_32, np.float64)
    f0, _ = env.reset()
    ora.reset()
    fo, co = ora.render(n_threads=8)
    frac, worst = frame_mismatch(_np(f0), fo)
    assert frac <= 0.005 and worst <= 1
    rng = np.random.RandomState(4)
    n_steps = 30 if res[0] <= 64 else 3
    for t in range(n_steps):
        a = rng.randint(0, 32, n).astype(np.int32)
        frames, r, term, trunc, info = env.step(a)
        ro, teo, tro = ora.step(table[a], 2)
        st = env.get_state()
        assert np.max(np.abs(_np(st["pos"]) - ora.pos)) < 1e-9 and np.max(np.abs(_np(st["ori"]) - ora.ori)) < 1e-9
        assert np.array_equal(_np(st["grid"]), ora.grid) and np.array_equal(_np(st["steps"]), ora.steps)
        assert np.array_equal(_np(term).astype(np.uint8), teo) and np.array_equal(_np(trunc).astype(np.uint8), tro)
        assert np.array_equal(_np(r), ro)
        ora.pos[:] = _np(st["pos"]); ora.ori[:] = _np(st["ori"])      # no compounding of last-bit differences
        if t % 5 == 0 or tro.any():
            fo, co = ora.render(n_threads=8)
            frac, worst = frame_mismatch(_np(frames), fo)
            assert frac <= 0.005 and worst <= 1, (t, frac, worst)
            assert np.array_equal(_np(info["command"]), co)
    if res[0] <= 64:
        assert env.get_state()["steps"].max() < 25      # truncation + SAME_STEP reset happened
    env.close()


def test_float_textures_take_general_path():
    """non-integer texels disable the packed byte copy; frames still follow the oracle, and with integer texels
    the packed path (one 16-byte load per filter row) gives the same frames as the float path"""
    tasks = [load_maze_golden(p)[1] for p in FILES]
    tab = build_tables(tasks)
    env_task = np.repeat(np.arange(len(tasks), dtype=np.int32), 4)
    n = len(env_task)
    lib_int = tex()
    lib_flt = {k: (v * np.float32(0.97) + np.float32(0.3)).astype(np.float32) for k, v in lib_int.items()}
    for lib in (lib_flt, lib_int):
        env = MazeWorldVecEnv(n, resolution=(40, 40), textures=lib, autoreset_mode="same_step", max_steps=25, seed=3)
        env.set_task(tasks, env_task_index=env_task)
        ora = oracle.MazeOracle(tab, lib, env_task, resolution=(40, 40), max_steps=25)
        f0, _ = env.reset()
        st = env.get_state()
        ora.reset()
        ora.pos[:] = _np(st["pos"]); ora.ori[:] = _np(st["ori"])
        fo, _ = ora.render(n_threads=8)
        frac, worst = frame_mismatch(_np(f0), fo)
        assert frac <= 0.005 and worst <= 1, (frac, worst)
        env.close()


def test_command_bar_and_final_obs():
    g, task = load_maze_golden(FILES[0])
    env = MazeWorldVecEnv(4, resolution=(32, 32), textures=tex(), autoreset_mode="same_step", max_steps=3,
                          command_in_observation=True, with_final_obs=True)
    env.set_task(task)
    f0, _ = env.reset()
    ora = oracle.MazeOracle(build_tables([task]), tex(), np.zeros(4, np.int32), resolution=(32, 32), max_steps=3,
                            command_in_observation=True)
    ora.reset()
    assert frame_mismatch(_np(f0), ora.render()[0])[1] <= 1
    for t in range(3):
        frames, r, term, trunc, info = env.step(np.full(4, 11, np.int32))
    assert bool(trunc.all())
    # the returned frame is the reset frame; final_obs is the frame at the truncated pose
    assert frame_mismatch(_np(frames), _np(f0))[0] == 0.0
    assert frame_mismatch(_np(info["final_obs"]), _np(f0))[0] > 0.01
    env.close()


def test_misuse():
    env = MazeWorldVecEnv(2, resolution=(32, 32), textures=tex())
    with pytest.raises(Exception, match="Must call \"set_task\" before reset"):
        env.reset()
    with pytest.raises(ValueError, match="Invalid Action Space Type"):
        MazeWorldVecEnv(2, action_space_type="Discrete8")
    env.close()


@pytest.mark.parametrize("path", FILES)
def test_f32_filter_stays_within_the_frame_budget_on_reference_frames(path):
    """precision="f32" (opt-in): the reference's own 64x64 and 32x32 frames within +-1 level on <= 0.5 % of the values"""
    g, task = load_maze_golden(path)
    for key_f, key_s, res in (("frames64", "frames64_steps", 64), ("frames", "frame_steps", int(g["res"]))):
        steps = g[key_s]
        env = MazeWorldVecEnv(len(steps), resolution=(res, res), textures=tex(), autoreset_mode="disabled", precision="f32")
        env.set_task(task)
        env.reset()
        env.set_state(pos=g["tr_pos"][steps].T.copy(), ori=g["tr_ori"][steps].copy(), cmd_idx=g["tr_cmd_idx"][steps])
        f = _np(env.render_frames())
        frac, worst = frame_mismatch(f, g[key_f])
        assert frac <= 0.005 and worst <= 1, (key_f, frac, worst)
        env.close()


@pytest.mark.parametrize("res", [(64, 64), (256, 256)])
def test_f32_filter_vs_exact_and_oracle_on_a_batch(res):
    """3 mazes x 16 random poses: f32 frames vs the oracle's exact ones (budget) and vs the exact device path"""
    tasks = [load_maze_golden(p)[1] for p in FILES]
    tab = build_tables(tasks)
    per = 16 if res[0] <= 64 else 3
    env_task = np.repeat(np.arange(len(tasks), dtype=np.int32), per)
    n = len(env_task)
    rng = np.random.RandomState(6)
    frames = {}
    for prec in ("exact", "f32"):
        env = MazeWorldVecEnv(n, resolution=res, textures=tex(), autoreset_mode="disabled", precision=prec, seed=1)
        env.set_task(tasks, env_task_index=env_task)
        env.reset()
        if prec == "exact":
            a = rng.randint(0, 16, (12, n)).astype(np.int32)
        for t in range(12):
            out = env.step(a[t])
        frames[prec] = _np(out[0])
        st = env.get_state()
        env.close()
    ora = oracle.MazeOracle(tab, tex(), env_task, resolution=res)
    ora.reset()
    ora.pos[:] = _np(st["pos"]); ora.ori[:] = _np(st["ori"]); ora.cmd_idx[:] = _np(st["cmd_idx"])
    fo, _ = ora.render(n_threads=8)
    assert frame_mismatch(frames["exact"], fo) == (0.0, 0)
    frac, worst = frame_mismatch(frames["f32"], fo)
    assert frac <= 0.005 and worst <= 1, (frac, worst)


@pytest.mark.parametrize("res,per,typing", [((64, 64), 700, "numpy2"), ((256, 256), 20, "numpy2"), ((96, 48), 150, "numba")])
def test_speculated_exact_filter_paints_the_bytes_of_the_direct_one(res, per, typing):
    """precision="exact" (default) runs the 16 taps with float64 sums and re-runs a pixel in the reference's typing when its
    byte is not certain (csrc/maze.hip: mz_interpolate_spec); "exact_direct" evaluates the reference's typing for every
    pixel.  The frames must be the same bytes — here on 8.6M / 3.9M / 2.1M pixels of batches walked 10 and 25 steps (walls
    at every distance, floors, ceilings, landmark overlays)."""
    tasks = [load_maze_golden(p)[1] for p in FILES]
    env_task = np.repeat(np.arange(len(tasks), dtype=np.int32), per)
    n = len(env_task)
    a = np.random.RandomState(11).randint(0, 16, (25, n)).astype(np.int32)
    frames = {}
    for prec in ("exact", "exact_direct"):
        env = MazeWorldVecEnv(n, resolution=res, textures=tex(), autoreset_mode="same_step", precision=prec, typing=typing,
                              seed=3)
        env.set_task(tasks, env_task_index=env_task)
        env.reset()
        got = []
        for t in range(25):
            out = env.step(a[t])
            if t in (9, 24):
                got.append(_np(out[0]).copy())
        frames[prec] = got
        env.close()
    for f0, f1 in zip(frames["exact"], frames["exact_direct"]):
        assert f0.shape == f1.shape and np.array_equal(f0, f1), int((f0 != f1).sum())


@pytest.mark.parametrize("res,per,prec,typing", [((64, 64), 300, "exact", "numpy2"), ((256, 256), 12, "exact", "numpy2"),
                                                 ((64, 64), 300, "f32", "numpy2"), ((256, 256), 12, "f32", "numba"),
                                                 ((136, 200), 20, "exact", "numba"), ((40, 24), 200, "f32", "numpy2")])
def test_rows_mapping_of_the_ray_caster_paints_the_bytes_of_the_columns_mapping(res, per, prec, typing):
    """set_raycast_mapping("rows"): every wave paints 64 rows of one column at a time from what the column left in LDS;
    "columns": a lane paints its column.  Same arithmetic per pixel, so the same bytes — exact (speculated) and fp32 filter,
    frames of one and of several column batches / row chunks (W up to 200 > 128 threads... and H = 200 > one chunk)"""
    tasks = [load_maze_golden(p)[1] for p in FILES]
    env_task = np.repeat(np.arange(len(tasks), dtype=np.int32), per)
    n = len(env_task)
    a = np.random.RandomState(13).randint(0, 16, (20, n)).astype(np.int32)
    frames = {}
    for mapping in ("columns", "rows", "auto"):
        env = MazeWorldVecEnv(n, resolution=res, textures=tex(), autoreset_mode="same_step", precision=prec, typing=typing,
                              seed=3, command_in_observation=True)
        env.set_task(tasks, env_task_index=env_task)
        env.set_raycast_mapping(mapping)
        env.reset()
        got = []
        for t in range(20):
            out = env.step(a[t])
            if t in (7, 19):
                got.append(_np(out[0]).copy())
        frames[mapping] = got
        env.close()
    for k in range(2):
        assert np.array_equal(frames["columns"][k], frames["rows"][k]), int((frames["columns"][k] != frames["rows"][k]).sum())
        assert np.array_equal(frames["columns"][k], frames["auto"][k])


@pytest.mark.parametrize("res,per", [((64, 64), 200), ((128, 96), 40), ((256, 256), 8)])
def test_the_three_window_fetches_of_the_exact_filter_paint_the_same_bytes(res, per, monkeypatch):
    """XV_MAZE_FILT (a measurement switch read at every render): 0 = the pair-interleaved texture copy on the columns
    mapping (prefetching pixel loop), 3 = the row-major copy on the columns mapping, 5 = the row-major copy on the rows mapping.
    Same arithmetic per pixel, three different fetch paths and register allocations: the same frames, byte for byte."""
    tasks = [load_maze_golden(p)[1] for p in FILES]
    env_task = np.repeat(np.arange(len(tasks), dtype=np.int32), per)
    n = len(env_task)
    a = np.random.RandomState(17).randint(0, 16, (12, n)).astype(np.int32)
    frames = {}
    for filt in ("0", "3", "5"):
        monkeypatch.setenv("XV_MAZE_FILT", filt)
        env = MazeWorldVecEnv(n, resolution=res, textures=tex(), autoreset_mode="same_step", seed=5, command_in_observation=True)
        env.set_task(tasks, env_task_index=env_task)
        env.reset()
        got = []
        for t in range(12):
            out = env.step(a[t])
            if t in (5, 11):
                got.append(_np(out[0]).copy())
        frames[filt] = got
        env.close()
    monkeypatch.delenv("XV_MAZE_FILT")
    for k in range(2):
        assert np.array_equal(frames["0"][k], frames["3"][k]), int((frames["0"][k] != frames["3"][k]).sum())
        assert np.array_equal(frames["0"][k], frames["5"][k]), int((frames["0"][k] != frames["5"][k]).sum())


@pytest.mark.parametrize("mode,space", [("same_step", "Discrete16"), ("next_step", "Discrete32"), ("disabled", "Continuous"),
                                        ("same_step", "turns")])
def test_nine_lane_move_kernel_equals_the_lane_per_env_kernel(mode, space):
    """two arrangements of the same arithmetic (xv_maze_set_move_kernel): bit-identical state, rewards and flags over a
    batch whose size is not a multiple of the 7 envs a wave holds, with wall contact, goal rules and resets"""
    tasks = [load_maze_golden(p)[1] for p in FILES]
    env_task = np.repeat(np.arange(len(tasks), dtype=np.int32), 37)[:-5]
    n = len(env_task)
    rng = np.random.RandomState(2)
    T = 40
    turns_only = space == "turns"      # Discrete16 actions 1..10 only turn: with the sorted kernel no env is walked at all
    if turns_only:
        space = "Discrete16"
        acts = rng.randint(1, 11, (T, n)).astype(np.int32)
    else:
        acts = rng.uniform(-1.2, 1.2, (T, n, 2)) if space == "Continuous" else \
            rng.randint(0, 16 if space == "Discrete16" else 32, (T, n)).astype(np.int32)
    recs = []
    # nine_lanes_compact: envs that only turn away from every wall are finished by the sorting kernel, the nine-lane
    # kernel walks the listed rest (Discrete16 / 32: both kinds present; Continuous: every env walks)
    for kern in ("lane_per_env", "nine_lanes", "three_lanes", "nine_lanes_compact"):
        env = MazeWorldVecEnv(n, resolution=(32, 32), textures=tex(), autoreset_mode=mode, max_steps=17,
                              action_space_type=space, seed=2)
        env.set_task(tasks, env_task_index=env_task)
        env.set_move_kernel(kern)
        env.reset()
        rec = []
        for t in range(T):
            f, r, te, tr, info = env.step(acts[t])
            st = env.get_state()
            rec += [_np(r), _np(te), _np(tr)] + [_np(st[k]) for k in ("pos", "ori", "grid", "steps", "cmd_idx", "cmd_age",
                                                                     "need_reset", "collision")]
            if mode == "disabled" and bool((te | tr).any()):
                env.reset(options={"reset_mask": _np(te | tr).astype(np.uint8)})
        recs.append(rec)
        assert turns_only or any(float(x.max()) > 0 for x in rec[10::11])   # walls were touched (collision > 0 somewhere)
        env.close()
    for a, b, c, d in zip(*recs):
        assert np.array_equal(a, b) and np.array_equal(a, c) and np.array_equal(a, d)


def test_numba_typing_variant_follows_its_oracle():
    """typing="numba" (float64 DDA and wall-column geometry, what numba infers for the reference's source): the device
    follows the oracle's restatement of that typing as closely as the default typing follows the goldens, and differs
    from the default frames on well under 0.5 % of the values"""
    tasks = [load_maze_golden(p)[1] for p in FILES]
    tab = build_tables(tasks)
    env_task = np.repeat(np.arange(len(tasks), dtype=np.int32), 8)
    n = len(env_task)
    rng = np.random.RandomState(3)
    frames = {}
    for typing in ("numpy2", "numba"):
        env = MazeWorldVecEnv(n, resolution=(64, 48), textures=tex(), autoreset_mode="same_step", max_steps=40,
                              action_space_type="Discrete16", typing=typing)
        env.set_task(tasks, env_task_index=env_task)
        ora = oracle.MazeOracle(tab, tex(), env_task, resolution=(64, 48), max_steps=40)
        env.reset(); ora.reset()
        acts = np.random.RandomState(3).randint(0, 16, (12, n)).astype(np.int32)
        for t in range(12):
            f, r, te, tr, info = env.step(acts[t])
        st = env.get_state()
        ora.pos[:] = _np(st["pos"]); ora.ori[:] = _np(st["ori"]); ora.cmd_idx[:] = _np(st["cmd_idx"])
        fo, _ = ora.render(n_threads=8, typing="numba" if typing == "numba" else "stub")
        frac, worst = frame_mismatch(_np(f), fo)
        assert frac <= 0.005 and worst <= 1, (typing, frac, worst)
        frames[typing] = _np(f)
        env.close()
    frac, worst = frame_mismatch(frames["numpy2"], frames["numba"])
    assert 0 < frac < 0.005, (frac, worst)


def test_numba_typing_kernel_against_the_mechanical_rule_fixture():
    """the ray-cast kernel with typing="numba" on the poses of tests/golden/raycast_numba_typing_frames.npz (the reference's
    source under numba's scalar typing, applied mechanically: oracle/gen_numba_typing.py): within the frame budget of
    the default typing against its goldens (+-1 level on <= 0.5 % of the values), and closer to that fixture than the
    default typing is"""
    import os
    from util import GOLD
    g = np.load(os.path.join(GOLD, "raycast_numba_typing_frames.npz"))
    diff = {"numba": 0, "numpy2": 0}
    for name in sorted(set(g["fixture"])):
        sel = g["fixture"] == name
        gg, task = load_maze_golden(os.path.join(GOLD, str(name)))
        steps = g["step"][sel]
        n = len(steps)
        for typing in ("numba", "numpy2"):
            env = MazeWorldVecEnv(n, resolution=(64, 64), textures=tex(), autoreset_mode="disabled",
                                  action_space_type="Discrete16", typing=typing)
            env.set_task(task)
            env.reset()
            env.set_state(pos=gg["tr_pos"][steps].T.copy(), ori=gg["tr_ori"][steps].copy(), cmd_idx=gg["tr_cmd_idx"][steps].copy())
            f = _np(env.render_frames())
            frac, worst = frame_mismatch(f, g["frames64"][sel])
            if typing == "numba":
                assert frac <= 0.005 and worst <= 1, (name, frac, worst)
            diff[typing] += int((f != g["frames64"][sel]).sum())
            env.close()
    assert diff["numba"] < diff["numpy2"], diff
