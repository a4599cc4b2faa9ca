"""MazeWorld task dicts -> device tables (the host half of MazeBase.set_task, maze_base.py:23-52).

Task dict schema: SURVEY.md §8(a) M1 (keys as produced by mazeworld/envs/task_sampler.py:156-170)."""
import numpy as np

LMAX = 15          # LandmarksRGB has 15 colours (ray_caster_utils.py:11-25); the sampler caps at 15 (:124-126)
NG_MAX = 64      # xv_maze_create accepts grids up to 64 x 64

DEFAULT_ACTION_SPACE_16 = [(0.0, 0.5), (0.05, 0.0), (-0.05, 0.0), (0.1, 0.0), (-0.1, 0.0), (0.2, 0.0), (-0.2, 0.0),
                           (0.3, 0.0), (-0.3, 0.0), (0.5, 0.0), (-0.5, 0.0), (0.0, 1.0), (0.05, 1.0), (-0.05, 1.0),
                           (0.10, 1.0), (-0.10, 1.0)]
DEFAULT_ACTION_SPACE_32 = [(0.0, 0.2), (0.02, 0.0), (-0.02, 0.0), (0.05, 0.0), (-0.05, 0.0), (0.1, 0.0), (-0.1, 0.0),
                           (0.2, 0.0), (-0.2, 0.0), (0.3, 0.0), (-0.3, 0.0), (0.4, 0.0), (-0.4, 0.0), (0.5, 0.0),
                           (-0.5, 0.0), (0.0, 0.5), (0.0, 1.0), (0.02, 0.5), (0.02, 1.0), (-0.02, 0.5), (-0.02, 1.0),
                           (0.05, 0.5), (0.05, 1.0), (-0.05, 0.5), (-0.05, 1.0), (0.10, 0.5), (0.10, 1.0),
                           (-0.10, 0.5), (-0.10, 1.0), (0.0, -0.2), (0.1, -0.2), (-0.1, -0.2)]
# (turn_rate, walk_speed) tables of mazeworld/envs/dynamics.py:16-46 — the values are the action-space definition


def build_tables(tasks):
    if isinstance(tasks, dict):
        tasks = [tasks]
    n_task = len(tasks)
    NG = max(int(np.shape(t["cell_walls"])[0]) for t in tasks)
    if NG > NG_MAX:
        raise ValueError("maze size %d exceeds the supported maximum %d" % (NG, NG_MAX))
    n_cmd = max(len(t["commands_sequence"]) for t in tasks)
    walls = np.ones((n_task, NG, NG), np.int8)
    texts = np.zeros((n_task, NG, NG), np.int32)
    landmarks = np.full((n_task, NG, NG), -1, np.int8)
    ints = np.zeros((n_task, 8), np.int32)
    dbl = np.zeros((n_task, 8), np.float64)
    commands = np.zeros((n_task, n_cmd), np.int32)
    lm_coord = np.zeros((n_task, LMAX, 2), np.int32)
    for i, t in enumerate(tasks):
        w = np.asarray(t["cell_walls"])
        n = w.shape[0]
        # the asserts of MazeBase.set_task (maze_base.py:50-52)
        assert t["agent_height"] < t["wall_height"] and t["agent_height"] > 0, \
            "the agent height must be > 0 and < wall height"
        assert w.shape == np.shape(t["cell_texts"]), "the dimension of walls must be equal to textures"
        assert w.shape[0] == w.shape[1], "only support square shape"
        if len(t["commands_sequence"]) != n_cmd:
            raise ValueError("all tasks of one batch must have the same commands_sequence length")
        lm = np.asarray(t["landmarks_coordinates"], np.int64).reshape(-1, 2)
        if len(lm) > LMAX:
            raise ValueError("at most %d landmarks" % LMAX)
        walls[i, :n, :n] = w
        texts[i, :n, :n] = np.asarray(t["cell_texts"])
        landmarks[i, :n, :n] = np.asarray(t["cell_landmarks"])
        ints[i, :6] = [n, int(t["start"][0]), int(t["start"][1]), int(t["ground_text"]), int(t["ceiling_text"]), len(lm)]
        dbl[i, :7] = [t["cell_size"], t["wall_height"], t["agent_height"], t["fol_angle"], t["step_reward"],
                      t["goal_reward"], t["collision_reward"]]
        dbl[i, 7] = np.tan(float(t["fol_angle"]) / 2)    # numpy.tan(vision_angle_h / 2), ray_caster_utils.py:145
        commands[i] = np.asarray(t["commands_sequence"], np.int64)
        lm_coord[i, :len(lm)] = lm
    return dict(NG=NG, n_cmd=n_cmd, walls=walls, texts=texts, landmarks=landmarks, ints=ints, dbl=dbl,
                commands=commands, lm_coord=lm_coord)
