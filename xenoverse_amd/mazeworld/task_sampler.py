"""MazeTaskSampler / Resampler — procedural maze tasks with the reference's signature and dict schema.

Reference: xenoverse/mazeworld/envs/task_sampler.py (MazeTaskManager.sample_task :92-177, resample_task
:179-225) and xenoverse/utils/grid_ops.py (genmaze_by_primwall :74-162).  Same keys, dtypes and value ranges
(SURVEY.md §8(a) M1, M7).  The topology generator is a re-statement of the idea (odd-lattice rooms joined by a
randomised Prim spanning tree, then extra walls opened until the wall density target is met when loops are
allowed); it does NOT reproduce the reference's random stream, so a given seed yields a different — equally
distributed in kind — maze.  Deterministic given `seed`.
"""
from copy import deepcopy

import numpy as np

PI = 3.1415926   # the reference's constant (mazeworld/envs/dynamics.py:7)


def genmaze(n, rng, allow_loops=True, wall_density=0.30):
    """int8[n, n], 1 = wall; border all walls; rooms on the odd lattice, all connected"""
    assert n % 2 == 1 and n >= 7
    w = np.ones((n, n), np.int8)
    rooms = [(i, j) for i in range(1, n - 1, 2) for j in range(1, n - 1, 2)]
    for c in rooms:
        w[c] = 0
    # randomised Prim over the room lattice
    start = rooms[rng.randint(len(rooms))]
    seen = {start}
    frontier = []

    def push(c):
        for d in ((2, 0), (-2, 0), (0, 2), (0, -2)):
            nb = (c[0] + d[0], c[1] + d[1])
            if 0 < nb[0] < n - 1 and 0 < nb[1] < n - 1 and nb not in seen:
                frontier.append((c, nb))
    push(start)
    while frontier:
        c, nb = frontier.pop(rng.randint(len(frontier)))
        if nb in seen:
            continue
        w[(c[0] + nb[0]) // 2, (c[1] + nb[1]) // 2] = 0
        seen.add(nb)
        push(nb)
    if allow_loops:   # open further interior walls until the density target is met
        inner = (n - 2) * (n - 2)
        cand = [(i, j) for i in range(1, n - 1) for j in range(1, n - 1) if w[i, j] and ((i % 2) != (j % 2))]
        rng.shuffle(cand)
        for c in cand:
            if w[1:-1, 1:-1].sum() <= inner * wall_density:
                break
            w[c] = 0
    return w


def _sample_cmds(rng, n_landmarks, length):
    xs = rng.randint(0, n_landmarks, length)
    for i in range(1, length):   # no immediate repeats (task_sampler.py:84-90)
        if xs[i] == xs[i - 1]:
            xs[i] = (xs[i] + rng.randint(1, n_landmarks)) % n_landmarks
    return xs


def _targets(rng, cell_walls, k):
    n = cell_walls.shape[0]
    like = rng.rand(n, n) - cell_walls
    idx = np.argsort(like, axis=None)[-k:]
    landmarks = [(int(i // n), int(i % n)) for i in idx]
    cl = np.full((n, n), -1, np.int8)
    for q, c in enumerate(landmarks):
        cl[c] = q
    return landmarks, cl


def _start(rng, cell_walls, cell_landmarks):
    n = cell_walls.shape[0]
    like = rng.rand(n, n) - cell_walls - cell_landmarks
    i = int(np.argsort(like, axis=None)[-1])
    return (i // n, i % n)


def MazeTaskSampler(n_range=(9, 25), allow_loops=True, cell_size_range=(1.5, 4.5), wall_height_range=(2.0, 6.0),
                    agent_height_range=(1.6, 2.0), wall_density_range=(0.2, 0.4), landmarks_number_range=(5, 15),
                    fol_angle_range=(0.3 * PI, 0.8 * PI), commands_sequence=200, step_reward=0.0,
                    collision_reward=-0.20, goal_reward=None, seed=None, verbose=False,
                    n_wall_textures=8, n_ground_textures=4, n_ceiling_textures=4):
    """Signature of the reference sampler (task_sampler.py:92-106) plus the texture-library sizes (the reference
    reads them from its JPG folder)."""
    rng = np.random.RandomState(seed)
    cell_size = rng.uniform(*cell_size_range)
    wall_height = rng.uniform(*wall_height_range)
    agent_height = rng.uniform(*agent_height_range)
    wall_density = rng.uniform(*wall_density_range)
    landmarks_number = min(int(rng.randint(*landmarks_number_range)), 15)
    n = int(rng.randint(*n_range))
    if n % 2 == 0:
        n += 1
    assert n > 6, "Minimum required cells are 7"
    assert landmarks_number > 1, "There must be at least 1 goal, thus landmarks_number must > 1"
    cell_walls = genmaze(n, rng, allow_loops=allow_loops, wall_density=wall_density)
    cell_texts = rng.randint(0, n_wall_textures, size=cell_walls.shape)
    inner = np.zeros_like(cell_walls, bool)
    inner[1:-1, 1:-1] = True
    cell_texts[inner & (cell_walls < 1)] = 0
    ground_text = int(rng.randint(0, n_ground_textures))
    ceiling_text = int(rng.randint(0, n_ceiling_textures))
    landmarks, cell_landmarks = _targets(rng, cell_walls, landmarks_number)
    start = _start(rng, cell_walls, cell_landmarks)
    fol_angle = rng.uniform(*fol_angle_range)
    def_goal_reward = n * np.sqrt(n) / 60.0 if goal_reward is None else goal_reward
    assert def_goal_reward > 0, "goal reward must be > 0"
    return {"start": start, "cell_walls": cell_walls, "cell_texts": cell_texts, "cell_size": float(cell_size),
            "ground_text": ground_text, "ceiling_text": ceiling_text, "step_reward": step_reward,
            "goal_reward": float(def_goal_reward), "collision_reward": collision_reward,
            "wall_height": float(wall_height), "agent_height": float(agent_height), "fol_angle": float(fol_angle),
            "commands_sequence": _sample_cmds(rng, len(landmarks), commands_sequence),
            "landmarks_coordinates": landmarks, "cell_landmarks": cell_landmarks}


def Resampler(task, resample_cmd=True, resample_start=True, resample_landmarks=False,
              resample_landmarks_color=False, seed=None, verbose=False):
    """Keep the scenario, re-draw start / commands (/ landmarks): task_sampler.py:179-225."""
    rng = np.random.RandomState(seed)
    new = deepcopy(task)
    k = len(task["landmarks_coordinates"])
    if resample_landmarks:
        landmarks, cell_landmarks = _targets(rng, task["cell_walls"], k)
    elif resample_landmarks_color:
        landmarks = list(task["landmarks_coordinates"])
        rng.shuffle(landmarks)
        cell_landmarks = np.full_like(task["cell_landmarks"], -1)
        for q, c in enumerate(landmarks):
            cell_landmarks[tuple(c)] = q
    else:
        landmarks, cell_landmarks = deepcopy(task["landmarks_coordinates"]), deepcopy(task["cell_landmarks"])
    new["landmarks_coordinates"], new["cell_landmarks"] = landmarks, cell_landmarks
    if resample_start:
        new["start"] = _start(rng, task["cell_walls"], cell_landmarks)
    if resample_cmd:
        new["commands_sequence"] = _sample_cmds(rng, k, len(task["commands_sequence"]))
    return new


def sample_batch(n, seed=None, **kwargs):
    """n mazes from MazeTaskSampler (task k uses seed + k) as ONE dict of stacked arrays — the tables
    `MazeWorldVecEnv.set_task` uploads as they are (mazeworld.tables.build_tables)."""
    from .tables import build_tables
    base = np.random.SeedSequence(seed).generate_state(1)[0] if seed is None else int(seed)
    return build_tables([MazeTaskSampler(seed=base + k, **kwargs) for k in range(n)])
