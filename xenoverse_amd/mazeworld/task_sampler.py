"""MazeTaskSampler / Resampler — procedural maze tasks with the reference's signature, dict schema and random stream.

Reference: xenoverse/mazeworld/envs/task_sampler.py (gentext :14-29, gentargets :34-46, genstart :48-52,
MazeTaskManager.sample_task :92-177, resample_task :179-225) and xenoverse/utils/grid_ops.py (Rectangle :13-49,
genmaze_largeroom :49-72, genmaze_by_primwall :74-162).  The sampler is a function of NumPy's legacy stream after
`numpy.random.seed(seed)`; this module consumes a `RandomState(seed)` in the same order — first a few large rooms dug
out of the interior, then walls released Prim-style until every open region is connected (and, with loops allowed, until
the wall density target is met) — so `MazeTaskSampler(seed=k)` returns the reference's maze for that seed, given the
same texture-library sizes (tests/golden/maze_15_seed*.npz hold three reference-sampled tasks).  The reference reads
the library sizes from its JPG folder (37 walls / 29 grounds / 21 ceilings); here they are arguments.
"""
from copy import deepcopy

import numpy as np

PI = 3.1415926   # the reference's constant (mazeworld/envs/dynamics.py:7)


def _dig_rooms(n, rng, n_rooms, size_range=(2, 4), tries=5):
    """genmaze_largeroom: up to `n_rooms` rectangles of 2..4 cells a side placed without touching each other.
    -> (occupied int8[n, n], walls int8[n, n], rooms [(r0, c0, r1, c1) in interior coordinates])"""
    m = n - 2
    occ = np.zeros((m, m), np.int8)
    rooms = []
    for _ in range(n_rooms):
        for _ in range(tries):
            h = rng.randint(size_range[0], size_range[1] + 1)          # randint, randint: the rectangle's sides
            w = rng.randint(size_range[0], size_range[1] + 1)
            # window sums of the occupancy map: a placement is free where its whole window is empty
            c = np.cumsum(np.cumsum(np.pad(occ.astype(np.int64), ((1, 0), (1, 0))), 0), 1)
            win = c[h:, w:] - c[:-h, w:] - c[h:, :-w] + c[:-h, :-w]
            rows, cols = np.where(win < 0.5)
            if rows.shape[0] == 0:
                continue
            k = rng.randint(0, rows.shape[0])                          # randint: which free placement
            r0, c0 = int(rows[k]), int(cols[k])
            r1, c1 = r0 + h - 1, c0 + w - 1
            occ[max(0, r0 - 1):min(m, r1 + 2), max(0, c0 - 1):min(m, c1 + 2)] = 1    # the room and a one-cell margin
            rooms.append((r0, c0, r1, c1))
            break
    walls = np.ones((m, m), np.int8)
    for r0, c0, r1, c1 in rooms:
        walls[r0:r1 + 1, c0:c1 + 1] = 0
    occ_full = np.ones((n, n), np.int8)
    wall_full = np.ones((n, n), np.int8)
    occ_full[1:n - 1, 1:n - 1] = occ
    wall_full[1:n - 1, 1:n - 1] = walls
    return occ_full, wall_full, rooms


def genmaze(n, rng, allow_loops=True, wall_density=0.30):
    """genmaze_by_primwall: int8[n, n], 1 = wall; border all walls.  Regions: every odd-lattice cell outside the rooms'
    margins and every room; a shuffled pass over the walls releases the first wall that joins two regions (or, once
    all are joined and loops are allowed, closes a loop / a random one with probability 0.2 per wall looked at); repeat
    until one region is left and the wall share is at most `wall_density`."""
    occ, walls, rooms = _dig_rooms(n, rng, rng.randint(0, (n - 2) ** 2 // 16))        # randint: how many rooms
    for i in range(1, n, 2):
        for j in range(1, n, 2):
            if not occ[i, j]:
                walls[i, j] = 0
    pending = {}            # interior walls still standing, in row-major insertion order
    region = {}             # open cell -> region id
    members = {}            # region id -> cells
    nxt = 0
    for i in range(1, n - 1):
        for j in range(1, n - 1):
            if walls[i, j]:
                pending[i, j] = 0
            elif not occ[i, j]:
                region[i, j] = nxt
                members[nxt] = [(i, j)]
                nxt += 1
    for r0, c0, r1, c1 in rooms:
        members[nxt] = []
        for i in range(r0 + 1, r1 + 2):
            for j in range(c0 + 1, c1 + 2):
                region[i, j] = nxt
                members[nxt].append((i, j))
        nxt += 1
    budget = (n - 2) * (n - 2) * wall_density
    while len(members) > 1 or (allow_loops and np.sum(walls[1:-1, 1:-1]) > budget):
        order = list(pending.keys())
        rng.shuffle(order)                                           # shuffle: the pass order
        keep, losers = -1, {}
        i = j = 0
        for i, j in order:
            keep, losers, seen, most = -1, {}, {}, 1
            for a, b in ((i - 1, j), (i + 1, j), (i, j - 1), (i, j + 1)):
                if 0 < a < n and 0 < b < n and walls[a, b] < 1:
                    rid = region[a, b]
                    seen[rid] = seen.get(rid, 0) + 1
                    most = max(most, seen[rid])
                    if rid < keep or keep < 0:       # the smallest region id survives a merge
                        if keep >= 0:
                            losers[keep] = True
                        keep = rid
                    elif rid != keep:
                        losers[rid] = True
            if losers and most < 2:                  # joins regions without closing a loop
                break
            if losers and most > 1 and allow_loops:
                break
            if allow_loops and len(members) < 2 and rng.random_sample() < 0.2:       # random(): only once all is joined
                break
        if keep < 0:                                 # the wall the pass ended on touches no open cell: next pass
            continue
        members[keep].append((i, j))
        region[i, j] = keep
        walls[i, j] = 0
        del pending[i, j]
        for rid in losers:
            members[keep].extend(members[rid])
            for cell in members[rid]:
                region[cell] = keep
            del members[rid]
    return walls


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
                    n_wall_textures=None, n_ground_textures=None, n_ceiling_textures=None):
    """Signature of the reference sampler (task_sampler.py:92-106) plus the texture-library sizes, which bound the
    texture ids it draws.  The reference reads them from its image folder (37 walls / 29 grounds / 21 ceilings); those
    are the defaults here, so `MazeTaskSampler(seed=k)` IS the reference's task for the seed; pass
    `*textures.texture_counts(lib)` when a library of another size is used."""
    from .textures import REFERENCE_TEXTURE_COUNTS
    n_wall_textures = REFERENCE_TEXTURE_COUNTS[0] if n_wall_textures is None else n_wall_textures
    n_ground_textures = REFERENCE_TEXTURE_COUNTS[1] if n_ground_textures is None else n_ground_textures
    n_ceiling_textures = REFERENCE_TEXTURE_COUNTS[2] if n_ceiling_textures is None else n_ceiling_textures
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
