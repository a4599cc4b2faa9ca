"""The reference's execution style, restated: ONE env per Python object, ONE step() per call, NumPy's global legacy
RNG (xenoverse/anymdp/anymdp_env.py: set_task :32-79, reset :81-90, single_step :92-110, step :112-132,
get_observation :145-159, MDP branch).  Test/bench infrastructure (bench.py's secondary CPU line):
it shows what the reference's per-object Python loop costs per core next to the vectorised C oracle; it is not the
parity oracle (that is oracle/xeno_oracle.c, pinned to the golden vectors) and the product never imports it."""
import numpy


class RefStyleAnyMDPEnv(object):
    def __init__(self, max_steps=5000):
        self.max_steps = max_steps
        self.task_set = False
        self.need_reset = True

    def set_task(self, task):
        for k, v in task.items():
            setattr(self, k, v)
        self.task_set = True
        self.need_reset = True

    def reset(self):
        if not self.task_set:
            raise Exception("Must call \"set_task\" first")
        self.steps = 0
        self.need_reset = False
        self._state = int(numpy.random.choice(self.s_0, replace=True, p=self.s_0_prob))
        return int(self.state_mapping[self._state]), {"steps": self.steps}

    def step(self, action):
        if self.need_reset or not self.task_set:
            raise Exception("Must \"set_task\" and \"reset\" before doing any actions")
        assert action < self.na, "Action must be less than the number of actions"
        self.steps += 1
        truncated = self.steps >= self.max_steps
        p = self.transition[self._state, action]
        ns = int(numpy.random.choice(self.ns, p=p))
        rew_gt = self.reward[self._state, action, ns]
        rew = numpy.random.normal(rew_gt, self.reward_noise[self._state, action, ns])
        terminated = ns in self.s_e
        self._state = ns
        if terminated or truncated:
            self.need_reset = True
        return (int(self.state_mapping[ns]), rew, terminated, truncated,
                {"steps": self.steps, "reward_gt": rew_gt})


def task_dict_from_tables(tab, k):
    """task k of a struct-of-arrays table set (oracle.anymdp_synth layout) -> the reference's task dict"""
    cdf = tab["cdf"][k]
    T = numpy.diff(numpy.concatenate([numpy.zeros(cdf.shape[:-1] + (1,)), cdf], axis=-1), axis=-1)
    S = cdf.shape[0]
    term = [s for s in range(S) if (int(tab["term_mask"][k][s >> 6]) >> (s & 63)) & 1]
    n0 = int(numpy.searchsorted(tab["s0_cdf"][k], 1.0)) + 1
    p0 = numpy.diff(numpy.concatenate([[0.0], tab["s0_cdf"][k][:n0]]))
    return dict(ns=S, na=cdf.shape[1], max_steps=float(tab["max_steps"][k]), state_mapping=tab["state_map"][k],
                s_0=tab["s0_ids"][k][:n0], s_0_prob=p0 / p0.sum(), s_e=term, transition=T,
                reward=tab["rs"][k][..., 0].astype(numpy.float64), reward_noise=tab["rs"][k][..., 1].astype(numpy.float64))


def time_python_loop(tab, seconds, seed=0):
    """env-steps/s of the per-object Python loop on ONE core: n envs (one task each), random actions, reset on done"""
    n = len(tab["max_steps"])
    envs = []
    for k in range(n):
        e = RefStyleAnyMDPEnv()
        e.set_task(task_dict_from_tables(tab, k))
        e.max_steps = float(tab["max_steps"][k])
        e.reset()
        envs.append(e)
    import time
    numpy.random.seed(seed)
    na = envs[0].na
    t0 = time.perf_counter()
    count = 0
    while time.perf_counter() - t0 < seconds:
        acts = numpy.random.randint(0, na, n)
        for e, a in zip(envs, acts):
            _, _, te, tr, _ = e.step(int(a))
            if te or tr:
                e.reset()
        count += n
    return count / (time.perf_counter() - t0), n
