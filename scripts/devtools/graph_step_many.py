import sys, time, torch
sys.path.insert(0, '.')
from xenoverse_amd import _lib
from xenoverse_amd.anymdp import AnyMDPVecEnv, row_lines
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
n_task, S, A, P = max(1, n // 64), 64, 8, 32
env = AnyMDPVecEnv(n, seed=1)
d = env.device
tab = dict(S=S, A=A, s0_max=4, rows=torch.empty((n_task, S, A, row_lines(S), 16), dtype=torch.float64, device=d),
           state_map=torch.empty((n_task, S), dtype=torch.int32, device=d), term_mask=torch.empty((n_task, 1), dtype=torch.int64, device=d),
           s0_cdf=torch.empty((n_task, 4), dtype=torch.float64, device=d), s0_ids=torch.empty((n_task, 4), dtype=torch.int32, device=d),
           max_steps=torch.empty(n_task, dtype=torch.int32, device=d))
_lib.check(env.lib.xv_anymdp_synth_tasks(env.engine.handle, 7, 0, n_task, S, A, 4, *[_lib.ptr(tab[k]) for k in ("rows", "state_map", "term_mask", "s0_cdf", "s0_ids", "max_steps")]))
env.set_task(tab); env.reset()
acts = torch.randint(0, A, (P, n), device=d, dtype=torch.int32)
ring = env.step_many(P, acts)
for g in (1, 0, 1, 0):
    env.set_step_many_graph(g)
    env.step_many(P * 4, acts, out=ring); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        env.step_many(P * 4, acts, out=ring)
    e1.record(); torch.cuda.synchronize()
    print("graph=%d state=%d: %.2f us/step" % (g, env.lib.xv_anymdp_step_many_graph_state(env._h), e0.elapsed_time(e1) * 1e3 / (20 * P * 4)))
