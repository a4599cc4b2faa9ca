#!/usr/bin/env python3
"""Quick tour (needs an MI355X): sample tasks with the reference's sampler API, step 4,096 AnyMDP envs from Python,
collect a teacher-labelled rollout on the device, and step a mixed batch of families.

    python examples/quickstart.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xenoverse_amd.anymdp import AnyMDPTaskSampler, AnyMDPVecEnv  # noqa: E402
from xenoverse_amd.linds import LinDSVecEnv, LinearDSSampler  # noqa: E402
from xenoverse_amd.metacontrol import CartPoleVecEnv, sample_cartpole  # noqa: E402
from xenoverse_amd.mixed import MixedBatch  # noqa: E402

# 1. tasks: the reference's dict schema, 16 tasks x 256 envs
tasks = [AnyMDPTaskSampler(state_space=16, action_space=4, seed=k) for k in range(16)]
env = AnyMDPVecEnv(num_envs=4096, seed=0, autoreset_mode="same_step", copy=False)
env.set_task(tasks)
obs, info = env.reset()

# 2. the closed loop [policy -> step] the way to run it: captured ONCE in a torch.cuda.graph and replayed — one graph launch per
#    `unroll` vector steps, the same trajectory as eager calls, bit for bit.  At 65,536 envs a captured loop costs 6.5-7.3 us per
#    [policy -> step] against 10.5-11.2 us for eager calls and a 5-us step kernel.  `unroll="auto"` = 8.
loop = env.capture(lambda o: (o + 1) % 4, obs, unroll="auto")
ret = 0.0
for _ in range(25):
    obs, reward, terminated, truncated, info = loop.replay()       # the 5-tuple of the last of the 8 steps (static buffers)
    ret += reward.mean().item()
print("captured loop: %d steps replayed, mean reward of the sampled steps %.4f" % (loop.steps_replayed, ret / 25))
loop.close()

# 2b. the same loop issued call by call (what a policy that cannot be captured needs): any torch op producing int actions
ret = torch.zeros(4096, device=env.device)
for t in range(200):
    actions = (obs + t) % 4
    obs, reward, terminated, truncated, info = env.step(actions.to(torch.int32))
    ret += reward
print("mean reward per step under the toy policy: %.4f" % (ret.mean().item() / 200))

# 3. teacher-labelled data: value iteration for all 16 tasks on the device, then a fused 64-step epsilon-greedy rollout
q, greedy, sweeps = env.solve(gamma=0.99)
data = env.rollout_teacher(64, epsilon=0.1)
print("teacher rollout:", {k: tuple(v.shape) for k, v in data.items()}, "mean reward %.4f" % data["reward"].mean().item())
print("device error flags:", env.check_errors())
env.close()

# 4. several families side by side (BASELINE config 5 in miniature)
mb = MixedBatch("cuda:0", seed=1)
mb.add("anymdp", AnyMDPVecEnv, 1024)
mb.add("linds", LinDSVecEnv, 512)
mb.add("cartpole", CartPoleVecEnv, 512, frameskip=1)
mb.set_task({"anymdp": tasks, "linds": [LinearDSSampler(16, 8, 8, seed=k) for k in range(8)],
             "cartpole": [sample_cartpole(seed=k) for k in range(512)]})
mb.reset()
out = mb.step({"anymdp": torch.zeros(1024, dtype=torch.int32, device="cuda"),
               "linds": torch.zeros((512, 8), device="cuda"),
               "cartpole": torch.ones(512, dtype=torch.int32, device="cuda")})
print({k: tuple(v[0].shape) for k, v in out.items()})
out = mb.step_fused({"anymdp": torch.zeros(1024, dtype=torch.int32, device="cuda"),      # the same step as ONE kernel launch
                     "linds": torch.zeros((512, 8), device="cuda"),
                     "cartpole": torch.ones(512, dtype=torch.int32, device="cuda")})
print("fused:", {k: tuple(v[0].shape) for k, v in out.items()})
mb.close()

# 5. mazes with the rule-based teacher on the device: SmartSLAMAgent explores, then walks to the commanded landmarks
from xenoverse_amd.mazeworld import MazeTaskSampler, MazeWorldVecEnv, SmartSLAMAgent, teacher_rollout  # noqa: E402
menv = MazeWorldVecEnv(256, resolution=(64, 64), action_space_type="Discrete16", max_steps=500)
menv.set_task([MazeTaskSampler(n_range=(11, 16), seed=k) for k in range(8)])
menv.reset()
agent = SmartSLAMAgent(maze_env=menv)              # the reference's constructor keywords
data = teacher_rollout(menv, agent, T=300)
print("maze teacher: goals reached per env in 300 steps: %.2f" % ((data["reward"] > 0.1).sum().item() / 256))
menv.close()

# 6. memory for speed: one table line per AnyMDP step (identical results).  "auto" asks the engine's census first and builds
#    the bucket lines only when its AUTO rule would use them
env = AnyMDPVecEnv(num_envs=4096, seed=0, copy=False)
env.set_task(tasks)
env.set_search("auto", n_bucket=16)
print("search in effect:", env.effective_search, "| share of draws the lines cannot answer: %.1e" % env.bucket_census()["p_fallback"])
obs, _ = env.reset()
print("bucket search:", tuple(env.rollout(torch.zeros((16, 4096), dtype=torch.int32, device=env.device))["obs"].shape))

# 7. (the captured loop is section 2)

# 8. long open-loop bursts: step_many with consecutive launches overlapped on two or three streams (each wave takes its envs over from
#    the same wave of the step before through a tag in the env record) — same results, ~1.3x the steps per second
actions = torch.randint(0, 4, (32, 4096), device=env.device, dtype=torch.int32)
env.reset(seed=123)
ring_a = {k: v.clone() for k, v in env.step_many(256, actions).items()}
env.reset(seed=123)
env.set_step_many_graph("on")
env.set_step_many_overlap(True)
ring_b = env.step_many(256, actions)
torch.cuda.synchronize()
print("overlapped step_many: taken =", env.step_many_overlap_state == 1, "| equal to the one-stream result:",
      all(torch.equal(ring_a[k], ring_b[k]) for k in ring_a), "| device error flags:", env.check_errors())

# 9. sub-batches: K envs over contiguous ranges of this env's envs, each on a stream of its own (same states, same draws).
#    Stepping the env and its sub-batches in turn is safe: each side starts past every tick the other has used.
env.set_step_many_overlap(False)
subs = env.split(4)
for sub in subs:
    with torch.cuda.stream(sub.stream):
        o = sub.step(torch.zeros(sub.num_envs, dtype=torch.int32, device=env.device))
torch.cuda.synchronize()
o_all = env.step(torch.zeros(env.num_envs, dtype=torch.int32, device=env.device))      # the whole env again: fresh draws
print("sub-batches:", len(subs), "x", subs[0].num_envs, "envs; observation of sub 2:", tuple(o[0].shape), "| whole env:", tuple(o_all[0].shape))
env.close()
