cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_linds.py tests/test_gpu_capture.py tests/test_gpu_mixed.py tests/test_gpu_fullsize.py -q -m gpu -k "linds or mixed or capture or config_3 or config_5" 2>&1 | grep -E "passed|failed" | tail -3
timeout 1500 python bench.py --sweep-envs 4096,16384,65536,131072 --steps 960 --warmup 96 --sweep-out gpurun_out/r04_z_anymdp_envs_sweep.json 2>/dev/null | python -c "
import sys, json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
for r in d['rows']: print({k: (round(v,3) if isinstance(v,float) else v) for k,v in r.items() if k in ('envs','search','us_per_step','frac','frac_traffic','frac_of_line_rate','frac_of_floor','frac_note','fused_rollout_us_per_step')})"
timeout 600 python scripts/bench_families.py --families linds_mfma --steps 600 2>/dev/null | cut -c1-300
timeout 600 python - <<'PY'
# python-level step() of LinDS: one launch per step since xv_linds_step_info
import sys, os, torch
sys.path.insert(0, "scripts")
from bench_families import timed, linds_tasks
from xenoverse_amd.linds import LinDSVecEnv
for copy in (True, False):
    env = LinDSVecEnv(65536, autoreset_mode="same_step", seed=1, copy=copy)
    env.set_task(linds_tasks(1024)); obs, _ = env.reset()
    st = {"o": obs}
    def it():
        st["o"] = env.step((st["o"][:, :8] * -0.3).clamp(-1, 1))[0]
    print("linds python loop copy=%s: %.1f us per [policy -> step]" % (copy, timed(it, 400, 20)))
    env.close()
PY
