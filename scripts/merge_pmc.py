"""Merge two pmc_traffic JSONs of scripts/gpu_pmc.sh (the --overlap off pass and the --overlap on pass: the latter adds the
HAND instantiation of the step kernel) into one profile.  usage: merge_pmc.py one_stream.json overlap.json out.json"""
import json
import sys

if __name__ == "__main__":
    a = json.load(open(sys.argv[1]))
    b = json.load(open(sys.argv[2]))
    assert a["bench_key"] == b["bench_key"], (a["bench_key"], b["bench_key"])
    for k, v in b["kernels"].items():
        a["kernels"].setdefault(k, v)
    a["note"] = ("merged: the one-stream pass (--overlap off) and the overlapped pass (--overlap on: the HAND instantiation "
                 "`..., true>` of the step kernel); scripts/gpu_pmc.sh twice, scripts/merge_pmc.py")
    json.dump(a, open(sys.argv[3], "w"), indent=1)
    for k, v in a["kernels"].items():
        if isinstance(v, dict) and "per_env_step_corrected_B" in v:
            print("%-70s %.1f B per env-step" % (k[:70], v["per_env_step_corrected_B"]))
