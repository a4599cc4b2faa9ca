"""What bounds a 65,536-env AnyMDP step from below on THIS box (bench.py's roofline.floor_us comes from the committed
output, profiles/*floor_probe*.json): runs the two microbenchmarks (prebuilt in scripts/devtools/_bin, sources beside them)
  latency_floor       an empty 1,024-wave launch; + coalesced streams; + 1 / 2 / 3 dependent random 128-byte lines per env
                      read cooperatively (8 lanes x 16 B), exactly the access structure of the step kernel
  gather_granularity  random lines per second the HBM system delivers (1,048,576 lanes, one 16-byte word of a random line each)
and writes a JSON summary.   python scripts/devtools/floor_probe.py [out.json]"""
import json
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def run(name):
    r = subprocess.run([os.path.join(HERE, "_bin", name)], capture_output=True, text=True, timeout=600)
    rows = []
    for ln in r.stdout.splitlines():
        m = re.match(r"(.*?)\s+n=(\d+)\s+([\d.]+) us per launch", ln)
        if m:
            rows.append((m.group(1).strip(), int(m.group(2)), float(m.group(3))))
    return rows, r.stdout


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(HERE, "..", "..", "gpurun_out", "floor_probe.json")
    lf, lf_txt = run("latency_floor")
    gg, gg_txt = run("gather_granularity")

    def pick(rows, name, n):
        return [t for (nm, k, t) in rows if nm.startswith(name) and k == n][0]
    d = {"what": "latency / line-rate floors of a 65,536-lane step on this box (scripts/devtools/floor_probe.py)",
         "empty_launch_us": pick(lf, "k0 empty", 65536),
         "streams_only_us": pick(lf, "k1 coalesced", 65536),
         "coop_lines_us": {"1": pick(lf, "cooperative: 1 random", 65536), "2": pick(lf, "cooperative: 2 dependent random", 65536),
                           "3": pick(lf, "cooperative: 3 dependent random", 65536)},
         "per_lane_gathers_us": {"1": pick(lf, "k2 + 1", 65536), "2": pick(lf, "k3 + 2", 65536), "3": pick(lf, "k4 + 3", 65536)}}
    t0, t1 = pick(gg, "no gather", 1048576), pick(gg, "16 B", 1048576)
    d["random_lines_per_s"] = 1048576 / ((t1 - t0) * 1e-6)
    d["random_line_traffic_GBs"] = d["random_lines_per_s"] * 128 / 1e9
    # what the figures belong to: the probe's own sources, and the step-kernel source of the tree they were taken beside
    # (bench.py prints both and says whether the kernel has changed since: roofline.floor_kernel_source_current)
    import hashlib
    sys.path.insert(0, os.path.join(HERE, "..", ".."))
    h = hashlib.sha256()
    for name in ("latency_floor.hip", "gather_granularity.hip"):
        h.update(open(os.path.join(HERE, name), "rb").read())
    d["probe_source_sha16"] = h.hexdigest()[:16]
    try:
        from xenoverse_amd.build import source_hash
        d["kernel_source_sha16"] = source_hash(("anymdp.hip", "philox.h", "xv_common.h"))
    except Exception as ex:
        d["kernel_source_sha16"] = None
        d["kernel_source_note"] = repr(ex)
    d["raw"] = {"latency_floor": lf_txt.splitlines(), "gather_granularity": gg_txt.splitlines()}
    os.makedirs(os.path.dirname(os.path.abspath(out)), exist_ok=True)
    json.dump(d, open(out, "w"), indent=1)
    print(json.dumps({k: v for k, v in d.items() if k != "raw"}))


if __name__ == "__main__":
    main()
