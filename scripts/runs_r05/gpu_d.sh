#!/bin/bash
# round 5, call D: kernel timeline of 20-step bursts, overlapped vs not (where do the ~30 us of a burst go?)
set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r05_d
mkdir -p $O
unset XV_PIPE_NOFORK
rm -rf $O/trace
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o burst -- python3 scripts/devtools/probe_chains.py --ks 1 --overlap --steps 20 --period 20 --repeats 3 --short 0 --tag burst > $O/burst.jsonl 2> $O/burst.err
echo "trace rc=$?"; cut -c1-300 $O/burst.jsonl
F=$(ls $O/trace/*kernel_trace.csv $O/trace/*/*kernel_trace.csv 2>/dev/null | head -1)
python3 - "$F" $O/burst_timeline.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
keep = [r for r in rows if any(k in r["Kernel_Name"] for k in ("step_kernel", "tick", "fill_u32"))]
out = open(sys.argv[2], "w")
# three windows: the last 70 kernels (non-overlapped second pass), and around the overlapped pass
def dump(rs, title):
    out.write("== %s\n" % title)
    t0 = int(rs[0]["Start_Timestamp"])
    for r in rs:
        nm = r["Kernel_Name"]
        short = "HAND" if "true, 1, true" in nm or "1, true>" in nm else ("step" if "step_kernel" in nm else nm[:24])
        out.write("q%s s%s %9.2f %9.2f  %6.2f  %s\n" % (r["Queue_Id"], r["Stream_Id"], (int(r["Start_Timestamp"]) - t0) / 1e3,
                  (int(r["End_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, short))
hand = [i for i, r in enumerate(keep) if "true>(AnyMDPArgs" in r["Kernel_Name"] and r["Kernel_Name"].count("true") >= 2]
print("kernels kept", len(keep), "hand kernels", len(hand))
if hand:
    dump(keep[hand[-1] - 65: hand[-1] + 3], "overlapped: the last three bursts")
dump(keep[-66:], "not overlapped: the last three bursts")
out.close()
PY
head -80 $O/burst_timeline.txt
rm -rf $O/trace
