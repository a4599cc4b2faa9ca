"""xv_engine_probe_side_streams before / after an RCCL communicator exists (devtool): which side-stream candidates the
overlapped step_many paths would accept beside the engine's stream."""
import sys

import torch

sys.path.insert(0, ".")
from xenoverse_amd.engine import Engine  # noqa: E402

if __name__ == "__main__":
    eng = Engine("cuda:0")
    if "--rccl-first" in sys.argv:
        from xenoverse_amd.distributed import RolloutGather
        g = RolloutGather((1 << 20,), device="cuda", transport="rccl", rank=0, world=1)
        torch.cuda.synchronize()
    for rnd in range(2):
        for row in eng.probe_side_streams():
            print("round", rnd, row, flush=True)
