"""A second process that keeps the GPU busy (devtool for soaks under time-slicing): large matmuls for N seconds."""
import sys
import time

import torch

if __name__ == "__main__":
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    x = torch.randn(8192, 8192, device="cuda")
    t0 = time.time()
    n = 0
    while time.time() - t0 < seconds:
        for _ in range(20):
            y = x @ x
        torch.cuda.synchronize()
        n += 20
    print("hog: %d matmuls in %.0f s" % (n, time.time() - t0))
