"""GPU box: does the stream -> hardware-queue assignment matter for the headline bench?  Creates K extra torch streams (kept
alive) before the bench draws its own from torch's pool, then runs `bench.py` with the remaining arguments in-process."""
import os, runpy, sys
import torch
k = int(sys.argv[1])
keep = [torch.cuda.Stream() for _ in range(k)]
sys.argv = [os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py")] + sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
