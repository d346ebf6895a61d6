"""K assembly + factorisation under rocprofv3 --kernel-trace --stats: time of assemble_lower_kernel by size.
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/asm -- python3 tools/prof_assemble.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from alabi_amd import HipGP
for N, d in ((2000, 10), (5000, 10), (10000, 20)):
    X = np.random.RandomState(N).uniform(-3, 3, (N, d))
    gp = HipGP(d, 0.0, -12.0, 0.0, np.log(np.full(d, 30.0 if d == 10 else 60.0)))
    for _ in range(3):
        gp.compute(X)
    torch.cuda.synchronize()
    print(N, d, gp.solver.log_determinant)
