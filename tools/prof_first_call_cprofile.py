import os, sys, time, cProfile, pstats
sys.path.insert(0, "/root/repo")
import numpy as np, torch
torch.zeros(1, device="cuda")
from alabi_amd import SurrogateModel
from alabi_amd.workloads import make_config
cfg = make_config("C3")
sm = SurrogateModel(lnlike_fn=cfg["fn"], bounds=cfg["bounds"], savedir="/tmp/alabi_fc", verbose=False, random_state=0, cache=False)
sm.init_samples(ntrain=2000)
pr = cProfile.Profile(); pr.enable()
sm.init_gp(hyperopt_method="cv")
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
