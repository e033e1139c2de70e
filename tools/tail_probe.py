"""Does the fused launch pay for a partial last round of workgroups?  It runs one item (8 samples of a particle) per wave, 4 waves
per workgroup, 5 workgroups per CU resident: 1280 workgroups fill the chip, config 3 has 4096 = 3.2 rounds.  Iteration time per
particle over particle counts that make whole and fractional rounds (S = 128, T = 64; store-free K-loop and storing).
usage: tail_probe.py"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B
dev = torch.device("cuda", 0)
for P in (640, 960, 1024, 1120, 1280, 1600, 1920, 2048):
    row = []
    for sf in (False, True):
        pl, obs, _ = B.build_planner(torch, "panda", P, 128, 64, torch.float32, dev, store_free=sf)
        pl._engine.set_option("no_step_pipeline", 1)          # whole-range launches: the rounds are those of ONE launch
        B.time_loop(torch, pl, obs, 150, 0)
        el = min(B.time_loop(torch, pl, obs, 100, 10) for _ in range(3))
        row.append(1e3 * el / 100)
        del pl
    wgs = P * 16 / 4
    print(f"P={P:5d}  workgroups {wgs:6.0f} = {wgs / 1280:4.2f} rounds   storing {row[0]:.4f} ms/iter = {1e3 * row[0] / P:.4f} us/particle   "
          f"store-free {row[1]:.4f} ms/iter = {1e3 * row[1] / P:.4f} us/particle")
