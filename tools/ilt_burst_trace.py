import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.getcwd())
import neurallaplacecontrol_amd as nlc
N, d, S = 655360, 5, 17
g = torch.Generator(device="cuda").manual_seed(1)
theta = (torch.rand(N, d, S, dtype=torch.float64, device="cuda", generator=g) * 2 - 1) * np.pi
phi = (torch.rand(N, d, S, dtype=torch.float64, device="cuda", generator=g) * 2 - 1) * (np.pi / 2)
t = torch.full((N,), 0.125, dtype=torch.float64, device="cuda")
torch.cuda.synchronize(); time.sleep(0.5)
for _ in range(300):
    nlc.ilt_reconstruct(theta, phi, t)
torch.cuda.synchronize(); time.sleep(0.3)
for _ in range(40):
    nlc.ilt_reconstruct(theta, phi, t)
    torch.cuda.synchronize(); time.sleep(0.002)
