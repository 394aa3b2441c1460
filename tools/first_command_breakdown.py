import os, sys, time
t00 = time.perf_counter()
import torch
sys.path.insert(0, os.getcwd())
import bench
import neurallaplacecontrol_amd as nlc
from neurallaplacecontrol_amd import _lib
t_import = time.perf_counter() - t00
torch.cuda.init(); torch.zeros(1, device="cuda"); torch.cuda.synchronize()
d, nu, A, T, K = 5, 1, 3.0, 40, 16384
model = bench.synthetic_state_dict(d, nu, 17).to("cuda")
torch.cuda.synchronize()
def stamp(msg, t0):
    torch.cuda.synchronize(); print(f"{msg:40s} {(time.perf_counter()-t0)*1e3:8.2f} ms"); return time.perf_counter()
for rep in range(2):
    t0 = time.perf_counter()
    ctx = _lib.Ctx(0); t0 = stamp("Ctx()", t0)
    del ctx
    p = nlc.MPPIDelay(nlc.NLDynamics(model, 0.05), nlc.EnvCost(bench.ENV), d, nlc.noise_sigma(nu), K, T, "cpu", lambda_=1.0,
                      u_min=torch.tensor(-A), u_max=torch.tensor(A), u_scale=A, noise_rng="philox", seed=0,
                      U_init=torch.zeros(T, nu, dtype=torch.float64))
    t0 = stamp("MPPIDelay()", t0)
    st, ab = nlc.initial_state(bench.ENV, torch.Generator().manual_seed(0)), torch.zeros(4, nu, dtype=torch.float64)
    with torch.cuda.device(0):
        p.ctx.use_torch_stream()
        t1 = time.perf_counter(); key = model.upload(p.ctx); t0 = stamp("  model.upload (nlc_set_model)", t1)
        p._model_key = key
        t1 = time.perf_counter(); p._configure(4); t0 = stamp("  _configure (nlc_mppi_configure + buffers)", t1)
    t1 = time.perf_counter(); p.command(st, ab); t0 = stamp("first command()", t1)
    t1 = time.perf_counter(); p.command(st, ab); t0 = stamp("second command()", t1)
    del p
print("import", t_import)
