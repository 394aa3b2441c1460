"""Watchdog harness for the fused one-launch planner body (tools only).  Runs ONE command() asynchronously; if it has not
finished after a few seconds, dumps the kernel's sync block (tickets, progress counters, census, flags) through a side
stream and exits hard, so a hand-off bug never holds the GPU box.  The progress counters and the timeline need a library
built with -DNLC_FUSED_TRACE=1 (make EXTRA_kernels_fused=-DNLC_FUSED_TRACE=1; NLC_LIB_PATH selects it).
    (the planner option fused_keep_sync = 1 set below keeps the merge kernel from zeroing the sync block after the command)
    python tools/fused_debug.py [K] [roll_cap] [<chain_first_tiles>x<partner_tiles>]"""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import neurallaplacecontrol_amd as nlc
FLAGS = 16 + 4096 + 16  # kFusedFlags (csrc/nlc_kernels.h)
if os.environ.get("NLC_LIB_PATH"):  # tools only: another build of the same library
    from neurallaplacecontrol_amd import _lib as _nlc_lib
    _nlc_lib.use_library(os.environ["NLC_LIB_PATH"])

K = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
cap = int(sys.argv[2]) if len(sys.argv) > 2 else 0
sched = sys.argv[3] if len(sys.argv) > 3 else None

T, d, nu = bench.HORIZON, 5, 1
model = bench.synthetic_state_dict(d, nu, bench.S_TERMS).to("cuda:0")
opts = {"rollout_variant": 3, "fused_keep_sync": 1}
if cap:
    opts["fused_roll_cap"] = cap
if sched is not None:
    opts["fused_chain_first_tiles"], opts["fused_partner_tiles"] = (int(x) for x in sched.split("x"))
p = nlc.MPPIDelay(nlc.NLDynamics(model, 0.05), nlc.EnvCost(bench.ENV), d, nlc.noise_sigma(nu), num_samples=K, horizon=T,
                  device="cuda:0", lambda_=1.0, u_min=torch.tensor(-3.0), u_max=torch.tensor(3.0), u_scale=3.0,
                  noise_rng="philox", seed=0, U_init=torch.zeros(T, nu, dtype=torch.float64), planner_options=opts)
state = nlc.initial_state(bench.ENV, torch.Generator().manual_seed(0))
ab = torch.zeros(4, nu, dtype=torch.float64)


def dump(tag):
    ntk = (K + 15) // 16
    words = FLAGS + (2 * T + 1) * ntk
    nd = ((words + 1) // 2 + 63) // 64 * 64
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        blk = p._ws[-nd:].clone()
        host = torch.empty_like(blk, device="cpu").pin_memory()
        host.copy_(blk, non_blocking=True)
    t0 = time.time()
    while not side.query() and time.time() - t0 < 3:
        time.sleep(0.05)
    w = host.view(torch.int32)
    print(tag, "enc_ticket", int(w[0]), "roll_ticket", int(w[1]), "timeout", int(w[2]), "| entered", int(w[4]), "roll_start",
          int(w[5]), "roll_done", int(w[6]), "enc_done", int(w[7]), "exited", int(w[8]), flush=True)
    u = lambda i: int(w[i]) & 0xffffffff
    t0 = (~u(9)) & 0xffffffff
    rel = lambda v: round(((v - t0) & 0xffffffff) / 100.0, 1)  # us (100 MHz)
    print(tag, "timeline us since first entry: rollout past first hand-off first/last", rel((~u(10)) & 0xffffffff), rel(u(11)),
          "| rollout done first/last", rel((~u(12)) & 0xffffffff), rel(u(13)), "| last encoder tile", rel(u(14)), flush=True)
    occ = w[16:16 + 2048]
    nz = occ[occ != 0]
    print(tag, "CUs seen", int((occ != 0).sum()), "max WG/CU", int(nz.max()) if nz.numel() else 0, flush=True)
    fl = w[FLAGS:FLAGS + T * ntk].view(T, ntk)
    print(tag, "flags set per horizon step:", [int(x) for x in (fl != 0).sum(1)], flush=True)
    if os.environ.get("FUSED_TIMELINE"):
        # trace build: flag words carry the tile's completion time, the block behind the owner row every chain's step times
        us = lambda x: (((x.to(torch.int64) & 0xffffffff) - t0) & 0xffffffff).double() / 100.0
        ft = us(fl)
        print(tag, "encoder tiles of horizon step t done, us (min / median / max):",
              [(int(ft[t].min()), int(ft[t].median()), int(ft[t].max())) for t in range(0, T, 3)], flush=True)
        ch = us(w[FLAGS + (T + 1) * ntk:FLAGS + (2 * T + 1) * ntk].view(T, ntk))
        print(tag, "chains past step t, us (min / median / max):",
              [(int(ch[t].min()), int(ch[t].median()), int(ch[t].max())) for t in range(0, T, 3)], flush=True)
        order = torch.sort(ft.reshape(-1)).values
        print(tag, "encoder tiles finished by 100 us marks:", [int((order <= m).sum()) for m in range(100, 1001, 100)], flush=True)


def watchdog():
    time.sleep(6)
    print("WATCHDOG: command still running after 6 s", flush=True)
    try:
        dump("hung:")
    finally:
        os._exit(3)


done = torch.cuda.Event()
threading.Thread(target=watchdog, daemon=True).start()
t0 = time.time()
a = p.command(state, ab)
done.record()
while not done.query():
    time.sleep(0.01)
print("command finished in", round(time.time() - t0, 3), "s; action", a.cpu().tolist(), flush=True)
dump("ok:")
for rep in range(3):
    a = p.command(state, ab)
    torch.cuda.synchronize()
    dump(f"rep{rep}:")
# compare with the two-launch path
q = nlc.MPPIDelay(nlc.NLDynamics(model, 0.05), nlc.EnvCost(bench.ENV), d, nlc.noise_sigma(nu), num_samples=K, horizon=T,
                  device="cuda:0", lambda_=1.0, u_min=torch.tensor(-3.0), u_max=torch.tensor(3.0), u_scale=3.0,
                  noise_rng="philox", seed=0, U_init=torch.zeros(T, nu, dtype=torch.float64), planner_options={"rollout_variant": 2})
b = q.command(state, ab)
print("two-launch action", b.cpu().tolist(), "states equal:", bool(torch.equal(p.states, q.states)),
      "cost equal:", bool(torch.equal(p.cost_total, q.cost_total)), flush=True)
os._exit(0)
