#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r5k; mkdir -p $O
for lib in tools/_ab/libnlc_nopair.so neurallaplacecontrol_amd/libnlc_hip.so; do
NLC_LIB_PATH=$lib python - <<'PY'
import os, sys, hashlib, time, torch
sys.path.insert(0, '.')
from neurallaplacecontrol_amd import _lib
_lib.use_library(os.environ["NLC_LIB_PATH"])
import neurallaplacecontrol_amd as nlc
name = os.path.basename(os.environ["NLC_LIB_PATH"])
for S, N in ((33, 20000), (17, 30000), (9, 1000), (3, 500), (5, 700)):
    d = 5
    g = torch.Generator(device="cuda").manual_seed(S)
    th = (torch.rand(N, d, S, dtype=torch.float64, device="cuda", generator=g) * 2 - 1) * 3.0
    ph = (torch.rand(N, d, S, dtype=torch.float64, device="cuda", generator=g) * 2 - 1) * 1.2
    t = torch.rand(N, dtype=torch.float64, device="cuda", generator=g) * 2 + 0.05
    x = nlc.ilt_reconstruct(th, ph, t, "dehoog")
    print(name, "fwd", S, N, hashlib.sha256(x.cpu().numpy().tobytes()).hexdigest()[:16])
# timing of the stand-alone forward at the bench's size
N, d, S = 655360, 5, 33
g = torch.Generator(device="cuda").manual_seed(1)
th = (torch.rand(N, d, S, dtype=torch.float64, device="cuda", generator=g) * 2 - 1) * 3.0
ph = (torch.rand(N, d, S, dtype=torch.float64, device="cuda", generator=g) * 2 - 1) * 1.2
t = torch.full((N,), 0.125, dtype=torch.float64, device="cuda")
for S2 in (33, 17):
    a, b = th[..., :S2].contiguous(), ph[..., :S2].contiguous()
    for _ in range(3): nlc.ilt_reconstruct(a, b, t, "dehoog")
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): nlc.ilt_reconstruct(a, b, t, "dehoog")
    torch.cuda.synchronize()
    print(name, "standalone forward S=%d N=%d: %.3f ms" % (S2, N, (time.perf_counter() - t0) / 10 * 1e3))
PY
done | tee $O/pair_bits_time.txt
timeout -k 10 500 python -m pytest tests -x -q -m gpu -k "dehoog or cfg5 or chain" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
timeout -k 10 300 python tools/split_ab.py --rounds 3 --only cfg5 tools/_ab/libnlc_nopair.so neurallaplacecontrol_amd/libnlc_hip.so > $O/split_ab_cfg5.json 2> $O/split_ab_cfg5.err; grep -v amdgpu.ids $O/split_ab_cfg5.err | tail -6
b() { timeout -k 10 120 python bench.py --config 4 --steps 30 --no-ilt --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', round(d['value'],1), round(d['ms_per_step'],3), {k:round(v['avg_ms'],4) for k,v in d['kernels_avg_ms'].items() if 'repfunc' in k or 'dehoog' in k})"; }
{ b; b --planner-opt dehoog_chain=1; b --planner-opt dehoog_chain=2; b --planner-opt dehoog_chain=0 --planner-opt dehoog_streams=1; b --planner-opt dehoog_chain=0 --planner-opt dehoog_streams=2; b --planner-opt dehoog_chain=0 --planner-opt dehoog_streams=3; } | tee $O/cfg5_forms.txt
