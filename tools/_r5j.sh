#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r5j; mkdir -p $O
timeout -k 10 120 python tools/_dhb_repro.py 16384 > $O/n16384.txt 2>&1; rc=$?; echo "16384 rc=$rc"; grep -v amdgpu.ids $O/n16384.txt | tail -2 | cut -c1-200
[ $rc -ne 0 ] && exit 1
timeout -k 10 300 python -m pytest tests/test_gpu_ilt.py tests/test_gpu_model.py -x -q -m gpu -k "dehoog or backward or trains" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
for rep in 1 2; do
  for lib in tools/_ab/libnlc_dhb_old.so neurallaplacecontrol_amd/libnlc_hip.so; do
    NLC_LIB_PATH=$lib timeout -k 10 200 python tools/dehoog_bwd_bench.py 16384 655360 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', [(r['N'], round(r['hip_forward_backward_ms'],3), round(r['hip_forward_only_ms'],3)) for r in d['results']])" || exit 1
  done
done | tee $O/dhb_ab.txt
for lib in tools/_ab/libnlc_dhb_old.so neurallaplacecontrol_amd/libnlc_hip.so; do
NLC_LIB_PATH=$lib python - <<'PY'
import os, sys, hashlib, torch
sys.path.insert(0, '.')
from neurallaplacecontrol_amd import _lib
_lib.use_library(os.environ["NLC_LIB_PATH"])
import neurallaplacecontrol_amd as nlc
for S, N in ((33, 20000), (17, 30000), (9, 1000)):
    d = 5
    g = torch.Generator(device="cuda").manual_seed(S)
    th = ((torch.rand(N, d, S, dtype=torch.float64, device="cuda", generator=g) * 2 - 1) * 3.0).requires_grad_()
    ph = ((torch.rand(N, d, S, dtype=torch.float64, device="cuda", generator=g) * 2 - 1) * 1.2).requires_grad_()
    t = torch.full((N,), 0.125, dtype=torch.float64, device="cuda")
    gx = torch.randn(N, d, dtype=torch.float64, device="cuda", generator=g)
    ga = torch.autograd.grad(nlc.ilt_reconstruct(th, ph, t, "dehoog"), (th, ph), gx)
    print(os.path.basename(os.environ["NLC_LIB_PATH"]), S, N, hashlib.sha256(ga[0].cpu().numpy().tobytes() + ga[1].cpu().numpy().tobytes()).hexdigest()[:16])
PY
done | tee $O/dhb_bits.txt
