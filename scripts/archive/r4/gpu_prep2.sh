#!/bin/bash
# binning of 20 M reads (ReadSet.from_flat_dev(order=True)): count / allocation / place, head library vs a variant, alternating
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for round in 1 2; do
for lib in head ${VARIANT:-noatom}; do
L=$R/build_tmp/lib_$lib.so; [ "$lib" = head ] && L=$R/strainscan_amd/lib/libstrainscan_hip.so
SS_LIB=$L timeout 300 python - "$lib" <<'PY' 2>&1 | grep -v amdgpu
import sys, time, ctypes
sys.path.insert(0, '.')
import numpy as np, torch, bench
from strainscan_amd import _lib
dev = torch.device("cuda", 0)
spec = bench.make_db(torch, dev, 103, seed=20231013)
reads = bench.make_reads(torch, dev, spec, 20_000_000, seed=2, hit_frac=0.05)
ts, br = [], []
for it in range(6):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); rs = _lib.ReadSet.from_flat_dev(reads.data_ptr(), reads.numel(), order=True); _lib.check(_lib.lib().ss_device_sync(), "sync"); ts.append((time.perf_counter() - t0) * 1e3)
    out = (ctypes.c_double * 3)(); _lib.lib().ss_reads_order_timing(out); br.append([round(x, 2) for x in out])
    rs.close()
print(sys.argv[1], "prepare ms", [round(t, 2) for t in ts], "count/alloc/place", br[-3:])
PY
done
done
