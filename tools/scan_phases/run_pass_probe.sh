#!/bin/bash
# Per-decision cost of the host scan's passes (slim + identity records of one 4096x4096 bf16 tensor), eight-wide vs scalar.
set -e
cd "$(dirname "$0")"
python - <<'PY'
import sys; sys.path.insert(0, '../..')
import numpy as np, torch
from oracle import mtq_oracle as orc
g = torch.Generator().manual_seed(0)
x = (torch.randn(4096, 4096, generator=g) * 0.02).to(torch.bfloat16).float().numpy()
st = orc.tile_stats(x, ["bf16", "bfp8", "bfp4", "bfp2"])
slim = np.concatenate([st[:, :2]] + [st[:, 2 + 5 * f:2 + 5 * f + 3] for f in (1, 2, 3)], axis=1)
np.ascontiguousarray(slim).tofile('/tmp/mtq_slim4096.bin')
PY
g++ -O2 -o /tmp/mtq_pass_probe pass_probe.cpp -L../../quantization_analysis_amd -lmtq_hip -Wl,-rpath,$(cd ../../quantization_analysis_amd && pwd) -Wl,-rpath,/opt/rocm/lib
echo "eight-wide:"; /tmp/mtq_pass_probe
echo "scalar:"; MTQ_SCAN_SCALAR=1 /tmp/mtq_pass_probe
grep -m1 "model name" /proc/cpuinfo
g++ -O2 -o /tmp/mtq_rng_probe rng_probe.cpp -L../../quantization_analysis_amd -lmtq_hip -Wl,-rpath,$(cd ../../quantization_analysis_amd && pwd) -Wl,-rpath,/opt/rocm/lib
echo "NumPy-compatible permutation:"; /tmp/mtq_rng_probe
