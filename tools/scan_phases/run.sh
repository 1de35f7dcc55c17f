#!/bin/bash
# Per-phase timing of the host greedy scan on this machine's CPU (records of one 4096x4096 bf16 tensor from K1).
set -e
cd "$(dirname "$0")"
python - <<'PY'
import sys; sys.path.insert(0, '../..')
import torch, bench
from quantization_analysis_amd import hip_backend as hb
x = bench.make_batch(1, 0, torch.device('cuda', 0))
hb.tile_stats(x[0], 0xF).cpu().numpy().tofile('/tmp/mtq_stats4096.bin')
PY
g++ -O2 -o /tmp/mtq_scan_phases scan_phases.cpp -L../../quantization_analysis_amd -lmtq_hip -Wl,-rpath,$(cd ../../quantization_analysis_amd && pwd) -Wl,-rpath,/opt/rocm/lib
/tmp/mtq_scan_phases
