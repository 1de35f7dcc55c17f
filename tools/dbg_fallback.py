import sys, os; sys.path.insert(0,'/root/repo')
import numpy as np, torch
from oracle import mtq_oracle as orc
from quantization_analysis_amd import hip_backend as hb
from tests.inputs import gen
ALL=["bf16","bfp8","bfp4","bfp2"]
base = gen("heavy_bf16", 5, (32, 128)).copy()
def run(tag, x):
    x = torch.from_numpy(x).to(torch.bfloat16).float().numpy()
    with np.errstate(all="ignore"):
        want = orc.tile_stats(x, ALL)
    got = hb.tile_stats(torch.from_numpy(x).cuda().to(torch.bfloat16), 0xF).cpu().numpy()
    print(tag, "got", got[0,[7,12,17]], "want", want[0,[7,12,17]], "allsame", np.array_equal(np.nan_to_num(got,nan=123.0), np.nan_to_num(want,nan=123.0)))
x=base.copy(); x[6,9]=np.nan; run("posnan", x)
x=base.copy(); x[6,9]=-np.nan; run("negnan?", x)
x=base.copy(); x[6,9]=np.float32(np.uint32(0xFFC00000).view(np.float32)); run("negnan bits", x)
x=base.copy(); x[6,9]=-np.inf; run("neginf", x)
x=base.copy(); x[5,7]=np.inf; run("posinf", x)
x=base.copy(); x[5,7]=np.inf; x[6,9]=-np.inf; run("both inf lanes 2,3", x)
x=base.copy(); x[5,7]=np.inf; x[20,9]=-np.inf; run("both inf far lanes", x)
x=base.copy(); x[5,7]=np.inf; x[5,20]=-np.inf; run("both inf same lane g2,g3", x)
x=base.copy(); x[7,:16]=3e38; run("huge", x)
x=base.copy(); x[7,:16]=3e38; x[6,9]=-np.inf; run("huge+neginf same lane", x)
