"""In-kernel cycle profile of grid_aggregate_pipe (build with: hipcc ... -DGRIDMM_AGG_PROF -c aggregate_pipe.hip, relink)."""
import ctypes, sys, torch
sys.path.insert(0, ".")
import bench, argparse
from gridmm_amd import _lib
sys.argv = ["bench.py", "--steps", "5", "--warmup", "3", "--no-cpu-baseline", "--no-torch-gpu-baseline", "--no-roofline"] + sys.argv[1:]
try:
    bench.main()
except SystemExit:
    pass
torch.cuda.synchronize()
lib = _lib.load()
buf = (ctypes.c_longlong * 64)()
lib.gridmm_debug_agg_prof.argtypes = [ctypes.c_void_p]
print("rc", lib.gridmm_debug_agg_prof(buf))
print("wave 3a tabread+mask wait dma work 3a(old) mfma flush")
for w in range(8):
    print(w, [buf[w * 8 + j] for j in range(8)])
