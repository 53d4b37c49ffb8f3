"""In-kernel cycle profile of grid_relevance_wide (build aggregate_rel.hip with -DGRIDMM_AGG_PROF and relink)."""
import ctypes, sys, torch
sys.path.insert(0, ".")
import bench
from gridmm_amd import _lib
sys.argv = ["bench.py", "--shape", "native", "--steps", "5", "--warmup", "3", "--no-cpu-baseline", "--no-torch-gpu-baseline", "--no-roofline"] + sys.argv[1:]
try:
    bench.main()
except SystemExit:
    pass
torch.cuda.synchronize()
lib = _lib.load()
buf = (ctypes.c_longlong * 64)()
lib.gridmm_debug_rel_prof.argtypes = [ctypes.c_void_p]
print("rc", lib.gridmm_debug_rel_prof(buf))
print("wave  work_after_C->vmwait  barrierA  Rphase  barrierB  tail  mfma_loop ntiles   (Rphase column = rounds only)")
for w in range(8):
    print(w, [buf[w * 8 + j] for j in range(7)])
