import ctypes, torch, sys
sys.path.insert(0, "/root/repo")
from planedepth_amd import _capi as C
lib = C.load()
torch.zeros(1, device="cuda")
out = (ctypes.c_int * 4)()
for W in (640, 1280):
    lib.pd_debug_rowquad_occupancy(W, 49, out)
    print("W", W, "fwd blocks/CU", out[0], "bwd blocks/CU", out[1], "block sizes", out[2], out[3])
