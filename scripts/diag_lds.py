import sys; sys.path.insert(0, "/root/repo")
import torch
from planedepth_amd import _capi as C
lib = C.load()
c = torch.zeros(1, dtype=torch.int32, device="cuda")
C.check(lib.pd_debug_count_lds_nans(C.ptr(c), C.stream_handle()), "count"); torch.cuda.synchronize(); print("before poison:", int(c))
c.zero_()
C.check(lib.pd_debug_poison_lds(C.stream_handle()), "poison")
C.check(lib.pd_debug_count_lds_nans(C.ptr(c), C.stream_handle()), "count"); torch.cuda.synchronize(); print("after poison:", int(c), "of", 2048 * 8192)
x = torch.randn(1 << 24, device="cuda"); y = (x * 2).sum()
c.zero_()
C.check(lib.pd_debug_count_lds_nans(C.ptr(c), C.stream_handle()), "count"); torch.cuda.synchronize(); print("after torch kernels:", int(c))
