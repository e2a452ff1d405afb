import ctypes, sys, os
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import __graft_entry__ as entry
P = entry.load_package(); L = P.lib()
m, rp, ci, v = entry.laplace5(600)
A = P.Matrix(0, m, m, rp, ci, v); d = P.Descr()
def once(op):
    C = ctypes.c_void_p()
    assert L.aoclsparse_sp2m(op, d.h, A.h, P.OP_NONE, d.h, A.h, P.STAGE_FULL, ctypes.byref(C)) == 0
    L.aoclsparse_destroy(ctypes.byref(C))
once(P.OP_NONE); once(P.OP_TRANSPOSE)
torch.cuda.synchronize(); f0 = torch.cuda.mem_get_info()[0]
import resource
r0 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
for i in range(200):
    once(P.OP_NONE if i % 2 else P.OP_TRANSPOSE)
torch.cuda.synchronize(); f1 = torch.cuda.mem_get_info()[0]
r1 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
print({"device_free_delta_MB": (f0 - f1) / 1e6, "host_maxrss_delta_MB": (r1 - r0) / 1e3})
