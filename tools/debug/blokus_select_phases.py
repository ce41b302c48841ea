import ctypes as C, torch, sys, os
sys.path.insert(0, "/root/repo")
from colosseumrl_amd import _native
from colosseumrl_amd.batched import BlokusBatch
bb = BlokusBatch(16384)
bb.rollout(288, 1)
torch.cuda.synchronize()
buf = (C.c_uint64 * 8)()
lib = _native.lib()
lib.crl_diag_sel_stamps(buf)
tot = sum(buf[:3])
for n, v in zip(["level 1 + setup", "fit build + sync", "walk + pick"], buf[:3]):
    print("%-18s %5.1f %%  %7.0f cycles/select-ish" % (n, 100.0 * v / tot, v / 16384 / 288))
