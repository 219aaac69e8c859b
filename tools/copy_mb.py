"""Developer micro-benchmark: device <-> pinned-host copies of the sizes the codec moves per frame (HIP events)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from opendcvc_amd import _lib, nn as L
from opendcvc_amd.entropy import EntropyCoder

lib = _lib.lib()
ec = EntropyCoder()
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for name, nbytes in [("z (int8)", 65280), ("y index half (uint8)", 522240), ("y symbols half (int8)", 522240), ("packed symbols (int16 x 2 halves)", 2088960)]:
    dev = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
    host = ec.pinned("mb_" + name, nbytes)
    for direction, fn in (("d2h", lambda: lib.dcvc_memcpy_d2h(ctypes.c_void_p(host.ptr), L._p(dev), nbytes, st)),
                          ("h2d", lambda: lib.dcvc_memcpy_h2d(L._p(dev), ctypes.c_void_p(host.ptr), nbytes, st))):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        print(f"{name:36s} {nbytes/1e6:6.3f} MB {direction}: {us:7.1f} us  {nbytes/us/1e3:6.1f} GB/s")
