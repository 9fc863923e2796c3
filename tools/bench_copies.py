#!/usr/bin/env python3
"""Large host <-> device copies (512 MiB, pageable host memory): the library's staged yakhip_upload / yakhip_download
against the runtime's own path as torch uses it.  GPU only.  Usage: python tools/bench_copies.py"""
import os
import sys, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from yak_amd import capi
ctx = capi.Context(0)
n = 512**3
vol = torch.arange(n, dtype=torch.int32, device="cuda")
host = np.zeros(n, dtype=np.int32)
for name, fn in (("yakhip_download (staged)", lambda: ctx.download(host.ctypes.data, vol.data_ptr(), n*4)),
                 ("torch .cpu() into new tensor", lambda: vol.cpu()),
                 ("torch copy_ into pageable numpy", lambda: torch.from_numpy(host).copy_(vol))):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("%-36s %.1f ms  %.1f GB/s" % (name, dt*1e3, n*4/dt/1e9))
up = np.arange(n, dtype=np.int32)
for name, fn in (("yakhip_upload (staged)", lambda: ctx.upload(vol.data_ptr(), up.ctypes.data, n*4)),
                 ("torch copy_ from pageable numpy", lambda: vol.copy_(torch.from_numpy(up)))):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("%-36s %.1f ms  %.1f GB/s" % (name, dt*1e3, n*4/dt/1e9))
