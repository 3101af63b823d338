#!/usr/bin/env python
"""How fast can a pitched feasibility mask reach host memory?  (the host-buffer form of ksched_eval: `end_to_end.host_arrays_to_mask`)
C3's mask: 100 000 rows of 79 words, 80 words apart on the device.  Pitched 2-D copy vs pack-on-device + 1-D copy, into pageable vs page-locked host memory.
usage: python tools/d2h_probe.py [P=100000] [W=79] [pitch=80]"""
import ctypes as C, sys, time
import numpy as np
import torch
P = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
W = int(sys.argv[2]) if len(sys.argv) > 2 else 79
pitch = int(sys.argv[3]) if len(sys.argv) > 3 else 80
path = next(l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l)
hip = C.CDLL(path)
D2H, D2D = 2, 3
dev = torch.device("cuda:0")
src = torch.randint(0, 1 << 62, (P, pitch), dtype=torch.int64, device=dev)
packed = torch.empty((P, W), dtype=torch.int64, device=dev)
pageable = np.empty((P, W), dtype=np.int64); pageable[:] = 1
pin = C.c_void_p()
assert hip.hipHostMalloc(C.byref(pin), C.c_size_t(P * W * 8), C.c_uint(0)) == 0
pinned = np.ctypeslib.as_array(C.cast(pin, C.POINTER(C.c_int64)), shape=(P, W))
pinned[:] = 1
vp = lambda a: C.c_void_p(a.ctypes.data if isinstance(a, np.ndarray) else a.data_ptr())
sz = C.c_size_t
def copy2d(dst): assert hip.hipMemcpy2D(vp(dst), sz(W * 8), vp(src), sz(pitch * 8), sz(W * 8), sz(P), C.c_int(D2H)) == 0
def pack1d(dst):
    assert hip.hipMemcpy2DAsync(vp(packed), sz(W * 8), vp(src), sz(pitch * 8), sz(W * 8), sz(P), C.c_int(D2D), None) == 0
    assert hip.hipMemcpy(vp(dst), vp(packed), sz(P * W * 8), C.c_int(D2H)) == 0
def full1d(dst_flat):  # the whole pitched buffer as it is (padding included): what a caller that accepts pitched rows would get
    assert hip.hipMemcpy(vp(dst_flat), vp(src), sz(dst_flat.nbytes), C.c_int(D2H)) == 0
want = src[:, :W].cpu().numpy()
mb = P * W * 8 / 1e6
for name, fn, dst in (("pitched 2-D copy -> pageable (what ksched_eval does)", copy2d, pageable), ("pitched 2-D copy -> page-locked", copy2d, pinned),
                      ("pack on the device + 1-D copy -> pageable", pack1d, pageable), ("pack on the device + 1-D copy -> page-locked", pack1d, pinned)):
    dst[:] = 0
    ts = []
    for _ in range(7):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(dst); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    ok = np.array_equal(dst, want)
    print(f"{name:60s} median {np.median(ts) * 1e3:7.2f} ms  best {min(ts) * 1e3:7.2f} ms  = {mb / np.median(ts) / 1e3:6.1f} GB/s  correct={ok}")
