import sys, time, os, ctypes as C
import numpy as np
sys.path.insert(0, os.getcwd())
from ephemeris_explorer_amd.workloads import plummer
# minimal ctypes driver that works with both libraries (no package import: the ABI symbol lists differ)
lib = C.CDLL(sys.argv[1])
vp, i32, i64, f64 = C.c_void_p, C.c_int32, C.c_int64, C.c_double
dp = C.POINTER(C.c_double)
lib.eph_nbody_create.argtypes = [i32, dp, dp, dp, f64, f64, C.c_char_p, C.POINTER(vp)]
lib.eph_nbody_advance.argtypes = [vp, i64]
lib.eph_nbody_sync.argtypes = [vp]
lib.eph_nbody_enable_timing.argtypes = [vp, i32]
lib.eph_nbody_kernel_time.argtypes = [vp, dp, C.POINTER(C.c_uint64)]
for n in (4096, 2048):
    pos, vel, mu = plummer(n)
    h = vp()
    assert lib.eph_nbody_create(n, pos.ctypes.data_as(dp), vel.ctypes.data_as(dp), mu.ctypes.data_as(dp), 0.0, 1.0 / 1024.0, b"QuinlanTremaine12", C.byref(h)) == 0
    lib.eph_nbody_advance(h, 12)
    t = time.time()
    while time.time() - t < 1.0:
        lib.eph_nbody_advance(h, 500); lib.eph_nbody_sync(h)
    lib.eph_nbody_enable_timing(h, 1)
    for rep in range(5):
        lib.eph_nbody_advance(h, 500); lib.eph_nbody_sync(h)
    ms, l = C.c_double(), C.c_uint64()
    lib.eph_nbody_kernel_time(h, C.byref(ms), C.byref(l))
    print(os.path.basename(sys.argv[1]), n, "%.2f us/step" % (ms.value / l.value * 1e3), flush=True)
