import ctypes, numpy as np, time, os, sys
lib = ctypes.CDLL("strainscan_amd/lib/libstrainscan_hip.so")
lib.ss_shuffle_split_bits.argtypes = [ctypes.c_uint64, ctypes.c_int, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_void_p]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
bits = np.zeros(n, dtype=np.uint32)
import hashlib
for it in range(8):
    t0 = time.perf_counter()
    rc = lib.ss_shuffle_split_bits(n, 20, n // 10, 0, bits.ctypes.data)
    dt = time.perf_counter() - t0
    print("rc", rc, "%.1f ms" % (dt * 1e3), hashlib.sha256(bits.tobytes()).hexdigest()[:16])
