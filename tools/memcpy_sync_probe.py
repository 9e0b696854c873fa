"""Is a device-to-device hipMemcpy on the null stream synchronous for the host?  (round 5: the plan-time copy of `in` into bufferDev1.)
Times the call's return and the following device synchronisation for a 2 GiB copy (about 0.75 ms of device time)."""
import ctypes, time
import torch
hip = ctypes.CDLL("libamdhip64.so")
n = 2 << 30
a = torch.empty(n, dtype=torch.uint8, device="cuda"); b = torch.empty_like(a)
a.fill_(1); torch.cuda.synchronize()
for name, fn in (("hipMemcpy D2D", lambda: hip.hipMemcpy(ctypes.c_void_p(b.data_ptr()), ctypes.c_void_p(a.data_ptr()), ctypes.c_size_t(n), 3)),
                 ("hipMemset", lambda: hip.hipMemset(ctypes.c_void_p(b.data_ptr()), 0, ctypes.c_size_t(n)))):
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter(); rc = fn(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        print(f"{name}: rc {rc}, call returned after {1e3 * (t1 - t0):.3f} ms, device idle {1e3 * (t2 - t1):.3f} ms later")
