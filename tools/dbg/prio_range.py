import ctypes, torch
hip = ctypes.CDLL("libamdhip64.so")
lo, hi = ctypes.c_int(), ctypes.c_int()
torch.cuda.init()
print("rc", hip.hipDeviceGetStreamPriorityRange(ctypes.byref(lo), ctypes.byref(hi)), "least", lo.value, "greatest", hi.value)
for p in (-3, -2, -1, 0, 1, 2, 3):
    h = ctypes.c_void_p()
    rc = hip.hipStreamCreateWithPriority(ctypes.byref(h), 1, p)   # hipStreamNonBlocking
    q = ctypes.c_int()
    rc2 = hip.hipStreamGetPriority(h, ctypes.byref(q)) if rc == 0 else -1
    print("create prio", p, "rc", rc, "-> actual", q.value if rc == 0 else None)
s = torch.cuda.Stream(priority=-1); print("torch -1 ->", s.priority); s = torch.cuda.Stream(priority=0); print("torch 0 ->", s.priority)
try:
    s = torch.cuda.Stream(priority=1); print("torch 1 ->", s.priority)
except Exception as e:
    print("torch 1 fails", e)
