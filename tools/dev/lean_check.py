import sys, ctypes as C
sys.path.insert(0, '/root/repo')
from flappie_amd import binding as B
eng = B.Engine(0)
L = B.lib()
L.ffhip_debug_lean_math_check.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_ulonglong)]
L.ffhip_debug_lean_math_check.restype = C.c_int
for steps in (0, 1, 2):
    tot = 0
    for ex in (0, 1, 2, 3, 4, 5, 6, 7, 23, 64, 100, 124, 125):
        n = C.c_ulonglong(0)
        assert L.ffhip_debug_lean_math_check(eng.h, ex, steps, C.byref(n)) == 0
        tot += n.value
        if n.value: print("steps", steps, "exponent", ex, "mismatches", n.value)
    print("steps", steps, "total mismatches", tot, flush=True)
