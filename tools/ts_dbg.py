import ctypes as C, sys
sys.path.insert(0, '.')
import numpy as np
from clover_amd.lib_binding import CloverHip
hip = CloverHip(); lib = hip.lib
n = 8192
q, s = hip.alloc(n // 2), hip.alloc(n // 16)
hip.check(lib.clv_fill_random_nibbles(q.ptr, q.nbytes, 41, 0, None))
hip.check(lib.clv_fill_random_scales(s.ptr, s.nbytes // 4, 42, 0, None))
for _ in range(3):
    hip.check(lib.clv_fill_random_nibbles(q.ptr, q.nbytes, 41, 0, None))
    hip.check(lib.clv4_threshold(q.ptr, s.ptr, n, n, 1024, None, None))
hip.sync()
out = (C.c_longlong * 32)()
print(lib.clvx_ts_dbg(out))
t = list(out)
names = {0:'start',1:'staged',2:'counted',3:'L0 hist',4:'L0 sel',5:'L1 hist',6:'L1 sel',7:'L2 hist',8:'L2 sel',9:'L3 hist',10:'L3 sel',11:'masks',12:'ranked',13:'end'}
prev = t[0]
for i in range(14):
    print(f"{names[i]:10s} +{t[i]-prev:6d}  (cum {t[i]-t[0]})"); prev = t[i]
