cat > /tmp/loop.py <<'PY'
import ctypes as C, sys, time
sys.path.insert(0,'.')
from clover_amd.lib_binding import CloverHip
hip=CloverHip(); lib=hip.lib
G=8192
A,B=hip.alloc(G*G//2),hip.alloc(G*G//2); Cc=hip.alloc(G*G*4)
sA,sB=hip.alloc((G//64)**2*4),hip.alloc((G//64)**2*4)
mode=sys.argv[1]
if mode.endswith("zero"):
    hip.check(lib.clv_memset(A.ptr,0,A.nbytes,None)); hip.check(lib.clv_memset(B.ptr,0,B.nbytes,None))
else:
    hip.check(lib.clv_fill_random_nibbles(A.ptr,A.nbytes,1,0,None)); hip.check(lib.clv_fill_random_nibbles(B.ptr,B.nbytes,2,0,None))
hip.check(lib.clv_fill_random_scales(sA.ptr,sA.nbytes//4,3,0,None)); hip.check(lib.clv_fill_random_scales(sB.ptr,sB.nbytes//4,4,0,None))
opA,opB=C.c_void_p(),C.c_void_p()
hip.check(lib.clm4_gemm_prepare(A.ptr,G,G,C.byref(opA),None)); hip.check(lib.clm4_gemm_prepare(B.ptr,G,G,C.byref(opB),None))
if mode.startswith("i32"):
    fn=lambda: hip.check(lib.clm4_gemm_i32_prepared(opA,None,G,G,opB,None,G,0,G//64,Cc.ptr,None))
else:
    fn=lambda: hip.check(lib.clm4_gemm_prepared(opA,None,sA.ptr,G,G,opB,None,sB.ptr,G,Cc.ptr,None))
t0=time.time(); n=0
while time.time()-t0 < 6:
    for _ in range(200): fn()
    hip.sync(); n+=200
el=time.time()-t0
print(mode, "ms/call", el/n*1e3)
PY
for m in gemm gemm_zero i32 i32_zero; do
  python /tmp/loop.py $m &
  PID=$!
  sleep 3.5
  rocm-smi --showpower --showclocks 2>/dev/null | grep -E "sclk|mclk|Power|fclk" | head -6
  sleep 0.7
  rocm-smi --showpower --showclocks 2>/dev/null | grep -E "sclk|Power" | head -3
  wait $PID
done
