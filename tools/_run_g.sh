for i in 1 2; do
for lib in clover_amd/lib/ab/libclover_hip_prev.so clover_amd/lib/libclover_hip.so; do
CLV_LIB=$lib python tools/kernel_bench.py > /tmp/kb.json 2>&1
python -c "
import json
d=json.load(open('/tmp/kb.json'))
print('$lib'.split('/')[-1], ' '.join(f'{k.replace(\"_n2^30\",\"\").replace(\"_32768^2\",\"\")}={v[\"ms\"]:.4f}' for k,v in d.items() if isinstance(v,dict) and 'ms' in v and ('2^30' in k or 'matrix' in k) and ('stoch' in k or 'scale' in k or k.startswith('quantize') or 'matrix' in k)))
"
done; done
