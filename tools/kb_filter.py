import json,sys
d=json.load(open(sys.argv[1]))
for k,v in d.items():
    if isinstance(v,dict) and "stoch" in k and "2^30" in k: print(k, v.get("ms"), v.get("frac_of_8TBs"))
