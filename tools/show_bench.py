import json,sys
j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(j["value"], j["ms_per_step"])
r=j["roofline"]
print({k:r[k] for k in ("kernel","achieved","frac","effective_tflops","avg_launch_ms","share_of_step")})
for k,v in r["all_kernels"].items(): print(k,v)
print(r["stages"])
if "extra" in j: print({k:v.get("ms_per_step") for k,v in j["extra"].items()})
