import json,sys
j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("oligo", j["value"], j["ms_per_step"], j["roofline"]["frac"], "ramp", j.get("ramp_steps"))
c=j["ctr_k31"]; print("ctr", c["value"], c["ms_per_step"], c["roofline"]["frac"], "ramp", c.get("ramp_steps"))
