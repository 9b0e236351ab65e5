"""projected_strong_scaling.cfg3 of a short bench run (A/B of shard-sized launches of the headline kernel)."""
import json, subprocess, sys
r = subprocess.run([sys.executable, "bench.py", "--no-configs", "--no-rows", "--no-variants", "--no-single", "--no-cpu-baseline", "--no-end-to-end"],
                   capture_output=True, text=True)
line = r.stdout.strip().splitlines()[-1]
d = json.loads(line)
print("value", d["value"], "frac", d["roofline"]["frac"])
det = json.load(open("bench_detail.json"))
for k, v in det["projected_strong_scaling"]["cfg3"].items():
    print(k, v)
