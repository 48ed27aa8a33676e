import sys, json
for line in sys.stdin:
    line = line.strip()
    if line.startswith('{'):
        d = json.loads(line)
        r = d["roofline"]
        print(d["config"]["name"], "fps", d["value"], "ms/step", d["ms_per_step"], "dom", r["kernel"], "GB/s", r["achieved"], "frac", r["frac"], r["kernel_ms_per_step"])
