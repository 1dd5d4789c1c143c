"""Print the headline fields of a bench.py JSON line: python tools/show_bench.py file.json"""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"{d['value']:.2f} {d['unit']}  {d['ms_per_step']:.2f} ms/step  n_gpus {d['n_gpus']}  jac {d['jacobian_ms']:.4f} ms "
      f"({d['jacobian_mnnz_per_s'] / 1e3:.0f} Gnnz/s, {100 * d['roofline_jacobian']['frac']:.1f} % HBM)")
r = d["roofline"]
if r.get("achieved") is not None:
    print(f"roofline: {r['kernel']} {r['achieved']:.1f} {r['unit']} = {100 * r['frac']:.1f} % of {r['peak']}; traffic {r.get('traffic')}")
else:
    print(f"roofline: {r['kernel']} ({r['bound']}), avg launch {r.get('avg_launch_ms')} ms")
if d.get("cpu_baseline"):
    print("cpu_baseline:", d["cpu_baseline"]["value"], d["cpu_baseline"]["unit"], "cores", d["cpu_baseline"]["cores"])
print("kernel_ms:", d.get("kernel_ms"))
