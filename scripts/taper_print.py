"""Dev helper: one line per workload from scripts/thin_slab_scaling.py's JSON lines on stdin (us per step and the projected efficiencies with the exchange)"""
import sys, json
for l in sys.stdin:
    if not l.startswith("{"):
        continue
    d = json.loads(l)
    for k, v in d.items():
        print(k, {x: v[x] for x in ("1_alone", "2_self_exchange", "4_alone", "4_self_exchange", "8_alone", "8_self_exchange")},
              {x: v["efficiency"][x] for x in ("2_self_exchange", "4_self_exchange", "8_self_exchange")})
