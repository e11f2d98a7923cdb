"""Merge the tables of a partial scripts/pmc_all.py run (OUTDIR/pmc_traffic.json, pmc_kernels.json) into profiles/pmc_traffic.json and
profiles/<round>_pmc_kernels.json:    python scripts/pmc_merge.py OUTDIR [ROUND]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, rnd = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "r06")
dst_t, dst_k = os.path.join(ROOT, "profiles", "pmc_traffic.json"), os.path.join(ROOT, "profiles", rnd + "_pmc_kernels.json")
new = json.load(open(os.path.join(src, "pmc_traffic.json")))
for v in new.values():
    v["source"] = v["source"].replace("profiles/pmc_kernels.json", "profiles/%s_pmc_kernels.json" % rnd)
old = json.load(open(dst_t)) if os.path.exists(dst_t) else {}
old.update(new)
json.dump(old, open(dst_t, "w"), indent=1)
kn = json.load(open(os.path.join(src, "pmc_kernels.json")))
ko = json.load(open(dst_k)) if os.path.exists(dst_k) else {}
ko.update(kn)
json.dump(ko, open(dst_k, "w"), indent=1)
print("merged", sorted(kn))
