"""Per-kernel LDS bank-conflict share from a `rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE` pass.
    python tools/lds_conflicts.py <pmc_dir> <out.json>
share = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE (extra cycles of conflicts / all LDS-array cycles), averaged per launch."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from profile_summarise import OURS, counters

c = counters(sys.argv[1])
out = {}
for name in sorted(c):
    if "grid=" in name or not any(k in name for k in OURS):
        continue
    bc, ia = c[name].get("SQ_LDS_BANK_CONFLICT", []), c[name].get("SQ_LDS_IDX_ACTIVE", [])
    if bc and ia and sum(ia) > 0:
        out[name] = {"launches": len(bc), "lds_bank_conflict_cycles": int(sum(bc) / len(bc)), "lds_idx_active_cycles": int(sum(ia) / len(ia)),
                     "conflict_share": round(sum(bc) / sum(ia), 4)}
json.dump({"doc": "rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE (own pass) on python3 bench.py --steps 1 --warmup 2; per launch", "kernels": out},
          open(sys.argv[2], "w"), indent=1)
for k, v in out.items():
    print(f"{k[:60]:60s} {v['conflict_share']:.3f}  ({v['lds_bank_conflict_cycles']} / {v['lds_idx_active_cycles']})")
