"""Merge the per-workload summaries scripts/pmc_traffic.sh leaves under gpurun_out/pmc_<tag>_<workload>/ into the one file bench.py
reads `roofline.traffic` from:   python scripts/merge_pmc_traffic.py <tag> [workload ...]  ->  profiles/<tag>_pmc_traffic.json"""
import json
import re
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
tag = sys.argv[1]
wls = sys.argv[2:] or ["c2", "c4", "c5"]


def model_name(k: str) -> str:
    k = re.sub(r"^void\s+", "", k)
    k = re.sub(r"<.*$", "", k)
    return "k_visual_cost" if k.startswith(("k_visual_cosine", "k_visual_euclid")) else k


out = {"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, --kernel-trace only) over `bench.py --workload W --steps 30`; "
               "KiB per launch as reported. On gfx950 FETCH_SIZE counts 64 B per 128-B request for wide coalesced reads "
               "(MI355X_MICROARCH.md, HBM): hbm_bytes = 2*FETCH*1024 + WRITE*1024. Collected by scripts/pmc_traffic.sh, merged by "
               "scripts/merge_pmc_traffic.py.", "workloads": {}}
for w in wls:
    f = ROOT / "gpurun_out" / f"pmc_{tag}_{w}" / "summary.json"
    if not f.exists():
        continue
    ks = json.loads(f.read_text())["kernels"]
    o = {}
    for k, v in ks.items():
        if "FETCH_SIZE" not in v or "WRITE_SIZE" not in v:
            continue
        fe, wr = v["FETCH_SIZE"]["avg"], v["WRITE_SIZE"]["avg"]
        o[model_name(k)] = {"fetch_kib": round(fe, 1), "write_kib": round(wr, 1), "hbm_bytes": int(2 * fe * 1024 + wr * 1024), "kernel": k}
    out["workloads"][w] = o
dst = ROOT / "profiles" / f"{tag}_pmc_traffic.json"
dst.write_text(json.dumps(out, indent=1))
print(dst, {w: {k: v["hbm_bytes"] for k, v in o.items() if k.startswith("k_")} for w, o in out["workloads"].items()})
