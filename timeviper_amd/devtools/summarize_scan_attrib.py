"""Dev tool: HBM-read attribution of the head-per-wave scan kernel from the passes of pmc_scan_attrib.sh.
usage: python timeviper_amd/devtools/summarize_scan_attrib.py r04 [tokens=163940]   (writes profiles/<tag>_ssd_scan_read_attribution.json)"""
import collections
import csv
import glob
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
tag = sys.argv[1]
tokens = int(sys.argv[2]) if len(sys.argv) > 2 else 163940
NAMES = {0: "all memory operations", 32: "no x copies", 64: "no B / C copies", 8: "no C.B^T loads", 128: "no dt loads",
         4: "no y stores", 236: "no x, B / C, C.B^T, dt reads and no y stores"}


def per_launch(d, counter):
    vals = collections.defaultdict(float)
    for f in glob.glob(str(ROOT / f"gpurun_out/{d}/**/p_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "ssd_head_kernel" in r["Kernel_Name"] and r["Counter_Name"] == counter:
                vals[r["Dispatch_Id"]] += float(r["Counter_Value"])
    v = sorted(vals.values())
    return sum(v) / len(v) if v else None


H, P, G, N = 128, 80, 8, 128
alg = {"x": tokens * 2 * H * P, "dt": tokens * 2 * H, "B + C": tokens * 4 * G * N, "C.B^T (6 KiB per chunk and group)": (tokens + 63) // 64 * G * 6144}
out = {"tokens": tokens, "algorithmic_read_bytes": alg, "runs": {}}
base = None
for dbg, name in NAMES.items():
    f = per_launch(f"attrib_{dbg}_FET", "FETCH_SIZE")
    if f is None:
        continue
    rd = f * 1024 * 2                      # KiB; x2: gfx950 counts 128-B requests at 64 B (MI355X_MICROARCH.md)
    hit, req = per_launch(f"attrib_{dbg}_TCC", "TCC_HIT_sum"), per_launch(f"attrib_{dbg}_TCC", "TCC_REQ_sum")
    out["runs"][str(dbg)] = {"what": name, "hbm_read_bytes": rd, "l2_requests": req, "l2_hit_rate": (hit / req if req else None)}
    if dbg == 0:
        base = rd
if base:
    out["attributed_read_bytes"] = {v["what"]: base - v["hbm_read_bytes"] for k, v in out["runs"].items() if k != "0"}
(ROOT / "profiles" / f"{tag}_ssd_scan_read_attribution.json").write_text(json.dumps(out, indent=1) + "\n")
print(json.dumps(out, indent=1))
