"""Dev tool: profiles/<tag>_ssd_scan_pmc.md and profiles/<tag>_ssd_scan_traffic.json from the rocprofv3 --pmc
passes of pmc_scan.sh (gpurun_out/pmc_slice*/**/p_counter_collection.csv).
usage: python timeviper_amd/devtools/summarize_pmc_scan.py r02 4 [tokens=163940]"""
import collections
import csv
import glob
import json
import re
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
from bench import scan_source_id  # noqa: E402

tag, impl = sys.argv[1], int(sys.argv[2])
tokens = int(sys.argv[3]) if len(sys.argv) > 3 else 163940
KERNELS = ["ssd_head_asm_kernel", "ssd_head_kernel<5, 4, 2>", "ssd_slice_kernel", "ssd_cb_kernel",
           "ssd_correct_list_kernel", "ssd_correct_kernel", "ssd_chain_prefix_kernel", "ssd_seg_chain_kernel", "ssd_seg_combine_kernel", "ssd_chunk_decay_kernel",
           "ssd_decay_prefix_kernel", "ssd_dt_transpose_kernel"]
agg = collections.defaultdict(lambda: collections.defaultdict(list))     # kernel -> counter -> per-dispatch sums
names = {}
for f in sorted(glob.glob(str(ROOT / "gpurun_out/pmc_slice*/**/p_counter_collection.csv"), recursive=True)):
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        k = next((n for n in KERNELS if n in r["Kernel_Name"]), None)
        if k is None:
            continue
        m_ = re.search(re.escape(k) + r"(<[^>]*>)?", r["Kernel_Name"])
        names[k] = m_.group(0) if m_ else k
        per[(k, r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
    for (k, _), cs in per.items():
        for c, v in cs.items():
            agg[k][c].append(v)
mean = {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in agg.items()}
launches = {k: max(len(v) for v in cs.values()) for k, cs in agg.items()}
# launches of a kernel per tv_ssd_scan_fwd call (the correction helpers run once per segment > 0)
per_call = {k: 1 for k in mean}
H, P, G, N = 128, 80, 8, 128
alg = tokens * (2 * H * P + 2 * H + 2 * 2 * G * N + 2 * H * P)
hbm = {}
for k, m in mean.items():
    if "FETCH_SIZE" in m:
        hbm[f"{k}_read"] = m["FETCH_SIZE"] * 1024 * 2        # KiB; x2: gfx950 counts 128-B requests at 64 B
    if "WRITE_SIZE" in m:
        hbm[f"{k}_write"] = m["WRITE_SIZE"] * 1024
hbm["total"] = sum(hbm.values())
out = [f"# rocprofv3 --pmc passes on `python3 timeviper_amd/devtools/bench_ops.py --ops scan --model-dt --dt-std 1.3 --impl {impl}` "
       f"({tokens} tokens, Nano-9B dims), {tag}\n",
       "One counter set per pass, no tracing (`timeviper_amd/devtools/pmc_scan.sh`); mean per launch, summed over the rows "
       "rocprofv3 reports per dispatch (`timeviper_amd/devtools/summarize_pmc_scan.py`).\n"]
for k in KERNELS:
    if k not in mean:
        continue
    m = mean[k]
    out.append(f"\n## {names[k]}  ({launches[k]} launches profiled)\n\n| counter | value per launch |\n|---|---|")
    out += [f"| {c} | {m[c]:.4g} |" for c in sorted(m)]
    if "SQ_WAVE_CYCLES" in m:
        out.append(f"\nwave-cycles: {100 * m['SQ_WAIT_ANY'] / m['SQ_WAVE_CYCLES']:.1f} % parked (s_waitcnt / barrier), "
                   f"{100 * m['SQ_WAIT_INST_ANY'] / m['SQ_WAVE_CYCLES']:.1f} % issue-stalled, "
                   f"{100 * m['SQ_ACTIVE_INST_ANY'] / m['SQ_WAVE_CYCLES']:.1f} % issuing; "
                   f"LDS: {100 * m['SQ_LDS_BANK_CONFLICT'] / max(m['SQ_LDS_IDX_ACTIVE'], 1):.1f} % of LDS-active cycles are bank conflicts.")
    if "TCC_REQ_sum" in m:
        out.append(f"L2 hit rate {100 * m['TCC_HIT_sum'] / m['TCC_REQ_sum']:.1f} % of {m['TCC_REQ_sum']:.3g} requests.")
out.append(f"\n## HBM traffic per tv_ssd_scan_fwd call\n\nalgorithmic bytes {alg / 1e9:.3f} GB; FETCH_SIZE (KiB) x 2 "
           f"(gfx950 correction, MI355X_MICROARCH.md) + WRITE_SIZE (KiB): {hbm['total'] / 1e9:.3f} GB = "
           f"**{hbm['total'] / alg:.3f} x algorithmic**.")
(ROOT / "profiles" / f"{tag}_ssd_scan_pmc.md").write_text("\n".join(out) + "\n")
(ROOT / "profiles" / f"{tag}_ssd_scan_traffic.json").write_text(json.dumps({
    "kernel": " + ".join(names[k] for k in KERNELS if k in names), "scan_impl": impl, "scan_source_id": scan_source_id(),
    "tokens": tokens, "algorithmic_bytes": alg, "hbm_bytes": hbm, "hbm_over_algorithmic": hbm["total"] / alg,
    "how": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, timeviper_amd/devtools/pmc_scan.sh) on "
           f"bench_ops.py --ops scan --model-dt --dt-std 1.3 --impl {impl}; KiB units; FETCH_SIZE x2 (gfx950 correction for 16 B/lane streaming "
           "reads, MI355X_MICROARCH.md); every kernel of the call summed (one launch each per call)"}, indent=1) + "\n")
print("\n".join(out))
