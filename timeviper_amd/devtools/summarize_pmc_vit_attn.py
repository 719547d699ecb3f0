"""Dev tool: profiles/<round>_vit_attention_pmc.md from the passes of pmc_vit_attn.sh
(gpurun_out/pmc_vit{0,5}_*/p_counter_collection.csv).   python summarize_pmc_vit_attn.py r06 <ms variant 0> <ms variant 5>"""
import collections
import csv
import glob
import sys

rnd = sys.argv[1] if len(sys.argv) > 1 else "r06"
ms = [float(v) for v in sys.argv[2:4]] if len(sys.argv) >= 4 else [1.08, 1.16]
names = {0: "flash_fwd_vit_kernel (round 6, default for bf16 frames: generated tile loop, 4 waves x 64 query rows, one wave per SIMD)",
         5: "flash_fwd_stream_kernel<bf16,5,3,3,true> (compiled: 8 waves x 32 query rows, two waves per SIMD; tv_flash_attn_set_variant(5))"}
kern = {0: "flash_fwd_vit_kernel", 5: "flash_fwd_stream_kernel"}
out = [f"# rocprofv3 --pmc passes on the two ViT attention kernels ({rnd}): generated tile loop and compiled\n",
       "`python3 timeviper_amd/devtools/bench_ops.py --ops attn` (256 frames x 729 tokens x 16 heads x 72, bf16) with "
       "`TV_FA_W64=0` / `5`; one counter set per pass, no tracing (`timeviper_amd/devtools/pmc_vit_attn.sh`); mean per "
       "launch, summed over the rows rocprofv3 reports per dispatch.  Durations: event timings of the same tool without "
       "the profiler.\n"]
rows = {}
for i, v in enumerate((0, 5)):
    agg = collections.defaultdict(list)
    for f in sorted(glob.glob(f"gpurun_out/pmc_vit{v}_*/**/*counter_collection.csv", recursive=True)):
        per = collections.defaultdict(lambda: collections.defaultdict(float))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if kern[v] not in k:
                continue
            per[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
        for cs in per.values():
            for c, val in cs.items():
                agg[c].append(val)
    m = {c: sum(x) / len(x) for c, x in agg.items()}
    rows[v] = m
    out.append(f"\n## {names[v]}: {ms[i]:.2f} ms per call\n\n| counter | value per launch |\n|---|---|")
    out += [f"| {c} | {m[c]:.4g} |" for c in sorted(m)]
    if "SQ_WAVE_CYCLES" in m:
        wc = m["SQ_WAVE_CYCLES"]
        parked = m.get("SQ_WAIT_ANY", 0) / wc
        stalled = m.get("SQ_WAIT_INST_ANY", 0) / wc
        out.append(f"\nwave-cycles: {100 * parked:.1f} % parked (s_waitcnt / barrier), {100 * stalled:.1f} % issue-stalled, "
                   f"{100 * (1 - parked - stalled):.1f} % issuing; {m.get('SQ_INSTS_VALU', 0) / max(m.get('SQ_INSTS_MFMA', 1), 1):.1f} VALU per MFMA instruction"
                   + (f"; LDS: {100 * m['SQ_LDS_BANK_CONFLICT'] / max(m.get('SQ_LDS_IDX_ACTIVE', 1), 1):.1f} % of LDS-active cycles are bank conflicts." if "SQ_LDS_BANK_CONFLICT" in m else "."))
if rows[0] and rows[5]:
    out.append("\n## side by side (compiled / generated)\n\n| counter | ratio |\n|---|---|")
    for c in sorted(set(rows[0]) & set(rows[5])):
        if rows[0][c]:
            out.append(f"| {c} | {rows[5][c] / rows[0][c]:.2f} |")
open(f"profiles/{rnd}_vit_attention_pmc.md", "w").write("\n".join(out) + "\n")
print("\n".join(out))
