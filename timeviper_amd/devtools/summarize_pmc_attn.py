"""Dev tool: profiles/r01_attention_pmc.md from the rocprofv3 --pmc passes of pmc_attn.sh
(gpurun_out/pmc_attn*/p_counter_collection.csv).  Durations: event timings of bench_ops.py without
the profiler, passed as  causal_ms vit_ms patch_ms."""
import collections
import csv
import glob
import sys

dur_ms = [float(v) for v in sys.argv[1:4]] if len(sys.argv) >= 4 else [12.03, 1.33, 0.66]
dur = dict(zip(["attention, causal d=128", "attention, ViT d=72", "patch embed"], [d * 1e-3 for d in dur_ms]))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
info = {}
for f in sorted(glob.glob("gpurun_out/pmc_attn*/p_counter_collection.csv")):
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "flash_fwd" in k:
            name = "attention, causal d=128" if "Li8ELi4E" in k else "attention, ViT d=72"
        elif "patch_embed_kernel" in k:
            name = "patch embed"
        else:
            continue
        per[(name, r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
        info[name] = (r["Workgroup_Size"], r["VGPR_Count"])
    for (name, _), cs in per.items():
        for c, v in cs.items():
            agg[name][c].append(v)
out = ["# rocprofv3 --pmc passes on `python3 timeviper_amd/devtools/bench_ops.py --ops attn,patch` (end of round 2)\n",
       "One counter set per pass (`timeviper_amd/devtools/pmc_attn.sh`, no tracing options); mean per launch, summed over "
       "the rows rocprofv3 reports per dispatch (`timeviper_amd/devtools/summarize_pmc_attn.py`). Kernels: "
       "`flash_fwd_kernel<bf16,8,4,8,3>` = causal GQA attention, L = 32 868, 40/8 heads x 128, 96-key tiles; "
       "`flash_fwd_stream_kernel<bf16,5,3,3>` = SigLIP ViT attention, 256 frames x 729 tokens x 16 heads x 72 (one "
       "resident work-group per CU streams the query blocks of its XCD's (frame, head) pairs); `patch_embed_kernel<bf16,packed,7>` = 256 frames x 384 px. Durations are the event timings of the same "
       "tool without the profiler.\n"]
for name in dur:
    m = {c: sum(v) / len(v) for c, v in agg[name].items()}
    wg, vg = info[name]
    out.append(f"\n## {name}  (work-group {wg}, {vg} VGPRs)\n\n| counter | value per launch |\n|---|---|")
    out += [f"| {c} | {m[c]:.4g} |" for c in sorted(m)]
    simd_cycles = dur[name] * 2.4e9 * 1024
    out.append(f"\nMFMA utilisation: SQ_VALU_MFMA_BUSY_CYCLES (cycles, 32 per 32x32x16 MFMA) / (1 024 SIMDs x "
               f"{dur[name] * 1e3:.2f} ms x 2.4 GHz) = **{100 * m['SQ_VALU_MFMA_BUSY_CYCLES'] / simd_cycles:.1f} %**; "
               f"{m['SQ_INSTS_MFMA']:.3g} MFMA and {m['SQ_INSTS_VALU']:.3g} VALU wave-instructions "
               f"({m['SQ_INSTS_VALU'] / m['SQ_INSTS_MFMA']:.1f} VALU per MFMA).")
    out.append(f"Wave-cycles: {100 * m['SQ_WAIT_ANY'] / m['SQ_WAVE_CYCLES']:.1f} % parked (s_waitcnt / barrier), "
               f"{100 * m['SQ_WAIT_INST_ANY'] / m['SQ_WAVE_CYCLES']:.1f} % waiting to issue.")
    out.append(f"LDS: {100 * m['SQ_LDS_BANK_CONFLICT'] / max(m['SQ_LDS_IDX_ACTIVE'], 1):.1f} % of LDS-active cycles are bank conflicts.")
    out.append(f"L2 hit rate {100 * m['TCC_HIT_sum'] / m['TCC_REQ_sum']:.1f} %; FETCH_SIZE (KiB) x 2 (gfx950 correction) = "
               f"{m['FETCH_SIZE'] * 1024 * 2 / 1e9:.2f} GB from the fabric per launch.")
print("\n".join(out))
