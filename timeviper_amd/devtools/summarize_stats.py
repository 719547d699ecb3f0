"""Dev tool: turn a rocprofv3 `--kernel-trace --stats` CSV (+ the bench line) into the markdown
summary kept under profiles/.
usage: python timeviper_amd/devtools/summarize_stats.py <kernel_stats.csv> <bench_line.json> <title> > out.md"""
import csv
import json
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
line = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
title = sys.argv[3] if len(sys.argv) > 3 else "bench under rocprofv3"
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"# {title}\n")
print("Command (on the GPU box): `rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 3 --warmup 1`\n")
print(f"Bench line of that run: **{line['value']} frames/s**, {line['ms_per_step']} ms/step (profiler attached).\n")
print(f"GPU time in kernels over warm-up + {line['steps']} steps: {tot / 1e9:.2f} s.\n")
print("| kernel | calls | avg µs | max µs | % of GPU time |\n|---|---:|---:|---:|---:|")
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:16]:
    print(f"| `{r['Name'][:64]}` | {r['Calls']} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['MaxNs']) / 1e3:.1f} "
          f"| {100 * float(r['TotalDurationNs']) / tot:.2f} |")


def share(pred):
    return 100 * sum(float(r["TotalDurationNs"]) for r in rows if pred(r["Name"])) / tot


print()
for pat in ("ssd_head_asm_kernel", "ssd_head_kernel", "ssd_dt_transpose", "ssd_slice_kernel", "ssd_cb_kernel", "ssd_correct_list_kernel", "ssd_correct_kernel", "ssd_seg_chain",
            "ssd_decay_prefix", "ssd_chain_prefix", "gemm_persist_kernel", "gemm_bf16_kernel", "layernorm_rows", "conv1d_xbc", "conv1d_bc_cb", "rmsnorm_gated"):
    for r in rows:
        if pat in r["Name"]:
            print(f"- `{r['Name'][:60]}`: {r['Calls']} calls, avg {float(r['AverageNs']) / 1e3:.1f} µs, "
                  f"max {float(r['MaxNs']) / 1e3:.1f} µs")
rf = line["roofline"]
print(f"\n`bench.py` measured the scan with events on the launch stream: {rf['launches']} launches, avg "
      f"{rf['avg_launch_us']} µs (all kernels of the operator), {rf['achieved']} GB/s algorithmic = **{100 * rf['frac']:.1f} % of "
      f"{rf['peak'] / 1000:.0f} TB/s**; HBM traffic {rf['traffic']} GB/s ({rf.get('traffic_source', '')}).\n")
print("Share of GPU time: hipBLASLt GEMMs %.1f %%, own GEMM (fc1 + GELU) %.1f %%, attention kernels %.1f %%, GELU %.1f %%, LayerNorm %.1f %%, "
      "patch embed %.1f %%, ToMe %.1f %%, Mamba kernels (scan, conv, gated norm) %.1f %%." % (
          share(lambda n: n.startswith("Cijk") or "Cijk_" in n), share(lambda n: "gemm_bf16_kernel" in n or "gemm_persist_kernel" in n), share(lambda n: "flash_fwd" in n),
          share(lambda n: "gelu_kernel" in n), share(lambda n: "layernorm" in n), share(lambda n: "patch_embed" in n),
          share(lambda n: "tome_" in n),
          share(lambda n: any(k in n for k in ("ssd_", "conv1d", "rmsnorm_gated")))))
