"""Dev tool: the per-rank critical path of the sequence-sharded 10 240-frame forward at N = 2 / 4 / 8 ranks, priced from
single-GPU measurements and xGMI link rates, written as profiles/rNN_sp_prediction.json — a FALSIFIABLE artefact: the
first SCALE run on a multi-GPU node can be judged against it component by component.

    python timeviper_amd/devtools/sp_prediction.py [--round 6] [--step-ms 8084] [--out profiles/r06_sp_prediction.json]

Inputs (all stated in the output): the single-GPU step and its split (profiles/rNN_bench_final_summary.md), the scan's
measured roofline fraction by sequence length (it falls with the shard: 0.406 at 163 940 tokens, 0.277 at 32 868, 0.145 at
4 196: profiles/r06_config{2,3}.json), 153 GB/s per xGMI link and direction, 30 us per collective.
"""
import argparse
import json
import math

FRAMES, TOKENS = 10240, 163940
SCAN_BYTES = 45312           # per token and layer
N_MAMBA, N_ATTN = 27, 4
STATE_BYTES = 128 * 80 * 128 * 4 + 512
LINK = 153e9                 # bytes/s per link and direction
LAT = 30e-6                  # per collective


def scan_frac(tokens):
    """measured points (tokens -> fraction of 8 TB/s), interpolated linearly in log2(tokens)"""
    pts = [(4196, 0.145), (32868, 0.277), (163940, 0.406)]        # profiles/r06_config{2,3}.json, r06_bench_plain_line.json
    if tokens <= pts[0][0]:
        return pts[0][1]
    for (a, fa), (b, fb) in zip(pts, pts[1:]):
        if tokens <= b:
            return fa + (fb - fa) * (math.log2(tokens) - math.log2(a)) / (math.log2(b) - math.log2(a))
    return pts[-1][1]


def predict(n, step_ms, vit_ms, rowlocal_ms, attn_ms, rank_ms):
    shard = TOKENS / n
    # frames: the causal-skew split gives rank 0 the most frames (split_frames: 8 ranks 1 331 of 10 240)
    frame_share = {1: 1.0, 2: 0.5 * 1.015, 4: 0.25 * 1.03, 8: 1331 / 10240}[n]
    vit = vit_ms * frame_share
    rowlocal = rowlocal_ms / n
    scan = N_MAMBA * SCAN_BYTES * shard / (scan_frac(shard) * 8e12) * 1e3
    links = min(n - 1, 7)
    gather_direct = N_MAMBA * (LAT + STATE_BYTES / LINK) * 1e3 if n > 1 else 0.0            # every peer on its own link
    gather_ring = N_MAMBA * (LAT + (n - 1) * STATE_BYTES / LINK) * 1e3 if n > 1 else 0.0    # one link pair, n - 1 hops
    # round 6: the halo rows are projected first and gathered asynchronously UNDER in_proj (distributed.py::_mamba): hidden
    # (in_proj of a shard takes >= 1.7 ms at 8 ranks against ~ 30 us of gather); the tiny product that makes them is priced
    halo_hidden = N_MAMBA * LAT * 1e3 if n > 1 else 0.0
    halo = N_MAMBA * 0.02 if n > 1 else 0.0
    correction = N_MAMBA * 0.15 if n > 1 else 0.0
    imbalance = {1: 1.0, 2: 1.25, 4: 1.40, 8: 1.48}[n]              # last rank's causal area over the mean, skewed split
    attn = attn_ms / n * imbalance
    kv_bytes = TOKENS * 2 * 1024 * 2 * (n - 1) / n                   # received per rank and attention layer
    kv_ms = (LAT + kv_bytes / (links * LINK)) * 1e3 if n > 1 else 0.0
    qproj_ms = 2.0 * TOKENS * 4480 * 5120 / 1.45e15 * 1e3 / n        # the projection the gather hides behind
    kv_exposed = N_ATTN * max(0.0, kv_ms - qproj_ms)
    ranking = rank_ms / n + (3 * (LAT + TOKENS / n * 40 * 4 * (n - 1) / (links * LINK)) * 1e3 if n > 1 else 0.0)
    total_lo = vit + rowlocal + scan + gather_direct + halo + correction + attn + kv_exposed + ranking
    total_hi = total_lo - gather_direct + gather_ring
    return {
        "ranks": n, "shard_tokens": round(shard), "scan_frac_at_shard": round(scan_frac(shard), 3),
        "per_rank_ms": {"vit_tome_projector": round(vit, 1), "llm_row_local": round(rowlocal, 1), "ssd_scan": round(scan, 1),
                        "state_all_gather_direct": round(gather_direct, 2), "state_all_gather_ring": round(gather_ring, 2),
                        "conv_halo": round(halo, 2), "conv_halo_gather_hidden_under_in_proj": round(halo_hidden, 2),
                        "carried_in_correction": round(correction, 1),
                        "causal_attention_last_rank": round(attn, 1), "kv_gather_per_layer": round(kv_ms, 2),
                        "q_proj_per_layer": round(qproj_ms, 2), "kv_gather_exposed": round(kv_exposed, 2),
                        "attn_ranking": round(ranking, 2)},
        # what a rank waits for with nothing else to do: the shard-state gather of every Mamba layer (its inputs are the
        # scan's last outputs and its result the correction's first input: nothing to run beside it), the K/V gather where it
        # outlasts q_proj, the ranking's logit exchange
        "exposed_collectives_ms": [round(gather_direct + kv_exposed + (ranking - rank_ms / n), 2),
                                   round(gather_ring + kv_exposed + (ranking - rank_ms / n), 2)],
        "step_ms": [round(total_lo, 1), round(total_hi, 1)],
        "frames_per_s": [round(FRAMES / total_hi * 1e3), round(FRAMES / total_lo * 1e3)],
        "speedup_vs_1": [round(step_ms / total_hi, 2), round(step_ms / total_lo, 2)],
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--round", type=int, default=6)
    ap.add_argument("--step-ms", type=float, default=8084.0, help="measured single-GPU step")
    ap.add_argument("--vit-frac", type=float, default=0.882, help="ViT + ToMe + projector share of the step")
    ap.add_argument("--attn-ms", type=float, default=318.0, help="4 causal attention layers, single GPU")
    ap.add_argument("--scan-ms", type=float, default=37.0, help="27 scans, single GPU")
    ap.add_argument("--rank-ms", type=float, default=3.0)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    vit = a.step_ms * a.vit_frac
    rowlocal = a.step_ms - vit - a.attn_ms - a.scan_ms - a.rank_ms
    doc = {
        "what": "predicted per-rank critical path of `bench.py --gpus N` (10 240 frames, sequence-sharded), to be judged against "
                "the first measured SCALE run; ranges = state all-gather as direct peer copies / as a single-link ring",
        "inputs": {"single_gpu_step_ms": a.step_ms, "vit_share": a.vit_frac, "llm_row_local_ms": round(rowlocal, 1),
                   "causal_attention_ms": a.attn_ms, "scan_ms": a.scan_ms, "scan_bytes_per_token_layer": SCAN_BYTES,
                   "scan_frac_by_tokens": {"4196": 0.145, "32868": 0.277, "163940": 0.406},
                   "xgmi_link_GBps": LINK / 1e9, "collective_latency_us": LAT * 1e6, "state_bytes_per_rank_layer": STATE_BYTES},
        "predictions": [predict(n, a.step_ms, vit, rowlocal, a.attn_ms, a.rank_ms) for n in (2, 4, 8)],
    }
    out = a.out or f"profiles/r{a.round:02d}_sp_prediction.json"
    with open(out, "w") as f:
        json.dump(doc, f, indent=1)
    for p in doc["predictions"]:
        print(p["ranks"], "ranks:", p["step_ms"], "ms ->", p["frames_per_s"], "frames/s, x", p["speedup_vs_1"])


if __name__ == "__main__":
    main()
