"""Keep-set comparison for the bf16 "attn" ranking (modeling_nano.py:1929-1942: top-k of the head-mean attention row).

In fp32 the HIP ranking and the oracle keep identical index sets.  In bf16 the two round at different points (the
reference rounds logits, probabilities and the head mean to bf16; the kernel keeps fp32 until the final score), so a
score can land one or two bf16 steps away from the oracle's, and ONLY tokens whose oracle score lies within that
rounding band of the k-th score can be kept by one and dropped by the other.  This asserts exactly that: every index
on which the two sets disagree has an oracle score inside the band around the threshold, every token strictly above
the band is kept by both, every token strictly below it by neither."""
import torch


def assert_keepsets_agree_outside_rounding_band(score_hip, score_ref, keep, ulps=2.0):
    sh, sr = score_hip.float().cpu().flatten(), score_ref.float().cpu().flatten()
    order = lambda t: torch.sort(t, descending=True, stable=True).indices[:keep]       # the defined tie-break: lower index first
    kh, kr = set(order(sh).tolist()), set(order(sr).tolist())
    tau = torch.sort(sr, descending=True).values[keep - 1].item()
    band = ulps * 2.0 ** -8 * abs(tau)              # bf16: 8 significant bits
    diff = sorted(kh ^ kr)
    outside = [i for i in diff if abs(sr[i].item() - tau) > band]
    assert not outside, f"{len(outside)} of {len(diff)} disagreeing tokens lie outside the rounding band: e.g. token " \
                        f"{outside[0]} oracle score {sr[outside[0]].item():.6g} vs threshold {tau:.6g} (band {band:.3g})"
    above = set(torch.nonzero(sr > tau + band).flatten().tolist())
    below = set(torch.nonzero(sr < tau - band).flatten().tolist())
    assert above <= kh and above <= kr, "a token strictly above the band was dropped"
    assert not (below & kh) and not (below & kr), "a token strictly below the band was kept"
    exact_ties = sum(1 for i in diff if sr[i].item() == tau)
    return {"disagree": len(diff), "exact_ties": exact_ties, "in_band": int(((sr - tau).abs() <= band).sum()), "tau": tau}
