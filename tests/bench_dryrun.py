"""TEST INFRASTRUCTURE (not a test module): `bench.py`'s N-rank job on a box WITHOUT GPUs — the launcher
(`launch_children`: torch.distributed.run, 127.0.0.1 rendezvous), the sequence-sharded step, the barrier + max-over-ranks
timing, rc propagation and the one-JSON-line contract — with the gloo backend, a toy model and the oracle-backed kernel
shims of tests/cpu_kernel_shim.py in place of the HIP operators.  tests/test_bench_dryrun_cpu.py runs it.
    python tests/bench_dryrun.py --gpus 2 --steps 1 --warmup 0 --frames 12"""
import contextlib
import os
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import bench  # noqa: E402


class _HostEvent:
    def record(self):
        self.t = time.perf_counter()

    def elapsed_time(self, other):
        return (other.t - self.t) * 1e3


class CpuEnv:
    backend = "gloo"
    pixels = 96

    def __init__(self):
        self.dev = torch.device("cpu")

    def init_process_group(self, world, rank):
        import torch.distributed as dist
        dist.init_process_group("gloo")          # RANK / WORLD_SIZE / MASTER_* from torch.distributed.run

    def sync(self):
        pass

    def event(self):
        return _HostEvent()

    def kernels(self):
        from cpu_kernel_shim import cpu_kernels
        return cpu_kernels()

    def ensure_built(self):
        pass

    def build_model(self, pd):
        from test_distributed_cpu import build
        torch.set_num_threads(2)
        vlm = build("CrossAttention" if pd else "no_merge")
        return vlm.llm_backbone.llm.config, vlm, "toy hybrid model on CPU shims (dry run of the N-rank job), {T} frames, {L} tokens"


if __name__ == "__main__":
    args = bench.parse_args()
    args.config = 0
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(bench.launch_children(__file__, sys.argv[1:], args.gpus))
    # OpTimers wraps the attributes of timeviper_amd.kernels that are current when it is entered: the shims must be in
    # place first, which `run` guarantees by entering env.kernels() before OpTimers
    bench.run(args, CpuEnv())
