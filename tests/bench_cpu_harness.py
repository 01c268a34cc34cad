"""TEST-ONLY entry point: runs bench.py's main() -- launcher, rendezvous, per-rank clips or sub-video shards, barrier + max-over-ranks
timing, the JSON line -- on the CPU over gloo with the stand-in models of tests/test_sharding_cpu.py.  It exists so that
``--gpus N`` (N > 1) is exercised end to end where no GPU is available; a real run never passes a runtime to bench.main().

    python tests/bench_cpu_harness.py --gpus 2 --steps 1 --warmup 0 --height 24 --width 32 --frames 12
(bench.py re-executes THIS file under torch.distributed.run when WORLD_SIZE is unset: argv[0] is what it relaunches.)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import bench  # noqa: E402


class CpuRuntime(bench.DeviceRuntime):
    backend = "gloo"
    graphs = False
    extras = False

    def n_devices(self):
        return None                      # no device limit on the CPU

    def device(self, local):
        import torch
        torch.set_num_threads(1)
        return torch.device("cpu")

    def init_library(self):
        pass

    def models(self, dev, raft_dtype):
        from tests.test_sharding_cpu import MODELS
        return MODELS

    def pin(self, t):
        return t

    def sync(self):
        pass

    def event(self):
        import time

        class _E:
            def record(self):
                self.t = time.perf_counter()

            def elapsed_time(self, other):
                return (other.t - self.t) * 1e3
        return _E()

    def reset_peak(self, dev):
        pass

    def peak_allocated(self, dev):
        return 0

    peak_reserved = reserved = peak_allocated

    def free_bytes(self, dev):
        return 1 << 40

    def total_memory(self, dev):
        return 1 << 40

    def empty_cache(self):
        pass


if __name__ == "__main__":
    bench.main(sys.argv, runtime=CpuRuntime())
