"""Where the HOST time of a training step goes (round 6): wall time inside every C-ABI entry point of the step
(ctypes calls timed in place) against the step's total enqueue time, on the bench's headline workload.
    python tools/host_profile.py [--steps 40]
"""
import argparse
import collections
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import bench                                                       # noqa: E402
from graphinvent_amd import lib as L, ops                          # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=40)
    args = ap.parse_args()
    device = torch.device("cuda:0")
    torch.cuda.set_device(device)
    wl = bench.Workload(bench.SHAPE, bench.MODEL, bench.BATCH, 0, device, args.steps + 20)
    lib = L.load()
    spent = collections.defaultdict(float)
    calls = collections.Counter()

    class Timed:
        def __init__(self, name, fn):
            self.name, self.fn = name, fn

        def __call__(self, *a):
            t = time.perf_counter()
            try:
                return self.fn(*a)
            finally:
                spent[self.name] += time.perf_counter() - t
                calls[self.name] += 1

    for _ in range(8):
        wl.run_step()
    torch.cuda.synchronize()
    for name in L.SIGNATURES:
        fn = getattr(lib, name, None)
        if fn is not None:
            setattr(lib, name, Timed(name, fn))
    wait0 = ops.HOST_WAIT[0]
    t0 = time.perf_counter()
    for _ in range(args.steps):
        wl.run_step()
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    wait = ops.HOST_WAIT[0] - wait0
    n = args.steps
    print(f"step {t_all / n * 1e3:.3f} ms; host: enqueue {(t_enq - wait) / n * 1e3:.3f} ms per step "
          f"(+ {wait / n * 1e3:.3f} ms waiting for the device)")
    inside = sum(spent.values())
    print(f"inside the C ABI: {inside / n * 1e3:.3f} ms per step; Python / torch around it: {(t_enq - wait - inside) / n * 1e3:.3f}")
    for name, s in sorted(spent.items(), key=lambda kv: -kv[1]):
        print(f"  {name:32s} {calls[name] / n:5.1f} calls/step  {s / n * 1e6:8.1f} us/step")


if __name__ == "__main__":
    main()
