"""What the box's memory system delivers to the simplest kernels (run on the GPU box): device-to-device copy
(read + write, the mix the filter's kernels have), fill (write only) and a reduction (read only), at a working set
that streams from HBM (2 x 2 GiB), at the headline step's own (2 x 238 MiB) and at one that fits the 256 MB Infinity
Cache (2 x 64 MiB).  The step's kernels
are judged against 8 TB/s in bench.py; this is the yardstick next to it: a kernel that moves bytes as fast as a
plain copy of the same size cannot be sped up by rearranging its memory accesses."""
import json

import torch


def rate(fn, nbytes, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return nbytes * reps / (a.elapsed_time(b) * 1e-3) / 1e12


def main():
    out = {}
    # (step_238MiB: the headline step reads 244 MB and writes 233 MB per step -- PMC, profiles/traffic_per_launch.json)
    for name, n in (("hbm_2GiB", 2 << 30), ("step_238MiB", 238 << 20), ("cache_64MiB", 64 << 20)):
        x = torch.empty(n // 8, dtype=torch.float64, device="cuda").normal_()
        y = torch.empty_like(x)
        reps = 20 if n > (1 << 30) else 400
        out[name] = {
            "copy_TBps_read_plus_write": round(rate(lambda: y.copy_(x), 2 * n, reps), 3),
            "fill_TBps": round(rate(lambda: y.fill_(1.0), n, reps), 3),
            "sum_TBps": round(rate(lambda: x.sum(), n, reps), 3),
        }
        del x, y
    print(json.dumps(out))


if __name__ == "__main__":
    main()
