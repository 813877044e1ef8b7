"""Which hardware queue does the process group's RCCL stream share?  (one rank, nccl backend)

HIP multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (rspnet_amd/streams.py).  ProcessGroupNCCL runs its collectives on an
internal stream of its own: if that stream shares a hardware queue with the step's main lane, a bucket all-reduce issued from the "w"
lane (ordered behind the weight gradients there) sits IN FRONT of the main lane's next graphs in that queue and holds them up until the
weight gradients are done.  Probe: stream A spins for ~0.4 ms; a tiny all-reduce is issued from an idle stream B that overlaps with A;
if it completes only after the spin, RCCL's stream is queued behind A."""
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rspnet_amd import streams  # noqa: E402


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    x = torch.zeros(1 << 20, device=dev)
    dist.all_reduce(x)                       # communicator + RCCL stream exist from here on
    torch.cuda.synchronize()
    main_s = torch.cuda.current_stream(dev)
    lanes = {n: streams.lane(dev, n) for n in ("q", "k", "w")}
    print("lanes overlap main:", streams.lanes_overlap(dev))
    cycles = streams._spin_cycles(dev)
    cands = {"main": main_s, **lanes}
    for i in range(6):
        cands[f"pool{i}"] = torch.cuda.Stream(device=dev)
    ev = torch.cuda.Event()

    def timed(a, b, busy):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if busy:
            with torch.cuda.stream(a):
                torch.cuda._sleep(cycles)
        with torch.cuda.stream(b):
            h = dist.all_reduce(x, async_op=True)
            h.wait()
            ev.record(b)
        ev.synchronize()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        return (t1 - t0) * 1e3

    for name, a in cands.items():
        # an issuing stream that runs beside `a`
        b = next((s for n, s in cands.items() if s.cuda_stream != a.cuda_stream and streams.overlap(a, s, dev)), None)
        if b is None:
            print(name, "no overlapping issuer")
            continue
        idle = min(timed(a, b, False) for _ in range(3))
        busy = min(timed(a, b, True) for _ in range(3))
        print(f"{name:6s} all-reduce from an idle stream: {idle:.3f} ms alone, {busy:.3f} ms while {name} spins -> "
              f"{'BLOCKED behind it (same hardware queue)' if busy > idle + 0.2 else 'runs beside it'}")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
