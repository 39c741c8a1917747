"""Do the streams a rank of the slab engine uses get hardware queues of their own?  HIP maps
streams onto GPU_MAX_HW_QUEUES queues (bench.py asks for 16); two streams that share a queue
are in order with each other, which is exactly what the engine's lean / fused / posting streams
must not be.  Blocks n-1 streams with a long sleep kernel each and times a tiny kernel on the
n-th: milliseconds = independent, ~the sleep = queued behind one of them."""
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", sys.argv[1] if len(sys.argv) > 1 else "16")
import torch  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
x = torch.zeros(1024, device=dev)
torch.cuda.synchronize()
clock_hz = 1.0e8  # _sleep counts the 100 MHz wall clock on this image (measured below)
t0 = time.perf_counter()
torch.cuda._sleep(int(1e8))
torch.cuda.synchronize()
unit = (time.perf_counter() - t0) / 1e8
print("GPU_MAX_HW_QUEUES=%s; _sleep(1e8) took %.3f s" % (os.environ["GPU_MAX_HW_QUEUES"], unit * 1e8))
cycles = int(0.5 / unit)  # half a second
for label, prios in (("2 high (chain, fused)", [-1, -1]),
                     ("2 high + 3 normal (chain, fused, bulk, post x 2)", [-1, -1, 0, 0, 0]),
                     ("2 high + 9 normal (+ six communicator streams)", [-1, -1] + [0] * 9),
                     ("2 high + 14 normal", [-1, -1] + [0] * 14)):
    streams = [torch.cuda.Stream(dev, priority=p) for p in prios]
    worst = 0.0
    for probe in range(len(streams)):
        torch.cuda.synchronize()
        for k, s in enumerate(streams):
            if k != probe:
                with torch.cuda.stream(s):
                    torch.cuda._sleep(cycles)
        e = torch.cuda.Event()
        t0 = time.perf_counter()
        with torch.cuda.stream(streams[probe]):
            x.add_(1.0)
            e.record()
        e.synchronize()
        worst = max(worst, time.perf_counter() - t0)
        torch.cuda.synchronize()
    print("%-55s slowest probe %.1f ms (sleep 500 ms)" % (label, worst * 1e3))
