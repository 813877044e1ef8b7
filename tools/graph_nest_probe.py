"""Does HIP stream capture accept NESTED forks (a stream forked from a forked stream, joined back the same way)?  Captures a
few trivial kernels in three topologies and replays them; prints ok / the exception per topology.  (A segmentation fault inside
hipStreamEndCapture is the other possible answer: run each topology in its own process.)"""
import sys
import torch

dev = torch.device("cuda", 0)
x = torch.zeros(1 << 20, device=dev)
topo = sys.argv[1] if len(sys.argv) > 1 else "flat"
A, B, C = (torch.cuda.Stream(dev) for _ in range(3))
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    main = torch.cuda.current_stream(dev)
    x.add_(1)
    if topo == "flat":                      # two siblings forked from the origin stream
        for s in (A, B):
            s.wait_stream(main)
            with torch.cuda.stream(s):
                x.mul_(1.0)
        for s in (A, B):
            main.wait_stream(s)
    elif topo == "nested":                  # B forked from A, joined into A, A joined into the origin
        A.wait_stream(main)
        with torch.cuda.stream(A):
            x.mul_(1.0)
            B.wait_stream(A)
            with torch.cuda.stream(B):
                x.mul_(1.0)
            A.wait_stream(B)
            x.mul_(1.0)
        main.wait_stream(A)
    elif topo == "nested2":                 # as nested, repeated several times with a second nested child
        for _ in range(4):
            A.wait_stream(main)
            with torch.cuda.stream(A):
                x.mul_(1.0)
                for s in (B, C):
                    s.wait_stream(A)
                    with torch.cuda.stream(s):
                        x.mul_(1.0)
                for s in (B, C):
                    A.wait_stream(s)
                x.mul_(1.0)
            x.add_(0)                       # origin-stream work beside A
            main.wait_stream(A)
    elif topo in ("prejoin", "prejoin2"):   # children join the capture by waiting on the ORIGIN first (flat), then take work from A
        reps = 1 if topo == "prejoin" else 4
        for _ in range(reps):
            for s in (A, B, C):
                s.wait_stream(main)
            with torch.cuda.stream(A):
                x.mul_(1.0)
                for s in (B, C):
                    s.wait_stream(A)            # cross edge between two streams that both forked from the origin
                    with torch.cuda.stream(s):
                        x.mul_(1.0)
                for s in (B, C):
                    A.wait_stream(s)
                x.mul_(1.0)
            x.add_(0)
            for s in (A, B, C):
                main.wait_stream(s)
    x.add_(1)
g.replay()
g.replay()
torch.cuda.synchronize()
print(topo, "ok", float(x[0]))
