"""Per-node cost of a replayed HIP graph: chains of trivial kernels, host enqueue vs steady-state time per replay.
GPU box: python scripts/graph_node_cost.py"""
import time, torch
dev = "cuda:0"
x = torch.zeros(64, device=dev)
big = torch.zeros(9696, 64, device=dev)
for n_nodes, tensor, label in ((64, x, "64-element add_"), (64, big, "9696x64 add_"), (16, x, "64-element add_")):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            tensor.add_(1.0)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n_nodes):
            tensor.add_(1.0)
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    N = 200
    t0 = time.perf_counter()
    for _ in range(N):
        g.replay()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{n_nodes:3d} chained nodes of {label:18s}: host enqueue {1e6*(t1-t0)/N:7.1f} us/replay, steady state {1e6*(t2-t0)/N:7.1f} us/replay = {1e6*(t2-t0)/N/n_nodes:5.2f} us/node")
