"""Does a zero-fill captured in a hipGraph clear its destination at every replay?  torch.zeros / zero_() (hipMemsetAsync
nodes or fill kernels, as torch chooses) followed by an accumulation, replayed."""
import os, sys, torch
dev = torch.device("cuda:0")
print("DEBUG_CLR_GRAPH_PACKET_CAPTURE =", os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE"))
for n in (1, 64, 768, 1 << 16, 1 << 22):
    out = torch.empty(n, device=dev)
    keep = torch.empty(n, device=dev)
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        a = torch.zeros(n, device=dev); a += 1
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        a = torch.zeros(n, device=dev)
        a += 1
        out.copy_(a)
        keep.zero_()
        keep += 2
    res = []
    for _ in range(4):
        g.replay(); torch.cuda.synchronize()
        res.append((float(out.max()), float(keep.max())))
    print(n, res)
