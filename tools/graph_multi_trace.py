import sys, torch
dev = "cuda:0"
n = 8
x, y = torch.randn(16384, device=dev), torch.randn(32768, device=dev)
def chain(v, k):
    for _ in range(k): v = v * 1.0001 + 0.5
    return v
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
ga, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
torch.cuda.synchronize()
with torch.cuda.graph(ga, stream=s1): a = chain(x, n).sum()
with torch.cuda.graph(gb, stream=s2): b = chain(y, n).sum()
torch.cuda.synchronize()
for _ in range(3):
    main = torch.cuda.current_stream()
    s1.wait_stream(main); s2.wait_stream(main)
    with torch.cuda.stream(s1): ga.replay()
    with torch.cuda.stream(s2): gb.replay()
    main.wait_stream(s1); main.wait_stream(s2)
torch.cuda.synchronize()
# eager, two streams
for _ in range(2):
    with torch.cuda.stream(s1): a = chain(x, n).sum()
    with torch.cuda.stream(s2): b = chain(y, n).sum()
torch.cuda.synchronize()
