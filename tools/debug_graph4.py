import torch
torch.manual_seed(0)
x = torch.rand(8, 5, 128, 128, device="cuda")
ref_sum, ref_mm = x.sum().item(), [v.item() for v in torch.aminmax(x)]
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2):
        a = x.sum(); b = torch.aminmax(x); c = (x * 2).mean()
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    a = x.sum(); mn, mx = torch.aminmax(x); c = (x * 2).mean()
for i in range(6):
    if i == 2:
        torch.cuda.current_stream().synchronize()
    g.replay()
    print(i, "sum %.3f (ref %.3f)  min %.3g max %.6f (ref %.3g %.6f) mean2 %.5f" % (a.item(), ref_sum, mn.item(), mx.item(), ref_mm[0], ref_mm[1], c.item()))
