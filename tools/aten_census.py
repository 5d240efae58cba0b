"""Which ATen operators (and at which shapes) still run inside the train step: one eager step under torch.profiler,
grouped by operator + input shapes, device time and call count.  Guides the next glue fusions."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import km_unet_amd
from km_unet_amd import train as T
from torch.profiler import profile, ProfilerActivity

torch.manual_seed(0)
dev = "cuda"
model = km_unet_amd.KM_UNetV3(num_classes=5).to(dev).train()
data = torch.rand(8, 10, 1, 128, 128, device=dev)
step = T.TrainStep(model, data, lr=1e-4, loss="hybrid")

for _ in range(3):
    step(data)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step(data)
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    dt = getattr(e, "self_device_time_total", None)
    if dt is None:
        dt = e.self_cuda_time_total
    if dt > 0 and e.key.startswith("aten::"):
        rows.append((dt, e.count, e.key, str(e.input_shapes)[:150]))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print("ATen device time %.3f ms in %d calls" % (tot / 1e3, sum(r[1] for r in rows)))
for dt, n, k, sh in rows[:int(sys.argv[1]) if len(sys.argv) > 1 else 70]:
    print("%8.1f us %4d x  %-28s %s" % (dt, n, k, sh))
