import sys; sys.path.insert(0,'/root/repo')
import torch, km_unet_amd
from km_unet_amd import train as T, dp
cnt={'slow':0,'fast':0}
orig=dp.FlatGradBucket.store
def store(self, grads):
    for v,g in zip(self.views, grads):
        if g.stride()!=v.stride() and not g.is_contiguous(): cnt['slow']+=1; print("straggler", tuple(g.shape), g.stride(), v.stride())
        else: cnt['fast']+=1
    return orig(self, grads)
dp.FlatGradBucket.store=store
m=km_unet_amd.KM_UNetV3(num_classes=5).cuda().train()
data=torch.rand(2,10,1,64,64,device='cuda')
st=T.TrainStep(m,data,loss='mse'); st(data); print(cnt)
