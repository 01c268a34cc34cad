import torch, sys
sys.path.insert(0,'/root/repo')
from propainter_amd import hip
hip.lib()
g=torch.Generator().manual_seed(0)
x=torch.randn(18,60,108,512,generator=g).cuda().half()
w=torch.randn(512,1,4,4,generator=g).cuda().float()
b=torch.randn(512,generator=g).cuda().float()
y=hip.depthwise_pool(x,w,b,4); torch.cuda.synchronize()
import torch.nn.functional as F
ref=F.conv2d(x.float().permute(0,3,1,2),w,b,stride=4,groups=512).permute(0,2,3,1)
print('max err', (y.float()-ref).abs().max().item())
e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): hip.depthwise_pool(x,w,b,4)
e1.record(); torch.cuda.synchronize()
print('us per launch', e0.elapsed_time(e1)/50*1e3, 'GB/s', x.numel()*2/ (e0.elapsed_time(e1)/50*1e-3)/1e9)
