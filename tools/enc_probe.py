import sys, time
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "image-compression_amd", "python"))
import torch
from khugle_amd import Codec
H,W = 4320,7680
c = Codec(H,W)
frames = [c.synth_dev(H,W,2025,i) for i in range(4)]
out = torch.zeros(64<<20, dtype=torch.uint8, device='cuda')
for i in range(3):
    c.encode_dev(frames[i%4],1,out); n=c.encode_result()
print('comp bytes', n, 'bpp', n*8/(H*W))
torch.cuda.synchronize()
t=time.time()
K=20
for i in range(K):
    c.encode_dev(frames[i%4],1,out)
n=c.encode_result()
torch.cuda.synchronize()
dt=(time.time()-t)/K
print('encode ms/frame %.3f  MP/s %.0f' % (dt*1e3, H*W/dt/1e6))
