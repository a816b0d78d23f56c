import sys, os, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, rtl_fm_player_amd as R
BL=262144
dev=torch.device("cuda:0")
cfg=R.wbfm_config(block_len=BL, math=R.MATH_FAST, rate_in=300000, rate_out2=48000, mode=2)
b=R.BatchDemod(cfg,256,device=0)
iq=torch.randint(0,256,(256,16,BL),dtype=torch.uint8,device=dev)
pcm=torch.zeros((256,16,b.pcm_stride),dtype=torch.int16,device=dev)
lens=torch.zeros((256,16),dtype=torch.int32,device=dev)
torch.cuda.synchronize()
ms=[]
for i in range(400):
    b.run_device(iq,16,pcm,lens); ms.append(b.last_kernel_ms())
ms=np.array(ms)
print([round(float(ms[i:i+20].mean()),3) for i in range(0,400,20)])
import time; time.sleep(2.0)
ms=[]
for i in range(60):
    b.run_device(iq,16,pcm,lens); ms.append(b.last_kernel_ms())
print("after 2 s idle:", [round(float(x),3) for x in ms[:12]], round(float(np.mean(ms[40:])),3))
