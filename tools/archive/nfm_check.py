import sys, os, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, rtl_fm_player_amd as R
from oracle import OracleStream
BL=262144
dev=torch.device("cuda:0")
kw=dict(rate_in=25000, rate_out2=12500, mode=1)
cfg=R.wbfm_config(block_len=BL, math=R.MATH_FAST, **kw)
S,B=int(os.environ.get("NSTREAMS","8")),16
b=R.BatchDemod(cfg,S,device=0)
g=torch.Generator(device=dev); g.manual_seed(12345)
iq=torch.randint(0,256,(S,B,BL),dtype=torch.uint8,device=dev,generator=g)
pcm=torch.zeros((S,B,b.pcm_stride),dtype=torch.int16,device=dev)
lens=torch.zeros((S,B),dtype=torch.int32,device=dev)
torch.cuda.synchronize(); b.run_device(iq,B,pcm,lens); b.sync()
for s in range(S):
    want,wl=OracleStream(**kw).run(iq[s].cpu().numpy().reshape(-1),BL)
    l=lens[s].cpu().numpy(); p=pcm[s].cpu().numpy()
    got=np.concatenate([p[k,:l[k]] for k in range(B)])
    d=np.abs(got.astype(np.int32)-want.astype(np.int32))
    if d.max() > 1 or s < 4: print(s, int(d.max()), np.bincount(d)[:4], "argmax", int(d.argmax()), "want", int(want[d.argmax()]), "got", int(got[d.argmax()]))
