import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import numpy as np, torch
import rtl_fm_player_amd as R
from oracle import lcg_bytes, OracleStream
BL = 262144; nb = 3
raw = lcg_bytes(nb * BL, 5)[0]
want, wl = OracleStream(rate_in=300000, rate_out2=48000, mode=2).run(raw, BL)
for use_wait in (True, False):
    dev = torch.device("cuda:0")
    b = R.BatchDemod(R.wbfm_config(math=R.MATH_EXACT, rate_in=300000, rate_out2=48000, mode=2), 1)
    host = torch.from_numpy(raw.copy()).pin_memory()
    iq = torch.zeros(nb * BL, dtype=torch.uint8, device=dev)
    pcm = torch.zeros(nb * b.pcm_stride, dtype=torch.int16, device=dev)
    lens = torch.zeros(nb, dtype=torch.int32, device=dev)
    big = torch.randn((8192, 8192), device=dev)
    torch.cuda.synchronize()
    for _ in range(30): big = (big @ big) * 1e-4
    iq.copy_(host, non_blocking=True)
    if use_wait: b.wait_stream()
    b.run_device(iq, nb, pcm, lens); b.sync()
    torch.cuda.synchronize()
    gl = lens.cpu().numpy()
    got = np.concatenate([pcm[k * b.pcm_stride:k * b.pcm_stride + int(gl[k])].cpu().numpy() for k in range(nb)])
    print("wait_stream", use_wait, "-> equal to the oracle:", bool(np.array_equal(gl, wl) and np.array_equal(got, want)))
    b.close()
