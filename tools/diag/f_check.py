import sys, numpy as np
sys.path.insert(0, '.')
import rtl_fm_player_amd as R
from oracle import OracleStream, lcg_bytes
bl, nb = 262144, 6
import sys as _s
CASES = (dict(rate_in=300000, rate_out2=48000, mode=2), dict(rate_in=240000, rate_out2=48000, mode=2), dict(rate_in=192000, rate_out2=48000, mode=2))
if len(_s.argv) > 1 and _s.argv[1] == "mono":
    CASES = (dict(rate_in=300000, rate_out2=48000, mode=1), dict(rate_in=240000, rate_out2=48000, mode=1), dict(rate_in=192000, rate_out2=48000, mode=1),
             dict(rate_in=25000, rate_out2=12500, mode=1), dict(rate_in=96000, rate_out2=32000, mode=1), dict(rate_in=384000, rate_out2=48000, mode=1))
for kw in CASES:
    iq, _ = lcg_bytes(nb * bl, 12345)
    want, wl = OracleStream(**kw).run(iq, bl)
    outs = {}
    for name, m in (("E", R.MATH_FAST_MFMA_E), ("F", R.MATH_FAST_MFMA_F)):
        b = R.BatchDemod(R.wbfm_config(block_len=bl, math=m, **kw), 1, device=0)
        # two launches of 3 blocks: exercises the carried state
        o1, l1 = b.run_host_concat(iq.reshape(1, nb, bl)[:, :3], 3)
        o2, l2 = b.run_host_concat(iq.reshape(1, nb, bl)[:, 3:], 3)
        out = np.concatenate([o1[0], o2[0]]); lens = np.concatenate([l1[0], l2[0]])
        d = np.abs(out.astype(np.int32) - want.astype(np.int32))
        print(kw['rate_in'], name, "family", b.math, "lens ok", np.array_equal(lens, wl), "max diff", int(d.max()), "n diff", int((d > 0).sum()), "of", d.size, "first bad", np.nonzero(d > 1)[0][:8])
        outs[name] = out
        b.close()
    print("   E vs F: differing values", int((outs["E"] != outs["F"]).sum()))
