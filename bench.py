#!/usr/bin/env python3
"""Benchmark of the IQ -> PCM hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

A step is one pass of the fused kernel over one resident batch of synthetic u8
IQ: `--streams` independent 2.4 Msps wide-FM stereo streams per GPU (default
256, BASELINE.json configs[2]) x `--blocks` reference blocks of 262144 bytes.
Streams shard across GPUs with no data-path collective (weak scaling); RCCL is
used only to gather the per-rank counters.  Rank 0 prints ONE JSON line.

N > 1: one process per GPU.  Under `python -m torch.distributed.run --nproc-per-node N
bench.py --gpus N ...` the ranks are already there (RANK / LOCAL_RANK / WORLD_SIZE in the
environment; --gpus must equal WORLD_SIZE).  Run plainly as `python bench.py --gpus N`,
this process becomes a launcher: BEFORE importing torch or touching HIP it starts N fresh
children of itself with those variables set, forwards rank 0's JSON line and exits
non-zero if any child fails (never an exec: a process that has initialised the GPU must
not be replaced).
"""
import argparse
import json
import math
import os
import socket
import subprocess
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BLOCK_LEN = 262144            # MAXIMUM_BUF_LENGTH, one reference block of u8 IQ
HBM_PEAK_GBS = 8000.0         # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec



def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--preheat", type=int, default=60,
                    help="untimed launches before the warm-up steps so that at least this many launches precede the "
                         "timed region: after idle the part needs ~40 launches (30 ms) to settle its clocks "
                         "(tools/ramp_check.py: 0.76 ms per launch at first, 0.63 ms from then on)")
    ap.add_argument("--streams", type=int, default=256, help="streams per GPU")
    ap.add_argument("--blocks", type=int, default=16, help="reference blocks per stream per step")
    ap.add_argument("--math", choices=["fast", "exact", "fast-valu", "fast-mfma", "fast-mfma-f"], default="fast",
                    help="fast = the +-1 LSB kernel family the library picks for the configuration; "
                         "fast-valu / fast-mfma / fast-mfma-f name one (A/B runs)")
    ap.add_argument("--mode", choices=["stereo", "mono", "nfm"], default="stereo",
                    help="stereo/mono: 2.4 Msps WBFM (rate_in 300k -> 48k); nfm: 200 ksps (25k -> 12.5k mono)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the CPU baseline leg")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--no-e2e", action="store_true", help="skip the H2D-inclusive leg (e2e_h2d)")
    ap.add_argument("--no-extra", action="store_true",
                    help="skip the extra legs of the line: `sustained` (>= --sustained-seconds of back-to-back launches) "
                         "`noise_input` / `quiet_input` (the same workload on uniform random bytes / on bytes in {127, 128} with mute "
                         "fills, parity-checked) and `modes` (mono and NFM on the same device)")
    ap.add_argument("--sustained-seconds", type=float, default=1.0)
    ap.add_argument("--only-sustained", action="store_true",
                    help="of the extra legs run `sustained` (with its hwmon power / clock samples) alone: tools/energy_ablate.sh")
    ap.add_argument("--e2e-streams", type=int, default=64)
    ap.add_argument("--e2e-jobs", type=int, default=20)
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed (nccl = RCCL) and run the counter gather through its collectives "
                         "even with one rank: executes the multi-GPU branch on a one-GPU box")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher / counter-gather plumbing only: no GPU work, gloo instead of RCCL (CPU tests)")
    ap.add_argument("--data", choices=["fm", "noise"], default="fm",
                    help="fm: synthetic FM broadcast per stream (stereo multiplex with 19 kHz pilot and L-R DSB, "
                         "BASELINE.json configs[1..3]; NFM: 1 kHz tone, 5 kHz deviation); noise: uniform random bytes")
    return ap.parse_args()


def usable_cores():
    """Host cores this process may really use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, q // per))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


CPU_CONFIGS = {   # BASELINE.json configs[0], [1], [4]
    "mono_2.4Msps": dict(rate_in=300000, rate_out2=48000, mode=1),
    "stereo_2.4Msps": dict(rate_in=300000, rate_out2=48000, mode=2),
    "nfm_200ksps": dict(rate_in=25000, rate_out2=12500, mode=1),
}


def _cpu_stream_factory():
    """The CPU side that is timed: oracle/_ref/libref.so - the reference's own full_demod()
    compiled from its sources by oracle/build_ref.py (kind "reference") - when it was built,
    otherwise the bit-exact restatement oracle/fm_oracle.c (kind "port")."""
    try:
        from oracle import refbind
        from oracle.build_ref import ref_status
        st = ref_status()
        # only the library build_ref made from the PINNED reference lines (oracle/ref_pin.json) and whose bytes on
        # disk still hash to what it recorded is timed as "the reference"
        if st and st["pinned"] and refbind.have_ref():
            return refbind.RefStream, "reference", ("oracle/_ref/libref.so = the reference's rotate_90_u8_f32 + full_demod, "
                                                    "gcc -O3; reference lines sha256 %s, library sha256 %s"
                                                    % (st["slices_sha256"][:16], st["so_sha256"][:16]))
    except Exception:                       # noqa: BLE001 - any problem with the optional library: use the port
        pass
    from oracle import OracleStream
    return OracleStream, "port", "oracle/fm_oracle.c, -O3 -ffp-contract=off"


def cpu_baseline(cfg_kw, seconds, iq_dev=None):
    """The reference's CPU full_demod() on this host's cores: one independent stream per thread
    on a bounded sample of the same workload (the first 8 blocks of the GPU run's own streams),
    plus 1 thread x 1 stream for each of BASELINE.json's single-stream configurations."""
    from oracle import lcg_bytes
    make, kind, what_lib = _cpu_stream_factory()
    n_thr = usable_cores()
    nb = 8
    if iq_dev is not None and iq_dev.shape[1] >= nb:
        iqs = [np.ascontiguousarray(iq_dev[t % iq_dev.shape[0], :nb].cpu().numpy().reshape(-1)) for t in range(n_thr)]
        what = "the run's own IQ"
    else:
        iqs = [lcg_bytes(nb * BLOCK_LEN, 12345 + t)[0] for t in range(n_thr)]
        what = "LCG u8 IQ"
    streams = [make(**cfg_kw) for _ in range(n_thr)]
    streams[0].run(iqs[0], BLOCK_LEN)                      # page in / warm up

    def one_thread(kw, budget):
        st = make(**kw)
        st.run(iqs[0], BLOCK_LEN)
        t1, n1 = time.perf_counter(), 0
        while time.perf_counter() - t1 < budget:
            st.run(iqs[0], BLOCK_LEN)
            n1 += nb * BLOCK_LEN // 2
        return round(n1 / (time.perf_counter() - t1) / 1e6, 2)

    per_cfg = min(2.0, seconds / 6)
    configs = {name: one_thread(kw, per_cfg) for name, kw in CPU_CONFIGS.items()}   # 1 thread x 1 stream
    single = one_thread(cfg_kw, per_cfg)
    done = [0] * n_thr
    stop = time.perf_counter() + seconds

    def work(t):
        while time.perf_counter() < stop:
            streams[t].run(iqs[t], BLOCK_LEN)
            done[t] += nb * BLOCK_LEN // 2

    t0 = time.perf_counter()
    thr = [threading.Thread(target=work, args=(t,)) for t in range(n_thr)]
    [x.start() for x in thr]
    [x.join() for x in thr]
    dt = time.perf_counter() - t0
    return {
        "value": round(sum(done) / dt / 1e6, 2),
        "unit": "Msamples/s",
        "cores": n_thr,
        "kind": kind,
        "single_thread": single,
        "configs_single_thread": configs,
        "sample": "%d threads (affinity capped by the cgroup CPU quota) x independent streams on %s, %d-block "
                  "runs of %s for %.0f s (%s); configs_single_thread: 1 thread x 1 stream, %.1f s each"
                  % (n_thr, cpu_model(), nb, what, dt, what_lib, per_cfg),
    }


def e2e_h2d(mode, n_streams, n_blocks, n_jobs, math_code, cfg_kw):
    """IQ that starts in HOST memory, measured without Python in the loop: rtl_fm_player_amd/fmd_e2e_bench (C, csrc/
    fmd_e2e_bench.c) runs as a child process - feeder pthreads in the dongle threads' role call fmd_ingest_callback one
    262144-byte transfer at a time (src/rtl_fm_player.c:790-837), its main thread is the demod thread (:855-933):
    fmd_batch_pump_begin / _end with two jobs in flight, H2D straight from the pinned rings, PCM back in host memory.
    Wall clock over everything.  (Round 2 fed the rings from Python threads: 12 GB/s, the harness's own limit.)"""
    exe = os.path.join(ROOT, "rtl_fm_player_amd", "fmd_e2e_bench")
    if not os.path.isfile(exe):
        raise RuntimeError("rtl_fm_player_amd/fmd_e2e_bench not built (python -c 'import __graft_entry__ as g; g.build()')")
    n_thr = max(1, min(usable_cores(), n_streams, 16))
    # the tool runs exactly the bench's configuration and kernel family, and says so in its line ("config")
    cmd = [exe, "-S", str(n_streams), "-B", str(n_blocks), "-J", str(max(2, n_jobs)), "-T", str(n_thr),
           "-m", str(cfg_kw["mode"]), "-i", str(cfg_kw["rate_in"]), "-o", str(cfg_kw["rate_out2"]), "-M", str(int(math_code))]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    if r.returncode != 0:
        raise RuntimeError("fmd_e2e_bench failed: " + r.stderr[-500:])
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])


def copy_bandwidth(torch, dev, stream, nbytes=1 << 30):
    """Device-to-device copy of 1 GiB (read + write counted), the practical HBM ceiling on this part."""
    src = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    dst = torch.empty_like(src)
    src.zero_()
    best = 0.0
    with torch.cuda.stream(stream):
        for _ in range(2):
            dst.copy_(src)
        for _ in range(5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(stream)
            dst.copy_(src)
            b.record(stream)
            b.synchronize()
            best = max(best, 2.0 * nbytes / (a.elapsed_time(b) * 1e-3) / 1e9)
    del src, dst
    return best


def synth_fm_iq(torch, dev, n_streams, n_samples, fs, wide, seed):
    """u8 IQ of n_streams synthetic FM stations, made on the device (float64 phase, reproducible).

    Wide FM: stereo multiplex 0.45 (L+R) + 0.45 (L-R) sin(2 w_p t) + 0.1 sin(w_p t), w_p = 19 kHz pilot,
    L / R tones per stream, +-75 kHz deviation.  Narrow FM: one tone, +-5 kHz.  The carrier sits at
    -fs/4 (rotate_90_u8_f32 centres it), amplitude 100 LSB around 127.5, +-2 LSB of uniform noise.
    """
    out = torch.empty((n_streams, 2 * n_samples), dtype=torch.uint8, device=dev)
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    n = torch.arange(n_samples, dtype=torch.float64, device=dev)
    quarter = (torch.arange(n_samples, device=dev) % 4).to(torch.float64) * (-math.pi / 2)   # -fs/4, exact per sample
    two_pi = 2.0 * math.pi
    for s in range(n_streams):
        k = seed * 131 + s
        f_l, f_r = 400.0 + 37.0 * (k % 97), 1000.0 + 53.0 * (k % 89)
        if wide:
            left, right = torch.sin(two_pi * f_l / fs * n), torch.sin(two_pi * f_r / fs * n)
            pilot = two_pi * 19000.0 / fs * n
            mpx = 0.45 * (left + right) + 0.45 * (left - right) * torch.sin(2.0 * pilot) + 0.1 * torch.sin(pilot)
            dev_hz = 75000.0
        else:
            mpx = torch.sin(two_pi * f_l / fs * n)
            dev_hz = 5000.0
        phase = torch.cumsum(mpx, 0) * (two_pi * dev_hz / fs) + quarter
        noise = torch.rand((2, n_samples), dtype=torch.float64, device=dev, generator=g) * 4.0 - 2.0
        i = torch.clamp(torch.round(127.5 + 100.0 * torch.cos(phase) + noise[0]), 0, 255)
        q = torch.clamp(torch.round(127.5 + 100.0 * torch.sin(phase) + noise[1]), 0, 255)
        out[s, 0::2] = i.to(torch.uint8)
        out[s, 1::2] = q.to(torch.uint8)
    return out


def measured_traffic(config):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (FETCH_SIZE x2 per the
    gfx950 correction + WRITE_SIZE, tools/profile_round.sh) when one exists for this workload."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_summary.json"))):
        try:
            t = json.load(open(f)).get("hbm_traffic")
        except (OSError, ValueError):
            continue
        w = t.get("workload") if t else None
        # same workload text and the same kernel family (profiles of earlier rounds carry only "math")
        if w and w.get("workload") == config.get("workload") and \
                w.get("kernel_family", {"fast": "fast-valu"}.get(w.get("math"), w.get("math"))) == config.get("kernel_family"):
            best = (int(t["traffic_bytes_per_launch"]), os.path.basename(f))
    return best


def measured_busy(summary_name):
    """Busy fractions of the CU's units over the kernel's duration, from the PMC passes of the committed profile summary the traffic
    figure comes from (tools/profile_round.sh; counters per launch): vector ALU = SQ_ACTIVE_INST_VALU x 4 / (SIMDs x cycles), matrix
    pipe = SQ_VALU_MFMA_BUSY_CYCLES / (SIMDs x cycles), LDS = SQ_LDS_IDX_ACTIVE / (CUs x cycles) where the summary holds it
    (SQ_ACTIVE_INST_LDS x 4 otherwise), cycles = GRBM_GUI_ACTIVE / 8 XCDs.  Measurements of the profiled build, not of this run."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", summary_name)))
    except (OSError, ValueError):
        return None
    c = {}
    for grp in d.values():
        if isinstance(grp, dict):
            for k, v in grp.items():
                if isinstance(v, dict) and "mean_per_launch" in v:
                    c[k] = float(v["mean_per_launch"])
    if not c.get("GRBM_GUI_ACTIVE") or "SQ_ACTIVE_INST_VALU" not in c:
        return None
    cyc, simds, cus = c["GRBM_GUI_ACTIVE"] / 8.0, 1024.0, 256.0
    out = {"valu": round(c["SQ_ACTIVE_INST_VALU"] * 4.0 / (simds * cyc), 3),
           "mfma": round(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (simds * cyc), 3),
           "lds": round(c["SQ_LDS_IDX_ACTIVE"] / (cus * cyc), 3) if "SQ_LDS_IDX_ACTIVE" in c else
                  round(c.get("SQ_ACTIVE_INST_LDS", 0.0) * 4.0 / (cus * cyc), 3),
           "cycles_per_xcd_under_counters": int(cyc), "source": "profiles/" + summary_name,
           "note": "rocprofv3 PMC per launch of the profiled build: SQ_ACTIVE_INST_VALU x 4, SQ_VALU_MFMA_BUSY_CYCLES over 1024 SIMDs x "
                   "(GRBM_GUI_ACTIVE / 8) cycles; LDS over 256 CUs"}
    return out


def kfd_gpu_count():
    """GPUs of this node as the kernel driver lists them (/sys/class/kfd: nodes with SIMDs), without opening a HIP
    device; None where the topology is not readable (the ranks then find out themselves)."""
    import glob
    nodes = glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties")
    if not nodes:
        return None
    n = 0
    for f in nodes:
        try:
            props = dict(l.split()[:2] for l in open(f) if len(l.split()) >= 2)
        except OSError:
            return None
        n += int(props.get("simd_count", "0")) > 0
    return n


def launch_ranks(args):
    """`python bench.py --gpus N` with no rank environment: start N fresh ranks of this script and supervise them.
    The launcher itself never imports torch and never touches HIP (the device count comes from sysfs); the ranks are
    new processes (subprocess), never an exec of this one.  If a rank exits non-zero the others - who would sit in the
    rendezvous until its timeout - are terminated and the launcher exits non-zero."""
    import tempfile
    n = args.gpus
    if not args.dry_run:
        have = kfd_gpu_count()
        if have is not None and have < n:
            raise SystemExit("bench.py: --gpus %d but this node shows %d HIP device(s)" % (n, have))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    with tempfile.TemporaryFile() as out0:
        for r in range(n):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                       MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
            env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                          stdout=out0 if r == 0 else subprocess.DEVNULL))
        print("bench.py: ranks " + " ".join(str(p.pid) for p in procs), file=sys.stderr, flush=True)
        failed = None
        while failed is None and any(p.poll() is None for p in procs):
            for r, p in enumerate(procs):
                if p.poll() not in (None, 0):
                    failed = r
            time.sleep(0.05)
        if failed is not None:                      # stop the others: SIGTERM, then SIGKILL after a grace period
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            deadline = time.time() + 5.0
            for p in procs:
                try:
                    p.wait(max(0.1, deadline - time.time()))
                except subprocess.TimeoutExpired:
                    p.kill()
                    p.wait()
        codes = [p.wait() for p in procs]
        out0.seek(0)
        text = out0.read().decode()
    sys.stdout.write(text)
    sys.stdout.flush()
    if any(codes):
        raise SystemExit("bench.py: rank %s failed; rank exit codes %s" % (failed, codes))
    if not any(l.startswith("{") for l in text.splitlines()):
        raise SystemExit("bench.py: rank 0 printed no JSON line")


def single_stream_leg(R, mode_kw, math_code, n_blocks=64):
    """BASELINE configs[1]: ONE 2.4 Msps stream, one reference block (262144 B = 54.6 ms of signal) per call - the operating
    point of the reference's demod thread (src/rtl_fm_player.c:855-933).  Milliseconds per block through
      (a) the reference-shaped surface: rotate_90_u8_f32(d) + full_demod(d) on a struct demod_state (host buffers in and out);
      (b) fmd_batch_run_host with 1 stream x 1 block;
      (c) the reference's own rotate_90_u8_f32 + full_demod on one host core (oracle/_ref/libref.so; the oracle port where the
          compiled reference did not travel), same blocks.
    Never `value`: a latency figure for the drop-in surface, host copies included."""
    import ctypes as C
    from oracle import lcg_bytes
    L = R.lib()
    iq = lcg_bytes(n_blocks * BLOCK_LEN, 777)[0].reshape(n_blocks, BLOCK_LEN)
    out = {"blocks": n_blocks, "block_bytes": BLOCK_LEN, "signal_ms_per_block": round(BLOCK_LEN / 2 / 2.4e6 * 1e3, 2)}
    # (a) reference-shaped calls
    prev_env = os.environ.get("FMD_MATH_FAST")
    if math_code != R.MATH_EXACT:
        os.environ["FMD_MATH_FAST"] = "1"
    d = R.DemodState()
    L.demod_init(C.byref(d))
    d.rate_in = d.rate_out = mode_kw["rate_in"]
    d.rate_out2 = mode_kw["rate_out2"]
    d.lpr.mode = mode_kw["mode"]
    d.lpr.size = 90 if mode_kw["mode"] == 2 else 128
    d.deemph_lambda = float(L.fmd_deemph_lambda(mode_kw["rate_out2"], 50e-6))
    L.init_u8_f32_table(); L.init_lp_f32(); L.init_lp_real_f32(C.byref(d))
    d.buf_len = BLOCK_LEN
    lens_a = []

    def call(k):
        C.memmove(d.buf, iq[k].ctypes.data, BLOCK_LEN)
        L.rotate_90_u8_f32(C.byref(d))
        L.full_demod(C.byref(d))
        lens_a.append(d.result_len)

    for k in range(4):
        call(k)                                        # builds the batch, settles allocations
    t0 = time.perf_counter()
    for k in range(4, n_blocks):
        call(k)
    out["dropin_ms_per_block"] = round((time.perf_counter() - t0) / (n_blocks - 4) * 1e3, 4)
    pcm_tail = np.ctypeslib.as_array(d.result)[:d.result_len].copy()
    L.deinit_lp_real_f32(C.byref(d))
    if prev_env is None:
        os.environ.pop("FMD_MATH_FAST", None)
    # (b) batch API, one stream, one block per call
    b = R.BatchDemod(R.wbfm_config(block_len=BLOCK_LEN, math=math_code, **mode_kw), 1)
    for k in range(4):
        b.run_host(iq[k:k + 1], 1)
    t0 = time.perf_counter()
    for k in range(4, n_blocks):
        pcm_b, lens_b = b.run_host(iq[k:k + 1], 1)
    out["batch_run_host_ms_per_block"] = round((time.perf_counter() - t0) / (n_blocks - 4) * 1e3, 4)
    b.close()
    # (c) the reference on one host core
    kind, cls = "port", None
    try:
        from oracle import refbind
        if refbind.have_ref():
            kind, cls = "reference", refbind.RefStream
    except ImportError:
        pass
    if cls is None:
        from oracle import OracleStream as cls
    o = cls(**mode_kw)
    o.run(iq[:4].reshape(-1), BLOCK_LEN)
    t0 = time.perf_counter()
    want, wl = o.run(iq[4:].reshape(-1), BLOCK_LEN)
    out["cpu_ms_per_block"] = round((time.perf_counter() - t0) / (n_blocks - 4) * 1e3, 4)
    out["cpu_kind"] = kind
    # the last block of (a) and (b) against (c): the drop-in really demodulated the stream
    tail = want[-int(wl[-1]):]
    tol = 0 if math_code == R.MATH_EXACT else 1
    da = int(np.abs(pcm_tail.astype(np.int32) - tail.astype(np.int32)).max()) if pcm_tail.size == tail.size else 1 << 20
    db = int(np.abs(pcm_b[0, 0, :lens_b[0, 0]].astype(np.int32) - tail.astype(np.int32)).max()) if lens_b[0, 0] == tail.size else 1 << 20
    out["parity_last_block_lsb"] = {"dropin": da, "batch": db, "tolerance": tol}
    assert da <= tol and db <= tol, "single-stream leg: PCM differs from the CPU path (%d / %d LSB)" % (da, db)
    out["dropin_vs_cpu"] = round(out["cpu_ms_per_block"] / out["dropin_ms_per_block"], 2)
    return out


def pick_device(local_rank, world, visible):
    """Device index of this rank.  A launcher may give every rank ALL devices (index = LOCAL_RANK) or exactly ONE
    (HIP_VISIBLE_DEVICES=k per rank, LOCAL_RANK=k still set): with fewer visible devices than LOCAL_RANK asks for and
    several ranks running, the rank takes LOCAL_RANK modulo what it sees instead of giving up - the first case is
    unchanged, the second binds device 0.  A single-rank run with an impossible LOCAL_RANK is still an error."""
    if visible < 1:
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    if local_rank < visible:
        return local_rank
    if world > 1:
        return local_rank % visible
    raise SystemExit("bench.py: LOCAL_RANK %d but only %d HIP device(s)" % (local_rank, visible))


def dry_run(args, rank, world):
    """The N-rank plumbing without a GPU (tests/test_bench_launcher.py): gloo, invented counters."""
    import torch
    import torch.distributed as dist
    from rtl_fm_player_amd.shard import gather_counters, shard_streams
    if os.environ.get("FMD_BENCH_DRYRUN_FAIL_RANK") == str(rank):      # tests: a rank that dies before the rendezvous
        sys.exit(3)
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    first, count = shard_streams(args.streams * world, world, rank)
    vis = os.environ.get("HIP_VISIBLE_DEVICES", os.environ.get("ROCR_VISIBLE_DEVICES"))
    n_vis = len([v for v in vis.split(",") if v.strip()]) if vis else int(os.environ.get("FMD_BENCH_DRYRUN_DEVICES", str(world)))
    bound = pick_device(int(os.environ.get("LOCAL_RANK", "0")), world, n_vis)      # what the real run would set_device() to
    samples = count * args.blocks * (BLOCK_LEN // 2) * args.steps
    rep = gather_counters(dist if world > 1 else None, torch.device("cpu"), 1.0 + 0.25 * rank, samples,
                          1000 + rank, first)
    devices = [bound]
    if world > 1:
        devices = [None] * world
        dist.all_gather_object(devices, bound)
    if rank == 0:
        print(json.dumps({"metric": "IQ Msamples/s through full_demod", "dry_run": True, "n_gpus": world, "bound_devices": devices,
                          "steps": args.steps, "warmup": args.warmup, "value": rep["samples"] / rep["elapsed_s"] / 1e6,
                          "unit": "Msamples/s", "scaling": "weak", "per_rank": rep["per_rank"]}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


POWER_BASE_W = 354.0        # package draw with every SIMD issuing s_nop at 2.4 GHz (profiles/r6b_power_price_live_operands.txt; 366 - 367 W on the devices of round 5)
NJ_PER_BYTE = 0.124         # a device-to-device copy above that base, per byte moved (profiles/archive/r33_stream_power.txt)


class PowerWatch:
    """Package power and shader clock of ONE device, read from its hwmon files (amdgpu: power1_input in microwatts, power1_cap,
    freq1_input in Hz) by a thread while a leg of the bench runs.  No rocm-smi process, no HIP call; silently absent where the files are
    (the device is found by its PCI address).  Reported beside `sustained`: the kernel is bound by the package power cap, not by a unit
    of the CU, and this is the driver-run line's own evidence of it."""

    def __init__(self, props, period=0.05):
        import glob
        self.dir, self.period, self.rows, self.th = None, period, [], None
        try:
            want = "%04x:%02x:%02x." % (props.pci_domain_id, props.pci_bus_id, props.pci_device_id)
        except Exception:
            return
        for d in glob.glob("/sys/class/drm/card*/device"):
            if want in os.path.realpath(d) + ".":
                h = glob.glob(d + "/hwmon/hwmon*/power1_input")
                if h:
                    self.dir = os.path.dirname(h[0])

    @staticmethod
    def _read(path):
        try:
            with open(path) as f:
                return float(f.read().split()[0])
        except Exception:
            return None

    def start(self):
        if not self.dir:
            return
        import threading
        self.stop_flag = threading.Event()

        def run():
            while not self.stop_flag.is_set():
                self.rows.append((self._read(self.dir + "/power1_input"), self._read(self.dir + "/freq1_input")))
                self.stop_flag.wait(self.period)
        self.th = threading.Thread(target=run, daemon=True)
        self.th.start()

    def stop(self):
        if not self.th:
            return None
        self.stop_flag.set()
        self.th.join()
        rows = self.rows[len(self.rows) // 4:]          # the first quarter: the sensor's averaging window still holds the time before the leg
        pw = [r[0] * 1e-6 for r in rows if r[0]]
        fq = [r[1] * 1e-6 for r in rows if r[1]]
        if not pw:
            return None
        cap = self._read(self.dir + "/power1_cap")
        return {"package_w_mean": round(sum(pw) / len(pw), 1), "package_w_max": round(max(pw), 1), "cap_w": round(cap * 1e-6, 1) if cap else None,
                "sclk_mhz_mean": round(sum(fq) / len(fq), 1) if fq else None, "samples": len(pw), "source": "hwmon power1_input / freq1_input, 50 ms period"}


def main():
    args = parse()
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return launch_ranks(args)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE is %d: launch one rank per GPU "
                         "(torch.distributed.run --nproc-per-node %d) or run `python bench.py --gpus %d` by itself"
                         % (args.gpus, world, args.gpus, args.gpus))
    if args.dry_run:
        return dry_run(args, rank, world)
    import torch
    import rtl_fm_player_amd as R

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    local = pick_device(local, world, torch.cuda.device_count())
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:             # --force-dist without a launcher: a free port of our own
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    stereo = args.mode == "stereo"
    if args.mode == "nfm":
        cfg_kw = dict(rate_in=25000, rate_out2=12500, mode=1)          # BASELINE.json configs[4]
    else:
        cfg_kw = dict(rate_in=300000, rate_out2=48000, mode=2 if stereo else 1)
    math_code = {"fast": R.MATH_FAST, "exact": R.MATH_EXACT, "fast-valu": R.MATH_FAST_VALU, "fast-mfma": R.MATH_FAST_MFMA,
            "fast-mfma-f": R.MATH_FAST_MFMA_F}[args.math]
    cfg = R.wbfm_config(block_len=BLOCK_LEN, math=math_code, **cfg_kw)
    S, B = args.streams, args.blocks
    batch = R.BatchDemod(cfg, S, device=local)

    # synthetic IQ resident in HBM: uniform random bytes, per-rank seed
    g = torch.Generator(device=dev)
    g.manual_seed(12345 + rank)
    if args.data == "noise":
        iq = torch.randint(0, 256, (S, B, BLOCK_LEN), dtype=torch.uint8, device=dev, generator=g)
    else:
        iq = synth_fm_iq(torch, dev, S, B * BLOCK_LEN // 2, 200e3 if args.mode == "nfm" else 2.4e6,
                         args.mode != "nfm", 12345 + rank).view(S, B, BLOCK_LEN)
    pcm = torch.zeros((S, B, batch.pcm_stride), dtype=torch.int16, device=dev)
    lens = torch.zeros((S, B), dtype=torch.int32, device=dev)
    stream = torch.cuda.Stream(device=dev)          # kernels and timing events share this stream
    torch.cuda.synchronize(dev)

    def step():
        batch.run_device(iq, B, pcm, lens, hip_stream=stream.cuda_stream)

    # ---- parity gate: EVERY stream of this rank against the oracle (host threads; the oracle releases the GIL) ----
    def parity_gate(iq_t):
        from concurrent.futures import ThreadPoolExecutor
        from oracle import OracleStream
        batch.reset()
        batch.run_device(iq_t, B, pcm, lens, hip_stream=stream.cuda_stream)
        torch.cuda.synchronize(dev)
        h_iq, h_pcm, h_lens = iq_t.cpu().numpy(), pcm.cpu().numpy(), lens.cpu().numpy()
        tol = 0 if args.math == "exact" else 1

        def check(s):
            want, wl = OracleStream(**cfg_kw).run(h_iq[s].reshape(-1), BLOCK_LEN)
            if not np.array_equal(h_lens[s], wl):
                return 1 << 20
            got = np.concatenate([h_pcm[s, b, :wl[b]] for b in range(B)])
            return int(np.abs(got.astype(np.int32) - want.astype(np.int32)).max()) if got.size else 0

        with ThreadPoolExecutor(max(1, min(usable_cores(), 32))) as ex:
            diffs = list(ex.map(check, range(S)))
        worst = max(diffs)
        assert worst < (1 << 20), "result_len mismatch on stream %d" % diffs.index(worst)
        assert worst <= tol, "PCM of stream %d differs from the CPU oracle by %d LSB (tolerance %d)" % (
            diffs.index(worst), worst, tol)
        batch.reset()
        return {"max_abs_lsb": worst, "tolerance_lsb": tol, "streams_checked": S}

    def timed_launches(iq_t, n):
        """n back-to-back launches on `stream` between one HIP event pair: (wall seconds, kernel ms per launch)"""
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(dev)
        w0 = time.perf_counter()
        e0.record(stream)
        for _ in range(n):
            batch.run_device(iq_t, B, pcm, lens, hip_stream=stream.cuda_stream)
        e1.record(stream)
        torch.cuda.synchronize(dev)
        return time.perf_counter() - w0, e0.elapsed_time(e1) / n

    parity = None
    if not args.no_check:
        parity = parity_gate(iq)

    batch.set_timing(False)                                  # no per-launch event pair inside the library
    for _ in range(max(0, args.preheat - args.warmup)):     # clock settling, see --preheat
        step()
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize(dev)
    if dist:
        dist.barrier()
    # the K launches go back to back; one HIP event pair on the launch stream brackets them all
    # (per-launch pairs, here or inside the library, put about 10 us of event handling between launches)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    ev0.record(stream)
    for _ in range(args.steps):
        step()
    ev1.record(stream)
    torch.cuda.synchronize(dev)
    if dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms = ev0.elapsed_time(ev1) / args.steps           # average launch duration over the timed region

    samples_per_step = S * B * (BLOCK_LEN // 2)
    pcm_bytes = int(lens.sum().item()) * 2
    algo_bytes = S * B * BLOCK_LEN + pcm_bytes                 # u8 IQ in + s16 PCM out, per launch

    # ---- gather the per-rank counters over RCCL (no data-path collective) ----
    from rtl_fm_player_amd.shard import gather_counters
    rep = gather_counters(dist, dev, elapsed, samples_per_step * args.steps, int(kernel_ms * 1e6),
                          int(pcm.view(torch.int16).sum().item()), force=args.force_dist)
    total_samples, elapsed = rep["samples"], rep["elapsed_s"]

    # ---- extra legs (never `value`): a sustained run, the worst-case and the quiet inputs, the other modes ----
    sustained = noise_leg = quiet_leg = modes_leg = None

    def input_leg(iq_t, what):
        """parity gate on EVERY stream, then the same preheat the timed workload got, then --steps timed launches"""
        par = None if args.no_check else parity_gate(iq_t)
        timed_launches(iq_t, max(args.preheat, args.warmup, 2))
        wall, k_ms = timed_launches(iq_t, args.steps)
        batch.reset()
        return {"data": what, "steps": args.steps, "untimed_launches_before_timing": max(args.preheat, args.warmup, 2),
                "ms_per_step": round(wall / args.steps * 1e3, 4), "kernel_ms": round(k_ms, 4),
                "value": round(samples_per_step * args.steps / wall / 1e6, 1), "parity": par,
                "slowdown_vs_timed_input": round(k_ms / kernel_ms, 3)}

    if not args.no_extra and not args.dry_run:
        n_sus = max(args.steps, int(math.ceil(args.sustained_seconds / (kernel_ms * 1e-3))))
        watch = PowerWatch(torch.cuda.get_device_properties(dev))   # package power and shader clock (hwmon) while the leg runs
        watch.start()
        wall, k_ms = timed_launches(iq, n_sus)
        power = watch.stop()
        sustained = {"launches": n_sus, "seconds": round(wall, 3), "ms_per_step": round(wall / n_sus * 1e3, 4),
                     "kernel_ms": round(k_ms, 4), "value": round(samples_per_step * n_sus / wall / 1e6, 1),
                     "frac": round(algo_bytes / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                     "note": "back-to-back launches of the timed workload right after the timed region; whole-rank rate"}
        if power:
            # what bounds this kernel: the package sits at its power cap and the shader clock is what the cap leaves
            sustained["power"] = power
        if args.data != "noise" and not args.only_sustained:
            # uniform random bytes = a pilot-less, full-band input: the ill-conditioned-sample redo paths of the +-1 LSB
            # kernels (branch cut / origin of the discriminator, carrier regeneration without a pilot) run here
            gn = torch.Generator(device=dev)
            gn.manual_seed(54321 + rank)
            iq_n = torch.randint(0, 256, (S, B, BLOCK_LEN), dtype=torch.uint8, device=dev, generator=gn)
            torch.cuda.synchronize(dev)
            noise_leg = input_leg(iq_n, "synthetic uniform random bytes")
            # quiet input: bytes in {127, 128} (a dongle without an antenna) with the reference's own mute fill - 4096 bytes of 127
            # at the head of every fourth block (src/rtl_fm_player.c:805-810) - and one stream in eight all-127: every decimated
            # sample lies next to the origin, where the reference's own rounding decides the phase and the +-1 LSB kernels
            # fall back to its arithmetic tile by tile
            iq_n.copy_(torch.randint(127, 129, (S, B, BLOCK_LEN), dtype=torch.uint8, device=dev, generator=gn))
            iq_n[:, ::4, :4096] = 127
            iq_n[::8] = 127
            torch.cuda.synchronize(dev)
            quiet_leg = input_leg(iq_n, "synthetic quiet input: bytes in {127, 128}, 4096-byte mute fills of 127, one stream in eight constant 127")
            del iq_n
        if world == 1 and args.mode == "stereo" and args.math == "fast" and not args.only_sustained:
            # the other modes of the path on this device, same shape of run (BASELINE.json configs[0] / [4] at 256 streams): never `value`
            modes_leg = {}
            for mname, mkw, rate in (("mono", dict(rate_in=300000, rate_out2=48000, mode=1), 2.4e6),
                                     ("nfm", dict(rate_in=25000, rate_out2=12500, mode=1), 200e3)):
                mb = R.BatchDemod(R.wbfm_config(block_len=BLOCK_LEN, math=math_code, **mkw), S, device=local)
                miq = synth_fm_iq(torch, dev, S, B * BLOCK_LEN // 2, rate, mname != "nfm", 12345 + rank).view(S, B, BLOCK_LEN)
                mpcm = torch.zeros((S, B, mb.pcm_stride), dtype=torch.int16, device=dev)
                mlens = torch.zeros((S, B), dtype=torch.int32, device=dev)
                torch.cuda.synchronize(dev)                 # the launches go to `stream`, torch made these on its own

                def mode_measure(t_iq):
                    """parity of EVERY stream against the oracle, the timed workload's preheat, then --steps timed launches of this mode's batch"""
                    par = None
                    if not args.no_check:
                        from concurrent.futures import ThreadPoolExecutor
                        from oracle import OracleStream
                        mb.reset()
                        mb.run_device(t_iq, B, mpcm, mlens, hip_stream=stream.cuda_stream)
                        torch.cuda.synchronize(dev)
                        h_iq, h_pcm, h_lens = t_iq.cpu().numpy(), mpcm.cpu().numpy(), mlens.cpu().numpy()

                        def mcheck(si):
                            want, wl = OracleStream(**mkw).run(h_iq[si].reshape(-1), BLOCK_LEN)
                            if not np.array_equal(h_lens[si], wl):
                                return 1 << 20
                            got = np.concatenate([h_pcm[si, b, :wl[b]] for b in range(B)])
                            return int(np.abs(got.astype(np.int32) - want.astype(np.int32)).max()) if got.size else 0

                        with ThreadPoolExecutor(max(1, min(usable_cores(), 32))) as ex:
                            worst = max(ex.map(mcheck, range(S)))
                        assert worst <= 1, "%s: PCM differs from the CPU oracle by %d LSB" % (mname, worst)
                        par = {"max_abs_lsb": worst, "tolerance_lsb": 1, "streams_checked": S}
                        mb.reset()
                    mb.set_timing(False)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    for _ in range(max(args.preheat, args.warmup, 2)):
                        mb.run_device(t_iq, B, mpcm, mlens, hip_stream=stream.cuda_stream)
                    torch.cuda.synchronize(dev)
                    w0 = time.perf_counter()
                    e0.record(stream)
                    for _ in range(args.steps):
                        mb.run_device(t_iq, B, mpcm, mlens, hip_stream=stream.cuda_stream)
                    e1.record(stream)
                    torch.cuda.synchronize(dev)
                    return par, time.perf_counter() - w0, e0.elapsed_time(e1) / args.steps

                mpar, mwall, mk = mode_measure(miq)
                mbytes = S * B * BLOCK_LEN + int(mlens.sum().item()) * 2
                modes_leg[mname] = {
                    "workload": "%d concurrent %s streams per GPU x %d blocks" % (S, "2.4 Msps mono WBFM" if mname == "mono" else "200 ksps narrow-FM mono", B),
                    "kernel_family": {R.MATH_FAST_VALU: "fast-valu", R.MATH_FAST_MFMA: "fast-mfma", R.MATH_FAST_MFMA_F: "fast-mfma-f"}.get(mb.math, str(mb.math)),
                    "steps": args.steps, "kernel_ms": round(mk, 4), "ms_per_step": round(mwall / args.steps * 1e3, 4),
                    "value": round(samples_per_step * args.steps / mwall / 1e6, 1), "unit": "Msamples/s",
                    "algorithmic_bytes_per_launch": mbytes, "achieved_gbs": round(mbytes / (mk * 1e-3) / 1e9, 1),
                    "frac": round(mbytes / (mk * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "parity": mpar}
                # ... and this mode on the quiet input of the stereo leg above (the driver's line carries it for every mode)
                gq = torch.Generator(device=dev)
                gq.manual_seed(97531 + rank)
                miq.copy_(torch.randint(127, 129, (S, B, BLOCK_LEN), dtype=torch.uint8, device=dev, generator=gq))
                miq[:, ::4, :4096] = 127
                miq[::8] = 127
                torch.cuda.synchronize(dev)
                qpar, qwall, qk = mode_measure(miq)
                modes_leg[mname]["quiet_input"] = {
                    "data": "synthetic quiet input: bytes in {127, 128}, 4096-byte mute fills of 127, one stream in eight constant 127",
                    "steps": args.steps, "kernel_ms": round(qk, 4), "ms_per_step": round(qwall / args.steps * 1e3, 4),
                    "value": round(samples_per_step * args.steps / qwall / 1e6, 1), "parity": qpar,
                    "slowdown_vs_timed_input": round(qk / mk, 3)}
                del mb, miq, mpcm, mlens

    if rank == 0:
        copy_gbs = copy_bandwidth(torch, dev, stream) if S * B * BLOCK_LEN <= (1 << 31) else 0.0
        value = total_samples / elapsed / 1e6
        achieved = algo_bytes / (kernel_ms * 1e-3) / 1e9
        out = {
            "metric": "IQ Msamples/s through full_demod",
            "value": round(value, 1),
            "unit": "Msamples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": ("f32" if batch.math in (R.MATH_EXACT, R.MATH_FAST_VALU) else
                      "f32 (stage A: int8-limb fixed point, exact sums)" if batch.math == R.MATH_FAST_MFMA else
                      "f32 (stages A, C, D: int8-limb fixed point, exact sums)" if stereo else "f32 (stages A, D: int8-limb fixed point, exact sums)"),
            "data": ("synthetic FM broadcast per stream (stereo multiplex: 19 kHz pilot + L-R DSB, tones, +-75 kHz)"
                     if args.data == "fm" and args.mode != "nfm" else
                     "synthetic narrow FM per stream (tone, +-5 kHz)" if args.data == "fm" else
                     "synthetic uniform random bytes"),
            "config": {
                "workload": ("%d concurrent 2.4 Msps %s WBFM streams per GPU x %d blocks of %d B u8 IQ "
                             "(rate_in 300k -> 48k PCM), IQ resident in HBM" % (S, args.mode, B, BLOCK_LEN))
                if args.mode != "nfm" else
                ("%d concurrent 200 ksps narrow-FM mono streams per GPU x %d blocks of %d B u8 IQ "
                 "(rate_in 25k -> 12.5k PCM), IQ resident in HBM" % (S, B, BLOCK_LEN)),
                "streams_per_gpu": S, "blocks_per_step": B, "math": args.math,
                # what ran: the kernel family FMD_MATH_FAST resolved to, and which stages used the matrix pipe
                "kernel_family": {R.MATH_EXACT: "exact", R.MATH_FAST_VALU: "fast-valu", R.MATH_FAST_MFMA: "fast-mfma", R.MATH_FAST_MFMA_F: "fast-mfma-f"}.get(batch.math, str(batch.math)),
                "mfma": batch.math in (R.MATH_FAST_MFMA, R.MATH_FAST_MFMA_F),
                "mfma_stages": {R.MATH_FAST_MFMA: "A (i8 decimator)",
                                R.MATH_FAST_MFMA_F: ("A (i8 decimator) + C (i8 pilot and L-R filters) + D (i8 decimating second stage at the emit instants: the composite L+R filter fm * fm and fm over (L-R) x carrier)"
                                                     if stereo else "A (i8 decimator) + D (i8 decimating low-pass at the emit instants)")}.get(batch.math, "none"),
                "sharding": "streams/%d" % world, "kernel": batch.kernel_name(),
                "untimed_launches_before_timing": max(args.preheat, args.warmup),
            },
            "roofline": {
                "bound": "hbm",
                "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": None,
                # what the CU's units were doing (counter-derived, filled in below where a committed profile matches this workload)
                "busy": None,
                "copy_kernel_gbs": round(copy_gbs, 1),           # measured d2d copy on this device (read + write)
                "frac_of_copy_kernel": round(achieved / copy_gbs, 4) if copy_gbs > 0 else None,
                "kernel_ms": round(kernel_ms, 4),
                "algorithmic_bytes_per_launch": algo_bytes,
                "bytes_per_sample": round(algo_bytes / samples_per_step, 4),
            },
        }
        tr = measured_traffic(out["config"])
        if tr:
            out["roofline"]["traffic"] = tr[0]
            out["roofline"]["traffic_source"] = "profiles/" + tr[1]
            out["roofline"]["busy"] = measured_busy(tr[1])
        if parity:
            out["parity"] = parity
        if sustained:
            out["sustained"] = sustained
            pw = sustained.get("power")
            if pw and pw.get("cap_w"):
                # The energy books of one launch: what the package drew while the sustained leg ran x the kernel's time there, split
                # into the base draw (every SIMD clocked and issuing s_nop: POWER_BASE_W, tools/ubench/power_price.hip), the bytes moved (NJ_PER_BYTE from
                # a device-to-device copy, tools/stream_power.py, x the traffic of the committed profile or the algorithmic bytes) and the arithmetic
                # (the rest).  bound_ms: the launch's energy at the cap - the time it would take if the cap were the only limit.
                t_s = sustained["kernel_ms"] * 1e-3
                joules = pw["package_w_mean"] * t_s
                moved = out["roofline"]["traffic"] or algo_bytes
                out["roofline"]["power"] = {
                    "cap_w": pw["cap_w"], "package_w": pw["package_w_mean"], "sclk_mhz": pw["sclk_mhz_mean"], "base_w": POWER_BASE_W, "nJ_per_byte": NJ_PER_BYTE,
                    "J_per_launch": round(joules, 4), "base_J": round(POWER_BASE_W * t_s, 4), "bytes_J": round(NJ_PER_BYTE * 1e-9 * moved, 4),
                    "arithmetic_J": round(joules - POWER_BASE_W * t_s - NJ_PER_BYTE * 1e-9 * moved, 4),
                    "bound_ms": round(joules / pw["cap_w"] * 1e3, 4), "kernel_ms": sustained["kernel_ms"],
                    "note": "sustained leg; bound_ms = J_per_launch / cap_w; the kernel sits at the cap when package_w ~ cap_w and sclk below its 2400 MHz maximum"}
        if noise_leg:
            out["noise_input"] = noise_leg
        if quiet_leg:
            out["quiet_input"] = quiet_leg
        if modes_leg:
            out["modes"] = modes_leg
        out["per_rank"] = rep["per_rank"]                      # samples, kernel ns per launch, PCM checksum of each rank
        if rep.get("backend"):
            out["counters_gathered_over"] = rep["backend"]     # "nccl" (= RCCL): all_reduce(MAX) + all_gather ran
        if not args.no_cpu and world == 1:
            out["cpu_baseline"] = cpu_baseline(cfg_kw, args.cpu_seconds, iq)
        elif not args.no_cpu:
            out["cpu_baseline"] = None
        if not args.no_extra and not args.only_sustained and world == 1:
            out["single_stream"] = single_stream_leg(R, cfg_kw, math_code)
        if not args.no_e2e and world == 1:
            out["e2e_h2d"] = e2e_h2d(args.mode, min(args.e2e_streams, S), B, args.e2e_jobs, math_code, cfg_kw)
        print(json.dumps(out), flush=True)
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
