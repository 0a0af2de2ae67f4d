"""PCIe-inclusive rate of the host-pointer entry points (what a Vec<Complex<f32>> node pays): the FIR node and the fused
metric chain over 2^24 samples, single shot (COMMS_HOST_PIPE_BYTES=0) against the chunked pipeline at several chunk sizes.
Each setting runs in its own process (the limits are read once).  usage: python scripts/bench_host_path.py"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import sys, time; sys.path.insert(0, %r)
import ctypes as C, numpy as np, comms_rs_amd as c
from comms_rs_amd._lib import check
n = 1 << 24
x = c.synth_iq(n)
taps = c.rrc_taps(255, 8.0, 0.35)
fir = c.BatchFirNode(taps)
chain = c.ChainNode(0.6283, 0.0, taps, 8, False, mixer_after_fir=True)
y = np.zeros(n, np.complex64); z = np.zeros(n // 8, np.complex64)
vp = lambda a: a.ctypes.data_as(C.c_void_p)
def best(fn):
    ts = []
    for _ in range(6):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return min(ts)
tf = best(lambda: check(c.lib().comms_fir_run(fir._h, vp(x), n, vp(y))))
tc = best(lambda: check(c.lib().comms_chain_run(chain._h, vp(x), n, vp(z))))
tfresh = best(lambda: fir.run(x))
print("%%-34s FIR %%6.2f ms = %%5.2f Gsamples/s (%%4.1f GB/s each way; fresh output array per call: %%5.2f Gsamples/s) | fused chain %%5.2f ms = %%5.2f Gsamples/s" %% (sys.argv[1], tf * 1e3, n / tf / 1e9, 8 * n / tf / 1e9, n / tfresh / 1e9, tc * 1e3, n / tc / 1e9))
''' % ROOT
for name, env in [("single shot", {"COMMS_HOST_PIPE_BYTES": "0"})] + [("pipelined, chunks of %d MiB" % mb, {"COMMS_HOST_CHUNK_BYTES": str(mb << 20)}) for mb in (2, 4, 8, 16, 32)] + [("single shot", {"COMMS_HOST_PIPE_BYTES": "0"})]:
    r = subprocess.run([sys.executable, "-c", CODE, name], env=dict(os.environ, **env), capture_output=True, text=True)
    print(r.stdout.strip() or r.stderr[-500:], flush=True)
