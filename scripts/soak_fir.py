"""Determinism soak: the same batch through the FIR kernels thousands of times, every output compared bit for bit
with the first (catches intermittent faults, e.g. a missed hazard wait state in the cross-lane radix-4)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import comms_rs_amd as c
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
s = torch.cuda.current_stream().cuda_stream
for n_taps, logn, algo in ((255, 24, c.FIR_OS1024), (63, 23, c.FIR_OS1024), (161, 20, c.FIR_OS1024), (4097, 22, c.FIR_OS16K)):
    n = 1 << logn
    x = torch.empty(n, dtype=torch.complex64, device="cuda:0"); c.synth_iq_dev(x.data_ptr(), n, 0, 5)
    taps = (c.rrc_taps(n_taps, 8.0, 0.35) * np.exp(0.1j * np.arange(n_taps))).astype(np.complex64)
    ref, y = torch.empty_like(x), torch.empty_like(x)
    node = c.BatchFirNode(taps).set_algo(algo)
    state0 = np.zeros(n_taps, np.complex64)
    node.run_dev(x.data_ptr(), n, ref.data_ptr(), s)
    bad = 0
    for i in range(reps):
        node.set_state(state0)
        node.run_dev(x.data_ptr(), n, y.data_ptr(), s)
        if not torch.equal(torch.view_as_real(y), torch.view_as_real(ref)):
            bad += 1
    print("%s taps=%d n=2^%d: %d of %d runs differ from the first" % (node.kernel_for(n), n_taps, logn, bad, reps), flush=True)
    assert bad == 0
