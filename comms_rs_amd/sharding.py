"""Stream sharding for multi-GPU runs: contiguous shards + one hand-over per neighbour pair.

A long IQ stream is cut into `world` contiguous shards, one per rank (one rank per GPU).
What a shard needs from its left neighbour, per node:

  FIR / pulse (N taps)   the N-1 samples before the shard = the reference's `state`
                         (fir_node.rs:193-211)                      -> state_from_halo + set_state
  mixer                  nothing: closed-form start phase           -> shard_mixer_phase
  decimate / upsample    nothing when shard starts are multiples of the rate (and of the
                         reference's batch length)
  FM demod               FM.prev = the sample before the shard (analog.rs:31): one sample
                         of the *demodulator's input*, i.e. of the decimated filter output
                         in a chain                                  -> prime_chain
  FFT batches            nothing (independent transforms)            -> shard_range over transforms

`prime_chain` is the general hand-over: the rank runs its own node(s) over a short prefix of
raw samples that precede its shard and throws the outputs away; afterwards the FIR history, the
oscillator phase and FM.prev are exactly what the un-sharded node would hold at the shard
boundary, and no rank ever waits for a neighbour's *results* -- only for a few hundred raw
input samples.  That send/recv pair (plus the optional scatter / gather of whole shards, the
"secondary" variant of SURVEY 8e) is the only communication on the path: there is no collective
in the data path.  Works on any torch.distributed backend: "nccl" (= RCCL over xGMI on MI355X)
with device tensors, "gloo" with CPU tensors in the tests and rehearsals.
"""


# The arithmetic lives in the C ABI (csrc/shard.cpp: comms_shard_range, comms_state_from_halo,
# comms_shard_mixer_phase, comms_chain_prefix_len) so that C++ / Rust hosts cut streams the same way; the functions
# here are thin bindings, plus the torch.distributed messages.
def shard_range(total, world, rank):
    """[start, stop) of `rank`'s contiguous shard of a `total`-sample stream."""
    import ctypes as C

    from ._lib import CommsError, check, lib

    if not (0 <= rank < world):
        raise ValueError("rank %d outside world %d" % (rank, world))
    a, b = C.c_size_t(), C.c_size_t()
    check(lib().comms_shard_range(total, world, rank, C.byref(a), C.byref(b)))
    return a.value, b.value


def halo_exchange(dist, tail, rank, world):
    """Ring-neighbour hand-over: every rank sends `tail` (its last N samples, a
    real-view tensor on the right device) to rank+1 and returns what rank-1 sent,
    or None on rank 0.  One P2P message per neighbour pair."""
    import torch

    halo = torch.zeros_like(tail) if rank > 0 else None
    ops = []
    if rank + 1 < world:
        ops.append(dist.P2POp(dist.isend, tail, rank + 1))
    if rank > 0:
        ops.append(dist.P2POp(dist.irecv, halo, rank - 1))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    return halo


def state_from_halo(halo_samples):
    """Reference `state` layout (newest first) from time-ordered halo samples."""
    import numpy as np

    from ._lib import check, lib

    h = np.ascontiguousarray(halo_samples, dtype=np.complex64)
    out = np.empty_like(h)
    check(lib().comms_state_from_halo(h.ctypes.data, h.size, out.ctypes.data))
    return out


def shard_mixer_phase(phase0, dphase, first_index):
    """Mixer start phase of the shard that begins at stream sample `first_index`:
    (phase0 + first_index * dphase) mod 2*pi, in extended precision -- the closed form of the
    reference's per-sample `phase += dphase` with wrap (src/mixer.rs:79-82), so every rank's
    MixerNode continues the un-sharded oscillator without any communication.  `first_index`
    may be negative (the prefix of prime_chain starts before the shard)."""
    import ctypes as C

    from ._lib import check, lib

    out = C.c_double()
    check(lib().comms_shard_mixer_phase(float(phase0), float(dphase), int(first_index), C.byref(out)))
    return out.value


def chain_prefix_len(n_taps, rate, fm_demod):
    """Raw samples a chain shard needs from its left neighbour, a multiple of `rate`.
    Without FM demod the FIR history is enough (n_taps - 1, rounded up).  With FM demod the
    shard must also reproduce the last decimated filter output before the boundary -- the
    output at input index (boundary - rate), which looks back another n_taps - 1 samples."""
    import ctypes as C

    from ._lib import check, lib

    out = C.c_size_t()
    check(lib().comms_chain_prefix_len(n_taps, rate, 1 if fm_demod else 0, C.byref(out)))
    return out.value


def prime_chain(chain, prefix_ptr, prefix_len, discard_ptr, stream=0):
    """Run `chain` (created with the oscillator phase of the prefix's first sample, see
    shard_mixer_phase) over the `prefix_len` raw samples that precede its shard, outputs into
    `discard_ptr` (prefix_len / rate elements, thrown away).  Afterwards the chain's FIR
    history, oscillator phase and FM.prev continue the un-sharded stream."""
    if prefix_len:
        chain.run_dev(prefix_ptr, prefix_len, discard_ptr, stream)


def scatter_shards(dist, full, shard, rank, world, root=0):
    """Secondary variant of SURVEY 8e ("includes transfer"): the root holds the whole stream
    (real-view tensor, `world` equal shards back to back) and sends shard r to rank r --
    `world - 1` point-to-point messages that leave the root on distinct xGMI links under
    RCCL; the root keeps its own shard by a local copy.  `shard` receives this rank's part."""
    per = shard.shape[0]
    if rank == root:
        shard.copy_(full[root * per:(root + 1) * per])
        ops = [dist.P2POp(dist.isend, full[r * per:(r + 1) * per], r) for r in range(world) if r != root]
    else:
        ops = [dist.P2POp(dist.irecv, shard, root)]
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    return shard


def gather_shards(dist, out_shard, full_out, rank, world, root=0):
    """The way back: every rank's output shard to the root, in rank order."""
    per = out_shard.shape[0]
    if rank == root:
        full_out[root * per:(root + 1) * per].copy_(out_shard)
        ops = [dist.P2POp(dist.irecv, full_out[r * per:(r + 1) * per], r) for r in range(world) if r != root]
    else:
        ops = [dist.P2POp(dist.isend, out_shard, root)]
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    return full_out
