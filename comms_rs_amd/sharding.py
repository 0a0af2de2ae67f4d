"""Stream sharding for multi-GPU runs: contiguous shards + one halo hand-over.

A long IQ stream is cut into `world` contiguous shards, one per rank (one rank
per GPU).  A FIR of N taps needs the N-1 samples that precede its shard (the
reference's `state`, fir_node.rs:193-211): rank r-1 hands its last N samples to
rank r once, before streaming starts.  That send/recv pair is the only
communication on the path -- there is no collective in the data path
(SURVEY.md section 8e).  Works on any torch.distributed backend: "nccl" (= RCCL
over xGMI on MI355X) with device tensors, "gloo" with CPU tensors in the tests.
"""


def shard_range(total, world, rank):
    """[start, stop) of `rank`'s contiguous shard of a `total`-sample stream."""
    if not (0 <= rank < world):
        raise ValueError("rank %d outside world %d" % (rank, world))
    base, rem = divmod(total, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


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
    return halo_samples[::-1].copy()


def shard_mixer_phase(phase0, dphase, first_index):
    """Mixer start phase of the shard that begins at stream sample `first_index`:
    (phase0 + first_index * dphase) mod 2*pi, in extended precision -- the closed form of the
    reference's per-sample `phase += dphase` with wrap (src/mixer.rs:79-82), so every rank's
    MixerNode continues the un-sharded oscillator without any communication."""
    import numpy as np

    two_pi = np.longdouble(2.0) * np.longdouble(np.pi)
    ph = np.fmod(np.longdouble(phase0) + np.longdouble(first_index) * np.longdouble(dphase), two_pi)
    return float(ph + two_pi if ph < 0 else ph)
