"""Data-parallel exchange for the chunk VQ-VAE step: one process per GPU, `torch.distributed` (backend "nccl" = RCCL
over xGMI on MI355X; "gloo" in the CPU tests).

The reference has no distributed code at all.  The step shards over the batch with exactly ONE exchange:
a SUM all-reduce of the engine's contiguous communication buffer

        comm = [ flat parameter gradients | cnt (K) | dw (K*E) | fault flag (4 floats, one used) ]

after the local backward.  Gradients are then averaged inside the fused clip+Adam kernel (grad_scale = 1/world, so the
clip sees the GLOBAL gradient norm, like clip_grad_norm_ after DDP averaging) and the EMA codebook update consumes the
GLOBAL assignment statistics, so every rank applies the identical update and the replicas never diverge.  BatchNorm
batch statistics stay per rank (north star: "all-reduce on gradients and on codebook EMA statistics only").  The last slot
carries each rank's fault latch of the persistent rollouts (0 / 1): a non-zero sum makes EVERY rank skip the step.

Message size at the BASELINE shape: 188,700 grads + 512 + 65,536 stats floats = 1.0 MB: latency-bound, so a single
fused collective per step is the right shape for the 7-link xGMI mesh (no bucketing, no overlap machinery)."""
from __future__ import annotations

from typing import Dict, Iterable, Optional

import torch
import torch.distributed as dist


class GradStatsAllReduce:
    """reduce_fn for VQVAEEngine.train_step: SUM all-reduce of the comm buffer, in place, on the current stream."""

    def __init__(self, group: Optional[dist.ProcessGroup] = None):
        self.group = group
        self.world = dist.get_world_size(group)

    def __call__(self, comm: torch.Tensor) -> torch.Tensor:
        dist.all_reduce(comm, op=dist.ReduceOp.SUM, group=self.group)
        return comm


def broadcast_state(tensors: Iterable[torch.Tensor], src: int = 0, group: Optional[dist.ProcessGroup] = None) -> None:
    """Make every rank start from rank `src`'s weights / codebook / EMA buffers (in place)."""
    for t in tensors:
        dist.broadcast(t, src=src, group=group)


def pack_comm(layout, offsets: Dict[str, tuple], n_flat: int, grads: Dict[str, torch.Tensor], cnt: torch.Tensor,
              dw: torch.Tensor) -> torch.Tensor:
    """Host-side mirror of the engine's comm layout (used by the CPU tests and by checkpoint tooling)."""
    K, E = dw.shape
    comm = torch.zeros(n_flat + K + K * E + 4, dtype=torch.float32)      # (+ the fault-flag slot, zero: no fault)
    for name, _ in layout:
        off, n, _shape = offsets[name]
        comm[off:off + n] = grads[name].reshape(-1)
    comm[n_flat:n_flat + K] = cnt
    comm[n_flat + K:n_flat + K + K * E] = dw.reshape(-1)
    return comm


def unpack_comm(layout, offsets: Dict[str, tuple], n_flat: int, comm: torch.Tensor, K: int, E: int):
    grads = {}
    for name, _ in layout:
        off, n, shape = offsets[name]
        grads[name] = comm[off:off + n].view(shape)
    return grads, comm[n_flat:n_flat + K], comm[n_flat + K:n_flat + K + K * E].view(K, E)


def flat_offsets(layout):
    """Same rule as VQVAEEngine.__init__: tensors in layout order, each padded to a multiple of 4 floats."""
    offsets, off = {}, 0
    for name, shp in layout:
        n = 1
        for s in shp:
            n *= s
        offsets[name] = (off, n, shp)
        off += (n + 3) // 4 * 4
    return offsets, off
