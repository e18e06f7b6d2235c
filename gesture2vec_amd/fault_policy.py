"""What happens to the persistent / cluster kernels' fast path after a residency fault, in ONE place (round 6).

The persistent rollouts and the cluster kernels (include/g2v.h: g2v_dec_rollout_set_persistent, g2v_gru_seq_set_cluster) spin on
their peer workgroups; a bounded wait that runs out latches a device-side fault word, every kernel that COMMITS a step to the
model state reads that word and leaves the state alone, and the trainers repeat the iteration on the per-step kernels.  Until
round 5 that switch was final for the process: a multi-day run that hiccuped once (another tenant on the device for a minute)
trained at per-step-kernel speed for the rest of its life.  Now the fast path is RE-ARMED after `rearm_after` fault-free
iterations on the per-step kernels, at most `max_rearms` times per process (a device that keeps faulting stays on the slow path).

The reference has no analogue (it has no persistent kernels); the policy is the build's own and changes no result: both kernel
families compute the same iteration (tests/test_gpu_ops.py: cluster == per-step), and a faulted iteration is never applied.
"""
from __future__ import annotations

import logging
import threading

from . import _lib


class PersistentPathPolicy:
    """ctx: the gesture2vec_amd._lib.Context whose switches this policy flips (an engine's own); None = the process's default
    context (the module-level trainers of Part d)."""

    def __init__(self, rearm_after: int = 1000, max_rearms: int = 3, ctx=None):
        self.ctx = ctx
        self.rearm_after = int(rearm_after)
        self.max_rearms = int(max_rearms)
        self.faults = 0            # faults seen by this process
        self.rearms = 0            # times the fast path was switched back on
        self.clean = 0             # fault-free iterations since the last fault
        self.off = False           # the fast path is off BECAUSE OF A FAULT (not because a caller chose the per-step kernels)
        self.generation = 0        # bumped whenever the selected kernel family changes: holders of captured graphs compare it
        self._lock = threading.Lock()

    def _switch(self, on: int) -> None:
        if self.ctx is not None:
            self.ctx.set(_lib.OPT_PERSISTENT, on)
            self.ctx.set(_lib.OPT_GRU_CLUSTER, on)
        else:
            lib = _lib.load()
            lib.g2v_dec_rollout_set_persistent(on)
            lib.g2v_gru_seq_set_cluster(on)

    def on_fault(self) -> None:
        """a trainer found the latch set: clear it, select the per-step kernels (the caller repeats the iteration)"""
        lib = _lib.load()
        with self._lock:
            lib.g2v_dec_rollout_persist_fault(1)
            self._switch(0)
            self.faults += 1
            self.clean = 0
            self.off = True
            self.generation += 1

    def tick(self) -> bool:
        """a trainer finished a fault-free iteration.  True: the fast path has just been re-armed -- drop captured graphs and
        cached launch plans (they were built for the per-step kernels) before the next iteration."""
        if not self.off:
            return False
        with self._lock:
            self.clean += 1
            if self.clean < self.rearm_after or self.rearms >= self.max_rearms:
                return False
            self._switch(1)
            self.rearms += 1
            self.clean = 0
            self.off = False
            self.generation += 1
        logging.warning("persistent kernels: %d fault-free iterations on the per-step kernels -- fast path re-armed (%d of %d)",
                        self.rearm_after, self.rearms, self.max_rearms)
        return True


POLICY = PersistentPathPolicy()
