"""step() loops recorded into a hipGraph, with the status word read behind every replay.

A recorded ``w2a_step`` is replayed without any of the host-side bookkeeping of ``csrc/w2a_bookkeeping.h``. The library keeps
the form of the state the recorded kernel steps current at the end of every API call; where it cannot (a masked reset or a
restored checkpoint took a packed batch out of lock step) it marks the mirror stale ON THE DEVICE and a replayed packed step
does nothing but raise ``W2A_ST_STALE_GRAPH`` (include/w2a.h). A loop that never reads the status word would then train on
frozen observations. ``RecordedSteps`` makes that impossible to miss without a synchronisation per replay: each replay is
followed by an asynchronous copy of the status word into pinned host memory, and replay k raises what replay k-1 left
there -- the host stays one replay ahead of the device, never more, so the device is never idle.

(The reference has no counterpart: its loop is Python, env.py:265-277.)
"""
from __future__ import annotations

import torch


class RecordedSteps:
    """``days`` calls of ``one_day()`` -- a policy on device tensors followed by ``env.step(actions)`` -- as one hipGraph.

    env       HeatAlertVecEnv with lockstep=False (episodes restart inside the step kernel) or autoreset="disabled"; a loop
              whose episode boundaries the host drives cannot be recorded (step() raises while capturing)
    warmup    run one_day() once before the capture (every capture needs one; it also puts a lock-step batch of
              >= 131 072 envs on its packed 16-B form, which the recorded steps then keep)
    """

    def __init__(self, env, one_day, days: int, warmup: bool = True):
        if days <= 0:
            raise ValueError("days must be positive")
        self.env, self.days = env, int(days)
        self.replays = 0
        with torch.cuda.device(env.device):
            if warmup:
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    one_day()
                torch.cuda.current_stream().wait_stream(side)
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                for _ in range(self.days):
                    one_day()
            self._host = torch.zeros(2, dtype=torch.int32).pin_memory()
            self._events = [torch.cuda.Event(), torch.cuda.Event()]
        self._pending = [False, False]

    def _collect(self, slot: int):
        if not self._pending[slot]:
            return
        self._events[slot].synchronize()
        self._pending[slot] = False
        bits = int(self._host[slot])
        if bits:
            self.env.check_status()  # reads and CLEARS the device word, raises the same errors as everywhere else

    def replay(self):
        """Launch the recorded block; raises what the PREVIOUS replay left in the status word."""
        slot = self.replays & 1
        self._collect(slot)  # (two replays ago: long complete)
        with torch.cuda.device(self.env.device):
            self.graph.replay()
            self._host[slot:slot + 1].copy_(self.env.status_word, non_blocking=True)
            self._events[slot].record()
        self._pending[slot] = True
        self.replays += 1
        self._collect(slot ^ 1)  # the previous replay: the host waits here while this one is queued behind it

    def finish(self):
        """Wait for the last replay and raise what it left in the status word."""
        for slot in (self.replays & 1, (self.replays & 1) ^ 1):
            self._collect(slot)
