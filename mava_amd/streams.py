"""Side streams that really run beside the launch stream.

HIP maps a process's streams onto a few hardware queues (round-robin as they are created): a new stream may land on the queue
of the stream it is meant to run beside, and its kernels then serialise behind that stream's however independent they are -
observed in bench.py, where the learner of a secondary workload is created after other learners' streams exist and its
"concurrent" critic chain ran strictly after the actor's.  `overlapping_stream` therefore PROBES: two spin kernels, one per
stream, must take the time of one; candidates that take the time of two are passed over."""
from __future__ import annotations

from typing import List, Optional

import torch

_SPIN_CYCLES = 1_500_000  # ~0.7 ms per probe kernel
_kept: List[torch.cuda.Stream] = []  # rejected candidates stay referenced, so the next candidate is another pool stream


def _pair_ms(main: torch.cuda.Stream, side: Optional[torch.cuda.Stream], device) -> float:
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(device)
    if side is not None:
        side.wait_stream(main)
    a.record(main)
    torch.cuda._sleep(_SPIN_CYCLES)
    if side is not None:
        with torch.cuda.stream(side):
            torch.cuda._sleep(_SPIN_CYCLES)
        main.wait_stream(side)
    b.record(main)
    b.synchronize()
    return a.elapsed_time(b)


def overlapping_stream(device, tries: int = 8) -> Optional[torch.cuda.Stream]:
    """A new stream on `device` whose kernels were seen to run concurrently with the current stream's, or None (the caller
    then keeps everything on the one stream)."""
    device = torch.device(device)
    if device.type != "cuda":
        return None
    with torch.cuda.device(device):
        main = torch.cuda.current_stream(device)
        try:
            _pair_ms(main, None, device)  # (clock ramp, lazy initialisation)
            single = _pair_ms(main, None, device)
            for _ in range(tries):
                cand = torch.cuda.Stream(device=device)
                if _pair_ms(main, cand, device) < 1.5 * single:
                    return cand
                _kept.append(cand)
        except (RuntimeError, AttributeError):  # no probe kernel in this torch build: take the stream unprobed
            return torch.cuda.Stream(device=device)
    return None
