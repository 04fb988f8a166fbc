"""GPU-resident circular replay buffer (host-side bookkeeping; the data lives in HBM).

Reference: `memory = CircularBuffer{Any}(MEM_SIZE)` of `[s, a, r, s', done]` (input.jl:139-140),
`remember` (memory_plotting_saving.jl:46-47), `getData` (:31-42).  Here: struct-of-arrays device
tensors (PyTorch is only the allocator) + a host push counter; kernels receive the raw pointers
through `shems_replay` (include/shems_hip.h).
"""
from __future__ import annotations

from . import _capi


class ReplayRing:
    def __init__(self, capacity, device=None, tensors=None):
        """tensors = (s, a, r, s2, done): views into memory the caller owns (a learner group's slab) instead of new buffers."""
        import torch
        self.capacity = int(capacity)
        if tensors is not None:
            self.s, self.a, self.r, self.s2, self.done = tensors
            self.device = self.s.device
            assert self.s.shape == (self.capacity, _capi.NSTATE) and self.done.dtype == torch.uint8 and self.done.shape == (self.capacity,)
        else:
            # default = the CURRENT device (one process per GPU: bench.py/run_charger.py call set_device(LOCAL_RANK) first);
            # a ring on another GPU than the env/agent would make every kernel dereference unmapped peer memory
            self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
            self.s = torch.zeros((self.capacity, _capi.NSTATE), dtype=torch.float32, device=self.device)
            self.a = torch.zeros((self.capacity, _capi.NACTION), dtype=torch.float32, device=self.device)
            self.r = torch.zeros((self.capacity,), dtype=torch.float32, device=self.device)
            self.s2 = torch.zeros((self.capacity, _capi.NSTATE), dtype=torch.float32, device=self.device)
            self.done = torch.zeros((self.capacity,), dtype=torch.uint8, device=self.device)
        self.pushed = 0                      # total transitions ever pushed (host state)

    def __len__(self):                       # length(memory)
        return min(self.pushed, self.capacity)

    @property
    def pos(self):
        return self.pushed % self.capacity

    def struct(self):
        return _capi.Replay(self.capacity, self.s.data_ptr(), self.a.data_ptr(), self.r.data_ptr(),
                            self.s2.data_ptr(), self.done.data_ptr())
