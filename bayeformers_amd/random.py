"""Counter-based RNG state of the Monte-Carlo path.

The reference draws eps from torch's global generator in layer-execution order
(/root/reference/bayeformers/nn/parameters/gaussian.py:100).  Here eps is a pure function of
(seed, sample index, 2*layer_id + tensor_id, element index) — csrc/bf_philox.h — so the only state is the seed
and the next unused Monte-Carlo sample index.  Every forward of a bnn.Model (or of a bare bnn.Linear) reserves S
consecutive sample indices; all layers inside one forward share them, which is what makes results independent
of kernel tiling, of S-batching and of how samples are sharded over GPUs.
"""
import threading

import torch

DEFAULT_SEED = 0x5EED

_DTYPES = {
    "bf16": torch.bfloat16, "bfloat16": torch.bfloat16, torch.bfloat16: torch.bfloat16,
    "fp16": torch.float16, "float16": torch.float16, "half": torch.float16, torch.float16: torch.float16,
    "fp32": torch.float32, "float32": torch.float32, torch.float32: torch.float32,
}


class _State(threading.local):
    def __init__(self):
        self.seed = DEFAULT_SEED
        self.next_sample = 0
        self.ctx = None  # (sample_base, S) while a bnn.Model forward is running
        self.compute_dtype = torch.bfloat16
        self.next_layer_id = 0


STATE = _State()


def manual_seed(seed: int, next_sample: int = 0) -> None:
    """Seed the Philox key and rewind the Monte-Carlo sample counter."""
    STATE.seed = int(seed) & (2 ** 64 - 1)
    STATE.next_sample = int(next_sample) & 0xFFFFFFFF


def get_state():
    return STATE.seed, STATE.next_sample


def reserve_samples(n: int) -> int:
    """Reserve n consecutive MC sample indices; returns the first."""
    base = STATE.next_sample
    STATE.next_sample = (base + int(n)) & 0xFFFFFFFF
    return base


def set_compute_dtype(dtype) -> None:
    """MFMA operand precision of the sampled-weight GEMM: 'bf16' (default), 'fp16', or 'fp32' (exact, 1/16 rate)."""
    STATE.compute_dtype = _DTYPES[dtype]


def get_compute_dtype() -> torch.dtype:
    return STATE.compute_dtype


def new_layer_id() -> int:
    i = STATE.next_layer_id
    STATE.next_layer_id += 1
    return i
