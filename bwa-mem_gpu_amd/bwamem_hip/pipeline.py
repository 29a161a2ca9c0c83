"""Host-side pipeline pieces shared by bench.py, the multi-GPU launcher and the tests.

Nothing here computes the hot path: seeding, chaining and extension go through the
C ABI (lib.py -> libbwamem_hip.so).  torch is used for device memory only: the reads
of a batch and the index arrays as HBM tensors.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np
import torch

from . import fmindex


@dataclass
class DeviceReads:
    ascii: torch.Tensor   # uint8 [n_bases]
    offs: torch.Tensor    # int32 [n_reads]
    lens: torch.Tensor    # int32 [n_reads]
    codes: torch.Tensor   # uint8 [n_reads, L] nt4 codes (fixed length batches only)

    @property
    def n(self) -> int:
        return int(self.lens.numel())


def reads_to_device(reads2d: np.ndarray, device) -> DeviceReads:
    n, L = reads2d.shape
    codes = torch.from_numpy(np.ascontiguousarray(reads2d)).to(device)
    lut = torch.tensor(list(b"ACGTN"), dtype=torch.uint8, device=device)
    ascii_ = lut[codes.long()].reshape(-1).contiguous()
    offs = (torch.arange(n, device=device, dtype=torch.int64) * L).to(torch.int32)
    lens = torch.full((n,), L, dtype=torch.int32, device=device)
    return DeviceReads(ascii_, offs, lens, codes)


def index_to_device_tensors(idx: fmindex.FMDIndex, device):
    """Pack the index arrays into HBM tensors, blocks padded to whole 32-byte units."""
    nblk = (idx.seq_len + 63) // 64 + 1
    bwt = torch.zeros(nblk * 8, dtype=torch.int32, device=device)
    w = torch.from_numpy(idx.bwt_words.view(np.int32))
    bwt[: w.numel()] = w.to(device)
    sa = torch.from_numpy(idx.sa.view(np.int32)).to(device)
    bits = torch.from_numpy(idx.sa_bits.view(np.int32)).to(device)
    return bwt, sa, bits
