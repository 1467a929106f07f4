"""Chunked real-time use of the frontend ("very-long-clip time tiling with EMA carry"; not part of the reference
surface -- the reference only ever sees whole clips).

``LeafStream(leaf)`` is fed consecutive chunks of B running waveforms and returns, chunk by chunk, exactly the frames
``leaf`` would produce for the whole recording: a frame is emitted as soon as every sample of its receptive field has
arrived (25 ms after its centre at the default geometry), the waveform history the next frames still need is kept on the
device (2 (K - 1) samples plus alignment), and the PCEN smoother's state travels between calls (``leaf_pcen_stream_f32``).
``flush()`` ends the stream: the remaining frames are produced with the reference's zero padding at the end of the clip.

Every call is two launches of the product kernels: the fused forward without compression on [history | chunk] (the
overlap-save or MFMA path ``LEAF_ALGO_AUTO`` picks for that length), then the stateful PCEN over the new frames.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import _native


class LeafStream:
    def __init__(self, leaf, log1p: bool = False):
        conv, pool = leaf._complex_conv, leaf._pooling
        self.leaf = leaf
        self.K, self.hop, self.F = conv._kernel_size, pool.strides, conv._filters
        self.pad_l = self.K // 2 + self.K % 2 - 1                          # utils.py:5-10
        self.pad_r = self.K // 2
        # a frame m' of the buffer is exact once the buffer starts a whole number of hops, >= 2 pad_l samples, before it (its
        # pooling window reaches pad_l back, the filters another pad_l) ...
        self.lead = -(-2 * self.pad_l // self.hop)                        # frames of lead-in
        # ... and holds every sample up to m' hop + 2 (K - 1 - pad_l)
        self.reach = 2 * (self.K - 1 - self.pad_l)
        self.log1p = log1p
        # (B, L): the samples still needed, from a whole number of hops before the next frame to emit.  At the start of a
        # stream the buffer begins at the recording's first sample -- the reference zero-pads the ENERGY in front of a clip
        # (frontend.py:15-19 then pooling.py:41), not the waveform, so the first frames must see the true clip start --
        # and `next` (the buffer's frame number of the next frame to emit) starts at 0; later it stays at `lead`.
        self.buf: Optional[torch.Tensor] = None
        self.next = 0
        self.state: Optional[torch.Tensor] = None                         # (B, F): PCEN smoother after the last emitted frame
        self.started = False                                              # a frame has been emitted (the smoother has a state)

    def _pooled(self, x2: torch.Tensor) -> torch.Tensor:
        """Floored pooled frames (B, F, n) of the buffer through the fused forward, compression off."""
        sd = self.leaf
        return _native.leaf_forward(x2, sd._complex_conv._kernel, sd._pooling.weights, sd._pooling._bias, None, None, None, None,
                                    self.K, self.hop, pcen=False, log1p=False, algo=_native.ALGO_AUTO)

    def _emit(self, first: int, last: int) -> torch.Tensor:
        """Frames first..last (buffer numbering) finalized with the carried smoother state."""
        pooled = self._pooled(self.buf)[:, :, first:last + 1].contiguous()
        c = self.leaf._compression
        if c is None:
            out, _ = _native.pcen_stream(pooled, None, None, None, None, 1e-12, None, log1p=self.log1p)
            return out
        out, self.state = _native.pcen_stream(pooled, c.alpha, c.delta, c.root, c.ema._weights, c._floor,
                                              self.state if self.started else None)
        self.started = True
        return out

    @torch.no_grad()
    def step(self, chunk: torch.Tensor) -> torch.Tensor:
        """chunk (B,1,Tc) or (B,Tc) float32 on the device -> the frames that became final, (B,F,n) with n >= 0."""
        _native.require_hip(chunk, "LeafStream.step")
        x2 = chunk[:, 0, :] if chunk.dim() == 3 else chunk
        self.buf = x2.float() if self.buf is None else torch.cat([self.buf, x2.to(self.buf.dtype)], dim=1)
        last = (self.buf.shape[1] - 1 - self.reach) // self.hop           # last frame whose receptive field is complete
        if last < self.next:
            return chunk.new_empty((x2.shape[0], self.F, 0), dtype=torch.float32)
        out = self._emit(self.next, last)
        self._advance(last + 1)
        return out

    def _advance(self, nxt: int) -> None:
        """Frame `nxt` (buffer numbering) is the next to emit: drop the whole hops in front that it no longer needs."""
        drop = max(0, nxt - self.lead)
        self.buf = self.buf[:, drop * self.hop:].contiguous()
        self.next = nxt - drop

    @torch.no_grad()
    def flush(self) -> torch.Tensor:
        """End of the stream: the frames still owed, with the reference's zero padding behind the last sample."""
        if self.buf is None:
            return torch.empty((0, self.F, 0), device="cuda")
        last = (self.buf.shape[1] - 1) // self.hop                        # frames of a clip of this length: floor((T - 1) / hop) + 1
        out = self._emit(self.next, last) if last >= self.next else self.buf.new_empty((self.buf.shape[0], self.F, 0))
        self.buf = self.state = None
        self.next = 0
        self.started = False
        return out
