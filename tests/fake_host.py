"""A fake Photoshop host for the FormatRecord-protocol shim: advanceState() feeds / collects rows of a numpy image
according to theRect32, abortProc can be armed to cancel after N polls, every rectangle request is logged."""
from __future__ import annotations

import ctypes

import numpy as np

import harness

pkg = harness.pkg
H = pkg.host

MODES = {(1, 8): H.plugInModeGrayScale, (1, 16): H.plugInModeGray16, (1, 32): H.plugInModeGray32,
         (3, 8): H.plugInModeRGBColor, (3, 16): H.plugInModeRGB48, (3, 32): H.plugInModeRGB96}


class FakeHost:
    def __init__(self, width, height, depth, planes, max_data=0, image=None, abort_after=None, fail_at_row=None):
        self.rects, self.polls = [], 0
        self.abort_after, self.fail_at_row = abort_after, fail_at_row
        dt = harness.src_dtype(depth)
        self.image = image if image is not None else np.zeros((height, width * planes), dtype=dt)
        self.writing = image is not None          # True: host -> plug-in (save); False: plug-in -> host (open)
        fr = H.FormatRecord()
        fr.HostSupports32BitCoordinates = 1
        fr.PluginUsing32BitCoordinates = 1        # AvifFormat.cpp:113-116
        fr.imageSize32.h, fr.imageSize32.v = width, height
        fr.imageSize.h, fr.imageSize.v = min(width, 32767), min(height, 32767)
        fr.depth, fr.planes = depth, planes
        fr.imageMode = MODES[(1 if planes <= 2 else 3, depth)]
        fr.maxData = max_data
        self._abort = H.TestAbortProc(self._abort_proc)
        self._advance = H.AdvanceStateProc(self._advance_state)
        self._progress = H.ProgressProc(lambda a, b: None)
        fr.abortProc, fr.advanceState, fr.progressProc = self._abort, self._advance, self._progress
        self.fr = fr

    def _abort_proc(self):
        self.polls += 1
        return 1 if (self.abort_after is not None and self.polls > self.abort_after) else 0

    def _advance_state(self):
        r = self.fr.theRect32
        self.rects.append((r.top, r.left, r.bottom, r.right))
        if self.fail_at_row is not None and r.top <= self.fail_at_row < r.bottom:
            return -36          # ioErr: any host error must come back unchanged
        n = r.bottom - r.top
        nbytes = n * self.fr.rowBytes
        buf = (ctypes.c_uint8 * nbytes).from_address(self.fr.data)
        view = np.frombuffer(buf, dtype=self.image.dtype).reshape(n, -1)
        if self.writing:
            view[:] = self.image[r.top:r.bottom]
        else:
            self.image[r.top:r.bottom] = view
        return 0
