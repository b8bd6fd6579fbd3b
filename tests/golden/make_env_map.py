#!/usr/bin/env python3
"""Writes tests/golden/assets/env_48x20.hdr, the synthetic environment map of the infinite-light scenes (the reference ships no
.hdr file): a sky gradient, a warm ground, one small very bright sun and one dim lamp; 48 x 20, neither a power of two, so the
light's constructor resamples it (mipmap.h:70-121).  Run from the repository root."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import raytracing_amd as ra  # noqa: E402

h, w = 20, 48
y, x = np.mgrid[0:h, 0:w]
t = y / (h - 1.0)
img = np.stack([0.25 + 0.35 * t, 0.35 + 0.3 * t, 0.9 - 0.55 * t], axis=2).astype(np.float32)
img[h // 2:, :, :] = np.array([0.22, 0.17, 0.12], np.float32) * (1 + 0.3 * np.sin(x[h // 2:, :, None] / 5.0))
img[4:6, 30:32] = [60.0, 52.0, 40.0]
img[8, 7] = [6.0, 2.0, 1.0]
ra.write_hdr(os.path.join(ROOT, "tests", "golden", "assets", "env_48x20.hdr"), img)


# env_rle_40x9.hdr: a run-length encoded Radiance file (runs and literal spans per channel, the way Radiance writes scanlines of 8 to 32767
# pixels) for the reader's RLE branch; tests/test_hardening.py rebuilds the same bytes from the same pixels and decodes both.
def write_rle_fixture():
    import tempfile
    rng = np.random.default_rng(4)
    h, w = 9, 40
    img = (rng.random((h, w, 3)) ** 4 * 50).astype(np.float32); img[2, 5:30] = [1.5, 0.25, 0.0]; img[3] = 0
    flat = os.path.join(tempfile.mkdtemp(), "flat.hdr")
    ra.write_hdr(flat, img)
    raw = open(flat, "rb").read()
    start = raw.index(b"+X %d\n" % w) + len(b"+X %d\n" % w)
    rgbe = np.frombuffer(raw[start:], np.uint8).reshape(h, w, 4)
    out = bytearray(raw[:start])
    for y in range(h):
        out += bytes([2, 2, w >> 8, w & 255])
        for c in range(4):
            row = rgbe[y, :, c].tolist(); i = 0
            while i < w:
                run = 1
                while i + run < w and run < 127 and row[i + run] == row[i]:
                    run += 1
                if run >= 3:
                    out += bytes([128 + run, row[i]]); i += run
                else:
                    j = i
                    while j < w and j - i < 128 and not (j + 2 < w and row[j] == row[j + 1] == row[j + 2]):
                        j += 1
                    j = max(j, i + 1)
                    out += bytes([j - i]) + bytes(row[i:j]); i = j
    open(os.path.join(ROOT, "tests", "golden", "assets", "env_rle_40x9.hdr"), "wb").write(bytes(out))


write_rle_fixture()
