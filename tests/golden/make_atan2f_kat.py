#!/usr/bin/env python3
"""Known answers of glibc 2.35's atan2f (the fdlibm float generation the reference binary of BASELINE.md was linked
against; /root/reference/atan2.h:7-10 calls it through cargf) on discriminator-shaped operands, written with THIS
image's libm: tests/golden/atan2f_kat.bin = uint32 triples (bits of y, bits of x, bits of atan2f(y, x)).
Purpose: the exact path restates that generation operation by operation, while the oracle and the
reference call whatever libm the host has.  On a host whose libm rounds atan2f differently the REFERENCE ITSELF prints
other soft symbols; the tests then say so and skip instead of failing (wm_exact.h is still held against these answers)."""
import ctypes
import os
import platform

import numpy as np

libm = ctypes.CDLL("libm.so.6")
libm.atan2f.restype = ctypes.c_float
libm.atan2f.argtypes = [ctypes.c_float, ctypes.c_float]
rng = np.random.default_rng(2035)
n = 8192
lim = np.where(np.arange(n) & 1, 2880, 1016)
v = [rng.integers(-lim, lim + 1) for _ in range(4)]
weak = [rng.random(n) < 0.3 for _ in range(4)]
v = [np.where(w, rng.integers(-12, 13, n), x).astype(np.float32) for w, x in zip(weak, v)]
re = (v[0] * v[2] - v[1] * (-v[3])).astype(np.float32)
im = (v[0] * (-v[3]) + v[1] * v[2]).astype(np.float32)
z = np.float32(0.0)
special = [(z, z), (-z, z), (z, -z), (-z, -z), (z, np.float32(3.5)), (-z, np.float32(3.5)), (z, np.float32(-3.5)), (-z, np.float32(-3.5)),
           (np.float32(2.25), z), (np.float32(2.25), -z), (np.float32(-2.25), z), (np.float32(-2.25), -z)]
ys = np.concatenate([im, np.array([s[0] for s in special], np.float32)])
xs = np.concatenate([re, np.array([s[1] for s in special], np.float32)])
out = np.array([libm.atan2f(float(y), float(x)) for y, x in zip(ys, xs)], np.float32)
kat = np.stack([ys.view(np.uint32), xs.view(np.uint32), out.view(np.uint32)], axis=1).astype("<u4")
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "atan2f_kat.bin")
kat.tofile(path)
print(f"{len(kat)} known answers from {platform.libc_ver()} -> {path}")
