"""The C-ABI library loads on a CPU-only box, exports what include/wmbus_hip.h declares, and
refuses to run without a HIP device (no CPU fallback)."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT


def declared_functions():
    src = open(os.path.join(ROOT, "include", "wmbus_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(wmbus_[a-z_]+)\s*\(", src)))


def test_header_symbols_are_exported(wm):
    L = wm.lib()
    names = declared_functions()
    assert len(names) >= 14
    for n in names:
        assert hasattr(L, n), n
    assert sorted(wm.EXPORTS) == names


def test_struct_sizes_match_ctypes(wm):
    # keeps the ctypes mirror honest: compile a tiny C program printing the sizes
    import subprocess, tempfile
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "s.c")
        open(c, "w").write('#include <stdio.h>\n#include "wmbus_hip.h"\nint main(void){printf("%zu %zu %zu %zu %zu\\n",'
                           'sizeof(wmbus_cfg),sizeof(wmbus_line),sizeof(wmbus_timing),sizeof(wmbus_batch_io),sizeof(wmbus_batch_stats));return 0;}\n')
        exe = os.path.join(d, "s")
        subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), "-o", exe, c], check=True)
        sizes = list(map(int, subprocess.run([exe], capture_output=True, text=True).stdout.split()))
    assert sizes == [ctypes.sizeof(wm.Cfg), ctypes.sizeof(wm.Line), ctypes.sizeof(wm.Timing), ctypes.sizeof(wm.BatchIo), ctypes.sizeof(wm.BatchStats)]


def test_no_cpu_fallback_without_device(wm):
    if wm.device_count() > 0:
        pytest.skip("a HIP device is present")
    with pytest.raises(wm.WmbusError, match="no HIP device|failed"):
        wm.Receiver(n_streams=1)


def test_batch_needs_a_device_too(wm):
    if wm.device_count() > 0:
        pytest.skip("a HIP device is present")
    with pytest.raises(wm.WmbusError, match="no HIP device|failed"):
        wm.Batch(n_streams=128)


def test_a_refused_open_can_be_closed_whatever_its_device_ordinal(wm):
    """wmbus_open hands back the handle of a refused open (the caller reads the message, then closes it);
    wmbus_close used to index a per-device table with the ordinal that had just been refused."""
    for dev in (64, -1, 1 << 20):
        with pytest.raises(wm.WmbusError, match="device must be"):
            wm.Receiver(n_streams=1, device=dev)
    for dev in (64, -1):                                       # through the batch API too (wmbus_batch_open closes what it opened)
        with pytest.raises(wm.WmbusError):
            wm.Batch(n_streams=64, device=dev)
    for bad in (dict(decimation=0), dict(decimation=17), dict(max_push_bytes=4097), dict(n_streams=0)):
        with pytest.raises(wm.WmbusError):
            wm.Receiver(**{"n_streams": 1, **bad})


def test_cli_usage_and_exit_codes(wm):
    import subprocess
    # unknown option (the reference's getopt string has no 'h'): usage on stdout, exit 1
    p = subprocess.run([wm.CLI_PATH, "-h"], capture_output=True, text=True, stdin=subprocess.DEVNULL)
    assert p.returncode == 1 and "Usage" in p.stdout and "-p [T,S]" in p.stdout and "1 ... 16" in p.stdout       # the one argv difference is in the usage text
    p = subprocess.run([wm.CLI_PATH, "-V"], capture_output=True, text=True, stdin=subprocess.DEVNULL)
    assert p.returncode == 0 and p.stdout.startswith("rtl_wmbus:") and len(p.stdout.splitlines()) == 2      # version, commit (rtl_wmbus.c:886-890)
    p = subprocess.run([wm.CLI_PATH, "-d", "17"], capture_output=True, text=True, stdin=subprocess.DEVNULL)
    assert p.returncode == 1 and "1..16" in p.stderr
    p = subprocess.run([wm.CLI_PATH, "-d", "0", "-s"], capture_output=True, text=True, stdin=subprocess.DEVNULL)
    assert p.returncode == 1
    p = subprocess.run([wm.CLI_PATH, "-r", "1"], capture_output=True, text=True, stdin=subprocess.DEVNULL)
    assert p.returncode == 1


def test_loading_the_library_leaves_the_environment_alone(wm):
    """no setenv behind the host application's back at dlopen; wmbus_runtime_init() is the explicit call."""
    import subprocess, sys
    code = ("import ctypes, os; os.environ.pop('GPU_MAX_HW_QUEUES', None); L = ctypes.CDLL(%r); "
            "a = os.environ.get('GPU_MAX_HW_QUEUES'); g = ctypes.CDLL(None).getenv; g.restype = ctypes.c_char_p; b = g(b'GPU_MAX_HW_QUEUES'); "
            "L.wmbus_runtime_init(); c = g(b'GPU_MAX_HW_QUEUES'); print(a, b, c)" % wm.LIB_PATH)
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert p.stdout.split() == ["None", "None", "b'16'"], p.stdout + p.stderr


def test_synth_is_deterministic(wm):
    a, fa = wm.synth_capture(seed=1234, n_samples=1 << 16, kinds=15, frames_per_s=200.0)
    b, fb = wm.synth_capture(seed=1234, n_samples=1 << 16, kinds=15, frames_per_s=200.0)
    c, _ = wm.synth_capture(seed=1235, n_samples=1 << 16, kinds=15, frames_per_s=200.0)
    assert np.array_equal(a, b) and fa == fb and not np.array_equal(a, c)


def test_batch_plan_splits_into_whole_groups_of_64(wm):
    """wmbus_batch_plan (no device needed): whole 64-capture groups per context where the batch has them, the remainder
    with the last context, 8 contexts by default, never an empty context."""
    assert wm.batch_plan(1024) == [128] * 8
    assert wm.batch_plan(1024, tolerance_mode=1) == [128] * 8
    assert wm.batch_plan(320) == [64] * 5
    assert wm.batch_plan(130) == [64, 66]
    assert wm.batch_plan(1500) == [192] * 7 + [156]
    assert wm.batch_plan(63) == [63] and wm.batch_plan(1) == [1]
    assert wm.batch_plan(1024, contexts=3) == [384, 320, 320]
    assert wm.batch_plan(5, contexts=8) == [1] * 5
    for s in range(1, 700, 7):
        for c in (0, 1, 2, 5, 8, 13):
            p = wm.batch_plan(s, contexts=c)
            assert sum(p) == s and min(p) >= 1 and (c == 0 or len(p) == min(c, s))
