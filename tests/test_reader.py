"""The drop-in CLI's live-stream input side (rtl-wmbus_amd/csrc/wm_reader.c, plain C, no GPU) behind a paced pipe:
latency of a staged block is bounded by -L whatever the push size, only whole 4096-byte blocks are pushed and in order,
the partial tail is dropped at end of input (/root/reference/rtl_wmbus.c:1304-1308), and with -f the staged blocks are
decoded BEFORE the flow time-out is reported (the reference processes every block it has read, :1300-1308)."""
import ctypes
import os
import subprocess
import threading
import time

import pytest

from conftest import ROOT

SRC = os.path.join(ROOT, "rtl-wmbus_amd", "csrc", "wm_reader.c")
PUSH_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_ubyte), ctypes.c_size_t)


class Cfg(ctypes.Structure):
    _fields_ = [("fd", ctypes.c_int), ("max_push", ctypes.c_size_t), ("max_latency_ms", ctypes.c_uint), ("flow_timeout_ms", ctypes.c_uint)]


@pytest.fixture(scope="module")
def reader(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("reader") / "libwm_reader.so")
    subprocess.run(["gcc", "-O2", "-Wall", "-Wextra", "-Werror", "-std=gnu11", "-fPIC", "-shared", "-o", so, SRC], check=True)
    L = ctypes.CDLL(so)
    L.wm_reader_run.argtypes = [ctypes.POINTER(Cfg), PUSH_FN, ctypes.c_void_p]
    return L


def run(reader, feed, max_push, latency, flow=0, fail_at=None):
    """feed(write_fd) runs on a thread; returns (return code, [(time, bytes)], total time)."""
    r, w = os.pipe()
    got = []

    def push(_user, buf, n):
        got.append((time.monotonic(), ctypes.string_at(buf, n)))
        return 1 if fail_at is not None and len(got) == fail_at else 0
    th = threading.Thread(target=feed, args=(w,))
    th.start()
    cb = PUSH_FN(push)
    rc = reader.wm_reader_run(ctypes.byref(Cfg(r, max_push, latency, flow)), cb, None)
    th.join()
    os.close(r)
    return rc, got


def test_paced_stream_is_pushed_within_the_latency_bound(reader):
    """3.2 MB/s in 16 KiB pieces (an RTL-SDR at 1.6 MS/s) into a 1 MiB push: without the bound the first byte would wait
    0.33 s for the push to fill."""
    piece, n_pieces, latency = 16384, 48, 40
    data = os.urandom(piece * n_pieces + 1000)                # + a partial tail
    sent = []

    def feed(w):
        for k in range(n_pieces):
            sent.append(time.monotonic())
            os.write(w, data[k * piece:(k + 1) * piece])
            time.sleep(piece / 3.2e6)
        os.write(w, data[n_pieces * piece:])
        os.close(w)
    rc, got = run(reader, feed, 1 << 20, latency)
    assert rc == 0
    out = b"".join(b for _, b in got)
    assert all(len(b) % 4096 == 0 and 0 < len(b) <= 1 << 20 for _, b in got)
    assert out == data[:len(data) // 4096 * 4096]             # whole blocks, in order, tail dropped
    assert len(got) >= 4                                      # several small pushes, not one full one at the end
    # every byte left within latency (+ scheduling slack) of its arrival
    pos = 0
    for t_push, b in got:
        first_piece = pos // piece
        assert t_push - sent[first_piece] < (latency + 200) / 1e3, (pos, t_push - sent[first_piece])     # generous slack: a loaded CI host
        pos += len(b)


def test_fast_input_fills_whole_pushes(reader):
    data = os.urandom(5 * (1 << 18) + 4096 * 3 + 17)

    def feed(w):
        os.write(w, data)
        os.close(w)
    rc, got = run(reader, feed, 1 << 18, 0)                   # -L 0: only full pushes (and the end)
    assert rc == 0
    assert [len(b) for _, b in got] == [1 << 18] * 5 + [4096 * 3]
    assert b"".join(b for _, b in got) == data[:len(data) // 4096 * 4096]


def test_flow_timeout_pushes_the_staged_blocks_first(reader):
    data = os.urandom(4096 * 5 + 100)
    t0 = time.monotonic()

    def feed(w):
        os.write(w, data)
        time.sleep(1.0)                                       # the writer stalls with the pipe still open
        os.close(w)
    rc, got = run(reader, feed, 1 << 20, 0, flow=300)
    assert rc == 1                                            # WM_READER_FLOW_STOPPED
    assert b"".join(b for _, b in got) == data[:4096 * 5]     # read before the stall: decoded, not lost with the process
    assert 0.25 < got[-1][0] - t0 < 0.95


def test_flow_timeout_counts_whole_blocks_not_bytes(reader):
    """the reference arms alarm(2) around the fread of ONE 4096-byte block (rtl_wmbus.c:1300-1302), so a source
    that dribbles single bytes -- never silent for long, never completing a block -- trips -f all the same."""
    t0 = time.monotonic()

    def feed(w):
        os.write(w, os.urandom(4096 * 2))
        try:
            for _ in range(12):                                   # one byte every 100 ms: no 300 ms without data
                time.sleep(0.1)
                os.write(w, b"x")
        except OSError:
            pass
        os.close(w)
    rc, got = run(reader, feed, 1 << 20, 0, flow=300)
    assert rc == 1
    assert sum(len(b) for _, b in got) == 4096 * 2
    assert 0.25 < got[-1][0] - t0 < 0.9


def test_a_failing_push_stops_the_reader(reader):
    def feed(w):
        try:
            os.write(w, os.urandom(4096 * 8))
        finally:
            os.close(w)
    rc, got = run(reader, feed, 4096 * 2, 0, fail_at=2)
    assert rc == -2 and len(got) == 2
