"""rtl-wmbus_amd -- MI355X-native back end for the rtl-wmbus cu8 -> datagram hot path.

Thin ctypes mirror of the C ABI in include/wmbus_hip.h (libwmbus_hip.so: hand-written HIP kernels
for gfx950 + host packet decoders).  The directory name contains a hyphen, so import it with

    import importlib; wm = importlib.import_module("rtl-wmbus_amd")

There is no CPU fallback: opening a receiver without a HIP device raises WmbusError.
"""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("WMBUS_HIP_LIB") or os.path.join(HERE, "libwmbus_hip.so")   # override: A/B of two builds
SYNTH_PATH = os.path.join(HERE, "libwmbus_synth.so")
CLI_PATH = os.path.join(HERE, "rtl_wmbus_hip")

CHAIN_T1C1, CHAIN_S1 = 0, 1
ALGO_RLA, ALGO_T2A = 0, 1
BLOCK_BYTES = 4096


class WmbusError(RuntimeError):
    pass


def build(force=False):
    """Compile every native target for gfx950 (hipcc cross-compiles without a GPU)."""
    if force:
        subprocess.run(["make", "-s", "-C", HERE, "clean"], check=True)
    subprocess.run(["make", "-s", "-C", HERE], check=True)


class Cfg(ctypes.Structure):
    _fields_ = [("decimation", ctypes.c_uint), ("simultaneous", ctypes.c_int), ("accurate_atan", ctypes.c_int),
                ("remove_dc", ctypes.c_int), ("t1c1_enabled", ctypes.c_int), ("s1_enabled", ctypes.c_int),
                ("rla_enabled", ctypes.c_int), ("time2_enabled", ctypes.c_int), ("show_algorithm", ctypes.c_int),
                ("fixed_timestamp", ctypes.c_int), ("n_streams", ctypes.c_uint), ("device", ctypes.c_int),
                ("max_push_bytes", ctypes.c_size_t), ("seg_len", ctypes.c_uint), ("rla_seg_len", ctypes.c_uint),
                ("warmup_t1c1", ctypes.c_uint),
                ("warmup_s1", ctypes.c_uint), ("rla_lookback", ctypes.c_uint), ("host_threads", ctypes.c_uint),
                ("keep_taps", ctypes.c_int), ("prefilter", ctypes.c_int), ("atan_mode", ctypes.c_int), ("spill_words", ctypes.c_uint), ("dedup_twins", ctypes.c_int), ("only_crc_ok", ctypes.c_int),
                ("input_windows", ctypes.c_uint), ("tolerance_mode", ctypes.c_int),
                # tuning and test knobs (0 = default), wmbus_hip.h
                ("rounds_on_host", ctypes.c_uint), ("rssi_full", ctypes.c_uint), ("rssi_dense_pm", ctypes.c_uint), ("bursts_to_host", ctypes.c_uint),
                ("burst_caps", ctypes.c_uint * 4), ("k1_small_tile", ctypes.c_uint), ("k1_tiles_per_block", ctypes.c_uint), ("clock_waves", ctypes.c_uint)]


class Line(ctypes.Structure):
    _fields_ = [("stream", ctypes.c_uint32), ("chain", ctypes.c_uint8), ("algo", ctypes.c_uint8),
                ("crc_ok", ctypes.c_uint8), ("pad", ctypes.c_uint8), ("sample", ctypes.c_uint64),
                ("text_off", ctypes.c_uint32), ("text_len", ctypes.c_uint32)]


class Timing(ctypes.Structure):
    _fields_ = [("demod_ms", ctypes.c_float), ("clock_ms", ctypes.c_float), ("rla_ms", ctypes.c_float),
                ("gather_ms", ctypes.c_float), ("d2h_ms", ctypes.c_float), ("gpu_total_ms", ctypes.c_float),
                ("host_decode_ms", ctypes.c_float), ("clock_reruns", ctypes.c_uint), ("rla_reruns", ctypes.c_uint),
                ("ema_retries", ctypes.c_uint), ("chips", (ctypes.c_uint64 * 2) * 2), ("bursts", ctypes.c_uint64),
                ("turn_wait_ms", ctypes.c_float), ("warnings", ctypes.c_uint), ("slow_path", ctypes.c_uint),
                ("rssi_ms", ctypes.c_float), ("rssi_mode", ctypes.c_uint), ("rssi_tiles", ctypes.c_uint),
                ("clock_round", ctypes.c_uint * 4), ("rla_round", ctypes.c_uint * 4)]

    def as_dict(self):
        d = {k: getattr(self, k) for k in ("demod_ms", "clock_ms", "rla_ms", "gather_ms", "d2h_ms", "gpu_total_ms",
                                           "host_decode_ms", "clock_reruns", "rla_reruns", "ema_retries", "bursts", "turn_wait_ms", "warnings", "slow_path",
                                           "rssi_ms", "rssi_mode", "rssi_tiles")}
        d["chips"] = [[int(self.chips[ch][al]) for al in range(2)] for ch in range(2)]      # [chain][algo]
        d["clock_round"] = [int(v) for v in self.clock_round]; d["rla_round"] = [int(v) for v in self.rla_round]
        return d


RSSI_EVERY_SAMPLE, RSSI_ON_DEMAND, RSSI_PAUSED, RSSI_FELL_BACK = 0, 1, 2, 3            # wmbus_timing.rssi_mode


FILL_FN = ctypes.CFUNCTYPE(ctypes.c_size_t, ctypes.c_void_p, ctypes.c_uint, ctypes.c_uint, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t)
LINES_FN = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.c_uint, ctypes.c_uint, ctypes.POINTER(Line), ctypes.c_size_t, ctypes.c_void_p,
                            ctypes.POINTER(Timing))


class BatchIo(ctypes.Structure):
    _fields_ = [("fill", FILL_FN), ("lines", LINES_FN), ("user", ctypes.c_void_p), ("resident_bytes", ctypes.c_size_t), ("passes", ctypes.c_uint),
                ("self_staged", ctypes.c_uint)]


class BatchStats(ctypes.Structure):
    _fields_ = [("samples", ctypes.c_uint64), ("lines", ctypes.c_uint64), ("seconds", ctypes.c_double), ("pushes", ctypes.c_uint),
                ("warnings", ctypes.c_uint)]


EXPORTS = ["wmbus_batch_plan", "wmbus_batch_open", "wmbus_batch_close", "wmbus_batch_last_error", "wmbus_batch_contexts", "wmbus_batch_context", "wmbus_batch_stage",
           "wmbus_batch_device_input", "wmbus_batch_run",
           "wmbus_runtime_init", "wmbus_default_cfg", "wmbus_open", "wmbus_close", "wmbus_last_error", "wmbus_stage", "wmbus_device_input",
           "wmbus_process", "wmbus_collect", "wmbus_lines", "wmbus_lines_text", "wmbus_get_timing", "wmbus_read_tap",
           "wmbus_read_chips", "wmbus_device_count", "wmbus_selftest_math", "wmbus_selftest_fir", "wmbus_alloc_pinned", "wmbus_free_pinned",
           "wmbus_debug_replay_decode"]

_lib = None


def lib():
    """Load libwmbus_hip.so (fails loudly if it has not been built)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise WmbusError(f"{LIB_PATH} is missing: run __graft_entry__.build() / make -C rtl-wmbus_amd")
        L = ctypes.CDLL(LIB_PATH)
        L.wmbus_runtime_init()           # before the first HIP call of the process (the hardware-queue default, see wmbus_hip.h)
        vp, u, sz = ctypes.c_void_p, ctypes.c_uint, ctypes.c_size_t
        L.wmbus_default_cfg.argtypes = [ctypes.POINTER(Cfg)]
        L.wmbus_open.argtypes = [ctypes.POINTER(Cfg), ctypes.POINTER(vp)]
        L.wmbus_close.argtypes = [vp]
        L.wmbus_last_error.argtypes = [vp]; L.wmbus_last_error.restype = ctypes.c_char_p
        L.wmbus_stage.argtypes = [vp, u, vp, sz]
        L.wmbus_device_input.argtypes = [vp, u]; L.wmbus_device_input.restype = vp
        L.wmbus_process.argtypes = [vp, sz]
        L.wmbus_collect.argtypes = [vp]
        L.wmbus_lines.argtypes = [vp, ctypes.POINTER(ctypes.POINTER(Line))]; L.wmbus_lines.restype = sz
        L.wmbus_lines_text.argtypes = [vp, ctypes.POINTER(sz)]; L.wmbus_lines_text.restype = vp
        L.wmbus_get_timing.argtypes = [vp, ctypes.POINTER(Timing)]
        L.wmbus_read_tap.argtypes = [vp, ctypes.c_char_p, ctypes.c_int, u, vp, sz]; L.wmbus_read_tap.restype = ctypes.c_long
        L.wmbus_read_chips.argtypes = [vp, ctypes.c_int, ctypes.c_int, u, vp, vp, sz]; L.wmbus_read_chips.restype = ctypes.c_long
        L.wmbus_device_count.restype = ctypes.c_int
        L.wmbus_alloc_pinned.argtypes = [sz]; L.wmbus_alloc_pinned.restype = vp
        L.wmbus_free_pinned.argtypes = [vp]
        L.wmbus_selftest_math.argtypes = [ctypes.c_int] + [vp] * 6 + [sz]
        L.wmbus_selftest_fir.argtypes = [ctypes.c_int, vp, vp, sz]
        L.wmbus_debug_replay_decode.argtypes = [vp, u, ctypes.POINTER(ctypes.c_double)]; L.wmbus_debug_replay_decode.restype = ctypes.c_long
        L.wmbus_batch_open.argtypes = [ctypes.POINTER(Cfg), u, ctypes.POINTER(vp)]
        L.wmbus_batch_plan.argtypes = [ctypes.POINTER(Cfg), u, ctypes.POINTER(u), u]; L.wmbus_batch_plan.restype = u
        L.wmbus_batch_close.argtypes = [vp]
        L.wmbus_batch_last_error.argtypes = [vp]; L.wmbus_batch_last_error.restype = ctypes.c_char_p
        L.wmbus_batch_contexts.argtypes = [vp]; L.wmbus_batch_contexts.restype = u
        L.wmbus_batch_context.argtypes = [vp, u, ctypes.POINTER(u), ctypes.POINTER(u)]; L.wmbus_batch_context.restype = vp
        L.wmbus_batch_stage.argtypes = [vp, u, vp, sz]
        L.wmbus_batch_device_input.argtypes = [vp, u]; L.wmbus_batch_device_input.restype = vp
        L.wmbus_batch_run.argtypes = [vp, ctypes.POINTER(BatchIo), ctypes.POINTER(BatchStats)]
        _lib = L
    return _lib


def device_count():
    return int(lib().wmbus_device_count())


def pinned_array(nbytes):
    """uint8 numpy array over page-locked host memory (kept alive by the array's base object)."""
    p = lib().wmbus_alloc_pinned(nbytes)
    if not p:
        raise WmbusError("wmbus_alloc_pinned failed")

    class _Owner:
        def __init__(self, ptr): self.ptr = ptr
        def __del__(self): lib().wmbus_free_pinned(self.ptr)
    buf = (ctypes.c_uint8 * nbytes).from_address(p)
    buf._owner = _Owner(p)
    return np.frombuffer(buf, np.uint8)


def selftest_math(a, b, device=0):
    """Device sqrt / divide / atan2f / discriminator of wm_exact.h on float32 arrays a, b."""
    a = np.ascontiguousarray(a, np.float32); b = np.ascontiguousarray(b, np.float32)
    outs = [np.empty_like(a) for _ in range(4)]
    rc = lib().wmbus_selftest_math(device, a.ctypes.data, b.ctypes.data, *[o.ctypes.data for o in outs], a.size)
    if rc:
        raise WmbusError(f"selftest_math failed: {rc}")
    return dict(sqrt=outs[0], div=outs[1], atan2=outs[2], disc=outs[3])


def selftest_fir(x, device=0):
    """The demodulation kernel's 11-tap and 46-tap low-pass filters on the device: x = 48 samples of history + n inputs
    (n a multiple of 4, <= 976); returns (y11[n], y46[n])."""
    x = np.ascontiguousarray(x, np.float32)
    n = x.size - 48
    out = np.empty(2 * n, np.float32)
    rc = lib().wmbus_selftest_fir(device, x.ctypes.data, out.ctypes.data, n)
    if rc:
        raise WmbusError(f"selftest_fir failed: {rc}")
    return out[:n], out[n:]


def _make_cfg(n_streams=1, max_push_bytes=4 << 20, decimation=2, simultaneous=False, accurate_atan=True,
              remove_dc=False, t1c1=True, s1=True, rla=True, time2=True, show_algorithm=True, device=0,
              seg_len=0, rla_seg_len=0, warmup_t1c1=0, warmup_s1=0, rla_lookback=0, host_threads=0, fixed_timestamp=True,
              prefilter=0, atan_mode=0, keep_taps=True, spill_words=0, input_windows=1, dedup_twins=False, only_crc_ok=False, tolerance_mode=0,
              rounds_on_host=False, rssi_full=False, rssi_dense_pm=0, bursts_to_host=False, burst_caps=None, k1_small_tile=False, k1_tiles_per_block=0, clock_waves=0):
    c = Cfg()
    lib().wmbus_default_cfg(ctypes.byref(c))
    # test campaigns (tests/README.md): the whole GPU suite once with every hand-off failure finished by the host-driven path,
    # once with every burst through the host packet decoders -- defaults of THIS wrapper, the library reads no environment
    rounds_on_host = rounds_on_host or os.environ.get("WMBUS_TEST_ROUNDS_ON_HOST") == "1"
    bursts_to_host = bursts_to_host or os.environ.get("WMBUS_TEST_BURSTS_TO_HOST") == "1"
    c.decimation, c.simultaneous, c.accurate_atan, c.remove_dc = decimation, int(simultaneous), int(accurate_atan), int(remove_dc)
    c.t1c1_enabled, c.s1_enabled, c.rla_enabled, c.time2_enabled = int(t1c1), int(s1), int(rla), int(time2)
    c.show_algorithm, c.fixed_timestamp = int(show_algorithm), int(fixed_timestamp)
    c.n_streams, c.device, c.max_push_bytes = n_streams, device, max_push_bytes
    c.seg_len, c.rla_seg_len, c.warmup_t1c1, c.warmup_s1 = seg_len, rla_seg_len, warmup_t1c1, warmup_s1
    c.rla_lookback, c.host_threads = rla_lookback, host_threads
    c.keep_taps, c.prefilter, c.atan_mode, c.spill_words, c.input_windows = int(keep_taps), prefilter, atan_mode, spill_words, input_windows
    c.dedup_twins, c.only_crc_ok, c.tolerance_mode = int(dedup_twins), int(only_crc_ok), int(tolerance_mode)
    c.rounds_on_host, c.rssi_full, c.rssi_dense_pm, c.bursts_to_host, c.k1_small_tile = int(rounds_on_host), int(rssi_full), int(rssi_dense_pm), int(bursts_to_host), int(k1_small_tile)
    c.k1_tiles_per_block = int(k1_tiles_per_block)
    c.clock_waves = int(clock_waves)
    for i, v in enumerate(burst_caps or ()):
        c.burst_caps[i] = int(v)
    return c


def batch_plan(n_streams, contexts=0, tolerance_mode=0):
    """Captures per context of a wmbus_batch of `n_streams` (no device needed)."""
    c = _make_cfg(n_streams=n_streams, tolerance_mode=tolerance_mode)
    out = (ctypes.c_uint * max(1, n_streams))()
    n = lib().wmbus_batch_plan(ctypes.byref(c), contexts, out, max(1, n_streams))
    return [int(out[i]) for i in range(n)]


class Batch:
    """wmbus_batch_*: `n_streams` captures on one device, split over several receiver contexts that the LIBRARY drives
    (include/wmbus_hip.h).  Keyword arguments as for Receiver; `contexts` = 0 takes the library's default split."""

    def __init__(self, n_streams, contexts=0, **kw):
        L = lib()
        kw.setdefault("keep_taps", False)                   # the product's configuration (the CLI's): no debug views, RSSI on demand
        self.cfg = _make_cfg(n_streams=n_streams, **kw)
        self.n_streams = n_streams
        self._h = ctypes.c_void_p()
        rc = L.wmbus_batch_open(ctypes.byref(self.cfg), contexts, ctypes.byref(self._h))
        if rc:
            msg = L.wmbus_batch_last_error(self._h).decode() if self._h else "allocation failed"
            L.wmbus_batch_close(self._h)
            self._h = None
            raise WmbusError(f"wmbus_batch_open failed ({rc}): {msg}")
        self.contexts = []                                  # (Receiver view, first stream, streams)
        for i in range(L.wmbus_batch_contexts(self._h)):
            f, n = ctypes.c_uint(), ctypes.c_uint()
            h = L.wmbus_batch_context(self._h, i, ctypes.byref(f), ctypes.byref(n))
            self.contexts.append((Receiver._view(h, n.value), f.value, n.value))

    def close(self):
        if self._h:
            lib().wmbus_batch_close(self._h)
            self._h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _chk(self, rc):
        if rc:
            raise WmbusError(f"error {rc}: {lib().wmbus_batch_last_error(self._h).decode()}")

    def stage(self, stream, cu8):
        cu8 = np.ascontiguousarray(cu8, dtype=np.uint8)
        self._chk(lib().wmbus_batch_stage(self._h, stream, cu8.ctypes.data, cu8.size))

    def run_resident(self, nbytes, passes, on_push=None, want_lines=True):
        """Every context makes `passes` pushes of the `nbytes` staged per stream.  on_push(first_stream, n_streams,
        [line dicts] (None unless want_lines), timing dict) is called per context push if given.  Returns the stats."""
        return self._run(BatchIo(FILL_FN(), self._sink(on_push, want_lines), None, nbytes, passes, 0))

    def run_from(self, fill, on_push=None, self_staged=False, want_lines=True):
        """Host-sourced run (input_windows=2): fill(first_stream, n_streams, slab) -> bytes per stream, where slab is a
        uint8 array [n_streams, max_push_bytes] over the library's page-locked staging memory (None with self_staged:
        the callback then stages from pinned memory of its own with Batch.stage); 0 ends that group."""
        def c_fill(user, first, n, slab, pitch, cap_):
            a = None
            if slab:
                a = np.ctypeslib.as_array(ctypes.cast(slab, ctypes.POINTER(ctypes.c_uint8)), shape=(n * pitch,)).reshape(n, pitch)[:, :cap_]
            return int(fill(first, n, a))
        return self._run(BatchIo(FILL_FN(c_fill), self._sink(on_push, want_lines), None, 0, 0, int(self_staged)))

    def _sink(self, on_push, want_lines=True):
        if on_push is None:
            return LINES_FN()

        def c_lines(user, first, n, lines, n_lines, text, timing):
            recs = None
            if want_lines:
                recs = []
                for k in range(n_lines):
                    ln = lines[k]
                    recs.append(dict(stream=ln.stream, chain=ln.chain, algo=ln.algo, crc_ok=ln.crc_ok, sample=ln.sample,
                                     text=ctypes.string_at(text + ln.text_off, ln.text_len).decode()))
            on_push(first, n, recs, timing.contents.as_dict())
        return LINES_FN(c_lines)

    def _run(self, io):
        st = BatchStats()
        self._io = io                                       # the callbacks must outlive the call
        self._chk(lib().wmbus_batch_run(self._h, ctypes.byref(io), ctypes.byref(st)))
        return dict(samples=int(st.samples), lines=int(st.lines), seconds=float(st.seconds), pushes=int(st.pushes), warnings=int(st.warnings))


class Receiver:
    """`n_streams` captures through the GPU back end, in lock step.

    Keyword arguments mirror the reference's switches: decimation (-d), simultaneous (-s),
    accurate_atan (not -a), remove_dc (-o), t1c1 / s1 (not -p T / -p S), rla (-r), time2 (-t),
    show_algorithm (-v).
    """

    def __init__(self, n_streams=1, **kw):
        L = lib()
        c = _make_cfg(n_streams=n_streams, **kw)
        self.cfg = c
        self.n_streams = n_streams
        self._h = ctypes.c_void_p()
        rc = L.wmbus_open(ctypes.byref(c), ctypes.byref(self._h))
        if rc:
            msg = L.wmbus_last_error(self._h).decode() if self._h else "allocation failed"
            L.wmbus_close(self._h)
            self._h = None
            raise WmbusError(f"wmbus_open failed ({rc}): {msg}")

    @classmethod
    def _view(cls, handle, n_streams):
        """A Receiver over a context someone else owns (a wmbus_batch): never closed from here."""
        r = cls.__new__(cls)
        r._h, r._borrowed, r.n_streams, r.cfg = ctypes.c_void_p(handle), True, n_streams, None
        return r

    def close(self):
        if getattr(self, "_h", None) and not getattr(self, "_borrowed", False):
            lib().wmbus_close(self._h)
        self._h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _chk(self, rc):
        if rc:
            raise WmbusError(f"error {rc}: {lib().wmbus_last_error(self._h).decode()}")

    def device_input(self, stream):
        return lib().wmbus_device_input(self._h, stream)

    def stage(self, stream, cu8):
        cu8 = np.ascontiguousarray(cu8, dtype=np.uint8)
        self._chk(lib().wmbus_stage(self._h, stream, cu8.ctypes.data, cu8.size))

    def process(self, nbytes):
        self._chk(lib().wmbus_process(self._h, nbytes))

    def collect(self):
        """Returns the datagram text of the push, streams in order, reference stdout order within."""
        L = lib()
        self._chk(L.wmbus_collect(self._h))
        n = ctypes.c_size_t()
        p = L.wmbus_lines_text(self._h, ctypes.byref(n))
        return ctypes.string_at(p, n.value).decode() if n.value else ""

    def lines(self):
        L = lib()
        p = ctypes.POINTER(Line)()
        n = L.wmbus_lines(self._h, ctypes.byref(p))
        sz = ctypes.c_size_t()
        t = L.wmbus_lines_text(self._h, ctypes.byref(sz))
        text = ctypes.string_at(t, sz.value) if sz.value else b""
        return [dict(stream=p[i].stream, chain=p[i].chain, algo=p[i].algo, crc_ok=p[i].crc_ok, sample=p[i].sample,
                     text=text[p[i].text_off:p[i].text_off + p[i].text_len].decode()) for i in range(n)]

    def lines_count(self):
        return int(lib().wmbus_lines(self._h, None))

    def timing(self):
        t = Timing()
        self._chk(lib().wmbus_get_timing(self._h, ctypes.byref(t)))
        return t.as_dict()

    def push(self, streams):
        """Stage one equally long cu8 array per stream, process, decode; returns the text."""
        n = None
        for s, a in enumerate(streams):
            a = np.ascontiguousarray(a, dtype=np.uint8)
            n = a.size if n is None else n
            assert a.size == n and n % BLOCK_BYTES == 0
            self.stage(s, a)
        self.process(n)
        return self.collect()

    def run(self, cu8, push_bytes=None):
        """Whole capture(s) through the receiver in pushes of `push_bytes`; partial tail dropped
        like the reference does at EOF (rtl_wmbus.c:1304-1308).  cu8: one array or a list."""
        streams = [cu8] if isinstance(cu8, np.ndarray) and cu8.ndim == 1 else list(cu8)
        streams = [np.ascontiguousarray(a, dtype=np.uint8) for a in streams]
        total = streams[0].size // BLOCK_BYTES * BLOCK_BYTES
        step = push_bytes or self.cfg.max_push_bytes
        per_stream = [[] for _ in streams]
        for off in range(0, total, step):
            n = min(step, total - off)
            self.push([a[off:off + n] for a in streams])
            for ln in self.lines():
                per_stream[ln["stream"]].append(ln["text"])
        return ["".join(x) for x in per_stream]

    def read_tap(self, what, chain, stream, n):
        dt = np.float32 if what == "dphi" else np.uint8
        out = np.zeros(n, dt)
        r = lib().wmbus_read_tap(self._h, what.encode(), chain, stream, out.ctypes.data, n)
        if r < 0:
            raise WmbusError(f"read_tap({what}) failed: {r}")
        return out[:r]

    def read_chips(self, chain, algo, stream, cap=1 << 22):
        w = np.zeros(cap, np.uint32)
        pos = np.zeros(cap, np.uint64)
        r = lib().wmbus_read_chips(self._h, chain, algo, stream, w.ctypes.data, pos.ctypes.data, cap)
        if r < 0:
            raise WmbusError(f"read_chips failed: {r}")
        if r > cap:
            raise WmbusError("read_chips: capacity too small")
        return w[:r], pos[:r]


# ---- synthetic captures (host-only helper library) -------------------------------------------------
class SynthCfg(ctypes.Structure):
    _fields_ = [("seed", ctypes.c_uint64), ("fs_khz", ctypes.c_uint), ("noise_sigma", ctypes.c_double),
                ("amplitude", ctypes.c_double), ("frames_per_s", ctypes.c_double), ("kinds", ctypes.c_uint),
                ("l_min", ctypes.c_int), ("l_max", ctypes.c_int), ("t1c1_center_khz", ctypes.c_double),
                ("s1_center_khz", ctypes.c_double), ("max_offset_khz", ctypes.c_double)]


class SynthFrame(ctypes.Structure):
    _fields_ = [("kind", ctypes.c_uint32), ("n_samples", ctypes.c_uint32), ("start_sample", ctypes.c_uint64),
                ("len", ctypes.c_uint16), ("complete", ctypes.c_uint8), ("pad", ctypes.c_uint8),
                ("telegram", ctypes.c_uint8 * 256)]


T1, C1A, C1B, S1 = 1, 2, 4, 8
_synth = None


def _synth_lib():
    global _synth
    if _synth is None:
        if not os.path.exists(SYNTH_PATH):
            raise WmbusError(f"{SYNTH_PATH} is missing: run make -C rtl-wmbus_amd")
        L = ctypes.CDLL(SYNTH_PATH)
        L.wmsynth_default_cfg.argtypes = [ctypes.POINTER(SynthCfg)]
        L.wmsynth_generate.restype = ctypes.c_size_t
        L.wmsynth_generate.argtypes = [ctypes.POINTER(SynthCfg), ctypes.c_void_p, ctypes.c_size_t,
                                       ctypes.POINTER(SynthFrame), ctypes.c_size_t]
        _synth = L
    return _synth


def synth_capture(seed, n_samples, fs_khz=1600, kinds=T1 | C1A | C1B, frames_per_s=20.0, amplitude=60.0,
                  noise_sigma=3.0, t1c1_center_khz=0.0, s1_center_khz=0.0, l_min=10, l_max=60, out=None,
                  max_frames=1024):
    """SURVEY.md section 8(d) recipe.  Returns (cu8 uint8[2*n_samples], [frame dicts])."""
    L = _synth_lib()
    c = SynthCfg()
    L.wmsynth_default_cfg(ctypes.byref(c))
    c.seed, c.fs_khz, c.kinds, c.frames_per_s = seed, fs_khz, kinds, frames_per_s
    c.amplitude, c.noise_sigma, c.t1c1_center_khz, c.s1_center_khz = amplitude, noise_sigma, t1c1_center_khz, s1_center_khz
    c.l_min, c.l_max = l_min, l_max
    buf = out if out is not None else np.empty(2 * n_samples, np.uint8)
    fr = (SynthFrame * max_frames)()
    k = L.wmsynth_generate(ctypes.byref(c), buf.ctypes.data, n_samples, fr, max_frames)
    frames = [dict(kind=fr[i].kind, start=fr[i].start_sample, n=fr[i].n_samples, complete=bool(fr[i].complete),
                   telegram=bytes(fr[i].telegram[:fr[i].len])) for i in range(min(k, max_frames))]
    return buf, frames
