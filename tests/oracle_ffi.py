"""ctypes bindings for oracle/libwmbus_oracle.so and the reference binary (TEST INFRASTRUCTURE).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes
import os
import re
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "libwmbus_oracle.so")
ORACLE_CLI = os.path.join(ORACLE_DIR, "wmbus_oracle_cli")
REF_BIN = os.path.join(ORACLE_DIR, "_ref", "rtl_wmbus")
REF_PROBE = os.path.join(ORACLE_DIR, "_ref", "ref_probe")

_TS = re.compile(rb"[0-9]{4}-[0-9]{2}-[0-9]{2} [0-9:.]+")


def build(ref=True):
    """make -C oracle (restatement always; reference binaries when /root/reference exists)."""
    subprocess.run(["make", "-s", "-C", ORACLE_DIR, "oracle"] + (["ref"] if ref else []), check=True)


class Opts(ctypes.Structure):
    _fields_ = [("decimation", ctypes.c_uint), ("simultaneous", ctypes.c_int),
                ("accurate_atan", ctypes.c_int), ("remove_dc", ctypes.c_int),
                ("t1c1_enabled", ctypes.c_int), ("s1_enabled", ctypes.c_int),
                ("rla_enabled", ctypes.c_int), ("time2_enabled", ctypes.c_int),
                ("show_algorithm", ctypes.c_int), ("fixed_timestamp", ctypes.c_int), ("prefilter", ctypes.c_int),
                ("atan_mode", ctypes.c_int)]


class Chip(ctypes.Structure):
    _fields_ = [("sample", ctypes.c_uint32), ("chain", ctypes.c_uint8), ("algo", ctypes.c_uint8),
                ("value", ctypes.c_uint8), ("rssi", ctypes.c_uint8)]


CHIP_DTYPE = np.dtype([("sample", "<u4"), ("chain", "u1"), ("algo", "u1"), ("value", "u1"), ("rssi", "u1")])


class Taps(ctypes.Structure):
    _fields_ = [("cap", ctypes.c_size_t), ("iq", ctypes.c_void_p * 2), ("dphi_raw", ctypes.c_void_p * 2),
                ("dphi", ctypes.c_void_p * 2), ("dphi_fir", ctypes.c_void_p * 2), ("rssi", ctypes.c_void_p * 2),
                ("clk", ctypes.c_void_p * 2),
                ("bit", ctypes.c_void_p * 2), ("chips", ctypes.c_void_p), ("chips_cap", ctypes.c_size_t),
                ("chips_len", ctypes.c_size_t)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(ORACLE_SO):
            build(ref=os.path.exists("/root/reference/rtl_wmbus.c"))
        L = ctypes.CDLL(ORACLE_SO)
        L.wmo_default_opts.argtypes = [ctypes.POINTER(Opts)]
        L.wmo_new.restype = ctypes.c_void_p
        L.wmo_new.argtypes = [ctypes.POINTER(Opts)]
        L.wmo_free.argtypes = [ctypes.c_void_p]
        L.wmo_set_taps.argtypes = [ctypes.c_void_p, ctypes.POINTER(Taps)]
        L.wmo_feed.restype = ctypes.c_size_t
        L.wmo_feed.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
        L.wmo_output.restype = ctypes.c_void_p
        L.wmo_output.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_size_t)]
        L.wmo_decimated_count.restype = ctypes.c_uint64
        L.wmo_decimated_count.argtypes = [ctypes.c_void_p]
        _lib = L
    return _lib


def make_opts(decimation=2, simultaneous=0, accurate_atan=1, remove_dc=0, t1c1=1, s1=1, rla=1,
              time2=1, show_algorithm=1, prefilter=0, atan_mode=0):
    o = Opts()
    lib().wmo_default_opts(ctypes.byref(o))
    o.decimation, o.simultaneous, o.accurate_atan, o.remove_dc = decimation, simultaneous, accurate_atan, remove_dc
    o.t1c1_enabled, o.s1_enabled, o.rla_enabled, o.time2_enabled = t1c1, s1, rla, time2
    o.show_algorithm, o.fixed_timestamp = show_algorithm, 1
    o.prefilter = prefilter
    o.atan_mode = atan_mode
    return o


def opts_to_argv(o):
    """The reference/oracle CLI switches equivalent to an Opts."""
    a = []
    if o.remove_dc: a.append("-o")
    if not o.accurate_atan: a.append("-a")
    if o.decimation != 2: a += ["-d", str(o.decimation)]
    if not o.t1c1_enabled: a += ["-p", "T"]
    if not o.s1_enabled: a += ["-p", "S"]
    if not o.rla_enabled: a += ["-r", "0"]
    if not o.time2_enabled: a += ["-t", "0"]
    if o.show_algorithm: a.append("-v")
    if o.simultaneous: a.append("-s")
    return a


def run(cu8, opts, taps=False, chips=False):
    """Run the restatement over a cu8 array.  Returns dict(text=..., m=..., taps...)."""
    L = lib()
    cu8 = np.ascontiguousarray(cu8, dtype=np.uint8)
    ctx = L.wmo_new(ctypes.byref(opts))
    out = {}
    keep = []
    t = Taps()
    m_cap = cu8.size // 2 // max(1, opts.decimation) + 1
    if taps or chips:
        t.cap = m_cap
        if taps:
            for name, dt, mult in [("iq", np.float32, 2), ("dphi_raw", np.float32, 1), ("dphi", np.float32, 1),
                                   ("dphi_fir", np.float32, 1), ("rssi", np.float32, 1), ("clk", np.float32, 1), ("bit", np.uint8, 1)]:
                arrs = [np.zeros(m_cap * mult, dt) for _ in range(2)]
                keep.append(arrs)
                out[name] = arrs
                getattr(t, name)[0] = arrs[0].ctypes.data
                getattr(t, name)[1] = arrs[1].ctypes.data
        if chips:
            ch = np.zeros(4 * m_cap, CHIP_DTYPE)
            keep.append(ch)
            t.chips = ch.ctypes.data
            t.chips_cap = ch.size
        L.wmo_set_taps(ctx, ctypes.byref(t))
    L.wmo_feed(ctx, cu8.ctypes.data, cu8.size)
    n = ctypes.c_size_t()
    p = L.wmo_output(ctx, ctypes.byref(n))
    out["text"] = ctypes.string_at(p, n.value).decode() if n.value else ""
    out["m"] = int(L.wmo_decimated_count(ctx))
    if taps:
        for k in ("iq", "dphi_raw", "dphi", "dphi_fir", "rssi", "clk", "bit"):
            mult = 2 if k == "iq" else 1
            out[k] = [a[: out["m"] * mult] for a in out[k]]
    if chips:
        out["chips"] = ch[: t.chips_len].copy()
    L.wmo_free(ctx)
    return out


def run_many(caps, opts, threads=None, passes=1):
    """Datagram text of many captures, farmed over the host's cores (the ctypes call releases the GIL).
    passes > 1: the capture is fed `passes` times to one oracle instance and only the text of the LAST
    pass is returned -- the twin of a receiver that has been pushed the same bytes that many times."""
    import concurrent.futures as cf
    L = lib()
    L.wmo_clear_output.argtypes = [ctypes.c_void_p]

    def one(cu8):
        cu8 = np.ascontiguousarray(cu8, dtype=np.uint8)
        ctx = L.wmo_new(ctypes.byref(opts))
        for k in range(passes):
            if k == passes - 1:
                L.wmo_clear_output(ctx)
            L.wmo_feed(ctx, cu8.ctypes.data, cu8.size)
        n = ctypes.c_size_t()
        p = L.wmo_output(ctx, ctypes.byref(n))
        text = ctypes.string_at(p, n.value).decode() if n.value else ""
        L.wmo_free(ctx)
        return text

    with cf.ThreadPoolExecutor(threads or min(len(caps), os.cpu_count() or 1)) as ex:
        return list(ex.map(one, caps))


def mask_ts(b):
    return _TS.sub(b"TS", b)


def run_reference(cu8, argv):
    """The unmodified reference binary on the same bytes (timestamp field masked)."""
    p = subprocess.run([REF_BIN] + list(argv), input=np.ascontiguousarray(cu8, np.uint8).tobytes(),
                       capture_output=True, check=True)
    return mask_ts(p.stdout).decode()


def have_reference():
    return os.path.exists(REF_BIN)


# ---- the two low-pass filters alone (oracle: wmo_fir; reference: ref_probe fir) ---------------------------------------------
FIR_NEG_TAPS = {0: [0, 1, 9, 10], 1: list(range(0, 11)) + list(range(35, 46))}     # taps with a negative coefficient (rtl_wmbus.c:369-391)
FIR_LEN = {0: 11, 1: 46}


def fir(which, x):
    """The oracle's low-pass (0: 11 taps, T1/C1; 1: 46 taps, S1) of x, from a zeroed history."""
    L = lib()
    L.wmo_fir.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]; L.wmo_fir.restype = None
    x = np.ascontiguousarray(x, np.float32)
    y = np.empty_like(x)
    L.wmo_fir(which, x.ctypes.data, y.ctypes.data, x.size)
    return y


def fir_rows(seed, n=976):
    """Rows of 48 + n floats for the low-pass tests: what a capture gives (angles in (-pi, pi]) and what it hardly ever does --
    rows of signed zeros, among them the pattern that turns a sum started at -(b0 x) instead of 0 + b0 x into -0 at the
    OUTPUT: +0 under every negative tap, -0 under every positive one, at several alignments -- zeros between sparse samples,
    subnormals, huge values."""
    rng = np.random.default_rng(seed)
    N = 48 + n
    pz, nz = np.float32(0.0), np.float32(-0.0)
    rows = [rng.uniform(-np.pi, np.pi, N).astype(np.float32),
            np.where(rng.integers(0, 2, N) == 1, nz, pz).astype(np.float32),
            np.full(N, nz, np.float32), np.full(N, pz, np.float32)]
    for which in (0, 1):
        ln, neg = FIR_LEN[which], set(FIR_NEG_TAPS[which])
        pat = np.array([pz if k in neg else nz for k in range(ln)], np.float32)[::-1]      # oldest first: tap k reads x[n - k]
        row = np.where(rng.integers(0, 2, N) == 1, nz, pz).astype(np.float32)
        for start in range(50 + which, N - ln, ln + 3):
            row[start:start + ln] = pat
        rows.append(row)
        row = row.copy(); row[rng.integers(0, N, 12)] = rng.standard_normal(12).astype(np.float32)
        rows.append(row)
    sparse = np.zeros(N, np.float32); idx = rng.integers(0, N, N // 8); sparse[idx] = rng.uniform(-3, 3, idx.size).astype(np.float32)
    sparse[rng.integers(0, N, N // 8)] = nz
    rows.append(sparse)
    rows.append((rng.standard_normal(N) * 1e-39).astype(np.float32))                        # subnormal products
    rows.append((rng.standard_normal(N).astype(np.float32) * np.float32(10.0) ** rng.integers(-30, 30, N).astype(np.float32)).astype(np.float32))
    return rows
