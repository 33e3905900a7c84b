"""The whole device pipeline on the host: K1 -> clock/time2 -> run-length -> k3_scan/k3_bursts, every kernel
from its DEVICE SOURCE (tests/emu), push by push with all carried state, then the product's host packet
decoders (wm_decoder.c) driven by a few lines of Python that mirror wmbus_collect (skip access codes that
passed while a decoder was busy, continuation bursts for decoders left busy at a push boundary).  The
datagram text must be the oracle's, byte for byte.  No GPU needed; the GPU suite checks the compiled code."""
import ctypes
import os

import numpy as np
import pytest

from cases import flags_to_oracle_opts
import test_burst_need as BN
import test_clock_emulated as CE
import test_k1_emulated as K1
import test_k3_emulated as K3
import test_rla_emulated as RE

emu_k1, emu_clock, emu_rla, emu_k3, emu_need = K1.emu, CE.emu, RE.emu, K3.emu_k3, BN.emu
K1_OD_ARGS = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint, ctypes.c_uint, ctypes.c_uint, ctypes.c_uint64, ctypes.c_uint, ctypes.c_uint,
              ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
F_SHIFT, F_ACCURATE, F_DC, F_T1C1, F_S1, F_RLA, F_T2A = 1, 2, 4, 8, 16, 32, 64


class HostPipeline:
    def __init__(self, libs, d=2, flags=F_ACCURATE | F_T1C1 | F_S1 | F_RLA | F_T2A, seg1=32768, seg0=8192, warm=(12288, 24576), lookback=1024,
                 max_push=1 << 20, polyphase=0, gpu_decode=True, on_demand=None):
        self.polyphase = polyphase
        # RSSI on demand, where the product uses it (wm_api.hip wmbus_open: default switches' kernel, d = 2..5, no debug taps)
        eligible = not polyphase and 2 <= d <= 5 and (flags & (F_ACCURATE | F_T1C1 | F_S1)) == (F_ACCURATE | F_T1C1 | F_S1) and not flags & (128 | 256)
        self.on_demand = eligible if on_demand is None else (on_demand and eligible)
        self.od_pushes = self.od_fallbacks = 0
        self.gpu_decode = gpu_decode                              # as the product: bursts inside the push are decoded by k3_bursts
        self.k1, self.clk, self.rla, self.k3, self.dec = libs
        self.d, self.flags, self.seg1, self.seg0, self.warm, self.lookback = d, flags, seg1, seg0, warm, lookback
        self.stride = (K1.HIST + max_push + K1.SLACK + 255) // 256 * 256
        self.row = np.full(self.stride, 128, np.uint8)
        self.ema = np.zeros(2, np.float32)
        self.clk_carry = np.zeros(2 * self.clk.wm_emu_clock_state_bytes(), np.uint8)
        sb = self.rla.wm_emu_rla_state_bytes()
        self.rla_carry = np.zeros(2 * sb, np.uint8)
        self.rla.wm_emu_rla_reset_state.argtypes = [ctypes.c_void_p]
        for r in range(2):
            self.rla.wm_emu_rla_reset_state(self.rla_carry[r * sb:].ctypes.data)
        self.n0 = 0
        db = self.dec.wm_emu_decoder_bytes()
        self.decs = {}
        self.dec.wm_decoder_init.argtypes = [ctypes.c_void_p, ctypes.c_int]
        self.dec.wm_decoder_chip.argtypes = [ctypes.c_void_p, ctypes.c_uint, ctypes.c_uint]
        self.dec.wm_decoder_chips_owed.argtypes = [ctypes.c_void_p]; self.dec.wm_decoder_chips_owed.restype = ctypes.c_uint
        self.dec.wm_decoder_format.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_uint, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_void_p]
        self.dec.wm_decoder_format.restype = ctypes.c_size_t
        self.dec.wm_packet_format.argtypes = [ctypes.c_int] * 5 + [ctypes.c_uint, ctypes.c_void_p, ctypes.c_uint, ctypes.c_uint, ctypes.c_char_p, ctypes.c_char_p,
                                              ctypes.c_char_p, ctypes.c_size_t]
        self.dec.wm_packet_format.restype = ctypes.c_size_t
        self.dec.wm_packet_crc_ok.argtypes = [ctypes.c_void_p, ctypes.c_uint, ctypes.c_int]
        for ch in range(2):
            for al in range(2):
                buf = ctypes.create_string_buffer(db)
                self.dec.wm_decoder_init(buf, ch)
                self.decs[(ch, al)] = [buf, 0]                  # decoder, chips owed

    def push(self, data):
        nb = data.size
        self.row[K1.HIST:K1.HIST + nb] = data
        n_new, d = nb // 2, self.d
        m0 = self.n0 // d
        M = (self.n0 + n_new) // d - m0
        lines = []
        if M > 0:
            Mcap = max(256, ((M + 975) // 976 * 976 + 255) // 256 * 256)
            dphi = np.zeros((2, Mcap), np.float32); rssi = np.zeros((2, Mcap), np.uint8); err = ctypes.c_uint(0)
            k1_flags = self.flags & (3 | 128 | 256) | F_T1C1 | F_S1
            def k1_full():
                assert self.k1.wm_emu_k1(self.row.ctypes.data, self.stride, 1, d, k1_flags, self.n0, n_new, Mcap, dphi.ctypes.data,
                                         rssi.ctypes.data, self.ema.ctypes.data, ctypes.byref(err), self.polyphase) >= 0 and err.value == 0
            def k1_on_demand(tile_flags):                       # the first pass without the RSSI + the RSSI of the flagged tiles; False: not provable
                self.k1.wm_emu_k1_od.restype = ctypes.c_long
                self.k1.wm_emu_k1_od.argtypes = K1_OD_ARGS
                state = np.zeros(2, np.float32)
                rssi[:] = 0
                r = self.k1.wm_emu_k1_od(self.row.ctypes.data, self.stride, 1, d, k1_flags, self.n0, n_new, Mcap, dphi.ctypes.data, rssi.ctypes.data,
                                         state.ctypes.data, tile_flags.ctypes.data)
                assert r in (0, 1), r
                if r == 0:
                    self.ema[:] = state
                return r == 0
            if self.on_demand:
                assert k1_on_demand(np.zeros((M + 975) // 976, np.uint32)) in (True, False)     # soft symbols (the last tile's RSSI comes with them here)
            else:
                k1_full()
            dphi[:, M:] = 0                                     # rows beyond M are scratch for the framers
            nseg1, cap1 = (M + self.seg1 - 1) // self.seg1, self.seg1 // 4 + 8
            bits = np.zeros((2, Mcap // 32), np.uint32); chips1 = np.zeros((2, nseg1, cap1), np.uint32); counts1 = np.zeros((2, nseg1), np.uint32)
            seen1 = np.zeros((2, nseg1), np.uint32)
            ctypes.c_void_p.in_dll(self.clk, "wm_emu_seen_out").value = seen1.ctypes.data
            r = self.clk.wm_emu_clock(dphi.ctypes.data, 1, M, Mcap, self.flags & (F_DC | F_T1C1 | F_S1 | F_T2A), self.seg1, self.warm[0], self.warm[1], cap1,
                                      self.clk_carry.ctypes.data, bits.ctypes.data, chips1.ctypes.data, counts1.ctypes.data, ctypes.byref(err), None)
            ctypes.c_void_p.in_dll(self.clk, "wm_emu_seen_out").value = None
            assert r >= 0 and err.value == 0
            # the product's layout (wm_api.hip): a primary region of half a chip per sample, then the segment's spill chain
            nseg0, cap0 = (M + self.seg0 - 1) // self.seg0, (self.seg0 // 2 + 8 + 7) // 8 * 8
            chips0 = np.zeros((2, nseg0, cap0), np.uint32); counts0 = np.zeros((2, nseg0), np.uint32); seen0 = np.zeros((2, nseg0), np.uint32)
            arena = np.zeros(1 << 20, np.uint32); chain = np.zeros((2, nseg0, self.rla.wm_emu_spill_levels()), np.uint32); nchain = np.zeros(2 * nseg0 + 1, np.uint32)
            spill = (arena.ctypes.data, arena.size, chain.ctypes.data, nchain.ctypes.data, nchain[2 * nseg0:].ctypes.data)
            self.rla.wm_emu_rla_set_spill(*spill)
            ctypes.c_void_p.in_dll(self.rla, "wm_emu_seen_out").value = seen0.ctypes.data
            r = 0 if not self.flags & F_RLA else self.rla.wm_emu_rla(bits.ctypes.data, 1, M, Mcap, self.flags & (F_T1C1 | F_S1), self.seg0, self.lookback, cap0, self.rla_carry.ctypes.data,
                                    chips0.ctypes.data, counts0.ctypes.data, ctypes.byref(err))
            ctypes.c_void_p.in_dll(self.rla, "wm_emu_seen_out").value = None
            if err.value:
                print("RLA overflow:", dict(M=M, seg0=self.seg0, flags=self.flags, n0=self.n0, d=self.d, counts=counts0.tolist(), cap=cap0))
            assert r >= 0 and err.value == 0, ("run-length framer", r, err.value, dict(M=M, seg0=self.seg0, lookback=self.lookback, flags=self.flags, n0=self.n0, max_count=int(counts0.max()), cap=cap0))
            fr = dict(geo=np.array([M, Mcap, self.flags, m0, self.seg0, self.seg1, nseg0, nseg1, cap0, cap1], np.uint64), chips=(chips0, chips1),
                      counts=(counts0, counts1), seen=(seen0, seen1))
            pending = [self.decs[(ch, al)][1] for al in range(2) for ch in range(2)]          # [algo][chain]
            fr["spill"] = spill
            self.spilled = getattr(self, "spilled", 0) + int(nchain[:2 * nseg0].sum())
            if self.on_demand:
                # k3_spans on the settled chips (the RSSI plays no part in it), the listed tiles' RSSI, then the bursts on rows
                # that hold nothing else; a tile that cannot be proven sends the push through the full pass, as in the product
                K3.bursts_on_host(self.k3, fr, np.zeros_like(rssi), pending, decode=self.gpu_decode)
                self.od_pushes += 1
                if not k1_on_demand(K3.bursts_on_host.last_spans.copy()):
                    self.od_fallbacks += 1
                    k1_full()
                dphi[:, M:] = 0
            if self.gpu_decode:
                hdr, words, pkts, pbytes = K3.bursts_on_host(self.k3, fr, rssi, pending, decode=True)
                lines = self.collect(hdr, words, pkts, pbytes)
            else:
                hdr, words = K3.bursts_on_host(self.k3, fr, rssi, pending)
                lines = self.collect(hdr, words)
        self.row[:K1.HIST] = self.row[nb:nb + K1.HIST].copy()
        self.n0 += n_new
        return lines

    def collect(self, hdr, words, pkts=(), pbytes=None):
        """wm_api.hip: wmbus_collect / decode_stream_range for one capture: candidate telegrams (bursts as chips, and
        telegrams the kernel has already assembled) per (chain, framer) in chip order."""
        out, seq = [], 0
        ents = [((int(h["chain"]), int(h["algo"]), 0 if h["flags"] & 1 else 1, int(h["chip0"])), 1, h) for h in hdr] + \
               [((int(p["chain"]), int(p["algo"]), 1, int(p["chip0"])), 0, p) for p in pkts]
        ents.sort(key=lambda e: e[0])
        line = ctypes.create_string_buffer(1024); ok = ctypes.c_int(0)
        group, next_free = None, 0
        for keyfull, raw, h in ents:
            key = keyfull[:2]
            if key != group:
                group, next_free = key, 0
            dec = self.decs[key]
            tag = b"rla;" if key[1] == 0 else b"t2a;"
            if not raw:
                if dec[1] != 0 or h["chip0"] < next_free:
                    continue                                    # the access code passed while the decoder was busy
                next_free = int(h["chip0"]) + int(h["consumed"])
                if h["status"] != 1:
                    continue
                fl, L = int(h["flags"]), int(h["L"])
                buf = (ctypes.c_uint8 * 296)(*pbytes[int(h["off"]):int(h["off"]) + max(L, 2)].tolist())
                # the kernel's CRC verdict must be the host's (t1_c1_packet_decoder.h:471-536 restated twice)
                assert bool(fl & 8) == bool(self.dec.wm_packet_crc_ok(buf, L, int(bool(fl & 2)))), "device CRC verdict"
                n = self.dec.wm_packet_format(key[0], int(bool(fl & 1)), int(bool(fl & 2)), int(bool(fl & 4)), int(bool(fl & 8)), L, buf,
                                              int(h["pkt_rssi"]), int(h["rssi_now"]), tag, b"TS", line, 1024)
                out.append((int(h["sample"]), key[0], key[1], seq, line.raw[:n].decode()))
                seq += 1
                continue
            cont = bool(h["flags"] & 1)
            if (dec[1] == 0) if cont else (dec[1] != 0 or h["chip0"] < next_free):
                continue                                        # the access code passed while the decoder was busy
            w = words[h["word_off"]:h["word_off"] + h["n_chips"]]
            st, k = (1 if cont else 0), 0
            while k < len(w):
                val, rs = int(w[k]) & 7, (int(w[k]) >> 3) & 0xFF
                if (val & 4) and st == 1:                       # the run-length framer reset itself
                    dec[0][0:2] = bytes(2)                     # wm_decoder_abort: step = 0
                    st = 0
                    break
                st = self.dec.wm_decoder_chip(dec[0], val & 3, rs)
                if st == 2:
                    n = self.dec.wm_decoder_format(dec[0], tag, b"TS", rs, line, 1024, ctypes.byref(ok))
                    out.append((int(h["pos0"]) + (int(w[k]) >> 11), key[0], key[1], seq, line.raw[:n].decode()))
                    seq += 1
                    st = 0
                if st == 0:
                    k += 1
                    break
                k += 1
            next_free = int(h["chip0"]) + k
            cut = st == 1
            assert not (cut and h["n_chips"] != h["avail"]), "burst too short"
            dec[1] = max(1, self.dec.wm_decoder_chips_owed(dec[0])) if cut else 0
        out.sort(key=lambda r: r[:4])
        return [r[4] for r in out]


def run_capture(libs, cu8, pushes, stats=None, **kw):
    p = HostPipeline(libs, **kw)
    text, off, k = [], 0, 0
    total = cu8.size // 4096 * 4096
    while off < total:
        nb = min(pushes[k % len(pushes)], total - off)
        text += p.push(cu8[off:off + nb])
        off += nb; k += 1
    if stats is not None:
        stats["od_pushes"] = stats.get("od_pushes", 0) + p.od_pushes; stats["od_fallbacks"] = stats.get("od_fallbacks", 0) + p.od_fallbacks
    return "".join(text)


@pytest.fixture(scope="module")
def libs(emu_k1, emu_clock, emu_rla, emu_k3, emu_need):
    return emu_k1, emu_clock, emu_rla, emu_k3, emu_need


@pytest.mark.parametrize("pushes", [[1 << 20], [4096 * 37, 4096 * 11], [4096 * 3]])
def test_bundled_capture_through_the_emulated_pipeline(libs, oracle, samples, pushes):
    cu8 = samples["samples2"][: 1 << 20]
    ref = oracle.run(cu8, flags_to_oracle_opts(oracle, ["-v"]))
    assert len(ref["text"].splitlines()) >= 2
    st = {}
    assert run_capture(libs, cu8, pushes, stats=st) == ref["text"]                 # RSSI on demand, as the product runs these switches
    assert st["od_pushes"] > 0 and st["od_fallbacks"] == 0
    assert run_capture(libs, cu8, pushes, on_demand=False) == ref["text"]          # the full pass (contexts with debug taps)
    assert run_capture(libs, cu8, pushes, gpu_decode=False) == ref["text"]        # every burst through the host decoders (cfg.bursts_to_host)


def test_synthetic_captures_through_the_emulated_pipeline(libs, oracle, wm):
    rng = np.random.default_rng(21 + int(os.environ.get("WMBUS_EMU_SEED", "0")))
    n_lines = 0
    for k in range(int(os.environ.get("WMBUS_EMU_N", "4"))):               # more for a bug hunt
        d = int(rng.choice([2, 2, 3, 4, 5]))
        shift, dc, fast = bool(rng.random() < 0.3), bool(rng.random() < 0.3), bool(rng.random() < 0.2)
        kw = dict(t1c1_center_khz=325.0, s1_center_khz=-325.0) if shift else {}
        cu8 = wm.synth_capture(seed=int(rng.integers(1, 1 << 30)), n_samples=int(rng.choice([1 << 17, 1 << 18])), fs_khz=K1.FS[d], kinds=15,
                               frames_per_s=200.0, amplitude=float(rng.choice([25.0, 60.0])), **kw)[0]
        cli = ["-v"] + (["-d", str(d)] if d != 2 else []) + (["-s"] if shift else []) + (["-o"] if dc else []) + (["-a"] if fast else [])
        ref = oracle.run(cu8, flags_to_oracle_opts(oracle, cli))
        pushes = [int(x) * 4096 for x in rng.integers(1, 4 if os.environ.get("WMBUS_EMU_TINY") else 40, 3)]      # TINY: pushes of 4 - 12 KB only
        flags = F_T1C1 | F_S1 | F_RLA | F_T2A | (F_SHIFT if shift else 0) | (F_DC if dc else 0) | (0 if fast else F_ACCURATE)
        got = run_capture(libs, cu8, pushes, d=d, flags=flags, seg1=int(rng.choice([4096, 32768])), seg0=int(rng.choice([1024, 8192])),
                          warm=(int(rng.choice([512, 12288])), int(rng.choice([512, 24576]))))
        assert got == ref["text"], (k, cli, pushes)
        n_lines += len(got.splitlines())
    assert n_lines > 10


def test_gpu_fuzz_configurations_through_the_emulated_pipeline(libs, oracle, wm):
    """The seeded random configurations of tests/test_gpu_fuzz.py (every switch combination, decimations, the
    polyphase pre-filter, push cuts, tunings, silent stretches), first capture of each, on the host emulation."""
    import test_gpu_fuzz as FZ
    from cases import flags_to_kwargs
    n = int(os.environ.get("WMBUS_EMU_N", "6"))
    ran = n_lines = 0
    for k in range(n):
        c = FZ.make_case(k, int(os.environ.get("WMBUS_EMU_SEED", "0")))
        rng = np.random.default_rng(c["seed"])
        kw = dict(seed=c["seed"], n_samples=min(c["n"], 1 << 17), fs_khz=FZ.FS[c["d"]], kinds=15, frames_per_s=90.0, amplitude=c["amp"])
        if c["simultaneous"]:
            kw.update(t1c1_center_khz=325.0, s1_center_khz=-325.0)
        cu8 = wm.synth_capture(**kw)[0]
        if c["silence"]:
            a = int(rng.integers(0, cu8.size // 2)) & ~1
            cu8[a:a + int(rng.integers(4096, cu8.size // 3))] = int(rng.choice([127, 128]))
        oo = flags_to_oracle_opts(oracle, c["flags"])
        oo.prefilter = c["prefilter"]
        ref = oracle.run(cu8, oo)
        pk = flags_to_kwargs(c["flags"])
        flags = ((F_SHIFT if pk.get("simultaneous") else 0) | (F_ACCURATE if pk.get("accurate_atan", True) else 0) | (F_DC if pk.get("remove_dc") else 0) |
                 (F_T1C1 if pk.get("t1c1", True) else 0) | (F_S1 if pk.get("s1", True) else 0) | (F_RLA if pk.get("rla", True) else 0) |
                 (F_T2A if pk.get("time2", True) else 0))
        t = c["tune"]
        got = run_capture(libs, cu8, [min(c["push"], 1 << 20)], d=c["d"], flags=flags, polyphase=c["prefilter"], seg1=t.get("seg_len", 32768),
                          seg0=t.get("rla_seg_len", 8192), warm=(t.get("warmup_t1c1", 12288), t.get("warmup_s1", 24576)), lookback=t.get("rla_lookback", 1024))
        want = ref["text"] if "-v" in c["flags"] else ref["text"]
        if "-v" not in c["flags"]:                           # the emulated collect always tags the framer: strip the tags
            got = "".join(ln.split(";", 1)[1] + "\n" for ln in got.splitlines())
        assert got == want, (k, c["flags"], c["tune"])
        ran += 1; n_lines += len(got.splitlines())
    assert ran >= n - 1 and (n < 6 or n_lines > 0)


@pytest.mark.parametrize("atan_mode", [1, 2])
def test_atan2_approximation_options_through_the_emulated_pipeline(libs, oracle, wm, atan_mode):
    cu8 = wm.synth_capture(seed=4242 + atan_mode, n_samples=1 << 18, kinds=15, frames_per_s=200.0, amplitude=60.0)[0]
    oo = flags_to_oracle_opts(oracle, ["-v"])
    oo.atan_mode = atan_mode
    ref = oracle.run(cu8, oo)
    got = run_capture(libs, cu8, [4096 * 21, 4096 * 4], flags=F_ACCURATE | F_T1C1 | F_S1 | F_RLA | F_T2A | (128 * atan_mode))
    assert got == ref["text"] and len(got.splitlines()) > 4
