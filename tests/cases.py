"""Shared test-case tables: reference switches x captures."""

S2 = "rtlsdr_868.950M_1M6_samples2.cu8"
I48 = "rtlsdr_868.625M_2M4_issue48.cu8"

# (capture, reference argv)
BUNDLED_CASES = [
    (S2, ["-v"]), (S2, []), (S2, ["-r", "0", "-v"]), (S2, ["-t", "0", "-v"]), (S2, ["-p", "S", "-v"]),
    (S2, ["-p", "T", "-v"]), (S2, ["-o", "-v"]), (S2, ["-a", "-v"]), (S2, ["-s", "-v"]), (S2, ["-d", "3", "-v"]),
    (S2, ["-o", "-a", "-v"]), (S2, ["-d", "1", "-v"]),
    (I48, ["-d", "3", "-s", "-o", "-v"]), (I48, ["-d", "3", "-s", "-v"]), (I48, ["-d", "3", "-s", "-o", "-a", "-v"]),
    (I48, ["-d", "3", "-v"]),
]

T1, C1A, C1B, S1 = 1, 2, 4, 8
ALL = T1 | C1A | C1B | S1

# synthetic captures: a pure function of (seed, config); flags = reference argv
SYNTH_CASES = [
    dict(id="t1c1_1600", seed=101, n=1 << 20, fs=1600, kinds=T1 | C1A | C1B, rate=60.0, amp=60.0, flags=["-v"]),
    dict(id="all_1600", seed=102, n=1 << 20, fs=1600, kinds=ALL, rate=60.0, amp=60.0, flags=["-v"]),
    dict(id="all_1600_weak", seed=103, n=1 << 20, fs=1600, kinds=ALL, rate=60.0, amp=9.0, flags=["-v"]),
    dict(id="all_1600_o", seed=104, n=1 << 20, fs=1600, kinds=ALL, rate=60.0, amp=20.0, flags=["-o", "-v"]),
    dict(id="all_1600_a", seed=105, n=1 << 20, fs=1600, kinds=ALL, rate=60.0, amp=20.0, flags=["-a", "-v"]),
    dict(id="all_2400_d3", seed=106, n=1 << 20, fs=2400, kinds=ALL, rate=60.0, amp=60.0, flags=["-d", "3", "-v"]),
    dict(id="all_4000_d5_s", seed=107, n=1 << 21, fs=4000, kinds=ALL, rate=60.0, amp=60.0, tc=325.0, sc=-325.0,
         flags=["-d", "5", "-s", "-v"]),
    dict(id="all_1600_s", seed=108, n=1 << 20, fs=1600, kinds=ALL, rate=60.0, amp=60.0, tc=325.0, sc=-325.0,
         flags=["-s", "-v"]),
    dict(id="noise_only", seed=109, n=1 << 19, fs=1600, kinds=0, rate=0.0, amp=0.0, flags=["-v"]),
    dict(id="long_frames", seed=110, n=1 << 20, fs=1600, kinds=ALL, rate=30.0, amp=60.0, lmin=150, lmax=250, flags=["-v"]),
    dict(id="t1_only_r0", seed=111, n=1 << 19, fs=1600, kinds=T1, rate=80.0, amp=40.0, flags=["-r", "0", "-v"]),
    dict(id="all_4800_d6_s", seed=113, n=1 << 21, fs=4800, kinds=ALL, rate=60.0, amp=60.0, tc=325.0, sc=-325.0,
         flags=["-d", "6", "-s", "-v"]),                     # a rate outside 2..5: the run-time-decimation kernel
    dict(id="t1c1_8000_d10", seed=114, n=1 << 21, fs=8000, kinds=T1 | C1A | C1B, rate=60.0, amp=60.0, flags=["-d", "10", "-o", "-v"]),
    dict(id="s1_only_t0", seed=112, n=1 << 19, fs=1600, kinds=S1, rate=40.0, amp=40.0, flags=["-t", "0", "-p", "T", "-v"]),
]


def synth_case_capture(wm, case):
    return wm.synth_capture(seed=case["seed"], n_samples=case["n"], fs_khz=case["fs"], kinds=case["kinds"],
                            frames_per_s=case["rate"], amplitude=case["amp"], t1c1_center_khz=case.get("tc", 0.0),
                            s1_center_khz=case.get("sc", 0.0), l_min=case.get("lmin", 10), l_max=case.get("lmax", 60))


def flags_to_kwargs(flags):
    """Reference argv -> rtl-wmbus_amd.Receiver keyword arguments."""
    kw = dict(show_algorithm=False)
    it = iter(flags)
    for f in it:
        if f == "-v": kw["show_algorithm"] = True
        elif f == "-o": kw["remove_dc"] = True
        elif f == "-a": kw["accurate_atan"] = False
        elif f == "-s": kw["simultaneous"] = True
        elif f == "-d": kw["decimation"] = int(next(it))
        elif f == "-r": next(it); kw["rla"] = False
        elif f == "-t": next(it); kw["time2"] = False
        elif f == "-p":
            v = next(it)
            if v in "Tt": kw["t1c1"] = False
            else: kw["s1"] = False
    return kw


def flags_to_oracle_opts(O, flags):
    kw = flags_to_kwargs(flags)
    return O.make_opts(decimation=kw.get("decimation", 2), simultaneous=int(kw.get("simultaneous", False)),
                       accurate_atan=int(kw.get("accurate_atan", True)), remove_dc=int(kw.get("remove_dc", False)),
                       t1c1=int(kw.get("t1c1", True)), s1=int(kw.get("s1", True)), rla=int(kw.get("rla", True)),
                       time2=int(kw.get("time2", True)), show_algorithm=int(kw["show_algorithm"]))
