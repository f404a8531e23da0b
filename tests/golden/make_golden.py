#!/usr/bin/env python3
"""Regenerates tests/golden/ from the reference checkout (run in the build
container where /root/reference is mounted; the GPU box only sees the copies).

Copies the reference's own decode fixtures (captures, their sensor metadata and
md5 digests -- tests/pcaps/, used by python/tests/test_core.py:272-279) and
writes the per-field hash snapshots of tests/frame_batcher_test.cpp:553-595
into snapshot_hashes.json.
"""
import json
import os
import re
import shutil

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
CAPTURES = [
    "OS-0-128-U1_v2.3.0_1024x10",                       # RNG15_RFL8_NIR8 (low bandwidth)
    "OS-0-32-U1_v2.2.0_1024x10",                        # RNG19_RFL8_SIG16_NIR16_DUAL
    "OS-1-128_767798045_1024x10_20230712_120049",       # FUSA_RNG15_RFL8_NIR8_DUAL
    "OS-2-128-U1_v2.3.0_1024x10",                       # RNG19_RFL8_SIG16_NIR16
    "OS-2-32-U0_v2.0.0_1024x10",                        # LEGACY
    "OS-1-32-G_v2.1.1_1024x10",                         # LEGACY
    "crc_test",                                         # RNG15_RFL8_NIR8 512x10, 34 packets w/ CRC64 footers
]


# OSF fixtures of the reference (tests/osfs/): PNG-encoded (16-bit gray, RGBA), PNG with 8-bit planes,
# ZPNG-encoded dual-return
OSFS = ["OS-1-128_v2.3.0_1024x10_lb_n3.osf", "OS-0-128_v3.0.1_1024x10_20241017_141645.osf", "single_scan_016.osf",
        "pose_delta_1_128.osf"]   # the last one: dual return 128 x 2048 with 1-D custom fields (IMU_*, POSITION_TIMESTAMP)


def osf_goldens():
    """Copies the OSF fixtures and pins the first one on the capture the reference wrote it from:
    tests/pcaps/OS-1-128_v2.3.0_1024x10_lb_n3.pcap is decoded here with the
    packet oracle (itself pinned on the reference's digests / snapshot hashes) and the sha256 of every
    plane and column header of its three frames goes to osf/lb_n3_pcap_planes.json."""
    import ctypes as C
    import hashlib
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from oracle import oracle as O
    dst = os.path.join(HERE, "osf")
    os.makedirs(dst, exist_ok=True)
    for name in OSFS:
        shutil.copyfile(os.path.join(REF, "tests", "osfs", name), os.path.join(dst, name))
    cal = O.calib_from_json(os.path.join(REF, "tests", "pcaps", "OS-1-128_v2.3.0_1024x10.json"))
    pf = cal.packet_format()
    pk = O.lidar_packets_from_pcap(os.path.join(REF, "tests", "pcaps", "OS-1-128_v2.3.0_1024x10_lb_n3.pcap"), pf)
    fids = [O.lib().ora_frame_id(C.byref(pf), p.ctypes.data) for p in pk]
    out = {}
    for fid in sorted(set(fids)):
        sel = [i for i, x in enumerate(fids) if x == fid]
        fr = O.Frame.for_profile(cal.profile, cal.h, cal.w, cal.cpp, with_window=False)
        fr.fill(0)
        b = O.Batcher(pf, init_id=O.lib().ora_init_id(C.byref(pf), pk[sel[0]].ctypes.data), expected_packets=len(sel))
        done = False
        for i in sel:
            done = b.batch(pk[i], 1 + i, fr)
        if not done:
            b.finalize(fr)
        sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
        entry = {"packets": len(sel)}
        for n in ("RANGE", "REFLECTIVITY", "NEAR_IR"):
            entry[n] = sha(fr.plane(n).astype(np.uint64))     # value digests: the OSF widens REFLECTIVITY to u16
        entry["timestamp"], entry["status"], entry["measurement_id"] = sha(fr.timestamp), sha(fr.status), sha(fr.measurement_id)
        out[str(fid)] = entry
    with open(os.path.join(dst, "lb_n3_pcap_planes.json"), "w") as f:
        json.dump({"source": "tests/pcaps/OS-1-128_v2.3.0_1024x10_lb_n3.pcap decoded by oracle/ (sha256 of the "
                             "uint64-widened planes, raw headers)", "frames": out}, f, indent=1, sort_keys=True)
    print("osf goldens:", {k: v["packets"] for k, v in out.items()})


def main():
    osf_goldens()
    dst = os.path.join(HERE, "pcaps")
    os.makedirs(dst, exist_ok=True)
    for base in CAPTURES:
        for ext in (".pcap", ".json", "_digest.json"):
            src = os.path.join(REF, "tests", "pcaps", base + ext)
            if os.path.exists(src):
                shutil.copyfile(src, os.path.join(dst, base + ext))

    # the pair the reference's parsing_benchmark_test.cpp reads besides the captures above (its names do not follow the
    # <base>.pcap / <base>.json pattern)
    # ... and the windowed captures of the reference's python/tests/test_batching.py::test_early_release
    for name in ("OS-1-128_v2.3.0_1024x10_lb_n3.pcap", "OS-1-128_v2.3.0_1024x10.json", "windowed_frame1.pcap",
                 "windowed_frame1_0.json", "windowed_frame2.pcap", "windowed_frame2_0.json"):
        shutil.copyfile(os.path.join(REF, "tests", "pcaps", name), os.path.join(dst, name))

    # snapshot hashes from the C++ test source
    text = open(os.path.join(REF, "tests", "frame_batcher_test.cpp")).read()
    block = text[text.index("FrameBatcherSnapshots"):text.index("struct matrix_hash")]
    out = {}
    for m in re.finditer(r'snapshot_param\{"([^"]+)\.pcap",\s*"[^"]+",\s*\{(.*?)\}\}\}', block, re.S):
        fields = {}
        for fm in re.finditer(r"ChanField::(\w+),\s*(0x[0-9a-fA-F]+|\d+)U?", m.group(2)):
            fields[fm.group(1)] = int(fm.group(2), 0)
        out[m.group(1)] = fields
    with open(os.path.join(HERE, "snapshot_hashes.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print({k: len(v) for k, v in out.items()})


if __name__ == "__main__":
    main()
