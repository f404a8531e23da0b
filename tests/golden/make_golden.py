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


def main():
    dst = os.path.join(HERE, "pcaps")
    os.makedirs(dst, exist_ok=True)
    for base in CAPTURES:
        for ext in (".pcap", ".json", "_digest.json"):
            src = os.path.join(REF, "tests", "pcaps", base + ext)
            if os.path.exists(src):
                shutil.copyfile(src, os.path.join(dst, base + ext))

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
