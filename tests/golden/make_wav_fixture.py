#!/usr/bin/env python3
"""Generates tests/golden/wav_header_{stereo,mono}.bin from the reference.

Runs only where /root/reference exists.  It parses the two 260-byte header tables the
reference writes at the start of every WAV file (include/rtl_fm_player.h:216-253) and stores
their bytes: data the parity test of fmd_wav_* compares against.
"""
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
src = open("/root/reference/include/rtl_fm_player.h").read()
for name, out in (("_WAVHeaderStereo", "wav_header_stereo.bin"), ("_WAVHeaderMono", "wav_header_mono.bin")):
    body = re.search(name + r"\[\]\s*=\s*\{(.*?)\};", src, re.S).group(1)
    data = bytes(int(x, 16) for x in re.findall(r"0x([0-9A-Fa-f]{2})", body))
    assert len(data) == 260, len(data)   # 16 rows of 16 + 4 (SURVEY.md says 276: miscounted)
    open(os.path.join(HERE, out), "wb").write(data)
    print(out, len(data), data[:44].hex())
