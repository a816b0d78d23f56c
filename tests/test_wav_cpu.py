"""WAV writer parity with the reference's InitWaveOut / CloseWaveOut (src/rtl_fm_player.c:1259-1328).

Golden data: tests/golden/wav_header_{stereo,mono}.bin are the reference's two header tables
(include/rtl_fm_player.h:216-253), extracted by tests/golden/make_wav_fixture.py."""
import ctypes as C
import os
import struct

import numpy as np
import pytest

import rtl_fm_player_amd as R

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def L():
    R.build_library()
    lib = R.lib()
    lib.fmd_wav_header.argtypes = [C.c_int, C.c_char_p]
    lib.fmd_wav_open.argtypes = [C.POINTER(C.c_void_p), C.c_char_p, C.c_int]
    lib.fmd_wav_write.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    lib.fmd_wav_close.argtypes = [C.c_void_p]
    return lib


@pytest.mark.parametrize("mode,name", [(2, "wav_header_stereo.bin"), (1, "wav_header_mono.bin"), (0, "wav_header_mono.bin")])
def test_header_equals_reference_table(L, mode, name):
    want = open(os.path.join(GOLD, name), "rb").read()
    buf = C.create_string_buffer(260)
    assert L.fmd_wav_header(mode, buf) == 0
    assert buf.raw == want


@pytest.mark.parametrize("mode", [1, 2])
def test_file_sizes_are_patched_like_closewaveout(L, tmp_path, mode):
    pcm = (np.arange(10001) % 2000 - 1000).astype(np.int16)
    path = str(tmp_path / "o.wav")
    h = C.c_void_p()
    assert L.fmd_wav_open(C.byref(h), path.encode(), mode) == 0
    assert L.fmd_wav_write(h, pcm.ctypes.data, pcm.size) == 0
    assert L.fmd_wav_write(h, pcm.ctypes.data, 5) == 0
    assert L.fmd_wav_close(h) == 0
    raw = open(path, "rb").read()
    size = 260 + 2 * (pcm.size + 5)
    assert len(raw) == size
    gold = open(os.path.join(GOLD, "wav_header_stereo.bin" if mode == 2 else "wav_header_mono.bin"), "rb").read()
    # everything but the two patched size fields is the reference header
    assert raw[:4] == gold[:4] and raw[8:40] == gold[8:40] and raw[44:260] == gold[44:260]
    assert struct.unpack("<I", raw[4:8])[0] == size - 8          # ftell - 8   (:1267-1272)
    assert struct.unpack("<I", raw[40:44])[0] == size - 44       # ftell - 44  (:1273-1278)
    assert np.array_equal(np.frombuffer(raw[260:260 + 2 * pcm.size], dtype=np.int16), pcm)
