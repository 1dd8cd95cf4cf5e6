"""GPU parity of the API-edge resampler with the reference's own function (api/openai_realtime/utils.py:5-12): int16 in -> int16 out,
BIT-EXACT for the client rates the realtime API uses (24 kHz <-> 16 kHz) and for 48 / 8 / 44.1 kHz inputs, ragged batches, empty and
one-sample chunks; plus the chunker / base64 bookkeeping."""

import base64

import numpy as np
import pytest

from oracle.audio_edge import reference_resample
from speech_to_speech_amd.audio_edge import EdgeResampler, PcmChunker, encode_audio_chunks

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("from_rate,to_rate", [(24000, 16000), (16000, 24000), (48000, 16000), (8000, 16000), (44100, 16000), (16000, 16000)])
def test_edge_resampler_is_bit_exact_with_the_reference_resample(from_rate, to_rate):
    rng = np.random.default_rng(from_rate + to_rate)
    lens = [4800, 1, 0, 959, 12345, 480]
    chunks = [(rng.standard_normal(n) * 9000).clip(-32768, 32767).astype(np.int16) for n in lens]
    chunks[3][:] = 32767                       # saturating input exercises the clip
    rs = EdgeResampler()
    got = rs.resample_batch([c.tobytes() for c in chunks], from_rate, to_rate)
    for c, g in zip(chunks, got):
        want = reference_resample(c.tobytes(), from_rate, to_rate)
        assert len(g) == len(want)
        a, b = np.frombuffer(g, np.int16).astype(np.int32), np.frombuffer(want, np.int16).astype(np.int32)
        assert np.array_equal(a, b), f"{from_rate}->{to_rate}: {np.abs(a - b).max()} LSB off in {(a != b).sum()} of {len(a)} samples"


def test_chunker_and_base64_bookkeeping():
    ch = PcmChunker()
    a = np.arange(700, dtype=np.int16).tobytes()
    out = ch.append(a)
    assert len(out) == 1 and len(out[0]) == 1024 and len(ch.remainder) == 2 * (700 - 512) and ch.has_data
    out = ch.append(np.arange(400, dtype=np.int16).tobytes())
    assert len(out) == 1 and np.frombuffer(out[0], np.int16)[0] == 512 and len(ch.remainder) == 2 * (700 + 400 - 1024)
    enc = encode_audio_chunks([b"\x01\x02", b""])
    assert enc == [base64.b64encode(b"\x01\x02").decode("ascii"), ""]
