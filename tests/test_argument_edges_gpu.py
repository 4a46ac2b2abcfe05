"""Constructor and argument edge cases through the Python mirror of the C ABI: bad geometry or devices are refused
with a message (jsp_last_error), degenerate-but-legal inputs behave as the reference does, nothing crashes."""
import numpy as np
import pytest

from jsplayer_amd import CodecError, DecoderState, MSVideo1_16bit, MSVideo1_8bit, ScreenPressor

pytestmark = pytest.mark.gpu


def dev(n):
    import torch
    return torch.zeros(n, dtype=torch.int32, device="cuda")


@pytest.mark.parametrize("make,msg", [
    (lambda: MSVideo1_16bit(0, 0), "bad frame size"),
    (lambda: MSVideo1_16bit(-4, 8), "bad frame size"),
    (lambda: ScreenPressor(0, 0, 24), "bad frame size"),
    (lambda: ScreenPressor(100000, 2, 24), "wider than 8192"),
    (lambda: MSVideo1_16bit(16, 16, device=99), "device_id out of range"),
    (lambda: MSVideo1_16bit(1 << 20, 1 << 20), "frame too large"),
])
def test_refused_constructions(make, msg):
    with pytest.raises(CodecError, match=msg):
        make()


def test_degenerate_but_legal_inputs():
    # frames smaller than one 4x4 block: no block is ever painted, the calls still answer as the reference does
    c = MSVideo1_16bit(3, 3)
    c.Preinit(36)
    d = dev(9)
    assert c.DecompressI(b"", d) == DecoderState.zero_state
    assert c.DecompressI(b"\x01\x02\x03\x04", d) == DecoderState.zero_state
    assert not d.cpu().numpy().any()
    with pytest.raises(CodecError, match="width\\*height"):
        c.DecompressI(b"", dev(4))
    # an 8-bit codec with a missing / short palette (the reference reads what is there)
    for pal in (b"", b"\x01\x02\x03"):
        p = MSVideo1_8bit(16, 16, pal)
        p.Preinit(0)
        assert p.DecompressI(b"\x05\x81" * 16, dev(256)) == DecoderState.zero_state
    # ScreenPressor: inter frame before any key frame = "no changes"; empty key frame = error_occured
    s = ScreenPressor(16, 16, 24)
    s.Preinit(-5)
    res = s.DecompressP(b"\x01\x02", dev(256))
    assert res.data_pnt is None and res.significant_changes is False
    assert s.DecompressI(b"", dev(256)) == DecoderState.error_occured
    assert ScreenPressor(16, 16, 7).DecompressI(b"\x12", dev(256)) in (DecoderState.zero_state, DecoderState.error_occured)
