"""jsplayer_amd — MI355X-native block-video decode path behind jsplayer's IVideoCodec.

Only the decode hot path of thedeemon/jsplayer is here (MSVideo1 + ScreenPressor): host stage in
C++ (parse / entropy -> descriptor tables), reconstruction in hand-written HIP for gfx950, exposed
through the C ABI of include/jsplayer_amd.h.  This package is the Python mirror of the reference's
plugin interface over that ABI.
"""
from .codec import (CodecError, DecoderState, FramePool, HostBuffer, MSVideo1_16bit, MSVideo1_8bit, PFrameResult, ScreenPressor,
                    StagedBatch)

__all__ = ["CodecError", "DecoderState", "FramePool", "HostBuffer", "MSVideo1_16bit", "MSVideo1_8bit", "PFrameResult", "ScreenPressor",
           "StagedBatch"]
