"""Run as a script by test_screenpressor_gpu.py with JSP_SP_IFRAME_KERNEL / JSP_SP_TILE_PPL / JSP_SP_GROUP_KERNEL set:
the kernel variants those knobs select (kept for A/B measurements) decode a few clips bit-exactly too."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)


def main():
    import torch
    from jsplayer_amd import ScreenPressor
    from jsplayer_amd import streamgen as sg
    checked = 0
    for (w, h) in [(64, 48), (320, 240), (2056, 40), (1024, 64)]:
        chunks, keys, frames = sg.sp_clip(994, w, h, 8, version=4, key_every=4, unchanged_at=(2,), flat_at=(6,))
        for band_rows in ("auto", "0", "7"):
            gpu = ScreenPressor(w, h, 24)
            gpu.Preinit(36)
            gpu.set_option("sp_band_rows", band_rows)
            dsts = [torch.full((w * h,), -1, dtype=torch.int32, device="cuda") for _ in chunks]
            st = gpu.stage_batch(chunks, dsts, is_key=keys)
            st.decode()
            gpu.sync()
            _, adopted, _ = st.results()
            for i, (d, img) in enumerate(zip(dsts, frames)):
                if adopted[i]:
                    assert np.array_equal(d.cpu().numpy().view(np.uint32), img), (w, h, band_rows, i)
                    checked += 1
            st.close()
            gpu.StopAndClean()
    print("variant ok", checked)


if __name__ == "__main__":
    main()
