cd $GRAFT_REPO_ROOT
python - <<'PY'
from jsplayer_amd import avi, streamgen as sg
fr, keys, _ = sg.msv1_clip(1, 320, 240, 6)
open('/tmp/a.avi','wb').write(avi.write_avi(320,240,fr,fourcc=b'CRAM',bpp=16,palette=None,key_flags=keys))
fr, keys, _ = sg.msv1_clip(2, 320, 240, 4)
open('/tmp/b.avi','wb').write(avi.write_avi(320,240,fr,fourcc=b'CRAM',bpp=16,palette=None,key_flags=keys))
PY
./examples/jsp_play /tmp/a.avi --pipelined --quiet --depth 8 --streams 1 --repeat 2 --device 0; echo rc=$?
./examples/jsp_play /tmp/a.avi,/tmp/b.avi --pipelined --quiet --depth 8 --streams 3 --repeat 2 --device 0; echo rc=$?
./examples/jsp_play /tmp/a.avi --pipelined --quiet --depth 8 --streams 1 --repeat 2; echo rc=$?
