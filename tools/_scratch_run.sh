cd /root/repo
MI_LIB_OVERRIDE=.ab_r6/libmi_timeline.so python tools/tile_timeline.py
