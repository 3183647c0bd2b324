"""Per-wave view of the split P2 tile (diagnostic build -DDSAT_STAMP_ROTATE: workgroup b's stamps come from wave b mod 16):
cycles from the tile's start to the end of each phase, by wave.  DSAT_LIB=tools/libs/libdsat_rot.so python tools/stamps_waves.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tools import stamps as S
st, meta = S.collect(2, 32)
t = st["round_p2"].astype(np.float64)
names = ["state rows", "finish (gather)", "layer 0", "layer 1 k-loop", "barrier after outputs", "end"]
cols = [1, 2, 5, 6, 3, 4]
print("wave " + " ".join("%22s" % n for n in names))
for w in range(16):
    rows = t[w::16]
    print("%4d " % w + " ".join("%22.0f" % np.median(rows[:, c] - rows[:, 0]) for c in cols))
