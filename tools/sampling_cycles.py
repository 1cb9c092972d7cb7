import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, "willow-inference-server_amd")
from wis_hip import _lib, ctranslate2 as ct2
model = ct2.Whisper("synthetic:large", max_batch=1, max_beam=5)
mel = np.ascontiguousarray(np.load("tests/golden/logmel_3sec.npz")["mel"][None].astype(np.float32))
P = [50258, 50259, 50359, 50363]
for _ in range(2):
    model.generate(ct2.StorageView.from_array(mel), [P], beam_size=5, fixed_new_tokens=16)
out = np.zeros((2, 16), np.uint64)
_lib.check(_lib.load().wis_debug_sampling_cycles(model._replicas[0].handle, out.ctypes.data_as(C.POINTER(C.c_uint64))))
for r in (0, 1):
    st = [int(v) for v in out[r] if v]
    if st:
        print("row", r, "phase cycles:", [st[i + 1] - st[i] for i in range(len(st) - 1)], "total", st[-1] - st[0])
