"""Is the process-to-process spread of the row-AND kernel (up to 3 %) memory placement or clock/thermal state?
One process: allocate + fill the C3 index, time K2, free, repeat; then idle and time again."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bigsi_amd import _lib
from bigsi_amd.storage import get_storage, hip_hbm

rows, cols, h = 10_000_000, 100_000, 4
rng = np.random.default_rng(1)
seqs = ["".join(rng.choice(list("ACGT"), size=1000)) for _ in range(256)]


def measure(st, n=20):
    b = st.new_batch(seqs, 31)
    _lib.check(_lib.lib().bigsi_hip_set_profiling(st.handle, 2))
    for _ in range(3):
        b.run(1.0)
    s = _lib.Stats()
    _lib.check(_lib.lib().bigsi_hip_stats(st.handle, _lib.C.byref(s), 1))
    for _ in range(n):
        b.run(1.0)
    _lib.check(_lib.lib().bigsi_hip_stats(st.handle, _lib.C.byref(s), 1))
    b.close()
    return s.and_ms / s.and_launches


for rep in range(4):
    st = get_storage({"storage-engine": "hip-hbm", "k": 31, "m": rows, "h": h, "storage-config": {"name": "pp%d" % rep, "max_cols": cols}})
    st.delete_all()
    for key, v in (("number_of_rows", rows), ("number_of_cols", cols), ("ksi:bloomfilter_size", rows), ("ksi:num_hashes", h)):
        st.set_integer(key, v)
    st.fill_synthetic(1, 0, 2)
    t = [measure(st) for _ in range(3)]
    print("allocation %d: K2 %.4f %.4f %.4f ms" % (rep, *t), flush=True)
    if rep == 3:
        time.sleep(45)
        print("  after 45 s idle: K2 %.4f ms" % measure(st), flush=True)
        for _ in range(40):
            measure(st, 50)
        print("  after ~5 s of continuous K2: %.4f ms" % measure(st), flush=True)
    st.delete_all()
    hip_hbm._RESIDENT.pop("pp%d" % rep).free()
